mkdir -p gpurun_out/r2w
timeout -k 10 600 python tools/kbench.py --steps 50 --warmup 50 --configs c2,toa6_65k,c4shard,c3,c3_f32,c5,iw8,ml,planar,planar_sens,t1k,t8k > gpurun_out/r2w/kbench_all.jsonl 2>/dev/null
python - <<'PY'
import json
for l in open('gpurun_out/r2w/kbench_all.jsonl'):
    d=json.loads(l); print(d['config'], d['tags'], d['us_per_launch'], 'steps/s %.3g'%d['tag_steps_per_s'], 'frac', round(d['hbm_frac'],4), 'pose_us', d['get_pose_us'], 'iters', round(d['mean_gain_iters'],2), round(d['mean_ml_iters'],2))
PY
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r2w/bench_driver.log 2>&1; grep '^{"metric"' gpurun_out/r2w/bench_driver.log | cut -c1-200
timeout -k 10 300 python bench.py > gpurun_out/r2w/bench_default.log 2>&1; grep '^{"metric"' gpurun_out/r2w/bench_default.log | cut -c1-200
timeout -k 10 600 python bench.py --config c4 --steps 40 --warmup 10 > gpurun_out/r2w/bench_c4_n1.log 2>&1; grep '^{"metric"' gpurun_out/r2w/bench_c4_n1.log | cut -c1-200
KFPOS_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 20 --warmup 5 --no-secondary > gpurun_out/r2w/bench_c3_gloo2.log 2>&1; grep '^{"metric"' gpurun_out/r2w/bench_c3_gloo2.log | cut -c1-200
