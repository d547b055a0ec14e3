mkdir -p gpurun_out/r2k
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_streaming_gpu.py tests/test_checkpoint_gpu.py -m gpu -x -q > gpurun_out/r2k/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2k/pytest.log; tail -4 gpurun_out/r2k/pytest.log
timeout -k 10 300 python tools/kbench.py --steps 50 --warmup 50 --configs c3,c3_f32 > gpurun_out/r2k/kbench.jsonl 2>/dev/null; python - <<'PY'
import json
for l in open('gpurun_out/r2k/kbench.jsonl'):
    d=json.loads(l); print(d['config'], d['us_per_launch'], round(d['hbm_frac'],4), d['mean_gain_iters'], d['rms_vs_truth'])
PY
timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r2k/bench.log 2>&1; grep '^{"metric"' gpurun_out/r2k/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('bench', d['value'], d['ms_per_step'], r['frac'], r['kernel_us_per_launch'], d['per_epoch_launch']['kernel_us_per_launch'])"
