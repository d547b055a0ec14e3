set -x
mkdir -p gpurun_out/r2a
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log; tail -5 gpurun_out/r2a/pytest.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r2a/bench_driver.log 2>&1 && tail -c 600 gpurun_out/r2a/bench_driver.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r2a/bench_default.log 2>&1 && tail -c 300 gpurun_out/r2a/bench_default.log
KFPOS_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --config c4 --total-tags 262144 --steps 40 --warmup 10 --gather epoch > gpurun_out/r2a/bench_c4_gloo2.log 2>&1; tail -c 400 gpurun_out/r2a/bench_c4_gloo2.log
timeout -k 10 600 python bench.py --config c4 --steps 40 --warmup 10 > gpurun_out/r2a/bench_c4_n1.log 2>&1; tail -c 600 gpurun_out/r2a/bench_c4_n1.log
