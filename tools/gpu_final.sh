# end-of-round validation: the GPU suite, smoke(), the driver's bench command, the default bench, and two-rank
# rehearsals of the N > 1 path (gloo, both ranks on the one card) for both bench configurations
set -e
mkdir -p gpurun_out/final
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/bench_driver.log 2>&1; grep '^{"metric"' gpurun_out/final/bench_driver.log | cut -c1-400
timeout -k 10 400 python bench.py > gpurun_out/final/bench_default.log 2>&1; grep '^{"metric"' gpurun_out/final/bench_default.log | cut -c1-200
KFPOS_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/final/bench_c3_gloo2.log 2>&1; grep '^{"metric"' gpurun_out/final/bench_c3_gloo2.log | cut -c1-300
KFPOS_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --config c4 --total-tags 262144 --steps 20 --warmup 5 --gather epoch > gpurun_out/final/bench_c4_gloo2.log 2>&1; grep '^{"metric"' gpurun_out/final/bench_c4_gloo2.log | cut -c1-300
