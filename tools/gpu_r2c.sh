set -x
mkdir -p gpurun_out/r2c
timeout -k 10 600 python -m pytest tests/test_streaming_gpu.py tests/test_gpu_parity.py tests/test_planar_gpu.py tests/test_ingest_batched.py tests/test_adaptor_replay.py tests/test_planar_adaptor.py tests/test_publish_messages.py tests/test_checkpoint_gpu.py -m gpu -x -q > gpurun_out/r2c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c/pytest.log; tail -15 gpurun_out/r2c/pytest.log
timeout -k 10 300 python tools/hostbench.py > gpurun_out/r2c/hostbench.json 2> gpurun_out/r2c/hostbench.err; cat gpurun_out/r2c/hostbench.json
g++ -O2 -std=c++17 -I include -I roskfpos_amd/csrc -o gpurun_out/r2c/adaptor_latency tools/adaptor_latency.cpp -L roskfpos_amd/csrc -lkfpos_hip -Wl,-rpath,$PWD/roskfpos_amd/csrc && timeout -k 10 120 gpurun_out/r2c/adaptor_latency 500 > gpurun_out/r2c/adaptor_latency.jsonl 2>&1; cat gpurun_out/r2c/adaptor_latency.jsonl
KFPOS_SMALL_BANK_ELEMS=0 timeout -k 10 120 gpurun_out/r2c/adaptor_latency 300 > gpurun_out/r2c/adaptor_latency_staged.jsonl 2>&1; cat gpurun_out/r2c/adaptor_latency_staged.jsonl
timeout -k 10 300 python tools/exp/halves.py > gpurun_out/r2c/halves.json 2> gpurun_out/r2c/halves.err; cat gpurun_out/r2c/halves.json
rm -f gpurun_out/r2c/adaptor_latency
