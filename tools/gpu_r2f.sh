set -x
mkdir -p gpurun_out/r2f
timeout -k 10 600 python -m pytest tests/test_streaming_gpu.py tests/test_ingest_batched.py tests/test_planar_adaptor.py tests/test_adaptor_replay.py -m gpu -x -q > gpurun_out/r2f/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2f/pytest.log; tail -5 gpurun_out/r2f/pytest.log
timeout -k 10 300 python tools/hostbench.py > gpurun_out/r2f/hostbench.json 2> gpurun_out/r2f/hostbench.err; cat gpurun_out/r2f/hostbench.json
g++ -O2 -std=c++17 -I include -I roskfpos_amd/csrc -o gpurun_out/r2f/ingestbench tools/ingestbench.cpp -L roskfpos_amd/csrc -lkfpos_hip -Wl,-rpath,$PWD/roskfpos_amd/csrc && (timeout -k 10 200 gpurun_out/r2f/ingestbench 65536 12 0; timeout -k 10 200 gpurun_out/r2f/ingestbench 65536 12 1) > gpurun_out/r2f/ingestbench.jsonl 2>&1; cat gpurun_out/r2f/ingestbench.jsonl
rm -f gpurun_out/r2f/ingestbench
