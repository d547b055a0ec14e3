set -x
mkdir -p gpurun_out/r2d
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2d/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2d/pytest.log; tail -12 gpurun_out/r2d/pytest.log
timeout -k 10 120 python tools/exp/pcie_probe.py > gpurun_out/r2d/pcie.json 2>/dev/null; cat gpurun_out/r2d/pcie.json
timeout -k 10 300 python tools/hostbench.py > gpurun_out/r2d/hostbench.json 2> gpurun_out/r2d/hostbench.err; cat gpurun_out/r2d/hostbench.json
g++ -O2 -std=c++17 -I include -I roskfpos_amd/csrc -o gpurun_out/r2d/ingestbench tools/ingestbench.cpp -L roskfpos_amd/csrc -lkfpos_hip -Wl,-rpath,$PWD/roskfpos_amd/csrc && (timeout -k 10 200 gpurun_out/r2d/ingestbench 65536 12 0; timeout -k 10 200 gpurun_out/r2d/ingestbench 65536 12 1; timeout -k 10 200 gpurun_out/r2d/ingestbench 65536 12 0 1; timeout -k 10 200 gpurun_out/r2d/ingestbench 4096 50 0) > gpurun_out/r2d/ingestbench.jsonl 2>&1; cat gpurun_out/r2d/ingestbench.jsonl
rm -f gpurun_out/r2d/ingestbench
