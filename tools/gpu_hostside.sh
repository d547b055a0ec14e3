# host-facing numbers with the library as it ships: single-tag adaptor objects, streaming slots, ranging ingest
set -e
mkdir -p gpurun_out/hostside
g++ -O2 -std=c++17 -I include -I roskfpos_amd/csrc -o gpurun_out/hostside/adaptor_latency tools/adaptor_latency.cpp -L roskfpos_amd/csrc -lkfpos_hip -Wl,-rpath,$PWD/roskfpos_amd/csrc
timeout -k 10 120 gpurun_out/hostside/adaptor_latency 500 > gpurun_out/hostside/adaptor_latency.jsonl 2>&1; cat gpurun_out/hostside/adaptor_latency.jsonl
g++ -O2 -std=c++17 -I include -I roskfpos_amd/csrc -o gpurun_out/hostside/ingestbench tools/ingestbench.cpp -L roskfpos_amd/csrc -lkfpos_hip -Wl,-rpath,$PWD/roskfpos_amd/csrc
(timeout -k 10 200 gpurun_out/hostside/ingestbench 65536 12 0; timeout -k 10 200 gpurun_out/hostside/ingestbench 65536 12 1) > gpurun_out/hostside/ingestbench.jsonl 2>&1; cat gpurun_out/hostside/ingestbench.jsonl
timeout -k 10 300 python tools/hostbench.py --steps 60 > gpurun_out/hostside/hostbench.json 2>/dev/null; cat gpurun_out/hostside/hostbench.json
rm -f gpurun_out/hostside/adaptor_latency gpurun_out/hostside/ingestbench
