mkdir -p gpurun_out/r2t
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2t/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2t/pytest.log; tail -3 gpurun_out/r2t/pytest.log
bash tools/gpu_ab.sh tools/exp/_build/libkfpos_prev.so roskfpos_amd/csrc/libkfpos_hip.so iw8,c5
