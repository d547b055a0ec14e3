mkdir -p gpurun_out/r2o
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r2o/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2o/pytest.log; tail -3 gpurun_out/r2o/pytest.log
bash tools/gpu_ab.sh tools/exp/_build/libkfpos_prev.so roskfpos_amd/csrc/libkfpos_hip.so c5
