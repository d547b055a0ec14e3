mkdir -p gpurun_out/r2q
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2q/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2q/pytest.log; tail -3 gpurun_out/r2q/pytest.log
bash tools/gpu_ab.sh tools/exp/_build/libkfpos_prev.so roskfpos_amd/csrc/libkfpos_hip.so c5,toa6_65k,c3,iw8,c2
