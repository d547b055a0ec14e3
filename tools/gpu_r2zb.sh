set -e
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_ml_estimator.py -m gpu -x -q 2>&1 | tail -4
bash tools/gpu_libs_ab.sh c5,c3 head:tools/exp/_build/libkfpos_head.so new:roskfpos_amd/csrc/libkfpos_hip.so
