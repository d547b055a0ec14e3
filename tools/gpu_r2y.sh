set -e
timeout -k 10 600 python -m pytest tests/test_pairs9_gpu.py tests/test_gpu_parity.py -x -q -k "pairs or imu or nine or baseline or c3 or fused" 2>&1 | tail -4
bash tools/gpu_libs_ab.sh c3 cur:tools/exp/_build/libkfpos_cur.so new:roskfpos_amd/csrc/libkfpos_hip.so
