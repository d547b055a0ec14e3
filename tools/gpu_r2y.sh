set -e
timeout -k 10 600 python -m pytest tests/test_pairs9_gpu.py -x -q 2>&1 | tail -5
bash tools/gpu_libs_ab.sh c3 head:tools/exp/_build/libkfpos_head.so nopairs:roskfpos_amd/csrc/libkfpos_hip.so:KFPOS_NO_PAIR9=1 pairs:roskfpos_amd/csrc/libkfpos_hip.so
