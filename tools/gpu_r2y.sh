set -e
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
mkdir -p gpurun_out/libs_ab
bash tools/gpu_libs_ab.sh c3 prev:tools/exp/_build/libkfpos_ship_prev.so new:roskfpos_amd/csrc/libkfpos_hip.so pairs:roskfpos_amd/csrc/libkfpos_hip.so:KFPOS_PAIR9=1 > gpurun_out/libs_ab/log.txt 2>&1
python - <<'PY'
import json,collections
d=collections.defaultdict(list)
for l in open("gpurun_out/libs_ab/ab.jsonl"):
    r=json.loads(l); d[(r["config"],r["variant"])].append(r["us"])
for k,v in sorted(d.items()): print(k, round(sum(v)/len(v),2), v)
PY
