set -e
mkdir -p gpurun_out/libs_ab
bash tools/gpu_libs_ab.sh c3,c5,toa6_65k,c4shard ship:roskfpos_amd/csrc/libkfpos_hip.so maxilp:tools/exp/_build/libkfpos_maxilp.so bias0:tools/exp/_build/libkfpos_bias0.so > gpurun_out/libs_ab/log.txt 2>&1
python - <<'PY'
import json,collections
d=collections.defaultdict(list)
for l in open("gpurun_out/libs_ab/ab.jsonl"):
    r=json.loads(l); d[(r["config"],r["variant"])].append(r["us"])
for k,v in sorted(d.items()): print(k, round(sum(v)/len(v),2), v)
PY
