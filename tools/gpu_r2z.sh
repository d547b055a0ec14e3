# full GPU suite, then same-box A/B of the previous commit's library against this one (all configs), then PMC of the
# 9-state kernel: previous commit / this one / this one with pairs
set -e
mkdir -p gpurun_out/r2z
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
bash tools/gpu_libs_ab.sh ml,c2,toa6_65k,c4shard,c3,c5,iw8,planar,planar_sens head:tools/exp/_build/libkfpos_head.so new:roskfpos_amd/csrc/libkfpos_hip.so > gpurun_out/r2z/ab_all.txt 2>&1
cp gpurun_out/libs_ab/ab.jsonl gpurun_out/r2z/ab_all.jsonl
bash tools/gpu_libs_ab.sh c3 head:tools/exp/_build/libkfpos_head.so new:roskfpos_amd/csrc/libkfpos_hip.so pairs:roskfpos_amd/csrc/libkfpos_hip.so:KFPOS_PAIR9=1 > gpurun_out/r2z/ab_c3.txt 2>&1
cp gpurun_out/libs_ab/ab.jsonl gpurun_out/r2z/ab_c3.jsonl
python - <<'PY'
import json,collections
for f in ("gpurun_out/r2z/ab_all.jsonl","gpurun_out/r2z/ab_c3.jsonl"):
    d=collections.defaultdict(list)
    for l in open(f):
        r=json.loads(l); d[(r["config"],r["variant"])].append(r["us"])
    for k,v in sorted(d.items()): print(k, round(sum(v)/len(v),2), v)
PY
bash tools/gpu_pmc_ab2.sh r2z/pmc c3 k_step_imu9 "0 tools/exp/_build/libkfpos_head.so" "0 roskfpos_amd/csrc/libkfpos_hip.so" "1 roskfpos_amd/csrc/libkfpos_hip.so"
