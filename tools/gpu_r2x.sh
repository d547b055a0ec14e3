# one box: iteration histogram, then product / cap10 / sweep2 timing variants of the 9-state kernel, alternating
set -e
mkdir -p gpurun_out/r2x
timeout -k 10 300 python tools/exp/iter_hist.py > gpurun_out/r2x/iter_hist.json
: > gpurun_out/r2x/variants.jsonl
for rep in 1 2 3; do
  for lib in roskfpos_amd/csrc/libkfpos_hip.so tools/exp/_build/libkfpos_cap10.so tools/exp/_build/libkfpos_sweep2.so; do
    KFPOS_LIB_PATH=$PWD/$lib timeout -k 10 200 python tools/kbench.py --steps 100 --warmup 50 --configs c3 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(json.dumps({'lib':'$lib','config':d['config'],'us':d['us_per_launch']}))" >> gpurun_out/r2x/variants.jsonl
  done
done
cat gpurun_out/r2x/variants.jsonl
head -c 1500 gpurun_out/r2x/iter_hist.json
