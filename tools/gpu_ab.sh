# A/B of two library builds on ONE box: alternating kbench runs. usage: gpu_ab.sh LIB_A LIB_B CONFIGS
mkdir -p gpurun_out/ab
: > gpurun_out/ab/ab.jsonl
for rep in 1 2 3; do
  for lib in "$1" "$2"; do
    KFPOS_LIB_PATH=$PWD/$lib timeout -k 10 200 python tools/kbench.py --steps 100 --warmup 50 --configs $3 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(json.dumps({'lib':'$lib','config':d['config'],'us':d['us_per_launch'],'hbm_frac':round(d['hbm_frac'],4)}))" >> gpurun_out/ab/ab.jsonl
  done
done
cat gpurun_out/ab/ab.jsonl
