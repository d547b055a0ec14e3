# same-box timing of several library builds. usage: gpu_libs_ab.sh CONFIGS "name:lib[:ENV=VAL]" ...
mkdir -p gpurun_out/libs_ab; CFG=$1; shift
: > gpurun_out/libs_ab/ab.jsonl
for rep in 1 2 3; do
  for spec in "$@"; do
    name=${spec%%:*}; rest=${spec#*:}; lib=${rest%%:*}; envs=""; [ "$rest" != "$lib" ] && envs=${rest#*:}
    env KFPOS_LIB_PATH=$PWD/$lib $envs timeout -k 10 200 python tools/kbench.py --steps 100 --warmup 50 --configs $CFG 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(json.dumps({'variant':'$name','config':d['config'],'us':d['us_per_launch']}))" >> gpurun_out/libs_ab/ab.jsonl
  done
done
cat gpurun_out/libs_ab/ab.jsonl
