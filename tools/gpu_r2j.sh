mkdir -p gpurun_out/r2j
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2j/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2j/pytest.log; tail -4 gpurun_out/r2j/pytest.log
timeout -k 10 300 python tools/kbench.py --steps 50 --warmup 50 --configs c3,c3_f32,toa6_65k,c5,iw8 > gpurun_out/r2j/kbench.jsonl 2>/dev/null; python - <<'PY'
import json
for l in open('gpurun_out/r2j/kbench.jsonl'):
    d=json.loads(l); print(d['config'], d['us_per_launch'], round(d['hbm_frac'],4), d['mean_gain_iters'])
PY
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r2j/bench.log 2>&1; tail -c 900 gpurun_out/r2j/bench.log
