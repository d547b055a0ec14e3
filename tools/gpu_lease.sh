#!/bin/bash
# Everything this repository runs on a GPU lease, in one place.   usage: bash tools/gpu_lease.sh TAG COMMAND [args]
# Output goes to gpurun_out/TAG/ (scratch); copy what should be judged into profiles/ (profiles/README.md).
#
#   tests                      the GPU test-suite
#   one PYTEST_ARGS...         part of it (e.g. tests/test_comm_world_gpu.py -x)
#   final                      end-of-round validation: suite, smoke(), the driver's bench command, the default bench, the
#                              plain two-rank command (bench.py spawns its ranks; gloo when they share a card), c4
#   kbench CONFIGS [STEPS]     tools/kbench.py step-kernel timings (CONFIGS: comma list, e.g. c2,c3,c5)
#   ab CONFIGS SPEC...         same-box timing of several library builds, three alternating rounds.
#                              SPEC = name:path/to/lib.so[:ENV=VAL] ; write down the commit each library was built from
#   pmc CONFIG MATCH [STEPS] [LIB...]   SQ counters of one kbench configuration (three --pmc passes) per library
#                              (default: the library in the tree); MATCH selects the kernel by name
#   profile                    bench.py under rocprofv3: --kernel-trace --stats of the driver's and the default command,
#                              FETCH_SIZE / WRITE_SIZE passes at two launch lengths -> traffic_latest.json
#   soak MODES [EPOCHS]        tests/soak.py (2 048 tags x 2 000 epochs against the oracle) per storage mode, e.g. mixed,p48
#   hostside                   single-tag adaptor latency, streaming slots, ranging ingest (tools/*.cpp, hostbench.py)
set -o pipefail
TAG=$1; CMD=$2; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
COMMIT=$(cat $R/.git/HEAD 2>/dev/null | head -c 60)   # (.git does not travel to the box: usually empty there)

kbench() { ( cd $R && timeout -k 10 300 python3 tools/kbench.py --steps ${2:-50} --warmup ${2:-50} --configs $1 ); }

case $CMD in
tests)
  cd $R && timeout -k 10 1100 python3 -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo rc=$? >> $OUT/tests.log; tail -5 $OUT/tests.log ;;
one)
  cd $R && timeout -k 10 900 python3 -m pytest "$@" -m gpu -q > $OUT/one.log 2>&1; echo rc=$? >> $OUT/one.log; tail -40 $OUT/one.log ;;
final)
  cd $R
  timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo rc=$? >> $OUT/tests.log; tail -3 $OUT/tests.log
  timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $OUT/smoke.log
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.log 2>$OUT/bench_driver.err; cut -c1-300 $OUT/bench_driver.log
  timeout -k 10 400 python3 bench.py > $OUT/bench_default.log 2>$OUT/bench_default.err; cut -c1-200 $OUT/bench_default.log
  timeout -k 10 300 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_c3_2ranks.log 2>$OUT/bench_c3_2ranks.err; cut -c1-200 $OUT/bench_c3_2ranks.log
  timeout -k 10 300 python3 bench.py --gpus 2 --config c4 --total-tags 262144 --steps 20 --warmup 5 --gather epoch > $OUT/bench_c4_2ranks.log 2>$OUT/bench_c4_2ranks.err; cut -c1-200 $OUT/bench_c4_2ranks.log
  timeout -k 10 300 python3 bench.py --config c4 --steps 20 --warmup 5 > $OUT/bench_c4_n1.log 2>$OUT/bench_c4_n1.err; cut -c1-200 $OUT/bench_c4_n1.log ;;
kbench)
  kbench $1 $2 2>/dev/null | grep '^{' | tee $OUT/kbench.jsonl | cut -c1-170 ;;
ab)
  CFG=$1; shift; : > $OUT/ab.jsonl
  for rep in 1 2 3; do
    for spec in "$@"; do
      name=${spec%%:*}; rest=${spec#*:}; lib=${rest%%:*}; envs=""; [ "$rest" != "$lib" ] && envs=${rest#*:}
      ( cd $R && env KFPOS_LIB_PATH=$R/$lib $envs timeout -k 10 300 python3 tools/kbench.py --steps 100 --warmup 50 --configs $CFG 2>/dev/null ) | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(json.dumps({'variant':'$name','config':d['config'],'us':d['us_per_launch'],'rep':$rep}))" >> $OUT/ab.jsonl
    done
  done
  python3 - <<PY
import json, collections
rows=[json.loads(l) for l in open("$OUT/ab.jsonl")]
by=collections.defaultdict(list)
for r in rows: by[(r["config"],r["variant"])].append(r["us"])
for (c,v),us in sorted(by.items()): print(f"{c:10s} {v:24s} " + " ".join(f"{u:7.2f}" for u in us) + f"   median {sorted(us)[len(us)//2]:.2f} us")
PY
  ;;
pmc)
  CFG=$1; MATCH=$2; STEPS=${3:-20}; shift 3 2>/dev/null || shift $#
  LIBS=("$@"); [ ${#LIBS[@]} -eq 0 ] && LIBS=(roskfpos_amd/csrc/libkfpos_hip.so)
  cd /tmp && export TMPDIR=/tmp
  for lib in "${LIBS[@]}"; do
    n=$(basename $lib .so)
    KFPOS_LIB_PATH=$R/$lib rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/pmcA_$n --output-format csv -- python3 $R/tools/kbench.py --steps $STEPS --warmup $STEPS --configs $CFG > $OUT/pmcA_$n.log 2>&1
    KFPOS_LIB_PATH=$R/$lib rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/pmcB_$n --output-format csv -- python3 $R/tools/kbench.py --steps $STEPS --warmup $STEPS --configs $CFG > $OUT/pmcB_$n.log 2>&1
    python3 $R/tools/pmc_summary.py $OUT/pmcA_$n $OUT/pmcB_$n --match $MATCH > $OUT/pmc_${CFG}_$n.json; rm -rf $OUT/pmcA_$n $OUT/pmcB_$n
    python3 - <<PY
import json
d=json.load(open("$OUT/pmc_${CFG}_$n.json"))
for k,c in d.items():
    e=$STEPS
    print("$n", k[-70:], "| VALU instr / wave-epoch %.0f"%(c["SQ_INSTS_VALU"]/c["SQ_WAVES"]/e), "| SALU %.0f"%(c["SQ_INSTS_SALU"]/c["SQ_WAVES"]/e),
          "| wave quad-cycles / epoch %.0f"%(c["SQ_WAVE_CYCLES"]/c["SQ_WAVES"]/e), "| VALU busy %.3f"%(c["SQ_ACTIVE_INST_VALU"]/c["SQ_WAVE_CYCLES"]),
          "| wait-inst %.3f"%(c["SQ_WAIT_INST_ANY"]/c["SQ_WAVE_CYCLES"]), "| waves %.0f"%c["SQ_WAVES"])
PY
  done ;;
profile)
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats -d $OUT/stats_driver --output-format csv -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_profiled.log 2>&1
  rocprofv3 --kernel-trace --stats -d $OUT/stats_default --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $OUT/bench_default_profiled.log 2>&1
  for E in 25 5; do
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch_$E --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-per-epoch --repeats 1 --epochs-per-launch $E --steps 100 --warmup 25 > $OUT/fetch_$E.log 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write_$E --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-per-epoch --repeats 1 --epochs-per-launch $E --steps 100 --warmup 25 > $OUT/write_$E.log 2>&1
  done
  cd $R
  KEY=$(python3 -c "import bench; print(bench.CONFIGS['c3']['kernel'])")
  # the headline kernel as rocprofv3 spells it (the same process also runs the P48 instantiation: secondary.c3_p48)
  MATCH=$(python3 -c "import bench; k=bench.CONFIGS['c3']['kernel']; print('k_step_imu9<(anonymous namespace)::p48' if 'p48' in k else k.split(',')[0] + ', ' + k.split(',')[1] + ',')")
  python3 tools/traffic_summary.py --match "$MATCH" --key "$KEY" --tags 65536 --run 25 $OUT/fetch_25 $OUT/write_25 --run 5 $OUT/fetch_5 $OUT/write_5 --note "bench.py --no-cpu-baseline --no-per-epoch --repeats 1 --steps 100 --warmup 25 at 25 and 5 epochs per launch ($TAG)" --merge $OUT/traffic_latest.json > /dev/null
  python3 tools/traffic_summary.py --match k_step_toa6 --key "k_step_toa6<true,double,double,8,0>" --tags 65536 --run 25 $OUT/fetch_25 $OUT/write_25 --run 5 $OUT/fetch_5 $OUT/write_5 --note "the secondary 6-state line of the same runs ($TAG)" --merge $OUT/traffic_latest.json > /dev/null
  head -40 $OUT/traffic_latest.json
  for d in stats_driver stats_default; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); cp $f $OUT/${d}_kernel_stats.csv; head -6 $f | cut -c1-160; done
  python3 - <<PY
import csv, glob, json
out = {}
for d in ("stats_driver", "stats_default"):
    f = glob.glob("$OUT/%s/**/*kernel_trace.csv" % d, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "$MATCH" in r["Kernel_Name"]]
    durs = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows)
    big = [x for x in durs if x > 0.5 * durs[-1]]
    out[d] = {"k_step_imu9_dispatches": len(durs), "full_size_launches": len(big), "mean_us_full_size_launch": sum(big) / len(big),
              "us_full_size_launches": big,
              "single_epoch_launches": len([x for x in durs if x < 100]), "mean_us_single_epoch_launch": (lambda v: sum(v) / len(v) if v else None)([x for x in durs if x < 100])}
json.dump(out, open("$OUT/dispatch_summary.json", "w"), indent=1); print(json.dumps(out, indent=1)[:1500])
PY
  rm -rf $OUT/stats_driver $OUT/stats_default $OUT/fetch_* $OUT/write_*
  grep '^{"metric"' $OUT/bench_driver_profiled.log | cut -c1-300 ;;
soak)
  cd $R
  for m in $(echo $1 | tr , ' '); do timeout -k 10 900 python3 tests/soak.py $m $2 > $OUT/soak_$m.log 2>&1; echo rc=$? >> $OUT/soak_$m.log; grep "ALL\|rc=" $OUT/soak_$m.log; done ;;
hostside)
  cd $R
  g++ -O2 -std=c++17 -I include -I roskfpos_amd/csrc -o $OUT/adaptor_latency tools/adaptor_latency.cpp -L roskfpos_amd/csrc -lkfpos_hip -Wl,-rpath,$R/roskfpos_amd/csrc
  timeout -k 10 120 $OUT/adaptor_latency 500 > $OUT/adaptor_latency.jsonl 2>&1; cat $OUT/adaptor_latency.jsonl
  g++ -O2 -std=c++17 -I include -I roskfpos_amd/csrc -o $OUT/ingestbench tools/ingestbench.cpp -L roskfpos_amd/csrc -lkfpos_hip -Wl,-rpath,$R/roskfpos_amd/csrc
  (timeout -k 10 200 $OUT/ingestbench 65536 12 0; timeout -k 10 200 $OUT/ingestbench 65536 12 1) > $OUT/ingestbench.jsonl 2>&1; cat $OUT/ingestbench.jsonl
  timeout -k 10 300 python3 tools/hostbench.py --steps 60 > $OUT/hostbench.json 2>/dev/null; cat $OUT/hostbench.json
  rm -f $OUT/adaptor_latency $OUT/ingestbench ;;
*) echo "unknown command $CMD"; exit 2 ;;
esac
