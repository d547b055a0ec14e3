# Profiles of bench.py for the round's profiles/ directory. usage: bash tools/gpu_profile_bench.sh TAG
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. the driver's exact command under --kernel-trace --stats
rocprofv3 --kernel-trace --stats -d $OUT/stats_driver --output-format csv -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_profiled.log 2>&1
# 2. the default command
rocprofv3 --kernel-trace --stats -d $OUT/stats_default --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $OUT/bench_default_profiled.log 2>&1
# 3. HBM traffic of the step kernels: FETCH_SIZE / WRITE_SIZE in separate passes, two launch lengths
for E in 25 5; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch_$E --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-per-epoch --epochs-per-launch $E --steps 100 --warmup 25 > $OUT/fetch_$E.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write_$E --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-per-epoch --epochs-per-launch $E --steps 100 --warmup 25 > $OUT/write_$E.log 2>&1
done
cd $R
python3 tools/traffic_summary.py --match k_step_imu9 --key "k_step_imu9<double,float,8,true>" --tags 65536 --run 25 $OUT/fetch_25 $OUT/write_25 --run 5 $OUT/fetch_5 $OUT/write_5 --note "bench.py --no-cpu-baseline --no-per-epoch --steps 100 --warmup 25 at 25 and 5 epochs per launch ($1)" --merge $OUT/traffic_latest.json > /dev/null
python3 tools/traffic_summary.py --match k_step_toa6 --key "k_step_toa6<true,double,double,8,0>" --tags 65536 --run 25 $OUT/fetch_25 $OUT/write_25 --run 5 $OUT/fetch_5 $OUT/write_5 --note "the secondary 6-state line of the same runs ($1)" --merge $OUT/traffic_latest.json > /dev/null
cat $OUT/traffic_latest.json | head -40
for d in stats_driver stats_default; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); cp $f $OUT/${d}_kernel_stats.csv; head -6 $f | cut -c1-160; done
python3 - <<PY
import csv, glob, json
out = {}
for d in ("stats_driver", "stats_default"):
    f = glob.glob("$OUT/%s/**/*kernel_trace.csv" % d, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "k_step_imu9" in r["Kernel_Name"]]
    durs = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows)
    big = [x for x in durs if x > 0.5 * durs[-1]]
    out[d] = {"k_step_imu9_dispatches": len(durs), "full_size_launches": len(big), "mean_us_full_size_launch": sum(big) / len(big),
              "single_epoch_launches": len([x for x in durs if x < 100]), "mean_us_single_epoch_launch": (lambda v: sum(v) / len(v) if v else None)([x for x in durs if x < 100])}
json.dump(out, open("$OUT/dispatch_summary.json", "w"), indent=1); print(json.dumps(out, indent=1))
PY
rm -rf $OUT/stats_driver $OUT/stats_default $OUT/fetch_* $OUT/write_*
grep '^{"metric"' $OUT/bench_driver_profiled.log | cut -c1-300
