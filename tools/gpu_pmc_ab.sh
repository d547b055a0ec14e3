# VALU instruction counts of one kbench configuration for two library builds. usage: gpu_pmc_ab.sh TAG CONFIG MATCH LIB_A LIB_B
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in $4 $5; do
  n=$(basename $lib .so)
  KFPOS_LIB_PATH=$R/$lib rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/pmc_$n --output-format csv -- python3 $R/tools/kbench.py --steps 20 --warmup 20 --configs $2 > $OUT/pmc_$n.log 2>&1
  python3 $R/tools/pmc_summary.py $OUT/pmc_$n --match $3 > $OUT/pmc_$2_$n.json; rm -rf $OUT/pmc_$n
  python3 - <<PY
import json
d=json.load(open("$OUT/pmc_$2_$n.json"))
for k,c in d.items():
    print("$n", k[28:70], "valu/wave-epoch %.0f"%(c["SQ_INSTS_VALU"]/c["SQ_WAVES"]/20), "salu %.0f"%(c["SQ_INSTS_SALU"]/c["SQ_WAVES"]/20), "wave quad-cycles/epoch %.0f"%(c["SQ_WAVE_CYCLES"]/c["SQ_WAVES"]/20), "valu busy %.3f"%(c["SQ_ACTIVE_INST_VALU"]/c["SQ_WAVE_CYCLES"]))
PY
done
