# VALU / LDS counters of one kbench configuration for library builds. usage: gpu_pmc_ab2.sh TAG CONFIG MATCH "ENV=VAL LIB" ...
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT; CFG=$2; MATCH=$3; shift 3
cd /tmp && export TMPDIR=/tmp
i=0
for spec in "$@"; do
  i=$((i+1))
  set -- $spec
  export KFPOS_PAIR9=$1
  lib=$2
  n=v${i}_$(basename $lib .so)_pair$1
  KFPOS_LIB_PATH=$R/$lib rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/pmcA_$n --output-format csv -- python3 $R/tools/kbench.py --steps 20 --warmup 20 --configs $CFG > $OUT/pmcA_$n.log 2>&1
  KFPOS_LIB_PATH=$R/$lib rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA -d $OUT/pmcB_$n --output-format csv -- python3 $R/tools/kbench.py --steps 20 --warmup 20 --configs $CFG > $OUT/pmcB_$n.log 2>&1
  python3 $R/tools/pmc_summary.py $OUT/pmcA_$n $OUT/pmcB_$n --match $MATCH > $OUT/pmc_${CFG}_$n.json; rm -rf $OUT/pmcA_$n $OUT/pmcB_$n
  python3 - <<PY
import json
d=json.load(open("$OUT/pmc_${CFG}_$n.json"))
for k,c in d.items():
    w=c["SQ_WAVES"]*20
    print("$n", k[28:64], "valu %.0f"%(c["SQ_INSTS_VALU"]/w), "salu %.0f"%(c["SQ_INSTS_SALU"]/w), "lds %.0f"%(c.get("SQ_INSTS_LDS",0)/w), "quad-cycles %.0f"%(c["SQ_WAVE_CYCLES"]/w), "valu busy %.3f"%(c["SQ_ACTIVE_INST_VALU"]/c["SQ_WAVE_CYCLES"]), "wait_lds %.3f"%(c.get("SQ_WAIT_INST_LDS",0)/c["SQ_WAVE_CYCLES"]), "wait_any %.3f"%(c["SQ_WAIT_INST_ANY"]/c["SQ_WAVE_CYCLES"]))
PY
done
