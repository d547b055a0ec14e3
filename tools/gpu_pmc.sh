# PMC passes for one kbench configuration. usage: bash tools/gpu_pmc.sh OUTDIR CONFIG MATCH [STEPS]
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
STEPS=${4:-20}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/pmcA --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --steps $STEPS --warmup $STEPS --configs $2 > $OUT/pmcA.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/pmcB --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --steps $STEPS --warmup $STEPS --configs $2 > $OUT/pmcB.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES -d $OUT/pmcC --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --steps $STEPS --warmup $STEPS --configs $2 > $OUT/pmcC.log 2>&1
cd $GRAFT_REPO_ROOT && python3 tools/pmc_summary.py $OUT/pmcA $OUT/pmcB $OUT/pmcC --match $3 > $OUT/pmc_$2.json; cat $OUT/pmc_$2.json; tail -2 $OUT/pmcC.log; rm -rf $OUT/pmcA $OUT/pmcB $OUT/pmcC
