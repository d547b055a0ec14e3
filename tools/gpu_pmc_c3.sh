# PMC passes for the 9-state bench kernel (kbench c3, 50 epochs per launch). usage: bash tools/gpu_pmc_c3.sh OUTDIR
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/pmcA --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --steps 50 --warmup 50 --configs c3 > $OUT/pmcA.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS -d $OUT/pmcB --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --steps 50 --warmup 50 --configs c3 > $OUT/pmcB.log 2>&1
cd $GRAFT_REPO_ROOT && python3 tools/pmc_summary.py $OUT/pmcA $OUT/pmcB --match k_step_imu9 > $OUT/pmc_c3.json; cat $OUT/pmc_c3.json; rm -rf $OUT/pmcA $OUT/pmcB
