mkdir -p gpurun_out/r2g
O=gpurun_out/r2g/sdma_matrix.jsonl
: > $O
timeout -k 10 120 python tools/hostbench.py --modes slots,reuse >> $O 2>/dev/null
HSA_ENABLE_SDMA=0 timeout -k 10 120 python tools/hostbench.py --modes slots,reuse >> $O 2>/dev/null
GPU_FORCE_BLIT_COPY_SIZE=0 timeout -k 10 120 python tools/hostbench.py --modes slots,reuse >> $O 2>/dev/null
GPU_FORCE_BLIT_COPY_SIZE=1048576 timeout -k 10 120 python tools/hostbench.py --modes slots,reuse >> $O 2>/dev/null
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r2g/prof -- python3 $GRAFT_REPO_ROOT/tools/hostbench.py --modes slots,reuse >> $GRAFT_REPO_ROOT/$O 2>/dev/null
cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/r2g/prof && cat $O
