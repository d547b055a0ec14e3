# end-of-round evidence with the final kernels: bench under rocprofv3 (driver command + default), HBM traffic passes,
# SQ counters of the 9-state kernel and of config 5, all-config kbench line
set -e
bash tools/gpu_profile_bench.sh r2zzc
bash tools/gpu_pmc_c3.sh r2zzc/pmc_c3
bash tools/gpu_pmc.sh r2zzc/pmc_c5 c5 k_step_toa6 10
timeout -k 10 300 python tools/kbench.py --steps 100 --warmup 50 --configs ml,c2,toa6_65k,c4shard,c3,c5,iw8,planar,planar_sens > gpurun_out/r2zzc/kbench_all.jsonl 2>/dev/null
cat gpurun_out/r2zzc/kbench_all.jsonl | cut -c1-200
