mkdir -p gpurun_out/r2i
O=gpurun_out/r2i/split_matrix.jsonl
: > $O
for rep in 1 2; do
for sb in 0 1048576 4194304; do
  echo "{\"KFPOS_SLOT_SPLIT_BYTES\": $sb}" >> $O
  KFPOS_SLOT_SPLIT_BYTES=$sb timeout -k 10 120 python tools/hostbench.py --modes slots,reuse,nopose >> $O 2>/dev/null
done
done
cat $O
