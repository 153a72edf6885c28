for n in 512 2048 8192; do
for sh in "" 16x16 16x12 16x10 64x3 64x4 8x16; do
  echo -n "n=$n shape=${sh:-default}: "; GBX_BSW_DIRECT_SHAPE=$sh python scripts/dbg_combined_call.py bsw $n 30 2>/dev/null | grep median
done; done
