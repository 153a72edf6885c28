# one phmm host call of the sizes the call combiner makes (16 ... 256 of the reference driver's batches), by pairs per unit of the
# stream path (GBX_PHMM_SEG) and by path (GBX_PHMM_SMALL=1: one pair per wavefront on the tiled kernels)
for units in 16 32 64 128 256 1024; do
  for seg in 1 2 4 8; do
    echo -n "batches $units seg $seg small 0: "; GBX_PHMM_SMALL=0 GBX_PHMM_SEG=$seg python scripts/dbg_combined_call.py phmm $units 15 2>/dev/null | grep median
  done
  echo -n "batches $units small 1: "; GBX_PHMM_SMALL=1 python scripts/dbg_combined_call.py phmm $units 15 2>/dev/null | grep median
done
