# A/B of BswChunkPrep::class_pairs (empty lane launches left out of a pipelined host call): sixty calls each way, twice
for r in 1 2; do
  python scripts/dbg_bsw_host_many.py 60
  GBX_BSW_SKIP_EMPTY=1 python scripts/dbg_bsw_host_many.py 60
done
