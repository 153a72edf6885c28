#!/usr/bin/env bash
# gbx_bsw_extend_host on 'large' by number of upload workers (GBX_HOST_THREADS; default 6), interleaved, 60 calls each:
# scripts/sweep_bsw_host_threads.sh > gpurun_out/<tag>_bsw_host_threads.txt
for round in 1 2 3; do
  for t in 6 3 4 5 2; do
    GBX_HOST_THREADS=$t python3 scripts/dbg_bsw_host_many.py 60 2>/dev/null | grep calls | cut -c1-130
  done
done
