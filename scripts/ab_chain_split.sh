#!/usr/bin/env bash
# gbx_chain_host on 'large': the longest jobs in a launch of their own with everything else downloaded meanwhile (GBX_CHAIN_SPLIT_TOP
# jobs; 0 = one launch, one download behind it = the form up to round 6), interleaved; the C entry's own timeline (GBX_HOST_TRACE) says
# when the results were home - the Python wrapper around it allocates the 660 MB of outputs on top: scripts/ab_chain_split.sh
for round in 1 2; do
  for top in 0 64 32 128; do
    echo -n "GBX_CHAIN_SPLIT_TOP=$top: "
    GBX_CHAIN_SPLIT_TOP=$top GBX_HOST_TRACE=1 python3 scripts/dbg_host_entries.py chain 2>&1 | grep -E "results downloaded|gbx_chain_host" | awk '/results downloaded/ {v[n++]=$4} /gbx_chain_host/ {line=$0} END {m=v[0]; for(i in v) if (v[i]<m) m=v[i]; printf "entry best %.1f ms of %d calls | wrapper: %s\n", m, n, line}'
  done
done
