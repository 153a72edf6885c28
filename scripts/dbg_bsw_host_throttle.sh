#!/usr/bin/env bash
# Are the occasional 5-8 ms stalls of gbx_bsw_extend_host the container's CPU quota (cpu.max: 16 cores on the pool's boxes)?  cpu.stat of
# the cgroup before and after sixty calls, with the default helper threads and with fewer: scripts/dbg_bsw_host_throttle.sh
stat() { cat /sys/fs/cgroup/cpu.stat 2>/dev/null | grep -E "nr_periods|nr_throttled|throttled_usec" | tr '\n' ' '; echo; }
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc: $(nproc)"
for env in "" "GBX_HOST_THREADS=4" "GBX_HOST_THREADS=8"; do
  echo "== ${env:-default}"
  echo -n "before: "; stat
  env $env python3 scripts/dbg_bsw_host_many.py 60 2>/dev/null | grep calls | cut -c1-120
  echo -n "after:  "; stat
done
