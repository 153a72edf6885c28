#!/usr/bin/env bash
# phmm stream kernels with the priors from LDS tables (in tree) against the compare-and-select form (GBX_PHMM_LUT=0), interleaved on one
# box: scripts/ab_phmm_lut.sh > gpurun_out/<tag>_phmm_lut_ab.txt
echo "== phmm 'large' (bench.py --kernel phmm, 10 steps): prior look-up tables in LDS (default) against compares and selects (GBX_PHMM_LUT=0), interleaved"
for round in 1 2 3; do
  for lut in 1 0; do
    echo "-- GBX_PHMM_LUT=$lut"
    GBX_PHMM_LUT=$lut python bench.py --kernel phmm --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d.get('kernels',{}).get('phmm',d)
print('   ms_per_step %.2f  value %.1f %s  verified %s' % (k['ms_per_step'], k['value'], k['unit'], str(k.get('cpu_baseline',{}).get('verified'))[:60]))"
  done
done
