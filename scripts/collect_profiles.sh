#!/bin/bash
# Round-end evidence on the GPU box: parity tests, smoke, the four bench lines (with cpu_baseline), rocprofv3 kernel
# stats and the HBM counters of the dominant kernels.  Everything lands in gpurun_out/<tag>_*; copy what is
# to be judged into profiles/.   Usage: bash scripts/collect_profiles.sh <tag>
tag=${1:-rXX}
out=gpurun_out
mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/${tag}_pytest_gpu.log 2>&1; tail -2 $out/${tag}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/${tag}_smoke.log 2>&1; tail -1 $out/${tag}_smoke.log
for k in bsw chain phmm poa; do
  timeout 900 python bench.py --kernel $k > $out/${tag}_${k}_bench.log 2>&1
  tail -1 $out/${tag}_${k}_bench.log > $out/${tag}_${k}_large_bench.json
  cut -c1-160 $out/${tag}_${k}_large_bench.json
done
for k in bsw chain phmm poa; do
  bash scripts/kstats.sh $k > $out/${tag}_${k}_kstats.txt 2>&1
  cp $out/kstats_$k.csv $out/${tag}_${k}_large_kernel_stats.csv 2>/dev/null
  head -4 $out/${tag}_${k}_kstats.txt
done
# HBM counters: not for poa (a --pmc pass over the poa kernel did not finish within 25 minutes on this pool)
for k in bsw chain phmm; do
  timeout 600 bash scripts/pmc.sh $k "FETCH_SIZE" "WRITE_SIZE" > /dev/null 2>&1
  python3 scripts/pmc_summary.py $out/pmc_$k $out/${tag}_${k}_pmc_hbm.json | head -3
done
