#!/bin/bash
# Round-end evidence on the GPU box (everything under gpurun_out/<tag>_*; copy what is to be judged into profiles/):
# the GPU suite, smoke, the default bench line (all six kernels, cpu baselines), rocprofv3 kernel stats of bsw and poa,
# and their counter passes (HBM bytes, VALU instructions / busy) stamped with the kernel source's hash at collection time.
#   bash scripts/collect_round.sh <tag> [kinds for the counter passes, default "bsw poa"]
tag=${1:-rXX}; kinds=${2:-"bsw poa"}
out=gpurun_out; mkdir -p $out
(time timeout 1500 python -m pytest tests -m gpu -x -q) > $out/${tag}_pytest_gpu.log 2>&1; tail -4 $out/${tag}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/${tag}_smoke.log 2>&1; tail -1 $out/${tag}_smoke.log
timeout 1500 python bench.py > $out/${tag}_bench_all_kernels.json 2> $out/${tag}_bench.err; grep "^kernels in full: " $out/${tag}_bench.err | cut -c18- > $out/${tag}_bench_kernels_full.json; python3 scripts/show_bench_line.py < $out/${tag}_bench_all_kernels.json 2>/dev/null | cut -c1-400 | head -8
for k in $kinds; do
  bash scripts/kstats.sh $k > $out/${tag}_${k}_kstats.txt 2>&1
  cp $out/kstats_$k.csv $out/${tag}_${k}_large_kernel_stats.csv 2>/dev/null
  head -6 $out/${tag}_${k}_kstats.txt
  timeout 900 bash scripts/pmc.sh $k "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES" > /dev/null 2>&1
  python3 scripts/pmc_summary.py $out/pmc_$k $out/${tag}_${k}_pmc.json | head -8
done
