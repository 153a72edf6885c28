#!/bin/bash
# rocprofv3 --kernel-trace --stats over a short bench run; prints per-kernel calls / mean / share.
# Usage (GPU box): bash scripts/kstats.sh <bsw|chain|phmm|poa|abea> [extra bench.py args]   -> gpurun_out/kstats_<kernel>.csv
k=$1; shift
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/kstats_$k; rm -rf $out; mkdir -p $out
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/bench.py --kernel $k --steps 2 --warmup 1 --no-cpu "$@" > $out.log 2>&1)
f=$(ls $out/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -z "$f" ] && { tail -5 $out.log; exit 1; }
cp $f $root/gpurun_out/kstats_$k.csv
python3 - $f <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(anonymous namespace\)::|gbx::|void ", "", r["Name"]).split("(")[0]
    print("%-44s calls %4s  avg %10.1f us  %5s %%" % (n[:44], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
