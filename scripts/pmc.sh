#!/bin/bash
# One rocprofv3 --pmc pass per argument group over a short bench run; prints per-kernel counter sums / launches.
# Usage (GPU box): bash scripts/pmc.sh <bench kernel: bsw|chain|phmm|poa|abea> "<CTR1 CTR2>" ["<CTR3 CTR4>" ...]
# Counters go in their own passes (no tracing domains combined with --pmc).  Output: gpurun_out/pmc_<kernel>.json
k=$1; shift
export TMPDIR=/tmp
root=$PWD
out=$root/gpurun_out/pmc_$k; rm -rf $out; mkdir -p $out
i=0
for grp in "$@"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $grp --output-format csv -d $out/p$i -- python3 $root/bench.py --kernel $k --steps 2 --warmup 1 --no-cpu $BENCH_ARGS > $out/p$i.log 2>&1)
done
python3 - $out <<'PY'
import sys, glob, csv, json, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("gbx::", "").split("(")[0][-60:]
        if kn.startswith("bsw_lane_kernel"): kn += "@" + str(int(r["Grid_Size"]) // 64 // 256)      # one symbol, a launch per LDS class: told apart by resident blocks per CU (dynamic LDS is not in the record)
        agg[kn][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (f, r["Dispatch_Id"])
        if key not in seen: seen.add(key); cnt[(f, kn)] += 1
launches = collections.defaultdict(int)
for (f, kn), c in cnt.items(): launches[kn] = max(launches[kn], c)
res = {kn: {"launches": launches[kn], **{c: v / max(launches[kn], 1) for c, v in d.items()}} for kn, d in agg.items()}
json.dump(res, open(out + ".json", "w"), indent=1)
for kn, d in sorted(res.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    print(kn, {k: (round(v) if isinstance(v, float) else v) for k, v in d.items()})
PY
