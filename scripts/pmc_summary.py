"""Per-kernel means of the counters collected by scripts/pmc.sh: python3 scripts/pmc_summary.py gpurun_out/pmc_<k>"""
import glob, csv, collections, re, sys, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+)(<[^>]*>)?\(", r["Kernel_Name"].replace("(anonymous namespace)::", ""))
        kn = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:40]
        if kn.startswith("bsw_lane_kernel"): kn += "@" + str(int(r["Grid_Size"]) // 64 // 256)      # one symbol, a launch per LDS class: told apart by resident blocks per CU (dynamic LDS is not in the record)
        agg[kn][r["Counter_Name"]] += float(r["Counter_Value"]); n[(kn, f)].add(r["Dispatch_Id"])
out = {}
for kn, d in agg.items():
    L = max(len(v) for (k, f), v in n.items() if k == kn)
    d = {k: v / L for k, v in d.items()}
    g = d.get("GRBM_GUI_ACTIVE", 0)
    if g < 2e5: continue
    if "SQ_ACTIVE_INST_VALU" in d: d["valu_busy"] = d["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * g / 8)
    if "SQ_INSTS_VALU" in d and d.get("SQ_WAVES"): d["valu_per_wave"] = d["SQ_INSTS_VALU"] / d["SQ_WAVES"]
    if "SQ_WAIT_INST_ANY" in d and d.get("SQ_WAVE_CYCLES"): d["wait_frac"] = d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"]
    out[kn] = {k: round(v, 3) for k, v in d.items()}
    print("%-36s launches %d" % (kn, L), {k: (round(v, 3) if v < 100 else int(v)) for k, v in d.items()})
# the hash of the kernel source these counters were collected on, recorded HERE (at collection time, on the box that ran
# them): scripts/make_profile_tables.py copies it, so that re-running that script later cannot re-stamp old counters
import os
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _root)
from genomicsbench_amd.srchash import KIND_SOURCE, tu_sha16
_kind = os.path.basename(sys.argv[1].rstrip("/")).replace("pmc_", "")
if _kind in KIND_SOURCE: out["_hip_sha16"] = tu_sha16(_kind)          # every file of the kind's translation unit
if len(sys.argv) > 2: json.dump(out, open(sys.argv[2], "w"), indent=1)
