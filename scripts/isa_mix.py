#!/usr/bin/env python3
"""Static VALU instruction mix of every gbx kernel against the measured issue rates (profiles/valu_peak.json) ->
profiles/valu_mix.json: per kernel the VALU instructions by rate class, the mean SIMD cycles one of them occupies, and the
lane-operation rate a chip running nothing but that mix would reach (the roof bench.py divides by).

Rate classes (scripts/valu_peak.hip on MI355X, 8 wavefronts per SIMD): `full` = 2.25 cycles per wave64 instruction
(v_add/sub/subrev_u32, v_and/or/xor_b32, v_lshrrev_b32, v_ashrrev_i32, v_mov_b32, v_add/sub/mul_f32, v_fmac_f32, 16-bit add/max
in their plain VOP1/VOP2 encodings), `quarter` = 8.2 (transcendentals), `half` = 4.17 (everything else: max/min/max3, v_lshlrev,
compares, selects, bit-field ops, v_perm, every DPP and SDWA form, every packed and 64-bit form, fma, conversions, readlane).
The mix is static (whole kernel, not weighted by trip counts): the kernels are unrolled row / column loops, so the static
mix is dominated by the loop bodies (phmm's stream kernels: the step bodies only - their per-unit set-up is large and rare).  Cross-compiles for gfx950; no GPU needed.

usage: python scripts/isa_mix.py            (writes profiles/valu_mix.json)"""
import collections
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "genomicsbench_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "scripts"))
FULL = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32",
        "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fmac_f32", "v_max_i16", "v_min_i16", "v_max_u16", "v_min_u16", "v_add_u16",
        "v_sub_u16", "v_subrev_u16", "v_not_b32", "v_lshrrev_b16", "v_ashrrev_i16"}
QUARTER = {"v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64",
           "v_rcp_iflag_f32", "v_exp_legacy_f32", "v_log_legacy_f32"}
CYC = {"full": 2.25, "half": 4.17, "quarter": 8.16}


def sha16(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


def rate_class(op):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base in QUARTER:
        return "quarter"
    if op.endswith(("_dpp", "_sdwa", "_e64")) or base.startswith("v_pk_"):
        return "half"
    return "full" if base in FULL else "half"


def main():
    from make_profile_tables import stage_name
    peak = json.load(open(os.path.join(ROOT, "profiles", "valu_peak.json")))
    simds, clk = peak["cus"] * 4, peak["clock_khz"] * 1e3
    out = {}
    tmp = "/tmp/gbx_isa_mix"
    os.makedirs(tmp, exist_ok=True)
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith("_kernels.hip"):
            continue
        extra = ["-fno-slp-vectorize"] if f == "phmm_kernels.hip" else []          # as csrc/Makefile builds it
        subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off"] + extra + ["-c", os.path.join(CSRC, f),
                        "-save-temps", "-o", f + ".o"], cwd=tmp, check=True, stderr=subprocess.DEVNULL)
        asm = open(os.path.join(tmp, f[:-4] + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
        for m in re.finditer(r"^(_Z\w+):\s*; @\1\n(.*?)s_endpgm", asm, re.S | re.M):
            sym, body = m.group(1), m.group(2)
            dem = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
            dem = dem.replace("(anonymous namespace)::", "").replace("void ", "").replace("gbx::", "").split("(")[0]
            counts = collections.Counter()
            lines = body.splitlines()
            if dem.startswith("phmm_stream_kernel"):
                # the stream kernels' set-up per unit (the prior tables, thirty selects and LDS writes per lane, unrolled) is a third of
                # their static instructions and a thousandth of the executed ones: the mix is that of the step bodies - the basic blocks
                # that hold a step's FMAs - which is what the counters weigh
                blocks, cur = [], []
                for line in lines:
                    if re.match(r"^\.LBB\d+_\d+:", line.strip()) or line.strip().startswith("; %bb."):
                        blocks.append(cur); cur = []
                    cur.append(line)
                blocks.append(cur)
                lines = [ln for b in blocks if sum(1 for x in b if x.strip().startswith("v_fmac_f32")) >= 2 * int(re.search(r"<(\d+)", dem).group(1)) for ln in b]
            for line in lines:
                t = line.strip()
                if t.startswith("v_") and not t.startswith(("v_cmpx",)):
                    counts[rate_class(t.split()[0])] += 1
            n = sum(counts.values())
            if n < 16:
                continue
            cyc = sum(CYC[k] * v for k, v in counts.items()) / n
            out[dem] = {"stage": stage_name(dem), "valu_static": n, **{k: counts.get(k, 0) for k in CYC},
                        "mean_cycles_per_valu": round(cyc, 3), "roof_lane_ops_per_s": 64.0 * simds * clk / cyc, "source": f}
    hashes = {f: sha16(os.path.join(CSRC, f)) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))}
    sys.path.insert(0, ROOT)
    from genomicsbench_amd.srchash import KIND_SOURCE, tu_sha16
    json.dump({"note": __doc__.split("\n\nusage")[0], "rates_cycles_per_wave64_instruction": CYC, "simds": simds, "clock_hz": clk,
               "csrc_sha16": hashes, "tu_sha16": {k: tu_sha16(k) for k in KIND_SOURCE}, "kernels": out}, open(os.path.join(ROOT, "profiles", "valu_mix.json"), "w"), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["valu_static"])[:50]:
        print("%-46s %-18s valu %5d  full %4d half %4d quarter %3d  %.2f cyc  roof %.2e" % (k[:46], v["stage"][:18], v["valu_static"], v["full"], v["half"], v["quarter"], v["mean_cycles_per_valu"], v["roof_lane_ops_per_s"]))


if __name__ == "__main__":
    main()
