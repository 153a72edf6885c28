#!/usr/bin/env python3
"""Turns the per-kernel counter means of scripts/pmc_summary.py (gpurun_out/<tag>_<k>_pmc.json, collected with
scripts/pmc.sh: one rocprofv3 --pmc pass per counter group, default 'large' sizes) into the tables bench.py reads:

  profiles/hbm_traffic.json   bytes per launch = FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024 where the kernel's reads
                              are wide coalesced vector loads (poa, phmm stream, chain ring refills), x 1 otherwise:
                              /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE are KB; on gfx950
                              FETCH_SIZE reports half the bytes of 16-byte-per-lane streaming reads, WRITE_SIZE is exact.
                              Both the raw and the corrected figures are kept.
  profiles/valu_busy.json     SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
  profiles/valu_insts.json    SQ_INSTS_VALU per launch (wave instructions; x 64 = lane operations)

usage: make_profile_tables.py <tag>      e.g. r02e      (kernels without a gpurun_out/<tag>_<k>_pmc.json keep their committed entries)
"""
import json
import os
import re
import sys

import hashlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genomicsbench_amd.srchash import tu_sha16  # noqa: E402
CSRC = os.path.join(ROOT, "genomicsbench_amd", "csrc")
KIND_SOURCE = {"bsw": "bsw_kernels.hip", "chain": "chain_kernels.hip", "phmm": "phmm_kernels.hip", "poa": "poa_kernels.hip", "abea": "abea_kernels.hip",
               "fmi": "fmi_kernels.hip"}


def sha16(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
WIDE_READS = ("poa_window", "phmm_stream", "phmm_f32", "phmm_f64", "abea_align")      # kernels whose reads are 16-byte-per-lane vectors


def stage_name(kname):
    """kernel symbol -> the Stage name gbx_profile_end reports (csrc/*.hip)."""
    m = re.match(r"bsw_rows_kernel<(\d+), (\d+), \w+>", kname)
    if m:
        return "bsw_rows_%sx%s" % (m.group(1), m.group(2))
    m = re.match(r"bsw_lane_kernel<\w+, (\w+)(?:, \w+){0,2}>(?:@(\d+))?", kname)
    if m:
        # one symbol, five launches per format: the launch is identified by its grid = resident blocks per CU, which
        # follows from its LDS size (bsw_kernels.hip: bsw_launch; scripts/pmc_summary.py appends it to the name)
        if not m.group(2):
            return "bsw_lane_compact" if m.group(1) == "true" else "bsw_lane_wide"
        per_cu = int(m.group(2))
        if m.group(1) == "true":
            return "bsw_lane_c%d" % {16: 47, 10: 79, 8: 99, 7: 135, 6: 159}.get(per_cu, 0)
        return "bsw_lane_w%d" % {14: 39, 7: 79, 5: 103, 4: 127, 3: 159}.get(per_cu, 0)
    m = re.match(r"phmm_stream_kernel<(\d+)(?:, true)?>", kname)            # (the table form; <N, false> = GBX_PHMM_LUT=0 is not the default and gets no entry)
    if m:
        return "phmm_stream_rpl" + m.group(1)
    if re.match(r"chain_kernel<\d+>", kname):
        return "chain_dp"
    m = re.match(r"phmm_f32_kernel<(\d+)>", kname)
    if m:
        return "phmm_f32_rpl" + m.group(1)
    if kname.startswith("fmi_smem_kernel<"):
        return "fmi_smem"
    if kname.startswith("poa_kernel<"):
        return "poa_window" if "false" in kname else "poa_window_long"
    return {"poa_kernel": "poa_window", "chain_kernel": "chain_dp", "chain_st_kernel": "chain_st", "bsw_lds_kernel": "bsw_lds",
            "bsw_classify_kernel": "bsw_classify", "phmm_f64_kernel<4>": "phmm_f64_redo", "bsw_unpack4_kernel": "bsw_unpack4",
            "abea_kernel": "abea_align"}.get(kname, kname)


def committed(name, key):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name))).get(key, {})
    except (OSError, ValueError):
        return {}



def main():
    tag = sys.argv[1]

    traffic, busy, insts = committed("hbm_traffic.json", "detail"), committed("valu_busy.json", "valu_busy"), committed("valu_insts.json", "valu_insts")
    # which source the counters of a kind were collected on: bench.py drops a kind's figures when its .hip file has changed since
    stamps = committed("hbm_traffic.json", "hip_sha16")
    for k in ("bsw", "chain", "phmm", "poa", "abea", "fmi"):
        path = os.path.join(ROOT, "gpurun_out", "%s_%s_pmc.json" % (tag, k))
        if not os.path.exists(path):
            continue
        table = json.load(open(path))
        # the stamp is the one recorded when the counters were collected (scripts/pmc_summary.py); a file without one (older
        # collections) is only accepted while the source still hashes to what the committed table says
        cur = tu_sha16(k)                                   # every file of the kind's translation unit (genomicsbench_amd/srchash.py)
        rec = table.pop("_hip_sha16", None)
        if rec is None and stamps.get(k) != cur:
            print("skipping %s: no collection-time stamp and the source has changed" % path)
            continue
        stamps[k] = rec or cur
        # entries of this kind carried over from earlier collections describe another source: drop them
        for d in (traffic, busy, insts):
            for name in [n for n in d if n.startswith(k + "_")]:
                del d[name]
        for kname, v in table.items():
            if not kname.startswith(("bsw_", "chain_", "phmm_", "poa_", "abea_", "fmi_")):
                continue
            name = stage_name(kname)
            if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                f, w = v["FETCH_SIZE"] * 1024.0, v["WRITE_SIZE"] * 1024.0
                wide = name.startswith(WIDE_READS)
                traffic[name] = {"fetch_raw": int(f), "write": int(w), "fetch_factor": 2 if wide else 1, "bytes": int(f * (2 if wide else 1) + w)}
            if "valu_busy" in v:
                busy[name] = min(1.0, v["valu_busy"])       # (SQ_ACTIVE_INST_VALU x 4 / SIMD cycles overshoots by a few per cent on saturated kernels: never a fraction above 1)
            if "SQ_INSTS_VALU" in v:
                insts[name] = int(v["SQ_INSTS_VALU"])
    note = ("rocprofv3 --pmc, separate passes per counter group (scripts/pmc.sh), default 'large' sizes, tag %s; units and the gfx950 "
            "FETCH_SIZE correction as /opt/skills/guides/MI355X_MICROARCH.md prescribes (see scripts/make_profile_tables.py)" % tag)
    json.dump({"note": note, "hip_sha16": stamps, "bytes_per_launch": {k: v["bytes"] for k, v in traffic.items()}, "detail": traffic},
              open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
    json.dump({"note": "fraction of SIMD cycles in which the VALU issues; " + note, "hip_sha16": stamps, "valu_busy": busy},
              open(os.path.join(ROOT, "profiles", "valu_busy.json"), "w"), indent=1)
    json.dump({"note": "SQ_INSTS_VALU per launch (wave-level instructions, x 64 lanes = lane operations); " + note, "hip_sha16": stamps, "valu_insts": insts},
              open(os.path.join(ROOT, "profiles", "valu_insts.json"), "w"), indent=1)
    print("kernels: traffic %d, valu_busy %d, valu_insts %d" % (len(traffic), len(busy), len(insts)))


if __name__ == "__main__":
    main()
