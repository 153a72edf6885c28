"""One hash per kernel kind over every source file its translation unit is made of: what the committed counter tables
(profiles/hbm_traffic.json, valu_busy.json, valu_insts.json, valu_mix.json) are stamped with and what bench.py compares
against before it quotes them.  Round 4 stamped the <kind>_kernels.hip file alone, so a change to a header the kernels
include (poa_graph.h, gbx_internal.h) left the tables reading "current"."""
import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
KIND_SOURCE = {"bsw": "bsw_kernels.hip", "chain": "chain_kernels.hip", "phmm": "phmm_kernels.hip", "poa": "poa_kernels.hip",
               "abea": "abea_kernels.hip", "fmi": "fmi_kernels.hip"}
KIND_HEADERS = {"poa": ("poa_graph.h",)}
COMMON_HEADERS = ("gbx_internal.h",)


def tu_files(kind):
    return (KIND_SOURCE[kind],) + KIND_HEADERS.get(kind, ()) + COMMON_HEADERS


def tu_sha16(kind):
    h = hashlib.sha256()
    for f in tu_files(kind):
        h.update(f.encode() + b"\0")
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]
