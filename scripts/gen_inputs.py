#!/usr/bin/env python3
"""Writes seeded synthetic datasets in the reference's file formats and directory layout
(<OUT>/{bsw,chain,phmm,poa,fmi}/{small,large}/..., R/scripts/run-cpu.sh:26-74), because the real
input-datasets tarball (R/README.md:18) cannot be downloaded here.

usage: gen_inputs.py <OUT_DIR> <small|large|tiny> [kernels...]       (fmi only when named: its index is built with torch)

fmi: <OUT>/fmi/broad = the index tables (the reference loads a bwa-mem2 index by that prefix, run-cpu.sh:27; ours are
written by genomicsbench_amd.fmi.save_index for a synthetic genome of 64 Mbp (small / large) or 1 Mbp (tiny)), and
<OUT>/fmi/<size>/SRR7733443_{1m,10m}_1.fastq = 151-bp reads sampled from it.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from genomicsbench_amd import io as gio  # noqa: E402
from genomicsbench_amd.datagen import gen_bsw, gen_chain, gen_phmm, gen_poa  # noqa: E402

SIZES = {          # (bsw pairs, chain calls, phmm batches, poa windows), seeds per SURVEY §8d
    "tiny": (2_000, 20, 10, 8),
    "small": (100_000, 1_000, 2_000, 1_000),
    "large": (2_000_000, 10_000, 20_000, 6_000),
}
FILES = {          # names used by run-cpu.sh
    "bsw": {"small": "bandedSWA_SRR7733443_100k_input.txt", "large": "bandedSWA_SRR7733443_1m_input.txt"},
    "chain": {"small": "in-1k.txt", "large": "c_elegans_40x.10k.in"},
    "phmm": {"small": "5m.in", "large": "large.in"},
    "poa": {"small": "input-1000.fasta", "large": "input.fasta"},
}


def main():
    out, size = sys.argv[1], sys.argv[2]
    kernels = sys.argv[3:] or ["bsw", "chain", "phmm", "poa"]
    nb, nc, nph, npo = SIZES[size]
    sub = "small" if size == "tiny" else size
    seed_off = 0 if size == "large" else -1
    if "fmi" in kernels:
        kernels = [k for k in kernels if k != "fmi"]
        from genomicsbench_amd import fmi as FM
        from genomicsbench_amd.datagen import gen_fmi_genome, gen_fmi_reads
        import torch
        g = gen_fmi_genome((1 << 20) if size == "tiny" else (64 << 20), 6001)
        os.makedirs(os.path.join(out, "fmi", sub), exist_ok=True)
        FM.save_index(FM.build_index(g, device="cuda:0" if torch.cuda.is_available() else None), os.path.join(out, "fmi", "broad"))
        n = {"tiny": 2_000, "small": 1_000_000, "large": 10_000_000}[size]
        path = os.path.join(out, "fmi", sub, "SRR7733443_%s_1.fastq" % ("10m" if size == "large" else "1m"))
        FM.write_reads(path, gen_fmi_reads(g, n, 6002 + seed_off), fastq=True)
        print("wrote", path, "and", os.path.join(out, "fmi", "broad"), flush=True)
    for k in kernels:
        d = os.path.join(out, k, sub)
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, FILES[k][sub])
        if k == "bsw":
            gio.write_bsw_pairs(path, gen_bsw(nb, 1002 + seed_off))
        elif k == "chain":
            gio.write_chain_calls(path, *gen_chain(nc, 2001 + seed_off))
        elif k == "phmm":
            gio.write_phmm_batches(path, gen_phmm(nph, 3001 + seed_off))
        else:
            gio.write_poa_windows(path, gen_poa(npo, 4001 + seed_off))
        print("wrote", path, flush=True)


if __name__ == "__main__":
    main()
