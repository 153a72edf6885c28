#!/bin/bash
# run-gpu-dp.sh <INPUTS_DIR> <small|large> — the fmi / bsw / phmm / chain / poa lines of the reference's
# scripts/run-cpu.sh (R/scripts/run-cpu.sh:26-43,57-74) on the MI355X drivers.  Same dataset-relative paths; the fmi
# line runs when <INPUTS_DIR>/fmi/broad exists (index tables written by scripts/gen_inputs.py ... fmi).
set -e
if [ $# -ne 2 ]; then echo "usage: run-gpu-dp.sh <INPUTS_DIR> <small|large>"; exit 1; fi
INPUTS_DIR=$1
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
BIN="$HERE/../genomicsbench_amd/bin"
if [ "$2" = "large" ]; then
    if [ -f "$INPUTS_DIR/fmi/broad" ]; then echo "Running fmi"; "$BIN/fmi" "$INPUTS_DIR/fmi/broad" "$INPUTS_DIR/fmi/large/SRR7733443_10m_1.fastq" 512 19 1; fi
    echo "Running bsw";   "$BIN/bsw" -pairs "$INPUTS_DIR/bsw/large/bandedSWA_SRR7733443_1m_input.txt" -t 1 -b 512
    echo "Running phmm";  "$BIN/phmm" -f "$INPUTS_DIR/phmm/large/large.in" -t 1
    echo "Running chain"; "$BIN/chain" -i "$INPUTS_DIR/chain/large/c_elegans_40x.10k.in" -o "$INPUTS_DIR/chain/large/c_elegans_40x.10k.out"
    echo "Running poa";   "$BIN/poa" -s "$INPUTS_DIR/poa/large/input.fasta" -t 1
else
    if [ -f "$INPUTS_DIR/fmi/broad" ]; then echo "Running fmi"; "$BIN/fmi" "$INPUTS_DIR/fmi/broad" "$INPUTS_DIR/fmi/small/SRR7733443_1m_1.fastq" 512 19 1; fi
    echo "Running bsw";   "$BIN/bsw" -pairs "$INPUTS_DIR/bsw/small/bandedSWA_SRR7733443_100k_input.txt" -t 1 -b 512
    echo "Running phmm";  "$BIN/phmm" -f "$INPUTS_DIR/phmm/small/5m.in" -t 1
    echo "Running chain"; "$BIN/chain" -i "$INPUTS_DIR/chain/small/in-1k.txt" -o "$INPUTS_DIR/chain/small/out-1k.txt"
    echo "Running poa";   "$BIN/poa" -s "$INPUTS_DIR/poa/small/input-1000.fasta" -t 1
fi
