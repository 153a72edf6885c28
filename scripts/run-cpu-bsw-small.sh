#!/bin/bash
# run-cpu-bsw-small.sh [INPUTS_DIR] — BASELINE.json config 0: the bsw 'small' input set (100 000 pairs) through the
# reference's own single-thread CPU driver exactly as R/scripts/run-cpu.sh:61 runs it
#     ../benchmarks/bsw/bsw -pairs $INPUTS_DIR/bsw/small/bandedSWA_SRR7733443_100k_input.txt -t 1 -b 512
# (binary: oracle/_ref/bsw_refdriver_cpu = the unmodified main_banded.cpp + bandedSWA.cpp, built by oracle/build_ref.sh
# where /root/reference exists).  Plumbing, no GPU needed; when a GPU is present the MI355X driver runs the same file
# beside it.  The input file is the seeded synthetic one (scripts/gen_inputs.py; the real dataset is not available
# offline).  Prints one JSON line with both timed regions.
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$HERE/.."
INPUTS_DIR=${1:-/tmp/gbx-inputs}
F="$INPUTS_DIR/bsw/small/bandedSWA_SRR7733443_100k_input.txt"
[ -f "$F" ] || python3 "$HERE/gen_inputs.py" "$INPUTS_DIR" small bsw >&2
REFBIN="$ROOT/oracle/_ref/bsw_refdriver_cpu"
cpu_s=null; cpu_cores=1
if [ -x "$REFBIN" ]; then
    # the driver returns 1 by design (main_banded.cpp:352)
    "$REFBIN" -pairs "$F" -t 1 -b 512 > /tmp/gbx-config0-cpu.log 2>&1 || true
    cpu_s=$(sed -n 's/^Overall SW cycles = [0-9]*, \([0-9.]*\) s/\1/p' /tmp/gbx-config0-cpu.log)
    cat /tmp/gbx-config0-cpu.log >&2
else
    echo "oracle/_ref/bsw_refdriver_cpu not built (no /root/reference at build time)" >&2
fi
gpu_ms=null; gpu_first_ms=null
if [ -x "$ROOT/genomicsbench_amd/bin/bsw" ] && python3 -c "import sys; sys.path.insert(0, '$ROOT'); from genomicsbench_amd import _native as N; sys.exit(0 if N.device_count() > 0 else 1)" 2>/dev/null; then
    "$ROOT/genomicsbench_amd/bin/bsw" -pairs "$F" -t 1 -b 512 --repeat 3 > /tmp/gbx-config0-gpu.log 2>&1 || true
    cat /tmp/gbx-config0-gpu.log >&2
    gpu_s=$(sed -n 's/^Overall SW time (H2D + kernels + D2H) = \([0-9.]*\) s/\1/p' /tmp/gbx-config0-gpu.log | head -1)
    [ -n "$gpu_s" ] && gpu_ms=$(python3 -c "print(1e3 * $gpu_s)")
    first_s=$(sed -n 's/^call 0: \([0-9.]*\) s/\1/p' /tmp/gbx-config0-gpu.log | head -1)
    [ -n "$first_s" ] && gpu_first_ms=$(python3 -c "print(1e3 * $first_s)")
fi
cells=$(python3 - "$F" <<'PY'
import sys
n = 0
with open(sys.argv[1]) as f:
    while True:
        h = f.readline()
        if not h:
            break
        t, q = f.readline().rstrip("\n"), f.readline().rstrip("\n")
        n += len(t) * len(q)
print(n)
PY
)
echo "{\"config\": \"bsw small, reference CPU driver -t 1 -b 512 (run-cpu.sh:61)\", \"pairs_file\": \"$F\", \"nominal_cells\": $cells, \"reference_cpu_seconds\": ${cpu_s:-null}, \"reference_cpu_threads\": $cpu_cores, \"gpu_driver_h2d_kernels_d2h_ms\": $gpu_ms, \"gpu_driver_first_call_ms\": $gpu_first_ms}"
