#!/usr/bin/env bash
# The GPU suite with a per-test time limit and a log that is complete up to the moment of a kill (a hanging test shows as the
# last line without a verdict, or as a pytest-timeout stack): scripts/run_gpu_tests.sh <log file> [pytest args...]
log="${1:-gpurun_out/pytest_gpu.log}"; shift || true
mkdir -p "$(dirname "$log")"
PYTHONUNBUFFERED=1 python -m pytest tests -m gpu -v -p no:cacheprovider --timeout "${GBX_TEST_TIMEOUT:-300}" "$@" > "$log" 2>&1
rc=$?
grep -E "passed|failed|error" "$log" | tail -3
exit $rc
