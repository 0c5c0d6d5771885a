#!/bin/bash
# Runs on the GPU box (through gpurun): smoke, GPU parity tests, bench, kernel-trace profile.
# Every GPU step has its own time limit; a fault, abort or time-out ends the script.
set -u
export GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$PWD}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
step() {   # step <seconds> <log> <cmd...>
    local secs=$1 log=$2; shift 2
    echo "== $* (limit ${secs}s)" | tee -a "$GRAFT_REPO_ROOT/gpurun_out/steps.log"
    timeout -k 10 "$secs" "$@" > "$GRAFT_REPO_ROOT/gpurun_out/$log" 2>&1
    local rc=$?
    echo "   exit $rc" | tee -a "$GRAFT_REPO_ROOT/gpurun_out/steps.log"
    case $rc in 124|137|134|139) echo "stopping: step ended abnormally" | tee -a "$GRAFT_REPO_ROOT/gpurun_out/steps.log"; tail -20 "$GRAFT_REPO_ROOT/gpurun_out/$log"; exit $rc;; esac
    return $rc
}
: > gpurun_out/steps.log
step 240 smoke.log python -u -c "import __graft_entry__ as g; g.smoke()" || { tail -30 gpurun_out/smoke.log; exit 1; }
tail -2 gpurun_out/smoke.log
step 600 pytest_gpu.log python -u -m pytest tests -m gpu -q --timeout 180 --timeout-method thread -p no:cacheprovider
tail -25 gpurun_out/pytest_gpu.log
step 300 bench.log python -u bench.py || { tail -30 gpurun_out/bench.log; exit 1; }
tail -3 gpurun_out/bench.log
if [ "${PROFILE:-1}" = "1" ]; then
    cd /tmp
    # headline launches only, so that the kernel's row in the stats table is the headline kernel's
    step 300 rocprof_stats.log rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o bench -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-other-configs
    cd "$GRAFT_REPO_ROOT"
    python3 scripts/headline_profile.py gpurun_out/prof/bench_kernel_trace.csv gpurun_out/rocprof_stats.log gpurun_out/prof/headline_summary.json
fi
