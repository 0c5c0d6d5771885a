#!/bin/bash
# N = 68 per-problem-pattern launch: in-tree library, the same with the equal-chunk kernel forced,
# and other builds, alternating twice on one box
cd "${GRAFT_REPO_ROOT:-$PWD}"
for rep in 1 2; do
for lib in "" "$@"; do
    echo "== lib=${lib:-in-tree}"
    PNP_HIP_LIB=${lib:+$PWD/$lib} timeout -k 10 150 python3 -u scripts/moment_phases.py 2>&1 | grep "own max_it" || exit 1
done
echo "== in-tree A/B build, equal chunks"
PNP_HIP_LIB=$PWD/pnp_solver_test_amd/libpnp_hip_ab.so PNP_MOMENT_PLAIN=1 timeout -k 10 150 python3 -u scripts/moment_phases.py 2>&1 | grep "own max_it" || exit 1
done
