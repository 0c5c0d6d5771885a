#!/bin/bash
# Shared pattern in moment form for 12..100 points: line-cut staging (in-tree) against the
# equal-chunk kernel (A/B build with PNP_MOMENT_PLAIN=1), alternated twice.
cd "${GRAFT_REPO_ROOT:-$PWD}"
for round in 1 2; do
    for arm in lines plain; do
        echo "== $arm $*"
        if [ $arm = plain ]; then export PNP_HIP_LIB=$PWD/pnp_solver_test_amd/libpnp_hip_ab.so PNP_MOMENT_PLAIN=1; else unset PNP_HIP_LIB PNP_MOMENT_PLAIN; fi
        timeout -k 10 200 python3 -u scripts/diag/lines_crossover.py shared "$@" 2>&1 | grep "^N" | cut -d" " -f2,5,7 | tr "\n" ";" || exit 1
        echo
    done
done
