#!/bin/bash
# A/B on one box: six-point launches whose lanes read their own rows (qeif_rows6_kernel,
# dlt_rows6_kernel) against the LDS-staged kernels (PNP_ROWS_STAGED=1, which only the A/B build of
# the library reads), by the cost fits.
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
export PNP_HIP_LIB="$PWD/pnp_solver_test_amd/libpnp_hip_ab.so"
for staged in 1 0 1 0; do
    echo "== PNP_ROWS_STAGED=$staged"
    PNP_ROWS_STAGED=$staged timeout -k 10 120 python3 -u scripts/iter_cost.py 2>&1 | grep -E "fit|max_it 5 " | tr "\n" ";" || exit 1
    echo
    PNP_ROWS_STAGED=$staged timeout -k 10 120 python3 -u scripts/dlt_cost.py 2>&1 | grep -E "fit|num_it 3 " | tr "\n" ";" || exit 1
    echo
done
