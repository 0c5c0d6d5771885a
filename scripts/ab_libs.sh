#!/bin/bash
# A/B on one box: the in-tree library against other builds of it (paths relative to the repo
# root), by the forced-iteration cost fit of the headline launch and of the DLT launch.
# usage: ab_libs.sh build_variants/libpnp_x.so [...]
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
for lib in "" "$@"; do
    echo "== lib=${lib:-in-tree}"
    PNP_HIP_LIB=${lib:+$PWD/$lib} timeout -k 10 120 python3 -u scripts/iter_cost.py 2>&1 | grep -E "fit|max_it 5 " || exit 1
    PNP_HIP_LIB=${lib:+$PWD/$lib} timeout -k 10 120 python3 -u scripts/dlt_cost.py 2>&1 | grep -E "fit" || exit 1
done
