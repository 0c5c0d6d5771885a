#!/bin/bash
# A/B on one box: the in-tree library against other builds of it (paths relative to the repo
# root), by the round-count cost fit of the DLT launch only.
# usage: ab_dlt.sh build_variants/libpnp_x.so [...]
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
for lib in "" "$@"; do
    echo "== lib=${lib:-in-tree}"
    PNP_HIP_LIB=${lib:+$PWD/$lib} timeout -k 10 120 python3 -u scripts/dlt_cost.py 2>&1 | grep -E "fit|num_it 3 " || exit 1
done
