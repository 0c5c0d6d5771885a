#!/bin/bash
# Per-problem patterns in moment form for 12..100 points: in-tree library against other builds.
cd "${GRAFT_REPO_ROOT:-$PWD}"
for lib in "" "$@"; do
    echo "== lib=${lib:-in-tree}"
    PNP_HIP_LIB=${lib:+$PWD/$lib} timeout -k 10 200 python3 -u scripts/diag/lines_crossover.py 2>&1 | grep "^N" | tr "\n" ";" || exit 1
    echo
done
