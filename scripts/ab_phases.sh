#!/bin/bash
# accumulation pass / iteration split of the N = 68 moment kernel: in-tree library against other builds
cd "${GRAFT_REPO_ROOT:-$PWD}"
for lib in "" "$@"; do
    echo "== lib=${lib:-in-tree}"
    PNP_HIP_LIB=${lib:+$PWD/$lib} timeout -k 10 150 python3 -u scripts/moment_phases.py 2>&1 | grep max_it || exit 1
done
