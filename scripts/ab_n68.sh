#!/bin/bash
# N = 15 / 32 / 68 launches in both QEIF forms: the in-tree library against other builds of it
cd "${GRAFT_REPO_ROOT:-$PWD}"
for lib in "" "$@"; do
    echo "== lib=${lib:-in-tree}"
    PNP_HIP_LIB=${lib:+$PWD/$lib} timeout -k 10 150 python3 -u scripts/n68_forms.py 2>&1 | grep -A1 -E "moments" | grep -E "moments|kernel_ms" | paste - - || exit 1
done
