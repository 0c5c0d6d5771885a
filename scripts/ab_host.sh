#!/bin/bash
# A/B on one box: the chunked host-pointer call through the in-tree library and other builds of it.
# usage: ab_host.sh build_variants/libpnp_x.so [...]
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
for lib in "" "$@"; do
    PNP_HIP_LIB=${lib:+$PWD/$lib} timeout -k 10 120 python3 -u scripts/host_lanes.py 2>&1 | grep -E "lib|out=|C ABI" || exit 1
done
