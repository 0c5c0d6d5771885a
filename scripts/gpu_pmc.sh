#!/bin/bash
# Counter passes (one rocprofv3 --pmc run each, as the pool requires) over scripts/pmc_probe.py.
set -u
export GRAFT_REPO_ROOT="${GRAFT_REPO_ROOT:-$PWD}"
OUT="$GRAFT_REPO_ROOT/gpurun_out/pmc"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
pass() {   # pass <name> <counters...>
    local name=$1; shift
    echo "== pmc pass $name: $*"
    timeout -s KILL 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o p -- python3 "$GRAFT_REPO_ROOT/scripts/pmc_probe.py" > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "   exit $rc"
    case $rc in 124|137|134|139) tail -5 "$OUT/$name.log"; exit $rc;; esac
}
pass fetch FETCH_SIZE
# read requests by size: FETCH_SIZE tallies a 128-byte request as 64 bytes on this chip, these do not
pass rdreq TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum
pass write WRITE_SIZE
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY
pass sq2 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS
pass grbm GRBM_GUI_ACTIVE
ls "$OUT"
