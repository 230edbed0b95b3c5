#!/bin/bash
# rocprofv3 kernel-trace + stats, then PMC passes, over one of the secondary configurations
# (run on the GPU box from the repo root).
# Usage: scripts/profile_config.sh <har|epic|logmel> <tag>      -> gpurun_out/prof_<tag>/
set -u
CFG=${1:-logmel}
TAG=${2:-$CFG}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $ROOT/scripts/bench_configs.py $CFG --iters 10"
run() { # name, rocprof args...
    local name=$1; shift
    timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- $BENCH > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "pass $name rc=$rc"
    return $rc
}
run stats --kernel-trace --stats || exit 1
run pmc_sq1 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR || exit 1
run pmc_sq2 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE || exit 1
run pmc_fetch --kernel-trace --pmc FETCH_SIZE || exit 1
run pmc_write --kernel-trace --pmc WRITE_SIZE || exit 1
run pmc_grbm --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT || exit 1
# matrix-pipe occupancy (the EPIC kernel's float64 MFMA form); a pass of its own, not fatal if a counter is missing
run pmc_mfma --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA || true
