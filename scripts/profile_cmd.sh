#!/bin/bash
# rocprofv3 kernel-trace + stats, then PMC passes (one counter-block budget per run), over any benchmark command
# (run on the GPU box from the repo root).  Environment variables set by the caller reach the command.
# Usage: scripts/profile_cmd.sh <tag> <passes: "stats sq1 sq2 fetch write grbm tcp tcc"> -- <program> [args...]   -> gpurun_out/prof_<tag>/
set -u
TAG=$1; PASSES=$2; shift; shift; [ "$1" = "--" ] && shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
run() { # name, rocprof args...
    local name=$1; shift
    timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- "${CMD[@]}" > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "pass $name rc=$rc"
    return $rc
}
CMD=("$@")
for P in $PASSES; do
  case $P in
    stats) run stats --kernel-trace --stats || exit 1 ;;
    sq1) run pmc_sq1 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR || exit 1 ;;
    sq2) run pmc_sq2 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE || exit 1 ;;
    sq3) run pmc_sq3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES || exit 1 ;;
    fetch) run pmc_fetch --kernel-trace --pmc FETCH_SIZE || exit 1 ;;
    write) run pmc_write --kernel-trace --pmc WRITE_SIZE || exit 1 ;;
    grbm) run pmc_grbm --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT || exit 1 ;;
    tcp) run pmc_tcp --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum || exit 1 ;;
    tcc) run pmc_tcc --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum || exit 1 ;;
  esac
done
