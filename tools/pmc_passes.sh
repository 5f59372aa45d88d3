#!/bin/bash
# rocprofv3 counter passes over tools/prof_forward.py; each pass in its own run (<= 8 SQ counters per pass).
# usage: tools/pmc_passes.sh <outdir> [graphs]
set -u
export TMPDIR=/tmp
OUT=$1; G=${2:-100}
mkdir -p "$OUT"
run() {  # name, counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 tools/prof_forward.py "$G" 1 > "$OUT/$name.log" 2>&1
  echo "$name exit $?"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES
run sq3 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM
run tcp TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_PENDING_STALL_CYCLES
run tcc TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_WRREQ
run grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY
