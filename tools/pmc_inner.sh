#!/bin/bash
# Run ON THE GPU BOX: counter passes (rocprofv3 --pmc, one pass per counter set, never together with a trace
# domain) of one inner-loop flavour of tools/bench_inner.py, e.g.
#   gpurun -- 'bash tools/pmc_inner.sh "rank1 full" gpurun_out/pmc_rank1'
# Sets (third argument selects a subset): SQ issue/wait, SQ LDS/VMEM, clocks, L2 (TCC) requests/hits, L2<->fabric (EA)
# requests, L1 (TCP) -> L2 requests.  Every pass runs under `timeout`: a counter set the hardware cannot collect makes
# rocprofv3 abort and then wait forever.
set -u
NAME=${1:-general full}
OUT=${2:-gpurun_out/pmc_inner}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
rm -rf "$OUT"; mkdir -p "$OUT"
CMD=(python tools/bench_inner.py --only "$NAME")
# a method name instead of an inner-loop flavour profiles whole calls: bash tools/pmc_inner.sh method:direct_separable out conv_xt
case "$NAME" in method:*) CMD=(python tools/run_method.py "${NAME#method:}");; esac
PAT=${4:-conv_}
SETS=${3:-sq1 sq2 grbm tcc1 tcc2 tcc3 tcp1}
run() { tag=$1; shift; case " $SETS " in *" $tag "*) timeout 150 rocprofv3 --pmc "$@" -d "$OUT/$tag" -o pmc -- "${CMD[@]}" > "$OUT/$tag.log" 2>&1;; esac; }
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY
run tcc1 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum
run tcc2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run tcc3 TCC_WRITE_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUSY_sum
run sq3 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_IFETCH_LEVEL
run sq4 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL
run sqc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES
run tcp1 TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
python tools/pmc_report.py "$OUT" "$PAT" 20 > "$OUT/report.txt" 2>&1
cat "$OUT/report.txt"
