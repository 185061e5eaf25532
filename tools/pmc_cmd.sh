#!/bin/bash
# Run ON THE GPU BOX: SQ counter passes of an arbitrary command, reported for kernels matching a pattern:
#   bash tools/pmc_cmd.sh OUTDIR PATTERN MIN_US -- python tools/bench_estimate.py --reps 5
set -u
OUT=$1; PAT=$2; MINUS=$3; shift 4
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
rm -rf "$OUT"; mkdir -p "$OUT"
run() { tag=$1; shift; timeout 200 rocprofv3 --pmc "$@" -d "$OUT/$tag" -o pmc -- "${CMD[@]}" > "$OUT/$tag.log" 2>&1; }
CMD=("$@")
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq3 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_IFETCH_LEVEL
run grbm GRBM_GUI_ACTIVE
python tools/pmc_report.py "$OUT" "$PAT" "$MINUS" > "$OUT/report.txt" 2>&1
cat "$OUT/report.txt"
