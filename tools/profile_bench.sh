#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + HBM-traffic PMC passes of a bench.py command;
# summaries land in gpurun_out/ and are then committed under profiles/.
#   gpurun -- 'bash tools/profile_bench.sh r04'          the default command (BASELINE config 2) + the rank-1 inner loop
#   gpurun -- 'bash tools/profile_bench.sh r04 cfg3'     another BASELINE config (per-GPU share): <tag>_bench_cfg3_*
# (every pass under `timeout`: a failed pass must not hang the box; counters in their own runs, never with the trace domains)
set -u
TAG=${1:-r05}
CFG=${2:-cfg2}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
if [ "$CFG" = cfg2 ]; then NAME=$TAG; STEPS=10; else NAME=${TAG}_bench_$CFG; STEPS=3; fi
OUT=gpurun_out/$NAME
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python bench.py --config $CFG --steps $STEPS --warmup 2 --no-cpu-baseline --no-context --settle-ms 0"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- $CMD > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.log"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- $CMD > /dev/null 2> "$OUT/pmc_fetch.log"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- $CMD > /dev/null 2> "$OUT/pmc_write.log"
if true; then
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -o pmc -- $CMD > /dev/null 2> "$OUT/pmc_sq.log"
fi
python tools/rocprof_summary.py "$OUT" "$NAME"
if [ "$CFG" = cfg2 ]; then
  # the north star's "separable-conv inner loop": rank-1 taps only (theta forced to 0), same 4K image
  IN="$OUT/inner"; mkdir -p "$IN"
  ICMD="python tools/bench_inner.py --only rank1"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$IN/trace" -o inner -- $ICMD > "$IN/bench_under_rocprof.json" 2> "$IN/trace.log"
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$IN/pmc_fetch" -o pmc -- $ICMD > /dev/null 2> "$IN/pmc_fetch.log"
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$IN/pmc_write" -o pmc -- $ICMD > /dev/null 2> "$IN/pmc_write.log"
  python tools/rocprof_summary.py "$IN" "${TAG}_inner_rank1"
fi
# keep what is small: the summaries, not the raw traces
rm -rf "$OUT"/trace "$OUT"/pmc_fetch "$OUT"/pmc_write "$OUT"/pmc_sq "$OUT"/inner/trace "$OUT"/inner/pmc_fetch "$OUT"/inner/pmc_write
ls -R "$OUT" | head -30
