#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + HBM-traffic PMC passes of the
# default bench.py command; summaries land in gpurun_out/ and are then committed under profiles/.
#   gpurun -- 'bash tools/profile_bench.sh r02'   (every pass under `timeout`: a failed pass must not hang the box)
set -u
TAG=${1:-r03}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-context"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- $CMD > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.log"
# counters in their own runs (never together with the trace domains)
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- $CMD > /dev/null 2> "$OUT/pmc_fetch.log"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- $CMD > /dev/null 2> "$OUT/pmc_write.log"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/pmc_sq" -o pmc -- $CMD > /dev/null 2> "$OUT/pmc_sq.log"
python tools/rocprof_summary.py "$OUT" "$TAG"
# the north star's "separable-conv inner loop": rank-1 taps only (theta forced to 0), same 4K image
IN="$OUT/inner"; mkdir -p "$IN"
ICMD="python tools/bench_inner.py --only rank1"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$IN/trace" -o inner -- $ICMD > "$IN/bench_under_rocprof.json" 2> "$IN/trace.log"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$IN/pmc_fetch" -o pmc -- $ICMD > /dev/null 2> "$IN/pmc_fetch.log"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$IN/pmc_write" -o pmc -- $ICMD > /dev/null 2> "$IN/pmc_write.log"
python tools/rocprof_summary.py "$IN" "${TAG}_inner_rank1"
ls -R "$OUT" | head -40
