#!/bin/bash
# Run ON THE GPU BOX: time the inner loop with every ablation build of tools/_abl/ (timing only: results are wrong)
cd "${GRAFT_REPO_ROOT:-.}"
for f in tools/_abl/lib_*.so; do
  cp "$f" polyblur_amd/lib/libpolyblur_hip.so
  echo "== $f: $(python tools/bench_inner.py --only "${1:-general full}" 2>&1 | grep -v amdgpu)"
done
