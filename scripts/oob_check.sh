#!/bin/bash
# Out-of-bounds detector for the GPU suite: with PyTorch's caching allocator off every tensor is its own hipMalloc mapping, so a
# kernel that reads or writes past a tensor faults ("Memory access fault by GPU") instead of silently touching a neighbour inside
# the allocator's block.  One pytest process per file (a fault kills the process).  Run on the GPU box: scripts/oob_check.sh
export PYTORCH_NO_CUDA_MEMORY_CACHING=1
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
rc=0
for f in tests/test_*gpu*.py; do
  r=$(timeout 900 python -m pytest "$f" -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-120)
  echo "$f: $r"
  case "$r" in *failed*|*error*) rc=1 ;; *passed*|*skipped*) ;; *) rc=1 ;; esac      # (a file whose tests all skip -- the capture tests, without the caching allocator -- is not a failure)
done
# the benchmark's launches (headline + supplementary steps; --no-graph: stream capture cannot free memory without the cache) and smoke()
r=$(timeout 1200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-60)
echo "bench.py: $r"
case "$r" in '{"metric"'*) ;; *) rc=1 ;; esac
r=$(python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1)
echo "smoke: $r"
case "$r" in "smoke ok") ;; *) rc=1 ;; esac
exit $rc
