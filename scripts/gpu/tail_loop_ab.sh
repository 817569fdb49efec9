#!/bin/bash
# the tight tail loop on one-word graphs: parity with it forced on, and the counting cases of scripts/bench_configs.py with it off / on / auto
cd ${GRAFT_REPO_ROOT:-.}
GSN_COUNT_TAIL_LOOP=1 timeout 900 python -m pytest tests/test_count_gpu.py tests/test_dataset_gpu.py -q -m gpu 2>&1 | tail -1
for v in 0 1 auto; do
  if [ $v = auto ]; then unset GSN_COUNT_TAIL_LOOP; else export GSN_COUNT_TAIL_LOOP=$v; fi
  echo "tail loop $v:"
  timeout 600 python scripts/bench_configs.py 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for c in d['counting']: print('   ', {k: c[k] for k in c if k in ('case','ms','ms_per_launch','graphs_per_s')} if isinstance(c, dict) else c)
"
done
