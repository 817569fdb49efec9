#!/bin/bash
# r06: counting rows of scripts/bench_configs.py with builds of the library on one box
cd ${GRAFT_REPO_ROOT:-.}
for v in "$@"; do
  if [ $v = default ]; then unset GSN_LIB_PATH; else export GSN_LIB_PATH=gsn_amd/lib/variants/libgsn_hip_$v.so; fi
  echo "== $v"
  timeout 600 python scripts/bench_configs.py 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for r in d['counting']: print('  ', {k: (round(v,3) if isinstance(v,float) else v) for k,v in r.items() if k in ('name','ms','graphs_per_s','ms_per_launch')})"
done
