cd ${GRAFT_REPO_ROOT:-.}
for v in w2w6 w2w5 w2w8; do
export GSN_LIB_PATH=gsn_amd/lib/variants/libgsn_hip_$v.so
for st in 2048 4096 6144 8192 12288; do
  echo -n "$v split_target $st: "
  GSN_COUNT_SPLIT_TARGET=$st timeout 300 python scripts/bench_counting_er.py --graphs 2048 --steps 5 --mode vertex 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['graphs_per_s'], d['ms_per_launch'])"
done
done
