mkdir -p gpurun_out/r3d
for v in NOMFMA NOLDS NOSTREAM NOGATHER NOSTORE NOLDSSTREAM NOALL; do
  (GSN_LIB_PATH=gsn_amd/lib/variants/libgsn_hip_$v.so timeout 120 python scripts/bench_layer.py --graphs 65536 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['fused'])") > gpurun_out/r3d/$v.log 2>&1
done
(timeout 120 python scripts/bench_layer.py --graphs 65536 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['fused'])") > gpurun_out/r3d/base.log 2>&1
for f in gpurun_out/r3d/*.log; do echo "== $f: $(cat $f | cut -c1-200)"; done
