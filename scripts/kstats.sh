#!/bin/bash
# scripts/kstats.sh <file.hip> <kernel-substring> [extra hipcc flags]: registers, spills, LDS and the static instruction mix of one kernel
# (device-only assembly; cross-compiles without a GPU)
set -e
f=$1; k=$2; shift 2
out=/tmp/kstats_$$.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -S --cuda-device-only "$@" gsn_amd/csrc/$f -o $out
python3 - "$out" "$k" <<'PY'
import re, sys
text = open(sys.argv[1]).read()
k = sys.argv[2]
for m in re.finditer(r"\.name:\s+(\S+)\n((?:.*\n)*?)\s+\.wavefront_size", text):
    if k in m.group(1):
        d = dict(re.findall(r"\.(\w+):\s+(\S+)", m.group(2)))
        print(m.group(1)[:90], {x: d.get(x) for x in ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size", "private_segment_fixed_size")})
# static instruction mix of the kernel body
for m in re.finditer(r"^(_Z\w+):.*\n((?:.*\n)*?)\s+s_endpgm", text, re.M):
    if k in m.group(1) and not m.group(1).startswith("."):
        body = m.group(2)
        ins = re.findall(r"^\s+([a-z_0-9]+)", body, re.M)
        from collections import Counter
        c = Counter()
        for i in ins:
            if i.startswith("v_mfma"): c["mfma"] += 1
            elif i.startswith("v_"): c["valu"] += 1
            elif i.startswith("s_"): c["salu"] += 1
            elif i.startswith("ds_"): c["lds"] += 1
            elif i.startswith(("global_", "buffer_", "flat_", "scratch_")): c["vmem" if not i.startswith("scratch_") else "scratch"] += 1
        print("static:", dict(c), "total", len(ins))
        top = Counter(i for i in ins if i.startswith("v_") and not i.startswith("v_mfma"))
        print("top valu:", top.most_common(14))
PY
rm -f $out
