#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of scripts/profile_bench.sh into the two tracked files under profiles/:
   <name>_kernel_stats.csv  = rocprofv3's own --stats table (gsn:: kernels first)
   <name>_pmc.csv           = per kernel: mean counter value per dispatch; FETCH/WRITE in KB per dispatch as reported
                              (FETCH_SIZE under-reports wide reads by 2x on gfx950 -- MI355X_MICROARCH.md; not applied here).
usage: summarise_profile.py gpurun_out/<tag> profiles/<name>"""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(d, stem):
    hits = sorted(glob.glob(os.path.join(d, "**", stem), recursive=True))
    return hits[0] if hits else None


def main():
    src, dst = sys.argv[1], sys.argv[2]
    ks = find(src, "r_kernel_stats.csv")
    if ks:
        rows = list(csv.reader(open(ks)))
        head, body = rows[0], rows[1:]
        body.sort(key=lambda r: (0 if "gsn::" in r[0] else 1, -float(r[2])))
        with open(dst + "_kernel_stats.csv", "w", newline="") as f:
            w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
            w.writerow(head)
            w.writerows(body)
    acc = defaultdict(lambda: defaultdict(list))
    for stem in ("p_counter_collection.csv", "q_counter_collection.csv", "f_counter_collection.csv", "w_counter_collection.csv"):
        p = find(src, stem)
        if not p:
            continue
        per_dispatch = defaultdict(float)
        for r in csv.DictReader(open(p)):
            if "gsn::" not in r["Kernel_Name"]:
                continue
            per_dispatch[(r["Kernel_Name"], r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (k, _d, c), v in per_dispatch.items():
            acc[k][c].append(v)
    counters = sorted({c for k in acc for c in acc[k]})
    with open(dst + "_pmc.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "dispatches"] + [c + "_per_dispatch" for c in counters])
        for k in sorted(acc):
            n = max(len(v) for v in acc[k].values())
            w.writerow([k, n] + ["%.6g" % (sum(acc[k][c]) / len(acc[k][c])) if acc[k][c] else "" for c in counters])
    print("wrote", dst + "_kernel_stats.csv", dst + "_pmc.csv")


if __name__ == "__main__":
    main()
