#!/usr/bin/env python3
"""Soak run of the dense-stage fuzzers with fresh seeds (tests/test_chain_fuzz_gpu.py's generators): random chain shapes at small and
at 20-90 k-row sizes and random wide linear shapes, each against fp64.  Prints every failing case; exit code 1 if any.

    python scripts/soak_dense.py [first_seed] [n_seeds]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_chain_fuzz_gpu as T  # noqa: E402


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda")
    fails = cases = 0
    for seed in range(first, first + n):
        rng = np.random.default_rng(seed)
        for large in (False, True):
            for _ in range(30 if not large else 10):
                out, ref, what = T._case(rng, dev, large=large)
                cases += 1
                err = (out.double() - ref).abs()
                bad = torch.isnan(err) | (err > 1e-5 * ref.abs().max() * (3.0 if what[0] == "seg" else 1.0))
                if out.shape != ref.shape or bool(bad.any()):
                    fails += 1
                    rows = bad.any(1).nonzero().flatten()
                    print("FAIL seed %d large %s %s: %d bad rows %s, max err %.2e" % (seed, large, what, rows.numel(), rows[:6].tolist(),
                                                                                    float(err[~torch.isnan(err)].max() / ref.abs().max())), flush=True)
    print("soak: %d cases, %d failures" % (cases, fails))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
