"""Diagnostics of GraphedTrainStep: eager run-to-run determinism, first differing step / tensor of replay vs eager, capture errors."""
import os, sys, types, traceback
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import test_graphed_train_gpu as T
from gsn_amd.graphs import GraphedTrainStep

def run_eager(kind, opt_name, n):
    torch.manual_seed(1234); torch.cuda.manual_seed(99)
    model, data, params, opt, loss_of, N, E = T._build(kind, opt_name, dropout=0.0)
    states = []
    for _ in range(n):
        T._eager_step(params, opt, loss_of)
        states.append(T._state(model))
    return states

def run_graph(kind, opt_name, n, warm):
    torch.manual_seed(1234); torch.cuda.manual_seed(99)
    model, data, params, opt, loss_of, N, E = T._build(kind, opt_name, dropout=0.0)
    step = GraphedTrainStep(loss_of, opt, params, warmup=warm)
    states = [None] * (warm - 1) + [T._state(model)]
    for _ in range(n - warm):
        step()
        states.append(T._state(model))
    return states

def cmp(a, b, tag):
    for i, (sa, sb) in enumerate(zip(a, b)):
        if sa is None or sb is None:
            continue
        bad = [(k, float((sa[k].double() - sb[k].double()).abs().max())) for k in sa if not torch.equal(sa[k], sb[k])]
        if bad:
            print(tag, "first difference after step", i + 1, ":", len(bad), "tensors; first", bad[:4]); return
    print(tag, "identical over", len(a), "steps")

for kind in ("zinc", "molhiv"):
    try:
        a = run_eager(kind, "sgd", 5); b = run_eager(kind, "sgd", 5)
        cmp(a, b, kind + " eager vs eager")
        g = run_graph(kind, "sgd", 5, 2)
        cmp(a, g, kind + " eager vs graph")
    except Exception:
        traceback.print_exc()
