"""A/B of the pipelined concatenation / scatter-add / readout mappings (csrc/propagate.hip: cat_pipe_kernel, GSN_PROP_CP = "lpr,unr[,blocks]";
"0" = the plain kernel) on the bench batch: time by HIP events, bit-identity against the plain kernel."""
import hashlib, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from gsn_amd import flags, layers

dev = "cuda"
G = int(os.environ.get("G", "65536"))
b = bench.make_batch(G, 5)
N, E = b.num_nodes, b.num_edges
ei = torch.from_numpy(b.edge_index).to(dev)
batch = torch.from_numpy(np.asarray(b.batch).astype(np.int64)).to(dev)
G4 = 4096
n4 = int(b.node_ptr[G4])
batch4 = batch[:n4].contiguous()


def run(f, byt, variants, name):
    ref = None
    for v in variants:
        os.environ["GSN_PROP_CP"] = v
        with torch.no_grad():
            y = f()
            for _ in range(5): f()
            torch.cuda.synchronize()
            flags.KERNEL_TIMER = {}
            for _ in range(20): f()
            torch.cuda.synchronize()
        evs = flags.KERNEL_TIMER.get("propagate_fwd", []); flags.KERNEL_TIMER = None
        ms = sorted(x.elapsed_time(z) for x, z, _ in evs)
        med = ms[len(ms) // 2]
        h = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:12]
        ref = ref or h
        print(json.dumps({"case": name, "variant": v, "ms": round(med, 4), "min": round(ms[0], 4), "hbm_frac": round(byt / med / 1e6 / 8000.0, 4), "same_bits": h == ref}), flush=True)


def edge_case(name, da, db, dc, per_node, variants):
    a = torch.randn(N, da, device=dev) if da else None
    bb = torch.randn(N if per_node else E, db, device=dev) if db else None
    c = torch.randn(E, dc, device=dev) if dc else None
    byt = 12.0 * E + 4.0 * (N + 1) + 4.0 * (N * da + (N if per_node else E) * db + E * dc) + 4.0 * N * (da + db + dc)
    run(lambda: layers.propagate(0, ei, 1, N, a=a, b=bb, c=c, b_per_node=per_node), byt, variants, name)


def pool_case(name, d, bt, n, g, variants):
    x = torch.randn(n, d, device=dev)
    byt = 4.0 * n * d + 4.0 * g * d + 8.0 * n
    run(lambda: layers.global_add_pool_sparse(x, bt, g), byt, variants, name)


which = sys.argv[1:] or ["scatter", "gin", "cat", "glob", "pool"]
if "scatter" in which:
    edge_case("scatter-add of messages d=128 (b only)", 0, 128, 0, False, ["0", "16,1", "16,2", "16,4", "8,1", "8,2", "32,2", "32,4", "16,2,4096", "0"])
if "gin" in which:
    edge_case("gin x_j only d=128", 128, 0, 0, False, ["0", "16,1", "16,2", "16,4", "8,2", "32,2"])
if "cat" in which:
    edge_case("gin cat(x_j, id_e, ef_e) 64+40+8", 64, 40, 8, False, ["0", "16,1", "16,2", "16,4", "8,2", "32,2"])
if "glob" in which:
    edge_case("gin cat(x_j, id_j) global 64+40", 64, 40, 0, True, ["0", "16,2", "16,4", "8,2"])
if "pool" in which:
    pool_case("readout d=300, 4096 graphs", 300, batch4, n4, G4, ["0", "16,1", "16,2", "32,1", "32,2", "32,4", "64,2", "64,4"])
    pool_case("readout d=128, 65536 graphs", 128, batch, N, G, ["0", "16,2", "32,1", "32,2", "32,4", "16,4"])
