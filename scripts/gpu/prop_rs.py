"""A/B of the relu-sum propagate mappings (csrc/propagate.hip: relu_sum3_kernel, GSN_PROP_RS = "lpr,unr,nt[,blocks]"; "0" = the generic
kernel) on the bench batch: time by HIP events, bit-identity against the generic kernel, with and without the self term."""
import hashlib, json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from gsn_amd import flags, layers

dev = "cuda"
b = bench.make_batch(int(os.environ.get("G", "65536")), 5)
N, E = b.num_nodes, b.num_edges
ei = torch.from_numpy(b.edge_index).to(dev)
d = int(os.environ.get("D", "300"))
a = torch.randn(N, d, device=dev); bb = torch.randn(E, d, device=dev); c = torch.randn(E, d, device=dev)
eps = torch.full((1,), 0.25, device=dev)
byt = 12.0 * E + 4.0 * (N + 1) + 4.0 * (N * d + 2 * E * d) + 4.0 * N * d
variants = sys.argv[1:] or ["0", "32,1,0", "32,2,0", "32,4,0", "32,2,1", "16,1,0", "16,2,0", "64,1,0", "64,2,0", "64,4,0", "32,2,0,4096", "32,2,0,2048", "32,1,0,4096", "0"]
ref = {}
for v in variants:
    os.environ["GSN_PROP_RS"] = v
    res = {"variant": v}
    for tag, kw in (("plain", {}), ("self", {"selfs": (a,), "eps": eps})):
        f = lambda: layers.propagate(1, ei, 1, N, a=a, b=bb, c=c, **kw)
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
        ref.setdefault(tag, h)
        extra = 4.0 * N * d if tag == "self" else 0.0
        res[tag] = {"ms": round(med, 4), "min": round(ms[0], 4), "hbm_frac": round((byt + extra) / med / 1e6 / 8000.0, 4), "same_bits": h == ref[tag]}
    print(json.dumps(res), flush=True)
