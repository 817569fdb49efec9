"""Switches of the host side in one place: A/B and fault-isolation switches read from the environment at import, and the run-time hooks
(kernel timers, the capture's list of raw-written tensors).  Every reader looks them up at call time (``flags.X``), so tests, scripts and
bench.py set them here: ``from gsn_amd import flags; flags.KERNEL_TIMER = {}``."""
import os

# Optional per-kernel timing hook for bench.py: a dict name -> list of (start_event, end_event, work) recorded on the
# launch stream around every kernel family; None = off (no events, no overhead).
KERNEL_TIMER = None

KERNEL_TIMER_ONLY = None       # a set of family names: only those are bracketed (an event pair per launch is not free: bench.py)

ZERO_ARENA = os.environ.get("GSN_ZERO_ARENA", "1") != "0"      # (0: every request is its own torch.zeros -- A/B and fault isolation)

LINEAR_F16X3 = os.environ.get("GSN_LINEAR_F16X3", "1") != "0"     # direct-row dense stages on the fp16x3 kernel (else bf16x6 / fp32)

LINEAR_F16X3_MIN_N = int(os.environ.get("GSN_LINEAR_F16X3_MIN_N", "128"))

# products of at most this many 128 x 128 output tiles stay on the bf16x6 kernel (its 32-row-tile twin, csrc/linear.hip): one launch of ~10-19 us
# instead of weight split + row pre-pass + product = three launches of ~20-30 us together (molhiv B = 32: 837 x 300 -> 600)
LINEAR_F16X3_MIN_TILES = int(os.environ.get("GSN_LINEAR_F16X3_MIN_TILES", "96"))

# train-mode BatchNorm stages too: the pre-BN rows AND their fp64 column statistics from one launch of the fp16x3 kernel
# (gsn_linear_f16x3_fwd_stats_hip: linear_fwd_bf16_kernel<STATS>'s contract at half its matrix work -- 105 k x 300 -> 600: 0.27 -> 0.18 ms with
# the row pre-pass; rows as accurate as the bf16x6 kernel's against fp64, scripts/gpu/stats_ab.py).  0: those stages stay on the bf16x6 kernel
LINEAR_F16X3_STATS = os.environ.get("GSN_LINEAR_F16X3_STATS", "1") != "0"

# weight gradients of stages whose forward product ran on the fp16x3 kernel: gsn_wgrad_f16x3_hip on the fp16 planes the step already holds (the
# forward product's row scratch of X, kept for the backward pass, and the input-gradient product's row scratch of gH) -- three plane products
# instead of six, no per-tile plane split: 105 k x 300 x 600: 0.38 -> 0.22 ms.  Values more than 2^-17 below the largest of their ROW keep an
# absolute precision of 2^-40 of that largest value instead of fp32's relative one (csrc/wgrad_f16.hip).  0: gsn_wgrad_hip (bf16x6) everywhere
WGRAD_F16X3 = os.environ.get("GSN_WGRAD_F16X3", "1") != "0"
# BatchNorm stages whose gH is read as planes only (input gradient on the fp16x3 kernel + the plane weight gradient): the adjoint pass writes the
# row scratch of gH directly (gsn_bn_act_bwd_planes_hip) -- no fp32 gH, no row pre-pass over it.  0: apply pass + pre-pass
BN_BWD_PLANES = os.environ.get("GSN_BN_BWD_PLANES", "1") != "0"
# the forward twin: a BatchNorm stage whose output only feeds the next product on the fp16x3 kernel writes the row scratch of its output
# (gsn_bn_act_planes_hip): no fp32 stage output, no row pre-pass in front of the next product.  0: gsn_bn_act_hip + pre-pass
BN_ACT_PLANES = os.environ.get("GSN_BN_ACT_PLANES", "1") != "0"
# (a stage whose input gradient is not wanted, or not computed on the fp16x3 kernel: the pre-pass over gH by itself, from this many rows on)
WGRAD_F16X3_SPLIT_ROWS = int(os.environ.get("GSN_WGRAD_F16X3_SPLIT_ROWS", "16384"))

# few-row products with an identity epilogue: K ranges of an output tile on several workgroups (gsn_linear_fwd_splitk_hip; 0: one workgroup per tile)
LINEAR_SPLITK = os.environ.get("GSN_LINEAR_SPLITK", "1") != "0"

STRIDED_WEIGHTS = os.environ.get("GSN_STRIDED_WEIGHTS", "1") != "0"      # transposed weight views read through their strides (0: a contiguous copy first)

VALIDATE_CACHES = os.environ.get("GSN_VALIDATE_CACHES", "0") != "0"   # re-derive-and-compare mode for the per-weight caches (below)

# Asynchronous validation of the per-weight caches (default on; GSN_ASYNC_VALIDATE=0 turns it off).  Behind every eval-mode forward of a
# layer ONE kernel fingerprints the layer's parameters and buffers (gsn_fingerprint_hip); the 8 bytes travel to pinned host memory behind
# it and are looked at -- without waiting -- at the layer's next forward.  A fingerprint that moved while no version counter did is a
# write through `.data`: the layer's caches are dropped there and then and a RuntimeWarning names the layer.  The forward(s) between the
# write and that point used the old derived weights (the check costs no synchronisation; GSN_VALIDATE_CACHES=1 checks BEFORE every
# forward at the price of one); `invalidate_caches` after such a write remains the contract for code that cannot afford one stale call.
ASYNC_VALIDATE = os.environ.get("GSN_ASYNC_VALIDATE", "1") != "0"

# at most one fingerprint per layer and interval (seconds of wall clock): a tight inference loop pays one 5 us launch per layer every 20 ms, not
# one per forward; a `.data` write is then noticed within the interval plus one forward.  0: behind every eval forward.
ASYNC_VALIDATE_INTERVAL = float(os.environ.get("GSN_ASYNC_VALIDATE_INTERVAL", "0.02"))

RAW_WRITTEN = None      # a list while gsn_amd.graphs.GraphedTrainStep captures: tensors that captured kernels write through raw pointers

SPLIT_EDGE_STAGE = os.environ.get("GSN_SPLIT_EDGE", "1") != "0"        # node part of a wide edge Linear once per node (K > SPLIT_EDGE_MIN_K)

SPLIT_EDGE_MIN_K = int(os.environ.get("GSN_SPLIT_EDGE_MIN_K", "160"))

FUSED_LAYER = os.environ.get("GSN_LAYER_FUSED", "1") != "0"   # one-launch `general` layer (gsn_layer_fused_fwd_hip) where it fits

CHAIN_ROW_EXPONENTS = os.environ.get("GSN_CHAIN_ROW_EXP", "1") != "0"      # 128-wide one-launch layers leave their output's row exponents for the next layer

GRAPH_ALIGNED_LAYER = os.environ.get("GSN_LAYER_GRAPHS", "1") != "0"   # d = 128 layers of a collated batch: node products on graph-aligned tiles (csrc/layer_g.hip)

PACK16_LAYER = os.environ.get("GSN_LAYER_PACK16", "1") != "0"   # tagged exact inputs: the packed-row kernel (csrc/layer_rp.hip)

FUSE_BN_ACT_ROWS = int(os.environ.get("GSN_FUSE_BN_ACT_ROWS", "16384"))      # train-mode stages of at most this many rows: finalize + normalise in one launch

NATIVE_DENSE_BACKWARD = True      # False: every mlp backward goes through the PyTorch twin (for comparison)

GATHER_CAT_TRAIN = os.environ.get("GSN_GATHER_CAT_TRAIN", "0") == "1"    # 1: training assembles the edge rows first (gsn_gather_cat_hip), as before r03

FOLD_KERNEL = os.environ.get("GSN_FOLD_KERNEL", "1") != "0"      # A/B switch: the fold as tensor ops over the dense stages (~18 launches per layer and step)

# GNN_OGB with the GSN_edge_sparse_ogb layers (msg = relu(x_j + id_e + e_e)): when the identifier encoder and the edge-feature encoder of a layer
# are both sums of embedding rows (multi_embedding 'sum', BondEncoder) of one width, ONE launch sums the rows of both encoders' tables
# (encoding.embed_columns over the concatenated code columns) and the layer reads one per-edge stream instead of two; one adjoint launch hands
# every table its gradient.  The sum is associated as ((id_0 + ..) + e_0 + ..) instead of (id) + (e): a reassociation of fp32 additions.
FUSE_EDGE_ENCODERS = os.environ.get("GSN_FUSE_EDGE_ENCODERS", "1") != "0"

# relu-sum layers whose own term is their gathered block (the ogb layers): the self term's adjoint inside the node pass of the propagate adjoint
FOLD_SELF_ADJOINT = os.environ.get("GSN_FOLD_SELF_ADJOINT", "1") != "0"

CODE_STATUS_CHECK = True   # read the out-of-range flag back after every code-gather launch (one host sync)
