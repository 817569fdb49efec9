// The one-launch `general` layer for WIDE node rows (d_x = 128: the hidden layers of a d = 128 model, edge rows of K = 256 + <= 16
// columns), layer_w.hip.  Selected inside the gsn_layer_fused_* entry points of layer_fused.hip.  Not part of the ABI.
#pragma once

#include <hip/hip_runtime.h>

#include "gsn_internal.h"
#include "layer_rr_inl.h"

namespace gsn {

// ---- shapes of the prepared buffer (shared with layer_g.hip, which runs on the same fragments) ----
constexpr int W_TN = 32;           // nodes per tile
constexpr int W_UE = 64;           // edge rows per unit (two sub-blocks of 32)
constexpr int W_DX = 128;          // node row width
constexpr int W_NXC = W_DX / 16;   // chunks of a node row
constexpr int W_NST = 2 * W_NXC + 1;   // chunk steps of the edge stage: the per-edge chunk first, then x_i, x_j
constexpr int W_MAXROLE = 3;
constexpr int W_HDR = 32;
constexpr unsigned W_MAGIC = 0x57573031u;
#ifndef W_SD_N
#define W_SD_N 4
#endif
constexpr int W_SD = W_SD_N;       // chunk steps a wave's share of the node-stage fragments is requested ahead of its write to the ring
constexpr int W_PDG = 2;           // gathered chunks in flight ahead of the one being converted

enum { WH_MAGIC = 0, WH_EE = 1, WH_E0 = 2, WH_E1 = 3, WH_EMIN = 4, WH_BAD = 5, WH_ACT = 6 };

struct WShape {
    static constexpr int WB = 4;
    static constexpr int NKS = 2 * WB;                       // chunks of S / of the hidden rows
    static constexpr int NK0 = NKS + W_NXC;                  // chunks of node stage 0 ([S | x]; deg enters with the bias)
    static constexpr int F_WE = 0;                           // edge stage [step][fb][plane]                      (LDS)
    static constexpr int F_LDS = F_WE + W_NST * WB * 2;
    static constexpr int F_W0 = F_LDS;                       // node stage 0 [c][fbo][plane], c < NK0             (streamed)
    static constexpr int F_W1 = F_W0 + NK0 * WB * 2;         // node stage 1 [c][fb][plane]                       (streamed)
    static constexpr int F_WB = F_W1 + NKS * WB * 2;         // node stage 0's (c0, deg column) as bf16 planes [fbo]   (read per tile)
    static constexpr int F_ALL = F_WB + WB;
    static constexpr int N_STREAM = F_WB - F_LDS;
    static constexpr int TAB_WORDS = 4 * 32 * WB;            // c0 of the edge stage (matrix units), c0 of node stage 0, its deg column, c0 of node stage 1
    static constexpr int LDS_BYTES = F_LDS * 1024 + 3 * 8192;     // the edge stage's fragments + a three-slot ring of node-stage chunk steps
    static constexpr int PREP_WORDS = W_HDR + F_ALL * 256 + TAB_WORDS;
};

struct WQuad { unsigned long long base; unsigned stride, role; };


// 1 when the shapes fit the wide kernel
int w_supported(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0, const gsn_chain_stage *node1);
// bytes of its prepared-weights buffer (0 when unsupported); 16-byte aligned
int64_t w_prepared_bytes(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0, const gsn_chain_stage *node1);
int w_prepare(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0, const gsn_chain_stage *node1, void *prepared, hipStream_t st);
// forward; GSN_OK or an error (GSN_E_UNSUPPORTED when this call's arguments are outside the kernel after all)
// `row_exp`: n_nodes ints of caller-owned scratch for the row exponents of x, or null (stream-ordered allocation per call; not under capture)
// `x_row_exp`: the row exponents of x when the caller has them (the pass over x is skipped); `out_row_exp`: those of the output rows, or null
int w_forward(int64_t n_nodes, int64_t n_edges, const int32_t *seg_ptr, const gsn_chain_stage *edge, const float *x, int64_t d_x,
              const gsn_chain_stage *node0, const gsn_chain_stage *node1, const void *prepared, float *out, int32_t *row_exp,
              const int32_t *x_row_exp, int32_t *out_row_exp, hipStream_t st);
// exponent field of max |x[v][:]| per row of 128 floats (255: Inf / NaN in the row)
int w_row_exponents(int64_t n_nodes, const float *x, int32_t *row_exp, hipStream_t st);

}  // namespace gsn
