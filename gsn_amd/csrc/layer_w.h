// The one-launch `general` layer for WIDE node rows (d_x = 128: the hidden layers of a d = 128 model, edge rows of K = 256 + <= 16
// columns), layer_w.hip.  Selected inside the gsn_layer_fused_* entry points of layer_fused.hip.  Not part of the ABI.
#pragma once

#include <hip/hip_runtime.h>

#include "gsn_internal.h"

namespace gsn {

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
