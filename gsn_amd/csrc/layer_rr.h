// Register-resident variant of the one-launch `general` layer (layer_rr.hip), selected inside the gsn_layer_fused_* entry points
// of layer_fused.hip for the shapes it covers.  Not part of the ABI.
#pragma once

#include <hip/hip_runtime.h>

#include "gsn_internal.h"

namespace gsn {

int rr_shape_ok(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0, const gsn_chain_stage *node1);
// 1 when the shapes fit the register-resident kernel (a subset of gsn_layer_fused_supported's domain)
int rr_supported(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0, const gsn_chain_stage *node1);
// bytes of its prepared-weights buffer (0 when unsupported); the buffer must be 16-byte aligned
int64_t rr_prepared_bytes(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0, const gsn_chain_stage *node1);
// pack16: the k-slot order of layer_rp.hip (x_i in slots 0..31 with the bias on slot 31, x_j in 32..63, the edge-level columns from 64; the
// in-degree of node stage 0 on slots d_x and d_x + 1)
int rr_prepare(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0, const gsn_chain_stage *node1, void *prepared, hipStream_t st,
               bool pack16 = false);
// forward; returns GSN_OK, an error, or 1 when this call's arguments (index arrays, sizes) are outside the kernel after all
int rr_forward(int64_t n_nodes, int64_t n_edges, const int32_t *seg_ptr, const gsn_chain_stage *edge, const float *x, int64_t d_x,
               const gsn_chain_stage *node0, const gsn_chain_stage *node1, const void *prepared, float *out, hipStream_t st);

// layer_rp.hip: the same layer on exact fp16 row packs (gsn_pack16).  rp_forward returns 1 when the call is outside its 32-bit offsets.
int rp_supported(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0, const gsn_chain_stage *node1);
int rp_forward(int64_t n_nodes, int64_t n_edges, const int32_t *seg_ptr, const gsn_chain_stage *edge, const float *x, int64_t d_x,
               const gsn_chain_stage *node0, const gsn_chain_stage *node1, const void *prepared, const gsn_pack16 *pack, int64_t edge_rows,
               float *out, hipStream_t st);

}  // namespace gsn
