// The one-launch `general` layer for WIDE node rows on GRAPH-ALIGNED tiles (layer_g.hip): same operation, same prepared weights as
// layer_w.hip, for a collated batch whose graph boundaries the caller knows (every graph <= 128 nodes).  Selected by
// gsn_layer_fused_fwd_graphs_hip (layer_fused.hip).  Not part of the ABI.
#pragma once

#include <hip/hip_runtime.h>

#include "gsn_internal.h"

namespace gsn {

constexpr int G_MAX_GRAPH_NODES = 128;     // nodes of the largest graph the kernel takes (one workgroup tile of whole graphs)

// 1 when the shapes fit (the shapes of layer_w.hip: w_supported)
int g_supported(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0, const gsn_chain_stage *node1);
// forward on the prepared buffer of w_prepare; GSN_OK, an error, or GSN_E_UNSUPPORTED (a graph above G_MAX_GRAPH_NODES, index arrays the
// kernel does not take)
int g_forward(int64_t n_nodes, int64_t n_edges, const int32_t *seg_ptr, const gsn_chain_stage *edge, const float *x, int64_t d_x,
              const gsn_chain_stage *node0, const gsn_chain_stage *node1, const void *prepared, int64_t n_graphs, const int64_t *node_ptr,
              int64_t max_nodes, float *out, hipStream_t st);

}  // namespace gsn
