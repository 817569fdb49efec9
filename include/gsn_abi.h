/*
 * gsn_abi.h -- C ABI of libgsn_hip.so, the MI355X-native (gfx950) replacement for GSN's two data-parallel
 * hot paths.  Plain pointers and sizes only; no torch / C++ types cross this boundary.
 *
 * Each entry point cites the reference interface it replaces (paths relative to gbouritsas/GSN).  The
 * reference's "plugin API" for these paths is plain Python imports; INTEGRATION.md shows the
 * ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - every function returns int: GSN_OK (0) or a negative GSN_E_* code; gsn_last_error() returns a thread-local
 *     message for the last failing call on this thread.  No C++ exception crosses the ABI.
 *   - the library never allocates or frees caller-visible memory: outputs and scratch are caller buffers
 *     (query sizes first).  Device entry points take device pointers plus `stream` (a hipStream_t passed as
 *     void*, e.g. torch.cuda.current_stream().cuda_stream) and are asynchronous on that stream.
 *   - host entry points (pattern analysis, plan building) are synchronous, re-entrant and thread-safe.
 *   - int64 index tensors are accepted as PyTorch hands them over (edge_index is int64 [2, E], row-major).
 */
#ifndef GSN_ABI_H
#define GSN_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSN_ABI_VERSION 1
#define GSN_KMAX 9 /* max pattern vertices the counting kernel handles: --k 8 of star_graph is a 9-vertex pattern (utils.py:59-62) */
#define GSN_SEG_RANGE_ROWS 16 /* rows one thread reduces in the fused scatter-add epilogue (see gsn_segsum_prepare_hip) */

enum {
    GSN_OK = 0,
    GSN_E_INVALID = -1,     /* bad argument */
    GSN_E_UNSUPPORTED = -2, /* valid in the reference but outside this build (e.g. k > GSN_KMAX, n > 768: INTEGRATION.md 11) */
    GSN_E_HIP = -3,         /* HIP runtime error (message has hipGetErrorString) */
    GSN_E_NOSPACE = -4,     /* caller buffer too small */
    GSN_E_NODEVICE = -5     /* no gfx950 device visible */
};

enum { GSN_MODE_VERTEX = 0, GSN_MODE_EDGE = 1 };

/* per-graph status written by gsn_count_hip into `status` (int32 per graph) */
enum {
    GSN_ST_OK = 0,
    GSN_ST_KEYERROR = 1, /* a match uses an edge direction that is not a column of edge_index: the reference raises
                            KeyError at utils_graph_processing.py:173 */
    GSN_ST_TOO_LARGE = 2, /* graph exceeds max_nodes / max_edges given to the call */
    GSN_ST_BAD_INDEX = 3  /* a vertex id outside [0, num_nodes) */
};

const char *gsn_last_error(void);
int gsn_version(void);
/* number of visible gfx950 devices (0 if none / no driver); never fails */
int gsn_device_count(void);
/* 0 when `stream` is not being captured into a HIP graph, else an id unique to the capture (hipStreamGetCaptureInfo); never fails.
 * Host-side scratch that a launch sequence zero-fills is keyed on it (a fill recorded in one capture does not run in another). */
int64_t gsn_stream_capture_id(void *stream);
/* 64-bit content fingerprint of n device tensors, ADDED into *acc (device, caller-zeroed): meta (device int64) = n base pointers, then
 * n sizes in 4-byte words; max_words = the largest of them (sizes the grid).  Order-independent sum of mixed (tensor, position, word)
 * values; one launch.  host_out (PINNED host memory, may be NULL): the value is also copied there asynchronously behind the kernel.
 * The host mirror enqueues it behind a layer forward and compares the value at the next forward (a parameter
 * written through `.data` does not move PyTorch's version counter, on which the derived-weight caches are keyed). */
int gsn_fingerprint_hip(int n_tensors, const int64_t *meta, int64_t max_words, unsigned long long *acc, unsigned long long *host_out,
                        void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * HP-1  pattern analysis (host).  Replaces utils_graph_processing.automorphism_orbits (:10-56) and
 * induced_edge_automorphism_orbits (:58-100): vertex orbits of Aut(H) numbered by rank of the orbit's smallest
 * vertex; the pattern's directed edges sorted by (u,v); edge-orbit ids in first-seen order of the key
 * {orbit(u),orbit(v)} (ordered pair iff directed_orbits); |Aut(H)|.
 *   directed_orbits  bit 0 (GSN_FLAG_DIRECTED_ORBITS): the reference's `directed_orbits`;  bit 1 (GSN_FLAG_DIRECTED): the
 *                    reference's `directed` (main.py:558, utils_graph_processing.py:14-16) -- every row (u,v) of `edges` is the
 *                    ARC u -> v, automorphisms preserve arcs, out_arcs lists the arcs themselves
 *   edges            [n_edges][2] int64, vertices 0..k-1; self loops / duplicates are dropped like the reference does
 *   out_vertex_orbit [GSN_KMAX]
 *   out_arcs         [k*(k-1)][2]  sorted directed edge list;  out_arc_orbit [k*(k-1)]
 * ---------------------------------------------------------------------------------------------------------------- */
#define GSN_FLAG_DIRECTED_ORBITS 1
#define GSN_FLAG_DIRECTED 2
int gsn_pattern_orbits(int64_t n_edges, const int64_t *edges, int directed_orbits, int64_t *out_k,
                       int64_t *out_vertex_orbit, int64_t *out_n_vertex_orbits, int64_t *out_arcs,
                       int64_t *out_arc_orbit, int64_t *out_n_arcs, int64_t *out_n_edge_orbits,
                       int64_t *out_aut_count);

/* ------------------------------------------------------------------------------------------------------------------
 * HP-1  vertex orbits of an arbitrary small graph (host).  The automorphism part of the reference's deprecated
 * edge_automorphism_orbits (utils_graph_processing.py:189-251, --edge_automorphism line_graph): there the graph is the
 * LINE graph of the pattern and graph-tool enumerates its automorphisms (:214-224); here orbits come from
 * "is there an automorphism with sigma(u) = v" searches, numbered like np.unique(..., return_inverse) numbers the
 * per-vertex orbit minima (:231): orbit id = rank of the orbit's smallest vertex.
 *   edges      [n_edges][2] int64, vertices 0..n_vertices-1 (n_vertices <= 64); self loops / duplicates dropped
 *   out_orbit  [n_vertices]
 * ---------------------------------------------------------------------------------------------------------------- */
int gsn_graph_vertex_orbits(int64_t n_vertices, int64_t n_edges, const int64_t *edges, int64_t *out_orbit,
                            int64_t *out_n_orbits);

/* ------------------------------------------------------------------------------------------------------------------
 * HP-1  counting plan (host).  Compiles a list of patterns (the reference's `subgraph_dicts`,
 * utils_data_gen.py:31-42) into the packed table the kernel executes: one rooted search per (pattern, vertex orbit)
 * [vertex mode] or per (pattern, directed-edge orbit) [edge mode], with symmetry-breaking order constraints for the
 * root's stabiliser.  Output columns follow utils_ids.py:19-25: patterns in the given order, orbit ids ascending.
 *   directed_orbits  flag bits as for gsn_pattern_orbits.  GSN_FLAG_DIRECTED (vertex mode only; edge mode ->
 *           GSN_E_UNSUPPORTED, the reference's directed edge counter dies on an unbound name, utils_graph_processing.py:146 vs
 *           :164): patterns AND the graphs later counted with this plan are digraphs (gt.Graph(directed=True), :108-113) -- a
 *           column (u,v) of edge_index is the arc u -> v, matches preserve arcs and, induced, non-arcs in each direction
 *   pat_ptr [n_patterns+1] into pat_edges [.][2]
 *   plan    caller buffer of `capacity` uint32 words (call with plan=NULL to get *out_words)
 * ---------------------------------------------------------------------------------------------------------------- */
int gsn_count_plan_build(int mode, int induced, int directed_orbits, int64_t n_patterns, const int64_t *pat_ptr,
                         const int64_t *pat_edges, uint32_t *plan, int64_t capacity, int64_t *out_words,
                         int64_t *out_n_cols);

/* ------------------------------------------------------------------------------------------------------------------
 * HP-1  batched counting (device).  Replaces utils_ids.subgraph_counts2ids (:7-29) applied to every graph of a batch,
 * i.e. the per-graph / per-pattern calls of subgraph_isomorphism_vertex_counts (utils_graph_processing.py:103-131) or
 * subgraph_isomorphism_edge_counts (:134-179) and the graph-tool VF2 enumeration under them.  One launch per batch.
 *   plan_host / plan_dev  the same plan table on host (launch configuration) and in device memory (kernel input)
 *   node_ptr [G+1], edge_ptr [G+1]   int64 device: graph g owns vertices node_ptr[g]..node_ptr[g+1] and columns
 *                                    edge_ptr[g]..edge_ptr[g+1] of edge_index
 *   edge_index            int64 device [2][E_total] (row 1 starts at edge_index + edge_row_stride); both directions
 *                         present as in PyG; self loops tolerated (their rows stay 0); duplicates: last column wins
 *   ids_are_global        1: vertex ids carry the batch offset node_ptr[g] (PyG collate), 0: graph-local ids
 *   graph_ids             optional int32 device [n_items]: process only these graphs (NULL = all G, n_items = G)
 *   max_nodes, max_edges  upper bounds over the processed graphs (sizes LDS); violators get GSN_ST_TOO_LARGE
 *   out                   int64 device [rows_total][n_cols] (rows = vertices or columns); fully overwritten for the
 *                         processed graphs, zeros included
 *   status                int32 device [G], written per processed graph (GSN_ST_*)
 * ---------------------------------------------------------------------------------------------------------------- */
int gsn_count_hip(const uint32_t *plan_host, const uint32_t *plan_dev, int64_t plan_words, int64_t n_graphs,
                  const int64_t *node_ptr, const int64_t *edge_ptr, const int64_t *edge_index,
                  int64_t edge_row_stride, int ids_are_global, const int32_t *graph_ids, int64_t n_items,
                  int64_t max_nodes, int64_t max_edges, int64_t *out, int32_t *status, void *stream);

/* Counting with the identifier encoding fused into the kernel's output: what the reference does in two steps -- int64 counts
 * (utils_ids.py:7-29), then DiscreteEmbedding('one_hot_encoder') on them in front of every GSN layer
 * (utils_graph_learning.py:78 / :170-187, models_graph_classification.py:222) -- in one launch, without the int64 rows going
 * to HBM and back and without the gsn_one_hot_hip launch.  Arguments as gsn_count_hip, plus
 *   n_classes  HOST int32 [n_cols] (as gsn_one_hot_hip; n_cols <= 64): column c becomes n_classes[c] floats, blocks in column order
 *   clamp      != 0: counts above n_classes[c] - 1 go to the last class (as gsn_one_hot_hip)
 *   enc_out    fp32 device [rows_total][sum n_classes]: per column a single 1 at the count (all 0 when the count is out of
 *              range and clamp == 0); fully overwritten for the processed graphs
 *   out        the int64 rows as well, or NULL (then only the encoded rows are written) */
int gsn_count_encode_hip(const uint32_t *plan_host, const uint32_t *plan_dev, int64_t plan_words, int64_t n_graphs,
                         const int64_t *node_ptr, const int64_t *edge_ptr, const int64_t *edge_index,
                         int64_t edge_row_stride, int ids_are_global, const int32_t *graph_ids, int64_t n_items,
                         int64_t max_nodes, int64_t max_edges, int64_t *out, int32_t *status, const int32_t *n_classes,
                         int clamp, float *enc_out, void *stream);
/* The same, and the encoded rows a second time as fp16 into columns pack_col0 .. of an exact row pack (gsn_pack16 below: the layout the
 * packed-row layer kernel reads; 1.0 = 0x3c00): `pack` fp16 device [rows_total][pack_stride], only the sum n_classes columns from
 * pack_col0 are written.  Written from the kernel's staged class indices: GSN_E_UNSUPPORTED when those are not staged (a graph split
 * over several workgroups, a column with more than 255 classes) -- pack the fp32 rows with gsn_pack16_rows_hip then.
 * enc_out may be NULL (r05): the fp32 rows are then not written at all -- the int64 counts (`out`) and the pack columns are what leaves the
 * kernel: what utils_ids.py:27 stores, and the encoder output (utils_graph_learning.py:170-187) in the form the packed-row layer reads. */
int gsn_count_encode_pack16_hip(const uint32_t *plan_host, const uint32_t *plan_dev, int64_t plan_words, int64_t n_graphs,
                                const int64_t *node_ptr, const int64_t *edge_ptr, const int64_t *edge_index,
                                int64_t edge_row_stride, int ids_are_global, const int32_t *graph_ids, int64_t n_items,
                                int64_t max_nodes, int64_t max_edges, int64_t *out, int32_t *status, const int32_t *n_classes,
                                int clamp, float *enc_out, uint16_t *pack, int64_t pack_stride, int64_t pack_col0, void *stream);

/* The same launch leaving what the first GSN layer needs beside the identifiers (r06).  Every fifth workgroup of the grid is a SIDE workgroup:
 * it takes the graphs of the four counting workgroups dispatched in front of it, sorts their columns in LDS and writes
 *   (i)   the target-sorted CSR of its graph's columns -- seg_ptr / perm / sorted_target / sorted_other exactly as
 *         gsn_csr_build_graphs_hip writes them (GSN_sparse.py:140-143: the reference re-sorts a COO tensor in every layer);
 *   (ii)  the node pack: one-hot of integer node codes (utils_graph_learning.py:170-187: DiscreteEmbedding('one_hot_encoder') on
 *         data.x) as fp16 [n_nodes][32], column 31 = 1.0 -- what gsn_one_hot_pack16_hip(col0 = 0, one_col = 31) writes;
 *   (iii) columns edge_col0 .. 15 of the edge pack fp16 [n_edges][16]: the one-hot of integer edge codes at edge_col0 .., zeros behind it (the
 *         counting workgroups write the identifier classes at pack_col0 .. as in gsn_count_encode_pack16_hip; with pack_col0 = 0 and
 *         edge_col0 = sum n_classes the whole row is written by the launch and the pack needs no zero fill).
 * Each part is optional (null pointers).  The counting workgroups run exactly what they run without side outputs (the work inside them, where
 * the graph already sits in LDS, was measured and lost: the molecule instantiation is at its register bound, profiles/r06_count_side_ab.txt).
 * Needs one workgroup per item (no graph list, no graph split over several workgroups) and graphs whose sort fits LDS, edge-mode or vertex-mode
 * plan alike; GSN_E_UNSUPPORTED otherwise, nothing launched -- the caller uses the separate entry points.
 * node_ptr[0] must be 0, node_ptr[n_graphs] = n_nodes, edge_ptr[n_graphs] = n_edges (a collated batch).  Statuses: a column that leaves
 * its graph raises GSN_ST_BAD_INDEX on that graph (its CSR entries are then unspecified but inside the batch's ranges), a graph
 * beyond max_nodes / max_edges GSN_ST_TOO_LARGE (sorted all the same when it fits the launch's LDS, else its vertices own no columns and
 * its columns map to themselves: as gsn_csr_build_graphs_hip); *code_status (device int32, caller-zeroed, may be NULL) gets 1 ORed in when a code lies outside its
 * column's classes and clamp is 0 (that column's segment stays zero, as gsn_one_hot_pack16_hip). */
typedef struct {
    int csr_row;                 /* which row of edge_index is the aggregation target (1: flow = source_to_target) */
    int32_t *seg_ptr;            /* [n_nodes + 1] or NULL: no CSR */
    int32_t *perm;               /* [n_edges] */
    int32_t *sorted_target;      /* [n_edges] or NULL */
    int32_t *sorted_other;       /* [n_edges] or NULL */
    int64_t n_nodes, n_edges;    /* batch totals */
    const int64_t *node_codes;   /* [n_nodes][node_code_cols] or NULL: no node pack */
    int node_code_cols;          /* 1 .. 4 */
    int node_n_classes[4];       /* sum <= 28 */
    int node_clamp;
    uint16_t *node_pack;         /* fp16 [n_nodes][32], 16-byte aligned */
    const int64_t *edge_codes;   /* [n_edges][edge_code_cols] or NULL: only the identifier columns of the edge pack are written */
    int edge_code_cols;          /* 1 .. 4 */
    int edge_n_classes[4];
    int edge_clamp;
    int edge_col0;               /* first pack column of the edge codes' one-hot: a multiple of 4, behind the identifier columns; <= 8 classes */
    int32_t *code_status;        /* or NULL */
} gsn_count_side;
int gsn_count_encode_pack16_side_hip(const uint32_t *plan_host, const uint32_t *plan_dev, int64_t plan_words, int64_t n_graphs,
                                     const int64_t *node_ptr, const int64_t *edge_ptr, const int64_t *edge_index,
                                     int64_t edge_row_stride, int ids_are_global, int64_t max_nodes, int64_t max_edges, int64_t *out,
                                     int32_t *status, const int32_t *n_classes, int clamp, uint16_t *pack, int64_t pack_stride,
                                     int64_t pack_col0, const gsn_count_side *side, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * HP-2  aggregation target index (device).  The scatter-add of the layers,
 *   torch.sparse.FloatTensor(edge_index, msgs, [N,N,d]) + torch.sparse.sum(msgs, aggr_dim).to_dense()
 * (GSN_sparse.py:140-143, GSN_edge_sparse.py:136-139, MPNN_*.py, *_ogb.py), is executed as a segmented sum over a
 * target-sorted edge permutation.  This builds that CSR once per batch (stable counting sort by target):
 *   index   int64 device [E]  aggregation targets (edge_index[1] for flow=source_to_target, [0] otherwise)
 *   seg_ptr int32 device [N+1] out; perm int32 device [E] out (edge ids grouped by target, original order kept inside)
 *   sorted_target int32 device [E] out or NULL: sorted_target[q] = index[perm[q]] (the target of the q-th sorted edge)
 *   other int64 device [E] or NULL (the other row of edge_index), sorted_other int32 device [E] out or NULL:
 *         sorted_other[q] = other[perm[q]] (the message source of the q-th sorted edge) -- lets a kernel that walks the
 *         edges in target order gather x_i / x_j / per-edge rows through ONE index load each
 *   scratch int32 device [gsn_csr_scratch_elems(N)]
 * ---------------------------------------------------------------------------------------------------------------- */
int64_t gsn_csr_scratch_elems(int64_t n_nodes);
int gsn_csr_build_hip(int64_t n_nodes, int64_t n_edges, const int64_t *index, const int64_t *other, int32_t *seg_ptr,
                      int32_t *perm, int32_t *sorted_target, int32_t *sorted_other, int32_t *scratch, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * HP-2  propagate: fused gather -> message -> segmented sum (device, fp32).
 *   out[t, :] = sum_{e : tgt(e) = t} msg_e,   msg_e assembled on the fly from up to three sources
 *     kind GSN_MSG_CAT  : msg_e = concat(a[src_e] (da), b_e or b[src_e] (db), c_e (dc))   -- 'gin' messages
 *                         (GSN_sparse.py:160-164, GSN_edge_sparse.py:154-158) and plain pre-computed messages (da=0,dc=0)
 *     kind GSN_MSG_RELU_SUM : msg_e = relu(a[src_e] + b_e or b[src_e] + c_e), all width d  -- 'ogb' messages
 *                         (GSN_edge_sparse_ogb.py:119-125, MPNN_edge_sparse_ogb.py message)
 *   a: per-node [N][da];  b: per-edge [E][db] (b_per_node=0) or per-node [N][db] gathered at src (b_per_node=1);
 *   c: per-edge [E][dc];  any of a/b/c may be NULL with width 0.   out: [N][d_out], fully overwritten.
 *   src int64 [E] message source vertex (edge_index[0] for source_to_target); seg_ptr/perm from gsn_csr_build_hip;
 *   sorted_src int32 [E] = src[perm[q]] (gsn_csr_build_hip's sorted_other) or NULL -- saves one dependent load per edge.
 * gsn_propagate_bwd_hip is the adjoint: given g_out [N][d_out] it writes g_a [N][da] (overwritten, gathers through the
 * source-sorted CSR seg_ptr_src/perm_src), g_b ([E][db] or [N][db]) and g_c [E][dc]; for RELU_SUM the forward inputs
 * are needed again to recompute the relu mask.
 * ---------------------------------------------------------------------------------------------------------------- */
enum { GSN_MSG_CAT = 0, GSN_MSG_RELU_SUM = 1 };

int gsn_propagate_fwd_hip(int kind, int64_t n_nodes, int64_t n_edges, const int64_t *src, const int32_t *seg_ptr,
                          const int32_t *perm, const int32_t *sorted_src, const float *a, int64_t da, const float *b, int64_t db,
                          int b_per_node, const float *c, int64_t dc, float *out, void *stream);

/* out[t][0 .. width) = sum over the rows q of t's segment of b[row(q)][0 .. width), where b is a COLUMN SLICE of wider rows (row stride ld
 * floats; row(q) = perm[q], or q when perm is NULL): gsn_propagate_fwd_hip's concatenation with the per-edge block alone, reading the slice in
 * place -- the per-vertex sums of a gathered block's input gradient (the x_i / x_j blocks of an edge stage, GSN_sparse.py:166-171). */
int gsn_segment_sum_rows_hip(int64_t n_nodes, int64_t n_edges, const int64_t *src, const int32_t *seg_ptr, const int32_t *perm,
                             const int32_t *sorted_src, const float *b, int64_t width, int64_t ld, float *out, void *stream);

int gsn_propagate_bwd_hip(int kind, int64_t n_nodes, int64_t n_edges, const int64_t *src, const int64_t *tgt,
                          const int32_t *seg_ptr_src, const int32_t *perm_src, const float *a, int64_t da,
                          const float *b, int64_t db, int b_per_node, const float *c, int64_t dc,
                          const float *g_out, float *g_a, float *g_b, float *g_c, void *stream);

/* The same pass with the layer's own term and the central encoders' column padding inside it (r03; what the reference does with
 * three elementwise tensor ops and two concatenations per layer):
 *     out[t] = (1 + *eps) * self[t] + sum_{e -> t} msg_e
 *   GSN_sparse.py:157-163 / GSN_edge_sparse.py:95-109 (gin):   self = cat(x, identifiers or their central value, central edge value);
 *   GSN_edge_sparse_ogb.py:63-84, :103-106 (ogb):              self = x (+ identifiers, global scope).
 * self_blocks: up to three blocks, concatenated (CAT: their widths add up to d_out) or added (RELU_SUM: each d_out wide); row_stride
 * in floats, 0 = ONE row for every vertex (central_encoder's constant value, utils_graph_learning.py:232-260).  eps: device pointer
 * to the layer's eps (parameter or buffer), NULL = 0.  pad_b / pad_c: zero columns in front of the per-edge blocks b / c of a
 * concatenation (utils_graph_learning.py:240-242: the extra first column of an extended one-hot encoding), so that
 * d_out = da + pad_b + db + pad_c + dc.  gsn_propagate_pad_bwd_hip: the adjoint of the messages with the same column layout;
 * gsn_propagate_self_bwd_hip: the adjoint of the self term.
 */
typedef struct gsn_self_block {
    const float *data;
    int64_t width;
    int64_t row_stride;
} gsn_self_block;

int gsn_propagate_self_fwd_hip(int kind, int64_t n_nodes, int64_t n_edges, const int64_t *src, const int32_t *seg_ptr,
                               const int32_t *perm, const int32_t *sorted_src, const float *a, int64_t da, const float *b, int64_t db,
                               int b_per_node, const float *c, int64_t dc, int64_t pad_b, int64_t pad_c, int n_self,
                               const gsn_self_block *self_blocks, const float *eps, float *out, void *stream);

/* Adjoint of the self term (one pass over g_out [N][d_out]): g_self[k] (fp32 [N][width_k], NULL = not wanted / single-row block)
 * receives (1 + eps) * the block's columns of g_out; g_colsum (fp64 [d_out], zero-filled, may be NULL) the column sums of
 * (1 + eps) * g_out -- the gradient of a single-row block is its slice; g_eps (fp64 [1], zero-filled, may be NULL) sum g_out . self. */
int gsn_propagate_self_bwd_hip(int kind, int64_t n_nodes, int64_t d_out, const float *g_out, int n_self,
                               const gsn_self_block *self_blocks, float *const *g_self, const float *eps, double *g_eps,
                               double *g_colsum, void *stream);

int gsn_propagate_pad_bwd_hip(int kind, int64_t n_nodes, int64_t n_edges, const int64_t *src, const int64_t *tgt,
                              const int32_t *seg_ptr_src, const int32_t *perm_src, const float *a, int64_t da,
                              const float *b, int64_t db, int b_per_node, const float *c, int64_t dc, int64_t pad_b, int64_t pad_c,
                              const float *g_out, float *g_a, float *g_b, float *g_c, void *stream);
/* The relu-sum adjoint of a layer whose own term is its gathered block (out = (1 + eps) a + sum relu(a[src] + b + c): GSN_edge_sparse_ogb.py:63-84,
 * :103-106 with self = x) in TWO launches: g_a receives the per-source sums of the masked per-edge gradients AND (1 + eps) g_out, g_eps (fp64, added
 * to; may be NULL) = sum g_out . a.  Replaces gsn_propagate_pad_bwd_hip + gsn_propagate_self_bwd_hip + the sum of their results.  b / c per edge,
 * widths d (or 0).  GSN_E_UNSUPPORTED (nothing launched): no edges, d > 320 or not a multiple of 4, unaligned rows, no per-edge gradient wanted. */
int gsn_propagate_bwd_fold_self_hip(int64_t n_nodes, int64_t n_edges, const int64_t *src, const int64_t *tgt, const int32_t *seg_ptr_src,
                                    const int32_t *perm_src, const float *a, int64_t d, const float *b, int64_t db, const float *c, int64_t dc,
                                    const float *g_out, float *g_a, float *g_b, float *g_c, const float *eps, double *g_eps, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * HP-2  fused dense stage (device; fp32 in / fp32 out; matrix products run as six exact bf16 plane products per fp32
 * product on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- fp32-equivalent error -- or, with GSN_LINEAR_BF16X6=0,
 * on v_mfma_f32_32x32x2_f32).  One models_misc.mlp layer
 * (models_misc.py:52-58)   Y = act( bn( X W^T + bias ) )   over M rows, where the rows of X are assembled on the fly
 * as a concatenation of up to five blocks, each either direct ([M][w]) or gathered through an int64 index
 * ([R][w] rows picked by idx[M]) -- this is torch.cat((x_i, x_j, identifiers.., edge_features), -1) feeding msg_fn
 * (GSN_sparse.py:166-171, GSN_edge_sparse.py:160-165) without materialising the cat or the gathers.
 *   W [n_out][k_total] row-major fp32 as nn.Linear stores it (k_total = sum of block widths); bias [n_out] or NULL
 *   bn_mean / bn_scale / bn_shift [n_out], all three or none:  y = (h - mean) * scale + shift  with
 *       scale = gamma / sqrt(var + eps), shift = beta  (eval: running stats; train: the batch stats of pass 1)
 *   act: 0 identity, 1 relu, 2 elu, 3 tanh   (models_misc.choose_activation)
 *   row_perm int32 [M] or NULL: output row r is computed from logical input row row_perm[r]
 *   out   [M][n_out] fp32, or NULL when only statistics are wanted
 *   stats double [2][n_out] or NULL: if given (train-mode BatchNorm1d, first pass) the kernel ADDS per-column sum and
 *         sum of squares of the PRE-BN values h = X W^T + bias into it (caller zeroes it) and skips bn / act; with
 *         `out` also given the same pass writes those raw h rows to it (then apply gsn_bn_act_hip)
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct {
    const float *data;     /* [rows][width] fp32, row stride = width */
    const int64_t *idx;    /* int64 gather index (a row of edge_index) or NULL */
    int64_t width;
    const int32_t *idx32;  /* int32 gather index (perm / sorted_target / sorted_source of gsn_csr_build_hip) or NULL;
                              at most one of idx / idx32; both NULL: direct (row r) */
} gsn_block;

int gsn_linear_fwd_hip(int64_t m_rows, int n_blocks, const gsn_block *blocks, const float *W, const float *bias,
                       int64_t n_out, const float *bn_mean, const float *bn_scale, const float *bn_shift, int act,
                       const int32_t *row_perm, float *out, double *stats, void *stream);
/* gsn_linear_fwd_hip with W given through element strides (W[j][k] at W + j * w_row_stride + k * w_col_stride): the input-gradient product
 * gX = gH W of a dense stage reads the stage's own weight as its transpose (1, K) -- no transposed copy per stage and step.  No row_perm.
 * GSN_E_UNSUPPORTED when the bf16x6 kernel is switched off (GSN_LINEAR_BF16X6=0). */
int gsn_linear_fwd_strided_hip(int64_t m_rows, int n_blocks, const gsn_block *blocks, const float *W, int64_t w_row_stride,
                               int64_t w_col_stride, const float *bias, int64_t n_out, const float *bn_mean, const float *bn_scale,
                               const float *bn_shift, int act, float *out, double *stats, void *stream);

/* Split-K form of the two entry points above for FEW rows (a dense backward at the reference's batch sizes, train_test_funcs.py:88-106 with
 * README.md:112,121's batch sizes: M = 10^3 rows is a third of the chip's CUs, each walking all K slices): gsn_linear_splitk_plan gives the
 * number of K ranges this library would use for the shape (1: take gsn_linear_fwd_hip / _strided_hip); gsn_linear_fwd_splitk_hip ADDS
 * blocks W^T (+ bias) to `out`, which must hold zeros on entry (float atomics: the order of the ranges' partial sums varies in the last bits).
 * Identity epilogue only.  w_row_stride = w_col_stride = 0: W is row-major [n_out][K]. */
int gsn_linear_splitk_plan(int64_t m_rows, int64_t k_total, int64_t n_out);
int gsn_linear_fwd_splitk_hip(int64_t m_rows, int n_blocks, const gsn_block *blocks, const float *W, int64_t w_row_stride,
                              int64_t w_col_stride, const float *bias, int64_t n_out, float *out, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * HP-2  fused MLP chain (device; fp32 in / fp32 out; matrix products as in gsn_linear_fwd_hip: bf16x6 by default,
 * GSN_CHAIN_BF16X6=0 for the fp32-MFMA kernels).  One or two dependent stages of models_misc.mlp (models_misc.py:52-58)
 * evaluated per 32- or 64-row tile without writing the intermediates to HBM:
 *     Y_s = act_s( bn_s( [ blocks_s | Y_{s-1} ] W_s^T + bias_s ) ),   out = Y_last   ([M][n_out_last])
 * i.e. the input of stage s > 0 is the concatenation of its own HBM blocks (e.g. x in update_fn's cat((x, agg)),
 * GSN_sparse.py:114 / GSN_edge_sparse.py:112) followed by the previous stage's output.  Weights are held in registers,
 * activations in LDS.  Same per-stage parameters as gsn_linear_fwd_hip.  `stats` (double [2][n_out_last]) replaces the
 * output by column sums / sums of squares of the LAST stage's pre-BN values (train-mode BatchNorm1d, pass 1).
 * gsn_mlp_chain_supported() tells whether a chain fits the fused kernel (<= 2 stages, every K_s <= 160, n_out_s <= 128,
 * at most 6 blocks, stage-1 blocks <= 64 columns, activations identity / relu); otherwise run the stages one by one
 * with gsn_linear_fwd_hip.
 */
typedef struct {
    const gsn_block *blocks;
    int n_blocks;
    const float *W;       /* [n_out][k_total] row-major; k_total = sum(block widths) + (s > 0 ? n_out of stage s-1 : 0) */
    const float *bias;    /* [n_out] or NULL */
    int64_t n_out;
    const float *bn_mean, *bn_scale, *bn_shift; /* all three or none */
    int act;
} gsn_chain_stage;

/* Fused scatter-add: with `seg_target` (int32 [M], the target of every row, rows visited in target-sorted order through
 * row_perm = perm and seg_target = sorted_target of gsn_csr_build_hip) the last stage's rows are summed per target
 * inside the kernel and `out` is [n_seg][n_out_last] -- the torch.sparse.sum of GSN_sparse.py:140-143 without writing
 * the [E, d] messages.  `out` must first be prepared with gsn_segsum_prepare_hip (zeroes the rows of empty segments
 * and of segments that straddle a GSN_SEG_RANGE_ROWS-row boundary, which the kernel adds to atomically; all other rows
 * are plain stores, so a segment's summation order is its row order except for segments longer than that range).
 * ---------------------------------------------------------------------------------------------------------------- */
int gsn_mlp_chain_supported(int n_stages, const gsn_chain_stage *stages);
int gsn_mlp_chain_fwd_hip(int64_t m_rows, int n_stages, const gsn_chain_stage *stages, const int32_t *row_perm,
                          const int32_t *seg_target, float *out, double *stats, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * HP-2  one whole `general` layer in ONE launch (device; fp32 in / fp32 out): GSN_sparse.forward / propagate / message
 * (GSN_sparse.py:93-176), GSN_edge_sparse (GSN_edge_sparse.py:82-170) and the MPNN twins with msg_kind='general':
 *     r_e   = act_e( bn_e( cat(edge blocks) We^T + be ) )                          per edge, rows in target-sorted order
 *     S_v   = sum_{e -> v} r_e                                                     torch.sparse.sum (:140-143)
 *     out_v = act_1( bn_1( act_0( bn_0( [x_v | S_v | deg_v,0,0,0] W0^T + b0 ) ) W1^T + b1 ) )
 * where the caller has folded msg_fn's last Linear into update_fn's first (W0 = [W3x | W3a W2 | W3a b2 | 0 0 0], layers.py)
 * and deg_v = seg_ptr[v+1] - seg_ptr[v].  r_e, S_v and the hidden rows stay in LDS: every input is read once and the
 * output written once.  Matrix products: both operands split into two fp16 planes after exact power-of-two row / matrix
 * scaling, three plane products on v_mfma_f32_32x32x16_f16 with fp32 accumulation (error vs fp64 as an fp32 FMA loop's;
 * two products for rows that are exact in fp16, e.g. one-hot encodings).  Summation order of S_v = row order.
 *   seg_ptr  int32 [n_nodes+1]  target-sorted CSR of gsn_csr_build_hip
 *   edge     stage with 1..6 blocks, every block gathered through an int32 index in sorted-row order (sorted_target,
 *            sorted_source or perm of gsn_csr_build_hip), widths multiples of 4, sum <= 80; n_out <= 128
 *   x        [n_nodes][d_x], d_x multiple of 4; d_x + edge.n_out + 4 <= 160
 *   node0    W [n0][d_x + edge.n_out + 4], no blocks;  node1  W [n1][n0], no blocks;  n0, n1 <= 128, multiples of 4
 *   act 0 identity / 1 relu; bn_* as in gsn_linear_fwd_hip (eval-mode / resolved statistics)
 * and, the hidden layers of a d = 128 model (GSN_edge_sparse.py:82-170 with K = 260 / 272 edge rows, MPNN_edge_sparse.py:110-151;
 * csrc/layer_w.hip): d_x = 128, every stage 128 wide, edge blocks = x through one index, x through another, then <= 16 further
 * columns (widths multiples of 4, each gathered through one of those two indices or a third).  For this shape the forward call
 * runs a small pass over x in front of the layer kernel (one int per node: the exponent of the row's largest |value|, stream-
 * ordered scratch from hipMallocAsync unless the caller brings it: gsn_layer_fused_fwd_ws_hip) and returns GSN_E_UNSUPPORTED when it would
 * have to allocate while the stream is being captured into a graph, or when the first two edge blocks are not `x` itself; the caller
 * then composes the layer from the other entry points.
 * gsn_layer_fused_supported() says whether the shapes fit; otherwise compose gsn_mlp_chain_fwd_hip launches.
 * ---------------------------------------------------------------------------------------------------------------- */
int gsn_layer_fused_supported(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0,
                              const gsn_chain_stage *node1);
/* The weights in the form the kernel keeps them in registers (per stage a power-of-two scale from max |W * bn_scale|, fp16
 * plane fragments per wave and K step) are made ONCE per weight version by gsn_layer_fused_prepare_hip into a device buffer of
 * gsn_layer_fused_prepared_bytes() bytes (16-byte aligned) and handed to every forward call as `prepared`; re-run it when W,
 * bias-independent: bn_scale or W of a stage changed.  (Split inside the forward kernel, every workgroup re-read every weight
 * row at the same moment: 50 - 400 k cycles of prologue per launch under load.) */
int64_t gsn_layer_fused_prepared_bytes(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0,
                                       const gsn_chain_stage *node1);
int gsn_layer_fused_prepare_hip(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0,
                                const gsn_chain_stage *node1, void *prepared, void *stream);
int gsn_layer_fused_fwd_hip(int64_t n_nodes, int64_t n_edges, const int32_t *seg_ptr, const gsn_chain_stage *edge,
                            const float *x, int64_t d_x, const gsn_chain_stage *node0, const gsn_chain_stage *node1,
                            const void *prepared, float *out, void *stream);
/* The same with caller-owned scratch: gsn_layer_fused_workspace_bytes() device bytes (0 for shapes that need none; today the d = 128
 * shape: 4 bytes per node), 4-byte aligned, contents undefined before and after.  With it the call allocates nothing and can be
 * captured into a HIP graph; `workspace` may be null when the size is 0.
 * Chaining layers of a d = 128 model: `out_row_exp` (int32 [n_nodes] or null; 128-wide outputs only) receives, per output row, the
 * exponent field of its largest |value| (255: the row holds an Inf / NaN) -- the d = 128 kernel writes it with the rows, the other
 * kernels by a pass over the rows they wrote; `x_row_exp` (or null) is that array of the layer that produced `x`: the d = 128 kernel
 * then skips its own pass over x (and needs no workspace).  It must describe the CURRENT contents of x. */
int64_t gsn_layer_fused_workspace_bytes(int64_t n_nodes, const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0,
                                        const gsn_chain_stage *node1);
int gsn_layer_fused_fwd_ws_hip(int64_t n_nodes, int64_t n_edges, const int32_t *seg_ptr, const gsn_chain_stage *edge,
                               const float *x, int64_t d_x, const gsn_chain_stage *node0, const gsn_chain_stage *node1,
                               const void *prepared, float *out, void *workspace, int64_t workspace_bytes,
                               const int32_t *x_row_exp, int32_t *out_row_exp, void *stream);

/* The d = 128 shape on GRAPH-ALIGNED tiles (csrc/layer_g.hip): the same layer, the same `prepared` buffer as gsn_layer_fused_fwd_hip for that
 * shape, for a collated batch whose graph boundaries the caller knows (torch_geometric's Batch.ptr; the pointers of
 * gsn_csr_build_graphs_hip): graph g owns the consecutive vertices node_ptr[g] .. node_ptr[g+1] and no edge leaves its graph.  The node
 * part of the edge stage, x [W_i | W_j]^T, is then computed once per NODE (MPNN_edge_sparse.py:139-151 evaluates it once per EDGE:
 * cat(x_i, x_j, e) W^T), on tiles of whole graphs, and the edge stage is a gather-add of P_i[target] + P_j[source] + z_e W_z^T.
 * No workspace, no row exponents.  Every graph must have <= 128 vertices (max_nodes, the caller's bound): GSN_E_UNSUPPORTED otherwise --
 * the caller then uses gsn_layer_fused_fwd_ws_hip.  node_ptr int64 [n_graphs + 1] device, node_ptr[0] = 0, node_ptr[n_graphs] = n_nodes. */
int gsn_layer_fused_graphs_supported(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0,
                                     const gsn_chain_stage *node1);
int gsn_layer_fused_fwd_graphs_hip(int64_t n_nodes, int64_t n_edges, const int32_t *seg_ptr, const gsn_chain_stage *edge,
                                   const float *x, int64_t d_x, const gsn_chain_stage *node0, const gsn_chain_stage *node1,
                                   const void *prepared, int64_t n_graphs, const int64_t *node_ptr, int64_t max_nodes,
                                   float *out, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * HP-2  the same one-launch layer on EXACT fp16 ROW PACKS (csrc/layer_rp.hip).  Layer 0 of every reference model reads one-hot /
 * small-integer encodings (utils_graph_learning.py:78-88, :170-187: DiscreteEmbedding('one_hot_encoder')): every value is exact in
 * fp16.  A producer that knows this (gsn_count_encode_hip, gsn_one_hot_hip, or gsn_pack16_rows_hip over an existing fp32 tensor) writes
 * the rows a second time as fp16 in the layout the matrix pipe reads, and the layer kernel skips what it otherwise does per edge row:
 * gathering fp32, converting, testing the conversion for exactness.  Same arithmetic, same results as gsn_layer_fused_fwd_hip.
 *   node_rows  fp16 [n_nodes][32]   columns 0 .. d_x-1 = x, columns d_x .. 30 = 0, column 31 = 1.0   (16-byte aligned)
 *   edge_rows  fp16 [edge_rows][16] the edge-level blocks of the edge stage (identifiers, edge features) concatenated in block order,
 *                                   zero padded; NULL when the edge stage is cat(x[i], x[j]) alone
 * Every value must be exactly what the fp32 tensors hold and < 2 in magnitude (the edge stage's weight scale is made for that bound);
 * gsn_pack16_rows_hip checks both and ORs 1 into *status (device int32, caller-zeroed) when a value is not.
 * The edge stage must be: block 0 = x through an int32 index, block 1 = x through another, blocks 2.. = the edge-level blocks, all through
 * ONE int32 index (perm of gsn_csr_build_hip), <= 16 columns together; widths as gsn_layer_fused_fwd_hip's register-resident shape
 * (every stage 128 wide, d_x + 4 <= 32).  The prepared weights are NOT those of gsn_layer_fused_prepare_hip (another k-slot order):
 * gsn_layer_fused_pack16_prepare_hip into gsn_layer_fused_pack16_prepared_bytes() bytes.  The fp32 `data` pointers of the blocks are not
 * read.  gsn_layer_fused_fwd_pack16_hip returns GSN_E_UNSUPPORTED when the packs are beyond its 32-bit byte offsets (n_nodes >= 2^25).
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct {
    const uint16_t *node_rows;
    const uint16_t *edge_rows;
} gsn_pack16;
int gsn_pack16_rows_hip(const float *src, int64_t rows, int64_t width, uint16_t *dst, int64_t dst_stride, int64_t col0,
                        int64_t one_col, int32_t *status, void *stream);
/* gsn_one_hot_hip's encoding (utils_graph_learning.py:170-187: DiscreteEmbedding('one_hot_encoder') over integer columns) written straight
 * into columns col0 .. col0 + sum(n_classes) of a pack (dst_stride fp16 columns per row, a multiple of 8, <= 64, rows 16-byte aligned);
 * one_col >= 0: that column = 1.0.  With col0 == 0 and one_col >= 0 the call owns the whole pack (a node pack: its remaining columns are
 * zero by contract) and writes every column of every row.
 * A code outside its column's classes leaves that column's segment zero (clamp: counted as the nearest class), as gsn_one_hot_hip, and
 * ORs 1 into *status (device int32, caller-zeroed; may be NULL) -- the reference's F.one_hot raises there. */
int gsn_one_hot_pack16_hip(int64_t m_rows, int n_cols, const int64_t *values, const int32_t *n_classes, int clamp, uint16_t *dst,
                           int64_t dst_stride, int64_t col0, int64_t one_col, int32_t *status, void *stream);
int gsn_layer_fused_pack16_supported(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0,
                                     const gsn_chain_stage *node1);
int64_t gsn_layer_fused_pack16_prepared_bytes(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0,
                                              const gsn_chain_stage *node1);
int gsn_layer_fused_pack16_prepare_hip(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0,
                                       const gsn_chain_stage *node1, void *prepared, void *stream);
int gsn_layer_fused_fwd_pack16_hip(int64_t n_nodes, int64_t n_edges, const int32_t *seg_ptr, const gsn_chain_stage *edge,
                                   const float *x, int64_t d_x, const gsn_chain_stage *node0, const gsn_chain_stage *node1,
                                   const void *prepared, const gsn_pack16 *pack, int64_t edge_rows, float *out, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * HP-1 + HP-2 as ONE host call (r06): the orbit counting of a collated batch with its side outputs
 * (gsn_count_encode_pack16_side_hip: int64 identifiers, CSR, node pack, whole edge pack rows) and, behind it on the same stream, layer 0
 * of the model on the packs (gsn_layer_fused_fwd_pack16_hip) -- what the reference reaches through utils_ids.py:7-29
 * (subgraph_counts2ids), utils_graph_learning.py:170-187 (the one-hot encoders) and GSN_edge_sparse.py:82-170 (forward) as three
 * Python-level stages.  Two kernel launches, no allocation, no host synchronisation, capturable into a HIP graph; the two structs carry
 * the arguments of the two entry points verbatim (same meaning, same checks, same status codes), so a host keeps them filled in and
 * pays one foreign call per step.  layer->seg_ptr and the edge stage's block indices are normally count->side's CSR arrays. */
typedef struct {
    const uint32_t *plan_host, *plan_dev;
    int64_t plan_words, n_graphs;
    const int64_t *node_ptr, *edge_ptr, *edge_index;
    int64_t edge_row_stride;
    int ids_are_global;
    int64_t max_nodes, max_edges;
    int64_t *out;
    int32_t *status;
    const int32_t *n_classes;
    int clamp;
    uint16_t *pack;
    int64_t pack_stride, pack_col0;
    const gsn_count_side *side;
} gsn_count_call;
typedef struct {
    int64_t n_nodes, n_edges;
    const int32_t *seg_ptr;
    const gsn_chain_stage *edge;
    const float *x;
    int64_t d_x;
    const gsn_chain_stage *node0, *node1;
    const void *prepared;
    const gsn_pack16 *pack;
    int64_t edge_rows;
    float *out;
} gsn_layer_pack16_call;
int gsn_count_layer_step_hip(const gsn_count_call *count, const gsn_layer_pack16_call *layer, void *event_between, void *stream);
/* event_between: a hipEvent_t recorded on `stream` between the two launches, or NULL (a measuring host brackets the two kernels of the one
 * call with it: bench.py's per-kernel HIP-event times). */

/* ------------------------------------------------------------------------------------------------------------------
 * HP-2  dense stage on DIRECT rows with fp16x3 matrix arithmetic (device): the same operation as gsn_linear_fwd_hip
 *     out[m, :] = act( bn( cat(blocks[0][m], blocks[1][m], ..) W^T + b ) )          (models_misc.py:52-58)
 * for blocks without a row index (node-level stages: the node product of a wide layer, its K = 260 node stage, jk projections,
 * the d = 300 ogb stages).  Two fp16 planes per operand after an exact power-of-two scaling (every input row by its largest
 * magnitude -- a pre-pass inside the call --, every output column of W by its largest entry), three plane products per fp32
 * product on the 16-bit matrix pipe, fp32 accumulation: the error of an fp32 FMA loop at half the matrix work of the bf16x6
 * kernel, over the whole fp32 exponent range; a row with an Inf / NaN comes out NaN.
 *   gsn_linear_f16x3_kpad(K)          K rounded up to whole K slices of 32 columns (two at least)
 *   gsn_linear_f16x3_prepare_hip      splits W [n_out][K] once: planes = 2 * n_out * kpad(K) fp16 values (laid out
 *                                     [n_out][kpad / 32][high | low][32]: one cache line per row and slice), col_inv = n_out floats
 *                                     (device buffers of the caller; valid until W changes)
 *   gsn_linear_f16x3_scratch_bytes    size of row_scratch for m_rows rows of K columns: the rows' inverse scales and their two
 *                                     fp16 planes (split once per call by a pre-pass, read by every column tile)
 *   gsn_linear_f16x3_fwd_hip          blocks: data + width only (idx / idx32 must be NULL), widths multiples of 4, 16-byte
 *                                     aligned; row_scratch: gsn_linear_f16x3_scratch_bytes(m_rows, K) bytes, 16-byte aligned;
 *                                     bias / bn_* / act as gsn_linear_fwd_hip
 *   gsn_linear_f16x3_fwd_stats_hip    the train-mode BatchNorm stage (models_misc.py:52-58, bn in train mode): out = the pre-BN rows
 *                                     x W^T + b, and their fp64 column sums / sums of squares ADDED to stats[2][n_out] by the same
 *                                     launch (gsn_linear_fwd_hip's `stats` contract); n_out a multiple of 4, out 16-byte aligned
 * ---------------------------------------------------------------------------------------------------------------- */
int64_t gsn_linear_f16x3_kpad(int64_t k_total);
int64_t gsn_linear_f16x3_mpad(int64_t m_rows);   /* rows of a row scratch: m_rows + at least 128 rows of zero planes, in whole 256-row tiles */
int64_t gsn_linear_f16x3_scratch_bytes(int64_t m_rows, int64_t k_total);
int gsn_linear_f16x3_prepare_hip(const float *W, int64_t n_out, int64_t k_total, void *planes, float *col_inv, void *stream);
/* the same from a weight given through element strides (W[j][k] at W + j * w_row_stride + k * w_col_stride; a transposed view of a row-major
 * matrix: 1, leading dimension) */
int gsn_linear_f16x3_prepare_strided_hip(const float *W, int64_t n_out, int64_t k_total, int64_t w_row_stride, int64_t w_col_stride,
                                         void *planes, float *col_inv, void *stream);
int gsn_linear_f16x3_fwd_hip(int64_t m_rows, int n_blocks, const gsn_block *blocks, const void *planes, const float *col_inv,
                             const float *bias, int64_t n_out, const float *bn_mean, const float *bn_scale, const float *bn_shift,
                             int act, float *row_scratch, float *out, void *stream);
int gsn_linear_f16x3_fwd_stats_hip(int64_t m_rows, int n_blocks, const gsn_block *blocks, const void *planes, const float *col_inv,
                                   const float *bias, int64_t n_out, float *row_scratch, float *out, double *stats, void *stream);
/* The rows' pre-pass by itself (row_scratch left exactly as gsn_linear_f16x3_fwd_hip leaves it: [mpad(m_rows)] inverse row scales,
 * then [mpad(m_rows)] rows of kpad(K) / 32 lines of 32 high | 32 low halfs; rows past m_rows hold zeros), and the product over rows split EARLIER (row_scratch is read, not written; blocks
 * give the widths only).  A training step splits every row set once: the forward product's scratch of X and the input-gradient product's
 * scratch of gH are what gsn_wgrad_f16x3_hip multiplies. */
int gsn_linear_f16x3_split_rows_hip(int64_t m_rows, int n_blocks, const gsn_block *blocks, float *row_scratch, void *stream);
int gsn_linear_f16x3_fwd_presplit_hip(int64_t m_rows, int n_blocks, const gsn_block *blocks, const void *planes, const float *col_inv,
                                      const float *bias, int64_t n_out, const float *bn_mean, const float *bn_scale, const float *bn_shift,
                                      int act, float *row_scratch, float *out, void *stream);
int gsn_linear_f16x3_fwd_stats_presplit_hip(int64_t m_rows, int n_blocks, const gsn_block *blocks, const void *planes, const float *col_inv,
                                            const float *bias, int64_t n_out, float *row_scratch, float *out, double *stats, void *stream);
/* gsn_bn_act_hip (out = act((h - mean) * scale + shift), models_misc.py:52-59's BatchNorm + activation on finished statistics) writing the row
 * scratch of its output -- for a stage output that only feeds the next product on the fp16x3 kernel (gsn_linear_f16x3_fwd_[stats_]presplit_hip) and
 * that product's plane weight gradient: no row pre-pass; `out` (fp32 rows) may be NULL.  n_cols a multiple of 4, at most 640. */
int gsn_bn_act_planes_hip(int64_t m_rows, int64_t n_cols, const float *h, const float *mean, const float *scale, const float *shift, int act,
                          float *out, float *row_scratch, void *stream);
/* HP-2 adjoint: weight gradient of a dense stage from fp16 planes (r06; torch.nn.Linear's weight.grad under models_misc.py:52-58):
 *     grad_w[n_out][K] += gH^T X
 * g_scratch = the row scratch of gH [m_rows][n_out] (its "K" is n_out), x_scratch = the row scratch of X [m_rows][K], both as
 * gsn_linear_f16x3_fwd_hip / _split_rows_hip leave them.  Three fp16 plane products per fp32 product; the per-row scales of the two
 * operands are reconciled per slab of rows inside the kernel (csrc/wgrad_f16.hip).  Same contract as gsn_wgrad_hip otherwise: grad_w is
 * ADDED to with float atomics (order varies in the last bits from run to run); a row with an Inf / NaN in gH or X makes the tile NaN. */
int gsn_wgrad_f16x3_hip(int64_t m_rows, int64_t n_out, int64_t k_total, const void *g_scratch, const void *x_scratch, float *grad_w,
                        void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * HP-2  edge stage of a `general` layer with the node part of its Linear taken out of the edge loop (device, fp32).
 * msg_fn's first Linear acts on cat(x_i, x_j, z_e) (GSN_sparse.py:166-171, GSN_edge_sparse.py:160-165, MPNN twins);
 * cat(x_i, x_j, z_e) W^T = x_i W_i^T + x_j W_j^T + z_e W_z^T, so the caller computes P = x [W_i | W_j]^T once per NODE
 * (gsn_linear_fwd_hip, N rows instead of E; bias and eval-mode BatchNorm folded into P_i and the weights) and this entry does
 *     out[t] = sum_{e : tgt(e) = t} act( P_i[t] + P_j[src(e)] + z_e W_z^T )            (GSN_edge_sparse.py:136-139 scatter-add)
 *   seg_ptr, sorted_src, perm   the target-sorted CSR of gsn_csr_build_hip / gsn_csr_build_graphs_hip
 *   p_i, p_j   fp32 device rows of `pitch` floats (two column ranges of one [N][2 d] matrix, or two matrices), d columns used
 *   z0, z1     per-edge blocks [E][w0], [E][w1] (identifiers / edge features; widths multiples of 4, w0 + w1 <= 16) or NULL
 *   wz_t       [w0 + w1][d] = W_z^T;   act 0 identity / 1 relu;   out [N][d] fully overwritten;  d multiple of 4, <= 256
 * ---------------------------------------------------------------------------------------------------------------- */
int gsn_edge_split_sum_hip(int64_t n_nodes, int64_t n_edges, const int32_t *seg_ptr, const int32_t *sorted_src,
                           const int32_t *perm, const float *p_i, const float *p_j, int64_t pitch, const float *z0,
                           int64_t w0, const float *z1, int64_t w1, const float *wz_t, int64_t d, int act, float *out,
                           void *stream);

/* The same CSR for a PyG-collated batch, one launch.  A batch is a disjoint union (torch_geometric's Batch / the reference's
 * DataLoader, main.py:19, utils_data_prep.py:35-60): graph g owns the consecutive columns edge_ptr[g] .. edge_ptr[g+1] of
 * edge_index and the consecutive vertices node_ptr[g] .. node_ptr[g+1], so every graph is sorted on its own in LDS (one wave
 * per graph) -- no global atomics, no global scan, same stable result as gsn_csr_build_hip.
 *   node_ptr, edge_ptr  int64 device [n_graphs + 1] (the arrays gsn_count_hip takes)
 *   max_nodes, max_edges  upper bounds over the graphs (size LDS: (2 (max_nodes + 1) + 2 max_edges) * 4 B <= 64 KiB, else
 *                       GSN_E_UNSUPPORTED -> use gsn_csr_build_hip)
 *   index, other, seg_ptr, perm, sorted_target, sorted_other   as gsn_csr_build_hip
 *   status              int32 device [1], zeroed by the caller: raised to GSN_ST_BAD_INDEX when a column's `index` entry lies
 *                       outside its graph's vertex range (the arrays do not describe a collated batch; outputs are then
 *                       unspecified), GSN_ST_TOO_LARGE when a graph exceeds max_nodes / max_edges */
int gsn_csr_build_graphs_hip(int64_t n_graphs, const int64_t *node_ptr, const int64_t *edge_ptr, int64_t n_nodes,
                             int64_t n_edges, int64_t max_nodes, int64_t max_edges, const int64_t *index,
                             const int64_t *other, int32_t *seg_ptr, int32_t *perm, int32_t *sorted_target,
                             int32_t *sorted_other, int32_t *status, void *stream);

int gsn_segsum_prepare_hip(int64_t n_seg, int64_t n_rows, const int32_t *seg_ptr, const int32_t *row_target,
                           int64_t n_out, float *out, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Identifier encoding (device).  The multi-hot encoding the reference applies to the integer identifiers in front of
 * every GSN layer: utils_graph_learning.one_hot_encoder.forward (:170-187) via DiscreteEmbedding('one_hot_encoder')
 * (models_graph_classification.py:222).  values int64 [M][C] -> out fp32 [M][sum n_classes], column c becoming
 * n_classes[c] floats with a single 1 at index values[r][c].  n_classes is a HOST array.  clamp != 0: indices are
 * clamped into [0, n_classes[c]-1] (the reference would raise on an out-of-range index).
 * ---------------------------------------------------------------------------------------------------------------- */
int gsn_one_hot_hip(int64_t m_rows, int n_cols, const int64_t *values, const int32_t *n_classes, int clamp, float *out,
                    void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Dataset-level dense recoding of integer columns (device): utils_encoding.one_hot_unique (utils_encoding.py:37-59),
 * i.e. per column np.unique(values[:, c], return_inverse=True): codes[r][c] = rank of values[r][c] among the distinct
 * values of column c, n_distinct[c] = their number (the reference's `d`).
 *   gsn_column_range_hip : col_min / col_max (device int64 [C]) of values int64 [M][C]; also what one_hot_max needs
 *                          (utils_encoding.py:62-69: d[c] = max + 1).
 *   gsn_column_ranks_hip : col_min as produced above; col_base (device int64 [C+1]) = prefix sums of the per-column table
 *                          sizes (max - min + 1), which the caller computes after reading the ranges back;
 *                          table = device scratch of table_elems = col_base[C] int32.
 * All asynchronous on `stream`.
 * ---------------------------------------------------------------------------------------------------------------- */
int gsn_column_range_hip(int64_t m_rows, int n_cols, const int64_t *values, int64_t *col_min, int64_t *col_max,
                         void *stream);
int gsn_column_ranks_hip(int64_t m_rows, int n_cols, const int64_t *values, const int64_t *col_min,
                         const int64_t *col_base, int64_t table_elems, int32_t *table, int64_t *codes,
                         int64_t *n_distinct, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Embedding of categorical columns (device): utils_graph_learning.multi_embedding.forward (:151-163), the
 * DiscreteEmbedding('embedding') of identifiers / degrees / node and edge types, and the same shape as ogb's
 * AtomEncoder / BondEncoder ('atom_encoder' / 'bond_encoder', :99-107).  codes int64 [M][C]; table c is fp32
 * [rows_c][d] row-major.  meta (device int64 [2C]) = table base addresses then rows_c.  concat != 0: out [M][C*d] is the
 * concatenation, else out [M][d] the sum over columns.  status (device int32, caller-zeroed) is raised to
 * GSN_ST_BAD_INDEX when a code is outside its table (that row's values are NaN).  bwd accumulates grad_out into the gradient tables named by
 * grad_meta (caller zero-fills them).  table_rows = the row counts again as a HOST array [C] (NULL allowed in fwd): they
 * pick the kernel -- tables that together have <= 448 rows are held in LDS per workgroup (64-wide slices), larger ones
 * are read / accumulated in HBM; the backward of summed embeddings over >= 4096 rows is the product OneHot(codes)^T grad_out on the
 * matrix pipe (exact bf16 planes, fp32 accumulation).
 * ---------------------------------------------------------------------------------------------------------------- */
int gsn_embed_fwd_hip(int64_t m_rows, int n_cols, int d, int concat, const int64_t *codes, const int64_t *meta,
                      const int64_t *table_rows, float *out, int32_t *status, void *stream);
int gsn_embed_bwd_hip(int64_t m_rows, int n_cols, int d, int concat, const int64_t *codes, const int64_t *grad_meta,
                      const int64_t *table_rows, const float *grad_out, void *stream);
/* The same backward with the gradient tables inside ONE caller-zeroed device allocation: table c starts at grad_flat +
 * table_offsets[c] floats (table_offsets, table_rows: HOST arrays [C]).  The table addresses travel as launch arguments: no device pointer
 * array, so no host-to-device copy per call and no memcpy node per embedding in a captured training step (utils_graph_learning.py:134-167
 * under train_test_funcs.py:88-106 at the reference's batch sizes).  gsn_embed_bwd_flat_supported: 1 when the shape is handled (<= 16
 * code columns; tables that fit the LDS slices or the matrix-pipe product), else 0 and gsn_embed_bwd_flat_hip returns
 * GSN_E_UNSUPPORTED -- use gsn_embed_bwd_hip then. */
int gsn_embed_bwd_flat_supported(int64_t m_rows, int n_cols, int concat, const int64_t *table_rows);
int gsn_embed_bwd_flat_hip(int64_t m_rows, int n_cols, int d, int concat, const int64_t *codes, float *grad_flat,
                           const int64_t *table_offsets, const int64_t *table_rows, const float *grad_out, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * First Linear of msg_fn over one-hot encoded inputs as a weight-row gather with fused bias / BatchNorm / activation
 * and fused scatter-add (device).  Replaces, for inputs that are DiscreteEmbedding('one_hot_encoder') outputs
 * (utils_graph_learning.py:78-81, models_graph_classification.py:205-222), the reference's sequence
 *     one_hot -> cat(x_i, x_j, ids, e) -> msg_fn.fc[0] -> bn -> act     (GSN_edge_sparse.py:152-170, models_misc.py:52-59)
 *     -> torch.sparse.sum scatter-add                                   (GSN_edge_sparse.py:136-139)
 * Row q of the target-sorted edge order takes, per slot s, the code  codes[idx[q] * stride + col]  (idx NULL: row q) and
 * adds row  w_off + code  of WT = W^T ([k_total][n_out] fp32 row-major):
 *     out[seg_target[q]] += act(bn(bias + sum_s WT[w_off_s + code_s]))
 * `out` [n_targets][n_out] must have been prepared with gsn_segsum_prepare_hip; seg_target int32 [m_rows] ascending.
 * stats != NULL: statistics pass instead -- fp64 [2][n_out] column sums of (bias + sum) and of its square (caller
 * zero-fills), no `out`.  status (device int32, caller-zeroed) is raised to GSN_ST_BAD_INDEX if a code is outside
 * [0, n_classes) (the code is then treated as 0).  act: 0 identity, 1 relu, 2 elu, 3 tanh.
 * ---------------------------------------------------------------------------------------------------------------- */
#define GSN_MAX_CODE_SLOTS 16
typedef struct gsn_code_slot {
    const int64_t *codes;  /* device, [rows][stride] */
    const int32_t *idx;    /* device int32 [m_rows] row of `codes` feeding sorted position q, or NULL */
    int32_t stride, col;   /* row pitch of `codes` in elements, column used */
    int32_t w_off;         /* first row of this slot's block in WT */
    int32_t n_classes;     /* number of rows of that block */
    int32_t clamp;         /* != 0: codes are clamped into [0, n_classes-1] instead of being reported (cf. gsn_one_hot_hip) */
    int32_t reserved;
} gsn_code_slot;
int gsn_code_stage_supported(int n_slots, int64_t k_total, int64_t n_out);
int gsn_code_stage_fwd_hip(int64_t m_rows, int n_slots, const gsn_code_slot *slots, const float *WT, int64_t k_total,
                           const float *bias, int64_t n_out, const float *bn_mean, const float *bn_scale,
                           const float *bn_shift, int act, const int32_t *seg_target, float *out, double *stats,
                           int32_t *status, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * BatchNorm + activation applied to materialised pre-BN rows (device): y = act((h - mean) * scale + shift) per column,
 * models_misc.mlp.forward (:52-59) for a train-mode stage whose statistics were taken in the same pass that wrote h
 * (gsn_linear_fwd_hip with both `out` and `stats`).  h, out fp32 [M][C] row-major (may alias); vectors may be NULL.
 * ---------------------------------------------------------------------------------------------------------------- */
int gsn_bn_act_hip(int64_t m_rows, int64_t n_cols, const float *h, const float *mean, const float *scale,
                   const float *shift, int act, float *out, void *stream);
/* The statistics of materialised pre-BN rows (device): ADDS per-column sum and sum of squares of h [M][C] into stats double [2][C]
 * (caller zeroes it) -- what gsn_linear_fwd_hip's `stats` argument takes inside its product, for rows that another kernel wrote
 * (the fp16x3 dense kernel of a train-mode stage, models_misc.py:52-59 with nn.BatchNorm1d in training mode). */
int gsn_column_stats_hip(int64_t m_rows, int64_t n_cols, const float *h, double *stats, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Backward of a dense mlp stage  H = X W^T + b, Z = bn(H), Y = act(Z)  (models_misc.py:52-59) for inputs that are plain
 * row-major blocks (device).  What PyTorch autograd does for the reference's nn.Linear / BatchNorm1d / activation.
 *   gsn_bn_act_bwd_hip : grad_h = d/dH from grad_y.  y = the stage output (activation derivative is taken from it);
 *       train_bn != 0: batch-statistics BatchNorm -- h = pre-BN rows, mean / invstd = the batch statistics, coef =
 *       gamma * invstd, sums = fp64 [2][C] scratch (zero-filled) that receives sum(gZ) = grad beta and sum(gZ * xhat) =
 *       grad gamma;  train_bn == 0: grad_h = gZ * coef (coef NULL = 1; h, mean, invstd, sums unused);  train_bn == 2: BatchNorm
 *       on its running statistics (module in eval mode, models_misc.py:41-45 under model.eval()) with gradients for gamma / beta:
 *       h = pre-BN rows, mean / invstd = the running statistics, sums as above, grad_h = gZ * coef.
 *       grad_bias (fp64 [C], zero-filled, may be NULL) receives the column sums of grad_h.  grad_h may alias grad_y.
 *   gsn_wgrad_hip : grad_w[n_out][K] += grad_h^T X  with X the concatenation of `blocks` (direct, or gathered through idx / idx32 as
 *       in gsn_linear_fwd_hip: the x_i / x_j blocks of an edge stage are read where they lie); caller zero-fills.
 *   The input gradient is gsn_linear_fwd_hip(grad_h, weight = W^T).
 * ---------------------------------------------------------------------------------------------------------------- */
int gsn_bn_act_bwd_hip(int64_t m_rows, int64_t n_cols, const float *grad_y, const float *y, const float *h,
                       const float *mean, const float *invstd, const float *coef, int train_bn, int act, double *sums,
                       float *grad_h, double *grad_bias, void *stream);
/* The same for a BatchNorm stage (train_bn 1 or 2) WITHOUT reading the stage output: the activation's derivative is taken from
 * z = (h - mean) * coef + shift, recomputed from the pre-BN rows with the forward pass's expression (gsn_bn_act_hip) -- one read of
 * [M][C] less in each of the two passes (they are HBM-bound: 3 -> 2 and 4 -> 3 array passes). */
int gsn_bn_act_bwd_from_h_hip(int64_t m_rows, int64_t n_cols, const float *grad_y, const float *h, const float *mean,
                              const float *invstd, const float *coef, const float *shift, int train_bn, int act, double *sums,
                              float *grad_h, double *grad_bias, void *stream);
/* The same again for a stage whose grad_h is consumed as fp16 planes only (r06): no fp32 grad_h is written; row_scratch
 * (gsn_linear_f16x3_scratch_bytes(m_rows, n_cols) bytes, 16-byte aligned) receives grad_h's inverse row scales and planes, ready for
 * gsn_linear_f16x3_fwd_presplit_hip (the input gradient grad_h W) and gsn_wgrad_f16x3_hip.  n_cols a multiple of 4, at most 640.  grad_bias
 * receives the column sums analytically: zero under batch statistics (the statistics absorb a shift of H), coef * sum(gZ) on running ones. */
int gsn_bn_act_bwd_planes_hip(int64_t m_rows, int64_t n_cols, const float *grad_y, const float *h, const float *mean, const float *invstd,
                              const float *coef, const float *shift, int train_bn, int act, double *sums, float *row_scratch,
                              double *grad_bias, void *stream);
int gsn_wgrad_hip(int64_t m_rows, int64_t n_out, const float *grad_h, int n_blocks, const gsn_block *blocks, float *grad_w,
                  void *stream);
/* The folded first weight of update_fn for a `general` layer whose aggregation runs in front of msg_fn's last Linear (W2, b2) --
 * GSN_edge_sparse.py:153-170 / GSN_sparse.py:166-171 with  update_fn.fc[0](cat(x, sum_e(W2 r_e + b2))) = cat(x, S, deg) out^T:
 *   out [rows][d_x + h_cols + 1 + pad_cols] = [ w3[:, :d_x] | w3[:, d_x:] W2 | w3[:, d_x:] b2 | 0 ],   w3 [rows][d_x + a_cols] (row stride ld3),
 *   w2 [a_cols][h_cols] (row stride ld2), b2 [a_cols]; pad_cols zero columns (the degree block is passed four floats wide so that the
 *   stage's rows are staged as float4).  One launch, fp32 FMA dot products.
 * _bwd: g [rows][>= d_x + h_cols + 1] (row stride ldg; columns past d_x + h_cols are not read) -> g_w3 [rows][d_x + a_cols], g_w2 [a_cols][h_cols], g_b2 [a_cols] (all written,
 *   contiguous), one launch.  A training step rebuilds the fold at every step (the three matrices move). */
int gsn_fold_weights_fwd_hip(int64_t rows, int64_t d_x, int64_t a_cols, int64_t h_cols, int64_t pad_cols, const float *w3, int64_t ld3,
                             const float *w2, int64_t ld2, const float *b2, float *out, void *stream);
int gsn_fold_weights_bwd_hip(int64_t rows, int64_t d_x, int64_t a_cols, int64_t h_cols, const float *g, int64_t ldg, const float *w3,
                             int64_t ld3, const float *w2, int64_t ld2, const float *b2, float *g_w3, float *g_w2, float *g_b2,
                             void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * out[r] = concat_b blocks[b].data[idx_b[r]]  (device): torch.cat((x_i, x_j, identifiers, edge_features), -1) of
 * GSN_sparse.py:166-171 / GSN_edge_sparse.py:160-165 materialised as fp32 [M][sum widths].  Only the training path uses
 * it (the assembled rows feed the weight gradient); inference gathers on the fly inside the dense kernels.
 * ---------------------------------------------------------------------------------------------------------------- */
int gsn_gather_cat_hip(int64_t m_rows, int n_blocks, const gsn_block *blocks, float *out, void *stream);

/* out[r] = x[r] + table[idx[r]]  (device; fp32 [M][d], table [T][d], idx int64 [M]): the virtual-node embedding added to the vertices of
 * its graph, models_graph_classification_ogb_original.py:236 `x + vn_embedding[data.batch]`, in one pass.  A row whose index lies
 * outside [0, T) comes out NaN.  Adjoint: grad x = grad out; grad table = the sum readout of grad out (gsn_propagate_fwd_hip). */
int gsn_add_gathered_hip(int64_t n_rows, int64_t d, const float *x, const float *table, const int64_t *idx, int64_t n_table, float *out,
                         void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Train-mode BatchNorm1d bookkeeping (device), what nn.BatchNorm1d does around the normalisation (models_misc.py:41-45):
 * stats fp64 [2][C] = column sum and sum of squares of the M pre-BN rows ->  mean, invstd = 1/sqrt(biased var + eps),
 * scale = gamma * invstd, shift = beta (gamma / beta NULL: 1 / 0), and running_mean / running_var (both or neither)
 * updated with `momentum` and the unbiased variance.  All vectors fp32 [C], one launch.
 * ---------------------------------------------------------------------------------------------------------------- */
int gsn_bn_finalize_hip(int64_t n_cols, int64_t m_rows, double eps, double momentum, const double *stats, const float *gamma,
                        const float *beta, float *running_mean, float *running_var, float *mean, float *invstd, float *scale,
                        float *shift, void *stream);
/* gsn_bn_finalize_count_hip and gsn_bn_act_hip in one launch (same expressions, same values): the batch statistics' vectors, the running
 * statistics and the counter are written, then out = act((h - mean) * scale + shift) over the M rows.  For row counts where a launch costs
 * more than the pass (the reference's batch sizes: 32 .. 128 graphs). */
int gsn_bn_finalize_act_hip(int64_t n_cols, int64_t m_rows, double eps, double momentum, const double *stats, const float *gamma,
                            const float *beta, float *running_mean, float *running_var, float *mean, float *invstd, float *scale,
                            float *shift, int64_t *num_batches_tracked, const float *h, int act, float *out, void *stream);

/* The same with nn.BatchNorm1d's `num_batches_tracked` (int64 [1], device; NULL = none) incremented by the kernel. */
int gsn_bn_finalize_count_hip(int64_t n_cols, int64_t m_rows, double eps, double momentum, const double *stats, const float *gamma,
                              const float *beta, float *running_mean, float *running_var, float *mean, float *invstd, float *scale,
                              float *shift, int64_t *num_batches_tracked, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GSN_ABI_H */
