// HP-1 counting kernel for gfx950 (MI355X) + its C-ABI launcher gsn_count_hip.
//
// One workgroup = one graph.  The graph lives in LDS for the whole kernel:
//   * adjacency bit matrix A[n][W] (W = ceil(n/64) 64-bit words per row), built from the edge_index columns with
//     LDS atomic-or (self loops dropped, parallel edges merged -> the simple undirected graph graph-tool matches on);
//     directed plans (main.py --directed, vertex mode): A holds the out-neighbour rows and a second matrix the in-neighbour rows,
//   * edge mode: the column endpoints (u8), a CSR rank  slot(u,v) = rowstart[u] + popcount(A[u] & below(v))  and
//     last[slot] = highest column holding (u,v)  ("last duplicate wins", utils_graph_processing.py:142-144),
//   * the packed plan table, a lane-interleaved candidate stack, and (if it fits) a staging copy of the output rows.
// Work is a pool of tasks (output column, output row); lanes pull tasks with a wave-aggregated LDS atomic
// (ballot + mbcnt prefix) so that lanes whose search finished early are refilled immediately -- rooted searches have
// wildly different lengths.  Every (row, column) cell is produced by exactly one lane: no atomics on the counts, no
// global atomics, deterministic.  Edge mode with undirected orbit classes: rows (u,v) and (v,u) are equal, the u < v row
// searches and writes both.  HBM traffic = read edge_index once (16 B/column) + write the int64 rows once.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "count_core.h"

namespace gsn {

constexpr int GSN_ENC_MAX_COLS = 64;

// What gsn_count_encode_pack16_side_hip adds to a counting launch (r06): the target-sorted CSR of the batch's columns, the node pack from
// integer node codes, the edge codes' one-hot columns of the edge pack.  They are made by SIDE WORKGROUPS of the same launch (side_block
// below), every (SIDE_EVERY + 1)-th workgroup of the grid; the counting workgroups run count_body exactly as without them.
struct SideArgs {
    int32_t *csr_seg, *csr_perm, *csr_tgt, *csr_oth;   // csr_seg == null: no CSR
    int64_t tot_nodes, tot_edges;
    const int64_t *ncode;      // node codes [tot_nodes][ncode_cols] or null
    uint16_t *npack;           // fp16 [tot_nodes][32]
    const int64_t *ecode;      // edge codes [tot_edges][ecode_cols] or null
    uint16_t *epack;           // fp16 [tot_edges][16]: columns ecode_col0 .. 15 are written
    int32_t *code_status;      // or null
    int csr_row;               // the row of edge_index that is the aggregation target
    int ncode_cols, ncode_clamp, ecode_cols, ecode_clamp, ecode_col0;
    unsigned ncode_ptr_w[2], ecode_ptr_w[2];    // first pack column of every code column's classes, one BYTE per column + the end (prefix sums; the
                                                // edge codes' start at ecode_col0) -- as words: byte-sized kernel arguments are vector loads on gfx9
};
#ifndef GSN_SIDE_EVERY
#define GSN_SIDE_EVERY 4
#endif
constexpr int SIDE_EVERY = GSN_SIDE_EVERY;  // counting workgroups per side workgroup (A/B: 2: +0.0xx, 8: see profiles/r06_count_side_ab.txt)

struct CountArgs {
    const uint32_t *plan;      // device
    int plan_words;
    int mode, n_cols, n_plans, plans_off, kmax;
    const int64_t *node_ptr, *edge_ptr, *src, *dst;
    int ids_are_global;
    const int32_t *graph_ids;  // or null
    int n_cap, e_cap;          // LDS capacities (rows of A, columns)
    int n_decl, e_decl;        // the caller's max_nodes / max_edges: a single graph beyond them is reported (GSN_ST_TOO_LARGE)
    int stack_skip;            // frames of the candidate stack that are never addressed (the smallest n_fixed of the plans)
    int pair;                  // 1: a workgroup takes the graphs 2 i, 2 i + 1 -- as ONE graph, their disjoint union, when that fits n_cap / e_cap
    int n_graphs;
    int stage_out;             // 1: output rows staged in LDS then written coalesced
    int sym;                   // edge mode with undirected orbit classes: rows (u,v) and (v,u) are equal -> search once
    int split;                 // workgroups per graph (each takes a contiguous slice of the (column,row) task space)
    int64_t *out;
    int32_t *status;
    // LDS byte offsets
    int off_valid, off_stack, off_plan, off_eu, off_ev, off_rowstart, off_last, off_out, off_misc;
    int off_prim, off_revof;   // edge mode: the rows that run searches, in column order; per column the last column of the reverse pair
    int off_ball;              // distance-pruning tables (radius 2, radius 3: n_cap rows each) or -1
    int off_degp;              // degree bit planes of the cores, [CORE_MAX + 1][DEG_PLANES][W] words (plans with a chain tail), or -1
    int degp_mask;             // bit d: some chain-tail plan lives in the d-core
    int any_tail;              // some plan ends in a closed form (plan_tail != 0)
    int tail_loop;             // graphs of <= 64 vertices: the last two levels in the tight loop too (dense graphs; GSN_COUNT_TAIL_LOOP)
    int off_core;              // d-cores of the graph, d = 0 .. CORE_MAX (W words each)
    int core_mask;             // bit d: some plan needs the d-core
    int off_ain;               // directed plans: the in-neighbour bit matrix
    int stride;                // words per plan (plan_stride)
    int pull_batch;            // idle lanes that wait before the pool's pull arm runs (1: every trip; GSN_PULL_BATCH)
    int zero_status;           // 1: a workgroup owns its graphs' status words (no split, no graph list) and zeroes them itself -- no memset launch in front
    // fused identifier encoding (gsn_count_encode_hip): column c of a finished cell also / instead leaves as n_classes[c] floats
    // with a single 1 (utils_graph_learning.one_hot_encoder, :170-187) -- the int64 round trip through HBM and the one-hot launch go
    unsigned short enc_n[GSN_ENC_MAX_COLS];   // n_classes per output column (blocks in column order)
    float *enc_out;            // [rows_total][enc_width] or null
    int enc_width, enc_clamp, off_enc;
    int enc_stage;             // 1: cells leave their class index in an LDS byte array, the rows are expanded and written coalesced at the end
    int off_encst;             // that array: [rows_cap][n_cols] bytes, 0xff = no class (count out of range, unclamped)
    int enc_from_counts;       // 1: no byte array -- the staged 16-bit counts (stage_out) give the class indices at the end (LDS per workgroup: occupancy)
    uint16_t *enc16;           // the same rows as fp16 into a column range of an exact row pack (gsn_count_encode_pack16_hip), or null; staged rows only
    int enc_no32;              // 1: the fp16 pack columns are the ONLY form of the encoded rows (enc_out is a placeholder that is never written)
    int enc16_stride, enc16_col0;
    int side_mask;             // bit 0: CSR, bit 1: node pack, bit 2: edge codes; != 0: the grid holds side workgroups
    int n_items;               // counting work items of the launch (graphs, or pairs of graphs)
    int lds_bytes;             // dynamic LDS of the launch (a side workgroup sorts as many graphs at a time as fit there)
    SideArgs side;
};

// The side arguments are read from the kernel-argument segment where they are used (scalar loads through a laundered pointer), not held in
// scalar registers from the kernel's entry: the molecule instantiation runs at its register bound (amdgpu_waves_per_eu).
typedef const SideArgs __attribute__((address_space(4))) *side_ptr_t;
__device__ __forceinline__ side_ptr_t side_late() {
    typedef const unsigned char __attribute__((address_space(4))) *kptr_t;
    kptr_t k = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return (side_ptr_t)(k + offsetof(CountArgs, side));
}
__device__ __forceinline__ int side_byte(const unsigned w0, const unsigned w1, const int c) { return (int)(((c < 4 ? w0 : w1) >> (8 * (c & 3))) & 0xffu); }

// One side workgroup: the graphs g_lo .. g_hi - 1 (those of the SIDE_EVERY counting workgroups dispatched right in front of it), taken in
// chunks of consecutive graphs that fit the launch's LDS together.  A chunk is a disjoint union -- consecutive vertex ids, consecutive
// columns, no column between two graphs -- so ONE stable counting sort by target sorts all its graphs (what csr_graphs_kernel does per
// graph in a launch of its own; GSN_sparse.py:140-143).  Per chunk: (A) every global load -- targets, sources, vertex codes, edge codes --
// in batches of four per lane, results into LDS; (B) histogram, scan, placement, order restore in LDS; (C) every global store: seg_ptr,
// perm / sorted targets / sorted sources, the node pack rows (four lanes per 64-byte row; what gsn_one_hot_pack16_hip writes,
// utils_graph_learning.py:170-187), the edge codes' columns of the edge pack.  Loads and stores are kept apart because gfx9 counts them in one
// counter: a load issued behind a store waits for the store's acknowledgement too.  Statuses stay with the counting workgroups (they test every
// column's endpoints against their graph); a code outside its classes ORs 1 into *code_status.
template <int T>
__device__ __forceinline__ void side_block(const CountArgs &a, unsigned char *smem, const int g_lo, const int g_hi) {
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    const side_ptr_t sd = side_late();
    const int tid = threadIdx.x;
    const bool do_csr = (a.side_mask & 1) != 0, do_np = (a.side_mask & 2) != 0, do_ec = (a.side_mask & 4) != 0;
    const int64_t *trow = sd->csr_row ? a.dst : a.src, *orow = sd->csr_row ? a.src : a.dst;
    auto need = [&](int64_t n, int64_t E) { return (n + 1) * 8 + n * 4 + E * 7 + 16; };
    if (do_csr) {
        // the ends of seg_ptr: vertices in front of the first graph (none in a collated batch) and the closing entries behind the last
        if (g_lo == 0) for (int64_t v = tid; v < a.node_ptr[0]; v += T) sd->csr_seg[v] = 0;
        if (g_hi == a.n_graphs) for (int64_t v = a.node_ptr[g_hi] + tid; v <= sd->tot_nodes; v += T) sd->csr_seg[v] = (int32_t)sd->tot_edges;
    }
    int g = g_lo;
    while (g < g_hi) {
        const int64_t n0 = a.node_ptr[g], e0 = a.edge_ptr[g];
        int g2 = g + 1;
        int64_t n64 = a.node_ptr[g2] - n0, E64 = a.edge_ptr[g2] - e0;
        const bool fits = n64 >= 0 && E64 >= 0 && n64 < 65535 && E64 < 65535 && need(n64, E64) <= a.lds_bytes;
        if (fits && a.ids_are_global)
            while (g2 < g_hi) {                          // (graph-local ids: one graph per chunk -- the offset of a column depends on its graph)
                const int64_t n2 = a.node_ptr[g2 + 1] - n0, E2 = a.edge_ptr[g2 + 1] - e0;
                if (n2 < n64 || E2 < E64 || n2 >= 65535 || E2 >= 65535 || need(n2, E2) > a.lds_bytes) break;
                n64 = n2; E64 = E2; ++g2;
            }
        g = g2;
        if (!fits) {
            // a graph beyond what the launch's LDS sorts (the counting workgroup reports it: GSN_ST_TOO_LARGE): its vertices own no columns, its
            // columns map to themselves (as csr_graphs_kernel); its rows of the packs are encoded by the plain loops
            if (n64 < 0 || E64 < 0) continue;
            if (do_csr) {
                for (int64_t v = tid; v < n64; v += T) sd->csr_seg[n0 + v] = (int32_t)e0;
                for (int64_t e = tid; e < E64; e += T) {
                    sd->csr_perm[e0 + e] = (int32_t)(e0 + e);
                    if (sd->csr_tgt) sd->csr_tgt[e0 + e] = (int32_t)n0;
                    if (sd->csr_oth) sd->csr_oth[e0 + e] = (int32_t)n0;
                }
            }
            if (do_np) {
                const unsigned w0 = sd->ncode_ptr_w[0], w1 = sd->ncode_ptr_w[1];
                for (int64_t i = tid; i < 4 * n64; i += T) {
                    const int64_t v = i >> 2;
                    const int q = (int)(i & 3);
                    unsigned m = 0x80000000u;
                    for (int c = 0; c < sd->ncode_cols; ++c) {
                        int64_t x = sd->ncode[(n0 + v) * sd->ncode_cols + c];
                        const int lo = side_byte(w0, w1, c), ncls = side_byte(w0, w1, c + 1) - lo;
                        if (sd->ncode_clamp) x = x < 0 ? 0 : (x >= ncls ? ncls - 1 : x);
                        if (x >= 0 && x < ncls) m |= 1u << (lo + (int)x);
                        else if (sd->code_status && q == 0) atomicOr(sd->code_status, 1);
                    }
                    const unsigned hot = m >> (8 * q);
                    auto word = [&](int kk) { return ((hot >> kk) & 1u ? 0x3c00u : 0u) | ((hot >> (kk + 1)) & 1u ? 0x3c000000u : 0u); };
                    *reinterpret_cast<u4v *>(sd->npack + (n0 + v) * 32 + 8 * q) = u4v{word(0), word(2), word(4), word(6)};
                }
            }
            if (do_ec) {
                const unsigned w0 = sd->ecode_ptr_w[0], w1 = sd->ecode_ptr_w[1];
                const int q0 = sd->ecode_col0 >> 2;
                for (int64_t i = tid; i < E64 * (4 - q0); i += T) {
                    const int64_t r = i / (4 - q0);
                    const int q = q0 + (int)(i - r * (4 - q0));
                    unsigned hot = 0;
                    for (int c = 0; c < sd->ecode_cols; ++c) {
                        int64_t x = sd->ecode[(e0 + r) * sd->ecode_cols + c];
                        const int lo = side_byte(w0, w1, c), ncls = side_byte(w0, w1, c + 1) - lo;
                        if (sd->ecode_clamp) x = x < 0 ? 0 : (x >= ncls ? ncls - 1 : x);
                        if (x >= 0 && x < ncls) hot |= 1u << (lo + (int)x);
                        else if (sd->code_status && q == q0) atomicOr(sd->code_status, 1);
                    }
                    hot >>= 4 * q;
                    uint2 o;
                    o.x = ((hot & 1u) ? 0x3c00u : 0u) | ((hot & 2u) ? 0x3c000000u : 0u);
                    o.y = ((hot & 4u) ? 0x3c00u : 0u) | ((hot & 8u) ? 0x3c000000u : 0u);
                    *reinterpret_cast<uint2 *>(sd->epack + (e0 + r) * 16 + 4 * q) = o;
                }
            }
            continue;
        }
        const int n = (int)n64, E = (int)E64;
        int *cstart = reinterpret_cast<int *>(smem);
        int *ccur = cstart + (n + 1);
        unsigned *nmask = reinterpret_cast<unsigned *>(ccur + (n + 1));
        uint16_t *tloc = reinterpret_cast<uint16_t *>(nmask + n);
        uint16_t *oloc = tloc + E;
        uint16_t *pl = oloc + E;
        unsigned char *ecls = reinterpret_cast<unsigned char *>(pl + E);      // bit mask of the edge codes' classes over pack columns ecode_col0 .. +7
        __syncthreads();                                 // (the chunk before this one has left LDS)
        for (int v = tid; v <= n; v += T) cstart[v] = 0;
        __syncthreads();
        // ---- (A) loads: four per lane in flight --------------------------------------------------------------------------
        const int64_t off = a.ids_are_global ? n0 : 0;   // (graph-local ids: the chunk is one graph)
        bool bad_code = false;
        if (do_csr)
            for (int c0 = 0; c0 < E; c0 += 4 * T) {
                int64_t tq[4], oq[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int c = c0 + j * T + tid; tq[j] = c < E ? trow[e0 + c] : off; oq[j] = c < E ? orow[e0 + c] : off; }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = c0 + j * T + tid;
                    if (c < E) {
                        const int64_t t = tq[j] - off, o = oq[j] - off;
                        const int tl = (t >= 0 && t < n) ? (int)t : 0, ol = (o >= 0 && o < n) ? (int)o : 0;     // (a column that leaves its graph: the counting workgroup reports it)
                        tloc[c] = (uint16_t)tl; oloc[c] = (uint16_t)ol;
                        atomicAdd(&cstart[tl], 1);
                    }
                }
            }
        if (do_np) {
            const unsigned w0 = sd->ncode_ptr_w[0], w1 = sd->ncode_ptr_w[1];
            const int nc = sd->ncode_cols;
            for (int v0 = 0; v0 < n; v0 += 4 * T) {
                if (nc == 1) {
                    int64_t xq[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const int v = v0 + j * T + tid; xq[j] = v < n ? sd->ncode[n0 + v] : 0; }
                    const int ncls = side_byte(w0, w1, 1);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int v = v0 + j * T + tid;
                        int64_t x = xq[j];
                        if (sd->ncode_clamp) x = x < 0 ? 0 : (x >= ncls ? ncls - 1 : x);
                        const bool ok = x >= 0 && x < ncls;
                        if (v < n) { nmask[v] = 0x80000000u | (ok ? 1u << (int)x : 0u); bad_code = bad_code || !ok; }
                    }
                } else {
                    for (int j = 0; j < 4; ++j) {
                        const int v = v0 + j * T + tid;
                        if (v >= n) break;
                        unsigned m = 0x80000000u;
                        for (int c = 0; c < nc; ++c) {
                            int64_t x = sd->ncode[(n0 + v) * nc + c];
                            const int lo = side_byte(w0, w1, c), ncls = side_byte(w0, w1, c + 1) - lo;
                            if (sd->ncode_clamp) x = x < 0 ? 0 : (x >= ncls ? ncls - 1 : x);
                            if (x >= 0 && x < ncls) m |= 1u << (lo + (int)x); else bad_code = true;
                        }
                        nmask[v] = m;
                    }
                }
            }
        }
        if (do_ec) {
            const unsigned w0 = sd->ecode_ptr_w[0], w1 = sd->ecode_ptr_w[1];
            const int nc = sd->ecode_cols, c00 = sd->ecode_col0;
            for (int r0 = 0; r0 < E; r0 += 4 * T) {
                if (nc == 1) {
                    int64_t xq[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const int r = r0 + j * T + tid; xq[j] = r < E ? sd->ecode[e0 + r] : 0; }
                    const int ncls = side_byte(w0, w1, 1) - c00;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = r0 + j * T + tid;
                        int64_t x = xq[j];
                        if (sd->ecode_clamp) x = x < 0 ? 0 : (x >= ncls ? ncls - 1 : x);
                        const bool ok = x >= 0 && x < ncls;
                        if (r < E) { ecls[r] = ok ? (unsigned char)(1u << (int)x) : (unsigned char)0; bad_code = bad_code || !ok; }
                    }
                } else {
                    for (int j = 0; j < 4; ++j) {
                        const int r = r0 + j * T + tid;
                        if (r >= E) break;
                        unsigned m = 0;
                        for (int c = 0; c < nc; ++c) {
                            int64_t x = sd->ecode[(e0 + r) * nc + c];
                            const int lo = side_byte(w0, w1, c) - c00, ncls = side_byte(w0, w1, c + 1) - c00 - lo;
                            if (sd->ecode_clamp) x = x < 0 ? 0 : (x >= ncls ? ncls - 1 : x);
                            if (x >= 0 && x < ncls) m |= 1u << (lo + (int)x); else bad_code = true;
                        }
                        ecls[r] = (unsigned char)m;
                    }
                }
            }
        }
        __syncthreads();
        // ---- (B) the sort, in LDS ----------------------------------------------------------------------------------------
        if (do_csr) {
            if (tid < 64) {                              // exclusive scan of the n + 1 counters, 64 at a time (start[n] becomes E)
                int carry = 0;
                for (int base = 0; base <= n; base += 64) {
                    const int v = base + tid;
                    const int cnt = v <= n ? cstart[v] : 0;
                    int incl = cnt;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const int t = __shfl_up(incl, o);
                        if (tid >= o) incl += t;
                    }
                    const int ex = carry + incl - cnt;
                    if (v <= n) { cstart[v] = ex; ccur[v] = ex; }
                    carry += __shfl(incl, 63);
                }
            }
            __syncthreads();
            for (int c = tid; c < E; c += T) pl[atomicAdd(&ccur[tloc[c]], 1)] = (uint16_t)c;
            __syncthreads();
            for (int v = tid; v < n; v += T) {          // column order inside every segment (the cursor hands out slots in arbitrary order)
                const int lo = cstart[v], hi = cstart[v + 1];
                for (int i = lo + 1; i < hi; ++i) {
                    const uint16_t x = pl[i];
                    int j = i - 1;
                    while (j >= lo && pl[j] > x) { pl[j + 1] = pl[j]; --j; }
                    pl[j + 1] = x;
                }
            }
            __syncthreads();
        }
        // ---- (C) stores --------------------------------------------------------------------------------------------------
        if (bad_code && sd->code_status) atomicOr(sd->code_status, 1);
        if (do_csr) {
            for (int v = tid; v < n; v += T) sd->csr_seg[n0 + v] = (int32_t)(e0 + cstart[v]);
            for (int i = tid; i < E; i += T) {
                const int le = pl[i];
                sd->csr_perm[e0 + i] = (int32_t)(e0 + le);
                if (sd->csr_tgt) sd->csr_tgt[e0 + i] = (int32_t)(n0 + tloc[le]);
                if (sd->csr_oth) sd->csr_oth[e0 + i] = (int32_t)(n0 + oloc[le]);
            }
        }
        if (do_np)
            for (int i = tid; i < 4 * n; i += T) {
                const int v = i >> 2, q = i & 3;
                const unsigned hot = nmask[v] >> (8 * q);
                auto word = [&](int kk) { return ((hot >> kk) & 1u ? 0x3c00u : 0u) | ((hot >> (kk + 1)) & 1u ? 0x3c000000u : 0u); };
                *reinterpret_cast<u4v *>(sd->npack + (n0 + v) * 32 + 8 * q) = u4v{word(0), word(2), word(4), word(6)};
            }
        if (do_ec) {
            const int q0 = sd->ecode_col0 >> 2, nq = 4 - q0;      // the 4-column groups from ecode_col0 to the end of the row
            for (int i = tid; i < E * nq; i += T) {
                const int r = nq == 1 ? i : i / nq;
                const int qq = i - r * nq;
                const unsigned hot = (unsigned)ecls[r] >> (4 * qq);
                uint2 o;
                o.x = ((hot & 1u) ? 0x3c00u : 0u) | ((hot & 2u) ? 0x3c000000u : 0u);
                o.y = ((hot & 4u) ? 0x3c00u : 0u) | ((hot & 8u) ? 0x3c000000u : 0u);
                *reinterpret_cast<uint2 *>(sd->epack + (e0 + r) * 16 + 4 * (q0 + qq)) = o;
            }
        }
    }
}

// diagnostic build (-DCOUNT_PROF, scripts/rr_variant.sh with RR_VARIANT_SRC=count): cycles of thread 0 per phase, summed over the workgroups;
// with -DCOUNT_PROF_STEP as well: the arms inside lane_step (count_core.h: Lane::prof -- 24 more registers per lane: for the large-graph
// instantiations, it distorts the molecule ones)
#ifdef COUNT_PROF
__device__ unsigned long long *g_count_prof;   // [items][8], set by the launcher
__device__ unsigned long long *g_count_prof2;  // [12] wave-level sums over the launch: the arms inside lane_step (count_core.h: Lane::prof)
#define COUNT_T(I) do { __syncthreads(); if (threadIdx.x == 0 && g_count_prof) { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); g_count_prof[(size_t)blockIdx.x * 16 + (I)] = t_now - t_prev; t_prev = t_now; } } while (0)
#else
#define COUNT_T(I)
#endif

// One pass of a workgroup over `ng` consecutive graphs g .. g + ng - 1 taken as ONE graph, their disjoint union (ng = 2 only for plans
// whose patterns are connected: an image of a connected pattern lies inside one component, so every rooted count of the union is the
// count in the root's own graph; vertex ids of the second graph follow the first's, rows and columns are contiguous in the batch
// anyway).  Two ZINC-sized graphs fill the 64 lanes of the wave that had one before: the set-up phases (adjacency, cores, distance
// tables, edge ranks: latency of dependent LDS round trips, half of a workgroup's life) are paid once per pair, and the task pool
// keeps 64 lanes busy twice as long.  report = false (a pair): nothing is reported, a pair that does not fit or holds an error
// returns 1 and the caller redoes its graphs one by one.
// MOL: the launch's invariants of the molecule datasets' identifier pass as compile-time constants -- edge mode with undirected orbit
// classes (symmetric rows), four output columns, one workgroup per pair of graphs (no split), encoded rows staged as class
// indices -- instead of ~12 launch arguments that are tested in every arm of the pool and held in scalar registers
// through the whole kernel (106 of them + spills before).  The launcher selects it when all of that holds (launch<>()).
template <int W, int T, bool DIR, bool TAIL, bool MOL>
__device__ __forceinline__ int count_body(const CountArgs &a, unsigned char *smem, const int item, const int part, const int g, const int ng, const bool report) {
#ifdef COUNT_PROF
    unsigned long long t_prev = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();                                    // (a pass before this one has finished with LDS)
    uint64_t *A = reinterpret_cast<uint64_t *>(smem);
    uint64_t *valid = reinterpret_cast<uint64_t *>(smem + a.off_valid);
    uint64_t *stack = reinterpret_cast<uint64_t *>(smem + a.off_stack) - (size_t)a.stack_skip * W * T;   // (frame l at stack + l W T: see the launcher)
    uint32_t *plan = reinterpret_cast<uint32_t *>(smem + a.off_plan);
    typedef typename std::conditional<(W <= 4), uint8_t, uint16_t>::type vid_t;   // column endpoints
    vid_t *eu = reinterpret_cast<vid_t *>(smem + a.off_eu);
    vid_t *ev = reinterpret_cast<vid_t *>(smem + a.off_ev);
    // (prefix sums of the distinct-neighbour degrees: one-word graphs -- at most 64 x 63 pairs -- keep them as 16-bit values: 128 bytes of
    //  a molecule pair's 6.9 KiB, the difference between 23 and 24 workgroups per CU)
    typedef typename std::conditional<(W == 1), uint16_t, int>::type rs_t;
    rs_t *rowstart = reinterpret_cast<rs_t *>(smem + a.off_rowstart);
    int *last = reinterpret_cast<int *>(smem + a.off_last);
    // (column tables as 16-bit values: a workgroup whose per-column state fits LDS has far fewer than 65535 columns; 0xffff = none)
    uint16_t *prim = reinterpret_cast<uint16_t *>(smem + a.off_prim);
    uint16_t *revof = reinterpret_cast<uint16_t *>(smem + a.off_revof);
    uint16_t *out_lds = reinterpret_cast<uint16_t *>(smem + a.off_out);   // staged counts as 16-bit values; 0xffff = this cell went to HBM by itself
    int *misc = reinterpret_cast<int *>(smem + a.off_misc);  // [0] next task  [1] n_active  [2] status
    uint64_t *balls = a.off_ball >= 0 ? reinterpret_cast<uint64_t *>(smem + a.off_ball) : nullptr;
    uint64_t *cores = reinterpret_cast<uint64_t *>(smem + a.off_core);   // [CORE_MAX + 1][W]
    uint64_t *degp = (TAIL && a.off_degp >= 0) ? reinterpret_cast<uint64_t *>(smem + a.off_degp) : nullptr;     // (TAIL = false: none of this exists)
    uint64_t *A_in = DIR ? reinterpret_cast<uint64_t *>(smem + a.off_ain) : nullptr;
    const int *enc = reinterpret_cast<const int *>(smem + a.off_enc);   // [2 * n_cols] (encoded output only)

    const int tid = threadIdx.x;
    (void)item;
    const int64_t n0 = a.node_ptr[g], e0 = a.edge_ptr[g];
    const int64_t n64 = a.node_ptr[g + ng] - n0, E64 = a.edge_ptr[g + ng] - e0;
    const int64_t nA64 = a.node_ptr[g + 1] - n0, EA64 = a.edge_ptr[g + 1] - e0;      // the first graph's share (ng = 1: everything)
    const bool edge_mode = MOL || a.mode == GSN_MODE_EDGE;
    const int64_t rows64 = edge_mode ? E64 : n64;
    const int64_t row0 = edge_mode ? e0 : n0;
    const int n_cols = MOL ? 4 : a.n_cols;
    const bool a_stage_out = a.stage_out != 0, a_sym = MOL || a.sym != 0, a_enc = MOL || a.enc_out != nullptr, a_enc_stage = MOL || a.enc_stage != 0;
    const bool a_enc_from_counts = a.enc_from_counts != 0;
    const int a_split = MOL ? 1 : a.split;

    if (ng > 1 && (n64 > a.n_cap || E64 > a.e_cap || n64 > W * 64)) return 1;
    if (ng == 1 && (n64 > a.n_decl || E64 > a.e_decl || n64 > W * 64)) {
        // caller under-declared max_nodes / max_edges: report, leave zeros
        if (part == 0) {
            if (a.out) for (int64_t i = tid; i < rows64 * n_cols; i += T) a.out[row0 * n_cols + i] = 0;
            if (a.enc_out && !a.enc_no32) for (int64_t i = tid; i < rows64 * a.enc_width; i += T) a.enc_out[row0 * a.enc_width + i] = 0.f;
            if (a.enc16) for (int64_t i = tid; i < rows64 * a.enc_width; i += T) a.enc16[(row0 + i / a.enc_width) * a.enc16_stride + a.enc16_col0 + i % a.enc_width] = 0;
            if (tid == 0) atomicMax(&a.status[g], (int)GSN_ST_TOO_LARGE);
        }
        return 0;
    }
    const int n = (int)n64, E = (int)E64, rows = (int)rows64;
    const int nA = (int)nA64, EA = (int)EA64;

    // one finished cell (row, column) = cnt: the int64 row (staged or direct), the staged class index or the encoded floats
    auto emit_cell = [&](int row, int col, uint64_t cnt) {
        if (a_stage_out) {
            // two bytes per staged cell (eight cost a ZINC pair 5.6 KiB: more than its LDS budget, so every cell went to HBM by itself from
            // inside the pool -- 12.5 M scattered 8-byte stores per 65 536 molecules); the rare count that does not fit goes out directly
            if (cnt < 0xffffull) out_lds[row * n_cols + col] = (uint16_t)cnt;
            else { out_lds[row * n_cols + col] = (uint16_t)0xffff; a.out[(row0 + row) * n_cols + col] = (int64_t)cnt; }
        }
        else if (a.out) a.out[(row0 + row) * n_cols + col] = (int64_t)cnt;
        if (a_enc) {
            const int *enc_t = reinterpret_cast<const int *>(smem + a.off_enc);
            const int eo = enc_t[2 * col], ncls = enc_t[2 * col + 1];
            uint64_t v = cnt;
            if (a.enc_clamp && v >= (uint64_t)ncls) v = (uint64_t)(ncls - 1);
            if (a_enc_stage) {
                if (!a_enc_from_counts) (smem + a.off_encst)[row * n_cols + col] = v < (uint64_t)ncls ? (unsigned char)v : (unsigned char)0xff;
            } else {
                float *d0 = a.enc_out + (row0 + row) * a.enc_width + eo;
                for (int j = 0; j < ncls; ++j) d0[j] = (uint64_t)j == v ? 1.f : 0.f;
            }
        }
    };

    // ---- phase 0: clear LDS state, copy the plan table ------------------------------------------------------------
    for (int i = tid; i < n * W; i += T) A[i] = 0ull;
    if (DIR)
        for (int i = tid; i < n * W; i += T) A_in[i] = 0ull;
    for (int i = tid; i < a.plan_words; i += T) plan[i] = a.plan[i];
    if (a_enc)                                          // (first float, n_classes) per column
        for (int c = tid; c < n_cols; c += T) {
            int o = 0;
            for (int x = 0; x < c; ++x) o += a.enc_n[x];
            reinterpret_cast<int *>(smem + a.off_enc)[2 * c] = o;
            reinterpret_cast<int *>(smem + a.off_enc)[2 * c + 1] = a.enc_n[c];
        }
    if (tid < 8) misc[tid] = 0;
    // staged outputs start at zero: rows that carry nothing (self loops, earlier duplicates) and rows whose roots lie outside every plan's core
    // then need no cell-by-cell zeros from the one wave that ranks the columns (8 LDS stores per such row: 7 k of a molecule pair's 60 k cycles)
    const bool bulk_zero = edge_mode && a_stage_out && (!a_enc || a_enc_stage);
    if (bulk_zero) {
        uint32_t *z = reinterpret_cast<uint32_t *>(out_lds);
        for (int i = tid; i < (rows * n_cols + 1) / 2; i += T) z[i] = 0u;
        if (a_enc && !a_enc_from_counts) {
            unsigned char *zb = smem + a.off_encst;
            for (int i = tid; i < rows * n_cols; i += T) zb[i] = 0;      // (count 0 = class 0)
        }
    }
    __syncthreads();

    COUNT_T(0);
    // ---- phase 1: adjacency bit matrix from the columns -----------------------------------------------------------
    for (int c = tid; c < E; c += T) {
        const bool second = c >= EA;                    // (a column of the pair's second graph: its vertices are nA .. n - 1)
        const int64_t off = a.ids_are_global ? n0 : (second ? -(int64_t)nA : 0);
        const int64_t u64 = a.src[e0 + c] - off, v64 = a.dst[e0 + c] - off;
        const int64_t lo = second ? nA : 0, hi = second ? n : nA;
        if (u64 < lo || v64 < lo || u64 >= hi || v64 >= hi) {
            atomicMax(&misc[2], (int)GSN_ST_BAD_INDEX);
            if (edge_mode) { eu[c] = 0; ev[c] = 0; }
            continue;
        }
        const int u = (int)u64, v = (int)v64;
        if (edge_mode) { eu[c] = (vid_t)u; ev[c] = (vid_t)v; }
        atomicMax(&misc[second ? 5 : 1], (u > v ? u : v) + 1);  // graph-tool creates vertices 0..max id, self-loop columns included
        if (u != v) {
            atomicOr(reinterpret_cast<unsigned long long *>(&A[u * W + (v >> 6)]), 1ull << (v & 63));
            atomicOr(reinterpret_cast<unsigned long long *>(&(DIR ? A_in : A)[v * W + (u >> 6)]), 1ull << (u & 63));
        }
    }
    __syncthreads();
    COUNT_T(1);
    // the vertices that exist: 0 .. largest id of the first graph, nA .. largest id of the second
    if (tid < W) valid[tid] = below_word(misc[1], tid) | (below_word(misc[5], tid) & ~below_word(nA, tid));
    // d-cores: every image of a pattern with minimum degree d lies in the d-core of the graph (its >= d pattern neighbours
    // are images too), so the plan's candidate universe is that core instead of all vertices -- on molecules the 2-core
    // (ring systems and what connects them) is a fraction of the graph and rooted searches from the rest end at once.
    __syncthreads();
    for (int d = 0; d <= CORE_MAX; ++d) {
        if (tid < W) cores[d * W + tid] = valid[tid];
    }
    __syncthreads();
    for (int d = 1; d <= CORE_MAX; ++d) {
        if (!((a.core_mask >> d) & 1)) continue;
        uint64_t *core = cores + d * W;
        if (W == 1 && T == 64) {                    // one wave, one vertex per lane: peel with ballots, no LDS round trips
            uint64_t cur = core[0];
            for (;;) {
                const bool in = (cur >> tid) & 1ull;
                const bool keep = in && tid < n && popc64(A[tid] & cur) >= d;
                const uint64_t nxt = __ballot(keep);
                if (nxt == cur) break;
                cur = nxt;
            }
            __syncthreads();
            if (tid == 0) core[0] = cur;
        } else {
            for (;;) {                               // Jacobi peeling: decide on a snapshot, then remove
                __syncthreads();
                if (tid == 0) misc[3] = 0;
                __syncthreads();
                uint32_t dropm = 0;                  // vertices tid + i*T, i < 12 (n <= 768, T >= 64)
                for (int v = tid, i = 0; v < n; v += T, ++i) {
                    const bool in = (core[v >> 6] >> (v & 63)) & 1ull;
                    if (in && !core_keeps<W>(A, core, v, d)) dropm |= 1u << i;
                }
                __syncthreads();
                for (int v = tid, i = 0; v < n; v += T, ++i)
                    if ((dropm >> i) & 1u) atomicAnd(reinterpret_cast<unsigned long long *>(&core[v >> 6]), ~(1ull << (v & 63)));
                if (dropm) misc[3] = 1;
                __syncthreads();
                if (misc[3] == 0) break;
            }
        }
    }
    __syncthreads();
    // degree bit planes of the cores that hold a chain-tail plan (count_core.h: tail_pairs, mode 3)
    if (TAIL && degp) {
        for (int i = tid; i < (CORE_MAX + 1) * DEG_PLANES * W; i += T) degp[i] = 0ull;
        __syncthreads();
        for (int d = 0; d <= CORE_MAX; ++d) {
            if (!((a.degp_mask >> d) & 1)) continue;
            for (int v = tid; v < n; v += T)
                deg_planes_vertex<W>(A, cores + d * W, v, degp + d * DEG_PLANES * W,
                                     [](uint64_t *wp, uint64_t bit) { atomicOr(reinterpret_cast<unsigned long long *>(wp), (unsigned long long)bit); });
        }
        __syncthreads();
    }
    COUNT_T(2);
    // distance pruning tables: vertices within 2 / 3 hops (count_core.h, candidates())
    if (balls) {
        for (int v = tid; v < n; v += T) ball_expand<W>(A, nullptr, v, balls);
        __syncthreads();
        for (int v = tid; v < n; v += T) ball_expand<W>(A, balls, v, balls + a.n_cap * W);
    }

    COUNT_T(3);
    // ---- phase 2 (edge mode): CSR rank of every directed pair, last-duplicate-wins column, the rows that search ---------
    if (edge_mode) {
        if (W == 1 && T == 64) {                 // one wave, n <= 64: the row starts are a wave prefix sum of the degrees
            const int d = tid < n ? popc64(A[tid]) : 0;
            int incl = d;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o);
                if (tid >= o) incl += t;
            }
            if (tid <= n) rowstart[tid] = (rs_t)(incl - d);
            if (tid == 63 && n == 64) rowstart[64] = (rs_t)incl;
        } else {
            for (int u = tid; u <= n; u += T) {
                int s = 0;
                for (int x = 0; x < u; ++x) {
#pragma unroll
                    for (int w = 0; w < W; ++w) s += popc64(A[x * W + w]);
                }
                rowstart[u] = (rs_t)s;
            }
        }
        for (int i = tid; i < E; i += T) last[i] = -1;
        __syncthreads();
        for (int c = tid; c < E; c += T) {
            const int u = eu[c], v = ev[c];
            if (u == v) continue;
            int r = rowstart[u];
#pragma unroll
            for (int w = 0; w < W; ++w) r += popc64(A[u * W + w] & below_word(v, w));
            revof[c] = (uint16_t)r;              // (its slot, until the pass below replaces it)
            atomicMax(&last[r], c);
        }
        __syncthreads();
        // Per column, once (not per task): is it the column that carries the counts of its pair (the last duplicate,
        // utils_graph_processing.py:142-144), which column carries the reverse pair, and does it search at all?  Undirected orbit
        // classes: every map that puts a pattern edge (a,b) on (u,v) puts (b,a), same class, on (v,u), so the two rows are equal
        // -- the u < v row searches and writes both.  Rows that carry nothing (self loops, earlier duplicates) get their zeros
        // here.  The searching rows are compacted IN COLUMN ORDER by one wave: every workgroup of a split graph enumerates the
        // same task list.
        if (tid < 64) {
            int n_prim = 0;
            for (int c0 = 0; c0 < E; c0 += 64) {
                const int c = c0 + tid;
                bool primary = false;
                if (c < E) {
                    const int u = eu[c], v = ev[c];
                    bool live = false;
                    int rev = -1;
                    if (u != v) {
                        live = last[revof[c]] == c;
                        int rr = rowstart[v];
#pragma unroll
                        for (int w = 0; w < W; ++w) rr += popc64(A[v * W + w] & below_word(u, w));
                        rev = last[rr];
                    }
                    revof[c] = rev < 0 ? (uint16_t)0xffffu : (uint16_t)rev;
                    primary = live && !(a_sym && rev >= 0 && u > v);
                    if (!live && part == 0 && !bulk_zero)
                        for (int col = 0; col < n_cols; ++col) emit_cell(c, col, 0ull);
#ifndef COUNT_NO_CORE_FILTER
                    if (primary) {
                        // A root outside the WEAKEST core any plan of the launch lives in is the image of nothing (cores are nested: it is outside
                        // every plan's core; lane_begin would find that out one task at a time -- a pull, a begin and a finish per cell).  Its
                        // cells, and those of the reverse row it would have written, are zero here and the row never enters the pool: on
                        // molecules (cycle plans: the 2-core = ring systems and what connects them) more than half of the rows.
                        const uint64_t *cm = cores + (__ffs(a.core_mask | (1 << CORE_MAX)) - 1) * W;
                        if (!(((cm[u >> 6] >> (u & 63)) & (cm[v >> 6] >> (v & 63))) & 1ull)) {
                            primary = false;
                            if (part == 0 && !bulk_zero)
                                for (int col = 0; col < n_cols; ++col) {
                                    emit_cell(c, col, 0ull);
                                    if (a_sym && rev >= 0) emit_cell(rev, col, 0ull);
                                }
                        }
                    }
#endif
                }
                const uint64_t pm = __ballot(primary);
                if (primary) prim[n_prim + __popcll(pm & ((1ull << tid) - 1ull))] = (uint16_t)c;
                n_prim += __popcll(pm);
            }
            if (tid == 0) misc[3] = n_prim;
        }
    }
    __syncthreads();
    if (misc[2] != 0) {  // bad index: zeros + status
        if (!report) return 1;
        if (part == 0) {
            if (a.out) for (int i = tid; i < rows * n_cols; i += T) a.out[row0 * n_cols + i] = 0;
            if (a.enc_out && !a.enc_no32) for (int i = tid; i < rows * a.enc_width; i += T) a.enc_out[row0 * a.enc_width + i] = 0.f;
            if (a.enc16 && a.enc_no32) for (int i = tid; i < rows * a.enc_width; i += T) a.enc16[(row0 + i / a.enc_width) * a.enc16_stride + a.enc16_col0 + i % a.enc_width] = 0;
            if (tid == 0) atomicMax(&a.status[g], misc[2]);
        }
        return 0;
    }

    COUNT_T(4);
    // ---- phase 3: task pool -- (column, row) cells, pulled by lanes as they go idle --------------------------------
    // this workgroup takes the tasks  part, part + split, part + 2*split, ...  (strided, so the heavy columns of a
    // pattern family are spread over all the workgroups of a graph); n_tasks = how many of them
    const int n_div = edge_mode ? misc[3] : rows;            // rows that run searches
    const int n_tasks_all = n_div * n_cols;
    const int n_tasks = n_tasks_all > part ? (n_tasks_all - part + a_split - 1) / a_split : 0;
    const float div_rcp = n_div > 0 ? 1.0f / (float)n_div : 0.f;
    const bool div_float = n_tasks_all < (1 << 22);        // task index exact in fp32: quotient by one multiply + one correction
    const uint32_t *col_ptr = plan + PLAN_HEADER_WORDS;
    const uint32_t *col_order = col_ptr + n_cols + 1;
    const uint32_t *plans = plan + a.plans_off;
    const int lane = tid & 63;
    const uint64_t lane_lt = (1ull << lane) - 1ull;

#ifdef COUNT_PROF
    unsigned prof_iters = 0, prof_lanes = 0;          // pool loop trips of the wave, lanes inside a rooted search summed over the trips
    unsigned prof_arm[4] = {0, 0, 0, 0};              // cycles of the wave in the pull / begin / step / finish arm
#define COUNT_ARM(I, T0) do { prof_arm[I] += (unsigned)(__builtin_amdgcn_s_memtime() - (T0)); } while (0)
#define COUNT_ARM_T0() __builtin_amdgcn_s_memtime()
#else
#define COUNT_ARM(I, T0)
#define COUNT_ARM_T0() 0ull
#endif
    Lane<W> s;
#ifdef COUNT_PROF_STEP
    for (int q = 0; q < 12; ++q) s.prof[q] = 0;
#endif
    s.l = -1; s.cnt = 0; s.k = 0; s.nfix = 0; s.fvec = fv_roots<W>(0, 0); s.plan = plans;
    s.balls = balls; s.ball_n = a.n_cap; s.degp = nullptr; s.loop = 0;
    if (TAIL) { s.degp = degp; s.loop = a.tail_loop; }
#pragma unroll
    for (int w = 0; w < W; ++w) s.used.w[w] = 0ull;
    bool has_task = false, exhausted = false;
    int t_row = 0, t_col = 0, p_i = 0, p_e = 0;
    FVec<W> roots = fv_roots<W>(0, 0);
    bool rev_missing = false;
    int mirror_row = -1;
    const uint64_t *lane_valid = valid;     // candidate universe of the lane's current plan (a core of the graph)

    // The loop is a state machine per lane (pull a cell, begin a plan, one search step, finish the cell) and a wave executes every arm
    // some lane is in: on molecules a trip is ~350 instructions of which the step arm is ~100, and the whole pool is ~9 trips of ~4 000
    // cycles (profiles/r05_count_phase_profile.txt) -- dependent LDS round trips, not arithmetic.  So (i) a cell that ends inside a begin
    // or a step is finished in the SAME trip (its own arm behind them), and (ii) idle lanes are refilled in batches: the pull arm (an LDS
    // atomic, the task decode, five dependent table reads) runs when GSN_PULL_BATCH lanes wait or when no lane is inside a search, not
    // in every trip for the one lane that happened to finish.
    for (;;) {
        const bool need = !has_task && !exhausted;
        uint64_t m = __ballot(need);
        if (m && __popcll(m) < a.pull_batch && __ballot(has_task) != 0ull) m = 0ull;
        const unsigned long long arm_t0 = COUNT_ARM_T0();
        if (m) {
            const int leader = __ffsll((unsigned long long)m) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(&misc[0], __popcll(m));
            base = __shfl(base, leader);
            if (need) {
                const int t = base + __popcll(m & lane_lt);
                if (t < n_tasks) {
                    const int tt = part + t * a_split;
                    int t_idx;
                    if (div_float) {
                        t_col = (int)((float)tt * div_rcp);
                        t_idx = tt - t_col * n_div;
                        if (t_idx < 0) { --t_col; t_idx += n_div; }
                        if (t_idx >= n_div) { ++t_col; t_idx -= n_div; }
                    } else {
                        t_col = tt / n_div;
                        t_idx = tt - t_col * n_div;
                    }
                    // columns in the plan compiler's order of falling estimated cost: the lanes that go idle at the end of the pool
                    // should be left with the short searches, not with the long ones
                    t_col = (int)col_order[t_col];
                    has_task = true;
                    s.cnt = 0; s.l = -1;
                    p_i = (int)col_ptr[t_col]; p_e = (int)col_ptr[t_col + 1];
                    rev_missing = false;
                    mirror_row = -1;
                    if (edge_mode) {
                        t_row = prim[t_idx];
                        const int rev = revof[t_row] == 0xffffu ? -1 : (int)revof[t_row];
                        rev_missing = rev < 0;
                        if (a_sym && rev >= 0) mirror_row = rev;
                        roots = fv_roots<W>(eu[t_row], ev[t_row]);
                    } else {
                        t_row = t_idx;
                        if (!((valid[t_row >> 6] >> (t_row & 63)) & 1ull)) p_i = p_e;  // vertex beyond the largest id: not a vertex of the matched graph
                        roots = fv_roots<W>(t_row, 0);
                    }
                } else {
                    exhausted = true;
                }
            }
        }
        if (m) COUNT_ARM(0, arm_t0);
        if (__ballot(has_task) == 0ull) break;
#ifdef COUNT_PROF
        prof_iters += 1; prof_lanes += (unsigned)__popcll(__ballot(has_task && s.l >= 0));
#endif
        if (has_task) {
            if (s.l < 0) {
                if (p_i < p_e) {
                    const unsigned long long t0 = COUNT_ARM_T0();
                    const uint32_t *pl = plans + p_i * (DIR ? PLAN_STRIDE_DIRECTED : PLAN_STRIDE_WORDS);
                    lane_valid = cores + plan_core(pl) * W;
                    lane_begin<W, DIR, TAIL>(s, pl, roots, A, lane_valid, stack, T, tid, A_in);
                    ++p_i;
                    COUNT_ARM(1, t0);
                }
            } else {
                const unsigned long long t0 = COUNT_ARM_T0();
                lane_step<W, DIR, TAIL>(s, A, lane_valid, stack, T, tid, A_in);
                COUNT_ARM(2, t0);
#ifdef COUNT_PROF_STEP
                s.prof[9] += 1; s.prof[10] += (unsigned long long)__popcll(__ballot(1));      // lane_step visits / active lanes
#endif
            }
            const unsigned long long t0f = COUNT_ARM_T0();
            if (s.l < 0 && p_i >= p_e) {
                // cell finished (its last plan ended in this trip's begin or step, or it had none)
                emit_cell(t_row, t_col, s.cnt);
                if (mirror_row >= 0) emit_cell(mirror_row, t_col, s.cnt);
                if (edge_mode && rev_missing && s.cnt != 0) atomicMax(&misc[2], (int)GSN_ST_KEYERROR);
                has_task = false;
                COUNT_ARM(3, t0f);
            }
        }
    }
    __syncthreads();

    COUNT_T(5);
#ifdef COUNT_PROF
#ifdef COUNT_PROF_STEP
    if ((threadIdx.x & 63) == 0 && g_count_prof2) for (int q = 0; q < 12; ++q) atomicAdd(&g_count_prof2[q], s.prof[q]);
#endif
    if (threadIdx.x == 0 && g_count_prof) { g_count_prof[(size_t)blockIdx.x * 16 + 7] = ((unsigned long long)prof_iters << 32) | prof_lanes; for (int q = 0; q < 4; ++q) g_count_prof[(size_t)blockIdx.x * 16 + 8 + q] = prof_arm[q]; }
#endif
    // ---- phase 4: coalesced write of the staged rows --------------------------------------------------------------
    if (a_stage_out) {   // (only with split == 1)
        int64_t *dst = a.out + row0 * n_cols;
        for (int i = tid; i < rows * n_cols; i += T) {
            const uint16_t v = out_lds[i];
            if (v != (uint16_t)0xffff) dst[i] = (int64_t)v;
        }
    }
    // ---- phase 4': encoded rows from the staged class indices, one float per thread and trip, consecutive addresses ----------
    if (a_enc && a_enc_stage) {
        const unsigned char *est = smem + a.off_encst;
        // class index of cell (r, c): the staged byte, or from the staged 16-bit count (0xffff = a count of 65 535 and more: the last class
        // when clamped, none otherwise -- class counts fit a byte here)
        auto cls_from_count = [&](unsigned v, int c) -> int {
            const int ncls = enc[2 * c + 1];
            if (a.enc_clamp) return (int)v >= ncls ? ncls - 1 : (int)v;
            return (int)v < ncls ? (int)v : 0xff;
        };
        float *dst = a.enc_out + row0 * a.enc_width;
        const int total = rows * a.enc_width;
        const bool pack_ok = !a.enc16 || (((a.enc16_stride | a.enc16_col0) & 3) == 0 && (reinterpret_cast<uintptr_t>(a.enc16) & 7) == 0);
        if ((MOL || n_cols == 4) && (a.enc_width & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0 && pack_ok) {
            // four identifier columns (the cycle / clique families of the reference's configurations): a thread owns a row, reads
            // its four class indices as one word and writes the row as float4s
            int hot0 = enc[0], hot1 = enc[2], hot2 = enc[4], hot3 = enc[6];
            for (int r = tid; r < rows; r += T) {
                int c0, c1, c2, c3;
                if (a_enc_from_counts) {
                    const uint2 q = *reinterpret_cast<const uint2 *>(out_lds + 4 * r);
                    c0 = cls_from_count(q.x & 0xffffu, 0); c1 = cls_from_count(q.x >> 16, 1);
                    c2 = cls_from_count(q.y & 0xffffu, 2); c3 = cls_from_count(q.y >> 16, 3);
                } else {
                    const unsigned cw = *reinterpret_cast<const unsigned *>(est + 4 * r);
                    c0 = (int)(cw & 0xffu); c1 = (int)((cw >> 8) & 0xffu); c2 = (int)((cw >> 16) & 0xffu); c3 = (int)(cw >> 24);
                }
                const int h0 = c0 == 0xff ? -1 : hot0 + c0, h1 = c1 == 0xff ? -1 : hot1 + c1;
                const int h2 = c2 == 0xff ? -1 : hot2 + c2, h3 = c3 == 0xff ? -1 : hot3 + c3;
                float4 *d4 = reinterpret_cast<float4 *>(dst + r * a.enc_width);
                if (!a.enc_no32)
                for (int k = 0; k < a.enc_width; k += 4) {
                    float4 o;
                    o.x = (h0 == k || h1 == k || h2 == k || h3 == k) ? 1.f : 0.f;
                    o.y = (h0 == k + 1 || h1 == k + 1 || h2 == k + 1 || h3 == k + 1) ? 1.f : 0.f;
                    o.z = (h0 == k + 2 || h1 == k + 2 || h2 == k + 2 || h3 == k + 2) ? 1.f : 0.f;
                    o.w = (h0 == k + 3 || h1 == k + 3 || h2 == k + 3 || h3 == k + 3) ? 1.f : 0.f;
                    d4[k >> 2] = o;
                }
                if (a.enc16) {                          // the same row as fp16 (1.0 = 0x3c00) into the pack's column range, 8 bytes per store
                    uint2 *p2 = reinterpret_cast<uint2 *>(a.enc16 + (row0 + r) * a.enc16_stride + a.enc16_col0);
                    for (int k = 0; k < a.enc_width; k += 4) {
                        uint2 o;
                        o.x = ((h0 == k || h1 == k || h2 == k || h3 == k) ? 0x3c00u : 0u) | ((h0 == k + 1 || h1 == k + 1 || h2 == k + 1 || h3 == k + 1) ? 0x3c000000u : 0u);
                        o.y = ((h0 == k + 2 || h1 == k + 2 || h2 == k + 2 || h3 == k + 2) ? 0x3c00u : 0u) | ((h0 == k + 3 || h1 == k + 3 || h2 == k + 3 || h3 == k + 3) ? 0x3c000000u : 0u);
                        p2[k >> 2] = o;
                    }
                }
            }
        } else
        // column c owns floats enc[2c] .. enc[2c] + enc[2c + 1] of a row; the table is short: a linear scan per float
        for (int i = tid; i < total; i += T) {
            const int r = i / a.enc_width;              // (exact: the reciprocal multiply was off by one for wide encodings, e.g. width 1000 from row 6100 on; this loop is bound by its stores)
            const int j = i - r * a.enc_width;
            int c = 0;
            while (c + 1 < n_cols && enc[2 * (c + 1)] <= j) ++c;
            const int k = j - enc[2 * c];
            const int cls = a_enc_from_counts ? cls_from_count(out_lds[r * n_cols + c], c) : (int)est[r * n_cols + c];
            const bool hot = k < enc[2 * c + 1] && cls == k;
            if (!a.enc_no32) dst[i] = hot ? 1.f : 0.f;
            if (a.enc16) a.enc16[(row0 + r) * a.enc16_stride + a.enc16_col0 + j] = hot ? (uint16_t)0x3c00 : (uint16_t)0;
        }
    }
    COUNT_T(6);
    if (misc[2] != 0) {
        if (!report) return 1;
        if (tid == 0) atomicMax(&a.status[g], misc[2]);   // status[] starts at zero: zeroed by this workgroup (zero_status) or by the launcher
    }
    return 0;
}

// (Measured and dropped: amdgpu_waves_per_eu(6 / 7) on the molecule instantiation spills 72 / 104 bytes per lane for 1.4 / 2 % on the kernel
//  and 0.5 % on the step; even a bound of 5 -- no tighter than what the allocator picks by itself -- changed its choices: 94 registers and
//  36 bytes of scratch instead of 83 and none, +3 % kernel time.  No occupancy attribute.)
// Register bound of the two-word instantiations (graphs of 65 .. 128 vertices: BASELINE config 5, ER G(128,1000)).  Left to itself the
// allocator takes 136 registers = 3 waves per SIMD for a kernel that is bound by the latency of dependent LDS reads at 0.30 lane activity
// (profiles/r06_count_er128_phase.txt); bounded to 80 (6 waves) it spills 188 bytes per lane and the launch is still 22 % faster:
// 43.6 k -> 53.5 k graphs/s at 2 048 graphs per launch (4 waves 50.0 k, 5: 52.9 k, 8: 52.4 k; scripts/gpu/r6_er_ab.sh).  The molecule
// instantiation has its own bound (COUNT_MOL_WAVES); the wider ones (W >= 4) are left alone: no BASELINE config runs them.
#ifndef COUNT_W2_WAVES
#define COUNT_W2_WAVES 6
#endif
#ifndef COUNT_W1_WAVES
#define COUNT_W1_WAVES 1
#endif
#ifndef COUNT_W4_WAVES
#define COUNT_W4_WAVES 1
#endif
template <int W, int T, bool DIR, bool TAIL, bool MOL = false>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(DIR ? 1 : (W == 1 ? COUNT_W1_WAVES : (W == 2 ? COUNT_W2_WAVES : (W == 4 ? COUNT_W4_WAVES : 1))))))
void count_kernel(CountArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int bid = (int)blockIdx.x;
    if (a.side_mask) {                                   // (launches with side workgroups: one workgroup per item, no graph list)
        const int grp = bid / (SIDE_EVERY + 1), j = bid - grp * (SIDE_EVERY + 1);
        if (j == SIDE_EVERY) {
            const int i0 = grp * SIDE_EVERY, i1 = i0 + SIDE_EVERY < a.n_items ? i0 + SIDE_EVERY : a.n_items;
            const int per = (MOL || a.pair) ? 2 : 1;
            side_block<T>(a, smem, per * i0, per * i1 < a.n_graphs ? per * i1 : a.n_graphs);
            return;
        }
        bid = grp * SIDE_EVERY + j;
        if (bid >= a.n_items) return;
    }
    const int item = MOL ? bid : bid / a.split, part = MOL ? 0 : bid - item * a.split;
    // one call site (the body is inlined once): pass 0 = the graph, or the pair 2 i, 2 i + 1 as one; passes 1, 2 = the pair's graphs one by
    // one when it did not fit or held an error
    const int g0 = (MOL || a.pair) ? 2 * item : (a.graph_ids ? a.graph_ids[item] : item);
    const bool two = (MOL || a.pair) && g0 + 1 < a.n_graphs;
    if (a.zero_status && threadIdx.x == 0) {           // (the thread that raises them later: same address, program order)
        a.status[g0] = 0;
        if (two) a.status[g0 + 1] = 0;
    }
    for (int pass = 0; pass < 3; ++pass) {
        const int g = pass == 2 ? g0 + 1 : g0;
        const int ng = (pass == 0 && two) ? 2 : 1;
        const int rc = count_body<W, T, DIR, TAIL, MOL>(a, smem, item, (MOL || a.pair) ? 0 : part, g, ng, ng == 1);
        if (!two || (pass == 0 && rc == 0)) break;
    }
}

// the molecule instantiation as its own kernel: its register bound is set apart from the generic instantiations
#ifndef COUNT_MOL_WAVES
#define COUNT_MOL_WAVES 6
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(COUNT_MOL_WAVES))) void count_kernel_mol(CountArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int item = (int)blockIdx.x;
    if (a.side_mask) {
        const int grp = item / (SIDE_EVERY + 1), j = item - grp * (SIDE_EVERY + 1);
        if (j == SIDE_EVERY) {
            const int i0 = grp * SIDE_EVERY, i1 = i0 + SIDE_EVERY < a.n_items ? i0 + SIDE_EVERY : a.n_items;
            side_block<64>(a, smem, 2 * i0, 2 * i1 < a.n_graphs ? 2 * i1 : a.n_graphs);
            return;
        }
        item = grp * SIDE_EVERY + j;
        if (item >= a.n_items) return;
    }
    const int g0 = 2 * item;
    const bool two = g0 + 1 < a.n_graphs;
    if (a.zero_status && threadIdx.x == 0) {
        a.status[g0] = 0;
        if (two) a.status[g0 + 1] = 0;
    }
    for (int pass = 0; pass < 3; ++pass) {
        const int g = pass == 2 ? g0 + 1 : g0;
        const int ng = (pass == 0 && two) ? 2 : 1;
        const int rc = count_body<1, 64, false, false, true>(a, smem, item, 0, g, ng, ng == 1);
        if (!two || (pass == 0 && rc == 0)) break;
    }
}

__global__ void status_zero_kernel(const int32_t *graph_ids, int n, int32_t *status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) status[graph_ids[i]] = 0;
}

static inline int align_up(int x, int a) { return (x + a - 1) / a * a; }

template <int W, int T, bool DIR, bool TAIL, bool MOL = false>
static int launch_d(CountArgs &a, int n_items, size_t lds, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&count_kernel<W, T, DIR, TAIL, MOL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return set_error(GSN_E_HIP, "hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
    hipLaunchKernelGGL((count_kernel<W, T, DIR, TAIL, MOL>), dim3((unsigned)n_items), dim3(T), lds, stream, a);
    e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "count_kernel launch: %s", hipGetErrorString(e));
#ifdef COUNT_PROF
    {
        (void)hipStreamSynchronize(stream);
        static int shown = 0;
        if (shown++ % 16 == 15) {
            unsigned long long *buf = nullptr;
            (void)hipMalloc(&buf, (size_t)n_items * 128);
            (void)hipMemset(buf, 0, (size_t)n_items * 128);
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_count_prof), &buf, sizeof(buf));
            unsigned long long *buf2 = buf + (size_t)n_items * 16 - 16;      // (the last item's slots 16..: its own row is read first)
            (void)hipMalloc(&buf2, 128); (void)hipMemset(buf2, 0, 128);
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_count_prof2), &buf2, sizeof(buf2));
            hipLaunchKernelGGL((count_kernel<W, T, DIR, TAIL, MOL>), dim3((unsigned)n_items), dim3(T), lds, stream, a);
            (void)hipStreamSynchronize(stream);
            unsigned long long *h = new unsigned long long[(size_t)n_items * 16];
            (void)hipMemcpy(h, buf, (size_t)n_items * 128, hipMemcpyDeviceToHost);
            double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            double it_sum = 0, ln_sum = 0, arm[4] = {0, 0, 0, 0};
            for (int i = 0; i < n_items; ++i) {
                for (int q = 0; q < 7; ++q) s[q] += (double)h[(size_t)i * 16 + q];
                it_sum += (double)(h[(size_t)i * 16 + 7] >> 32); ln_sum += (double)(h[(size_t)i * 16 + 7] & 0xffffffffull);
                for (int q = 0; q < 4; ++q) arm[q] += (double)h[(size_t)i * 16 + 8 + q];
            }
            fprintf(stderr, "countprof pool: %.1f loop trips per workgroup (wave 0), %.1f lanes of 64 inside a search per trip; cycles per workgroup in the arms: pull %.0f begin %.0f step %.0f finish %.0f\n",
                    it_sum / n_items, it_sum > 0 ? ln_sum / it_sum : 0.0, arm[0] / n_items, arm[1] / n_items, arm[2] / n_items, arm[3] / n_items);
            fprintf(stderr, "countprof W %d T %d items %d: cycles per workgroup: clear+plan %.0f adjacency %.0f cores %.0f balls %.0f edge ranks %.0f task pool %.0f write %.0f\n", W, T, n_items,
                    s[0] / n_items, s[1] / n_items, s[2] / n_items, s[3] / n_items, s[4] / n_items, s[5] / n_items, s[6] / n_items);
            delete[] h;
            {
                unsigned long long p2[16];
                (void)hipMemcpy(p2, buf2, 96, hipMemcpyDeviceToHost);
                fprintf(stderr, "countprof step arms (summed over the launch's waves): tail_loop %.3g cycles in %.3g visits, %.1f lanes per visit, %.1f iterations per lane, %.1f = the longest lane's per visit | "
                                "tail_pairs %.3g cycles in %.3g visits, %.1f lanes per visit | lane_step %.3g visits, %.1f lanes per visit\n",
                        (double)p2[0], (double)p2[1], p2[1] ? (double)p2[2] / p2[1] : 0.0, p2[2] ? (double)p2[3] / p2[2] : 0.0, p2[1] ? (double)p2[4] / p2[1] : 0.0,
                        (double)p2[5], (double)p2[6], p2[6] ? (double)p2[7] / p2[6] : 0.0, (double)p2[9], p2[9] ? (double)p2[10] / p2[9] : 0.0);
                unsigned long long *nul2 = nullptr;
                (void)hipMemcpyToSymbol(HIP_SYMBOL(g_count_prof2), &nul2, sizeof(nul2));
                (void)hipFree(buf2);
            }
            unsigned long long *nul = nullptr;
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_count_prof), &nul, sizeof(nul));
            (void)hipFree(buf);
        }
    }
#endif
    return GSN_OK;
}
template <int W, int T>
static int launch(CountArgs &a, int n_items, size_t lds, hipStream_t stream) {
    // TAIL: the instantiation with the closed forms / the tight loop of the last two levels (count_core.h) -- graphs above 64 vertices, or a
    // plan that ends in a closed form; the molecule workloads with cycle / clique patterns run the instantiation without them
    if constexpr (W >= 2) {            // (always with the tails: no second instantiation to compile)
        return a.off_ain >= 0 ? launch_d<W, T, true, true>(a, n_items, lds, stream) : launch_d<W, T, false, true>(a, n_items, lds, stream);
    } else {
        const bool tail = a.any_tail != 0 || a.tail_loop != 0;
        if (a.off_ain >= 0) return tail ? launch_d<W, T, true, true>(a, n_items, lds, stream) : launch_d<W, T, true, false>(a, n_items, lds, stream);
        if constexpr (T == 64) {
#ifdef COUNT_PROF
            static const bool mol_on = false;           // (the phase profile is read from the generic instantiation)
#else
            static const bool mol_on = [] { const char *d = getenv("GSN_COUNT_MOL"); return !d || atoi(d) != 0; }();
#endif
            const bool mol = mol_on && !tail && a.mode == GSN_MODE_EDGE && a.sym && a.n_cols == 4 && a.pair && a.split == 1 && a.enc_out &&
                             a.enc_stage && !a.graph_ids;
            if (getenv("GSN_CHAIN_TRACE"))
                fprintf(stderr, "gsn count: molecule instantiation %d (tail %d mode %d sym %d cols %d pair %d split %d stage %d out %d enc %d enc_stage %d ids %d)\n", (int)mol, (int)tail,
                        a.mode, a.sym, a.n_cols, a.pair, a.split, a.stage_out, a.out != nullptr, a.enc_out != nullptr, a.enc_stage, a.graph_ids != nullptr);
            if (mol) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&count_kernel_mol), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return set_error(GSN_E_HIP, "hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
                hipLaunchKernelGGL(count_kernel_mol, dim3((unsigned)n_items), dim3(64), lds, stream, a);
                e = hipGetLastError();
                if (e != hipSuccess) return set_error(GSN_E_HIP, "count_kernel_mol launch: %s", hipGetErrorString(e));
                return GSN_OK;
            }
        }
        return tail ? launch_d<W, T, false, true>(a, n_items, lds, stream) : launch_d<W, T, false, false>(a, n_items, lds, stream);
    }
}

}  // namespace gsn

using namespace gsn;

static int count_launch(const uint32_t *plan_host, const uint32_t *plan_dev, int64_t plan_words, int64_t n_graphs,
                        const int64_t *node_ptr, const int64_t *edge_ptr, const int64_t *edge_index,
                        int64_t edge_row_stride, int ids_are_global, const int32_t *graph_ids, int64_t n_items,
                        int64_t max_nodes, int64_t max_edges, int64_t *out, int32_t *status, const int32_t *n_classes,
                        int enc_clamp, float *enc_out, void *stream, uint16_t *enc16 = nullptr, int64_t enc16_stride = 0, int64_t enc16_col0 = 0, int enc_no32 = 0,
                        const gsn_count_side *side = nullptr) {
    if (!plan_host || !plan_dev || plan_words < PLAN_HEADER_WORDS || plan_host[0] != PLAN_MAGIC)
        return set_error(GSN_E_INVALID, "gsn_count_hip: not a plan table (build it with gsn_count_plan_build)");
    if (!node_ptr || !edge_ptr || (!out && !enc_out) || !status) return set_error(GSN_E_INVALID, "gsn_count_hip: null pointer argument");
    if (enc_out && !n_classes) return set_error(GSN_E_INVALID, "gsn_count_encode_hip: n_classes is null");
    if (!graph_ids) n_items = n_graphs;
    if (n_items <= 0) return GSN_OK;
    if (max_nodes > 768)
        return set_error(GSN_E_UNSUPPORTED, "graphs with more than 768 vertices (%lld) are outside this build", (long long)max_nodes);
    if (max_nodes < 1) max_nodes = 1;
    if (max_edges < 0) max_edges = 0;

    CountArgs a{};
    a.plan = plan_dev; a.plan_words = (int)plan_words;
    a.mode = (int)plan_host[1]; a.n_plans = (int)plan_host[3]; a.n_cols = (int)plan_host[4]; a.kmax = (int)plan_host[5];
    a.plans_off = (int)plan_host[7];
    a.sym = (a.mode == GSN_MODE_EDGE && (plan_host[6] & 1u) == 0) ? 1 : 0;
    const bool directed = (plan_host[6] & 2u) != 0;
    a.stride = plan_stride(plan_host[6]);
    {
        static const int pull_batch = [] { const char *d = getenv("GSN_PULL_BATCH"); const int v = d ? atoi(d) : 8; return v < 1 ? 1 : (v > 64 ? 64 : v); }();
        a.pull_batch = pull_batch;
    }
    if (directed && a.mode != GSN_MODE_VERTEX) return set_error(GSN_E_UNSUPPORTED, "gsn_count_hip: directed plans are vertex-mode plans");
    a.node_ptr = node_ptr; a.edge_ptr = edge_ptr;
    a.src = edge_index; a.dst = edge_index ? edge_index + edge_row_stride : nullptr;
    a.ids_are_global = ids_are_global; a.graph_ids = graph_ids;
    a.out = out; a.status = status;
    a.enc_out = enc_out; a.enc_width = 0; a.enc_clamp = enc_clamp;
    int64_t enc_width = 0;
    bool enc_bytes = true;                              // every class index fits a byte (0xff = none)
    if (enc_out) {
        if (a.n_cols > GSN_ENC_MAX_COLS)
            return set_error(GSN_E_UNSUPPORTED, "gsn_count_encode_hip: %d output columns; the fused encoding handles <= %d", a.n_cols, GSN_ENC_MAX_COLS);
        for (int c = 0; c < a.n_cols; ++c) {
            if (n_classes[c] < 1 || n_classes[c] > 65535) return set_error(GSN_E_INVALID, "gsn_count_encode_hip: n_classes[%d] = %d", c, n_classes[c]);
            a.enc_n[c] = (unsigned short)n_classes[c];
            enc_width += n_classes[c];
            enc_bytes = enc_bytes && n_classes[c] <= 255;
        }
        a.enc_width = (int)enc_width;
    }
    if (max_edges > 0 && !edge_index) return set_error(GSN_E_INVALID, "gsn_count_hip: edge_index is null");
    // side outputs (gsn_count_encode_pack16_side_hip): validated here; the grid gets its side workgroups below
    if (side) {
        if (graph_ids) return set_error(GSN_E_UNSUPPORTED, "gsn_count_encode_pack16_side_hip: a graph list (the side outputs cover every graph of the batch)");
        if (side->seg_ptr) {
            if ((max_edges > 0 && !side->perm) || side->n_nodes < 0 || side->n_edges < 0 || side->n_nodes >= ((int64_t)1 << 31) - 1 || side->n_edges >= (int64_t)1 << 31)
                return set_error(GSN_E_INVALID, "gsn_count_encode_pack16_side_hip: CSR outputs / totals");
            a.side_mask |= 1;
            a.side.csr_seg = side->seg_ptr; a.side.csr_perm = side->perm; a.side.csr_tgt = side->sorted_target; a.side.csr_oth = side->sorted_other;
        }
        a.side.csr_row = side->csr_row ? 1 : 0; a.side.tot_nodes = side->n_nodes; a.side.tot_edges = side->n_edges;
        if (side->node_codes) {
            if (!side->node_pack || (reinterpret_cast<uintptr_t>(side->node_pack) & 15) || side->node_code_cols < 1 || side->node_code_cols > 4)
                return set_error(GSN_E_INVALID, "gsn_count_encode_pack16_side_hip: node pack (16-byte aligned) and 1..4 node code columns");
            int o = 0;
            for (int c = 0; c < side->node_code_cols; ++c) {
                if (side->node_n_classes[c] < 1) return set_error(GSN_E_INVALID, "gsn_count_encode_pack16_side_hip: node_n_classes[%d] < 1", c);
                a.side.ncode_ptr_w[c >> 2] |= (unsigned)o << (8 * (c & 3)); o += side->node_n_classes[c];
                if (o > 28) return set_error(GSN_E_INVALID, "gsn_count_encode_pack16_side_hip: more than 28 encoded node columns");
            }
            a.side.ncode_ptr_w[side->node_code_cols >> 2] |= (unsigned)o << (8 * (side->node_code_cols & 3));
            a.side_mask |= 2;
            a.side.ncode = side->node_codes; a.side.npack = side->node_pack; a.side.ncode_cols = side->node_code_cols; a.side.ncode_clamp = side->node_clamp ? 1 : 0;
        }
        if (side->edge_codes) {
            if (!enc16 || enc16_stride != 16 || (reinterpret_cast<uintptr_t>(enc16) & 15) || side->edge_code_cols < 1 || side->edge_code_cols > 4)
                return set_error(GSN_E_INVALID, "gsn_count_encode_pack16_side_hip: edge codes need a 16-column, 16-byte-aligned edge pack and 1..4 code columns");
            if (side->edge_col0 < 0 || (side->edge_col0 & 3) || side->edge_col0 < enc16_col0 + enc_width)
                return set_error(GSN_E_INVALID, "gsn_count_encode_pack16_side_hip: edge_col0 %d must be a multiple of 4 behind the identifier columns (%lld .. %lld)",
                                 side->edge_col0, (long long)enc16_col0, (long long)(enc16_col0 + enc_width));
            int o = side->edge_col0;
            for (int c = 0; c < side->edge_code_cols; ++c) {
                if (side->edge_n_classes[c] < 1) return set_error(GSN_E_INVALID, "gsn_count_encode_pack16_side_hip: edge_n_classes[%d] < 1", c);
                a.side.ecode_ptr_w[c >> 2] |= (unsigned)o << (8 * (c & 3)); o += side->edge_n_classes[c];
            }
            a.side.ecode_ptr_w[side->edge_code_cols >> 2] |= (unsigned)o << (8 * (side->edge_code_cols & 3));
            if (o > 16 || o - side->edge_col0 > 8)
                return set_error(GSN_E_INVALID, "gsn_count_encode_pack16_side_hip: %d edge code classes at column %d (<= 8, inside the 16 columns)", o - side->edge_col0, side->edge_col0);
            a.side_mask |= 4;
            a.side.ecode = side->edge_codes; a.side.epack = enc16; a.side.ecode_cols = side->edge_code_cols; a.side.ecode_clamp = side->edge_clamp ? 1 : 0;
            a.side.ecode_col0 = side->edge_col0;
        }
        a.side.code_status = side->code_status;
    }

    const int W = max_nodes <= 64 ? 1 : (max_nodes <= 128 ? 2 : (max_nodes <= 256 ? 4 : (max_nodes <= 512 ? 8 : 12)));
    const bool edge_mode = a.mode == GSN_MODE_EDGE;
    const int64_t rows_cap = edge_mode ? max_edges : max_nodes;
    // large graphs (W > 4: 16-bit vertex ids, adjacency up to 72 KiB) keep one wave per workgroup so that the candidate
    // stack stays small; their parallelism comes from `split` workgroups per graph
    const int T = (rows_cap * a.n_cols <= 512 || W > 4) ? 64 : 256;
    a.n_decl = (int)max_nodes; a.e_decl = (int)max_edges;
    a.n_graphs = (int)n_graphs;
    // Two consecutive graphs per workgroup, as one graph (count_body): whole batches of small graphs whose plans hold connected patterns
    // only (every enumerated level of every plan has an adjacency constraint); GSN_COUNT_PAIR=0 switches it off.
    const int64_t tasks_cap1 = rows_cap * a.n_cols;
    bool pair = !graph_ids && n_graphs >= 2 && W == 1 && T == 64 && !(n_items < 2048 && tasks_cap1 >= 1024);
    if (pair) { const char *d = getenv("GSN_COUNT_PAIR"); if (d && atoi(d) == 0) pair = false; }
    for (int p = 0; pair && p < a.n_plans; ++p) {
        const uint32_t *w = plan_host + a.plans_off + (int64_t)p * a.stride;
        const int k = (int)(w[0] & 0xffu), nfix = (int)((w[0] >> 8) & 0xffu);
        for (int l = nfix; l < k; ++l) {
            const uint32_t adj = (w[2 + l] & 0xffu) | (directed ? (w[PLAN_STRIDE_WORDS + l] & 0xffu) : 0u);
            if (adj == 0) pair = false;
        }
    }
    a.pair = pair ? 1 : 0;
    if (pair) {
        max_nodes = 2 * max_nodes < 64 ? 2 * max_nodes : 64;
        max_edges = 2 * max_edges;
    }
    const int64_t rows_cap_u = edge_mode ? max_edges : max_nodes;      // (of what a workgroup holds: a graph, or the union of two)
    a.n_cap = (int)max_nodes; a.e_cap = (int)max_edges;

    int o = align_up((int)max_nodes * W * 8, 16);
    a.off_ain = -1;
    if (directed) { a.off_ain = o; o += align_up((int)max_nodes * W * 8, 16); }
    a.off_valid = o; o += align_up(W * 8, 16);
    // candidate stack: one frame per enumerated level below the last (levels n_fixed .. k - 2; .. k - 3 for W >= 2); the frames of the root levels are
    // never touched, so the array starts at the smallest n_fixed of the plans (stack_skip frames in front of it do not exist)
    int nfix_min = 255;
    for (int p = 0; p < a.n_plans; ++p) { const int nf = (int)((plan_host[a.plans_off + (int64_t)p * a.stride] >> 8) & 0xffu); nfix_min = nf < nfix_min ? nf : nfix_min; }
    if (nfix_min > a.kmax - 1 || a.n_plans == 0) nfix_min = 0;
    a.stack_skip = nfix_min;
    // (W >= 2: levels k - 2 and k - 1 never get a frame -- count_core.h counts them in closed form or in tail_loop)
    const int last_frame = W >= 2 ? a.kmax - 2 : a.kmax - 1;
    const int depth = last_frame - nfix_min > 1 ? last_frame - nfix_min : 1;
    a.off_stack = o; o += depth * W * T * 8;
    a.off_plan = o; o += align_up((int)plan_words * 4, 16);
    const int vid_bytes = W > 4 ? 2 : 1;
    a.off_eu = o; o += edge_mode ? align_up((int)max_edges * vid_bytes, 16) : 0;
    a.off_ev = o; o += edge_mode ? align_up((int)max_edges * vid_bytes, 16) : 0;
    a.off_rowstart = o; o += edge_mode ? align_up(((int)max_nodes + 1) * (W == 1 ? 2 : 4), 16) : 0;
    a.off_last = o; o += edge_mode ? align_up((int)max_edges * 4, 16) : 0;
    if (edge_mode && max_edges >= 65535) return set_error(GSN_E_UNSUPPORTED, "gsn_count_hip: %lld columns per workgroup (16-bit column tables; LDS ends far earlier)", (long long)max_edges);
    a.off_prim = o; o += edge_mode ? align_up((int)max_edges * 2, 16) : 0;
    a.off_revof = o; o += edge_mode ? align_up((int)max_edges * 2, 16) : 0;
    a.off_misc = o; o += 32;
    a.off_enc = o; o += enc_out ? align_up(2 * a.n_cols * 4, 16) : 0;
    a.off_core = o; o += align_up((CORE_MAX + 1) * W * 8, 16);
    a.core_mask = 0;
    for (int p = 0; p < a.n_plans; ++p) a.core_mask |= 1 << plan_core(plan_host + a.plans_off + p * a.stride);
    // pruning tables only for graphs of <= 64 vertices (molecules): on larger, denser targets the balls are (nearly)
    // everything and the extra AND per step costs more than it saves (measured: ER G(128,1000) +8 %)
    a.off_ball = -1;
    if (W == 1 && a.kmax >= 4 && !directed) { a.off_ball = o; o += align_up(2 * (int)max_nodes * W * 8, 16); }
    // degree planes only when a plan ends in a chain (patterns.cpp: plan_tail_mode == 3): cycles, cliques -- the molecule workloads -- do not
    a.off_degp = -1; a.degp_mask = 0; a.any_tail = 0;
    for (int p = 0; p < a.n_plans; ++p) {
        const uint32_t *w = plan_host + a.plans_off + (int64_t)p * a.stride;
        if (plan_tail(w)) a.any_tail = 1;
        if (plan_tail(w) == 3) a.degp_mask |= 1 << plan_core(w);
    }
    if (a.degp_mask) { a.off_degp = o; o += align_up((CORE_MAX + 1) * DEG_PLANES * W * 8, 16); }
    // one-word graphs: the tight loop over the images of level k - 2 pays where those sets are long -- dense graphs (average degree >= 8 by
    // the caller's capacities: clique-rich ego networks 1.81 -> 0.96 ms per 1000, 12-regular n = 25 +5 %), not molecules (ZINC 0.073 ->
    // 0.082 ms with it).  GSN_COUNT_TAIL_LOOP=0 / 1 forces it off / on.
    a.tail_loop = 0;
    if (W == 1) {
        static const int forced = [] { const char *e = getenv("GSN_COUNT_TAIL_LOOP"); return e ? atoi(e) : -1; }();
        a.tail_loop = forced >= 0 ? (forced != 0) : ((int64_t)a.e_decl >= 8 * (int64_t)a.n_decl);     // (the caller's capacities, before pairing)
    }
    a.off_out = o;
    const int64_t stage_bytes = align_up((int)(rows_cap_u * a.n_cols * 2 < ((int64_t)1 << 28) ? rows_cap_u * a.n_cols * 2 : ((int64_t)1 << 28)), 16);      // (16-bit staged counts)
    // Stage the output rows in LDS (coalesced final write) only while that keeps the workgroup small: the search is
    // latency-bound on dependent LDS reads, so heavy graphs want as many co-resident workgroups per CU as possible
    // (>= 8 waves per SIMD) and write their cells straight to HBM instead.
    const int64_t lds_budget = (T == 64 ? (pair ? 160 * 1024 / 16 : 160 * 1024 / 32) : 160 * 1024 / 8);
    a.stage_out = (out && o + stage_bytes <= lds_budget) ? 1 : 0;
    // few heavy graphs: several workgroups per graph so that every CU gets >= 8 of them
    a.split = 1;
    const int64_t tasks_cap = rows_cap * a.n_cols;
    static const int64_t split_target = [] { const char *e = getenv("GSN_COUNT_SPLIT_TARGET"); const int64_t v = e ? atoll(e) : 0; return v > 0 ? v : (int64_t)2048; }();
    if (!pair && n_items < split_target && tasks_cap >= 1024) {
        int64_t sp = (split_target + n_items - 1) / n_items;
        if (sp > 32) sp = 32;
        if (sp > tasks_cap / 256) sp = tasks_cap / 256;
        if (sp > 1) { a.split = (int)sp; a.stage_out = 0; }
    }
    if (a.stage_out) o += (int)stage_bytes;
    // class indices of the encoded rows: staged whenever one workgroup owns the whole graph and the indices fit a byte; else
    // every cell writes its floats itself
    a.enc_stage = 0; a.off_encst = o; a.enc_from_counts = 0;
    a.enc16 = enc16; a.enc16_stride = (int)enc16_stride; a.enc16_col0 = (int)enc16_col0;
    a.enc_no32 = (enc16 && enc_no32) ? 1 : 0;
    if (enc_out && a.split == 1 && enc_bytes && rows_cap_u * enc_width < (int64_t)1 << 24 && o + rows_cap_u * a.n_cols <= 150 * 1024) {
        a.enc_stage = 1;
        static const bool bytes_forced = getenv("GSN_COUNT_ENC_BYTES") != nullptr;      // (A/B: keep the byte array beside staged counts)
        if (a.stage_out && !bytes_forced) a.enc_from_counts = 1;      // the staged 16-bit counts hold what the class indices need: no second array
        else o += align_up((int)(rows_cap_u * a.n_cols), 16);         // (16-byte aligned: with four columns a row's indices are read as one word)
    }
    if (a.side_mask) {
        if (a.split != 1)
            return set_error(GSN_E_UNSUPPORTED, "gsn_count_encode_pack16_side_hip: this launch splits a graph over %d workgroups (few heavy graphs): the side workgroups ride a "
                                                "one-workgroup-per-item grid; use gsn_csr_build_graphs_hip / gsn_one_hot_pack16_hip", a.split);
        // a side workgroup sorts a graph of the declared sizes in the launch's LDS (several at a time where they fit)
        const int64_t need = ((int64_t)a.n_decl + 1) * 8 + (int64_t)a.n_decl * 4 + (int64_t)a.e_decl * 7 + 16;
        if (a.n_decl >= 65535 || a.e_decl >= 65535 || need > 160 * 1024)
            return set_error(GSN_E_UNSUPPORTED, "gsn_count_encode_pack16_side_hip: graphs of %d vertices / %d columns do not sort in LDS (%lld B)", a.n_decl, a.e_decl, (long long)need);
        if (need > o) o = align_up((int)need, 16);
    }
    if (o > 160 * 1024) return set_error(GSN_E_UNSUPPORTED, "graph too large for LDS (%d B needed)", o);
    if (enc16 && !a.enc_stage)
        return set_error(GSN_E_UNSUPPORTED, "gsn_count_encode_pack16_hip: the fp16 rows are written from the staged class indices (one workgroup per graph, "
                                            "n_classes <= 255); pack the fp32 rows with gsn_pack16_rows_hip instead");

    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // per-graph status words start at OK; workgroups raise them with atomicMax.  One workgroup per graph (or pair) and every graph taken:
    // the workgroup zeroes its own words -- the memset was a 6 us launch (+ the gap behind it) in front of every counting launch
    a.zero_status = (!graph_ids && a.split == 1) ? 1 : 0;
    if (!a.zero_status) {
        hipError_t e = graph_ids ? hipSuccess : hipMemsetAsync(status, 0, sizeof(int32_t) * (size_t)n_graphs, st);
        if (graph_ids) hipLaunchKernelGGL(status_zero_kernel, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, st, graph_ids, (int)n_items, status);
        if (e != hipSuccess) return set_error(GSN_E_HIP, "hipMemsetAsync(status): %s", hipGetErrorString(e));
    }
    int items = pair ? (int)((n_graphs + 1) / 2) : (int)n_items * a.split;
    a.n_items = items; a.lds_bytes = o;
    if (a.side_mask) items = (items + SIDE_EVERY - 1) / SIDE_EVERY * (SIDE_EVERY + 1);      // every (SIDE_EVERY + 1)-th workgroup is a side workgroup
    if (getenv("GSN_CHAIN_TRACE")) fprintf(stderr, "gsn count: count_kernel<%d,%d> workgroups %d pair %d split %d lds %d\n", W, T, items, a.pair, a.split, o);
    if (W == 1 && T == 64) return launch<1, 64>(a, items, (size_t)o, st);
    if (W == 1) return launch<1, 256>(a, items, (size_t)o, st);
    if (W == 2 && T == 64) return launch<2, 64>(a, items, (size_t)o, st);
    if (W == 2) return launch<2, 256>(a, items, (size_t)o, st);
    if (W == 4 && T == 64) return launch<4, 64>(a, items, (size_t)o, st);
    if (W == 4) return launch<4, 256>(a, items, (size_t)o, st);
    if (W == 8) return launch<8, 64>(a, items, (size_t)o, st);
    return launch<12, 64>(a, items, (size_t)o, st);
}

extern "C" int gsn_count_hip(const uint32_t *plan_host, const uint32_t *plan_dev, int64_t plan_words, int64_t n_graphs,
                             const int64_t *node_ptr, const int64_t *edge_ptr, const int64_t *edge_index,
                             int64_t edge_row_stride, int ids_are_global, const int32_t *graph_ids, int64_t n_items,
                             int64_t max_nodes, int64_t max_edges, int64_t *out, int32_t *status, void *stream) {
    if (!out) return set_error(GSN_E_INVALID, "gsn_count_hip: null pointer argument");
    return count_launch(plan_host, plan_dev, plan_words, n_graphs, node_ptr, edge_ptr, edge_index, edge_row_stride, ids_are_global,
                        graph_ids, n_items, max_nodes, max_edges, out, status, nullptr, 0, nullptr, stream);
}

extern "C" int gsn_count_encode_hip(const uint32_t *plan_host, const uint32_t *plan_dev, int64_t plan_words, int64_t n_graphs,
                                    const int64_t *node_ptr, const int64_t *edge_ptr, const int64_t *edge_index,
                                    int64_t edge_row_stride, int ids_are_global, const int32_t *graph_ids, int64_t n_items,
                                    int64_t max_nodes, int64_t max_edges, int64_t *out, int32_t *status,
                                    const int32_t *n_classes, int clamp, float *enc_out, void *stream) {
    if (!enc_out) return set_error(GSN_E_INVALID, "gsn_count_encode_hip: enc_out is null");
    return count_launch(plan_host, plan_dev, plan_words, n_graphs, node_ptr, edge_ptr, edge_index, edge_row_stride, ids_are_global,
                        graph_ids, n_items, max_nodes, max_edges, out, status, n_classes, clamp, enc_out, stream);
}

extern "C" int gsn_count_encode_pack16_hip(const uint32_t *plan_host, const uint32_t *plan_dev, int64_t plan_words, int64_t n_graphs,
                                           const int64_t *node_ptr, const int64_t *edge_ptr, const int64_t *edge_index,
                                           int64_t edge_row_stride, int ids_are_global, const int32_t *graph_ids, int64_t n_items,
                                           int64_t max_nodes, int64_t max_edges, int64_t *out, int32_t *status,
                                           const int32_t *n_classes, int clamp, float *enc_out, uint16_t *pack, int64_t pack_stride,
                                           int64_t pack_col0, void *stream) {
    if (!pack) return set_error(GSN_E_INVALID, "gsn_count_encode_pack16_hip: pack is null");
    // enc_out == NULL (r05): the pack columns are the only form of the encoded rows -- the fp32 one-hot rows (48 bytes per row of four 3-class
    // columns) are not written at all; the kernel takes the same path with a placeholder address it never stores to
    const int no32 = enc_out ? 0 : 1;
    if (no32) enc_out = reinterpret_cast<float *>(pack);
    int64_t w = 0;
    if (n_classes && plan_host && plan_words >= PLAN_HEADER_WORDS) for (uint32_t c = 0; c < plan_host[4] && c < GSN_ENC_MAX_COLS; ++c) w += n_classes[c];
    if (pack_stride <= 0 || pack_stride > 0x7fffffff || pack_col0 < 0 || pack_col0 + w > pack_stride)
        return set_error(GSN_E_INVALID, "gsn_count_encode_pack16_hip: columns %lld .. %lld outside a pack row of %lld", (long long)pack_col0, (long long)(pack_col0 + w), (long long)pack_stride);
    return count_launch(plan_host, plan_dev, plan_words, n_graphs, node_ptr, edge_ptr, edge_index, edge_row_stride, ids_are_global,
                        graph_ids, n_items, max_nodes, max_edges, out, status, n_classes, clamp, enc_out, stream, pack, pack_stride, pack_col0, no32);
}

extern "C" int gsn_count_encode_pack16_side_hip(const uint32_t *plan_host, const uint32_t *plan_dev, int64_t plan_words, int64_t n_graphs,
                                                const int64_t *node_ptr, const int64_t *edge_ptr, const int64_t *edge_index,
                                                int64_t edge_row_stride, int ids_are_global, int64_t max_nodes, int64_t max_edges, int64_t *out,
                                                int32_t *status, const int32_t *n_classes, int clamp, uint16_t *pack, int64_t pack_stride,
                                                int64_t pack_col0, const gsn_count_side *side, void *stream) {
    if (!side) return set_error(GSN_E_INVALID, "gsn_count_encode_pack16_side_hip: side is null");
    if (!out) return set_error(GSN_E_INVALID, "gsn_count_encode_pack16_side_hip: the int64 counts are always written (out is null)");
    if (!pack)          // (no identifier pack: plain counts + the CSR / node pack; edge codes need the pack and are refused in count_launch)
        return count_launch(plan_host, plan_dev, plan_words, n_graphs, node_ptr, edge_ptr, edge_index, edge_row_stride, ids_are_global, nullptr, n_graphs,
                            max_nodes, max_edges, out, status, nullptr, 0, nullptr, stream, nullptr, 0, 0, 0, side);
    int64_t w = 0;
    if (n_classes && plan_host && plan_words >= PLAN_HEADER_WORDS) for (uint32_t c = 0; c < plan_host[4] && c < GSN_ENC_MAX_COLS; ++c) w += n_classes[c];
    if (pack_stride <= 0 || pack_stride > 0x7fffffff || pack_col0 < 0 || pack_col0 + w > pack_stride)
        return set_error(GSN_E_INVALID, "gsn_count_encode_pack16_side_hip: columns %lld .. %lld outside a pack row of %lld", (long long)pack_col0, (long long)(pack_col0 + w), (long long)pack_stride);
    return count_launch(plan_host, plan_dev, plan_words, n_graphs, node_ptr, edge_ptr, edge_index, edge_row_stride, ids_are_global, nullptr, n_graphs,
                        max_nodes, max_edges, out, status, n_classes, clamp, reinterpret_cast<float *>(pack), stream, pack, pack_stride, pack_col0, 1, side);
}
