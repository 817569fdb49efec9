// HP-2 propagate stage for gfx950: target-sorted CSR build + fused gather -> message -> segmented sum (fwd and adjoint).
//
// Replaces  torch.sparse.FloatTensor(edge_index, msgs, [N,N,d]) + torch.sparse.sum(.., aggr_dim).to_dense()
// (GSN_sparse.py:140-143, GSN_edge_sparse.py:136-139 and the MPNN / ogb twins) and the index_select gathers in front of
// it (GSN_*.py:125-136).  HBM-bound, zero dense flops: every input row is read once through L2, the output row is
// written once, no [E,d] message tensor and no COO sort.  Rows are summed in edge order inside a segment, so the fp32
// result is deterministic (no float atomics).
//
// Work mapping: one "row group" of LPR lanes per target vertex (LPR = lanes per row, a power of two chosen so that
// LPR * VEC >= d_out where possible), 64/LPR vertices per wave, 4 waves per workgroup; lanes stride over the feature
// dimension with float4 (VEC=4) accesses when every block width is a multiple of 4, scalar otherwise.
#include <hip/hip_runtime.h>
#include <cstdio>

#include "gsn_internal.h"

namespace gsn {

// ----------------------------------------------------------------------------------------------------------------
// CSR build: stable counting sort of edge ids by aggregation target
// ----------------------------------------------------------------------------------------------------------------
__global__ void csr_zero(int32_t *cnt, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cnt[i] = 0;
}

__global__ void csr_hist(const int64_t *__restrict__ index, int64_t n_edges, int32_t *cnt) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_edges; e += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(&cnt[index[e]], 1);
}

// One-wave workgroups throughout the CSR build: it usually runs on a second stream next to the counting kernel, whose
// one-wave workgroups refill every freed wave slot at once -- a 256-thread workgroup needs four free slots on one CU at the
// same moment and starves until the counting kernel drains (measured: csr_hist 254 us instead of 30).
constexpr int CSR_T = 64;
constexpr int SCAN_T = 64;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_T * SCAN_ITEMS;

// per-tile sums
__global__ __launch_bounds__(SCAN_T) void scan_tile_sums(const int32_t *__restrict__ in, int64_t n, int32_t *tile_sums) {
    __shared__ int32_t red[SCAN_T / 64];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
    int32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int64_t j = base + threadIdx.x * SCAN_ITEMS + i;
        if (j < n) s += in[j];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t t = 0;
        for (int i = 0; i < SCAN_T / 64; ++i) t += red[i];
        tile_sums[blockIdx.x] = t;
    }
}

// exclusive scan of the tile sums by one wave (n_tiles is small: n / 1024)
__global__ __launch_bounds__(64) void scan_tile_offsets(int32_t *tile_sums, int64_t n_tiles) {
    const int lane = threadIdx.x;
    int32_t carry = 0;
    for (int64_t base = 0; base < n_tiles; base += 64) {
        const int64_t j = base + lane;
        const int32_t v = j < n_tiles ? tile_sums[j] : 0;
        int32_t incl = v;
        for (int o = 1; o < 64; o <<= 1) {
            const int32_t t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (j < n_tiles) tile_sums[j] = carry + incl - v;
        carry += __shfl(incl, 63);
    }
}

// out[j] = tile_offset + exclusive scan inside the tile;  also copies the result into `cursor`
__global__ __launch_bounds__(SCAN_T) void scan_apply(const int32_t *__restrict__ in, int64_t n, const int32_t *__restrict__ tile_off,
                                                    int32_t *out, int32_t *cursor) {
    __shared__ int32_t wsum[SCAN_T / 64];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    int32_t v[SCAN_ITEMS];
    int32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = (base + i < n) ? in[base + i] : 0;
        s += v[i];
    }
    // wave inclusive scan of s
    int32_t incl = s;
    const int lane = threadIdx.x & 63;
    for (int o = 1; o < 64; o <<= 1) {
        const int32_t t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    int32_t woff = 0;
    for (int i = 0; i < (int)(threadIdx.x >> 6); ++i) woff += wsum[i];
    int32_t run = tile_off[blockIdx.x] + woff + incl - s;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) {
            out[base + i] = run;
            cursor[base + i] = run;
        }
        run += v[i];
    }
}

__global__ void csr_fill(const int64_t *__restrict__ index, int64_t n_edges, int32_t *cursor, int32_t *perm) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_edges; e += (int64_t)gridDim.x * blockDim.x) {
        const int32_t pos = atomicAdd(&cursor[index[e]], 1);
        perm[pos] = (int32_t)e;
    }
}

// restore original edge order inside every segment (atomics above place them in arbitrary order)
__global__ void csr_sort_segments(const int32_t *__restrict__ seg_ptr, int64_t n_nodes, int32_t *perm, int32_t *sorted_target,
                                  const int64_t *__restrict__ other, int32_t *sorted_other) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_nodes) return;
    const int32_t lo = seg_ptr[t], hi = seg_ptr[t + 1];
    for (int32_t i = lo + 1; i < hi; ++i) {
        const int32_t x = perm[i];
        int32_t j = i - 1;
        while (j >= lo && perm[j] > x) { perm[j + 1] = perm[j]; --j; }
        perm[j + 1] = x;
    }
    if (sorted_target)
        for (int32_t i = lo; i < hi; ++i) sorted_target[i] = (int32_t)t;
    if (sorted_other)
        for (int32_t i = lo; i < hi; ++i) sorted_other[i] = (int32_t)other[perm[i]];
}

// Small batches (the reference's 32..128 graphs per step): the whole build in ONE workgroup -- histogram, scan, fill and
// the per-segment order restore in LDS -- instead of seven launches whose enqueue cost exceeds their run time.
constexpr int CSR_SMALL_NODES = 12287;   // n_nodes + 1 counters + the same number of cursors in LDS (2 x 48 KiB)
constexpr int CSR_SMALL_EDGES = 1 << 15;

__global__ __launch_bounds__(1024) void csr_small_kernel(const int64_t *__restrict__ index, int64_t n_edges, int n_nodes,
                                                         int32_t *seg_ptr, int32_t *perm, int32_t *sorted_target,
                                                         const int64_t *__restrict__ other, int32_t *sorted_other) {
    extern __shared__ int32_t sm[];
    int32_t *cnt = sm;                    // [n_nodes + 1]
    int32_t *cur = sm + n_nodes + 1;      // [n_nodes + 1]
    __shared__ int32_t wsum[16];
    __shared__ int32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i <= n_nodes; i += 1024) cnt[i] = 0;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int64_t e = tid; e < n_edges; e += 1024) atomicAdd(&cnt[index[e]], 1);
    __syncthreads();
    for (int base = 0; base <= n_nodes; base += 1024) {       // exclusive scan, 1024 counters per pass
        const int i = base + tid;
        const int32_t v = i <= n_nodes ? cnt[i] : 0;
        int32_t x = v;
        for (int d = 1; d < 64; d <<= 1) {
            const int32_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int32_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        const int32_t carry = carry_s;
        if (i <= n_nodes) {
            const int32_t ex = carry + woff + x - v;
            cur[i] = ex;
            seg_ptr[i] = ex;
        }
        __syncthreads();
        if (tid == 1023) carry_s = carry + woff + x;
        __syncthreads();
    }
    for (int64_t e = tid; e < n_edges; e += 1024) {
        const int32_t pos = atomicAdd(&cur[index[e]], 1);
        perm[pos] = (int32_t)e;
    }
    __threadfence_block();
    __syncthreads();
    for (int t = tid; t < n_nodes; t += 1024) {               // cur[t] is now the segment end
        const int32_t hi = cur[t], lo = hi - cnt[t];
        for (int32_t i = lo + 1; i < hi; ++i) {
            const int32_t x = perm[i];
            int32_t j = i - 1;
            while (j >= lo && perm[j] > x) { perm[j + 1] = perm[j]; --j; }
            perm[j + 1] = x;
        }
        if (sorted_target)
            for (int32_t i = lo; i < hi; ++i) sorted_target[i] = (int32_t)t;
        if (sorted_other)
            for (int32_t i = lo; i < hi; ++i) sorted_other[i] = (int32_t)other[perm[i]];
    }
}

// Output rows the fused segmented-sum epilogue (chain.hip) reaches with atomics or not at all must start at zero:
// empty segments, and segments that straddle a GSN_SEG_RANGE_ROWS-row boundary of the target-sorted row space.
__global__ __launch_bounds__(256) void segsum_prepare_kernel(const int32_t *__restrict__ seg_ptr, const int32_t *__restrict__ row_target,
                                                             int64_t n_seg, int64_t n_rows, int n_out, float *out) {
    const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n_bound = n_rows / GSN_SEG_RANGE_ROWS;  // boundaries at rows R, 2R, ...
    int row = -1;
    if (item < n_seg) {
        if (seg_ptr[item] == seg_ptr[item + 1]) row = (int)item;
    } else if (item - n_seg < n_bound) {
        const int64_t r = (item - n_seg + 1) * GSN_SEG_RANGE_ROWS;
        if (r < n_rows && row_target[r - 1] == row_target[r]) row = row_target[r];
    }
    // the wave zeroes the rows its lanes found, one row at a time, all 64 lanes on the columns
    unsigned long long m = __ballot(row >= 0);
    const int lane = threadIdx.x & 63;
    while (m) {
        const int src = __ffsll(m) - 1;
        m &= m - 1;
        const int r = __shfl(row, src);
        for (int c = lane; c < n_out; c += 64) out[(int64_t)r * n_out + c] = 0.f;
    }
}

// ----------------------------------------------------------------------------------------------------------------
// forward: out[t] = sum over the segment of t of msg_e
// ----------------------------------------------------------------------------------------------------------------
struct PropArgs {
    int kind;
    int64_t n_nodes, n_edges;
    const int64_t *src;
    const int32_t *seg_ptr, *perm;
    const int32_t *sorted_src;   // src[perm[q]] per segment position, or null
    const float *a, *b, *c;
    int da, db, dc, d_out;
    int64_t ldb;                 // row stride of b in floats (= db unless b is a column slice of wider rows: gsn_segment_sum_rows_hip)
    int b_per_node;
    float *out;
    // r03: the layer's own term and the central encoders' column padding inside the same pass (GSN_sparse.py:157-163,
    // GSN_edge_sparse_ogb.py:103-106:  (1 + eps) * self + sum of messages;  utils_graph_learning.py:240-246: a zero column in front
    // of the neighbours' one-hot block).  CAT: the self blocks are concatenated, RELU_SUM: added; row_stride 0 = one row for every node.
    int pad_b, pad_c;
    int n_self;
    const float *self_data[3];
    int self_w[3];
    int64_t self_stride[3];
    const float *eps;
};

template <int VEC>
struct VecT;
template <>
struct VecT<1> { using type = float; };
template <>
struct VecT<4> { using type = float4; };

// rows that are read exactly once (the per-edge blocks of a concatenation) and the output rows are streamed past the caches: scatter-add of
// 128-wide messages 0.494 -> 0.464 ms, the output store alone -3 % on every shape (-DPROP_NT=0: plain accesses; profiles/r03_propagate_nt.txt)
#ifndef PROP_NT
#define PROP_NT 3        // bit 0: loads, bit 1: stores
#endif
__device__ __forceinline__ float vload_once(const float *p) { return (PROP_NT & 1) ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ float4 vload_once(const float4 *p) {
    if (!(PROP_NT & 1)) return *p;
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void vstore_once(float *p, float v) { if (PROP_NT & 2) __builtin_nontemporal_store(v, p); else *p = v; }
__device__ __forceinline__ void vstore_once(float4 *p, float4 v) {
    if (!(PROP_NT & 2)) { *p = v; return; }
    typedef float f4v __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(f4v{v.x, v.y, v.z, v.w}, reinterpret_cast<f4v *>(p));
}
__device__ __forceinline__ float vadd(float x, float y) { return x + y; }
__device__ __forceinline__ float4 vadd(float4 x, float4 y) { return make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w); }
__device__ __forceinline__ float vfma(float s, float x, float y) { return s * x + y; }     // (1 + eps) * self + sum: two roundings, as the reference's two ops
__device__ __forceinline__ float4 vfma(float s, float4 x, float4 y) { return make_float4(s * x.x + y.x, s * x.y + y.y, s * x.z + y.z, s * x.w + y.w); }
__device__ __forceinline__ float vrelu(float x) { return x > 0.f ? x : 0.f; }
__device__ __forceinline__ float4 vrelu(float4 x) { return make_float4(vrelu(x.x), vrelu(x.y), vrelu(x.z), vrelu(x.w)); }
__device__ __forceinline__ void vzero(float &x) { x = 0.f; }
__device__ __forceinline__ void vzero(float4 &x) { x = make_float4(0.f, 0.f, 0.f, 0.f); }

// MAXC: column chunks of width LPR*VEC each lane accumulates (d_out <= MAXC*LPR*VEC).  With `sorted_src` the source
// vertex comes from the segment-ordered copy the CSR build writes (one dependent load less than src[perm[q]]).
// (Taking the edges of a segment four at a time with all row loads in flight together was measured and is no faster:
// the kernel is limited by DRAM efficiency on 512-byte gathers, not by load latency.)
// EXT: the layer's own term and / or zero columns (gsn_propagate_self_fwd_hip); the plain kernel is compiled without either
// U4 (launches with less than one workgroup per CU -- the graph-level readouts and the reference's batch sizes, where nothing hides a
// wave's load latency): a segment's elements four at a time, indices and rows of the four in flight together, added in the same order
// (the result is bit-identical).  Per element the plain loop pays perm -> src -> row as one dependent chain: 34 us for the 26-row
// segments of a 32-graph readout, d = 300.
template <int VEC, int LPR, int MAXC, bool EXT, bool U4 = false>
__global__ __launch_bounds__(256) void propagate_fwd_kernel(PropArgs p) {
    using V = typename VecT<VEC>::type;
    constexpr int RPW = 64 / LPR;  // rows (targets) per wave
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR, li = lane % LPR;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    // the message element of edge e (source vertex s) in this lane's column chunk i
    auto msg = [&](int64_t e, int64_t s, int i) -> V {
        const int col = (i * LPR + li) * VEC;
        V m;
        if (p.kind == GSN_MSG_CAT) {
            // column layout: a | pad_b zeros | b | pad_c zeros | c      (pads only on the scalar path, VEC == 1)
            constexpr bool PADS = EXT && VEC == 1;
            int o = col - p.da;
            if (o < 0) m = *reinterpret_cast<const V *>(p.a + s * p.da + col);
            else if (PADS && o < p.pad_b) vzero(m);
            else if ((o -= (PADS ? p.pad_b : 0)) < p.db)
                m = p.b_per_node ? *reinterpret_cast<const V *>(p.b + s * p.ldb + o) : vload_once(reinterpret_cast<const V *>(p.b + e * p.ldb + o));
            else if (PADS && (o - p.db) < p.pad_c) vzero(m);
            else m = vload_once(reinterpret_cast<const V *>(p.c + e * p.dc + (o - p.db - (PADS ? p.pad_c : 0))));
        } else {
            vzero(m);
            if (p.a) m = vadd(m, *reinterpret_cast<const V *>(p.a + s * p.d_out + col));
            // (three d-wide streams per edge: streaming the two per-edge ones past the caches was measured 6 % slower here)
            if (p.b) m = vadd(m, *reinterpret_cast<const V *>(p.b + (p.b_per_node ? s : e) * p.d_out + col));
            if (p.c) m = vadd(m, *reinterpret_cast<const V *>(p.c + e * p.d_out + col));
            m = vrelu(m);
        }
        return m;
    };
    for (int64_t t0 = wave * RPW; t0 < p.n_nodes; t0 += n_waves * RPW) {
        const int64_t t = t0 + sub;
        if (t >= p.n_nodes) continue;
        V acc[MAXC];
#pragma unroll
        for (int i = 0; i < MAXC; ++i) vzero(acc[i]);
        const int32_t lo = p.seg_ptr[t], hi = p.seg_ptr[t + 1];
        int32_t q = lo;
        if (U4) {
            for (; q + 4 <= hi; q += 4) {
                int64_t e[4], sv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) e[u] = p.perm ? (int64_t)p.perm[q + u] : (int64_t)(q + u);
#pragma unroll
                for (int u = 0; u < 4; ++u) sv[u] = p.sorted_src ? (int64_t)p.sorted_src[q + u] : p.src[e[u]];
                V m[4][MAXC];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int i = 0; i < MAXC; ++i)
                        if ((i * LPR + li) * VEC < p.d_out) m[u][i] = msg(e[u], sv[u], i);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int i = 0; i < MAXC; ++i)
                        if ((i * LPR + li) * VEC < p.d_out) acc[i] = vadd(acc[i], m[u][i]);
            }
        }
        for (; q < hi; ++q) {
            const int64_t e = p.perm ? (int64_t)p.perm[q] : (int64_t)q;
            const int64_t s = p.sorted_src ? (int64_t)p.sorted_src[q] : p.src[e];
#pragma unroll
            for (int i = 0; i < MAXC; ++i)
                if ((i * LPR + li) * VEC < p.d_out) acc[i] = vadd(acc[i], msg(e, s, i));
        }
        if (EXT && p.n_self) {
            const float sc = 1.f + (p.eps ? *p.eps : 0.f);
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                const int col = (i * LPR + li) * VEC;
                if (col < p.d_out) {
                    V sv;
                    vzero(sv);
                    int o = col;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        if (k < p.n_self) {
                            if (p.kind == GSN_MSG_CAT) {
                                if (o >= 0 && o < p.self_w[k]) sv = *reinterpret_cast<const V *>(p.self_data[k] + t * p.self_stride[k] + o);
                                o -= p.self_w[k];
                            } else {
                                sv = vadd(sv, *reinterpret_cast<const V *>(p.self_data[k] + t * p.self_stride[k] + col));
                            }
                        }
                    }
                    acc[i] = vfma(sc, sv, acc[i]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int col = (i * LPR + li) * VEC;
            if (col < p.d_out) vstore_once(reinterpret_cast<V *>(p.out + t * p.d_out + col), acc[i]);
        }
    }
}

// The ogb message  relu(x_j + id_e + e_e)  (GSN_edge_sparse_ogb.py:103-106: three d-wide streams per edge, float4 columns) with the index
// chain taken out of the row loop: per target the kernel above walks seg_ptr -> perm / sorted_src -> rows as three dependent round trips and
// only the last one carries data.  Here a lane group keeps a three-deep pipeline over its targets (grid stride): the segment bounds of the
// target two steps ahead, the first four edges' indices of the next target and the rows of this one are requested back to back, and the rows
// of UNR edges are in flight together.  Same order of additions as propagate_fwd_kernel (bit-identical result).
template <int LPR, int MAXC, int UNR, bool NT, bool EXT, bool HASC = true>
__global__ __launch_bounds__(256) void relu_sum3_kernel(PropArgs p) {
    constexpr int RPW = 64 / LPR;
    constexpr int NPF = 4;                       // edges per target whose indices are fetched ahead (longer segments: plain tail loop)
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR, li = lane % LPR;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t step = n_waves * RPW, nn = p.n_nodes;
    const int q4 = p.d_out >> 2;
    const float4 *A = reinterpret_cast<const float4 *>(p.a), *B = reinterpret_cast<const float4 *>(p.b), *C = reinterpret_cast<const float4 *>(p.c);
    auto ldseg = [&](int64_t tt, int32_t &lo, int32_t &hi) {
        lo = hi = 0;
        if (tt < nn) { lo = p.seg_ptr[tt]; hi = p.seg_ptr[tt + 1]; }
    };
    auto ldidx = [&](int32_t lo, int32_t hi, int32_t *e, int32_t *s) {
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int32_t q = lo + u;
            e[u] = 0; s[u] = 0;
            if (q < hi) {
                e[u] = p.perm ? p.perm[q] : q;
                s[u] = p.sorted_src ? p.sorted_src[q] : (int32_t)p.src[e[u]];
            }
        }
    };
    auto row = [&](const float4 *base, int32_t r, int i) -> float4 { return base[(int64_t)r * q4 + (i * LPR + li)]; };
    auto row_once = [&](const float4 *base, int32_t r, int i) -> float4 {
        const float4 *ptr = base + (int64_t)r * q4 + (i * LPR + li);
        return NT ? vload_once(ptr) : *ptr;
    };
    int64_t t = wave * RPW + sub;
    int32_t lo0, hi0, lo1, hi1, e0[NPF], s0[NPF];
    ldseg(t, lo0, hi0);
    ldseg(t + step, lo1, hi1);
    ldidx(lo0, hi0, e0, s0);
    for (int64_t t0 = wave * RPW; t0 < nn; t0 += step, t += step) {
        int32_t lo2, hi2, e1[NPF], s1[NPF];
        ldseg(t + 2 * step, lo2, hi2);
        ldidx(lo1, hi1, e1, s1);
        float4 acc[MAXC];
#pragma unroll
        for (int i = 0; i < MAXC; ++i) vzero(acc[i]);
        const int deg = hi0 - lo0;
#pragma unroll
        for (int u0 = 0; u0 < NPF; u0 += UNR) {
            if (u0 < deg) {
                float4 ra[UNR][MAXC], rb[UNR][MAXC], rc[UNR][HASC ? MAXC : 1];
#pragma unroll
                for (int u = 0; u < UNR; ++u)
#pragma unroll
                    for (int i = 0; i < MAXC; ++i) {
                        vzero(ra[u][i]); vzero(rb[u][i]);
                        if (HASC) vzero(rc[u][HASC ? i : 0]);
                        if (u0 + u < deg && (i * LPR + li) < q4) {
                            ra[u][i] = row(A, s0[u0 + u], i);
                            rb[u][i] = row_once(B, e0[u0 + u], i);
                            if (HASC) rc[u][HASC ? i : 0] = row_once(C, e0[u0 + u], i);
                        }
                    }
#pragma unroll
                for (int u = 0; u < UNR; ++u)
                    if (u0 + u < deg) {
#pragma unroll
                        for (int i = 0; i < MAXC; ++i) {
                            float4 m = vadd(ra[u][i], rb[u][i]);
                            if (HASC) m = vadd(m, rc[u][HASC ? i : 0]);
                            acc[i] = vadd(acc[i], vrelu(m));
                        }
                    }
            }
        }
        for (int32_t q = lo0 + NPF; q < hi0; ++q) {
            const int32_t e = p.perm ? p.perm[q] : q;
            const int32_t s = p.sorted_src ? p.sorted_src[q] : (int32_t)p.src[e];
#pragma unroll
            for (int i = 0; i < MAXC; ++i)
                if ((i * LPR + li) < q4) {
                    float4 m = vadd(row(A, s, i), row_once(B, e, i));
                    if (HASC) m = vadd(m, row_once(C, e, i));
                    acc[i] = vadd(acc[i], vrelu(m));
                }
        }
        if (t < nn) {
            if (EXT && p.n_self) {
                const float sc = 1.f + (p.eps ? *p.eps : 0.f);
#pragma unroll
                for (int i = 0; i < MAXC; ++i) {
                    const int col = (i * LPR + li) * 4;
                    if (col < p.d_out) {
                        float4 sv;
                        vzero(sv);
#pragma unroll
                        for (int k = 0; k < 3; ++k)
                            if (k < p.n_self) sv = vadd(sv, *reinterpret_cast<const float4 *>(p.self_data[k] + t * p.self_stride[k] + col));
                        acc[i] = vfma(sc, sv, acc[i]);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                const int col = (i * LPR + li) * 4;
                if (col < p.d_out) vstore_once(reinterpret_cast<float4 *>(p.out + t * p.d_out + col), acc[i]);
            }
        }
        lo0 = lo1; hi0 = hi1; lo1 = lo2; hi1 = hi2;
#pragma unroll
        for (int u = 0; u < NPF; ++u) { e0[u] = e1[u]; s0[u] = s1[u]; }
    }
}

// The same pipeline for a concatenation of float4-aligned blocks without zero columns (the gin messages, the scatter-add of per-edge message
// rows, the readouts): a | b | c, b per edge or per source vertex.  Segments longer than the four prefetched edges (a readout's ~23 rows per
// graph) continue four rows at a time, their indices and rows in flight together, added in segment order (bit-identical to the plain loop).
template <int LPR, int MAXC, int UNR, bool EXT>
__global__ __launch_bounds__(256) void cat_pipe_kernel(PropArgs p) {
    constexpr int RPW = 64 / LPR;
    constexpr int NPF = 4;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR, li = lane % LPR;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t step = n_waves * RPW, nn = p.n_nodes;
    auto ldseg = [&](int64_t tt, int32_t &lo, int32_t &hi) {
        lo = hi = 0;
        if (tt < nn) { lo = p.seg_ptr[tt]; hi = p.seg_ptr[tt + 1]; }
    };
    auto ldidx = [&](int32_t lo, int32_t hi, int32_t *e, int32_t *s) {
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int32_t q = lo + u;
            e[u] = 0; s[u] = 0;
            if (q < hi) {
                e[u] = p.perm ? p.perm[q] : q;
                s[u] = p.sorted_src ? p.sorted_src[q] : (int32_t)p.src[e[u]];
            }
        }
    };
    // this lane's chunk i of the message of edge e (source vertex s); every block boundary is a multiple of four floats
    auto msg = [&](int32_t e, int32_t s, int i) -> float4 {
        const int col = (i * LPR + li) * 4;
        int o = col - p.da;
        if (o < 0) return *reinterpret_cast<const float4 *>(p.a + (int64_t)s * p.da + col);
        if (o < p.db)
            return p.b_per_node ? *reinterpret_cast<const float4 *>(p.b + (int64_t)s * p.ldb + o)
                                : vload_once(reinterpret_cast<const float4 *>(p.b + (int64_t)e * p.ldb + o));
        return vload_once(reinterpret_cast<const float4 *>(p.c + (int64_t)e * p.dc + (o - p.db)));
    };
    int64_t t = wave * RPW + sub;
    int32_t lo0, hi0, lo1, hi1, e0[NPF], s0[NPF];
    ldseg(t, lo0, hi0);
    ldseg(t + step, lo1, hi1);
    ldidx(lo0, hi0, e0, s0);
    for (int64_t t0 = wave * RPW; t0 < nn; t0 += step, t += step) {
        int32_t lo2, hi2, e1[NPF], s1[NPF];
        ldseg(t + 2 * step, lo2, hi2);
        ldidx(lo1, hi1, e1, s1);
        float4 acc[MAXC];
#pragma unroll
        for (int i = 0; i < MAXC; ++i) vzero(acc[i]);
        const int deg = hi0 - lo0;
#pragma unroll
        for (int u0 = 0; u0 < NPF; u0 += UNR) {
            if (u0 < deg) {
                float4 m[UNR][MAXC];
#pragma unroll
                for (int u = 0; u < UNR; ++u)
#pragma unroll
                    for (int i = 0; i < MAXC; ++i) {
                        vzero(m[u][i]);
                        if (u0 + u < deg && (i * LPR + li) * 4 < p.d_out) m[u][i] = msg(e0[u0 + u], s0[u0 + u], i);
                    }
#pragma unroll
                for (int u = 0; u < UNR; ++u)
                    if (u0 + u < deg) {
#pragma unroll
                        for (int i = 0; i < MAXC; ++i) acc[i] = vadd(acc[i], m[u][i]);
                    }
            }
        }
        for (int32_t q = lo0 + NPF; q < hi0; q += 4) {
            int32_t e[4], sv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                e[u] = 0; sv[u] = 0;
                if (q + u < hi0) {
                    e[u] = p.perm ? p.perm[q + u] : q + u;
                    sv[u] = p.sorted_src ? p.sorted_src[q + u] : (int32_t)p.src[e[u]];
                }
            }
            float4 m[4][MAXC];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < MAXC; ++i) {
                    vzero(m[u][i]);
                    if (q + u < hi0 && (i * LPR + li) * 4 < p.d_out) m[u][i] = msg(e[u], sv[u], i);
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (q + u < hi0) {
#pragma unroll
                    for (int i = 0; i < MAXC; ++i) acc[i] = vadd(acc[i], m[u][i]);
                }
        }
        if (t < nn) {
            if (EXT && p.n_self) {
                const float sc = 1.f + (p.eps ? *p.eps : 0.f);
#pragma unroll
                for (int i = 0; i < MAXC; ++i) {
                    const int col = (i * LPR + li) * 4;
                    if (col < p.d_out) {
                        float4 sv;
                        vzero(sv);
                        int o = col;
#pragma unroll
                        for (int k = 0; k < 3; ++k)
                            if (k < p.n_self) {
                                if (o >= 0 && o < p.self_w[k]) sv = *reinterpret_cast<const float4 *>(p.self_data[k] + t * p.self_stride[k] + o);
                                o -= p.self_w[k];
                            }
                        acc[i] = vfma(sc, sv, acc[i]);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                const int col = (i * LPR + li) * 4;
                if (col < p.d_out) vstore_once(reinterpret_cast<float4 *>(p.out + t * p.d_out + col), acc[i]);
            }
        }
        lo0 = lo1; hi0 = hi1; lo1 = lo2; hi1 = hi2;
#pragma unroll
        for (int u = 0; u < NPF; ++u) { e0[u] = e1[u]; s0[u] = s1[u]; }
    }
}

// ----------------------------------------------------------------------------------------------------------------
// backward
// ----------------------------------------------------------------------------------------------------------------
struct PropBwdArgs {
    int kind;
    int64_t n_nodes, n_edges;
    const int64_t *src, *tgt;
    const int32_t *seg_ptr_src, *perm_src;
    const float *a, *b, *c, *g_out;
    int da, db, dc, d_out;
    int b_per_node;
    int pad_b, pad_c;         // CAT: zero columns in front of b / c (no gradient)
    float *g_a, *g_b, *g_c;
    const float *g_edge;      // RELU_SUM: the masked per-edge gradient the edge kernel has just written (g_c, or a per-edge g_b), or null
    // r05: the layer's own term (1 + eps) * x folded into the node pass when the self block IS a (gsn_propagate_bwd_fold_self_hip)
    int fold_self;
    const float *eps;
    double *g_eps;
};

// per-edge gradients: g_b[e] (if per-edge) and g_c[e]; one row group per edge, scalar columns
__global__ __launch_bounds__(256) void propagate_bwd_edge_kernel(PropBwdArgs p) {
    const int64_t total = p.n_edges * (int64_t)p.d_out;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i / p.d_out;
        const int col = (int)(i - e * p.d_out);
        const int64_t t = p.tgt[e];
        float g = p.g_out[t * p.d_out + col];
        if (p.kind == GSN_MSG_CAT) {
            const int ob = col - p.da - p.pad_b, oc = ob - p.db - p.pad_c;
            if (ob >= 0 && ob < p.db) {
                if (!p.b_per_node && p.g_b) p.g_b[e * p.db + ob] = g;
            } else if (oc >= 0) {
                if (p.g_c) p.g_c[e * p.dc + oc] = g;
            }
        } else {
            const int64_t s = p.src[e];
            float pre = 0.f;
            if (p.a) pre += p.a[s * p.d_out + col];
            if (p.b) pre += p.b[(p.b_per_node ? s : e) * p.d_out + col];
            if (p.c) pre += p.c[e * p.d_out + col];
            g = pre > 0.f ? g : 0.f;
            if (!p.b_per_node && p.g_b) p.g_b[e * p.d_out + col] = g;
            if (p.g_c) p.g_c[e * p.d_out + col] = g;
        }
    }
}

// concatenation, every width a multiple of four floats, no zero columns (the readout's and the virtual node's adjoints: a pure row
// broadcast g_b[e] = g_out[tgt[e]]): a row group of 16 lanes per edge, float4 columns, the target read once per group -- the element-per-thread
// kernel above pays a 64-bit division and an 8-byte index load per float (70 -> ~25 us at 105 k x 300)
__global__ __launch_bounds__(256) void propagate_bwd_edge_cat4_kernel(PropBwdArgs p) {
    const int lane = threadIdx.x & 63, sub = lane >> 4, li = lane & 15;
    const int qa = p.da >> 2, qb = p.db >> 2, qc = p.dc >> 2;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t e0 = wave * 4; e0 < p.n_edges; e0 += n_waves * 4) {
        const int64_t e = e0 + sub;
        if (e >= p.n_edges) continue;
        const float4 *go = reinterpret_cast<const float4 *>(p.g_out + p.tgt[e] * p.d_out);
        if (!p.b_per_node && p.g_b) {
            float4 *gb = reinterpret_cast<float4 *>(p.g_b + e * p.db);
            for (int k = li; k < qb; k += 16) gb[k] = go[qa + k];
        }
        if (p.g_c) {
            float4 *gc = reinterpret_cast<float4 *>(p.g_c + e * p.dc);
            for (int k = li; k < qc; k += 16) gc[k] = go[qa + qb + k];
        }
    }
}

// the same for relu-sum messages whose width is a multiple of four floats (the d = 300 ogb layers): a row group of 16 lanes per edge,
// float4 columns, the edge's two indices read once per group instead of once per element (r03: 500 -> ~330 us at E = 214 k, d = 300)
__global__ __launch_bounds__(256) void propagate_bwd_edge_relu4_kernel(PropBwdArgs p) {
    const int lane = threadIdx.x & 63, sub = lane >> 4, li = lane & 15;
    const int q = p.d_out >> 2;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t e0 = wave * 4; e0 < p.n_edges; e0 += n_waves * 4) {
        const int64_t e = e0 + sub;
        if (e >= p.n_edges) continue;
        const int64_t t = p.tgt[e], s = p.src[e];
        const float4 *go = reinterpret_cast<const float4 *>(p.g_out + t * p.d_out);
        const float4 *pa = p.a ? reinterpret_cast<const float4 *>(p.a + s * p.d_out) : nullptr;
        const float4 *pb = p.b ? reinterpret_cast<const float4 *>(p.b + (p.b_per_node ? s : e) * p.d_out) : nullptr;
        const float4 *pc = p.c ? reinterpret_cast<const float4 *>(p.c + e * p.d_out) : nullptr;
        float4 *gb = (!p.b_per_node && p.g_b) ? reinterpret_cast<float4 *>(p.g_b + e * p.d_out) : nullptr;
        float4 *gc = p.g_c ? reinterpret_cast<float4 *>(p.g_c + e * p.d_out) : nullptr;
        for (int c4 = li; c4 < q; c4 += 16) {
            float4 g = go[c4];
            float4 pre = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pa) pre = vadd(pre, pa[c4]);
            if (pb) pre = vadd(pre, p.b_per_node ? pb[c4] : vload_once(pb + c4));
            if (pc) pre = vadd(pre, vload_once(pc + c4));
            g.x = pre.x > 0.f ? g.x : 0.f; g.y = pre.y > 0.f ? g.y : 0.f; g.z = pre.z > 0.f ? g.z : 0.f; g.w = pre.w > 0.f ? g.w : 0.f;
            if (gb) gb[c4] = g;
            if (gc) gc[c4] = g;
        }
    }
}

// the same with every column chunk of the edge's rows in flight together (d <= 320: five chunks of 16 float4) and the indices of the group's
// next edge requested before this edge's rows (the kernel above: index -> rows as two dependent round trips per edge, one chunk at a time)
__global__ __launch_bounds__(256) void propagate_bwd_edge_relu4p_kernel(PropBwdArgs p) {
    const int lane = threadIdx.x & 63, sub = lane >> 4, li = lane & 15;
    const int q = p.d_out >> 2;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = (((int64_t)gridDim.x * blockDim.x) >> 6) * 4;
    int64_t e = wave * 4 + sub;
    int64_t t = 0, s = 0;
    if (e < p.n_edges) { t = p.tgt[e]; s = p.src[e]; }
    for (int64_t e0 = wave * 4; e0 < p.n_edges; e0 += stride, e += stride) {
        int64_t tn = 0, sn = 0;
        if (e + stride < p.n_edges) { tn = p.tgt[e + stride]; sn = p.src[e + stride]; }
        if (e < p.n_edges) {
            const float4 *go = reinterpret_cast<const float4 *>(p.g_out + t * p.d_out);
            const float4 *pa = p.a ? reinterpret_cast<const float4 *>(p.a + s * p.d_out) : nullptr;
            const float4 *pb = p.b ? reinterpret_cast<const float4 *>(p.b + (p.b_per_node ? s : e) * p.d_out) : nullptr;
            const float4 *pc = p.c ? reinterpret_cast<const float4 *>(p.c + e * p.d_out) : nullptr;
            float4 *gb = (!p.b_per_node && p.g_b) ? reinterpret_cast<float4 *>(p.g_b + e * p.d_out) : nullptr;
            float4 *gc = p.g_c ? reinterpret_cast<float4 *>(p.g_c + e * p.d_out) : nullptr;
            float4 g[5], va[5], vb[5], vc[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int c4 = li + 16 * k;
                vzero(g[k]); vzero(va[k]); vzero(vb[k]); vzero(vc[k]);
                if (c4 < q) {
                    g[k] = go[c4];
                    if (pa) va[k] = pa[c4];
                    if (pb) vb[k] = p.b_per_node ? pb[c4] : vload_once(pb + c4);
                    if (pc) vc[k] = vload_once(pc + c4);
                }
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int c4 = li + 16 * k;
                if (c4 < q) {
                    // (0 + a) + b + c: the forward kernel's order of additions -- the mask must be the forward's mask
                    float4 pre = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (pa) pre = vadd(pre, va[k]);
                    if (pb) pre = vadd(pre, vb[k]);
                    if (pc) pre = vadd(pre, vc[k]);
                    float4 m = g[k];
                    m.x = pre.x > 0.f ? m.x : 0.f; m.y = pre.y > 0.f ? m.y : 0.f; m.z = pre.z > 0.f ? m.z : 0.f; m.w = pre.w > 0.f ? m.w : 0.f;
                    if (gb) gb[c4] = m;
                    if (gc) gc[c4] = m;
                }
            }
        }
        t = tn; s = sn;
    }
}

// per-node gradients through the source-sorted CSR: g_a[s] (and g_b[s] if per node) = sum over edges leaving s
__global__ __launch_bounds__(256) void propagate_bwd_node_kernel(PropBwdArgs p) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int width = p.kind == GSN_MSG_CAT ? p.da + (p.b_per_node ? p.db : 0) : p.d_out;
    for (int64_t s = wave; s < p.n_nodes; s += n_waves) {
        const int32_t lo = p.seg_ptr_src[s], hi = p.seg_ptr_src[s + 1];
        for (int col = lane; col < width; col += 64) {
            float acc = 0.f;
            for (int32_t q = lo; q < hi; ++q) {
                const int64_t e = p.perm_src[q];
                float g;
                if (p.g_edge) {
                    // relu-sum with a per-edge gradient already written by propagate_bwd_edge_kernel: that row IS relu'(pre) * g_out[t]
                    // (one stream instead of the three forward inputs + the gathered g_out row)
                    g = p.g_edge[e * p.d_out + col];
                } else {
                    const int64_t t = p.tgt[e];
                    g = p.g_out[t * p.d_out + col];
                    if (p.kind != GSN_MSG_CAT) {
                        float pre = 0.f;
                        if (p.a) pre += p.a[s * p.d_out + col];
                        if (p.b) pre += p.b[(p.b_per_node ? s : e) * p.d_out + col];
                        if (p.c) pre += p.c[e * p.d_out + col];
                        g = pre > 0.f ? g : 0.f;
                    }
                }
                acc += g;
            }
            if (p.kind == GSN_MSG_CAT) {
                if (col < p.da) { if (p.g_a) p.g_a[s * p.da + col] = acc; }
                else if (p.g_b) p.g_b[s * p.db + (col - p.da)] = acc;
            } else {
                if (p.g_a) p.g_a[s * p.d_out + col] = acc;
                if (p.b_per_node && p.g_b) p.g_b[s * p.d_out + col] = acc;
            }
        }
    }
}

// relu-sum node gradients from the masked per-edge rows (g_edge), float4 columns: a row group of 16 lanes per source vertex
__global__ __launch_bounds__(256) void propagate_bwd_node_edge4_kernel(PropBwdArgs p) {
    const int lane = threadIdx.x & 63, sub = lane >> 4, li = lane & 15;
    const int q = p.d_out >> 2;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t s0 = wave * 4; s0 < p.n_nodes; s0 += n_waves * 4) {
        const int64_t s = s0 + sub;
        if (s >= p.n_nodes) continue;
        const int32_t lo = p.seg_ptr_src[s], hi = p.seg_ptr_src[s + 1];
        // five column chunks of a row per lane and pass (d <= 320 in one): an edge's index is read once and its row's loads are in flight together
        for (int c0 = li; c0 < q; c0 += 80) {
            float4 acc[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int32_t qq = lo; qq < hi; ++qq) {
                const float4 *row = reinterpret_cast<const float4 *>(p.g_edge + (int64_t)p.perm_src[qq] * p.d_out);
#pragma unroll
                for (int k = 0; k < 5; ++k)
                    if (c0 + 16 * k < q) acc[k] = vadd(acc[k], row[c0 + 16 * k]);
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int c4 = c0 + 16 * k;
                if (c4 < q) {
                    if (p.g_a) reinterpret_cast<float4 *>(p.g_a + s * p.d_out)[c4] = acc[k];
                    if (p.b_per_node && p.g_b) reinterpret_cast<float4 *>(p.g_b + s * p.d_out)[c4] = acc[k];
                }
            }
        }
    }
}

// the same with the segment bounds of the group's next vertex and its first four edge ids requested before this vertex's rows, and the rows of two
// edges in flight together (d <= 320); same order of additions
// FOLD (the ogb layers: out = (1 + eps) x + sum relu(x_j + ..), the self block is the gathered block): g_a[s] += (1 + eps) g_out[s] in the same
// pass, g_eps = sum g_out . a as fp64 partial sums -- instead of an elementwise pass over [N d] that writes g_self, and the gradient-accumulation
// add of g_a + g_self behind it (two launches and six [N d] streams per layer and step)
template <bool FOLD>
__global__ __launch_bounds__(256) void propagate_bwd_node_edge4p_kernel(PropBwdArgs p) {
    const int lane = threadIdx.x & 63, sub = lane >> 4, li = lane & 15;
    const int q = p.d_out >> 2;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t stride = (((int64_t)gridDim.x * blockDim.x) >> 6) * 4, nn = p.n_nodes;
    auto ldseg = [&](int64_t v, int32_t &lo, int32_t &hi) {
        lo = hi = 0;
        if (v < nn) { lo = p.seg_ptr_src[v]; hi = p.seg_ptr_src[v + 1]; }
    };
    auto ldidx = [&](int32_t lo, int32_t hi, int32_t *e) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { e[u] = 0; if (lo + u < hi) e[u] = p.perm_src[lo + u]; }
    };
    int64_t s = wave * 4 + sub;
    int32_t lo0, hi0, lo1, hi1, e0[4];
    const float fold_sc = FOLD ? 1.f + (p.eps ? *p.eps : 0.f) : 0.f;
    double fold_dot = 0.0;
    ldseg(s, lo0, hi0);
    ldseg(s + stride, lo1, hi1);
    ldidx(lo0, hi0, e0);
    for (int64_t s0 = wave * 4; s0 < nn; s0 += stride, s += stride) {
        int32_t lo2, hi2, e1[4];
        ldseg(s + 2 * stride, lo2, hi2);
        ldidx(lo1, hi1, e1);
        float4 acc[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) vzero(acc[k]);
        const int deg = hi0 - lo0;
#pragma unroll
        for (int u0 = 0; u0 < 4; u0 += 2) {
            if (u0 < deg) {
                float4 r[2][5];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float4 *row = reinterpret_cast<const float4 *>(p.g_edge + (int64_t)e0[u0 + u] * p.d_out);
#pragma unroll
                    for (int k = 0; k < 5; ++k) {
                        vzero(r[u][k]);
                        if (u0 + u < deg && li + 16 * k < q) r[u][k] = row[li + 16 * k];
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (u0 + u < deg) {
#pragma unroll
                        for (int k = 0; k < 5; ++k) acc[k] = vadd(acc[k], r[u][k]);
                    }
            }
        }
        for (int32_t qq = lo0 + 4; qq < hi0; ++qq) {
            const float4 *row = reinterpret_cast<const float4 *>(p.g_edge + (int64_t)p.perm_src[qq] * p.d_out);
#pragma unroll
            for (int k = 0; k < 5; ++k)
                if (li + 16 * k < q) acc[k] = vadd(acc[k], row[li + 16 * k]);
        }
        if (s < nn) {
            if (FOLD) {
                const float4 *go = reinterpret_cast<const float4 *>(p.g_out + s * p.d_out);
                const float4 *av = reinterpret_cast<const float4 *>(p.a + s * p.d_out);
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const int c4 = li + 16 * k;
                    if (c4 < q) {
                        const float4 g = go[c4];
                        acc[k] = make_float4(acc[k].x + g.x * fold_sc, acc[k].y + g.y * fold_sc, acc[k].z + g.z * fold_sc, acc[k].w + g.w * fold_sc);
                        if (p.g_eps) {
                            const float4 v = av[c4];
                            fold_dot += (double)g.x * v.x + (double)g.y * v.y + (double)g.z * v.z + (double)g.w * v.w;
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int c4 = li + 16 * k;
                if (c4 < q) {
                    if (p.g_a) reinterpret_cast<float4 *>(p.g_a + s * p.d_out)[c4] = acc[k];
                    if (p.b_per_node && p.g_b) reinterpret_cast<float4 *>(p.g_b + s * p.d_out)[c4] = acc[k];
                }
            }
        }
        lo0 = lo1; hi0 = hi1; lo1 = lo2; hi1 = hi2;
#pragma unroll
        for (int u = 0; u < 4; ++u) e0[u] = e1[u];
    }
    if (FOLD && p.g_eps) {
        double se = fold_dot;
        for (int off = 32; off > 0; off >>= 1) se += __shfl_down(se, off);
        __shared__ double red[4];
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = se;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(p.g_eps, red[0] + red[1] + red[2] + red[3]);
    }
}

// the self term's adjoint when it is ONE per-node block as wide as the messages and no single-row sums are wanted (the ogb layers with
// per-edge identifiers): an elementwise pass over the flat [N d] arrays in float4s, eps from a block reduction
__global__ __launch_bounds__(256) void propagate_self_bwd_flat_kernel(int64_t n4, const float4 *__restrict__ g_out, const float4 *__restrict__ self,
                                                                      float4 *__restrict__ g_self, const float *eps, double *g_eps) {
    const float sc = 1.f + (eps ? *eps : 0.f);
    double se = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 g = g_out[i];
        if (g_eps) {
            const float4 v = self[i];
            se += (double)g.x * v.x + (double)g.y * v.y + (double)g.z * v.z + (double)g.w * v.w;
        }
        if (g_self) g_self[i] = make_float4(g.x * sc, g.y * sc, g.z * sc, g.w * sc);
    }
    if (g_eps) {
        for (int off = 32; off > 0; off >>= 1) se += __shfl_down(se, off);
        __shared__ double red[4];
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = se;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(g_eps, red[0] + red[1] + red[2] + red[3]);
    }
}

// adjoint of the self term  out[t] += (1 + eps) * self[t]:  g_self_k = (1 + eps) * (columns of g_out) for per-node blocks, the column
// sums of (1 + eps) * g_out for single-row blocks (fp64, the host slices them), g_eps = sum g_out . self (fp64).  One pass over g_out;
// lane = column, waves stride over rows (the layout of bn_act_bwd_reduce_kernel).
struct SelfBwdArgs {
    int kind;
    int64_t n_nodes;
    int d_out, n_self;
    const float *g_out;
    const float *self_data[3];
    int self_w[3];
    int64_t self_stride[3];
    float *g_self[3];
    const float *eps;
    double *g_eps, *g_colsum;
};

__global__ __launch_bounds__(256) void propagate_self_bwd_kernel(SelfBwdArgs p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + lane;
    const bool cok = c < p.d_out;
    const float sc = 1.f + (p.eps ? *p.eps : 0.f);
    // CAT: the block this column belongs to and the offset inside it
    int kb = 0, o = c;
    if (p.kind == GSN_MSG_CAT) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (k < p.n_self - 1 && kb == k && o >= p.self_w[k]) { o -= p.self_w[k]; kb = k + 1; }
    }
    double se = 0.0, scol = 0.0;
    if (cok) {
        for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < p.n_nodes; t += (int64_t)gridDim.x * 4) {
            const float g = p.g_out[t * p.d_out + c];
            const float gs = g * sc;
            float sv = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k < p.n_self) {
                    if (p.kind == GSN_MSG_CAT) {
                        if (k == kb) {
                            sv = p.self_data[k][t * p.self_stride[k] + o];
                            if (p.g_self[k]) p.g_self[k][t * p.self_w[k] + o] = gs;
                        }
                    } else {
                        sv += p.self_data[k][t * p.self_stride[k] + c];
                        if (p.g_self[k]) p.g_self[k][t * p.d_out + c] = gs;
                    }
                }
            }
            se += (double)g * (double)sv;
            scol += (double)gs;
        }
    }
    __shared__ double red[2][4][64];
    red[0][wave][lane] = se;
    red[1][wave][lane] = scol;
    __syncthreads();
    if (wave == 0) {
        double a = red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane];
        const double b = red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane];
        if (p.g_colsum && cok) atomicAdd(&p.g_colsum[c], b);
        if (p.g_eps) {
            for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
            if (lane == 0) atomicAdd(p.g_eps, a);
        }
    }
}

// Few targets, long segments (the graph-level readouts at the reference's batch sizes: 32 .. 128 graphs, ~26 rows each): ONE WORKGROUP per
// target, its rows dealt to the four waves (row q to wave q mod 4, four rows in flight per wave), partial sums added in wave order through
// LDS -- a fixed order, so the result is deterministic (it differs in the last bit from the sequential order of propagate_fwd_kernel,
// which spends ~1.3 us of load latency per row when one wave walks a segment alone: 29 us per readout at d = 300).
__global__ __launch_bounds__(256) void segment_sum_wg_kernel(PropArgs p) {
    __shared__ float4 red[3][256];
    const int t = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int32_t lo = p.seg_ptr[t], hi = p.seg_ptr[t + 1];
    const int q4 = p.d_out >> 2;
    for (int c0 = 0; c0 < q4; c0 += 64) {
        const int col = c0 + lane;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col < q4) {
            int32_t q = lo + wave;
            for (; q + 12 < hi; q += 16) {
                int64_t e[4];
                float4 m[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) e[u] = p.perm ? (int64_t)p.perm[q + 4 * u] : (int64_t)(q + 4 * u);
#pragma unroll
                for (int u = 0; u < 4; ++u) m[u] = *reinterpret_cast<const float4 *>(p.b + e[u] * p.ldb + 4 * col);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = vadd(acc, m[u]);
            }
            for (; q < hi; q += 4) {
                const int64_t e = p.perm ? (int64_t)p.perm[q] : (int64_t)q;
                acc = vadd(acc, *reinterpret_cast<const float4 *>(p.b + e * p.ldb + 4 * col));
            }
        }
        if (wave) red[wave - 1][lane] = acc;
        __syncthreads();
        if (wave == 0 && col < q4) {
            acc = vadd(vadd(vadd(acc, red[0][lane]), red[1][lane]), red[2][lane]);
            *reinterpret_cast<float4 *>(p.out + (int64_t)t * p.d_out + 4 * col) = acc;
        }
        __syncthreads();
    }
}

static int hip_check(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "%s: %s", what, hipGetErrorString(e));
    return GSN_OK;
}

template <int VEC, int LPR, int MAXC>
static int launch_fwd(const PropArgs &p, hipStream_t st) {
    constexpr int RPW = 64 / LPR;
    const int64_t waves_needed = (p.n_nodes + RPW - 1) / RPW;
    int64_t blocks = (waves_needed + 3) / 4;
    const int64_t cap = 256 * 8 * 4;  // enough workgroups to fill 256 CUs several times; grid-stride beyond
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    if (p.n_self || p.pad_b || p.pad_c) hipLaunchKernelGGL((propagate_fwd_kernel<VEC, LPR, MAXC, true>), dim3((unsigned)blocks), dim3(256), 0, st, p);
    else if (blocks <= 128) hipLaunchKernelGGL((propagate_fwd_kernel<VEC, LPR, MAXC, false, true>), dim3((unsigned)blocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((propagate_fwd_kernel<VEC, LPR, MAXC, false>), dim3((unsigned)blocks), dim3(256), 0, st, p);
    return hip_check("propagate_fwd_kernel");
}

template <int LPR, int MAXC, int UNR, bool NT>
static int launch_rs3(const PropArgs &p, hipStream_t st, int64_t cap) {
    constexpr int RPW = 64 / LPR;
    const int64_t waves_needed = (p.n_nodes + RPW - 1) / RPW;
    int64_t blocks = (waves_needed + 3) / 4;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    if (!p.c) {        // two streams: relu(x_j + e_e), or the identifier and edge-feature embeddings summed by their encoder (models.GNN_OGB)
        if (p.n_self) hipLaunchKernelGGL((relu_sum3_kernel<LPR, MAXC, UNR, NT, true, false>), dim3((unsigned)blocks), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((relu_sum3_kernel<LPR, MAXC, UNR, NT, false, false>), dim3((unsigned)blocks), dim3(256), 0, st, p);
    } else if (p.n_self) hipLaunchKernelGGL((relu_sum3_kernel<LPR, MAXC, UNR, NT, true>), dim3((unsigned)blocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((relu_sum3_kernel<LPR, MAXC, UNR, NT, false>), dim3((unsigned)blocks), dim3(256), 0, st, p);
    return hip_check("relu_sum3_kernel");
}

template <int LPR, int MAXC, int UNR>
static int launch_cp(const PropArgs &p, hipStream_t st, int64_t cap) {
    constexpr int RPW = 64 / LPR;
    const int64_t waves_needed = (p.n_nodes + RPW - 1) / RPW;
    int64_t blocks = (waves_needed + 3) / 4;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    if (p.n_self) hipLaunchKernelGGL((cat_pipe_kernel<LPR, MAXC, UNR, true>), dim3((unsigned)blocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((cat_pipe_kernel<LPR, MAXC, UNR, false>), dim3((unsigned)blocks), dim3(256), 0, st, p);
    return hip_check("cat_pipe_kernel");
}

// Where cat_pipe_kernel is taken (A/B on one box, 65 536 ZINC-shaped graphs, scripts/gpu/prop_cp.py, plain kernel -> pipelined): long segments
// (the readouts: every row of a graph is one "edge") d = 300 over 4 096 graphs 74.5 -> 44.9 us with 64 lanes x 2 chunks and four rows in flight,
// d = 128 over 65 536 graphs 176.8 -> 161.3 us (32 lanes, four rows); the gin aggregation of x_j alone, d = 128: 0.4005 -> 0.3819 ms (32 lanes, two
// edges).  NOT taken where it measured no better: the scatter-add of 128-wide per-edge rows (0.450 plain, 0.448-0.494 pipelined), the 112-wide and
// 104-wide gin concatenations (0.479 -> 0.470 at best, 0.384 -> 0.400).  GSN_PROP_CP = "lpr,unr[,blocks]" forces a mapping ("0": the plain
// kernel everywhere); -1 = not taken
static int launch_cat_pipe(const PropArgs &p, hipStream_t st) {
    const int q = p.d_out / 4;
    const bool long_segments = p.n_edges >= 8 * p.n_nodes;
    int l = 0, u = 4, b = 256 * 8 * 4;
    if (long_segments) l = q <= 8 ? 8 : q <= 16 ? 16 : q <= 32 ? 32 : 64;
    else if (p.da && !p.db && !p.dc && q > 16 && q <= 32) { l = 32; u = 2; }
    if (l && (q + l - 1) / l > 2) u = 2;
    if (const char *e = getenv("GSN_PROP_CP")) sscanf(e, "%d,%d,%d", &l, &u, &b);
    if (l == 0) return -1;
    const int m = (q + l - 1) / l;
    const int64_t cap = b;
#define CP(L, M) do { if (u == 1) return launch_cp<L, M, 1>(p, st, cap); if (u == 4 && M <= 2) return launch_cp<L, M, (M <= 2 ? 4 : 2)>(p, st, cap); return launch_cp<L, M, 2>(p, st, cap); } while (0)
    if (l == 8) { if (m <= 1) CP(8, 1); if (m <= 2) CP(8, 2); if (m <= 4) CP(8, 4); }
    if (l == 16) { if (m <= 1) CP(16, 1); if (m <= 2) CP(16, 2); if (m <= 4) CP(16, 4); if (m <= 5) CP(16, 5); }
    if (l == 32) { if (m <= 1) CP(32, 1); if (m <= 2) CP(32, 2); if (m <= 3) CP(32, 3); }
    if (l == 64) { if (m <= 1) CP(64, 1); if (m <= 2) CP(64, 2); if (m <= 4) CP(64, 4); }
#undef CP
    return -1;
}

// Mapping of relu_sum3_kernel by row width (A/B on one box, 65 536 ZINC-shaped graphs, d = 300, scripts/gpu/prop_rs.py; generic kernel 2.405 ms plain /
// 2.766 ms with the self term):  16 lanes x 5 chunks, 2 edges' rows in flight 2.010 / 2.082;  16 x 5, 1 edge 2.073 / 2.028;  32 x 3, 4 edges 2.075 / 2.250;
// 32 x 3, 2 edges 2.117 / 2.052;  64 x 2 2.14-2.25;  nontemporal per-edge loads +4 %;  fewer workgroups +1..4 %.  All variants: the same bits.
// GSN_PROP_RS = "lpr,unr,nt[,blocks]" forces one ("0": the generic kernel); -1 = not taken
static int launch_relu_sum3(const PropArgs &p, hipStream_t st) {
    const int q = p.d_out / 4;
    int l = q <= 80 ? 16 : 32, u = 0, n = 0, b = 256 * 8 * 4;
    if (const char *e = getenv("GSN_PROP_RS")) sscanf(e, "%d,%d,%d,%d", &l, &u, &n, &b);
    if (l == 0) return -1;
    if (u == 0) u = l == 16 ? (p.n_self ? 1 : 2) : (p.n_self ? 2 : 4);
    const bool nt = n != 0;
    const int64_t cap = b;
#define RS3(L, M, U) return nt ? launch_rs3<L, M, U, true>(p, st, cap) : launch_rs3<L, M, U, false>(p, st, cap)
    if (l == 32 && q <= 96) { if (u == 1) RS3(32, 3, 1); if (u == 2) RS3(32, 3, 2); if (u == 4) RS3(32, 3, 4); }
    if (l == 16 && q <= 80) { if (u == 1) RS3(16, 5, 1); if (u == 2) RS3(16, 5, 2); }
    if (l == 64 && q <= 128) { if (u == 1) RS3(64, 2, 1); if (u == 2) RS3(64, 2, 2); if (u == 4) RS3(64, 2, 4); }
#undef RS3
    return -1;
}


// ----------------------------------------------------------------------------------------------------------------
// Edge stage of a `general` layer with the node part of its Linear taken out of the edge loop.  The first Linear of msg_fn acts
// on cat(x_i, x_j, z_e) (GSN_sparse.py:166-171, GSN_edge_sparse.py:160-165, MPNN twins), and
//     cat(x_i, x_j, z_e) W^T = x_i W_i^T + x_j W_j^T + z_e W_z^T,
// so the two node terms are ONE product per NODE (P = x [W_i | W_j]^T, N rows instead of E, gsn_linear_fwd_hip) and what is left
// per edge is a gather-add:   S[t] = sum_{e -> t} act( P_i[t] + P_j[src_e] + z_e W_z^T )   (bias and the eval-mode BatchNorm are
// folded into P_i and the weights by the caller).  HBM-bound: one 4 d-byte row gather per edge + the z_e row, one read and one
// write per node; z_e W_z^T (d_z <= 16 columns) is done in registers.  One row group of LPR lanes per target, float4 columns.
// ----------------------------------------------------------------------------------------------------------------
struct SplitSumArgs {
    int64_t n_nodes;
    const int32_t *seg_ptr, *sorted_src, *perm;
    const float *tb, *a;            // P_i and P_j: rows of `pitch` floats
    int64_t pitch;
    const float *z0, *z1;           // per-edge blocks (rows addressed through perm) or null
    int w0, w1;                     // their widths (multiples of 4)
    const float *wz;                // [w0 + w1][d] = W_z^T
    int d, act;
    float *out;
};

template <int MAXC, int NZ4>
__global__ __launch_bounds__(256) void edge_split_sum_kernel(SplitSumArgs p) {
    constexpr int LPR = 32, RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR, li = lane % LPR;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    // this lane's columns of W_z^T
    float4 wz[NZ4 > 0 ? NZ4 * 4 : 1][MAXC];
#pragma unroll
    for (int k = 0; k < NZ4 * 4; ++k)
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int col = (i * LPR + li) * 4;
            wz[k][i] = (k < p.w0 + p.w1 && col < p.d) ? *reinterpret_cast<const float4 *>(p.wz + (int64_t)k * p.d + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    const int nz0 = p.w0 >> 2, nz = (p.w0 + p.w1) >> 2;
    for (int64_t t0 = wave * RPW; t0 < p.n_nodes; t0 += n_waves * RPW) {
        const int64_t t = t0 + sub;
        if (t >= p.n_nodes) continue;
        float4 m0[MAXC], acc[MAXC];
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int col = (i * LPR + li) * 4;
            vzero(acc[i]); vzero(m0[i]);
            if (col < p.d) m0[i] = *reinterpret_cast<const float4 *>(p.tb + t * p.pitch + col);
        }
        const int32_t lo = p.seg_ptr[t], hi = p.seg_ptr[t + 1];
        for (int32_t q = lo; q < hi; ++q) {
            const int64_t sv = p.sorted_src[q];
            float4 m[MAXC];
#pragma unroll
            for (int i = 0; i < MAXC; ++i) {
                const int col = (i * LPR + li) * 4;
                m[i] = m0[i];
                if (col < p.d) m[i] = vadd(m[i], *reinterpret_cast<const float4 *>(p.a + sv * p.pitch + col));
            }
            if (NZ4 > 0) {
                const int64_t e = p.perm[q];
#pragma unroll
                for (int k4 = 0; k4 < NZ4; ++k4) {
                    if (k4 < nz) {
                        const float4 z = k4 < nz0 ? *reinterpret_cast<const float4 *>(p.z0 + e * p.w0 + 4 * k4)
                                                  : *reinterpret_cast<const float4 *>(p.z1 + e * p.w1 + 4 * (k4 - nz0));
                        const float zz[4] = {z.x, z.y, z.z, z.w};
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int i = 0; i < MAXC; ++i) {
                                const float4 w = wz[4 * k4 + r][i];
                                m[i].x = fmaf(zz[r], w.x, m[i].x); m[i].y = fmaf(zz[r], w.y, m[i].y);
                                m[i].z = fmaf(zz[r], w.z, m[i].z); m[i].w = fmaf(zz[r], w.w, m[i].w);
                            }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < MAXC; ++i) acc[i] = vadd(acc[i], p.act ? vrelu(m[i]) : m[i]);
        }
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int col = (i * LPR + li) * 4;
            if (col < p.d) *reinterpret_cast<float4 *>(p.out + t * p.d + col) = acc[i];
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------
// CSR build of a PyG batch, one wave per GRAPH.  A collated batch is a disjoint union: graph g owns the consecutive
// columns edge_ptr[g] .. edge_ptr[g+1] of edge_index and the consecutive vertices node_ptr[g] .. node_ptr[g+1], so its
// columns keep that range in the target-sorted order as well and the whole stable counting sort of a graph -- histogram,
// scan, placement, order restore -- runs in LDS: one launch, no global atomics, no global scan, every input column read
// once and every output written once (the generic build above: seven launches over the same data).
// ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void csr_graphs_kernel(const int64_t *__restrict__ node_ptr, const int64_t *__restrict__ edge_ptr, int n_graphs,
                                                        int64_t n_nodes, int64_t n_edges, int n_cap, int e_cap,
                                                        const int64_t *__restrict__ index, const int64_t *__restrict__ other,
                                                        int32_t *seg_ptr, int32_t *perm, int32_t *sorted_target, int32_t *sorted_other,
                                                        int32_t *status) {
    extern __shared__ int32_t csr_lds[];
    int32_t *start = csr_lds;                  // [n_cap + 1] histogram -> exclusive scan
    int32_t *cur = start + (n_cap + 1);        // [n_cap + 1] placement cursors
    int32_t *tloc = cur + (n_cap + 1);         // [e_cap] graph-local target of every column
    int32_t *pl = tloc + e_cap;                // [e_cap] graph-local column ids in target order
    const int g = blockIdx.x, lane = threadIdx.x;
    const int64_t n0 = node_ptr[g], e0 = edge_ptr[g];
    const int64_t n64 = node_ptr[g + 1] - n0, E64 = edge_ptr[g + 1] - e0;
    // The pointers must describe THIS batch: graph g owns vertices [node_ptr[g], node_ptr[g+1]) and columns [edge_ptr[g], edge_ptr[g+1])
    // inside [0, n_nodes] x [0, n_edges], the last graph ends at n_edges (a column behind it would stay unsorted) and the first starts
    // at column 0.  Anything else raises GSN_ST_BAD_INDEX and this graph writes nothing outside its clamped vertex range.
    const bool ptr_ok = n0 >= 0 && e0 >= 0 && n64 >= 0 && E64 >= 0 && n0 + n64 <= n_nodes && e0 + E64 <= n_edges &&
                        (g != 0 || e0 == 0) && (g != n_graphs - 1 || e0 + E64 == n_edges);
    if (!ptr_ok) {
        if (lane == 0) atomicMax(status, (int)GSN_ST_BAD_INDEX);
        // keep every seg_ptr entry this workgroup is responsible for inside [0, n_edges]: later kernels walk seg_ptr
        const int64_t a0 = n0 < 0 ? 0 : (n0 > n_nodes ? n_nodes : n0);
        int64_t a1 = n0 + n64; a1 = a1 < a0 ? a0 : (a1 > n_nodes ? n_nodes : a1);
        for (int64_t v = a0 + lane; v < a1; v += 64) seg_ptr[v] = (int32_t)n_edges;
        if (g == n_graphs - 1) for (int64_t v = a1 + lane; v <= n_nodes; v += 64) seg_ptr[v] = (int32_t)n_edges;
        if (g == 0) for (int64_t v = lane; v < a0; v += 64) seg_ptr[v] = 0;
        return;
    }
    if (g == n_graphs - 1)                     // vertices behind the last graph (none in a collated batch) own no columns
        for (int64_t v = n0 + n64 + lane; v <= n_nodes; v += 64) seg_ptr[v] = (int32_t)n_edges;
    if (g == 0)
        for (int64_t v = lane; v < n0; v += 64) seg_ptr[v] = 0;
    if (n64 > n_cap || E64 > e_cap) {
        if (lane == 0) atomicMax(status, (int)GSN_ST_TOO_LARGE);
        // (the caller falls back to the generic build; until then the graph's vertices own no columns and its columns map to themselves)
        for (int64_t v = lane; v < n64; v += 64) seg_ptr[n0 + v] = (int32_t)e0;
        for (int64_t e = lane; e < E64; e += 64) {
            perm[e0 + e] = (int32_t)(e0 + e);
            if (sorted_target) sorted_target[e0 + e] = (int32_t)n0;
            if (sorted_other) sorted_other[e0 + e] = (int32_t)n0;
        }
        return;
    }
    const int n = (int)n64, E = (int)E64;
    for (int v = lane; v <= n; v += 64) start[v] = 0;
    __syncthreads();
    bool bad = false;
    for (int e = lane; e < E; e += 64) {
        const int64_t t = index[e0 + e] - n0;
        const bool ok = t >= 0 && t < n;
        bad = bad || !ok;
        const int tl = ok ? (int)t : 0;
        tloc[e] = tl;
        atomicAdd(&start[tl], 1);
    }
    if (bad) atomicMax(status, (int)GSN_ST_BAD_INDEX);      // a column whose target lies outside its graph: not a collated batch
    __syncthreads();
    // exclusive scan of the n + 1 counters (start[n] = 0 becomes E), 64 at a time
    int carry = 0;
    for (int base = 0; base <= n; base += 64) {
        const int v = base + lane;
        const int c = v <= n ? start[v] : 0;
        int incl = c;
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        const int ex = carry + incl - c;
        if (v <= n) { start[v] = ex; cur[v] = ex; }
        if (v < n) seg_ptr[n0 + v] = (int32_t)(e0 + ex);
        carry += __shfl(incl, 63);
    }
    __syncthreads();
    for (int e = lane; e < E; e += 64) pl[atomicAdd(&cur[tloc[e]], 1)] = e;
    __syncthreads();
    // restore column order inside every segment (the cursor hands out slots in arbitrary order)
    for (int v = lane; v < n; v += 64) {
        const int lo = start[v], hi = start[v + 1];
        for (int i = lo + 1; i < hi; ++i) {
            const int x = pl[i];
            int j = i - 1;
            while (j >= lo && pl[j] > x) { pl[j + 1] = pl[j]; --j; }
            pl[j + 1] = x;
        }
    }
    __syncthreads();
    for (int i = lane; i < E; i += 64) {
        const int le = pl[i];
        perm[e0 + i] = (int32_t)(e0 + le);
        if (sorted_target) sorted_target[e0 + i] = (int32_t)(n0 + tloc[le]);
        if (sorted_other) sorted_other[e0 + i] = (int32_t)other[e0 + le];
    }
}

}  // namespace gsn

using namespace gsn;

extern "C" int64_t gsn_csr_scratch_elems(int64_t n_nodes) { return (n_nodes + 1) + (n_nodes + 1) / SCAN_TILE + 2; }

extern "C" int gsn_segsum_prepare_hip(int64_t n_seg, int64_t n_rows, const int32_t *seg_ptr, const int32_t *row_target,
                                      int64_t n_out, float *out, void *stream) {
    if (!seg_ptr || !out || (n_rows > 0 && !row_target) || n_out <= 0) return set_error(GSN_E_INVALID, "gsn_segsum_prepare_hip: bad argument");
    const int64_t items = n_seg + n_rows / GSN_SEG_RANGE_ROWS;
    if (items <= 0) return GSN_OK;
    hipLaunchKernelGGL(segsum_prepare_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       seg_ptr, row_target, n_seg, n_rows, (int)n_out, out);
    return hip_check("gsn_segsum_prepare_hip");
}

extern "C" int gsn_csr_build_hip(int64_t n_nodes, int64_t n_edges, const int64_t *index, const int64_t *other, int32_t *seg_ptr,
                                 int32_t *perm, int32_t *sorted_target, int32_t *sorted_other, int32_t *scratch, void *stream) {
    if (sorted_other && !other) return set_error(GSN_E_INVALID, "gsn_csr_build_hip: sorted_other needs `other`");
    if (n_nodes < 0 || n_edges < 0 || !seg_ptr || !scratch || (n_edges > 0 && (!index || !perm)))
        return set_error(GSN_E_INVALID, "gsn_csr_build_hip: bad argument");
    if (n_edges >= (int64_t)1 << 31 || n_nodes >= ((int64_t)1 << 31) - 1)
        return set_error(GSN_E_UNSUPPORTED, "gsn_csr_build_hip: more than 2^31 edges or vertices");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (n_nodes <= CSR_SMALL_NODES && n_edges <= CSR_SMALL_EDGES) {
        static DeviceOnce lds_set;
        const int lds_dev = current_device();
        if (!lds_set.done(lds_dev)) {   // up to 2 x 48 KiB of dynamic LDS
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&csr_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    2 * (CSR_SMALL_NODES + 1) * (int)sizeof(int32_t)) != hipSuccess)
                return set_error(GSN_E_HIP, "gsn_csr_build_hip: cannot raise the LDS limit of csr_small_kernel");
            lds_set.mark(lds_dev);
        }
        hipLaunchKernelGGL(csr_small_kernel, dim3(1), dim3(1024), (size_t)(2 * (n_nodes + 1)) * sizeof(int32_t), st, index, n_edges,
                           (int)n_nodes, seg_ptr, perm, sorted_target, other, sorted_other);
        return hip_check("gsn_csr_build_hip");
    }
    const int64_t n1 = n_nodes + 1;
    int32_t *cnt = scratch;                 // [n1] histogram, later the fill cursor
    int32_t *tiles = scratch + n1;          // [n_tiles]
    const int64_t n_tiles = (n1 + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(csr_zero, dim3((unsigned)((n1 + CSR_T - 1) / CSR_T)), dim3(CSR_T), 0, st, cnt, n1);
    if (n_edges > 0) {
        int64_t blocks = (n_edges + CSR_T - 1) / CSR_T;
        if (blocks > 32768) blocks = 32768;
        hipLaunchKernelGGL(csr_hist, dim3((unsigned)blocks), dim3(CSR_T), 0, st, index, n_edges, cnt);
    }
    hipLaunchKernelGGL(scan_tile_sums, dim3((unsigned)n_tiles), dim3(SCAN_T), 0, st, cnt, n1, tiles);
    hipLaunchKernelGGL(scan_tile_offsets, dim3(1), dim3(64), 0, st, tiles, n_tiles);
    hipLaunchKernelGGL(scan_apply, dim3((unsigned)n_tiles), dim3(SCAN_T), 0, st, cnt, n1, tiles, seg_ptr, cnt);
    if (n_edges > 0) {
        int64_t blocks = (n_edges + CSR_T - 1) / CSR_T;
        if (blocks > 32768) blocks = 32768;
        hipLaunchKernelGGL(csr_fill, dim3((unsigned)blocks), dim3(CSR_T), 0, st, index, n_edges, cnt, perm);
        hipLaunchKernelGGL(csr_sort_segments, dim3((unsigned)((n_nodes + CSR_T - 1) / CSR_T)), dim3(CSR_T), 0, st, seg_ptr, n_nodes, perm, sorted_target, other, sorted_other);
    }
    return hip_check("gsn_csr_build_hip");
}

extern "C" int gsn_propagate_fwd_hip(int kind, int64_t n_nodes, int64_t n_edges, const int64_t *src, const int32_t *seg_ptr,
                                     const int32_t *perm, const int32_t *sorted_src, const float *a, int64_t da, const float *b, int64_t db,
                                     int b_per_node, const float *c, int64_t dc, float *out, void *stream) {
    return gsn_propagate_self_fwd_hip(kind, n_nodes, n_edges, src, seg_ptr, perm, sorted_src, a, da, b, db, b_per_node, c, dc, 0, 0, 0,
                                      nullptr, nullptr, out, stream);
}

static int propagate_fwd_impl(int kind, int64_t n_nodes, int64_t n_edges, const int64_t *src, const int32_t *seg_ptr,
                              const int32_t *perm, const int32_t *sorted_src, const float *a, int64_t da, const float *b,
                              int64_t db, int64_t ldb, int b_per_node, const float *c, int64_t dc, int64_t pad_b, int64_t pad_c,
                              int n_self, const gsn_self_block *self_blocks, const float *eps, float *out, void *stream);

extern "C" int gsn_propagate_self_fwd_hip(int kind, int64_t n_nodes, int64_t n_edges, const int64_t *src, const int32_t *seg_ptr,
                                          const int32_t *perm, const int32_t *sorted_src, const float *a, int64_t da, const float *b,
                                          int64_t db, int b_per_node, const float *c, int64_t dc, int64_t pad_b, int64_t pad_c,
                                          int n_self, const gsn_self_block *self_blocks, const float *eps, float *out, void *stream) {
    return propagate_fwd_impl(kind, n_nodes, n_edges, src, seg_ptr, perm, sorted_src, a, da, b, db, db, b_per_node, c, dc, pad_b, pad_c, n_self,
                              self_blocks, eps, out, stream);
}

// out[t] = sum over the rows of t's segment of b[row][0 .. width), b a COLUMN SLICE of wider rows (row stride ld floats): the per-vertex sums of
// a gathered block's input gradient, taken where the input-gradient product left it (no contiguous copy of the slice)
extern "C" int gsn_segment_sum_rows_hip(int64_t n_nodes, int64_t n_edges, const int64_t *src, const int32_t *seg_ptr, const int32_t *perm,
                                        const int32_t *sorted_src, const float *b, int64_t width, int64_t ld, float *out, void *stream) {
    if (width < 1 || ld < width) return set_error(GSN_E_INVALID, "gsn_segment_sum_rows_hip: width %lld, row stride %lld", (long long)width, (long long)ld);
    return propagate_fwd_impl(GSN_MSG_CAT, n_nodes, n_edges, src, seg_ptr, perm, sorted_src, nullptr, 0, b, width, ld, 0, nullptr, 0, 0, 0, 0, nullptr,
                              nullptr, out, stream);
}

static int propagate_fwd_impl(int kind, int64_t n_nodes, int64_t n_edges, const int64_t *src, const int32_t *seg_ptr,
                              const int32_t *perm, const int32_t *sorted_src, const float *a, int64_t da, const float *b,
                              int64_t db, int64_t ldb, int b_per_node, const float *c, int64_t dc, int64_t pad_b, int64_t pad_c,
                              int n_self, const gsn_self_block *self_blocks, const float *eps, float *out, void *stream) {
    if (kind != GSN_MSG_CAT && kind != GSN_MSG_RELU_SUM) return set_error(GSN_E_INVALID, "gsn_propagate_fwd_hip: unknown kind %d", kind);
    if (pad_b < 0 || pad_c < 0 || n_self < 0 || n_self > 3 || (n_self > 0 && !self_blocks))
        return set_error(GSN_E_INVALID, "gsn_propagate_self_fwd_hip: bad pads / self blocks");
    if ((pad_b || pad_c) && (kind != GSN_MSG_CAT || (pad_b && b_per_node)))
        return set_error(GSN_E_INVALID, "gsn_propagate_self_fwd_hip: zero columns only in front of the per-edge blocks of a concatenation");
    if (!seg_ptr || !out || (n_edges > 0 && !src)) return set_error(GSN_E_INVALID, "gsn_propagate_fwd_hip: null pointer");
    // widths are authoritative; a null pointer is only legal for a block that is never dereferenced
    if ((da > 0 && !a && n_edges > 0) || (db > 0 && !b && n_edges > 0) || (dc > 0 && !c && n_edges > 0))
        return set_error(GSN_E_INVALID, "gsn_propagate_fwd_hip: a block has width > 0 but no data");
    if (da < 0 || db < 0 || dc < 0) return set_error(GSN_E_INVALID, "gsn_propagate_fwd_hip: negative width");
    int64_t d_out;
    if (kind == GSN_MSG_CAT) d_out = da + (db ? pad_b + db : 0) + (dc ? pad_c + dc : 0);
    else {
        d_out = da > db ? da : db;
        d_out = d_out > dc ? d_out : dc;
        if ((da && da != d_out) || (db && db != d_out) || (dc && dc != d_out))
            return set_error(GSN_E_INVALID, "gsn_propagate_fwd_hip: relu-sum blocks must share one width");
    }
    if ((pad_b && !db) || (pad_c && !dc)) return set_error(GSN_E_INVALID, "gsn_propagate_self_fwd_hip: padding in front of an empty block");
    uintptr_t self_align = 0;
    int64_t self_or = 0, self_sum = 0;
    for (int k = 0; k < n_self; ++k) {
        if (!self_blocks[k].data || self_blocks[k].width <= 0 || self_blocks[k].row_stride < 0)
            return set_error(GSN_E_INVALID, "gsn_propagate_self_fwd_hip: self block %d is empty", k);
        if (kind == GSN_MSG_RELU_SUM && self_blocks[k].width != d_out)
            return set_error(GSN_E_INVALID, "gsn_propagate_self_fwd_hip: self block %d: width %lld, expected %lld", k,
                             (long long)self_blocks[k].width, (long long)d_out);
        self_align |= (uintptr_t)self_blocks[k].data;
        self_or |= self_blocks[k].width | self_blocks[k].row_stride;
        self_sum += self_blocks[k].width;
    }
    if (n_self && kind == GSN_MSG_CAT) {
        if (d_out == 0) d_out = self_sum;      // (no edges' worth of blocks: the self term alone)
        if (self_sum != d_out)
            return set_error(GSN_E_INVALID, "gsn_propagate_self_fwd_hip: self blocks are %lld columns wide, messages %lld",
                             (long long)self_sum, (long long)d_out);
    }
    if (d_out <= 0 || n_nodes <= 0) return GSN_OK;
    if (d_out > 1024) return set_error(GSN_E_UNSUPPORTED, "gsn_propagate_fwd_hip: message width %lld > 1024", (long long)d_out);
    PropArgs p{};
    p.pad_b = db ? (int)pad_b : 0; p.pad_c = dc ? (int)pad_c : 0; p.n_self = n_self; p.eps = eps;
    for (int k = 0; k < n_self; ++k) {
        p.self_data[k] = self_blocks[k].data; p.self_w[k] = (int)self_blocks[k].width; p.self_stride[k] = self_blocks[k].row_stride;
    }
    p.kind = kind; p.n_nodes = n_nodes; p.n_edges = n_edges; p.src = src; p.seg_ptr = seg_ptr; p.perm = perm;
    p.sorted_src = sorted_src;
    p.a = da ? a : nullptr; p.b = db ? b : nullptr; p.c = dc ? c : nullptr;
    p.da = (int)da; p.db = (int)db; p.dc = (int)dc; p.d_out = (int)d_out; p.ldb = ldb;
    p.b_per_node = b_per_node; p.out = out;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool aligned = ((da | db | dc | ldb | pad_b | pad_c | self_or) % 4 == 0) && pad_b == 0 && pad_c == 0 &&
                         (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)out | self_align) % 16 == 0);
    if (aligned && kind == GSN_MSG_CAT && !da && !dc && db && !b_per_node && !n_self && n_nodes <= 512 && n_edges >= 8 * n_nodes) {
        hipLaunchKernelGGL(segment_sum_wg_kernel, dim3((unsigned)n_nodes), dim3(256), 0, st, p);
        return hip_check("segment_sum_wg_kernel");
    }
    if (aligned && kind == GSN_MSG_RELU_SUM && p.a && (p.b || p.c) && !b_per_node && d_out >= 132 && d_out <= 384) {
        if (!p.b) { p.b = p.c; p.c = nullptr; }        // (one per-edge stream: it is the kernel's block b)
        const int rc3 = launch_relu_sum3(p, st);
        if (rc3 != -1) return rc3;
    }
    if (aligned && kind == GSN_MSG_CAT) {
        const int rcp = launch_cat_pipe(p, st);
        if (rcp != -1) return rcp;
    }
    if (aligned) {
        const int64_t q = d_out / 4;  // float4 per row
        if (q <= 8) return launch_fwd<4, 8, 1>(p, st);
        if (q <= 16) return launch_fwd<4, 16, 1>(p, st);
        // lanes per target row: the widths up to 128 floats take 16 (four targets per wave: four rows' loads in flight per wave instead of two
        // -- 0.656 -> 0.463 ms for the 112-wide concatenation, 0.422 -> 0.394 at d = 128, r04); GSN_PROP_LPR=8/16/32/64 forces a mapping
        static const int forced = [] { const char *e = getenv("GSN_PROP_LPR"); return e ? atoi(e) : 0; }();
        if (forced) {
            const int m = (int)((q + forced - 1) / forced);
            if (forced == 8) { if (m <= 1) return launch_fwd<4, 8, 1>(p, st); if (m <= 2) return launch_fwd<4, 8, 2>(p, st); if (m <= 4) return launch_fwd<4, 8, 4>(p, st); }
            if (forced == 16) { if (m <= 1) return launch_fwd<4, 16, 1>(p, st); if (m <= 2) return launch_fwd<4, 16, 2>(p, st); if (m <= 4) return launch_fwd<4, 16, 4>(p, st);
                                if (m <= 5) return launch_fwd<4, 16, 5>(p, st); }
            if (forced == 32) { if (m <= 1) return launch_fwd<4, 32, 1>(p, st); if (m <= 2) return launch_fwd<4, 32, 2>(p, st); if (m <= 3) return launch_fwd<4, 32, 3>(p, st); }
        }
        // (short segments -- ~2 edges per target, the message aggregation -- want many targets per wave; long ones -- a readout's ~23 rows per
        //  graph -- walk their rows serially per lane group and want the wider group: 42.7 vs 59.9 us per 16 384-graph readout)
        if (q <= 32) return (q > 16 && n_edges >= 8 * n_nodes) ? launch_fwd<4, 32, 1>(p, st) : launch_fwd<4, 16, 2>(p, st);
        if (q <= 64) return launch_fwd<4, 64, 1>(p, st);
        if (q <= 96) return launch_fwd<4, 32, 3>(p, st);       // (d = 300: 75 float4 per row -- 2.57 -> 2.51 ms against 64 lanes x 2)
        if (q <= 128) return launch_fwd<4, 64, 2>(p, st);
        return launch_fwd<4, 64, 4>(p, st);
    }
    if (d_out <= 16) return launch_fwd<1, 16, 1>(p, st);
    if (d_out <= 32) return launch_fwd<1, 32, 1>(p, st);
    if (d_out <= 64) return launch_fwd<1, 64, 1>(p, st);
    if (d_out <= 256) return launch_fwd<1, 64, 4>(p, st);
    return launch_fwd<1, 64, 16>(p, st);
}

extern "C" int gsn_propagate_bwd_hip(int kind, int64_t n_nodes, int64_t n_edges, const int64_t *src, const int64_t *tgt,
                                     const int32_t *seg_ptr_src, const int32_t *perm_src, const float *a, int64_t da,
                                     const float *b, int64_t db, int b_per_node, const float *c, int64_t dc,
                                     const float *g_out, float *g_a, float *g_b, float *g_c, void *stream) {
    return gsn_propagate_pad_bwd_hip(kind, n_nodes, n_edges, src, tgt, seg_ptr_src, perm_src, a, da, b, db, b_per_node, c, dc, 0, 0, g_out,
                                     g_a, g_b, g_c, stream);
}

static int propagate_bwd_impl(int kind, int64_t n_nodes, int64_t n_edges, const int64_t *src, const int64_t *tgt,
                              const int32_t *seg_ptr_src, const int32_t *perm_src, const float *a, int64_t da,
                              const float *b, int64_t db, int b_per_node, const float *c, int64_t dc, int64_t pad_b,
                              int64_t pad_c, const float *g_out, float *g_a, float *g_b, float *g_c, void *stream,
                              int fold_self, const float *eps, double *g_eps);

extern "C" int gsn_propagate_pad_bwd_hip(int kind, int64_t n_nodes, int64_t n_edges, const int64_t *src, const int64_t *tgt,
                                         const int32_t *seg_ptr_src, const int32_t *perm_src, const float *a, int64_t da,
                                         const float *b, int64_t db, int b_per_node, const float *c, int64_t dc, int64_t pad_b,
                                         int64_t pad_c, const float *g_out, float *g_a, float *g_b, float *g_c, void *stream) {
    return propagate_bwd_impl(kind, n_nodes, n_edges, src, tgt, seg_ptr_src, perm_src, a, da, b, db, b_per_node, c, dc, pad_b, pad_c, g_out, g_a, g_b,
                              g_c, stream, 0, nullptr, nullptr);
}

// The relu-sum adjoint of a layer whose own term is its gathered block:  out = (1 + eps) a + sum_{e -> t} relu(a[src_e] + b_e + c_e)
// (GSN_edge_sparse_ogb.py:63-84 / :103-106 with self = x).  g_a receives BOTH contributions -- the per-source sums of the masked per-edge
// gradients and (1 + eps) g_out -- in the node pass, g_eps (fp64, added to; may be null) = sum g_out . a.  Replaces gsn_propagate_pad_bwd_hip +
// gsn_propagate_self_bwd_hip + the sum of their two results.  GSN_E_UNSUPPORTED (nothing launched) where the folded pass does not apply: no
// edges, widths above 320 or not multiples of four, unaligned rows, no per-edge gradient asked for.
extern "C" int gsn_propagate_bwd_fold_self_hip(int64_t n_nodes, int64_t n_edges, const int64_t *src, const int64_t *tgt,
                                               const int32_t *seg_ptr_src, const int32_t *perm_src, const float *a, int64_t d, const float *b,
                                               int64_t db, const float *c, int64_t dc, const float *g_out, float *g_a, float *g_b, float *g_c,
                                               const float *eps, double *g_eps, void *stream) {
    static const bool pipe = [] { const char *e = getenv("GSN_PROP_BWD_PIPE"); return !e || atoi(e) != 0; }();
    static const bool fold = [] { const char *e = getenv("GSN_PROP_FOLD_SELF"); return !e || atoi(e) != 0; }();
    const bool need_edge = (g_b && db) || (g_c && dc);
    if (!fold || !pipe || !a || !g_a || !g_out || d <= 0 || d > 320 || (d & 3) || n_edges <= 0 || !need_edge || (db && db != d) || (dc && dc != d) ||
        (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)g_out | (uintptr_t)g_a | (uintptr_t)g_b | (uintptr_t)g_c) % 16) != 0)
        return set_error(GSN_E_UNSUPPORTED, "gsn_propagate_bwd_fold_self_hip: shape outside the folded pass");
    return propagate_bwd_impl(GSN_MSG_RELU_SUM, n_nodes, n_edges, src, tgt, seg_ptr_src, perm_src, a, d, b, db, 0, c, dc, 0, 0, g_out, g_a, g_b, g_c,
                              stream, 1, eps, g_eps);
}

static int propagate_bwd_impl(int kind, int64_t n_nodes, int64_t n_edges, const int64_t *src, const int64_t *tgt,
                              const int32_t *seg_ptr_src, const int32_t *perm_src, const float *a, int64_t da,
                              const float *b, int64_t db, int b_per_node, const float *c, int64_t dc, int64_t pad_b,
                              int64_t pad_c, const float *g_out, float *g_a, float *g_b, float *g_c, void *stream,
                              int fold_self, const float *eps, double *g_eps) {
    if (kind != GSN_MSG_CAT && kind != GSN_MSG_RELU_SUM) return set_error(GSN_E_INVALID, "gsn_propagate_bwd_hip: unknown kind %d", kind);
    if (!g_out || (n_edges > 0 && (!src || !tgt))) return set_error(GSN_E_INVALID, "gsn_propagate_bwd_hip: null pointer");
    if (pad_b < 0 || pad_c < 0 || ((pad_b || pad_c) && (kind != GSN_MSG_CAT || (pad_b && b_per_node))))
        return set_error(GSN_E_INVALID, "gsn_propagate_pad_bwd_hip: zero columns only in front of the per-edge blocks of a concatenation");
    if (!db) pad_b = 0;
    if (!dc) pad_c = 0;
    int64_t d_out;
    if (kind == GSN_MSG_CAT) d_out = da + pad_b + db + pad_c + dc;
    else { d_out = da > db ? da : db; d_out = d_out > dc ? d_out : dc; }
    if (d_out <= 0 || n_nodes <= 0) return GSN_OK;
    // (per-edge blocks of an edge-less batch are empty tensors: no pointer to give)
    if (kind == GSN_MSG_RELU_SUM && ((da && !a) || (db && !b && (b_per_node || n_edges > 0)) || (dc && !c && n_edges > 0)))
        return set_error(GSN_E_INVALID, "gsn_propagate_bwd_hip: relu-sum needs the forward inputs");
    PropBwdArgs p{};
    p.kind = kind; p.n_nodes = n_nodes; p.n_edges = n_edges; p.src = src; p.tgt = tgt;
    p.seg_ptr_src = seg_ptr_src; p.perm_src = perm_src;
    p.a = da ? a : nullptr; p.b = db ? b : nullptr; p.c = dc ? c : nullptr; p.g_out = g_out;
    p.da = (int)da; p.db = (int)db; p.dc = (int)dc; p.d_out = (int)d_out; p.b_per_node = b_per_node;
    p.pad_b = (int)pad_b; p.pad_c = (int)pad_c;
    p.g_a = g_a; p.g_b = g_b; p.g_c = g_c;
    p.fold_self = fold_self; p.eps = eps; p.g_eps = g_eps;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool need_edge = (g_b && !b_per_node && db) || (g_c && dc);
    if (need_edge && n_edges > 0) {
        const bool vec4 = kind == GSN_MSG_RELU_SUM && d_out % 4 == 0 &&
                          (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)g_out | (uintptr_t)g_b | (uintptr_t)g_c) % 16 == 0);
        const bool cat4 = kind == GSN_MSG_CAT && !pad_b && !pad_c && ((da | db | dc) & 3) == 0 &&
                          (((uintptr_t)g_out | (uintptr_t)g_b | (uintptr_t)g_c) % 16 == 0);
        if (vec4 || cat4) {
            int64_t blocks = (n_edges + 15) / 16;
            if (blocks > 16384) blocks = 16384;
            static const bool pipe = [] { const char *e = getenv("GSN_PROP_BWD_PIPE"); return !e || atoi(e) != 0; }();
            if (vec4 && pipe && d_out <= 320) hipLaunchKernelGGL(propagate_bwd_edge_relu4p_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
            else if (vec4) hipLaunchKernelGGL(propagate_bwd_edge_relu4_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
            else hipLaunchKernelGGL(propagate_bwd_edge_cat4_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
        } else {
            int64_t blocks = (n_edges * d_out + 255) / 256;
            if (blocks > 16384) blocks = 16384;
            hipLaunchKernelGGL(propagate_bwd_edge_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
        }
    }
    const bool need_node = (g_a && da) || (g_b && b_per_node && db);
    if (kind == GSN_MSG_RELU_SUM && need_edge && n_edges > 0) p.g_edge = (g_c && dc) ? g_c : g_b;
    if (need_node) {
        if (!seg_ptr_src || (n_edges > 0 && !perm_src)) return set_error(GSN_E_INVALID, "gsn_propagate_bwd_hip: source CSR missing");
        const bool edge4 = p.g_edge && kind == GSN_MSG_RELU_SUM && d_out % 4 == 0 &&
                           (((uintptr_t)p.g_edge | (uintptr_t)g_a | (uintptr_t)(b_per_node ? g_b : nullptr)) % 16 == 0);
        if (edge4) {
            int64_t blocks = (n_nodes + 15) / 16;
            if (blocks > 16384) blocks = 16384;
            static const bool pipe = [] { const char *e = getenv("GSN_PROP_BWD_PIPE"); return !e || atoi(e) != 0; }();
            if (p.fold_self) {
                if (!(pipe && d_out <= 320)) return set_error(GSN_E_UNSUPPORTED, "gsn_propagate_bwd_fold_self_hip: the folded self term rides the pipelined node pass (d <= 320)");
                hipLaunchKernelGGL(propagate_bwd_node_edge4p_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, p);
            } else if (pipe && d_out <= 320) hipLaunchKernelGGL(propagate_bwd_node_edge4p_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, p);
            else hipLaunchKernelGGL(propagate_bwd_node_edge4_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
        } else {
            int64_t blocks = (n_nodes + 3) / 4;
            if (blocks > 16384) blocks = 16384;
            hipLaunchKernelGGL(propagate_bwd_node_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
        }
    }
    return hip_check("gsn_propagate_bwd_hip");
}

extern "C" int gsn_propagate_self_bwd_hip(int kind, int64_t n_nodes, int64_t d_out, const float *g_out, int n_self,
                                          const gsn_self_block *self_blocks, float *const *g_self, const float *eps, double *g_eps,
                                          double *g_colsum, void *stream) {
    if (kind != GSN_MSG_CAT && kind != GSN_MSG_RELU_SUM) return set_error(GSN_E_INVALID, "gsn_propagate_self_bwd_hip: unknown kind %d", kind);
    if (n_self < 1 || n_self > 3 || !self_blocks || !g_out || d_out < 1 || d_out > 1024)
        return set_error(GSN_E_INVALID, "gsn_propagate_self_bwd_hip: bad arguments");
    SelfBwdArgs p{};
    p.kind = kind; p.n_nodes = n_nodes; p.d_out = (int)d_out; p.n_self = n_self; p.g_out = g_out; p.eps = eps; p.g_eps = g_eps; p.g_colsum = g_colsum;
    int64_t sum = 0;
    for (int k = 0; k < n_self; ++k) {
        if (!self_blocks[k].data || self_blocks[k].width <= 0 || self_blocks[k].row_stride < 0)
            return set_error(GSN_E_INVALID, "gsn_propagate_self_bwd_hip: self block %d is empty", k);
        if (kind == GSN_MSG_RELU_SUM && self_blocks[k].width != d_out)
            return set_error(GSN_E_INVALID, "gsn_propagate_self_bwd_hip: self block %d: width %lld, expected %lld", k,
                             (long long)self_blocks[k].width, (long long)d_out);
        p.self_data[k] = self_blocks[k].data; p.self_w[k] = (int)self_blocks[k].width; p.self_stride[k] = self_blocks[k].row_stride;
        p.g_self[k] = g_self ? g_self[k] : nullptr;
        sum += self_blocks[k].width;
    }
    if (kind == GSN_MSG_CAT && sum != d_out)
        return set_error(GSN_E_INVALID, "gsn_propagate_self_bwd_hip: self blocks are %lld columns wide, g_out %lld", (long long)sum, (long long)d_out);
    if (n_nodes <= 0) return GSN_OK;
    if (n_self == 1 && !g_colsum && self_blocks[0].width == d_out && self_blocks[0].row_stride == d_out && (n_nodes * d_out) % 4 == 0 &&
        (((uintptr_t)g_out | (uintptr_t)self_blocks[0].data | (uintptr_t)p.g_self[0]) % 16 == 0)) {
        const int64_t n4 = n_nodes * d_out / 4;
        int64_t bx = (n4 + 255) / 256;
        if (bx > 4096) bx = 4096;
        hipLaunchKernelGGL(propagate_self_bwd_flat_kernel, dim3((unsigned)bx), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n4,
                           reinterpret_cast<const float4 *>(g_out), reinterpret_cast<const float4 *>(self_blocks[0].data),
                           reinterpret_cast<float4 *>(p.g_self[0]), eps, g_eps);
        return hip_check("propagate_self_bwd_flat_kernel");
    }
    int64_t bx = (n_nodes + 3) / 4;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(propagate_self_bwd_kernel, dim3((unsigned)bx, (unsigned)((d_out + 63) / 64)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), p);
    return hip_check("propagate_self_bwd_kernel");
}

extern "C" int gsn_csr_build_graphs_hip(int64_t n_graphs, const int64_t *node_ptr, const int64_t *edge_ptr, int64_t n_nodes,
                                        int64_t n_edges, int64_t max_nodes, int64_t max_edges, const int64_t *index,
                                        const int64_t *other, int32_t *seg_ptr, int32_t *perm, int32_t *sorted_target,
                                        int32_t *sorted_other, int32_t *status, void *stream) {
    if (sorted_other && !other) return set_error(GSN_E_INVALID, "gsn_csr_build_graphs_hip: sorted_other needs `other`");
    if (n_graphs < 1 || !node_ptr || !edge_ptr || n_nodes < 0 || n_edges < 0 || !seg_ptr || !status || (n_edges > 0 && (!index || !perm)))
        return set_error(GSN_E_INVALID, "gsn_csr_build_graphs_hip: bad argument");
    if (n_edges >= (int64_t)1 << 31 || n_nodes >= ((int64_t)1 << 31) - 1 || n_graphs >= (int64_t)1 << 31)
        return set_error(GSN_E_UNSUPPORTED, "gsn_csr_build_graphs_hip: more than 2^31 edges, vertices or graphs");
    if (max_nodes < 1) max_nodes = 1;
    if (max_edges < 1) max_edges = 1;
    const int64_t lds = (2 * (max_nodes + 1) + 2 * max_edges) * (int64_t)sizeof(int32_t);
    if (lds > 64 * 1024)
        return set_error(GSN_E_UNSUPPORTED, "gsn_csr_build_graphs_hip: graphs of up to %lld vertices / %lld columns need %lld B of LDS (limit 64 KiB); use gsn_csr_build_hip",
                         (long long)max_nodes, (long long)max_edges, (long long)lds);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    static DeviceOnce lds_set;
    const int lds_dev = current_device();
    if (!lds_set.done(lds_dev)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&csr_graphs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess)
            return set_error(GSN_E_HIP, "gsn_csr_build_graphs_hip: cannot raise the LDS limit of csr_graphs_kernel");
        lds_set.mark(lds_dev);
    }
    hipLaunchKernelGGL(csr_graphs_kernel, dim3((unsigned)n_graphs), dim3(64), (size_t)lds, st, node_ptr, edge_ptr, (int)n_graphs, n_nodes, n_edges,
                       (int)max_nodes, (int)max_edges, index, other, seg_ptr, perm, sorted_target, sorted_other, status);
    return hip_check("gsn_csr_build_graphs_hip");
}

template <int MAXC, int NZ4>
static int launch_split_sum(const SplitSumArgs &p, hipStream_t st) {
    int64_t blocks = ((p.n_nodes + 1) / 2 + 3) / 4;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((edge_split_sum_kernel<MAXC, NZ4>), dim3((unsigned)blocks), dim3(256), 0, st, p);
    return hip_check("edge_split_sum_kernel");
}

extern "C" int gsn_edge_split_sum_hip(int64_t n_nodes, int64_t n_edges, const int32_t *seg_ptr, const int32_t *sorted_src,
                                      const int32_t *perm, const float *p_i, const float *p_j, int64_t pitch, const float *z0,
                                      int64_t w0, const float *z1, int64_t w1, const float *wz_t, int64_t d, int act, float *out,
                                      void *stream) {
    if (n_nodes < 0 || n_edges < 0 || !seg_ptr || !p_i || !p_j || !out || (n_edges > 0 && !sorted_src))
        return set_error(GSN_E_INVALID, "gsn_edge_split_sum_hip: bad argument");
    if (d < 4 || (d & 3) || d > 256 || pitch < d || (pitch & 3)) return set_error(GSN_E_UNSUPPORTED, "gsn_edge_split_sum_hip: d must be a multiple of 4 up to 256 (pitch a multiple of 4)");
    if (w0 < 0 || w1 < 0 || (w0 & 3) || (w1 & 3) || w0 + w1 > 16 || (w0 > 0 && !z0) || (w1 > 0 && !z1) || (w0 == 0 && w1 > 0))
        return set_error(GSN_E_UNSUPPORTED, "gsn_edge_split_sum_hip: per-edge blocks must be multiples of 4 wide, 16 columns in total at most");
    if (w0 + w1 > 0 && (!wz_t || !perm)) return set_error(GSN_E_INVALID, "gsn_edge_split_sum_hip: per-edge blocks need W_z^T and perm");
    if (act != 0 && act != 1) return set_error(GSN_E_UNSUPPORTED, "gsn_edge_split_sum_hip: identity / relu only");
    if ((reinterpret_cast<uintptr_t>(p_i) | reinterpret_cast<uintptr_t>(p_j) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(wz_t) |
         reinterpret_cast<uintptr_t>(z0) | reinterpret_cast<uintptr_t>(z1)) & 15)
        return set_error(GSN_E_INVALID, "gsn_edge_split_sum_hip: pointers must be 16-byte aligned");
    if (n_nodes == 0) return GSN_OK;
    SplitSumArgs p{};
    p.n_nodes = n_nodes; p.seg_ptr = seg_ptr; p.sorted_src = sorted_src; p.perm = perm; p.tb = p_i; p.a = p_j; p.pitch = pitch;
    p.z0 = z0; p.z1 = z1; p.w0 = (int)w0; p.w1 = (int)w1; p.wz = wz_t; p.d = (int)d; p.act = act; p.out = out;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nz4 = (int)((w0 + w1) >> 2);
    const bool wide = d > 128;
    if (wide && nz4 > 2) return set_error(GSN_E_UNSUPPORTED, "gsn_edge_split_sum_hip: d > 128 with more than 8 per-edge columns");
    if (!wide) {
        switch (nz4) {
            case 0: return launch_split_sum<1, 0>(p, st);
            case 1: return launch_split_sum<1, 1>(p, st);
            case 2: return launch_split_sum<1, 2>(p, st);
            case 3: return launch_split_sum<1, 3>(p, st);
            default: return launch_split_sum<1, 4>(p, st);
        }
    }
    switch (nz4) {
        case 0: return launch_split_sum<2, 0>(p, st);
        case 1: return launch_split_sum<2, 1>(p, st);
        default: return launch_split_sum<2, 2>(p, st);
    }
}
