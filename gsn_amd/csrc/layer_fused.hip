// One `general` message-passing layer in ONE launch (GSN_sparse.py:93-176 / GSN_edge_sparse.py:82-170 and the MPNN twins):
//
//     r_e  = act_e( bn_e( cat(x_i, x_j, ids.., e) W1^T + b1 ) )              per edge      (edge stage, K_e <= 80)
//     S_v  = sum_{e -> v} r_e                                                per node      (torch.sparse.sum, :140-143)
//     h_v  = act_0( bn_0( [x_v | S_v | deg_v] W0'^T + b0 ) )                 per node      (W0' = [W3x | W3a W2 | W3a b2], layers.py)
//     out_v = act_1( bn_1( h_v W1'^T + b1' ) )                               per node
//
// r_e, S_v and h_v never touch HBM: every input is read once, the output written once (B_alg).
//
// Work decomposition.  A PyG batch is a disjoint union and the rows of the target-sorted CSR (gsn_csr_build_hip) are
// node-contiguous, so a workgroup owns a contiguous NODE range and walks it in tiles of <= 32 nodes whose in-edges it
// processes in chunks of <= 64 rows (the tile takes as many nodes as fit a whole number of chunks: ZINC-shaped graphs give
// ~31 nodes / 64 edges per tile, one chunk; a hub or a dense graph gives several chunks per tile).  Group S0 runs the tile
// iterator over seg_ptr (one 33-entry window load per tile, fetched a tile ahead) and publishes every chunk descriptor four
// steps ahead through LDS (three packed words each), so every scheduling decision is a wave-uniform scalar read; group S1, which
// must never wait for a load behind its output stores, follows a four-word record per step.
//
// Roles (12 waves, 3 per SIMD; a wave keeps ONE stage's weights in registers for the whole kernel).  Every step has two
// phases with one LDS-only barrier after each, and every group alternates a matrix phase with a staging phase, so that the
// matrix pipe and the vector pipe are both busy in both phases; every LDS buffer is written in one phase and read in the other
// (only H, which crosses from phase 2 to the next step's phase 2, is double):
//                 phase 1                                              phase 2
//   group E  (waves 0-3)   edge stage of chunk i: IN_E -> Y (fp32)      gathered rows -> planes IN_E of chunk i + 1;
//                          (+ issues the gathers of chunk i + 1 and     the x rows -> XR (raw; group S0 stages them next step)
//                          the x-row loads of a completing tile)
//   group S0 (waves 4-7)   tile iterator; [S | x | deg] of the         node stage 0: IN_N -> H[i & 1] (fp32);  per node 0..15 of the
//                          finished tile -> IN_N                        tile, sum of its Y rows in row order -> S (no atomics)
//   group S1 (waves 8-11)  node stage 1 of the tile staged last step:   H[(i - 1) & 1] -> planes MID;  the sums of nodes 16..31
//                          MID -> out rows (range-checked buffer stores)
// No register spill inside the loops (a scratch reload waits, in order, behind every load / store in flight): uniform scale
// factors are read back from an LDS table at the point of use, group S0 holds no load over its matrix phase (group E fetches the
// x rows), descriptors are packed.  WG > 0 instantiations know the three widths at compile time (d = 128 / 64).
//
// Matrix arithmetic: fp16x3.  Both operands are split into two fp16 planes (x = x_h + x_l, 11 + 11 significant bits, round
// to nearest) after an exact power-of-two scaling that puts the largest magnitude of every A row (and of each stage's weight
// matrix) into [2^14, 2^15); x w ~ x_h w_h + x_h w_l + x_l w_h with fp32 accumulation inside v_mfma_f32_32x32x16_f16; the
// dropped x_l w_l is < 2^-22 |x w|.  scripts/micro/bf16x6_check.hip: max error vs fp64 2.0e-7 of max|C| at K = 160 (an fp32
// FMA loop: 3.1e-7; the six bf16 plane products of chain_*_bf16.hip: 2.9e-7) at HALF the matrix work of the bf16 scheme.
// The accumulator is un-scaled in the epilogue (one fma with the row's inverse scale).  Rows that are exactly representable
// in fp16 without scaling (one-hot / small-integer encodings: every layer-0 input of the reference models) need no row
// maximum and no low plane: a chunk made only of such rows runs two products instead of three.
// Non-finite values: an edge row, node row or hidden row that holds an Inf or a NaN gives NaN in ALL of that row's outputs
// (and so, through the per-node sums, in the rows that aggregate it); fp32 arithmetic would keep a signed Inf where no Inf - Inf
// or 0 * Inf occurs.  An Inf / NaN weight or folded BatchNorm factor makes every output row NaN.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "chain_common.h"
#include "layer_rr.h"
#include "layer_w.h"
#include "layer_g.h"

namespace gsn {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 lf_h2 __attribute__((ext_vector_type(2)));
typedef float lf_f2 __attribute__((ext_vector_type(2)));
typedef unsigned lf_u4 __attribute__((ext_vector_type(4)));
typedef unsigned lf_u2 __attribute__((ext_vector_type(2)));

constexpr int LF_TE = 64;     // edge rows per chunk
constexpr int LF_TN = 32;     // nodes per tile
constexpr int LF_MAXB = 6;    // edge-stage input blocks
constexpr int LF_PY = 132;    // fp32 row pitch of Y / S_acc / H in floats (rows 16-byte aligned, banks staggered by 4)
constexpr int LF_SEGW = 36;   // ints per seg-window slot (33 used)
constexpr int LF_NSLOT = 8;   // seg-window slots: tiles between formation (three chunks ahead) and the node stage's staging
constexpr int LF_NXJ = 2;     // [x | deg] float4 chunks a stager thread may own (d_x <= 60)

struct LfStage {
    const float *W, *bias, *bn_mean, *bn_scale, *bn_shift;
    int k_total, n_out, act;
};

struct LfArgs {
    int n_nodes, n_edges;
    const int32_t *seg_ptr;                 // [n_nodes + 1] target-sorted CSR
    int e_nblocks;
    const float *e_data[LF_MAXB];           // edge-stage blocks, row r of the sorted order reads row e_idx[b][r] of e_data[b]
    const int32_t *e_idx[LF_MAXB];
    int e_width[LF_MAXB];
    LfStage e, s0, s1;
    const float *x;                         // [n_nodes][d_x] first block of the node stage
    int d_x;
    float *out;                             // [n_nodes][s1.n_out]
    // the three stages' weights as this kernel's register fragments (gsn_layer_fused_prepare_hip): [0..11] per-wave maxima of
    // |W * bn_scale| (the stage scales), then per stage [wave][k-step][plane h, l][lane] 16-byte fragments
    const unsigned *prep;
};
constexpr int LF_PREP_HDR = 16;             // words in front of the fragments

// One chunk of one tile; every field wave-uniform.  Packed into three words (the descriptors of four chunks in flight, the
// three tiles of the node pipeline and the iterator all live in scalar registers: unpacked they spill into vector lanes).
struct LfDesc {
    int m0, e0, pk;                         // pk: valid | first << 1 | last << 2 | slot << 3 | nn << 6 | ne << 12
    __device__ __forceinline__ int valid() const { return pk & 1; }
    __device__ __forceinline__ int first() const { return (pk >> 1) & 1; }
    __device__ __forceinline__ int last() const { return (pk >> 2) & 1; }
    __device__ __forceinline__ int slot() const { return (pk >> 3) & 7; }
    __device__ __forceinline__ int nn() const { return (pk >> 6) & 63; }
    __device__ __forceinline__ int ne() const { return (pk >> 12) & 127; }
    __device__ __forceinline__ int completes() const { return (pk & 5) == 5; }        // valid && last: the tile's sums are finished in this step
};
__device__ __forceinline__ LfDesc lf_desc_none() { LfDesc d; d.m0 = 0; d.e0 = 0; d.pk = 0; return d; }
static_assert(LF_TN < 64 && LF_TE < 128 && LF_NSLOT <= 8, "descriptor field widths");

struct LfIter {
    const int32_t *seg;
    int n_nodes, m_next, m_end;
    int m0, nn, eb, ee, ec, pending, slot;
    int win;                                // lane l: seg_ptr[m_next + l] (prefetched window of the NEXT tile)
};

__device__ __forceinline__ void lf_iter_load(LfIter &it, int lane) {
    int idx = it.m_next + lane;
    idx = idx < it.n_nodes ? idx : it.n_nodes;
    it.win = it.seg[idx];
}

// next chunk in node order; forms a new tile from the prefetched window when the current one is exhausted
__device__ __forceinline__ LfDesc lf_iter_next(LfIter &it, int lane, int *segl, bool writer) {
    LfDesc d = lf_desc_none();
    if (!(it.pending || it.ec < it.ee)) {
        if (it.m_next >= it.m_end) return d;
        int nmax = it.m_end - it.m_next;
        nmax = nmax < LF_TN ? nmax : LF_TN;
        const int w0 = __builtin_amdgcn_readfirstlane(it.win);
        const int cnt = it.win - w0;                                    // lane l: in-edges of the tile's first l nodes
        const int ne_all = __builtin_amdgcn_readlane(cnt, nmax);
        int nn = nmax;
        if (ne_all > LF_TE) {
            const int cap = ne_all / LF_TE * LF_TE;                     // a whole number of chunks
            const unsigned long long ok = __ballot(lane <= nmax && cnt <= cap);
            nn = __popcll(ok) - 1;                                      // cnt is monotone in the lane; lane 0 always passes
            nn = nn < 1 ? 1 : nn;                                       // a single node with more than `cap` edges
        }
        nn = __builtin_amdgcn_readfirstlane(nn);
        it.m0 = it.m_next; it.nn = nn; it.eb = w0; it.ee = __builtin_amdgcn_readlane(it.win, nn);
        it.ec = it.eb; it.pending = 1; it.slot = (it.slot + 1) & (LF_NSLOT - 1);
        if (writer && lane <= LF_TN) segl[it.slot * LF_SEGW + lane] = it.win;
        it.m_next += nn;
        lf_iter_load(it, lane);
    }
    d.m0 = it.m0; d.e0 = it.ec;
    const int left = it.ee - it.ec;
    const int ne = left < LF_TE ? left : LF_TE;
    d.pk = 1 | ((it.ec == it.eb) << 1) | ((it.ec + LF_TE >= it.ee) << 2) | (it.slot << 3) | (it.nn << 6) | (ne << 12);
    it.ec += LF_TE; it.pending = 0;
    return d;
}

// power-of-two scale that puts a magnitude with these (sign-less) float bits into [2^14, 2^15), and its inverse.  The
// biased exponent is clamped to [15, 254] so that both factors stay normal floats.
__device__ __forceinline__ void lf_scale(unsigned maxbits, float &scale, float &inv) {
    int e = (int)(maxbits >> 23);
    e = e < 15 ? 15 : (e > 254 ? 254 : e);
    scale = __uint_as_float((unsigned)(268 - e) << 23);
    inv = __uint_as_float((unsigned)(e - 14) << 23);
}

// two floats -> packed fp16 high parts, packed fp16 low parts (round to nearest), and the fp32 residual bits
__device__ __forceinline__ void lf_split2(lf_f2 v, unsigned &hi, unsigned &lo, unsigned &resbits) {
    const lf_h2 h = __builtin_convertvector(v, lf_h2);
    const lf_f2 r = v - __builtin_convertvector(h, lf_f2);
    const lf_h2 l = __builtin_convertvector(r, lf_h2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
    resbits = __float_as_uint(r.x) | __float_as_uint(r.y);
}

__device__ __forceinline__ unsigned lf_absmax4(unsigned m, float4 v) {
    const unsigned a = __float_as_uint(v.x) & 0x7fffffffu, b = __float_as_uint(v.y) & 0x7fffffffu;
    const unsigned c = __float_as_uint(v.z) & 0x7fffffffu, d = __float_as_uint(v.w) & 0x7fffffffu;
    m = max(max(m, a), b);
    return max(max(m, c), d);
}

// reductions over the 8 consecutive lanes that share one staged row
__device__ __forceinline__ unsigned lf_or8(unsigned v) {
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true);   // row_half_mirror
    return v;
}
__device__ __forceinline__ unsigned lf_max8(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true));
    return v;
}

// row of NCH float4 chunks held by 8 lanes -> scaled fp16 planes at `dst` (+ plane stride); chunk j is stored only where
// wr[j] (a per-lane predicate on the store alone: the arithmetic is branch-free).  Returns the inverse row scale -- NaN, with
// `nonfinite` set, when the row holds an Inf or a NaN: the epilogue of that tile then writes NaN rows (see lf_epilogue).
template <int NCH>
__device__ __forceinline__ float lf_split_row_scaled(const float4 (&v)[NCH], const bool (&wr)[NCH], _Float16 *dst, int plane_halfs, const int (&koff)[NCH], bool &nonfinite) {
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < NCH; ++j) m = lf_absmax4(m, v[j]);
    m = lf_max8(m);
    float rs, inv;
    lf_scale(m, rs, inv);
    if (m >= 0x7f800000u) { inv = __uint_as_float(0x7fc00000u); nonfinite = true; }   // Inf / NaN in the row: its outputs become NaN
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        unsigned h0, l0, h1, l1, rb;
        lf_split2(lf_f2{v[j].x * rs, v[j].y * rs}, h0, l0, rb);
        lf_split2(lf_f2{v[j].z * rs, v[j].w * rs}, h1, l1, rb);
        if (wr[j]) {
            *reinterpret_cast<lf_u2 *>(dst + koff[j]) = lf_u2{h0, h1};
            *reinterpret_cast<lf_u2 *>(dst + plane_halfs + koff[j]) = lf_u2{l0, l1};
        }
    }
    return inv;
}

// weights of one lane (output column `col`, plane column k' = 16 s + 8 lh + e), BN scale folded in, times the stage's
// power-of-two scale.  Plane column k' holds original weight column  k' < h1 ? dx + k' : (k' < h1 + dx ? k' - h1 : k')
// (node stage 0 keeps its input planes as [S | x | deg]; h1 = dx = 0: identity); the first h1 plane columns are multiplied by
// sfac (the S rows arrive times the edge stage's weight scale).
template <int NK>
__device__ __forceinline__ void lf_weight_planes(const LfStage &st, int col, bool cok, int lh, float bnscale, float wscale, int h1, int dx, float sfac, lf_u4 *Bh, lf_u4 *Bl) {
#pragma unroll
    for (int s = 0; s < NK; ++s) {
        unsigned h[4], l[4], rb;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float wv[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int kp = 16 * s + 8 * lh + 2 * q + e;
                const int k = kp < h1 ? dx + kp : (kp < h1 + dx ? kp - h1 : kp);
                const bool ok = kp < st.k_total && cok;
                const float w0 = st.W[ok ? (int64_t)col * st.k_total + k : 0];
                wv[e] = ok ? w0 * bnscale * wscale * (kp < h1 ? sfac : 1.f) : 0.f;     // (sfac: exact power of two)
            }
            lf_split2(lf_f2{wv[0], wv[1]}, h[q], l[q], rb);
        }
        Bh[s] = lf_u4{h[0], h[1], h[2], h[3]};
        Bl[s] = lf_u4{l[0], l[1], l[2], l[3]};
    }
}

// largest |W[col][k] * bnscale| of this lane's column
__device__ __forceinline__ unsigned lf_weight_absmax(const LfStage &st, int col, bool cok, float bnscale) {
    unsigned m = 0;
    if (cok)
        for (int k = 0; k < st.k_total; ++k) m = max(m, __float_as_uint(st.W[(int64_t)col * st.k_total + k] * bnscale) & 0x7fffffffu);
    return m;
}

#define LF_MF(A, B) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), acc, 0, 0, 0)

// acc[32 x 32] = A[32 rows][16 NK] (planes at ap, ap + plane) x this wave's weight planes, three plane products per k-step.
// Two independent accumulator chains over the two halves of K, issued alternately: a dependent MFMA that does not follow its
// predecessor back to back loses the forwarding path (~40 cycles, scripts/micro/valu_rates.hip), and with the LDS reads of the
// next fragments in between it never does; with two chains no MFMA waits for the one issued just before it.
template <int NK>
__device__ __forceinline__ f32x16 lf_mma_k2(const _Float16 *ap, int plane, const lf_u4 *Bh, const lf_u4 *Bl) {
    static_assert(NK % 2 == 0, "two chains over the halves of K");
    constexpr int H = NK / 2;
    f32x16 acc, acb;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acb[r] = 0.f; }
    // The low-plane fragments feed ONE product each, issued first: their registers are reloaded for the next k-step right after
    // it (the five other products of the step cover the LDS latency); only the high-plane fragments are double-buffered.
    lf_u4 ha[2], hb[2], la, lb;
    ha[0] = *reinterpret_cast<const lf_u4 *>(ap);
    la = *reinterpret_cast<const lf_u4 *>(ap + plane);
    hb[0] = *reinterpret_cast<const lf_u4 *>(ap + 16 * H);
    lb = *reinterpret_cast<const lf_u4 *>(ap + plane + 16 * H);
#pragma unroll
    for (int s = 0; s < H; ++s) {
        const int c = s & 1, n = c ^ 1;
        LF_MF(la, Bh[s]);
        acb = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, lb), __builtin_bit_cast(f16x8, Bh[s + H]), acb, 0, 0, 0);
        if (s + 1 < H) {
            la = *reinterpret_cast<const lf_u4 *>(ap + plane + 16 * (s + 1));
            lb = *reinterpret_cast<const lf_u4 *>(ap + plane + 16 * (s + 1 + H));
            ha[n] = *reinterpret_cast<const lf_u4 *>(ap + 16 * (s + 1));
            hb[n] = *reinterpret_cast<const lf_u4 *>(ap + 16 * (s + 1 + H));
        }
        LF_MF(ha[c], Bl[s]);
        acb = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, hb[c]), __builtin_bit_cast(f16x8, Bl[s + H]), acb, 0, 0, 0);
        LF_MF(ha[c], Bh[s]);
        acb = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, hb[c]), __builtin_bit_cast(f16x8, Bh[s + H]), acb, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += acb[r];
    return acc;
}

// one 32-row tile of the edge stage: a single accumulator chain (the edge group's register budget also holds the gathered
// rows of the next chunk), fragments of k-step s + 1 read before the products of step s are issued
template <int NK, bool THREE>
__device__ __forceinline__ f32x16 lf_mma_tile(const _Float16 *ap, int plane, const lf_u4 *Bh, const lf_u4 *Bl, float init) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = init;
    lf_u4 ha[2], la[2];
    ha[0] = *reinterpret_cast<const lf_u4 *>(ap);
    la[0] = THREE ? *reinterpret_cast<const lf_u4 *>(ap + plane) : ha[0];
#pragma unroll
    for (int s = 0; s < NK; ++s) {
        const int c = s & 1, n = c ^ 1;
        if (s + 1 < NK) {
            ha[n] = *reinterpret_cast<const lf_u4 *>(ap + 16 * (s + 1));
            la[n] = THREE ? *reinterpret_cast<const lf_u4 *>(ap + plane + 16 * (s + 1)) : ha[n];
        }
        if (THREE) LF_MF(la[c], Bh[s]);
        LF_MF(ha[c], Bl[s]);
        LF_MF(ha[c], Bh[s]);
    }
    return acc;
}

// epilogue of a 32 x 32 accumulator tile: y = max(acc * comb[row] + c0, lo) for this lane's 16 rows; f(g, r, y) for row
// 4 lh + 8 g + r of the 32-row tile.  comb: this tile's 32 inverse scales (LDS).  NANROWS (tiles whose stager met a non-finite value,
// rare): rows whose inverse scale is NaN come out as NaN -- the plain max would drop the NaN (IEEE maxNum semantics).
template <bool NANROWS, typename F>
__device__ __forceinline__ void lf_epilogue(const f32x16 &acc, const float *comb, int lh, float c0, float lo, F f) {
    const float *cp = comb + 4 * lh;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 cm = *reinterpret_cast<const float4 *>(cp + 8 * g);
        const float cv[4] = {cm.x, cm.y, cm.z, cm.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float y = __builtin_amdgcn_fmed3f(fmaf(acc[4 * g + r], cv[r], c0), lo, INFINITY);   // = max(., lo)
            if (NANROWS) y = cv[r] != cv[r] ? cv[r] : y;
            f(g, r, y);
        }
    }
}

// Per-node sums of a chunk's activated rows (group E left them in Y times its weight scale), in row order: deterministic, no
// atomics.  256 threads handle 16 nodes (node_base ..): 16 lanes per node, float4 chunks l16 and l16 + 16 of the row.  A tile's
// sums accumulate in S over its chunks; the last chunk divides the weight scale out.
template <int WG>
__device__ __forceinline__ void lf_node_sums(const float *ytile, float *sacc, int w0, int w1, int node_base, int u, int cs_rt, int nn, int e0,
                                             bool first, bool last, float inv_w_e) {
    const int cs = WG ? 8 * WG : cs_rt;
    const int node = node_base + (u >> 4), l16 = u & 15;
    int a0 = 0, a1 = 0;
    if (node < nn) { a0 = w0; a1 = w1; }          // w0, w1 = seg window entries node, node + 1 (read by the caller, early)
    const int lo = (a0 > e0 ? a0 : e0) - e0, hi = (a1 < e0 + LF_TE ? a1 : e0 + LF_TE) - e0;
    const bool on0 = l16 < cs, on1 = l16 + 16 < cs;
    float *sp = sacc + node * LF_PY + 4 * l16;
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
    if (!first) {
        if (on0) v0 = *reinterpret_cast<const float4 *>(sp);
        if (on1) v1 = *reinterpret_cast<const float4 *>(sp + 64);
    }
    for (int r = lo; r < hi; ++r) {
        const float *y = ytile + r * LF_PY + 4 * l16;
        float4 y0 = make_float4(0.f, 0.f, 0.f, 0.f), y1 = y0;
        if (on0) y0 = *reinterpret_cast<const float4 *>(y);
        if (on1) y1 = *reinterpret_cast<const float4 *>(y + 64);
        v0.x += y0.x; v0.y += y0.y; v0.z += y0.z; v0.w += y0.w;
        v1.x += y1.x; v1.y += y1.y; v1.z += y1.z; v1.w += y1.w;
    }
    if (last) {
        v0.x *= inv_w_e; v0.y *= inv_w_e; v0.z *= inv_w_e; v0.w *= inv_w_e;
        v1.x *= inv_w_e; v1.y *= inv_w_e; v1.z *= inv_w_e; v1.w *= inv_w_e;
    }
    if (on0) *reinterpret_cast<float4 *>(sp) = v0;
    if (on1) *reinterpret_cast<float4 *>(sp + 64) = v1;
}

// a chunk descriptor through LDS (group S0 runs the tile iterator and publishes every chunk four steps ahead)
__device__ __forceinline__ void lf_desc_put(int *ring, const LfDesc &d) {
    *reinterpret_cast<int4 *>(ring) = make_int4(d.m0, d.e0, d.pk, 0);
}
__device__ __forceinline__ LfDesc lf_desc_get(const int *ring) {
    const int4 v = *reinterpret_cast<const int4 *>(ring);
    LfDesc d;
    d.m0 = __builtin_amdgcn_readfirstlane(v.x); d.e0 = __builtin_amdgcn_readfirstlane(v.y); d.pk = __builtin_amdgcn_readfirstlane(v.z);
    return d;
}

// WG > 0: all three widths (edge stage, node stage 0, node stage 1) are 32 WG, known at compile time -- the shape of every
// reference configuration (d = 64, 128); WG = 0: widths read from the arguments (any multiple of 32 / 4).
template <int NKE, int NK0, int NK1, int WG, bool PROF>
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3))) void layer_fused_kernel(LfArgs a, unsigned long long *prof, int prio) {
    auto clk = [&]() -> unsigned long long {
        if (!PROF) return 0;
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long v = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
        return v;
    };
    // diagnostic build only: bits 8.. of `prio` switch parts of the work off (GSN_FUSED_ABLATE; results are then garbage) to see
    // what the step time is sensitive to
    const int abl = PROF ? (prio >> 8) : 0;
    prio &= 0xff;
    unsigned long long pc[6] = {0, 0, 0, 0, 0, 0};     // phase-1 work, barrier 1, phase-2 work, barrier 2, bookkeeping, steps
    unsigned long long pe[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // per-group detail, see lf_launch
    constexpr int KE = 16 * NKE, KPE = KE + 8, PLE = LF_TE * KPE;        // halfs
    constexpr int K0 = 16 * NK0, KP0 = K0 + 8, PL0 = LF_TN * KP0;
    constexpr int K1 = 16 * NK1, KP1 = K1 + 8, PL1 = LF_TN * KP1;
    constexpr int NCHE = (KE / 4 + 7) / 8;                               // float4 chunks per stager thread and row
    constexpr int NCH1 = K1 / 32;
    constexpr int NJS = WG ? WG : 4, NJX = LF_NXJ, NCH0 = NJS + NJX;     // node rows: <= 4 chunks of S and <= LF_NXJ of [x | deg] per thread
    static_assert(K1 % 32 == 0, "stage-1 input planes in whole 32-column groups");
    static_assert(WG == 0 || (2 * WG == NK1 && 2 * WG < NK0), "compile-time widths: K1 = 32 WG, K0 > 32 WG");
    constexpr int OFF_INE = 0, SZ_INE = 2 * PLE * 2;
    constexpr int OFF_Y = OFF_INE + SZ_INE, SZ_Y = LF_TE * LF_PY * 4;
    constexpr int OFF_SACC = OFF_Y + SZ_Y, SZ_SACC = LF_TN * LF_PY * 4;       // per-node sums of the tile being reduced
    constexpr int OFF_INN = OFF_SACC + SZ_SACC, SZ_INN = 2 * PL0 * 2;
    constexpr int OFF_H = OFF_INN + SZ_INN, SZ_H = 2 * LF_TN * LF_PY * 4;       // hidden rows of two tiles (step parity)
    constexpr int OFF_MID = OFF_H + SZ_H, SZ_MID = 2 * PL1 * 2;
    constexpr int OFF_XR = OFF_MID + SZ_MID, SZ_XR = LF_NXJ * LF_TN * 32 * 4;  // raw x rows of the tile whose sums complete in this step
    constexpr int OFF_TAB = OFF_XR + SZ_XR;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16 *in_e = reinterpret_cast<_Float16 *>(smem + OFF_INE);
    float *ytile = reinterpret_cast<float *>(smem + OFF_Y);
    float *sacc = reinterpret_cast<float *>(smem + OFF_SACC);
    _Float16 *in_n = reinterpret_cast<_Float16 *>(smem + OFF_INN);
    float *htile = reinterpret_cast<float *>(smem + OFF_H);
    _Float16 *mid = reinterpret_cast<_Float16 *>(smem + OFF_MID);
    float *xrl = reinterpret_cast<float *>(smem + OFF_XR);                // [LF_NXJ][LF_TN][32]: chunk q8 + 8 j of row r8 at ((j * LF_TN + r8) * 8 + q8) * 4
    int *rsrc = reinterpret_cast<int *>(smem + OFF_TAB);                 // [LF_MAXB][LF_TE]
    float *comb_e = reinterpret_cast<float *>(rsrc + LF_MAXB * LF_TE);   // [LF_TE]  inverse row scale x inverse weight scale
    float *comb_n = comb_e + LF_TE;                                      // [LF_TN]
    float *comb_h = comb_n + LF_TN;                                      // [LF_TN]
    int *segl = reinterpret_cast<int *>(comb_h + LF_TN);                 // [LF_NSLOT][LF_SEGW] seg_ptr windows of the tiles in flight
    int *flag_e = segl + LF_NSLOT * LF_SEGW;                             // [4]  ordinals of: a chunk with a scaled row; a chunk / node tile / hidden tile with a non-finite row
    unsigned *wmax = reinterpret_cast<unsigned *>(flag_e + 4);           // [12] per-wave weight maxima (prologue)
    int *pub = reinterpret_cast<int *>(wmax + 12);                        // [4 (+8)] step record for group S1 (published by group S0 every step)
    int *dring = pub + 12;                                                // [8][4] chunk descriptors, published four steps ahead
    float *wtab = reinterpret_cast<float *>(dring + 64);                   // [4] inverse weight scales of groups E, S0, S1 (NaN: non-finite weights).  Read
                                                                          // back into a scalar register at every use: held in a vector register over the
                                                                          // whole loop they get spilled, and a reload waits for every load / store in flight

    const int tid = threadIdx.x;
    const int grp = __builtin_amdgcn_readfirstlane(tid >> 8);            // 0: E, 1: S0, 2: S1
    const int t = tid & 255;
    const int lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int q8 = t & 7, r8 = t >> 3;                                    // stager map: 8 lanes per row, 32 rows per pass

    const unsigned long long t_entry = clk();
    for (int i = tid; i < OFF_TAB / 4; i += 768) reinterpret_cast<float *>(smem)[i] = 0.f;   // padded columns stay zero
    if (tid < 4) flag_e[tid] = -1;

    // ---- this wave's stage: folded BatchNorm, weight scale ------------------------------------------------------------
    const LfStage &st = grp == 0 ? a.e : (grp == 1 ? a.s0 : a.s1);
    const int st_n_out = WG ? 32 * WG : st.n_out;
    const int h1 = WG ? 32 * WG : a.e.n_out;                              // width of the activated edge rows = of the per-node sums
    const int col = 32 * w + li;
    const bool cok = col < st_n_out;
    const bool active = 32 * w < st_n_out;
    const float bias = (cok && st.bias) ? st.bias[col] : 0.f;
    float bnscale = 1.f, c0 = bias;
    if (cok && st.bn_scale) { bnscale = st.bn_scale[col]; c0 = (bias - st.bn_mean[col]) * bnscale + st.bn_shift[col]; }
    // ---- node range of this workgroup; group S0 runs the tile iterator (identical in its four waves) and publishes every chunk
    //      four steps ahead through LDS: chunks 0 and 1 here, before the prologue's barrier
    LfIter it;
    it.seg = a.seg_ptr; it.n_nodes = a.n_nodes;
    it.m_next = (int)((int64_t)a.n_nodes * blockIdx.x / gridDim.x);
    it.m_end = (int)((int64_t)a.n_nodes * (blockIdx.x + 1) / gridDim.x);
    it.m0 = 0; it.nn = 0; it.eb = 0; it.ee = 0; it.ec = 0; it.pending = 0; it.slot = 0; it.win = 0;
    const bool seg_writer = tid >= 256 && tid < 320;
    LfDesc d3 = lf_desc_none();                                           // (group S0 only) chunk i + 3
    LfDesc dc0 = d3;
    if (grp == 1) {
        lf_iter_load(it, lane);
        dc0 = lf_iter_next(it, lane, segl, seg_writer);
        d3 = lf_iter_next(it, lane, segl, seg_writer);
        if (tid == 256) { lf_desc_put(dring, dc0); lf_desc_put(dring + 4, d3); }
    }
    if (tid < 12) wmax[tid] = a.prep[tid];                                // per-wave weight maxima (prepared once per weight version)
    __syncthreads();
    const unsigned long long t_max = clk();
    float wscale, inv_w;
    lf_scale(max(max(wmax[4 * grp], wmax[4 * grp + 1]), max(wmax[4 * grp + 2], wmax[4 * grp + 3])), wscale, inv_w);
    wscale = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(wscale)));       // (wave-uniform: scalar registers)
    inv_w = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(inv_w)));
    const float act_lo = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(st.act == 1 ? 0.f : -INFINITY)));   // relu / identity as one max
    // an Inf / NaN weight (or folded BatchNorm factor): every output row of this stage is NaN, through the same path as a non-finite input row
    const bool w_bad = __builtin_amdgcn_readfirstlane((int)(max(max(wmax[4 * grp], wmax[4 * grp + 1]), max(wmax[4 * grp + 2], wmax[4 * grp + 3])) >= 0x7f800000u)) != 0;
    if (w_bad) inv_w = __uint_as_float(0x7fc00000u);
    if (t == 0) wtab[grp] = inv_w;                                         // (visible after the first barrier of the loop; first used after it)
    auto uniform_lds = [](const float *p) -> float { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(*p))); };
    // this wave's weight fragments: consecutive lanes read consecutive 16 bytes (the in-kernel split read 64 different rows per
    // instruction from every workgroup at once: 50 - 400 k cycles of prologue under load)
    const lf_u4 *pfrag = reinterpret_cast<const lf_u4 *>(a.prep + LF_PREP_HDR) +
                         ((grp == 0 ? 0 : (grp == 1 ? 4 * NKE : 4 * (NKE + NK0))) * 2 + (grp == 0 ? NKE : (grp == 1 ? NK0 : NK1)) * 2 * w) * 64 + lane;

    // ---- step bookkeeping (the tile iterator itself lives in group S0, see below) ------------------------------------------
    LfDesc d0 = lf_desc_none(), d1 = d0, d2 = d0;                         // chunks i, i+1, i+2 of step i
    if (grp == 0) d2 = lf_desc_get(dring);                                // chunk 0 (published before the prologue's barrier)
    if (grp == 1) d2 = dc0;
    // the edge stage's weight scale: its activated rows carry it until the per-node sums (groups S0 and S1) divide it out
    {
        float ws_e_unused, inv_w_e;
        lf_scale(max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])), ws_e_unused, inv_w_e);
        if (max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])) >= 0x7f800000u) inv_w_e = __uint_as_float(0x7fc00000u);
        if (tid == 0) wtab[3] = inv_w_e;
    }
    const int cs_e = h1 >> 2;                                              // float4 chunks of an activated row
    // tiles in the node pipeline at step i (m0 + the descriptor word of the chunk that completed the tile; pk = 0: none):
    //   ts: sums complete, group S0 stages it in phase 1 and multiplies it in phase 2 -> H[i & 1];
    //   th: its H was written in step i - 1, group S1 splits it in phase 2 -> MID;  tm: MID staged in step i - 1, S1 multiplies it in phase 1
    int ts_pk = 0, ts_m0 = 0, th_pk = 0, th_m0 = 0, tm_pk = 0, tm_m0 = 0;
    auto pk_nn = [](int pk) { return (pk >> 6) & 63; };
    auto pk_slot = [](int pk) { return (pk >> 3) & 7; };
    int step = -2;

    if (grp == 0) {
        // =============================================================================================================
        // group E
        // =============================================================================================================
        lf_u4 Bh[NKE], Bl[NKE];
#pragma unroll
        for (int q = 0; q < NKE; ++q) { Bh[q] = pfrag[(2 * q) * 64]; Bl[q] = pfrag[(2 * q + 1) * 64]; }
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the fragments are HERE -- left "in flight" into the loop, their first uses get
                                                 // a wait in every iteration, and in this group's loop that wait covers the gathers as well
        // staging map: thread -> rows r8 and r8 + 32, float4 chunks q8 + 8 j of each
        const float *gbase[NCHE];
        int gbw[NCHE], gblk[NCHE], gk[NCHE];
        bool gon[NCHE];
        unsigned gmask[NCHE];
#pragma unroll
        for (int j = 0; j < NCHE; ++j) {
            const int kc = 4 * (q8 + 8 * j);
            gon[j] = kc < a.e.k_total;                                    // a real input column: loaded
            gmask[j] = gon[j] ? 0xffffffffu : 0u;
            int blk = 0, c = kc;
#pragma unroll
            for (int b = 0; b < LF_MAXB - 1; ++b)
                if (blk == b && b + 1 < a.e_nblocks && c >= a.e_width[b]) { c -= a.e_width[b]; blk = b + 1; }
            if (!gon[j]) { blk = 0; c = 0; }
            const float *bd = a.e_data[0];
            int bw = a.e_width[0];
#pragma unroll
            for (int b = 1; b < LF_MAXB; ++b)
                if (blk == b) { bd = a.e_data[b]; bw = a.e_width[b]; }
            gbase[j] = bd + c; gbw[j] = bw; gblk[j] = blk * LF_TE; gk[j] = kc;
        }
        // row sources: thread -> tile row t & 63 of block t >> 6 (and of block 4 + (t >> 6))
        const int rs_r = lane, rs_b = w;                                 // (the block is wave-uniform: its index pointer lives in scalar registers)
        const int32_t *rs_p0 = a.e_idx[0], *rs_p1 = a.e_idx[0];
#pragma unroll
        for (int b = 0; b < LF_MAXB; ++b) {
            if (b == rs_b && b < a.e_nblocks) rs_p0 = a.e_idx[b];
            if (b == rs_b + 4 && b < a.e_nblocks) rs_p1 = a.e_idx[b];
        }
        const bool rs_on0 = rs_b < a.e_nblocks, rs_on1 = rs_b + 4 < a.e_nblocks;
        const int e_last = a.n_edges > 0 ? a.n_edges - 1 : 0;
        const float c0w = c0 * wscale;                                    // the accumulators' start value in a chunk of exact rows
        int raw0 = 0, raw1 = 0;
        bool lo_dirty[2] = {false, false};                                // (LDS starts zeroed; a thread always stages the same two tile rows)
        float4 pf[2][NCHE];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int j = 0; j < NCHE; ++j) pf[rr][j] = make_float4(0.f, 0.f, 0.f, 0.f);
        // The x rows of the node stage travel through this group as well (group S0 keeps 80 registers of weights and must not hold
        // loads in flight over its matrix phase): loaded in phase 1 of the step that completes the tile's sums, parked raw in LDS in
        // phase 2, staged by group S0 in the next step's phase 1.  Thread -> row r8, chunks q8 + 8 j.
        const int cx_e = a.d_x >> 2, njx_e = (cx_e + 1 + 7) >> 3;
        float4 xq[NJX];
#pragma unroll
        for (int j = 0; j < NJX; ++j) xq[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        (void)xq;
        float *const yp_lane = ytile + col + 4 * lh * LF_PY;              // this lane's column, row 4 lh of the tile (MFMA row map: 4 lh + 8 g + r)
        while (d0.valid() | d1.valid() | d2.valid() | ts_pk | th_pk | tm_pk | (step < 0)) {
            const unsigned long long c_0 = clk();
            const LfDesc dn = lf_desc_get(dring + 4 * ((step + 3) & 7));  // chunk i + 3 (group S0 published it a step ago)
            // ---------------- phase 1 ----------------
            const unsigned long long e_1 = clk();
            if (d2.valid() && a.n_edges > 0) {                            // row sources of chunk i + 2 (consumed in phase 2)
                const int ne2 = d2.ne();
                int er = d2.e0 + (rs_r < ne2 ? rs_r : (ne2 > 0 ? ne2 - 1 : 0));
                er = er < e_last ? er : e_last;
                if (rs_on0) raw0 = rs_p0[er];
                if (rs_on1) raw1 = rs_p1[er];
            }
            const bool e1_on = d1.valid() && d1.ne() > 0 && !(abl & 128);
            if (e1_on) {                                                  // gathers of chunk i + 1 (consumed in phase 2)
                int sr[2][NCHE];
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                    for (int j = 0; j < NCHE; ++j) sr[rr][j] = rsrc[gblk[j] + r8 + 32 * rr];
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                    for (int j = 0; j < NCHE; ++j)                        // (a chunk past k_total reads a valid address; it is masked when the
                        pf[rr][j] = *reinterpret_cast<const float4 *>(gbase[j] + (int64_t)sr[rr][j] * gbw[j]);   // row is staged: any use here would wait for the data)
            }
            const bool xon = d0.completes() != 0;
            const int d0nn = d0.nn();
            if (xon) {                                                    // (clamped addresses, masked when parked: a select or a zero written to
#pragma unroll                                                            //  these registers here would wait for the gathers just issued)
                for (int j = 0; j < NJX; ++j) {
                    const int xrow = d0.m0 + (r8 < d0nn ? r8 : d0nn - 1), xc = q8 + 8 * j < cx_e ? q8 + 8 * j : cx_e - 1;
                    if (j < njx_e) xq[j] = *reinterpret_cast<const float4 *>(a.x + (int64_t)xrow * a.d_x + 4 * xc);
                }
            }
            const unsigned long long e_2 = clk();
            if (prio) __builtin_amdgcn_s_setprio(2);
            const int d0ne = d0.ne();
            if (d0.valid() && d0ne > 0 && active && !(abl & 1)) {
                const _Float16 *ap0 = in_e + li * KPE + 8 * lh, *ap1 = ap0 + 32 * KPE;
                // The activated rows are kept TIMES the weight scale (Y' = ws * Y, an exact power of two that the per-node sums divide
                // out again): a chunk whose rows are all exact in fp16 starts its accumulators at ws * c0 and needs no multiply at all.
                const bool nanrows = flag_e[1] == step || w_bad;
                const bool three = flag_e[0] == step || nanrows;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (h == 1 && d0ne <= 32) break;
                    const _Float16 *ap = h ? ap1 : ap0;
                    float *y0 = yp_lane + 32 * h * LF_PY;
                    if (three) {
                        const f32x16 acc = lf_mma_tile<NKE, true>(ap, PLE, Bh, Bl, 0.f);
                        if (cok) {
                            if (nanrows) lf_epilogue<true>(acc, comb_e + 32 * h, lh, c0w, act_lo, [&](int g, int r, float y) { y0[(8 * g + r) * LF_PY] = y; });
                            else lf_epilogue<false>(acc, comb_e + 32 * h, lh, c0w, act_lo, [&](int g, int r, float y) { y0[(8 * g + r) * LF_PY] = y; });
                        }
                    } else {
                        const f32x16 acc = lf_mma_tile<NKE, false>(ap, PLE, Bh, Bl, c0w);
                        if (cok) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) y0[((r & 3) + 8 * (r >> 2)) * LF_PY] = __builtin_amdgcn_fmed3f(acc[r], act_lo, INFINITY);
                        }
                    }
                }
            }
            if (prio) __builtin_amdgcn_s_setprio(0);
            const unsigned long long c_1 = clk();
            lds_barrier();
            const unsigned long long c_2 = clk();
            // ---------------- phase 2 ----------------
            if (d2.valid()) {
                if (rs_on0) rsrc[rs_b * LF_TE + rs_r] = raw0;
                if (rs_on1) rsrc[(rs_b + 4) * LF_TE + rs_r] = raw1;
            }
            const unsigned long long e_3 = clk();
            if (e1_on && !(abl & 2)) {
                bool scaled_any = false, nf_any = false;
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const int row = r8 + 32 * rr;
                    _Float16 *dst = in_e + row * KPE;
                    unsigned hi[NCHE][2], bits = 0;
#pragma unroll
                    for (int j = 0; j < NCHE; ++j) {
                        unsigned rb0, rb1, lo0, lo1;
                        lf_split2(lf_f2{pf[rr][j].x, pf[rr][j].y}, hi[j][0], lo0, rb0);
                        lf_split2(lf_f2{pf[rr][j].z, pf[rr][j].w}, hi[j][1], lo1, rb1);
                        bits |= (rb0 | rb1) & gmask[j];                   // (columns past k_total hold whatever the clamped address had)
                    }
                    bits = lf_or8(bits);
                    float comb = 1.f;
                    if (bits == 0) {                                      // the whole row is exact in fp16: no scale, low plane zero
#pragma unroll
                        for (int j = 0; j < NCHE; ++j)
                            if (gon[j]) {                                 // (columns k_total .. KE are never written: they stay zero)
                                *reinterpret_cast<lf_u2 *>(dst + gk[j]) = lf_u2{hi[j][0], hi[j][1]};
                                // the low plane of this row is zero already unless this thread staged a scaled row here last time
                                if (lo_dirty[rr]) *reinterpret_cast<lf_u2 *>(dst + PLE + gk[j]) = lf_u2{0u, 0u};
                            }
                        lo_dirty[rr] = false;
                    } else {
                        lo_dirty[rr] = true;
                        float4 pz[NCHE];
#pragma unroll
                        for (int j = 0; j < NCHE; ++j) pz[j] = gon[j] ? pf[rr][j] : make_float4(0.f, 0.f, 0.f, 0.f);
                        comb = lf_split_row_scaled<NCHE>(pz, gon, dst, PLE, gk, nf_any);
                        scaled_any = true;
                    }
                    if (q8 == 0) comb_e[row] = w_bad ? __uint_as_float(0x7fc00000u) : comb;
                }
                if (scaled_any) flag_e[0] = step + 1;                     // (every writer stores the same value)
                if (nf_any) flag_e[1] = step + 1;
            }
            if (xon) {
#pragma unroll
                for (int j = 0; j < NJX; ++j)
                    if (j < njx_e) {
                        const bool on = q8 + 8 * j < cx_e && r8 < d0nn;
                        *reinterpret_cast<float4 *>(xrl + ((j * LF_TN + r8) * 8 + q8) * 4) = on ? xq[j] : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
            }
            const unsigned long long e_4 = clk();
            const unsigned long long c_3 = clk();
            lds_barrier();
            const unsigned long long c_4 = clk();
            tm_pk = th_pk; tm_m0 = th_m0;
            th_pk = ts_pk; th_m0 = ts_m0;
            ts_pk = d0.completes() ? d0.pk : 0; ts_m0 = d0.m0;
            d0 = d1; d1 = d2; d2 = dn;
            ++step;
            if (PROF) { const unsigned long long c_5 = clk(); pc[0] += c_1 - c_0; pc[1] += c_2 - c_1; pc[2] += c_3 - c_2; pc[3] += c_4 - c_3; pc[4] += c_5 - c_4; pc[5] += 1;
                        pe[0] += e_1 - c_0; pe[1] += e_2 - e_1; pe[2] += c_1 - e_2; pe[3] += e_3 - c_2; pe[4] += e_4 - e_3; pe[5] += c_3 - e_4; }
        }
        if (PROF && prof && lane == 0 && blockIdx.x == 0) {
            for (int q = 0; q < 6; ++q) prof[(tid >> 6) * 6 + q] = pc[q];
            for (int q = 0; q < 6; ++q) prof[72 + (tid >> 6) * 6 + q] = pe[q];
        }
        return;
    }

    if (grp == 1) {
        // =============================================================================================================
        // group S0.  Input planes of a node row: [S (h1 columns) | x (d_x) | deg, 0, 0, 0]; the weight planes follow that order.
        // Thread -> row r8, S chunks q8 + 8 j (j < njs) and chunks q8 + 8 j of the [x | deg] part (j < njx).
        // =============================================================================================================
        const int cs = h1 >> 2, cx = a.d_x >> 2;
        const int njs = WG ? WG : cs >> 3, njx = (cx + 1 + 7) >> 3;      // (h1 is a multiple of 32)
        lf_u4 Bh[NK0], Bl[NK0];
#pragma unroll
        for (int q = 0; q < NK0; ++q) { Bh[q] = pfrag[(2 * q) * 64]; Bl[q] = pfrag[(2 * q + 1) * 64]; }
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the fragments are HERE -- left "in flight" into the loop, their first uses get
                                                 // a wait in every iteration, and in this group's loop that wait covers the gathers as well
        const unsigned long long t_planes = PROF ? (unsigned long long)(Bh[0][0] != 0x12345u) + clk() : 0ull;
        int koff[NCH0];
        bool wr[NCH0], isdeg[NJX];
#pragma unroll
        for (int j = 0; j < NJS; ++j) { koff[j] = 4 * (q8 + 8 * j); wr[j] = j < njs; }
#pragma unroll
        for (int j = 0; j < NJX; ++j) {
            const int c = q8 + 8 * j;
            koff[NJS + j] = h1 + 4 * c;
            isdeg[j] = c == cx;
            wr[NJS + j] = c <= cx;
        }
        float *const hp_lane = htile + col + 4 * lh * LF_PY;
        while (d0.valid() | d1.valid() | d2.valid() | ts_pk | th_pk | tm_pk | (step < 0)) {
            const unsigned long long c_0 = clk();
            const LfDesc dn = lf_iter_next(it, lane, segl, seg_writer);  // chunk i + 4 (its window arrived a step ago)
            // ---------------- phase 1 ----------------
            const unsigned long long e_1 = clk();
            if (tid == 256) {
                lf_desc_put(dring + 4 * ((step + 4) & 7), dn);            // group E reads it at the top of the next step
                // what group S1 needs to follow the tiles and to sum its half of this chunk's nodes (it never loads seg_ptr: a
                // wait for such a load would drain its output stores): the tile staged now, "more steps follow", the current chunk
                const int more = d1.valid() | d0.completes();
                *reinterpret_cast<int4 *>(pub) = make_int4(ts_pk, ts_m0, d0.pk | (more << 20), d0.e0);
            }
            const unsigned long long e_2 = clk();
            if (ts_pk && !(abl & 4)) {                                    // [S | x | deg] of the tile whose sums were finished in the last step
                const int ts_nn = pk_nn(ts_pk);
                const int *sw = segl + pk_slot(ts_pk) * LF_SEGW + r8;
                int a0 = 0, a1 = 0;
                if (r8 < ts_nn) { a0 = sw[0]; a1 = sw[1]; }
                const float *sp = sacc + r8 * LF_PY;
                float4 v[NCH0];
#pragma unroll
                for (int j = 0; j < NJS; ++j) {
                    v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (j < njs) v[j] = *reinterpret_cast<const float4 *>(sp + koff[j]);
                }
                const float degf = (float)(a1 - a0);
#pragma unroll
                for (int j = 0; j < NJX; ++j) {
                    v[NJS + j] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (j < njx) v[NJS + j] = *reinterpret_cast<const float4 *>(xrl + ((j * LF_TN + r8) * 8 + q8) * 4);   // (group E parked it; zero past d_x)
                    if (isdeg[j]) v[NJS + j] = make_float4(degf, 0.f, 0.f, 0.f);
                }
                bool nf = false;
                const float inv = lf_split_row_scaled<NCH0>(v, wr, in_n + r8 * KP0, PL0, koff, nf);
                if (q8 == 0) comb_n[r8] = inv * uniform_lds(wtab + 1);
                if (nf) flag_e[2] = step;
            }
            const unsigned long long e_3 = clk();
            const unsigned long long c_1 = clk();
            lds_barrier();
            const unsigned long long c_2 = clk();
            // ---------------- phase 2 ----------------
            // (this chunk's segment bounds for the per-node sums below: read now, under the matrix products)
            const int *swp = segl + d0.slot() * LF_SEGW + (t >> 4);
            const int sw0 = swp[0], sw1 = swp[1];
            if (prio) __builtin_amdgcn_s_setprio(2);
            if (ts_pk && active && !(abl & 8)) {
                const f32x16 acc = lf_mma_k2<NK0>(in_n + li * KP0 + 8 * lh, PL0, Bh, Bl);
                float *hp = hp_lane + (step & 1) * (LF_TN * LF_PY);
                if (cok) {
                    if (flag_e[2] == step || w_bad) lf_epilogue<true>(acc, comb_n, lh, c0, act_lo, [&](int g, int r, float y) { hp[(8 * g + r) * LF_PY] = y; });
                    else lf_epilogue<false>(acc, comb_n, lh, c0, act_lo, [&](int g, int r, float y) { hp[(8 * g + r) * LF_PY] = y; });
                }
            }
            if (prio) __builtin_amdgcn_s_setprio(0);
            const unsigned long long e_4 = clk();
            if (d0.valid() && !(abl & 16))                                // nodes 0 .. 15 of this chunk's tile (group S1: 16 .. 31)
                lf_node_sums<WG>(ytile, sacc, sw0, sw1, 0, t, cs_e, d0.nn(), d0.e0, d0.first() != 0, d0.last() != 0, uniform_lds(wtab + 3));
            const unsigned long long c_3 = clk();
            lds_barrier();
            const unsigned long long c_4 = clk();
            tm_pk = th_pk; tm_m0 = th_m0;
            th_pk = ts_pk; th_m0 = ts_m0;
            ts_pk = d0.completes() ? d0.pk : 0; ts_m0 = d0.m0;
            d0 = d1; d1 = d2; d2 = d3; d3 = dn;
            ++step;
            if (PROF) { const unsigned long long c_5 = clk(); pc[0] += c_1 - c_0; pc[1] += c_2 - c_1; pc[2] += c_3 - c_2; pc[3] += c_4 - c_3; pc[4] += c_5 - c_4; pc[5] += 1;
                        pe[0] += e_1 - c_0; pe[1] += e_2 - e_1; pe[2] += e_3 - e_2; pe[3] += c_1 - e_3; pe[4] += e_4 - c_2; pe[5] += c_3 - e_4; }
        }
        if (PROF && prof && lane == 0 && blockIdx.x == 0) {
            for (int q = 0; q < 6; ++q) prof[(tid >> 6) * 6 + q] = pc[q];
            for (int q = 0; q < 6; ++q) prof[72 + (tid >> 6) * 6 + q] = pe[q];
            if (tid == 256) { prof[140] = t_max - t_entry; prof[141] = t_planes - t_max; prof[142] = clk() - t_planes; }
        }
        if (PROF && prof && tid == 256 && blockIdx.x == 128) { prof[136] = t_max - t_entry; prof[137] = t_planes - t_max; prof[138] = clk() - t_planes; prof[139] = t_entry; }
        if (PROF && prof && tid == 256 && blockIdx.x == 0) prof[143] = t_entry;
        if (PROF && prof && tid == 256) { prof[144 + 2 * blockIdx.x] = t_entry; prof[145 + 2 * blockIdx.x] = clk(); }   // entry / exit time of every workgroup
        return;
    }

    // =================================================================================================================
    // group S1
    // =================================================================================================================
    lf_u4 Bh[NK1], Bl[NK1];
#pragma unroll
    for (int q = 0; q < NK1; ++q) { Bh[q] = pfrag[(2 * q) * 64]; Bl[q] = pfrag[(2 * q + 1) * 64]; }
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0), see group E
    const int njh = WG ? WG : a.s0.n_out >> 5;                            // 32-column groups of H that carry data (n_out multiple of 32)
    int koff[NCH1];
    bool wr[NCH1];
#pragma unroll
    for (int j = 0; j < NCH1; ++j) { wr[j] = j < njh; koff[j] = 4 * (q8 + 8 * j); }
    // Output rows leave through a buffer resource per tile (base = the tile's first row, extent = its nn rows): the rows past nn
    // of a short tile are dropped by the hardware's range check instead of one compare + exec mask per store, and the address of
    // a store is one per-lane byte offset fixed before the loop plus an immediate.
    const int rstride = st_n_out * 4;                                     // bytes per output row
    int voff[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) voff[g] = ((8 * g + 4 * lh) * st_n_out + col) * 4;
    // This group follows the tiles through the record group S0 publishes every step instead of running the iterator: its
    // only global memory operations are the output stores, and it never waits for them.
    int more = 1;
    while (more | th_pk | tm_pk | (step < 0)) {
        const unsigned long long c_0 = clk();
        // ---------------- phase 1 ----------------
        if (prio) __builtin_amdgcn_s_setprio(2);
        unsigned long long e_1 = c_0;
        if (tm_pk && active && !(abl & 32)) {
            const f32x16 acc = lf_mma_k2<NK1>(mid + li * KP1 + 8 * lh, PL1, Bh, Bl);
            if (PROF) { pe[5] += (unsigned long long)(acc[0] != 12345.f); e_1 = clk(); }   // (forces the products to finish before the reading)
            const __amdgpu_buffer_rsrc_t orow = __builtin_amdgcn_make_buffer_rsrc(a.out + (int64_t)tm_m0 * st_n_out, 0, pk_nn(tm_pk) * rstride, 0x00020000);
            if (cok) {
                auto put = [&](int g, int r, float y) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), orow, voff[g] + r * rstride, 0, 0); };
                if (flag_e[3] == step || w_bad) lf_epilogue<true>(acc, comb_h, lh, c0, act_lo, put);
                else lf_epilogue<false>(acc, comb_h, lh, c0, act_lo, put);
            }
        }
        if (prio) __builtin_amdgcn_s_setprio(0);
        const unsigned long long c_1 = clk();
        lds_barrier();
        const unsigned long long c_2 = clk();
        // ---------------- phase 2 ----------------
        const int4 rec = *reinterpret_cast<const int4 *>(pub);
        const int nts_pk = __builtin_amdgcn_readfirstlane(rec.x), nts_m0 = __builtin_amdgcn_readfirstlane(rec.y);
        const int c_pk = __builtin_amdgcn_readfirstlane(rec.z), c_e0 = __builtin_amdgcn_readfirstlane(rec.w);
        more = (c_pk >> 20) & 1;
        const int *swp = segl + pk_slot(c_pk) * LF_SEGW + 16 + (t >> 4);   // segment bounds of this group's nodes (16 ..) of the current chunk
        const int sw0 = swp[0], sw1 = swp[1];
        const unsigned long long e_3 = clk();
        if (th_pk && !(abl & 64)) {                                       // H of that tile was written in the last step's phase 2
            const float *hp = htile + ((step - 1) & 1) * (LF_TN * LF_PY) + r8 * LF_PY;
            float4 v[NCH1];
#pragma unroll
            for (int j = 0; j < NCH1; ++j) {
                v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (j < njh) v[j] = *reinterpret_cast<const float4 *>(hp + koff[j]);
            }
            bool nf = false;
            const float inv = lf_split_row_scaled<NCH1>(v, wr, mid + r8 * KP1, PL1, koff, nf);
            if (q8 == 0) comb_h[r8] = inv * uniform_lds(wtab + 2);
            if (nf) flag_e[3] = step + 1;
        }
        const unsigned long long e_4 = clk();
        if ((c_pk & 1) && !(abl & 16))                                    // nodes 16 .. 31 of this chunk's tile (group S0: 0 .. 15)
            lf_node_sums<WG>(ytile, sacc, sw0, sw1, 16, t, cs_e, pk_nn(c_pk), c_e0, ((c_pk >> 1) & 1) != 0, ((c_pk >> 2) & 1) != 0, uniform_lds(wtab + 3));
        const unsigned long long c_3 = clk();
        lds_barrier();
        const unsigned long long c_4 = clk();
        tm_pk = th_pk; tm_m0 = th_m0;
        th_pk = nts_pk; th_m0 = nts_m0;
        ++step;
        if (PROF) { const unsigned long long c_5 = clk(); pc[0] += c_1 - c_0; pc[1] += c_2 - c_1; pc[2] += c_3 - c_2; pc[3] += c_4 - c_3; pc[4] += c_5 - c_4; pc[5] += 1;
                    pe[0] += e_1 - c_0; pe[1] += c_1 - e_1; pe[2] += e_3 - c_2; pe[3] += e_4 - e_3; pe[4] += c_3 - e_4; }
    }
    if (PROF && prof && lane == 0 && blockIdx.x == 0) {
        for (int q = 0; q < 6; ++q) prof[(tid >> 6) * 6 + q] = pc[q];
        for (int q = 0; q < 5; ++q) prof[72 + (tid >> 6) * 6 + q] = pe[q];
    }
}

// The weights of the three stages in the form the layer kernel keeps them in registers, made ONCE per weight version by one
// workgroup: per wave the largest |W * bn_scale| of its 32 columns (-> the stage's power-of-two scale), then every wave's fp16
// plane fragments.  Layout: see LfArgs::prep.
template <int NKE, int NK0, int NK1>
__global__ __launch_bounds__(768) void layer_fused_prepare_kernel(LfArgs a, unsigned *prep) {
    __shared__ unsigned wmax[12];
    const int tid = threadIdx.x;
    const int grp = tid >> 8, t = tid & 255, lane = t & 63, w = t >> 6, li = lane & 31, lh = lane >> 5;
    const LfStage &st = grp == 0 ? a.e : (grp == 1 ? a.s0 : a.s1);
    const int col = 32 * w + li;
    const bool cok = col < st.n_out;
    float bnscale = 1.f;
    if (cok && st.bn_scale) bnscale = st.bn_scale[col];
    {
        unsigned m = lf_weight_absmax(st, col, cok, bnscale);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
        if (lane == 0) { wmax[4 * grp + w] = m; prep[4 * grp + w] = m; }
    }
    __syncthreads();
    float wscale, inv_w;
    lf_scale(max(max(wmax[4 * grp], wmax[4 * grp + 1]), max(wmax[4 * grp + 2], wmax[4 * grp + 3])), wscale, inv_w);
    lf_u4 *frag = reinterpret_cast<lf_u4 *>(prep + LF_PREP_HDR);
    if (grp == 0) {
        lf_u4 Bh[NKE], Bl[NKE];
        lf_weight_planes<NKE>(st, col, cok, lh, bnscale, wscale, 0, 0, 1.f, Bh, Bl);
        lf_u4 *f = frag + (NKE * 2 * w) * 64 + lane;
#pragma unroll
        for (int q = 0; q < NKE; ++q) { f[(2 * q) * 64] = Bh[q]; f[(2 * q + 1) * 64] = Bl[q]; }
    } else if (grp == 1) {
        lf_u4 Bh[NK0], Bl[NK0];
        lf_weight_planes<NK0>(st, col, cok, lh, bnscale, wscale, a.e.n_out, a.d_x, 1.f, Bh, Bl);
        lf_u4 *f = frag + (4 * NKE * 2 + NK0 * 2 * w) * 64 + lane;
#pragma unroll
        for (int q = 0; q < NK0; ++q) { f[(2 * q) * 64] = Bh[q]; f[(2 * q + 1) * 64] = Bl[q]; }
    } else {
        lf_u4 Bh[NK1], Bl[NK1];
        lf_weight_planes<NK1>(st, col, cok, lh, bnscale, wscale, 0, 0, 1.f, Bh, Bl);
        lf_u4 *f = frag + (4 * (NKE + NK0) * 2 + NK1 * 2 * w) * 64 + lane;
#pragma unroll
        for (int q = 0; q < NK1; ++q) { f[(2 * q) * 64] = Bh[q]; f[(2 * q + 1) * 64] = Bl[q]; }
    }
}

#undef LF_MF

template <int NKE, int NK0, int NK1, int WG = 0, bool PROF = false>
static int lf_launch(const LfArgs &a, hipStream_t st) {
    constexpr size_t lds = (size_t)2 * LF_TE * (16 * NKE + 8) * 2 + (size_t)LF_TE * LF_PY * 4 + (size_t)LF_TN * LF_PY * 4 +
                           (size_t)2 * LF_TN * (16 * NK0 + 8) * 2 + (size_t)2 * LF_TN * LF_PY * 4 + (size_t)2 * LF_TN * (16 * NK1 + 8) * 2 +
                           (size_t)LF_NXJ * LF_TN * 32 * 4 +
                           ((size_t)LF_MAXB * LF_TE + LF_TE + 2 * LF_TN + LF_NSLOT * LF_SEGW + 4 + 12 + 12 + 64 + 4) * 4;
    static_assert(lds <= 160 * 1024, "LDS budget");
    const void *fn = reinterpret_cast<const void *>(&layer_fused_kernel<NKE, NK0, NK1, WG, PROF>);
    static DeviceOnce attr_set;                                        // (the attribute is per device)
    const int attr_dev = current_device();
    if (!attr_set.done(attr_dev)) {
        hipError_t e0 = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e0 != hipSuccess) return set_error(GSN_E_HIP, "hipFuncSetAttribute(layer_fused_kernel): %s", hipGetErrorString(e0));
        attr_set.mark(attr_dev);
    }
    int64_t gx = 256;
    { const char *d = getenv("GSN_FUSED_GRID"); if (d && atoi(d) > 0) gx = atoi(d); }
    const int64_t n_tiles = ((int64_t)a.n_nodes + LF_TN - 1) / LF_TN;
    if (gx > n_tiles) gx = n_tiles;
    if (getenv("GSN_CHAIN_TRACE")) fprintf(stderr, "gsn chain: layer_fused_kernel<%d,%d,%d,%d> nodes %d edges %d grid %lld\n", NKE, NK0, NK1, WG, a.n_nodes, a.n_edges, (long long)gx);
    unsigned long long *prof = nullptr;
    if (PROF) { (void)hipMalloc(&prof, (24 * 6 + 512) * 8); (void)hipMemset(prof, 0, (24 * 6 + 512) * 8); }
    static const int prio_env = [] { const char *d = getenv("GSN_FUSED_PRIO"); return d ? atoi(d) : 1; }();   // matrix phases at raised wave priority (~1 %)
    int prio = prio_env & 0xff;
    if (PROF) { const char *d = getenv("GSN_FUSED_ABLATE"); if (d) prio |= atoi(d) << 8; }
    hipLaunchKernelGGL((layer_fused_kernel<NKE, NK0, NK1, WG, PROF>), dim3((unsigned)gx), dim3(768), lds, st, a, prof, prio);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "layer_fused_kernel: %s", hipGetErrorString(e));
    if (PROF) {
        unsigned long long h[24 * 6 + 512];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost);
        (void)hipFree(prof);
        static int shown = 0;
        if (shown % 8 == 7 && gx == 256) {           // when did the workgroups enter and leave (relative to the first entry)?
            unsigned long long t0 = ~0ull;
            // (s_memtime counters are per XCD: only the workgroups of one XCD, blockIdx % 8 == 0, are comparable)
            for (int b = 0; b < 256; b += 8) if (h[144 + 2 * b] && h[144 + 2 * b] < t0) t0 = h[144 + 2 * b];
            fprintf(stderr, "fusedprof workgroup entry / exit on XCD 0 (cycles after its first entry):");
            for (int b = 0; b < 256; b += 40) fprintf(stderr, " [%d] %lld / %lld", b, (long long)(h[144 + 2 * b] - t0), (long long)(h[145 + 2 * b] - t0));
            long long emax = 0, xmin = 1ll << 62, xmax = 0;
            for (int b = 0; b < 256; b += 8) {
                const long long e = (long long)(h[144 + 2 * b] - t0), x = (long long)(h[145 + 2 * b] - t0);
                emax = e > emax ? e : emax; xmin = x < xmin ? x : xmin; xmax = x > xmax ? x : xmax;
            }
            fprintf(stderr, "\nfusedprof last entry %lld, first exit %lld, last exit %lld\n", emax, xmin, xmax);
        }
        if (shown++ % 8 == 7)
            for (int w = 0; w < 12; ++w) {
                const unsigned long long *o = h + w * 6;
                if (o[5]) fprintf(stderr, "fusedprof %s%d steps %llu: phase1 %llu barrier %llu phase2 %llu barrier %llu bookkeeping %llu (cycles per step)\n",
                                  w < 4 ? "E" : (w < 8 ? "S0-" : "S1-"), w & 3, o[5], o[0] / o[5], o[1] / o[5], o[2] / o[5], o[3] / o[5], o[4] / o[5]);
                const unsigned long long *e = h + 72 + w * 6;
                if (w < 4 && o[5])
                    fprintf(stderr, "fusedprof E%d detail: iterator+publish %llu sources+gathers %llu matrix+epilogue %llu | table %llu split %llu sums %llu\n", w,
                            e[0] / o[5], e[1] / o[5], e[2] / o[5], e[3] / o[5], e[4] / o[5], e[5] / o[5]);
                else if (w < 8 && o[5])
                    fprintf(stderr, "fusedprof S0-%d detail: iterator %llu publish %llu staging %llu x-loads %llu | matrix+epilogue %llu sums %llu\n", w & 3,
                            e[0] / o[5], e[1] / o[5], e[2] / o[5], e[3] / o[5], e[4] / o[5], e[5] / o[5]);
                if (w == 4) fprintf(stderr, "fusedprof prologue of workgroup 0 / 128 (wave S0-0): LDS clear + tables %llu / %llu, weight fragments %llu / %llu cycles\n",
                                    h[140], h[136], h[141], h[137]);
                if (w >= 8 && o[5])
                    fprintf(stderr, "fusedprof S1-%d detail: matrix %llu epilogue+stores %llu | record %llu split %llu sums %llu\n", w & 3,
                            e[0] / o[5], e[1] / o[5], e[2] / o[5], e[3] / o[5], e[4] / o[5]);
            }
    }
    return GSN_OK;
}

static bool lf_stage_ok(const gsn_chain_stage &g) {
    if (!g.W || g.n_out < 1 || g.n_out > 128 || (g.n_out & 3)) return false;
    if (g.act != 0 && g.act != 1) return false;
    if ((g.bn_scale != nullptr) != (g.bn_shift != nullptr) || (g.bn_scale != nullptr) != (g.bn_mean != nullptr)) return false;
    return true;
}

}  // namespace gsn

using namespace gsn;

extern "C" int gsn_layer_fused_supported(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0,
                                         const gsn_chain_stage *node1) {
    if (!edge || !node0 || !node1) return 0;
    if (w_supported(edge, d_x, node0, node1)) return 1;                // d_x = 128 (layer_w.hip)
    if (!lf_stage_ok(*edge) || !lf_stage_ok(*node0) || !lf_stage_ok(*node1)) return 0;
    if ((edge->n_out & 31) || (node0->n_out & 31)) return 0;           // the stagers own whole 32-column groups of S and H
    if (edge->n_blocks < 1 || edge->n_blocks > LF_MAXB || !edge->blocks) return 0;
    int64_t ke = 0;
    for (int b = 0; b < edge->n_blocks; ++b) {
        const gsn_block &bl = edge->blocks[b];
        if (!bl.data || !bl.idx32 || bl.idx || bl.width <= 0 || (bl.width & 3)) return 0;   // int32 row sources, float4 gathers
        if (reinterpret_cast<uintptr_t>(bl.data) & 15) return 0;
        ke += bl.width;
    }
    if (ke > 80) return 0;
    if (d_x < 4 || (d_x & 3) || d_x + 4 > 32 * LF_NXJ) return 0;
    const int64_t k0 = d_x + edge->n_out + 4;
    if (k0 > 160) return 0;
    if (node0->n_blocks != 0 || node1->n_blocks != 0) return 0;
    return 1;
}

// shared by the prepare and the forward entry: the stage descriptions of the kernel and its (NK0, NK1) instantiation
static void lf_fill_stages(LfArgs &a, const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0, const gsn_chain_stage *node1) {
    a.e_nblocks = edge->n_blocks;
    int ke = 0;
    for (int b = 0; b < edge->n_blocks; ++b) {
        a.e_data[b] = edge->blocks[b].data; a.e_idx[b] = edge->blocks[b].idx32; a.e_width[b] = (int)edge->blocks[b].width;
        ke += a.e_width[b];
    }
    auto fill = [](LfStage &s, const gsn_chain_stage &g, int k) {
        s.W = g.W; s.bias = g.bias; s.bn_mean = g.bn_mean; s.bn_scale = g.bn_scale; s.bn_shift = g.bn_shift;
        s.k_total = k; s.n_out = (int)g.n_out; s.act = g.act;
    };
    fill(a.e, *edge, ke);
    fill(a.s0, *node0, (int)(d_x + edge->n_out + 4));
    fill(a.s1, *node1, (int)node0->n_out);
    a.d_x = (int)d_x;
}
static int lf_nk0(const LfArgs &a) { return a.s0.k_total <= 96 ? 6 : 10; }
static int lf_nk1(const LfArgs &a) { return a.s1.k_total <= 64 ? 4 : 8; }
// the prepared buffer: this file's fragments, then (where the shape fits) those of the register-resident kernel (layer_rr.hip)
static int64_t lf_prep_bytes_own(const LfArgs &a) { return (((int64_t)(LF_PREP_HDR + 4 * (5 + lf_nk0(a) + lf_nk1(a)) * 2 * 64 * 4) * 4) + 255) / 256 * 256; }

extern "C" int64_t gsn_layer_fused_prepared_bytes(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0,
                                                  const gsn_chain_stage *node1) {
    if (!gsn_layer_fused_supported(edge, d_x, node0, node1)) return 0;
    if (w_supported(edge, d_x, node0, node1)) return w_prepared_bytes(edge, d_x, node0, node1);
    LfArgs a{};
    lf_fill_stages(a, edge, d_x, node0, node1);
    return lf_prep_bytes_own(a) + rr_prepared_bytes(edge, d_x, node0, node1);
}

extern "C" int gsn_layer_fused_prepare_hip(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0,
                                           const gsn_chain_stage *node1, void *prepared, void *stream) {
    if (!gsn_layer_fused_supported(edge, d_x, node0, node1))
        return set_error(GSN_E_UNSUPPORTED, "gsn_layer_fused_prepare_hip: shape outside the fused layer kernel");
    if (!prepared || (reinterpret_cast<uintptr_t>(prepared) & 15)) return set_error(GSN_E_INVALID, "gsn_layer_fused_prepare_hip: prepared must be a 16-byte aligned device buffer");
    if (w_supported(edge, d_x, node0, node1)) return w_prepare(edge, d_x, node0, node1, prepared, reinterpret_cast<hipStream_t>(stream));
    LfArgs a{};
    lf_fill_stages(a, edge, d_x, node0, node1);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    unsigned *pp = reinterpret_cast<unsigned *>(prepared);
    const int nk0 = lf_nk0(a), nk1 = lf_nk1(a);
    if (nk0 == 6 && nk1 == 4) hipLaunchKernelGGL((layer_fused_prepare_kernel<5, 6, 4>), dim3(1), dim3(768), 0, st, a, pp);
    else if (nk0 == 6) hipLaunchKernelGGL((layer_fused_prepare_kernel<5, 6, 8>), dim3(1), dim3(768), 0, st, a, pp);
    else if (nk1 == 4) hipLaunchKernelGGL((layer_fused_prepare_kernel<5, 10, 4>), dim3(1), dim3(768), 0, st, a, pp);
    else hipLaunchKernelGGL((layer_fused_prepare_kernel<5, 10, 8>), dim3(1), dim3(768), 0, st, a, pp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "layer_fused_prepare_kernel: %s", hipGetErrorString(e));
    if (rr_supported(edge, d_x, node0, node1))
        return rr_prepare(edge, d_x, node0, node1, reinterpret_cast<unsigned char *>(prepared) + lf_prep_bytes_own(a), st);
    return GSN_OK;
}

extern "C" int64_t gsn_layer_fused_workspace_bytes(int64_t n_nodes, const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0,
                                                   const gsn_chain_stage *node1) {
    if (n_nodes <= 0 || !w_supported(edge, d_x, node0, node1)) return 0;
    return n_nodes * 4;                                  // the row exponents of x (layer_w.hip)
}

static int lf_forward(int64_t n_nodes, int64_t n_edges, const int32_t *seg_ptr, const gsn_chain_stage *edge,
                      const float *x, int64_t d_x, const gsn_chain_stage *node0, const gsn_chain_stage *node1,
                      const void *prepared, float *out, void *workspace, int64_t workspace_bytes, const int32_t *x_row_exp,
                      int32_t *out_row_exp, void *stream);

extern "C" int gsn_layer_fused_fwd_hip(int64_t n_nodes, int64_t n_edges, const int32_t *seg_ptr, const gsn_chain_stage *edge,
                                       const float *x, int64_t d_x, const gsn_chain_stage *node0, const gsn_chain_stage *node1,
                                       const void *prepared, float *out, void *stream) {
    return lf_forward(n_nodes, n_edges, seg_ptr, edge, x, d_x, node0, node1, prepared, out, nullptr, 0, nullptr, nullptr, stream);
}

extern "C" int gsn_layer_fused_fwd_ws_hip(int64_t n_nodes, int64_t n_edges, const int32_t *seg_ptr, const gsn_chain_stage *edge,
                                          const float *x, int64_t d_x, const gsn_chain_stage *node0, const gsn_chain_stage *node1,
                                          const void *prepared, float *out, void *workspace, int64_t workspace_bytes,
                                          const int32_t *x_row_exp, int32_t *out_row_exp, void *stream) {
    if (out_row_exp && (!node1 || node1->n_out != 128))
        return set_error(GSN_E_INVALID, "gsn_layer_fused_fwd_ws_hip: out_row_exp is defined for 128-wide output rows");
    const int rc = lf_forward(n_nodes, n_edges, seg_ptr, edge, x, d_x, node0, node1, prepared, out, workspace, workspace_bytes, x_row_exp, out_row_exp, stream);
    if (rc != GSN_OK || !out_row_exp || n_nodes <= 0 || w_supported(edge, d_x, node0, node1)) return rc;
    return w_row_exponents(n_nodes, out, out_row_exp, reinterpret_cast<hipStream_t>(stream));     // (the other kernels: a pass over the rows they wrote)
}

static int lf_forward(int64_t n_nodes, int64_t n_edges, const int32_t *seg_ptr, const gsn_chain_stage *edge,
                      const float *x, int64_t d_x, const gsn_chain_stage *node0, const gsn_chain_stage *node1,
                      const void *prepared, float *out, void *workspace, int64_t workspace_bytes, const int32_t *x_row_exp,
                      int32_t *out_row_exp, void *stream) {
    if (!gsn_layer_fused_supported(edge, d_x, node0, node1))
        return set_error(GSN_E_UNSUPPORTED, "gsn_layer_fused_fwd_hip: shape outside the fused layer kernel (edge K <= 80, d_x + n_msg + 4 <= 160, "
                                            "widths <= 128 and multiples of 4, int32 row sources, identity / relu)");
    if (!seg_ptr || !x || !out || !prepared) return set_error(GSN_E_INVALID, "gsn_layer_fused_fwd_hip: null seg_ptr / x / out / prepared (gsn_layer_fused_prepare_hip)");
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(prepared)) & 15) return set_error(GSN_E_INVALID, "gsn_layer_fused_fwd_hip: x and prepared must be 16-byte aligned");
    if (n_nodes > (int64_t)2000000000 || n_edges > (int64_t)2000000000) return set_error(GSN_E_UNSUPPORTED, "gsn_layer_fused_fwd_hip: 32-bit row arithmetic");
    if (n_nodes <= 0) return GSN_OK;
    if (w_supported(edge, d_x, node0, node1)) {
        if (workspace && (workspace_bytes < n_nodes * 4 || (reinterpret_cast<uintptr_t>(workspace) & 3)))
            return set_error(GSN_E_INVALID, "gsn_layer_fused_fwd_ws_hip: workspace smaller than gsn_layer_fused_workspace_bytes() or misaligned");
        return w_forward(n_nodes, n_edges, seg_ptr, edge, x, d_x, node0, node1, prepared, out, reinterpret_cast<int32_t *>(workspace), x_row_exp, out_row_exp,
                         reinterpret_cast<hipStream_t>(stream));
    }
    LfArgs a{};
    a.n_nodes = (int)n_nodes; a.n_edges = (int)n_edges; a.seg_ptr = seg_ptr;
    lf_fill_stages(a, edge, d_x, node0, node1);
    a.x = x; a.out = out; a.prep = reinterpret_cast<const unsigned *>(prepared);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // every stage 128 wide, d_x + 4 <= 32: the register-resident kernel (layer_rr.hip; GSN_FUSED_RR=0 keeps this file's kernel)
    if (rr_supported(edge, d_x, node0, node1)) {
        const int rc = rr_forward(n_nodes, n_edges, seg_ptr, edge, x, d_x, node0, node1, reinterpret_cast<const unsigned char *>(prepared) + lf_prep_bytes_own(a), out, st);
        if (rc != 1) return rc;
    }
    const int k0 = a.s0.k_total, k1 = a.s1.k_total;
    // every width 128 (or 64): the reference's d = 128 / 64 layers -- widths as compile-time constants
    const bool generic = getenv("GSN_FUSED_GENERIC") != nullptr;
    const bool w128 = !generic && a.e.n_out == 128 && a.s0.n_out == 128 && a.s1.n_out == 128 && k0 <= 160 && k0 > 96 && k1 > 64;
    const bool w64 = !generic && a.e.n_out == 64 && a.s0.n_out == 64 && a.s1.n_out == 64 && k0 <= 96;
    { const char *d = getenv("GSN_FUSED_PROF"); if (d && atoi(d) && w128) return lf_launch<5, 10, 8, 4, true>(a, st); }
    if (w128) return lf_launch<5, 10, 8, 4>(a, st);
    if (w64) return lf_launch<5, 6, 4, 2>(a, st);
    if (k0 <= 96) return k1 <= 64 ? lf_launch<5, 6, 4>(a, st) : lf_launch<5, 6, 8>(a, st);
    return k1 <= 64 ? lf_launch<5, 10, 4>(a, st) : lf_launch<5, 10, 8>(a, st);
}

// ---- the d = 128 layer on graph-aligned tiles (layer_g.hip): same shapes, same prepared buffer as the d = 128 kernel above ------------
extern "C" int gsn_layer_fused_graphs_supported(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0,
                                                const gsn_chain_stage *node1) {
    if (!edge || !node0 || !node1) return 0;
    return g_supported(edge, d_x, node0, node1);
}

extern "C" int gsn_layer_fused_fwd_graphs_hip(int64_t n_nodes, int64_t n_edges, const int32_t *seg_ptr, const gsn_chain_stage *edge,
                                              const float *x, int64_t d_x, const gsn_chain_stage *node0, const gsn_chain_stage *node1,
                                              const void *prepared, int64_t n_graphs, const int64_t *node_ptr, int64_t max_nodes,
                                              float *out, void *stream) {
    if (!gsn_layer_fused_graphs_supported(edge, d_x, node0, node1))
        return set_error(GSN_E_UNSUPPORTED, "gsn_layer_fused_fwd_graphs_hip: shape outside the graph-aligned d = 128 kernel");
    if (!seg_ptr || !x || !out || !prepared || !node_ptr) return set_error(GSN_E_INVALID, "gsn_layer_fused_fwd_graphs_hip: null seg_ptr / x / out / prepared / node_ptr");
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(prepared)) & 15) return set_error(GSN_E_INVALID, "gsn_layer_fused_fwd_graphs_hip: x and prepared must be 16-byte aligned");
    if (n_nodes > (int64_t)2000000000 || n_edges > (int64_t)2000000000 || n_graphs > (int64_t)2000000000) return set_error(GSN_E_UNSUPPORTED, "gsn_layer_fused_fwd_graphs_hip: 32-bit row arithmetic");
    if (n_nodes <= 0) return GSN_OK;
    return g_forward(n_nodes, n_edges, seg_ptr, edge, x, d_x, node0, node1, prepared, n_graphs, node_ptr, max_nodes, out, reinterpret_cast<hipStream_t>(stream));
}

// ---- the register-resident layer on exact fp16 row packs (layer_rp.hip) ------------------------------------------------------------
extern "C" int gsn_layer_fused_pack16_supported(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0,
                                                const gsn_chain_stage *node1) {
    if (!edge || !node0 || !node1) return 0;
    return rp_supported(edge, d_x, node0, node1);
}

extern "C" int64_t gsn_layer_fused_pack16_prepared_bytes(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0,
                                                         const gsn_chain_stage *node1) {
    if (!gsn_layer_fused_pack16_supported(edge, d_x, node0, node1)) return 0;
    return rr_prepared_bytes(edge, d_x, node0, node1);
}

extern "C" int gsn_layer_fused_pack16_prepare_hip(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0,
                                                  const gsn_chain_stage *node1, void *prepared, void *stream) {
    if (!gsn_layer_fused_pack16_supported(edge, d_x, node0, node1))
        return set_error(GSN_E_UNSUPPORTED, "gsn_layer_fused_pack16_prepare_hip: shape outside the packed-row layer kernel");
    if (!prepared || (reinterpret_cast<uintptr_t>(prepared) & 15)) return set_error(GSN_E_INVALID, "gsn_layer_fused_pack16_prepare_hip: prepared must be a 16-byte aligned device buffer");
    return rr_prepare(edge, d_x, node0, node1, prepared, reinterpret_cast<hipStream_t>(stream), true);
}

extern "C" int gsn_layer_fused_fwd_pack16_hip(int64_t n_nodes, int64_t n_edges, const int32_t *seg_ptr, const gsn_chain_stage *edge,
                                              const float *x, int64_t d_x, const gsn_chain_stage *node0, const gsn_chain_stage *node1,
                                              const void *prepared, const gsn_pack16 *pack, int64_t edge_rows, float *out, void *stream) {
    if (!gsn_layer_fused_pack16_supported(edge, d_x, node0, node1))
        return set_error(GSN_E_UNSUPPORTED, "gsn_layer_fused_fwd_pack16_hip: shape outside the packed-row layer kernel");
    if (!seg_ptr || !out || !prepared || !pack) return set_error(GSN_E_INVALID, "gsn_layer_fused_fwd_pack16_hip: null seg_ptr / out / prepared / pack");
    if (reinterpret_cast<uintptr_t>(prepared) & 15) return set_error(GSN_E_INVALID, "gsn_layer_fused_fwd_pack16_hip: prepared must be 16-byte aligned");
    if (n_nodes > (int64_t)2000000000 || n_edges > (int64_t)2000000000 || edge_rows < 0) return set_error(GSN_E_UNSUPPORTED, "gsn_layer_fused_fwd_pack16_hip: 32-bit row arithmetic");
    if (n_nodes <= 0) return GSN_OK;
    const int rc = rp_forward(n_nodes, n_edges, seg_ptr, edge, x, d_x, node0, node1, prepared, pack, edge_rows, out, reinterpret_cast<hipStream_t>(stream));
    if (rc == 1) return set_error(GSN_E_UNSUPPORTED, "gsn_layer_fused_fwd_pack16_hip: packs beyond 2 GiB (32-bit buffer offsets); use gsn_layer_fused_fwd_hip");
    return rc;
}
