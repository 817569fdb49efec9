// The one-launch `general` layer for WIDE node rows: d_x = 128, edge rows cat(x_i, x_j, ids.., e) of K = 256 + (<= 16) columns --
// the hidden layers of a d = 128 model (GSN_edge_sparse.py:82-170 with K = 260 / 272, MPNN_edge_sparse.py:110-151), which the
// register-resident kernel of layer_rr.hip (d_x + 4 <= 32, K <= 76) does not take:
//
//     r_e  = act_e( bn_e( cat(x_i, x_j, ids.., e) W1^T + b1 ) )              per edge
//     S_v  = sum_{e -> v} r_e                                                per node      (torch.sparse.sum)
//     h_v  = act_0( bn_0( [x_v | S_v | deg_v] W0'^T + b0 ) )                 per node
//     out_v = act_1( bn_1( h_v W1'^T + b1' ) )                               per node
//
// Same data flow as layer_rr.hip (every wave owns a range of nodes and takes tiles of <= 32 nodes through all stages; accumulator
// tiles turn into operand fragments in place; per-node sums through an incidence product), other budget: 332 KiB of fp16 plane
// fragments instead of 184, and 2.2x the products per tile.  So
//   * ONE wave per SIMD (256 threads per workgroup, one workgroup per CU): 512 registers per lane (256 of them accumulator
//     registers).  The edge stage keeps the accumulators of a UNIT of 64 edge rows (two 32-row sub-blocks x four feature blocks =
//     128 registers) and walks the 17 16-column chunks of the rows once: per chunk 24 products that share 8 weight fragments, with
//     the NEXT chunk's gathered fp32 values converted to scaled fp16 planes, the fragment reads of the next step and the gathers
//     two chunks ahead issued in between -- a wave issues in order, and what is placed between two MFMAs runs under them
//     (scripts/micro/issue_mix.hip: MFMA + 4 vector instructions = 34 cycles; scripts/micro/wide_edge_loop.hip: this loop at
//     850-880 cycles per chunk of 768).  With one wave per SIMD nothing else hides anything: the order of issue is prescribed
//     (W_MIX / W_MIX_EDGE), and a single register spill costs a tile ~4 000 cycles (its reload waits for every gather in flight).
//   * the edge stage's fragments (136 KiB) live in LDS.  Those of the two node stages (192 KiB) cannot, and a CU's path to L2
//     carries only ~30 bytes per cycle (measured: four private streams per CU were the bound, at any depth of prefetch and with
//     half the CUs idle alike): they stream ONCE PER WORKGROUP.  After its tile's edge stage a wave meets the other three; then for
//     each of the 24 chunk steps of the node stages wave w loads fragments 2 w, 2 w + 1 of the step four steps ahead into
//     registers, writes them into a three-slot ring in LDS two steps ahead, and all four waves read the step's eight fragments
//     from there.  One barrier per step; a wave whose range is exhausted keeps carrying its share until every range is.
//   * the power-of-two row scale of an edge row must be known before its first chunk is converted, and the row is 1 KiB: it is
//     made from the row exponents of the two node rows (a side array of one int per node, written by a small pass over x in
//     front of this kernel: layer_w_row_exp_kernel) and the per-edge columns (gathered first).
//   * node stage 0's bias and deg column enter as ONE bf16 product per feature block (three bf16 planes of c0 and w_deg against
//     the lane's row scale and deg x row scale -- powers of two and small integers are exact in bf16 at any exponent) instead of
//     128 values per lane that would have to be loaded and held.
//
// Matrix arithmetic: fp16x3 everywhere (two fp16 planes per operand after exact power-of-two row / matrix scaling, three plane
// products, fp32 accumulation); the activated edge rows enter the incidence product as three bf16 planes (exact at any magnitude).
// Non-finite values as in layer_rr.hip: an edge / node / hidden row that holds an Inf or a NaN makes exactly the output rows that
// see it NaN.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "chain_common.h"
#include "layer_rr_inl.h"
#include "layer_w.h"

namespace gsn {

struct WArgs {
    int n_nodes, n_edges;
    const int32_t *seg_ptr;
    const int32_t *ridx[W_MAXROLE];   // row-index arrays: of x_i, of x_j, of the per-edge blocks
    WQuad zq[4];                      // the four 16-byte quads of the per-edge chunk
    const float *x;
    const int32_t *xe;                // exponent field of max |x[v][:]| per node (255: the row holds an Inf / NaN)
    int32_t *xe_out;                  // the same of the output rows, written with them (or null)
    float *out;
    const unsigned *prep;
    int n_ranges;
};

// ------------------------------------------------------------------------------------------------------------------------------
// tile iterator (layer_rr.hip's, handing out UNITS of <= 64 edge rows)
struct WDesc {
    int m0, e0, pk;                         // pk: valid | first << 1 | last << 2 | nn << 6 | ne << 12
    __device__ __forceinline__ int valid() const { return pk & 1; }
    __device__ __forceinline__ int last() const { return (pk >> 2) & 1; }
    __device__ __forceinline__ int nn() const { return (pk >> 6) & 63; }
    __device__ __forceinline__ int ne() const { return (pk >> 12) & 127; }
};

struct WIter {
    const int32_t *seg;
    int n_nodes, m_next, m_end;
    int m0, nn, eb, ee, ec, pending;
    int win;
};

__device__ __forceinline__ void w_iter_load(WIter &it, int lane) {
    int idx = it.m_next + lane;
    idx = idx < it.n_nodes ? idx : it.n_nodes;
    it.win = it.seg[idx];
}

__device__ __forceinline__ WDesc w_iter_next(WIter &it, int lane) {
    WDesc d; d.m0 = 0; d.e0 = 0; d.pk = 0;
    if (!(it.pending || it.ec < it.ee)) {
        if (it.m_next >= it.m_end) return d;
        int nmax = it.m_end - it.m_next;
        nmax = nmax < W_TN ? nmax : W_TN;
        const int w0 = __builtin_amdgcn_readfirstlane(it.win);
        const int cnt = it.win - w0;
        const int ne_all = __builtin_amdgcn_readlane(cnt, nmax);
        int nn = nmax;
        if (ne_all > W_UE) {
            const int cap = ne_all / W_UE * W_UE;
            const unsigned long long ok = __ballot(lane <= nmax && cnt <= cap);
            nn = __popcll(ok) - 1;
            nn = nn < 1 ? 1 : nn;
        }
        nn = __builtin_amdgcn_readfirstlane(nn);
        it.m0 = it.m_next; it.nn = nn; it.eb = w0; it.ee = __builtin_amdgcn_readlane(it.win, nn);
        it.ec = it.eb; it.pending = 1;
        it.m_next += nn;
        w_iter_load(it, lane);
    }
    d.m0 = it.m0; d.e0 = it.ec;
    const int left = it.ee - it.ec;
    const int ne = left < W_UE ? left : W_UE;
    d.pk = 1 | ((it.ec == it.eb) << 1) | ((it.ec + W_UE >= it.ee) << 2) | (it.nn << 6) | (ne << 12);
    it.ec += W_UE; it.pending = 0;
    return d;
}

// row indices of a unit's edge rows (sub-block sb, row li), segment bounds and row exponent of the lane's target
struct WIdx {
    int r[2][W_MAXROLE];
    int pt, pt1, xet;
};

__device__ __forceinline__ void w_idx_load(const WArgs &a, const WDesc &d, int li, WIdx &ix) {
    const int ne = d.ne();
    if (d.valid() && ne > 0) {
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            const int row = 32 * sb + li;
            const int e = d.e0 + (row < ne ? row : ne - 1);
#pragma unroll
            for (int q = 0; q < W_MAXROLE; ++q) ix.r[sb][q] = a.ridx[q][e];
        }
    }
    if (d.valid()) {
        const int nn = d.nn();
        int t = d.m0 + li;
        t = t < a.n_nodes ? t : a.n_nodes - 1;          // (lanes past nn are masked when used)
        ix.pt = a.seg_ptr[t];
        ix.pt1 = a.seg_ptr[t + 1];
        ix.xet = a.xe[d.m0 + (li < nn ? li : nn - 1)];
    }
}

typedef const __attribute__((address_space(1))) rr_f4 *w_gptr;

// issue order inside one chunk step: NM x (one MFMA, then -- while there are any -- one LDS read and one global load, then NV vector
// instructions).  A wave issues in order; a ds_read_b128 holds the issue port ~16-20 cycles (scripts/micro/lds_frag_stream.hip), an MFMA 4
// of the 32 its pipe is busy: the eight fragment reads of a step cost 160 cycles in front of the products and nothing between them.
#ifdef RR_NOMIX
#define W_MIX(NM, ND, NG, NV)
#else
#define W_MIX(NM, ND, NG, NV) _Pragma("unroll") for (int mix_q = 0; mix_q < (NM); ++mix_q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); \
    if (mix_q < (ND)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); if (mix_q < (NG)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); \
    __builtin_amdgcn_sched_group_barrier(0x002, (NV), 0); }
#endif
// the edge stage's step: a feature block's two fragments are replaced behind its six products (reads behind products 6, 7, 12, 13, ...)
#ifndef W_MIXV
#define W_MIXV 2
#endif
#ifdef RR_NOMIX
#define W_MIX_EDGE()
#else
#define W_MIX_EDGE() _Pragma("unroll") for (int mix_q = 0; mix_q < 24; ++mix_q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); \
    if (mix_q >= 5 && (mix_q - 5) % 6 < 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); if (mix_q < 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); \
    __builtin_amdgcn_sched_group_barrier(0x002, W_MIXV, 0); }
#endif

// ------------------------------------------------------------------------------------------------------------------------------
template <bool PROF>
__global__ __launch_bounds__(256) void layer_fused_kernel_w(WArgs a, unsigned long long *prof) {
    auto clk = [&]() -> unsigned {
        if (!PROF) return 0u;
        __builtin_amdgcn_sched_barrier(0);
        const unsigned v = (unsigned)__builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
        return v;
    };
    unsigned pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // scales, edge loop, edge epilogue + prefetch, rendezvous, stage 0, stage 1, units, tiles
    unsigned pd[4] = {0, 0, 0, 0};               // inside `stage 1`: between the stages, its eight steps, advance, stores
    using SH = WShape;
    constexpr int WB = SH::WB, NKS = SH::NKS, NK0 = SH::NK0;
    constexpr int NSTEP = NK0 + NKS;              // chunk steps of the two node stages (8 streamed fragments each)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane0 = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li0 = lane0 & 31, lh0 = lane0 >> 5;
    // LDS address of this lane's 16 bytes inside fragment 0; the bases 64 and 128 KiB further (16-bit immediate offsets) are made where
    // they are used, opaque to the compiler (layer_rr_inl.h: rr_lds_frag)
    const unsigned ldsb0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem + 16u * (unsigned)lane0;
    auto lds_bases = [&](unsigned (&b)[3]) {
        b[0] = ldsb0;
        asm volatile("" : "+v"(b[0]));
        b[1] = b[0] + 0x10000u; b[2] = b[0] + 0x20000u;
        asm volatile("" : "+v"(b[1]), "+v"(b[2]));
    };

    // ---- prologue: edge-stage fragments -> LDS ----------------------------------------------------------------------------------
    {
        const rr_u4 *src = reinterpret_cast<const rr_u4 *>(a.prep + W_HDR);
        rr_u4 *dst = reinterpret_cast<rr_u4 *>(smem);
        for (int i = tid; i < SH::F_LDS * 64; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const int Ee = (int)a.prep[WH_EE], E0 = (int)a.prep[WH_E0], E1 = (int)a.prep[WH_E1], e_min = (int)a.prep[WH_EMIN];
    const bool w_bad = a.prep[WH_BAD] != 0;
    const unsigned acts = a.prep[WH_ACT];
    auto sgpr = [](float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); };
    const float lo_e = sgpr((acts & 1) ? 0.f : -3.0e38f), lo_0 = sgpr((acts & 2) ? 0.f : -INFINITY), lo_1 = sgpr((acts & 4) ? 0.f : -INFINITY);
    const __amdgpu_buffer_rsrc_t wstream = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(a.prep) + W_HDR + SH::F_W0 * 256, 0, (SH::F_ALL - SH::F_W0) * 1024, 0x00020000);
    const float *tab0 = reinterpret_cast<const float *>(a.prep + W_HDR + SH::F_ALL * 256);     // (folded biases: global memory, cache hits)

    // ---- this wave's node range (an empty one still walks through the workgroup's rendezvous) -------------------------------------
    const int range = wave * (int)gridDim.x + (int)blockIdx.x;
    WIter it;
    it.seg = a.seg_ptr; it.n_nodes = a.n_nodes;
    it.m_next = range < a.n_ranges ? (int)((int64_t)a.n_nodes * range / a.n_ranges) : 0;
    it.m_end = range < a.n_ranges ? (int)((int64_t)a.n_nodes * (range + 1) / a.n_ranges) : 0;
    it.m0 = 0; it.nn = 0; it.eb = 0; it.ee = 0; it.ec = 0; it.pending = 0; it.win = 0;
    w_iter_load(it, lane0);
    WDesc cur = w_iter_next(it, lane0);
    WIdx ixc, ixn;
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int q = 0; q < W_MAXROLE; ++q) { ixc.r[sb][q] = 0; ixn.r[sb][q] = 0; }
    ixc.pt = ixc.pt1 = ixn.pt = ixn.pt1 = 0; ixc.xet = ixn.xet = 0;
    w_idx_load(a, cur, li0, ixc);
    WDesc nxt = w_iter_next(it, lane0);
    w_idx_load(a, nxt, li0, ixn);

    // what is in flight for a unit before its edge stage starts: the per-edge chunk, the row exponents of its node rows, the first
    // W_PDG chunks of x_i; and the row addresses the remaining gathers are made from
    struct Pre {
        unsigned long long rowp[2][2];
        int xf[2][2];
        rr_f4 z[2][2];
    };
    rr_f4 raw[W_PDG + 1][2][2];                        // ring of gathered chunks [chunk % 4][sub-block][16-byte half of the lane's 8 columns]
    auto pre_issue = [&](const WIdx &ix, int lh, Pre &p) {
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
#pragma unroll
            for (int role = 0; role < 2; ++role) {
                const unsigned row = (unsigned)ix.r[sb][role];
                p.rowp[sb][role] = reinterpret_cast<unsigned long long>(a.x) + (unsigned long long)row * (W_DX * 4) + (unsigned long long)(32 * lh);
                p.xf[sb][role] = a.xe[row];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const WQuad &q0 = a.zq[j], &q1 = a.zq[2 + j];
                const unsigned long long base = lh ? q1.base : q0.base;
                const unsigned stride = lh ? q1.stride : q0.stride, role = lh ? q1.role : q0.role;
                const unsigned m1 = 0u - (unsigned)(role == 1u), m2 = 0u - (unsigned)(role == 2u);
                const unsigned row = ((unsigned)ix.r[sb][0] & ~(m1 | m2)) | ((unsigned)ix.r[sb][1] & m1) | ((unsigned)ix.r[sb][2] & m2);
                p.z[sb][j] = *reinterpret_cast<w_gptr>(base + (unsigned long long)row * stride);
            }
        }
#pragma unroll
        for (int c = 0; c < W_PDG; ++c)
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int j = 0; j < 2; ++j) raw[c][sb][j] = *reinterpret_cast<w_gptr>(p.rowp[sb][0] + 64 * c + 16 * j);
    };
    Pre pre;
    pre_issue(ixc, lh0, pre);


    // The fragments of the two node stages stream from L2 once per WORKGROUP and tile round (a CU's path to L2 carries ~30 bytes per
    // cycle: four private streams of 192 KiB per tile cost more than the products they feed): wave w loads fragments 2 w, 2 w + 1
    // of every chunk step five steps ahead into registers, writes them to a three-slot ring in LDS two steps ahead, and all four
    // waves read the step's eight fragments from there.  One barrier per step.
#define W_SHARE(K, J) __builtin_bit_cast(rr_u4, __builtin_amdgcn_raw_buffer_load_b128(wstream, 16 * lane, (8 * (K) + (J)) * 1024 + wave * 2048, 0))
    constexpr unsigned RING = SH::F_LDS * 1024 - 0x20000;       // byte offset of the ring behind ldsb[2]
    typedef __attribute__((address_space(3))) rr_u4 *w_ldsw;

    const unsigned t_start = clk();
    while (true) {
        int li = li0, lh = lh0;
        asm volatile("" : "+v"(li), "+v"(lh));
        const int lane = li + 32 * lh;
        const float *tab = tab0;
        asm volatile("" : "+s"(tab));                    // (loop-invariant loads: hoisted out of the tile loop and spilled otherwise)
        // ---- this wave's tile: its units through the edge stage ----------------------------------------------------------------------
        f32x16 sacc[WB];                               // S^T tiles: row = feature in block, column = target
#pragma unroll
        for (int fb = 0; fb < WB; ++fb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[fb][r] = 0.f;
        unsigned badt = 0;
        const bool tile_valid = cur.valid() != 0;
        const int t_m0 = cur.m0, t_nn = tile_valid ? cur.nn() : 0;
        int pt = ixc.pt, pt1 = ixc.pt1;
        const int xet = ixc.xet;
        if (li >= t_nn) { pt = 0; pt1 = 0; }
        Pre pren;
        WIdx ixu = ixn;
        WDesc nn2; nn2.m0 = 0; nn2.e0 = 0; nn2.pk = 0;
        // the next unit: everything its edge stage needs before the first product, the descriptor after it, that one's row indices
        auto advance = [&]() {
            int lhu = lh;
            asm volatile("" : "+v"(lhu));                // (else the quad selections are hoisted out of every loop and spilled)
            pre_issue(ixn, lhu, pren);
            ixu = ixn;                                   // (pt, pt1, xet of the next unit's tile)
            nn2 = w_iter_next(it, lane);
            w_idx_load(a, nn2, li, ixn);
        };
        auto rotate = [&]() { cur = nxt; nxt = nn2; ixc = ixu; pre = pren; };
        bool more = tile_valid;
        while (more) {
            const unsigned t0 = clk();
            const int ne = cur.ne();
            const bool last = cur.last() != 0;
            float inv[2] = {1.f, 1.f};
            bool bad_e[2] = {false, false};
            f32x16 acc[2][WB];
            if (ne > 0) {
                // =================================================================================================================
                // row scales of the unit's edge rows: exponent fields of x_i, x_j (side array) and of the per-edge columns
                // =================================================================================================================
                float rs[2];
                rr_u4 Ah[2], Al[2];
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) {
                    unsigned m = 0;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const rr_f4 v = pre.z[sb][j];
                        m = max(max(m, __float_as_uint(v.x) & 0x7fffffffu), __float_as_uint(v.y) & 0x7fffffffu);
                        m = max(max(m, __float_as_uint(v.z) & 0x7fffffffu), __float_as_uint(v.w) & 0x7fffffffu);
                    }
                    m = rr_xhalf_max(m);
                    int e = max(max((int)(m >> 23), pre.xf[sb][0]), pre.xf[sb][1]);
                    bad_e[sb] = e >= 255;
                    e = e < 15 ? 15 : (e > 254 ? 254 : e);
                    rs[sb] = __uint_as_float((unsigned)(268 - e) << 23);
                    inv[sb] = __uint_as_float((unsigned)(e - 14) << 23);
                    unsigned h[4], l[4];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const rr_f4 v = pre.z[sb][j];
                        rr_split2s(v.x, v.y, rs[sb], h[2 * j], l[2 * j]);
                        rr_split2s(v.z, v.w, rs[sb], h[2 * j + 1], l[2 * j + 1]);
                    }
                    Ah[sb] = rr_u4{h[0], h[1], h[2], h[3]};
                    Al[sb] = rr_u4{l[0], l[1], l[2], l[3]};
                }
                const unsigned t1 = clk();
                // =================================================================================================================
                // edge stage: 17 chunk steps x (2 sub-blocks x 4 feature blocks x 3 plane products)
                // =================================================================================================================
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int fb = 0; fb < WB; ++fb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[sb][fb][r] = 0.f;
                unsigned ldsb[3];
                lds_bases(ldsb);
                rr_u4 fr[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) fr[q] = rr_lds_frag(ldsb, SH::F_WE + q);
                auto issue = [&](int c) {                  // chunk c of the gathered node rows: c < 8 from x_i, else x_j
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            raw[c % (W_PDG + 1)][sb][j] = *reinterpret_cast<w_gptr>(pre.rowp[sb][c >> 3] + 64 * (c & 7) + 16 * j);
                };
                rr_u4 Nh[2], Nl[2];
                auto convert = [&](int c) {
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb) {
                        unsigned h[4], l[4];
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const rr_f4 v = raw[c % (W_PDG + 1)][sb][j];
                            rr_split2s(v.x, v.y, rs[sb], h[2 * j], l[2 * j]);
                            rr_split2s(v.z, v.w, rs[sb], h[2 * j + 1], l[2 * j + 1]);
                        }
                        Nh[sb] = rr_u4{h[0], h[1], h[2], h[3]};
                        Nl[sb] = rr_u4{l[0], l[1], l[2], l[3]};
                    }
                };
#pragma unroll
                for (int s = 0; s < W_NST; ++s) {
                    // step s multiplies chunk s - 1 of the node rows (s = 0: the per-edge chunk); chunk s is converted under it, chunk s + W_PDG issued
                    if (s + W_PDG < 2 * W_NXC) issue(s + W_PDG);
                    // (feature block by feature block, the two sub-blocks' accumulators alternately; a block's two fragments are replaced by
                    //  the next step's as soon as its six products are issued: ten fragments in registers instead of sixteen)
#pragma unroll
                    for (int fb = 0; fb < WB; ++fb) {
                        RR_MFH(Al[0], fr[2 * fb], acc[0][fb]);
                        RR_MFH(Al[1], fr[2 * fb], acc[1][fb]);
                        RR_MFH(Ah[0], fr[2 * fb + 1], acc[0][fb]);
                        RR_MFH(Ah[1], fr[2 * fb + 1], acc[1][fb]);
                        RR_MFH(Ah[0], fr[2 * fb], acc[0][fb]);
                        RR_MFH(Ah[1], fr[2 * fb], acc[1][fb]);
                        if (s + 1 < W_NST) {
                            fr[2 * fb] = rr_lds_frag(ldsb, SH::F_WE + 8 * (s + 1) + 2 * fb);
                            fr[2 * fb + 1] = rr_lds_frag(ldsb, SH::F_WE + 8 * (s + 1) + 2 * fb + 1);
                        }
                    }
                    if (s + 1 < W_NST) convert(s);
                    W_MIX_EDGE()
                    RR_SB();
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb) { Ah[sb] = Nh[sb]; Al[sb] = Nl[sb]; }
                }
                const unsigned t2 = clk();
                if (PROF) { pc[0] += t1 - t0; pc[1] += t2 - t1; }
            }
            const unsigned t2b = clk();
            // ---- inside a tile the next unit's gathers fly under this one's activation (the ring of gathered chunks is free now); behind
            //      the tile's last unit they are issued behind the node stages (registers)
            float cbe[WB];
#pragma unroll
            for (int fb = 0; fb < WB; ++fb) cbe[fb] = tab[32 * fb + li0];
            if (!last) advance();
            if (ne > 0) {
                // =================================================================================================================
                // activation + per-node sums: the activated rows as three bf16 planes into the incidence product
                // =================================================================================================================
                int li = li0, lh = lh0;                   // (made here: everything computed from the lane index is otherwise hoisted in front of
                asm volatile("" : "+v"(li), "+v"(lh));    //  the edge stage and spilled across it)
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) {
                    const unsigned bm = rr_edge_mask(pt, pt1, cur.e0 + 32 * sb);
                    rr_u4 M[2];
                    rr_incidence(bm, lh, M);
                    const unsigned badrows = (unsigned)__builtin_amdgcn_ballot_w64(bad_e[sb]);
                    if (badrows & bm) badt = 1;
                    float iv = inv[sb];
                    if (bad_e[sb]) iv = 0.f;               // (its products are NaN; the clamp below makes them finite, its target is marked)
                    float invr[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) invr[r] = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * rr_crow(r, lh), __float_as_int(iv)));
#pragma unroll
                    for (int fb = 0; fb < WB; ++fb) {
                        const float cb = cbe[fb];
#pragma unroll
                        for (int cc = 0; cc < 2; ++cc) {
                            unsigned y1[4], y2[4], y3[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int r0 = 8 * cc + 2 * q;
                                const float ya = __builtin_amdgcn_fmed3f(fmaf(acc[sb][fb][r0], invr[r0], cb), lo_e, 3.0e38f);
                                const float yb = __builtin_amdgcn_fmed3f(fmaf(acc[sb][fb][r0 + 1], invr[r0 + 1], cb), lo_e, 3.0e38f);
                                rr_split3b(ya, yb, y1[q], y2[q], y3[q]);
                            }
                            const rr_u4 p1 = rr_u4{y1[0], y1[1], y1[2], y1[3]}, p2 = rr_u4{y2[0], y2[1], y2[2], y2[3]}, p3 = rr_u4{y3[0], y3[1], y3[2], y3[3]};
                            RR_MFB(p3, M[cc], sacc[fb]);
                            RR_MFB(p2, M[cc], sacc[fb]);
                            RR_MFB(p1, M[cc], sacc[fb]);
                        }
                    }
                }
            }
            if (PROF) { pc[2] += clk() - t2b; pc[6] += 1; }
            if (last) break;
            rotate();
        }
        // ---- this wave's share of the first five node-stage steps (a wave without a tile still carries its share for the others); the row
        //      scale of node stage 0 and the accumulators' start values are made while they fly --------------------------------------------
        rr_u4 sq[W_SD][2];                             // share of steps k + 2 .. k + 1 + W_SD (registers), slot = step % W_SD
        rr_u4 s01[2][2];                               // share of steps 0 and 1 (written to the ring at the rendezvous)
        f32x16 hacc[WB];
        asm volatile("" : "+v"(li), "+v"(lh));
        // (a wave's loads return in order: what is needed first is requested first)
        rr_u4 bfr[WB];
#pragma unroll
        for (int fbo = 0; fbo < WB; ++fbo) bfr[fbo] = __builtin_bit_cast(rr_u4, __builtin_amdgcn_raw_buffer_load_b128(wstream, 16 * lane, (SH::N_STREAM + fbo) * 1024, 0));
        float cb1[WB];
#pragma unroll
        for (int fb = 0; fb < WB; ++fb) cb1[fb] = tab[3 * 32 * WB + 32 * fb + li];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            s01[0][j] = W_SHARE(0, j); s01[1][j] = W_SHARE(1, j);
#pragma unroll
            for (int k = 2; k < 2 + W_SD; ++k) sq[k % W_SD][j] = W_SHARE(k, j);
        }
        // ---- row scale from max(|S|, |x|, deg) in true units (sacc = 2 se S) ------------------------------------------------
        float ms = 0.f;
#pragma unroll
        for (int fb = 0; fb < WB; ++fb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) ms = fmaxf(fmaxf(fabsf(sacc[fb][r]), fabsf(sacc[fb][r + 1])), ms);
        const float degf = (float)(pt1 - pt);
        const unsigned msb = __float_as_uint(ms);
        int es = (int)(msb >> 23) - 1 - Ee;
        es = msb == 0 ? 0 : es;
        unsigned fld = (unsigned)max(max(max(es, xet), (int)(__float_as_uint(degf) >> 23)), 0);
        fld = rr_xhalf_max(fld);
        bool badrow = msb >= 0x7f800000u || xet >= 255 || badt != 0 || w_bad;
        badrow = rr_xhalf_or(badrow ? 1u : 0u) != 0;
        int e_t = (int)fld;
        e_t = e_t < e_min ? e_min : (e_t > 254 ? 254 : e_t);
        const float rs0 = __uint_as_float((unsigned)(268 - e_t) << 23);
        float fs = rr_pow2(267 - e_t - Ee);
        if (badrow) fs = __uint_as_float(0x7fc00000u);
        // ---- accumulators start at (c0 + deg w_deg) s0 rs0: ONE bf16 product per feature block.  The prepared fragment holds three bf16
        //      planes of c0 2^E0 in k-slots 0..2 and of w_deg 2^E0 in 3..5 and 6..8; this lane's operand rs0 (a power of two: exact in
        //      bf16 at any exponent) in 0..2, the upper 8 bits of deg times rs0 in 3..5, the lower in 6..8 (exact for deg < 65536)
        {
            const float dh = __uint_as_float(__float_as_uint(degf) & 0xffff0000u), dl = degf - dh;
            const unsigned rsb = __float_as_uint(rs0) >> 16, dhb = __float_as_uint(dh * rs0) >> 16, dlb = __float_as_uint(dl * rs0) >> 16;
            const rr_u4 bv = lh ? rr_u4{dlb, 0u, 0u, 0u} : rr_u4{rsb | (rsb << 16), rsb | (dhb << 16), dhb | (dhb << 16), dlb | (dlb << 16)};
#pragma unroll
            for (int fbo = 0; fbo < WB; ++fbo) {
#pragma unroll
                for (int r = 0; r < 16; ++r) hacc[fbo][r] = 0.f;
                RR_MFB(bfr[fbo], bv, hacc[fbo]);
            }
        }
        // ---- rendezvous of the workgroup's four waves: is there a tile at all?  (votes in the first words of the ring) ------------------
        const unsigned tr0 = clk();
        {
            unsigned *vote = reinterpret_cast<unsigned *>(smem + SH::F_LDS * 1024);
            if (lane == 0) vote[wave] = tile_valid ? 1u : 0u;
            lds_barrier();
            const unsigned any = vote[0] | vote[1] | vote[2] | vote[3];
            lds_barrier();
            if (__builtin_amdgcn_readfirstlane(any) == 0) break;
        }
        unsigned ldsb[3];
        lds_bases(ldsb);
        const unsigned wbase = ldsb[2] + RING + (unsigned)wave * 2048u;       // this wave's two fragments inside a ring slot
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int j = 0; j < 2; ++j) *reinterpret_cast<w_ldsw>(wbase + 8192u * k + 1024u * j) = s01[k][j];
        lds_barrier();
        const unsigned tr1 = clk();
        {
            const int nn = t_nn;
            // =====================================================================================================================
            // node stage 0 (transposed): H^T = W0 [S | x]^T, deg with the bias
            // =====================================================================================================================
            f32x16 oacc[WB];
            rr_f4 xr[3][2];                               // ring of the x row's chunks (lane (t, h): columns 16 c + 8 h ..+8)
            int xrow = t_m0 + (li < nn ? li : nn - 1);
            xrow = nn > 0 ? xrow : 0;
            const unsigned long long xp = reinterpret_cast<unsigned long long>(a.x) + (unsigned long long)xrow * (W_DX * 4) + (unsigned long long)(32 * lh);
            auto xissue = [&](int c) {
#pragma unroll
                for (int j = 0; j < 2; ++j) xr[c % 3][j] = *reinterpret_cast<w_gptr>(xp + 64 * c + 16 * j);
            };
            rr_u4 fr[8], nf[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) fr[q] = *reinterpret_cast<rr_ldsp>(ldsb[2] + RING + 1024u * q);
            // per chunk step k of the two stages: this wave's share of step k + 2 goes from registers to the ring, its share of step k + 5
            // is requested, the eight fragments of step k + 1 are read, the twelve products of step k issued; one barrier
            unsigned ph[4], pl[4], nph[4], npl[4];
            float f2 = 0.f, inv2 = 0.f;
            bool anybad = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) rr_split2s(sacc[0][2 * q], sacc[0][2 * q + 1], fs, ph[q], pl[q]);
            unsigned t5 = 0, t5a = 0;
#pragma unroll
            for (int k = 0; k < NSTEP; ++k) {
#ifndef W_ABL_NORINGWRITE
                if (k + 2 < NSTEP)
#else
                if (false)
#endif
                {
#pragma unroll
                    for (int j = 0; j < 2; ++j) *reinterpret_cast<w_ldsw>(wbase + 8192u * ((k + 2) % 3) + 1024u * j) = sq[(k + 2) % W_SD][j];
                }
                if (k + 2 + W_SD < NSTEP) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) sq[(k + 2) % W_SD][j] = W_SHARE(k + 2 + W_SD, j);
                }
#ifdef W_ABL_NORINGREAD
#pragma unroll
                for (int q = 0; q < 8; ++q) nf[q] = fr[q];                                 // (diagnostic build: wrong results)
#else
                if (k + 1 < NSTEP) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) nf[q] = *reinterpret_cast<rr_ldsp>(ldsb[2] + RING + 8192u * ((k + 1) % 3) + 1024u * q);
                }
#endif
                const rr_u4 bh = rr_u4{ph[0], ph[1], ph[2], ph[3]}, bl = rr_u4{pl[0], pl[1], pl[2], pl[3]};
                if (k < NK0) {
                    // ---- node stage 0, chunk k: S (the S^T tiles as operand fragments, 16 features at a time), then x (chunk cx is
                    //      requested at step cx + 5 and converted under step cx + 7) -------------------------------------------------
                    if (k >= 5 && k - 5 < W_NXC) xissue(k - 5);
#pragma unroll
                    for (int fbo = 0; fbo < WB; ++fbo) RR_MFH(fr[2 * fbo], bl, hacc[fbo]);
#pragma unroll
                    for (int fbo = 0; fbo < WB; ++fbo) RR_MFH(fr[2 * fbo + 1], bh, hacc[fbo]);
#pragma unroll
                    for (int fbo = 0; fbo < WB; ++fbo) RR_MFH(fr[2 * fbo], bh, hacc[fbo]);
                    if (k + 1 < NKS) {
                        const int c1 = k + 1, fb1 = c1 >> 1, cc1 = c1 & 1;
#pragma unroll
                        for (int q = 0; q < 4; ++q) rr_split2s(sacc[fb1][8 * cc1 + 2 * q], sacc[fb1][8 * cc1 + 2 * q + 1], fs, nph[q], npl[q]);
                    } else if (k + 1 < NK0) {
                        const int cx = k + 1 - NKS;
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            rr_split2s(xr[cx % 3][j].x, xr[cx % 3][j].y, rs0, nph[2 * j], npl[2 * j]);
                            rr_split2s(xr[cx % 3][j].z, xr[cx % 3][j].w, rs0, nph[2 * j + 1], npl[2 * j + 1]);
                        }
                    }
                    W_MIX(12, 8, 2, 2)
                    RR_SB();
                    if (k + 1 == NK0) {
                        // ---- between the stages: activation, row scale of H, the first 16 hidden features as fragments --------------------
                        t5 = clk();
                        float m2 = 0.f;
                        rr_mfma_settle();
#pragma unroll
                        for (int fbo = 0; fbo < WB; ++fbo)
#pragma unroll
                            for (int r = 0; r < 16; ++r) hacc[fbo][r] = rr_max(hacc[fbo][r], lo_0);
#pragma unroll
                        for (int fbo = 0; fbo < WB; ++fbo)
#pragma unroll
                            for (int r = 0; r < 16; r += 2) m2 = fmaxf(fmaxf(fabsf(hacc[fbo][r]), fabsf(hacc[fbo][r + 1])), m2);
                        const unsigned m2b = rr_xhalf_max(__float_as_uint(m2));
                        int e2 = (int)(m2b >> 23) + (e_t - 141 - E0);
                        e2 = e2 < 15 ? 15 : (e2 > 254 ? 254 : e2);
                        f2 = rr_pow2(e_t - e2 - E0 + 127);
                        inv2 = rr_pow2(e2 - 14 - E1);
                        if (m2b >= 0x7f800000u || badrow) { f2 = __uint_as_float(0x7fc00000u); inv2 = f2; badrow = true; }
                        anybad = __builtin_amdgcn_ballot_w64(badrow) != 0ull;
#pragma unroll
                        for (int q = 0; q < 4; ++q) rr_split2s(hacc[0][2 * q], hacc[0][2 * q + 1], f2, nph[q], npl[q]);
#pragma unroll
                        for (int fb = 0; fb < WB; ++fb)
#pragma unroll
                            for (int r = 0; r < 16; ++r) oacc[fb][r] = 0.f;
                        t5a = clk();
                    }
                } else {
                    // ---- node stage 1, chunk c of the hidden rows: OUT = H W1^T ---------------------------------------------------------
                    const int c = k - NK0;
#pragma unroll
                    for (int fb = 0; fb < WB; ++fb) RR_MFH(bl, fr[2 * fb], oacc[fb]);
#pragma unroll
                    for (int fb = 0; fb < WB; ++fb) RR_MFH(bh, fr[2 * fb + 1], oacc[fb]);
#pragma unroll
                    for (int fb = 0; fb < WB; ++fb) RR_MFH(bh, fr[2 * fb], oacc[fb]);
                    if (c + 1 < NKS) {
                        const int c1 = c + 1, fb1 = c1 >> 1, cc1 = c1 & 1;
#pragma unroll
                        for (int q = 0; q < 4; ++q) rr_split2s(hacc[fb1][8 * cc1 + 2 * q], hacc[fb1][8 * cc1 + 2 * q + 1], f2, nph[q], npl[q]);
                    }
                    W_MIX(12, 8, 2, 2)
                    RR_SB();
                }
                lds_barrier();
#pragma unroll
                for (int q = 0; q < 8; ++q) fr[q] = nf[q];
#pragma unroll
                for (int q = 0; q < 4; ++q) { ph[q] = nph[q]; pl[q] = npl[q]; }
            }
            const unsigned t5b = clk();
            advance();                                  // (the next tile's first gathers fly under the stores)
            const unsigned t5c = clk();
            int li = li0, lh = lh0;
            asm volatile("" : "+v"(li), "+v"(lh));
            // ---- rows leave as 128-byte row segments ---------------------------------------------------------------------------------
            float invr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) invr[r] = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * rr_crow(r, lh), __float_as_int(inv2)));
            const int voff_lane = (4 * lh * 32 * WB + li) * 4;
            const __amdgpu_buffer_rsrc_t orow = __builtin_amdgcn_make_buffer_rsrc(a.out + (int64_t)t_m0 * (32 * WB), 0, nn * (32 * WB * 4), 0x00020000);
            unsigned mrow[16];                          // largest |value| (bits) of this lane's columns in every row of the tile
#pragma unroll
            for (int r = 0; r < 16; ++r) mrow[r] = 0u;
#pragma unroll
            for (int fb = 0; fb < WB; ++fb) {
                const float cb = cb1[fb];
                auto put = [&](auto nanrows) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float y = fmaxf(fmaf(oacc[fb][r], invr[r], cb), lo_1);
                        if (decltype(nanrows)::value) y = invr[r] != invr[r] ? invr[r] : y;     // (the max drops a NaN)
                        mrow[r] = max(mrow[r], __float_as_uint(y) & 0x7fffffffu);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), orow, voff_lane + ((r & 3) + 8 * (r >> 2)) * (32 * WB * 4) + 32 * fb * 4, 0, 0);
                    }
                };
                if (anybad) put(std::true_type{}); else put(std::false_type{});
            }
            if (a.xe_out) {
                // the row exponents of the OUTPUT rows for the layer that takes them as its x (it then skips its pass over them): maximum over
                // the 32 lanes of a half by DPP (row_shr 1, 2, 4, 8: lane 15 of a row of 16 holds the row's maximum; row_bcast15: lane 31 /
                // 63 that of the half), lanes 31 and 63 write their sixteen rows
                const __amdgpu_buffer_rsrc_t erow = __builtin_amdgcn_make_buffer_rsrc(a.xe_out + t_m0, 0, nn * 4, 0x00020000);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    unsigned v = mrow[r];
                    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));
                    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));
                    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));
                    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));
                    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true));
                    if (li == 31) __builtin_amdgcn_raw_buffer_store_b32(v >> 23, erow, ((r & 3) + 8 * (r >> 2) + 4 * lh) * 4, 0, 0);
                }
            }
            if (PROF) { const unsigned t6 = clk(); pc[7] += tile_valid ? 1 : 0; pc[3] += tr1 - tr0; pc[4] += t5 - tr1; pc[5] += t6 - t5; pd[0] += t5a - t5; pd[1] += t5b - t5a; pd[2] += t5c - t5b; pd[3] += t6 - t5c; }
            rotate();
        }
    }
    if (PROF && prof && lane0 == 0 && (range == 0 || range == a.n_ranges / 2)) {
        unsigned long long *o = prof + (range == 0 ? 0 : 16);
        for (int q = 0; q < 8; ++q) o[q] = pc[q];
        o[8] = clk() - t_start;
        for (int q = 0; q < 4; ++q) o[9 + q] = pd[q];
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// exponent field of max |x[v][:]| per node row of 128 floats (255 when the row holds an Inf or a NaN): one half-wave per row
__global__ __launch_bounds__(256) void layer_w_row_exp_kernel(const float *x, int n_nodes, int32_t *xe) {
    const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
    const int64_t hw = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + half, nhw = (int64_t)gridDim.x * 8;
    for (int64_t row0 = hw * 4; row0 < n_nodes; row0 += nhw * 4) {
        unsigned m[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t row = row0 + u < n_nodes ? row0 + u : n_nodes - 1;
            const float4 v = *reinterpret_cast<const float4 *>(x + row * W_DX + 4 * l31);
            m[u] = max(max(__float_as_uint(v.x) & 0x7fffffffu, __float_as_uint(v.y) & 0x7fffffffu), max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu));
        }
#pragma unroll
        for (int d = 1; d < 32; d <<= 1)
#pragma unroll
            for (int u = 0; u < 4; ++u) m[u] = max(m[u], (unsigned)__shfl_xor((int)m[u], d, 64));
        if (l31 < 4 && row0 + l31 < n_nodes) xe[row0 + l31] = (int32_t)((l31 == 0 ? m[0] : (l31 == 1 ? m[1] : (l31 == 2 ? m[2] : m[3]))) >> 23);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Prepared weights: header, fp16 plane fragments of the three stages in the kernel's operand layouts, folded biases.
struct WPrepArgs {
    const float *W[3], *bias[3], *bn_mean[3], *bn_scale[3], *bn_shift[3];
    int k_total[3], act[3];
};

__device__ __forceinline__ float w_prep_bn(const WPrepArgs &p, int st, int row) { return p.bn_scale[st] ? p.bn_scale[st][row] : 1.f; }
__device__ __forceinline__ float w_prep_c0(const WPrepArgs &p, int st, int row) {
    const float b = p.bias[st] ? p.bias[st][row] : 0.f;
    if (!p.bn_scale[st]) return b;
    return (b - p.bn_mean[st][row]) * p.bn_scale[st][row] + p.bn_shift[st][row];
}

__global__ __launch_bounds__(1024) void layer_w_prepare_kernel(WPrepArgs p, unsigned *prep) {
    using SH = WShape;
    constexpr int WB = SH::WB, NKS = SH::NKS, NK0 = SH::NK0, Wd = 32 * WB;
    __shared__ unsigned red[4];                        // max |We|, max |W0|, max |W1|, max |c0 of node stage 0|
    __shared__ int sE[4];
    const int tid = threadIdx.x;
    if (tid < 4) red[tid] = 0;
    __syncthreads();
    for (int st = 0; st < 3; ++st)
        for (int i = tid; i < Wd * p.k_total[st]; i += 1024) {
            const int row = i / p.k_total[st];
            atomicMax(&red[st], __float_as_uint(p.W[st][i] * w_prep_bn(p, st, row)) & 0x7fffffffu);
        }
    if (tid < Wd) {
        atomicMax(&red[3], __float_as_uint(w_prep_c0(p, 1, tid)) & 0x7fffffffu);
        // (the other folded constants only have to be finite)
        const unsigned c = (__float_as_uint(w_prep_c0(p, 0, tid)) & 0x7fffffffu) >= 0x7f800000u || (__float_as_uint(w_prep_c0(p, 2, tid)) & 0x7fffffffu) >= 0x7f800000u;
        if (c) atomicMax(&red[0], 0x7f800000u);
    }
    __syncthreads();
    if (tid < 3) {
        int e = (int)(red[tid] >> 23);
        e = e < 15 ? 15 : (e > 254 ? 254 : e);
        sE[tid] = 141 - e;                              // scale 2^E puts the maximum into [2^14, 2^15)
    }
    __syncthreads();
    const int Ee = sE[0], E0 = sE[1], E1 = sE[2];
    if (tid == 0) {
        const bool bad = red[0] >= 0x7f800000u || red[1] >= 0x7f800000u || red[2] >= 0x7f800000u || red[3] >= 0x7f800000u;
        int emin = 15;
        emin = max(emin, 15 - Ee);
        if (red[3] != 0) emin = max(emin, E0 + 16 + ((int)(red[3] >> 23) - 126));
        emin = min(emin, 254);
        prep[WH_MAGIC] = W_MAGIC; prep[WH_EE] = (unsigned)Ee; prep[WH_E0] = (unsigned)E0; prep[WH_E1] = (unsigned)E1;
        prep[WH_EMIN] = (unsigned)emin; prep[WH_BAD] = bad ? 1u : 0u;
        prep[WH_ACT] = (p.act[0] == 1 ? 1u : 0u) | (p.act[1] == 1 ? 2u : 0u) | (p.act[2] == 1 ? 4u : 0u);
        for (int i = 7; i < W_HDR; ++i) prep[i] = 0;
    }
    rr_u4 *frag = reinterpret_cast<rr_u4 *>(prep + W_HDR);
    for (int i = tid; i < SH::F_ALL * 64; i += 1024) {
        const int f = i >> 6, lane = i & 63, l31 = lane & 31, h = lane >> 5;
        int st, blk, c, plane;
        if (f >= SH::F_WB) {
            // (c0, deg column) of node stage 0 in matrix units as bf16 planes: lane (g, h), k-slot 8 h + s -> 0..2 planes of c0, 3..5 and 6..8 planes of w_deg
            const int row = 32 * (f - SH::F_WB) + l31;
            const float s0 = rr_pow2(E0 + 127);
            const float cv = w_prep_c0(p, 1, row) * s0, wd = p.W[1][(int64_t)row * p.k_total[1] + W_DX + Wd] * w_prep_bn(p, 1, row) * s0;
            unsigned c1, c2, c3, d1, d2, d3;
            rr_split3b(cv, 0.f, c1, c2, c3);
            rr_split3b(wd, 0.f, d1, d2, d3);
            c1 &= 0xffffu; c2 &= 0xffffu; c3 &= 0xffffu; d1 &= 0xffffu; d2 &= 0xffffu; d3 &= 0xffffu;
            frag[i] = h ? rr_u4{d3, 0u, 0u, 0u} : rr_u4{c1 | (c2 << 16), c3 | (d1 << 16), d2 | (d3 << 16), d1 | (d2 << 16)};
            continue;
        }
        if (f < SH::F_W0) { st = 0; const int q = f - SH::F_WE; plane = q & 1; blk = (q >> 1) % WB; c = (q >> 1) / WB; }          // c = chunk step
        else if (f < SH::F_W1) { st = 1; const int q = f - SH::F_W0; plane = q & 1; blk = (q >> 1) % WB; c = (q >> 1) / WB; }
        else { st = 2; const int q = f - SH::F_W1; plane = q & 1; blk = (q >> 1) % WB; c = (q >> 1) / WB; }
        const int row = 32 * blk + l31;
        const float sc = rr_pow2((st == 0 ? Ee : (st == 1 ? E0 : E1)) + 127) * w_prep_bn(p, st, row);
        float wv[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            int k;
            if (st == 0) {
                if (c == 0) { k = 2 * W_DX + 8 * h + s; k = k < p.k_total[0] ? k : -1; }      // the per-edge chunk
                else k = 16 * (c - 1) + 8 * h + s;
            } else if (st == 1) {
                if (c < NKS) k = W_DX + rr_kslot_feature(c, h, s);                             // columns of [x | S | deg 0 0 0]
                else k = 16 * (c - NKS) + 8 * h + s;
            } else k = rr_kslot_feature(c, h, s);
            wv[s] = k >= 0 ? p.W[st][(int64_t)row * p.k_total[st] + k] * sc : 0.f;
        }
        unsigned o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned hi, lo;
            rr_split2(wv[2 * q], wv[2 * q + 1], hi, lo);
            o[q] = plane ? lo : hi;
        }
        frag[i] = rr_u4{o[0], o[1], o[2], o[3]};
    }
    float *tab = reinterpret_cast<float *>(prep + W_HDR + SH::F_ALL * 256);
    if (tid < Wd) {
        tab[tid] = w_prep_c0(p, 0, tid) * rr_pow2(Ee + 127);
        tab[Wd + tid] = w_prep_c0(p, 1, tid);
        tab[2 * Wd + tid] = p.W[1][(int64_t)tid * p.k_total[1] + W_DX + Wd] * w_prep_bn(p, 1, tid);
        tab[3 * Wd + tid] = w_prep_c0(p, 2, tid);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
static bool w_enabled() {
    static const int on = [] { const char *d = getenv("GSN_FUSED_W"); return d ? atoi(d) : 1; }();
    return on != 0;
}

static bool w_stage_ok(const gsn_chain_stage &g, int width) {
    if (!g.W || g.n_out != width) return false;
    if (g.act != 0 && g.act != 1) return false;
    if ((g.bn_scale != nullptr) != (g.bn_shift != nullptr) || (g.bn_scale != nullptr) != (g.bn_mean != nullptr)) return false;
    return true;
}

int w_supported(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0, const gsn_chain_stage *node1) {
    if (!w_enabled() || !edge || !node0 || !node1) return 0;
    const int width = 128;
    if (d_x != W_DX) return 0;
    if (!w_stage_ok(*edge, width) || !w_stage_ok(*node0, width) || !w_stage_ok(*node1, width)) return 0;
    if (edge->n_blocks < 2 || edge->n_blocks > 6 || !edge->blocks) return 0;
    int64_t kz = 0;
    const void *roles[W_MAXROLE] = {nullptr, nullptr, nullptr};
    for (int b = 0; b < edge->n_blocks; ++b) {
        const gsn_block &bl = edge->blocks[b];
        if (!bl.data || !bl.idx32 || bl.idx || bl.width <= 0 || (bl.width & 3)) return 0;
        if (reinterpret_cast<uintptr_t>(bl.data) & 15) return 0;
        if (b < 2) {
            if (bl.width != W_DX) return 0;
            roles[b] = bl.idx32;
        } else {
            kz += bl.width;
            if (bl.idx32 != roles[0] && bl.idx32 != roles[1]) {
                if (roles[2] && roles[2] != bl.idx32) return 0;
                roles[2] = bl.idx32;
            }
        }
    }
    if (edge->blocks[0].data != edge->blocks[1].data) return 0;        // (both are x: one side array of row exponents)
    if (kz > 16) return 0;
    if (node0->n_blocks != 0 || node1->n_blocks != 0) return 0;
    return 1;
}

int64_t w_prepared_bytes(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0, const gsn_chain_stage *node1) {
    if (!w_supported(edge, d_x, node0, node1)) return 0;
    return (int64_t)WShape::PREP_WORDS * 4;
}

int w_prepare(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0, const gsn_chain_stage *node1, void *prepared, hipStream_t st) {
    WPrepArgs p{};
    const gsn_chain_stage *gs[3] = {edge, node0, node1};
    int ke = 0;
    for (int b = 0; b < edge->n_blocks; ++b) ke += (int)edge->blocks[b].width;
    const int kt[3] = {ke, (int)(d_x + edge->n_out + 4), (int)node0->n_out};
    for (int s = 0; s < 3; ++s) {
        p.W[s] = gs[s]->W; p.bias[s] = gs[s]->bias; p.bn_mean[s] = gs[s]->bn_mean; p.bn_scale[s] = gs[s]->bn_scale; p.bn_shift[s] = gs[s]->bn_shift;
        p.k_total[s] = kt[s]; p.act[s] = gs[s]->act;
    }
    hipLaunchKernelGGL(layer_w_prepare_kernel, dim3(1), dim3(1024), 0, st, p, reinterpret_cast<unsigned *>(prepared));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "layer_w_prepare_kernel: %s", hipGetErrorString(e));
    return GSN_OK;
}

int w_row_exponents(int64_t n_nodes, const float *x, int32_t *row_exp, hipStream_t st) {
    if (n_nodes <= 0) return GSN_OK;
    int64_t gb = (n_nodes + 31) / 32;
    gb = gb < 1 ? 1 : (gb > 2048 ? 2048 : gb);
    if (getenv("GSN_CHAIN_TRACE")) fprintf(stderr, "gsn chain: layer_w_row_exp_kernel nodes %lld\n", (long long)n_nodes);
    hipLaunchKernelGGL(layer_w_row_exp_kernel, dim3((unsigned)gb), dim3(256), 0, st, x, (int)n_nodes, row_exp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "layer_w_row_exp_kernel: %s", hipGetErrorString(e));
    return GSN_OK;
}

int w_forward(int64_t n_nodes, int64_t n_edges, const int32_t *seg_ptr, const gsn_chain_stage *edge, const float *x, int64_t d_x,
              const gsn_chain_stage *node0, const gsn_chain_stage *node1, const void *prepared, float *out, int32_t *row_exp,
              const int32_t *x_row_exp, int32_t *out_row_exp, hipStream_t st) {
    using SH = WShape;
    (void)d_x; (void)node0; (void)node1;
    if (edge->blocks[0].data != x) return set_error(GSN_E_UNSUPPORTED, "gsn_layer_fused_fwd_hip: the wide kernel takes x itself as its first two edge blocks");
    if (!row_exp && !x_row_exp) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
            return set_error(GSN_E_UNSUPPORTED, "gsn_layer_fused_fwd_hip: without a workspace the d = 128 kernel allocates its row-exponent array per call (no stream capture): gsn_layer_fused_fwd_ws_hip");
    }
    WArgs a{};
    a.n_nodes = (int)n_nodes; a.n_edges = (int)n_edges; a.seg_ptr = seg_ptr;
    a.x = x; a.out = out; a.prep = reinterpret_cast<const unsigned *>(prepared);
    a.ridx[0] = edge->blocks[0].idx32; a.ridx[1] = edge->blocks[1].idx32; a.ridx[2] = nullptr;
    int nq = 0;
    for (int b = 2; b < edge->n_blocks; ++b) {
        const gsn_block &bl = edge->blocks[b];
        unsigned role;
        if (bl.idx32 == a.ridx[0]) role = 0;
        else if (bl.idx32 == a.ridx[1]) role = 1;
        else { a.ridx[2] = bl.idx32; role = 2; }
        for (int q = 0; q < (int)bl.width / 4; ++q) {
            a.zq[nq].base = reinterpret_cast<unsigned long long>(bl.data) + 16ull * q;
            a.zq[nq].stride = (unsigned)(bl.width * 4);
            a.zq[nq].role = role;
            ++nq;
        }
    }
    if (!a.ridx[2]) a.ridx[2] = a.ridx[0];
    for (int q = nq; q < 4; ++q) {                        // columns past K: finite data of the same rows (their weights are zero)
        if (nq > 0) a.zq[q] = a.zq[0];
        else { a.zq[q].base = reinterpret_cast<unsigned long long>(x); a.zq[q].stride = W_DX * 4; a.zq[q].role = 0; }
    }
    if (n_edges == 0)
        for (int q = 0; q < 4; ++q) { a.zq[q].base = reinterpret_cast<unsigned long long>(x); a.zq[q].stride = 0; a.zq[q].role = 0; }
    const int64_t n_tiles = (n_nodes + W_TN - 1) / W_TN;
    int64_t gx = 256;
    { const char *d = getenv("GSN_FUSED_GRID"); if (d && atoi(d) > 0) gx = atoi(d); }
    int64_t ranges = gx * 4;
    if (ranges > n_tiles) ranges = n_tiles;
    if (gx > ranges) gx = ranges;
    a.n_ranges = (int)ranges;
    static const bool prof_on = [] { const char *d = getenv("GSN_FUSED_PROF"); return d && atoi(d) != 0; }();
    static DeviceOnce attr_set;
    const int attr_dev = current_device();
    if (!attr_set.done(attr_dev)) {
        hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(&layer_fused_kernel_w<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e0 == hipSuccess) e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(&layer_fused_kernel_w<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e0 != hipSuccess) return set_error(GSN_E_HIP, "hipFuncSetAttribute(layer_fused_kernel_w): %s", hipGetErrorString(e0));
        attr_set.mark(attr_dev);
    }
    static_assert(SH::LDS_BYTES <= 160 * 1024, "LDS budget");
    // the row exponents of x: one int per node, stream-ordered scratch
    int32_t *xe = row_exp;
    if (!xe && !x_row_exp) {
        hipError_t em = hipMallocAsync(reinterpret_cast<void **>(&xe), (size_t)n_nodes * 4, st);
        if (em != hipSuccess) return set_error(GSN_E_HIP, "hipMallocAsync(row exponents): %s", hipGetErrorString(em));
    }
    a.xe = x_row_exp ? x_row_exp : xe;
    a.xe_out = out_row_exp;
    const bool trace = getenv("GSN_CHAIN_TRACE") != nullptr;
    if (!x_row_exp) {
        int64_t gb = (n_nodes + 31) / 32;
        gb = gb < 1 ? 1 : (gb > 2048 ? 2048 : gb);
        if (trace) fprintf(stderr, "gsn chain: layer_w_row_exp_kernel nodes %d\n", a.n_nodes);
        hipLaunchKernelGGL(layer_w_row_exp_kernel, dim3((unsigned)gb), dim3(256), 0, st, x, a.n_nodes, xe);
    }
    if (trace) fprintf(stderr, "gsn chain: layer_fused_kernel_w nodes %d edges %d grid %lld ranges %d\n", a.n_nodes, a.n_edges, (long long)gx, a.n_ranges);
    if (prof_on) {
        unsigned long long *prof = nullptr;
        (void)hipMalloc(&prof, 32 * 8); (void)hipMemsetAsync(prof, 0, 32 * 8, st);
        hipLaunchKernelGGL((layer_fused_kernel_w<true>), dim3((unsigned)gx), dim3(256), SH::LDS_BYTES, st, a, prof);
        unsigned long long h[32];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost);
        (void)hipFree(prof);
        static int shown = 0;
        if (shown++ % 8 == 7)
            for (int w = 0; w < 2; ++w) {
                const unsigned long long *o = h + 16 * w;
                const double nu = o[6] ? (double)o[6] : 1.0, nt = o[7] ? (double)o[7] : 1.0;
                fprintf(stderr, "wprof range %s: units %llu tiles %llu total %llu cycles | per unit: scales %.0f edge loop %.0f epilogue %.0f rendezvous %.0f | per tile: stage0 %.0f stage1 %.0f (between the stages %.0f, steps %.0f, advance %.0f, stores %.0f)\n",
                        w ? "mid" : "0", o[6], o[7], o[8], o[0] / nu, o[1] / nu, o[2] / nu, o[3] / nu, o[4] / nt, o[5] / nt, o[9] / nt, o[10] / nt, o[11] / nt, o[12] / nt);
            }
    } else {
        hipLaunchKernelGGL((layer_fused_kernel_w<false>), dim3((unsigned)gx), dim3(256), SH::LDS_BYTES, st, a, (unsigned long long *)nullptr);
    }
    hipError_t e = hipGetLastError();
    if (!row_exp && !x_row_exp) (void)hipFreeAsync(xe, st);
    if (e != hipSuccess) return set_error(GSN_E_HIP, "layer_fused_kernel_w: %s", hipGetErrorString(e));
    return GSN_OK;
}

}  // namespace gsn
