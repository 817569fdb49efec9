// The one-launch `general` layer with every intermediate in REGISTERS (GSN_sparse.py:93-176 / GSN_edge_sparse.py:82-170 and the
// MPNN twins; same operation, same arithmetic as layer_fused.hip, which remains the kernel for the shapes this one does not take):
//
//     r_e  = act_e( bn_e( cat(x_i, x_j, ids.., e) W1^T + b1 ) )              per edge      (edge stage, K_e <= 80)
//     S_v  = sum_{e -> v} r_e                                                per node      (torch.sparse.sum, :140-143)
//     h_v  = act_0( bn_0( [x_v | S_v | deg_v] W0'^T + b0 ) )                 per node
//     out_v = act_1( bn_1( h_v W1'^T + b1' ) )                               per node
//
// Why a second kernel.  layer_fused.hip keeps a stage's WEIGHTS in the registers of four waves and moves the rows through LDS:
// every 32 x 16 operand fragment is read by four waves, three role groups meet at two barriers per step, and the step takes
// ~7 150 cycles of which the matrix pipe works 2 370 (profiles/r02_fused_*).  Here it is the other way round: the WEIGHTS sit
// in LDS (152 KiB of prepared fp16 plane fragments, read with one conflict-free ds_read_b128 per fragment; the low plane of the
// last stage streams from L2), every wave owns a contiguous range of NODES and takes tiles of <= 32 nodes / <= 64 in-edges through
// all four stages by itself, and a tile's rows never leave the wave's registers:
//
//   * gfx950's 32x32x16 MFMA has the same lane layout for its A and B operands (lane l: row / column l & 31, k = 8 (l >> 5) ..+8)
//     and its C layout gives a lane 16 values of ONE column.  So an accumulator tile, converted to fp16 planes in place, IS an
//     operand fragment of the next product whose contraction runs over the accumulator's ROW index -- with the k-slots in the
//     order (r & 3) + 8 (r >> 2) + 4 (l >> 5), which the prepared weight fragments follow (scripts/emulate_layer_rr.py checks
//     every index formula of this file on the CPU).
//   * edge stage:  Y[e][f]   = Z[e][:] We^T      A = gathered rows (straight from global memory into the operand layout: lane
//                                                (e, h) loads columns 16 c + 8 h .. + 8 of its edge's concatenated row), B = We.
//   * per-node sums: S^T[f][t] = sum_e Y^T[f][e] M[e][t]   the activated rows as the A operand, M the 0 / 2.0 incidence matrix of
//                                                the tile (edge e enters target t), built from seg_ptr with four bit operations
//                                                per register.  No LDS, no atomics, fp32 accumulation inside the MFMA.
//   * node stage 0 (transposed): H^T[g][t] = W0[g][:] IN^T[:][t]   B = [S | x | deg] of target t from the S^T accumulators
//   * node stage 1: OUT[t][f] = H[t][:] W1^T     A = the H^T accumulators, C has 32 consecutive floats of a row in 32 lanes.
//   No barrier after the prologue; a wave waits only for its own loads.  2 waves per SIMD (<= 256 registers), 8 waves per CU.
//
// Matrix arithmetic: fp16x3 as in layer_fused.hip (two fp16 planes per operand after an exact power-of-two scaling of every
// input row and of every weight matrix, x w ~ x_h w_h + x_h w_l + x_l w_h, fp32 accumulation), two products for edge rows that
// are exact in fp16 (one-hot / small-integer encodings).  The incidence product takes the activated rows as two fp16 planes
// under the edge stage's weight scale when the chunk's input rows are exact and < 2 in magnitude (then |Y| is bounded by the
// weights' row sums, which the scale is made from) and as three bf16 planes (exact, any magnitude) otherwise.
// Non-finite values: as layer_fused.hip -- an edge / node / hidden row that holds an Inf or a NaN makes exactly the output rows
// that see it NaN (the activated rows are clamped to finite values before the incidence product -- 0 x NaN would reach every
// target of the tile -- and the targets of such rows are marked instead).
#include <hip/hip_runtime.h>

#include <cstdint>

#ifndef RR_STORE_AUX
#define RR_STORE_AUX 2        // cache policy bits of the output stores (2: nt -- the rows are written once and read by the NEXT launch; as in layer_rp.hip)
#endif
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "chain_common.h"
#include "layer_rr.h"
#include "layer_rr_inl.h"
#include "layer_rr_core.h"

namespace gsn {

// what one half of the lanes loads for one 16-byte slot of an edge row: address = base + row_index[role] * stride
struct RrSlotHalf { unsigned long long base; unsigned stride, role; };

struct RrArgs {
    int n_nodes, n_edges;
    const int32_t *seg_ptr;
    const int32_t *role_idx[RR_MAXROLE];
    RrSlotHalf slot[RR_NSLOT][2];
    const float *x;
    int d_x;
    float *out;
    const unsigned *prep;
    int n_ranges;                  // node ranges (one per wave slot; slot = wave * gridDim.x + blockIdx.x)
};

// what a block needs before its gathers can be issued / its incidence operand built: the row indices of its edge rows, the
// segment bounds of the lane's target
struct RrIdx {
    int r[RR_MAXROLE];
    int pt, pt1;
};

__device__ __forceinline__ void rr_idx_load(const RrArgs &a, const RrDesc &d, int li, RrIdx &ix) {
    const int ne = d.ne();
    if (d.valid() && ne > 0) {
        const int e = d.e0 + (li < ne ? li : ne - 1);
#pragma unroll
        for (int q = 0; q < RR_MAXROLE; ++q) ix.r[q] = a.role_idx[q][e];
    }
    if (d.valid()) {
        int t = d.m0 + li;
        t = t < a.n_nodes ? t : a.n_nodes - 1;          // (lanes past nn are masked when used)
        ix.pt = a.seg_ptr[t];
        ix.pt1 = a.seg_ptr[t + 1];
    }
}

// the 16-byte loads of one edge block: lane (e, h) reads slot s of its row = columns 8 s' of the concatenated row, from whichever
// block holds them (table in LDS: base, row stride and index role per slot and lane half).  Always issued -- with the row indices
// of an earlier block when there is no next one -- so that the destination registers are dead between their last use and here.
__device__ __forceinline__ void rr_gather_issue(const rr_u4 *slot_tab, int lh, const int (&r)[RR_MAXROLE], rr_f4 (&g)[RR_NSLOT]) {
    rr_u4 ent[RR_NSLOT];
#pragma unroll
    for (int s = 0; s < RR_NSLOT; ++s) ent[s] = slot_tab[2 * s + lh];           // (all table reads first: one LDS latency, not ten)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < RR_NSLOT; ++s) {
        const unsigned m1 = 0u - (unsigned)(ent[s].w == 1u), m2 = 0u - (unsigned)(ent[s].w == 2u);      // (a select chain became divergent branches)
        const unsigned row = ((unsigned)r[0] & ~(m1 | m2)) | ((unsigned)r[1] & m1) | ((unsigned)r[2] & m2);
        const unsigned long long addr = ((unsigned long long)ent[s].y << 32 | ent[s].x) + (unsigned long long)row * ent[s].z;
#ifdef RR_ABL_NOGATHER
        g[s] = rr_f4{(float)(addr & 1), 0.f, 1.f, 0.f};
#else
        g[s] = *reinterpret_cast<const __attribute__((address_space(1))) rr_f4 *>(addr);
#endif
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
template <int WB, int NKX, bool PROF>
__global__ __launch_bounds__(64 * RR_NW) __attribute__((amdgpu_waves_per_eu(RR_NW / 4, RR_NW / 4))) void layer_fused_kernel_rr(RrArgs a, unsigned long long *prof) {
    // diagnostic build (GSN_FUSED_PROF=1): cycles per phase of wave 0 of workgroup 0 and of one wave in the middle of the grid
    auto clk = [&]() -> unsigned {
        if (!PROF) return 0u;
        __builtin_amdgcn_sched_barrier(0);
        const unsigned v = (unsigned)__builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
        return v;
    };
    unsigned pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // wait+convert, issue, edge, stage 0, split, stage 1, blocks, tiles
    unsigned pw[4] = {0, 0, 0, 0};               // explicit waits (diagnostic build): gathers at the block top, x rows, everything in flight before advance(), LDS at the top of stage 1
    auto wait_all = [&](int which) { if (PROF) { const unsigned a0 = clk(); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); pw[which] += clk() - a0; } };
    using SH = RrShape<WB, NKX>;
    constexpr int NKS = SH::NKS, NK0 = SH::NK0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    rr_u4 *slot_tab0 = reinterpret_cast<rr_u4 *>(smem + SH::F_LDS * 1024 + SH::TAB_WORDS * 4);   // (tables behind the fragments: c0 of the three stages, gather slots)

    const int tid = threadIdx.x;
    const int lane0 = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li0 = lane0 & 31, lh0 = lane0 >> 5;
    unsigned ldsb[3];
    ldsb[0] = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem + 16u * (unsigned)lane0;
    ldsb[1] = ldsb[0] + 0x10000u; ldsb[2] = ldsb[0] + 0x20000u;
    asm volatile("" : "+v"(ldsb[1]), "+v"(ldsb[2]));          // (opaque: see rr_lds_frag)

    // ---- prologue: weights -> LDS ---------------------------------------------------------------------------------------------
    {
        const rr_u4 *src = reinterpret_cast<const rr_u4 *>(a.prep + RR_HDR);
        rr_u4 *dst = reinterpret_cast<rr_u4 *>(smem);
        for (int i = tid; i < SH::F_LDS * 64; i += 64 * RR_NW) dst[i] = src[i];
        const float *tsrc = reinterpret_cast<const float *>(a.prep + RR_HDR + SH::F_ALL * 256);
        float *tdst = reinterpret_cast<float *>(smem + SH::F_LDS * 1024);
        for (int i = tid; i < SH::TAB_WORDS; i += 64 * RR_NW) tdst[i] = tsrc[i];
        if (tid < RR_NSLOT * 2) {
            const RrSlotHalf &sh = a.slot[tid >> 1][tid & 1];
            slot_tab0[tid] = rr_u4{(unsigned)sh.base, (unsigned)(sh.base >> 32), sh.stride, sh.role};
        }
    }
    __syncthreads();
    const int Ee = (int)a.prep[RRH_EE], E0 = (int)a.prep[RRH_E0], E1 = (int)a.prep[RRH_E1], e_min = (int)a.prep[RRH_EMIN];
    const bool w_bad = a.prep[RRH_BAD] != 0;
    if (a.prep[RRH_PACK] != 0u) __builtin_trap();              // (prepared for layer_rp.hip: another k-slot order)
    const unsigned acts = a.prep[RRH_ACT];
    // (wave-uniform: scalar registers -- as vector registers they are spilled, and a reload waits for every load in flight)
    auto sgpr = [](float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); };
    const float lo_e = sgpr((acts & 1) ? 0.f : -3.0e38f), lo_0 = sgpr((acts & 2) ? 0.f : -INFINITY), lo_1 = sgpr((acts & 4) ? 0.f : -INFINITY);
    // node stage 0's low S planes stream from L2 (LDS holds 152 KiB of the 184): a buffer resource over them, lane offset + fragment offset
    const __amdgpu_buffer_rsrc_t wstream = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(a.prep) + RR_HDR + SH::F_W0SL * 256, 0, NKS * WB * 1024, 0x00020000);

    // ---- this wave's node range -----------------------------------------------------------------------------------------------
    const int range = wave * (int)gridDim.x + (int)blockIdx.x;
    if (range >= a.n_ranges) return;
    RrIter it;
    it.seg = a.seg_ptr; it.n_nodes = a.n_nodes;
    it.m_next = (int)((int64_t)a.n_nodes * range / a.n_ranges);
    it.m_end = (int)((int64_t)a.n_nodes * (range + 1) / a.n_ranges);
    it.m0 = 0; it.nn = 0; it.eb = 0; it.ee = 0; it.ec = 0; it.pending = 0; it.win = 0;
    rr_iter_load(it, lane0);
    RrDesc cur = rr_iter_next(it, lane0);
    RrIdx ixc, ixn;
#pragma unroll
    for (int q = 0; q < RR_MAXROLE; ++q) { ixc.r[q] = 0; ixn.r[q] = 0; }
    ixc.pt = ixc.pt1 = ixn.pt = ixn.pt1 = 0;
    rr_idx_load(a, cur, li0, ixc);
    RrDesc nxt = rr_iter_next(it, lane0);
    rr_idx_load(a, nxt, li0, ixn);
    rr_f4 g[RR_NSLOT];                                // the gathered rows of the CURRENT block (in flight at the top of the loop)
    rr_gather_issue(slot_tab0, lh0, ixc.r, g);
    int pt = ixc.pt, pt1 = ixc.pt1;

    f32x16 sacc[WB];                                   // S^T tiles: row = feature in block, column = target
#pragma unroll
    for (int fb = 0; fb < WB; ++fb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[fb][r] = 0.f;
    unsigned badt = 0;                                 // this lane's target saw a non-finite edge row
    const int cx = a.d_x >> 2;

#ifdef RR_STAGGER
    // the second wave of every SIMD starts half a tile late: while one is in its edge stage (vector-heavy) the other is in the node
    // stages (matrix-heavy)
    if (wave >= 4) { for (int q = 0; q < RR_STAGGER; ++q) __builtin_amdgcn_s_sleep(127); }
#endif
    const unsigned t_start = clk();
    while (cur.valid()) {
        const unsigned t0 = clk();
        // (every LDS read below is loop-invariant to the compiler, which hoists what it likes out of the loop and then spills it: the
        //  bases are made opaque once per block)
        asm volatile("" : "+v"(ldsb[0]), "+v"(ldsb[1]), "+v"(ldsb[2]));
        // (and so is everything made from the lane index: 16 permute addresses, clamped x offsets, ... -- recomputed where used)
        int li = li0, lh = lh0;
        asm volatile("" : "+v"(li), "+v"(lh));
        const int lane = li + 32 * lh;
        const float *tab = reinterpret_cast<const float *>(rr_lds_generic(ldsb[0] - 16u * (unsigned)lane + SH::F_LDS * 1024));
        const rr_u4 *slot_tab = reinterpret_cast<const rr_u4 *>(tab + SH::TAB_WORDS);
        const int nn = cur.nn(), ne = cur.ne();
        if (li >= nn) { pt = 0; pt1 = 0; }
        // the x rows of the tile's nodes, in the operand layout of node stage 0 (lane (t, h): quads 4 c + 2 h + j); used behind the tile's
        // last block, loaded with every block (a load under `last` here and a use under `last` there would keep the registers alive
        // around the loop): in front of the edge stage when that runs the short path, behind it otherwise (registers)
        float4 xq[NKX][2];
        auto load_x = [&]() {
            const int xrow = cur.m0 + (li < nn ? li : nn - 1);
#pragma unroll
            for (int c = 0; c < NKX; ++c)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    int q = 4 * c + 2 * lh + j;
                    q = q < cx ? q : cx - 1;
                    xq[c][j] = *reinterpret_cast<const float4 *>(a.x + (int64_t)xrow * a.d_x + 4 * q);
                }
        };
        // =========================================================================================================================
        // the gathered rows as fp16 fragments; are they all exact (and < 2)?  Then the next block's gathers take their registers.
        // =========================================================================================================================
        wait_all(0);
        rr_u4 Ah[RR_NKE], Al[RR_NKE];
        unsigned res = 0, big = 0;
#pragma unroll
        for (int c = 0; c < RR_NKE; ++c) {
            unsigned h[4];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const rr_f4 v = g[2 * c + j];
                rr_split2r(v.x, v.y, h[2 * j], res);
                rr_split2r(v.z, v.w, h[2 * j + 1], res);
                big |= __float_as_uint(v.x) | __float_as_uint(v.y) | __float_as_uint(v.z) | __float_as_uint(v.w);
            }
            Ah[c] = rr_u4{h[0], h[1], h[2], h[3]};
            Al[c] = Ah[c];
        }
        const bool inexact = ((res & 0x7fffffffu) | (big & 0x40000000u)) != 0;
        // k-slot 79 (chunk 4, upper lane half, last slot) carries the constant 1 whose weights are the folded bias: the accumulators of
        // the short path start at zero (an inline constant of the first product instead of sixteen moves per 32 x 32 tile)
        Ah[RR_NKE - 1][3] = lh ? (Ah[RR_NKE - 1][3] & 0xffffu) | 0x3c000000u : Ah[RR_NKE - 1][3];
        const bool exact_blk = ne == 0 || __builtin_amdgcn_ballot_w64(inexact) == 0ull;
        float inv_e = 1.f;
        bool bad_e = false;
#ifndef RR_EXP_NOSLOW
        if (!exact_blk) {
            // rows that need the power-of-two row scale: both planes of the scaled rows
            unsigned m = 0;
#pragma unroll
            for (int s = 0; s < RR_NSLOT; ++s) {
                const rr_f4 v = g[s];
                m = max(max(m, __float_as_uint(v.x) & 0x7fffffffu), __float_as_uint(v.y) & 0x7fffffffu);
                m = max(max(m, __float_as_uint(v.z) & 0x7fffffffu), __float_as_uint(v.w) & 0x7fffffffu);
            }
            m = rr_xhalf_max(m);
            int e = (int)(m >> 23);
            e = e < 15 ? 15 : (e > 254 ? 254 : e);
            const float rs = __uint_as_float((unsigned)(268 - e) << 23);
            inv_e = __uint_as_float((unsigned)(e - 14) << 23);
            bad_e = m >= 0x7f800000u;
#pragma unroll
            for (int c = 0; c < RR_NKE; ++c) {
                unsigned h[4], l[4];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const rr_f4 v = g[2 * c + j];
                    rr_split2s(v.x, v.y, rs, h[2 * j], l[2 * j]);
                    rr_split2s(v.z, v.w, rs, h[2 * j + 1], l[2 * j + 1]);
                }
                Ah[c] = rr_u4{h[0], h[1], h[2], h[3]};
                Al[c] = rr_u4{l[0], l[1], l[2], l[3]};
            }
            // (scaled rows: no constant column -- the row scale has no fp16 value; the bias enters in the epilogue)
            Ah[RR_NKE - 1][3] = lh ? Ah[RR_NKE - 1][3] & 0xffffu : Ah[RR_NKE - 1][3];
            Al[RR_NKE - 1][3] = lh ? Al[RR_NKE - 1][3] & 0xffffu : Al[RR_NKE - 1][3];
        }
#endif
        const unsigned t1 = clk();
        // ---- next block: its gathers, the descriptor after it, that one's row indices.  Inside a tile right here (they fly under this
        //      block's edge stage); behind the tile's last block after node stage 0, whose accumulators need the registers ------------
        int npt = 0, npt1 = 0;
        RrDesc nn2; nn2.m0 = 0; nn2.e0 = 0; nn2.pk = 0;
        auto advance = [&]() {
            wait_all(2);
            rr_gather_issue(slot_tab, lh, ixn.r, g);
            npt = ixn.pt; npt1 = ixn.pt1;
            nn2 = rr_iter_next(it, lane);
            rr_idx_load(a, nn2, li, ixn);
        };
        if (!cur.last() && exact_blk) advance();       // (scaled rows: behind the edge stage, which then needs the registers itself)

        const unsigned t2 = clk();
        // =========================================================================================================================
        // edge stage + per-node sums of this block
        // =========================================================================================================================
        if (ne > 0) {
            const unsigned bm = rr_edge_mask(pt, pt1, cur.e0);
            rr_u4 M[2];
            rr_incidence(bm, lh, M);
#ifndef RR_EXP_NOSLOW
            if (exact_blk)
#endif
            {
                // ---- exact rows: two plane products, Y'' = se Y bounded by 2^15, two fp16 planes into the incidence product.
                //      Two feature blocks at a time = two accumulator chains issued alternately: an MFMA that waits for the accumulator of
                //      the one issued just before it, with anything in between, costs ~75 cycles instead of 32 (guide: +43 for the first
                //      extra issue slot).  Weight fragments of step c + 1 are read before the products of step c.
#pragma unroll
                for (int fp = 0; fp < WB; fp += 2) {
                    f32x16 acc0, acc1;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
                    rr_u4 f0h = rr_lds_frag(ldsb, SH::F_WE + 2 * (fp * RR_NKE)), f0l = rr_lds_frag(ldsb, SH::F_WE + 2 * (fp * RR_NKE) + 1);
                    rr_u4 f1h = rr_lds_frag(ldsb, SH::F_WE + 2 * ((fp + 1) * RR_NKE)), f1l = rr_lds_frag(ldsb, SH::F_WE + 2 * ((fp + 1) * RR_NKE) + 1);
#pragma unroll
                    for (int c = 0; c < RR_NKE; ++c) {
                        rr_u4 n0h = f0h, n0l = f0l, n1h = f1h, n1l = f1l;
                        if (c + 1 < RR_NKE) {
                            n0h = rr_lds_frag(ldsb, SH::F_WE + 2 * (fp * RR_NKE + c + 1)); n0l = rr_lds_frag(ldsb, SH::F_WE + 2 * (fp * RR_NKE + c + 1) + 1);
                            n1h = rr_lds_frag(ldsb, SH::F_WE + 2 * ((fp + 1) * RR_NKE + c + 1)); n1l = rr_lds_frag(ldsb, SH::F_WE + 2 * ((fp + 1) * RR_NKE + c + 1) + 1);
                        }
                        RR_MFH(Ah[c], f0l, acc0);
                        RR_MFH(Ah[c], f1l, acc1);
                        RR_MFH(Ah[c], f0h, acc0);
                        RR_MFH(Ah[c], f1h, acc1);
                        RR_SB();
                        f0h = n0h; f0l = n0l; f1h = n1h; f1l = n1l;
                    }
                    rr_mfma_settle();
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const f32x16 &acc = u ? acc1 : acc0;
                        unsigned yh[8], yl[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) rr_split2(rr_max(acc[2 * q], lo_e), rr_max(acc[2 * q + 1], lo_e), yh[q], yl[q]);
                        // (four products on one accumulator: low planes first, the other chain's split in between)
                        const rr_u4 pl0 = rr_u4{yl[0], yl[1], yl[2], yl[3]}, pl1 = rr_u4{yl[4], yl[5], yl[6], yl[7]};
                        const rr_u4 ph0 = rr_u4{yh[0], yh[1], yh[2], yh[3]}, ph1 = rr_u4{yh[4], yh[5], yh[6], yh[7]};
                        RR_MFH(pl0, M[0], sacc[fp + u]);
                        RR_MFH(pl1, M[1], sacc[fp + u]);
                        RR_MFH(ph0, M[0], sacc[fp + u]);
                        RR_MFH(ph1, M[1], sacc[fp + u]);
                    }
                    RR_SB();
                }
            }
#ifndef RR_EXP_NOSLOW
            else {
                // ---- scaled rows: three plane products; the activated rows as three bf16 planes (exact at any magnitude) -----------
                // targets of non-finite rows (lane e of either half holds row e's flag)
                const unsigned badrows = (unsigned)__builtin_amdgcn_ballot_w64(bad_e);
                if (badrows & bm) badt = 1;
                if (bad_e) inv_e = 0.f;                            // (its products are NaN; the clamp below makes them finite, its target is marked)
#pragma unroll
                for (int fb = 0; fb < WB; ++fb) {
                    f32x16 acc;
                    const float cb = tab[32 * fb + li];
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                    for (int c = 0; c < RR_NKE; ++c) {
                        const rr_u4 bh = rr_lds_frag(ldsb, SH::F_WE + ((fb * RR_NKE + c) * 2)), bl = rr_lds_frag(ldsb, SH::F_WE + ((fb * RR_NKE + c) * 2 + 1));
                        RR_MFH(Al[c], bh, acc);
                        RR_MFH(Ah[c], bl, acc);
                        RR_MFH(Ah[c], bh, acc);
                        RR_SB();
                    }
                    // (eight rows at a time: the inverse row scale travels to the accumulator's REGISTER -- row rr_crow(r, h) of the block)
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
                        unsigned y1[4], y2[4], y3[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int r0 = 8 * cc + 2 * q;
                            const float i0 = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * rr_crow(r0, lh), __float_as_int(inv_e)));
                            const float i1 = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * rr_crow(r0 + 1, lh), __float_as_int(inv_e)));
                            const float ya = __builtin_amdgcn_fmed3f(fmaf(acc[r0], i0, cb), lo_e, 3.0e38f);
                            const float yb = __builtin_amdgcn_fmed3f(fmaf(acc[r0 + 1], i1, cb), lo_e, 3.0e38f);
                            rr_split3b(ya, yb, y1[q], y2[q], y3[q]);
                        }
                        const rr_u4 p1 = rr_u4{y1[0], y1[1], y1[2], y1[3]}, p2 = rr_u4{y2[0], y2[1], y2[2], y2[3]}, p3 = rr_u4{y3[0], y3[1], y3[2], y3[3]};
                        RR_MFB(p3, M[cc], sacc[fb]);
                        RR_MFB(p2, M[cc], sacc[fb]);
                        RR_MFB(p1, M[cc], sacc[fb]);
                        RR_SB();
                    }
                }
            }
#endif
        }

        if (!cur.last() && !exact_blk) advance();
        const unsigned t3 = clk();
        if (PROF) { pc[0] += t1 - t0; pc[1] += t2 - t1; pc[2] += t3 - t2; pc[6] += 1; }

        unsigned t4g = 0, t5g = 0;
        if (cur.last()) {
            // =====================================================================================================================
            // node stage 0 (transposed): H^T = W0 [S | x | deg]^T
            // =====================================================================================================================
            f32x16 hacc[WB];
            load_x();                                                   // (rows this tile's edge blocks have just gathered: cache hits)
            wait_all(1);
            // ---- row scale from max(|S|, |x|, deg) in true units (sacc = 2 se S) ------------------------------------------------
            float ms = 0.f;
#pragma unroll
            for (int fb = 0; fb < WB; ++fb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) ms = fmaxf(fmaxf(fabsf(sacc[fb][r]), fabsf(sacc[fb][r + 1])), ms);
            unsigned mx = 0;
            const float degf = (float)(pt1 - pt);
            float4 xv[NKX][2];
#pragma unroll
            for (int c = 0; c < NKX; ++c)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int q = 4 * c + 2 * lh + j;
                    float4 v = xq[c][j];
                    if (q == cx) v = make_float4(degf, 0.f, 0.f, 0.f);
                    if (q > cx || li >= nn) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    xv[c][j] = v;
                    mx = max(max(mx, __float_as_uint(v.x) & 0x7fffffffu), __float_as_uint(v.y) & 0x7fffffffu);
                    mx = max(max(mx, __float_as_uint(v.z) & 0x7fffffffu), __float_as_uint(v.w) & 0x7fffffffu);
                }
            const unsigned msb = __float_as_uint(ms);
            int es = (int)(msb >> 23) - 1 - Ee;                       // exponent field of the largest |S| in true units
            es = msb == 0 ? 0 : es;
            unsigned fld = (unsigned)max(max(es, (int)(mx >> 23)), 0);
            fld = rr_xhalf_max(fld);
            bool badrow = msb >= 0x7f800000u || mx >= 0x7f800000u || badt != 0 || w_bad;
            badrow = rr_xhalf_or(badrow ? 1u : 0u) != 0;
            int e_t = (int)fld;                                        // exponent field the row scale of target li is made from
            e_t = e_t < e_min ? e_min : (e_t > 254 ? 254 : e_t);
            const float rs = __uint_as_float((unsigned)(268 - e_t) << 23);     // 2^(141 - e_t)
            float fs = rr_pow2(267 - e_t - Ee);                                // rs / (2 se): sacc -> scaled planes
            const float sc = rr_pow2(E0 + 268 - e_t);                          // s0 rs: the bias in accumulator units
            if (badrow) fs = __uint_as_float(0x7fc00000u);
            // ---- the [x | deg] fragments ------------------------------------------------------------------------------------
            rr_u4 Xh[NKX], Xl[NKX];
#pragma unroll
            for (int c = 0; c < NKX; ++c) {
                unsigned h[4], l[4];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    rr_split2s(xv[c][j].x, xv[c][j].y, rs, h[2 * j], l[2 * j]);
                    rr_split2s(xv[c][j].z, xv[c][j].w, rs, h[2 * j + 1], l[2 * j + 1]);
                }
                Xh[c] = rr_u4{h[0], h[1], h[2], h[3]};
                Xl[c] = rr_u4{l[0], l[1], l[2], l[3]};
            }
            // ---- accumulators start at the bias (row = hidden feature: per register; column = target: this lane's scale) -----
#pragma unroll
            for (int fbo = 0; fbo < WB; ++fbo)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 cv = *reinterpret_cast<const float4 *>(tab + 32 * WB + 32 * fbo + 8 * i + 4 * lh);
                    hacc[fbo][4 * i] = cv.x * sc; hacc[fbo][4 * i + 1] = cv.y * sc; hacc[fbo][4 * i + 2] = cv.z * sc; hacc[fbo][4 * i + 3] = cv.w * sc;
                }
            // ---- [x | deg] part first (its fragments' registers are free for the rest of the stage).  Everywhere in this stage: hidden
            //      feature blocks in pairs = two accumulator chains issued alternately (see the edge stage), fragments one step ahead ----
            {
                auto fr = [&](int i, rr_u4 &h0, rr_u4 &l0, rr_u4 &h1, rr_u4 &l1) {          // step i = (cq, pair)
                    const int cq = i / (WB / 2), fa = 2 * (i % (WB / 2));
                    h0 = rr_lds_frag(ldsb, SH::F_W0H + fa * NK0 + NKS + cq); l0 = rr_lds_frag(ldsb, SH::F_W0XL + fa * NKX + cq);
                    h1 = rr_lds_frag(ldsb, SH::F_W0H + (fa + 1) * NK0 + NKS + cq); l1 = rr_lds_frag(ldsb, SH::F_W0XL + (fa + 1) * NKX + cq);
                };
                rr_u4 a0h, a0l, a1h, a1l;
                fr(0, a0h, a0l, a1h, a1l);
#pragma unroll
                for (int i = 0; i < NKX * (WB / 2); ++i) {
                    const int cq = i / (WB / 2), fa = 2 * (i % (WB / 2));
                    rr_u4 n0h = a0h, n0l = a0l, n1h = a1h, n1l = a1l;
                    if (i + 1 < NKX * (WB / 2)) fr(i + 1, n0h, n0l, n1h, n1l);
                    RR_MFH(a0h, Xl[cq], hacc[fa]);
                    RR_MFH(a1h, Xl[cq], hacc[fa + 1]);
                    RR_MFH(a0l, Xh[cq], hacc[fa]);
                    RR_MFH(a1l, Xh[cq], hacc[fa + 1]);
                    RR_MFH(a0h, Xh[cq], hacc[fa]);
                    RR_MFH(a1h, Xh[cq], hacc[fa + 1]);
                    RR_SB();
                    a0h = n0h; a0l = n0l; a1h = n1h; a1l = n1l;
                }
            }
            // ---- S part: the S^T tiles become operand fragments, 16 features at a time; the weights' low planes arrive from L2, PD
            //      fragments ahead (nothing else of this wave is in flight here: the next block's gathers are issued behind this stage);
            //      the high planes one step ahead from LDS; the split of chunk c + 1 is spread under the products of chunk c
#ifdef RR_ABL_NOSTREAM
#define RR_STREAM(I) rr_u4{(unsigned)(I), (unsigned)lane, 0x3c003c00u, 0u}
#else
#define RR_STREAM(I) __builtin_bit_cast(rr_u4, __builtin_amdgcn_raw_buffer_load_b128(wstream, 16 * lane, (I) * 1024, 0))
#endif
            constexpr int PD = RR_PD, NSL = NKS * WB;
            static_assert(PD % 2 == 0, "the stream is consumed two fragments per step");
            rr_u4 ql[PD];
#pragma unroll
            for (int i = 0; i < PD; ++i) ql[i] = RR_STREAM(i);
            {
                rr_u4 a0h = rr_lds_frag(ldsb, SH::F_W0H), a1h = rr_lds_frag(ldsb, SH::F_W0H + NK0);
                unsigned ph[4], pl[4], nph[4], npl[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) rr_split2s(sacc[0][2 * q], sacc[0][2 * q + 1], fs, ph[q], pl[q]);
#pragma unroll
                for (int c = 0; c < NKS; ++c) {
                    const rr_u4 bh = rr_u4{ph[0], ph[1], ph[2], ph[3]}, bl = rr_u4{pl[0], pl[1], pl[2], pl[3]};
#pragma unroll
                    for (int fa = 0; fa < WB; fa += 2) {
                        const int i = c * WB + fa;                 // stream position of (c, fa); (c, fa + 1) follows
                        rr_u4 n0h = a0h, n1h = a1h;
                        if (i + 2 < NSL) {
                            const int c1 = (i + 2) / WB, f1 = (i + 2) % WB;
                            n0h = rr_lds_frag(ldsb, SH::F_W0H + f1 * NK0 + c1); n1h = rr_lds_frag(ldsb, SH::F_W0H + (f1 + 1) * NK0 + c1);
                        }
                        const rr_u4 a0l = ql[i % PD], a1l = ql[(i + 1) % PD];
                        RR_MFH(a0h, bl, hacc[fa]);
                        RR_MFH(a1h, bl, hacc[fa + 1]);
                        RR_MFH(a0l, bh, hacc[fa]);
                        RR_MFH(a1l, bh, hacc[fa + 1]);
                        RR_MFH(a0h, bh, hacc[fa]);
                        RR_MFH(a1h, bh, hacc[fa + 1]);
                        if (i + PD < NSL) {
                            ql[i % PD] = RR_STREAM(i + PD);
                            ql[(i + 1) % PD] = RR_STREAM(i + 1 + PD);
                        }
                        if (c + 1 < NKS) {                         // half of the next chunk's planes
                            const int c1 = c + 1, fb1 = c1 >> 1, cc1 = c1 & 1;
                            rr_split2s(sacc[fb1][8 * cc1 + 2 * fa], sacc[fb1][8 * cc1 + 2 * fa + 1], fs, nph[fa], npl[fa]);
                            rr_split2s(sacc[fb1][8 * cc1 + 2 * fa + 2], sacc[fb1][8 * cc1 + 2 * fa + 3], fs, nph[fa + 1], npl[fa + 1]);
                        }
                        RR_SB();
                        a0h = n0h; a1h = n1h;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) { ph[q] = nph[q]; pl[q] = npl[q]; }
                }
            }
            // (the sums are consumed: zero for the next tile HERE, not under a condition at the top of the loop -- the compiler cannot
            //  know that a tile's last block is followed by a first one and would keep the 64 registers alive through node stage 1)
#pragma unroll
            for (int fb = 0; fb < WB; ++fb)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[fb][r] = 0.f;
            badt = 0;
            const unsigned t4 = clk(); t4g = t4;
            // =====================================================================================================================
            // node stage 1: OUT = H W1^T, rows leave as 128-byte row segments
            // =====================================================================================================================
            // ---- activation, row scale of H, fragments ----------------------------------------------------------------------------
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
            int e2 = (int)(m2b >> 23) + (e_t - 141 - E0);                // exponent field of the largest |h| in true units
            e2 = e2 < 15 ? 15 : (e2 > 254 ? 254 : e2);
            float f2 = rr_pow2(e_t - e2 - E0 + 127);                     // accumulator units -> scaled planes
            float inv2 = rr_pow2(e2 - 14 - E1);                          // 1 / (row scale x matrix scale of stage 1)
            if (m2b >= 0x7f800000u || badrow) { f2 = __uint_as_float(0x7fc00000u); inv2 = f2; badrow = true; }
            const bool anybad = __builtin_amdgcn_ballot_w64(badrow) != 0ull;
            // (the fp16 planes of H are made 16 hidden features at a time UNDER the first products of node stage 1, below)
            rr_u4 Hh[NKS], Hl[NKS];
            auto hsplit = [&](int c) {
                const int fbo = c >> 1, cc = c & 1;
                unsigned h[4], l[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) rr_split2s(hacc[fbo][8 * cc + 2 * q], hacc[fbo][8 * cc + 2 * q + 1], f2, h[q], l[q]);
                Hh[c] = rr_u4{h[0], h[1], h[2], h[3]};
                Hl[c] = rr_u4{l[0], l[1], l[2], l[3]};
            };
            hsplit(0);
            float invr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) invr[r] = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * rr_crow(r, lh), __float_as_int(inv2)));
            const unsigned t5 = clk(); t5g = t5;
            advance();                                                  // (the hidden rows are fp16 fragments now: room for the next block's gathers)
            const int voff_lane = (4 * lh * 32 * WB + li) * 4;          // byte offset of (row 4 h, column li) in an output tile
            const __amdgpu_buffer_rsrc_t orow = __builtin_amdgcn_make_buffer_rsrc(a.out + (int64_t)cur.m0 * (32 * WB), 0, nn * (32 * WB * 4), 0x00020000);
#pragma unroll
            for (int fp = 0; fp < WB; fp += 2) {
                // two output feature blocks = two accumulator chains issued alternately; weight fragments one step ahead
                f32x16 o0, o1;
#pragma unroll
                for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
                rr_u4 b0h = rr_lds_frag(ldsb, SH::F_W1 + 2 * (fp * NKS)), b0l = rr_lds_frag(ldsb, SH::F_W1 + 2 * (fp * NKS) + 1);
                rr_u4 b1h = rr_lds_frag(ldsb, SH::F_W1 + 2 * ((fp + 1) * NKS)), b1l = rr_lds_frag(ldsb, SH::F_W1 + 2 * ((fp + 1) * NKS) + 1);
#pragma unroll
                for (int c = 0; c < NKS; ++c) {
                    rr_u4 n0h = b0h, n0l = b0l, n1h = b1h, n1l = b1l;
                    if (c + 1 < NKS) {
                        n0h = rr_lds_frag(ldsb, SH::F_W1 + 2 * (fp * NKS + c + 1)); n0l = rr_lds_frag(ldsb, SH::F_W1 + 2 * (fp * NKS + c + 1) + 1);
                        n1h = rr_lds_frag(ldsb, SH::F_W1 + 2 * ((fp + 1) * NKS + c + 1)); n1l = rr_lds_frag(ldsb, SH::F_W1 + 2 * ((fp + 1) * NKS + c + 1) + 1);
                    }
                    RR_MFH(Hl[c], b0h, o0);
                    RR_MFH(Hl[c], b1h, o1);
                    RR_MFH(Hh[c], b0l, o0);
                    RR_MFH(Hh[c], b1l, o1);
                    RR_MFH(Hh[c], b0h, o0);
                    RR_MFH(Hh[c], b1h, o1);
                    if (fp == 0 && c + 1 < NKS) { hsplit(c + 1); RR_MIX(6, 4) }       // the planes of the next 16 hidden features, under the products
                    RR_SB();
                    b0h = n0h; b0l = n0l; b1h = n1h; b1l = n1l;
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const f32x16 &o = u ? o1 : o0;
                    const float cb = tab[2 * 32 * WB + 32 * (fp + u) + li];
                    auto put = [&](auto nanrows) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float y = rr_max(fmaf(o[r], invr[r], cb), lo_1);
                            if (decltype(nanrows)::value) y = invr[r] != invr[r] ? invr[r] : y;     // (the max drops a NaN)
#ifdef RR_ABL_NOSTORE
                            asm volatile("" :: "v"(y));
                            if (false)
#endif
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), orow, voff_lane + ((r & 3) + 8 * (r >> 2)) * (32 * WB * 4) + 32 * (fp + u) * 4, 0, RR_STORE_AUX);
                        }
                    };
                    if (anybad) put(std::true_type{}); else put(std::false_type{});
                }
                RR_SB();
            }
        }
        if (PROF && cur.last()) { const unsigned t6 = clk(); pc[7] += 1; pc[5] += t6 - t5g; pc[3] += t4g - t3; pc[4] += t5g - t4g; }
        cur = nxt; nxt = nn2;
        pt = npt; pt1 = npt1;
    }
    if (PROF && prof && lane0 == 0 && (range == 0 || range == a.n_ranges / 2)) {
        unsigned long long *o = prof + (range == 0 ? 0 : 16);
        for (int q = 0; q < 8; ++q) o[q] = pc[q];
        o[8] = clk() - t_start;
        for (int q = 0; q < 4; ++q) o[9 + q] = pw[q];
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Prepared weights: header, fp16 plane fragments of the three stages in the kernel's operand layouts, folded biases.  One
// workgroup; run once per weight version.
struct RrPrepArgs {
    const float *W[3], *bias[3], *bn_mean[3], *bn_scale[3], *bn_shift[3];
    int k_total[3], n_out[3], act[3];
    int d_x;
    int pack16;                    // k-slot order of layer_rp.hip
};

__device__ __forceinline__ float rr_prep_bn(const RrPrepArgs &p, int st, int row) { return p.bn_scale[st] ? p.bn_scale[st][row] : 1.f; }
__device__ __forceinline__ float rr_prep_c0(const RrPrepArgs &p, int st, int row) {
    const float b = p.bias[st] ? p.bias[st][row] : 0.f;
    if (!p.bn_scale[st]) return b;
    return (b - p.bn_mean[st][row]) * p.bn_scale[st][row] + p.bn_shift[st][row];
}

template <int WB, int NKX>
__global__ __launch_bounds__(1024) void layer_rr_prepare_kernel(RrPrepArgs p, unsigned *prep) {
    using SH = RrShape<WB, NKX>;
    constexpr int NKS = SH::NKS, NK0 = SH::NK0, Wd = 32 * WB;
    __shared__ unsigned red[4];                        // edge bound, max |W0|, max |W1|, max |c0 of node stage 0|
    __shared__ int sE[4];
    const int tid = threadIdx.x;
    if (tid < 4) red[tid] = 0;
    __syncthreads();
    // ---- maxima ---------------------------------------------------------------------------------------------------------------
    if (tid < Wd) {
        const float bn = rr_prep_bn(p, 0, tid);
        float l1 = 0.f;
        unsigned nf = 0;
        for (int k = 0; k < p.k_total[0]; ++k) { const float w = p.W[0][(int64_t)tid * p.k_total[0] + k] * bn; l1 += fabsf(w); nf |= (__float_as_uint(w) & 0x7fffffffu) >= 0x7f800000u; }
        const float bound = 2.f * l1 + fabsf(rr_prep_c0(p, 0, tid));
        atomicMax(&red[0], nf ? 0x7f800000u : __float_as_uint(bound));
    }
    for (int st = 1; st < 3; ++st)
        for (int i = tid; i < Wd * p.k_total[st]; i += 1024) {
            const int row = i / p.k_total[st];
            atomicMax(&red[st], __float_as_uint(p.W[st][i] * rr_prep_bn(p, st, row)) & 0x7fffffffu);
        }
    if (tid < Wd) atomicMax(&red[3], __float_as_uint(rr_prep_c0(p, 1, tid)) & 0x7fffffffu);
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
        // smallest exponent field a node row's scale may be made from: s0 * rs * |c0| and rs / (2 se) must stay finite
        int emin = 15;
        emin = max(emin, 15 - Ee);
        if (red[3] != 0) emin = max(emin, E0 + 16 + ((int)(red[3] >> 23) - 126));
        emin = min(emin, 254);
        prep[RRH_MAGIC] = RR_MAGIC; prep[RRH_EE] = (unsigned)Ee; prep[RRH_E0] = (unsigned)E0; prep[RRH_E1] = (unsigned)E1;
        prep[RRH_EMIN] = (unsigned)emin; prep[RRH_BAD] = bad ? 1u : 0u;
        prep[RRH_ACT] = (p.act[0] == 1 ? 1u : 0u) | (p.act[1] == 1 ? 2u : 0u) | (p.act[2] == 1 ? 4u : 0u);
        prep[RRH_PACK] = p.pack16 ? 1u : 0u;
        for (int i = 8; i < RR_HDR; ++i) prep[i] = 0;
    }
    // ---- fragments: one thread per (fragment, lane) --------------------------------------------------------------------------------
    rr_u4 *frag = reinterpret_cast<rr_u4 *>(prep + RR_HDR);
    for (int i = tid; i < SH::F_ALL * 64; i += 1024) {
        const int f = i >> 6, lane = i & 63, l31 = lane & 31, h = lane >> 5;
        int st, blk, c, plane;
        if (f < SH::F_W0H) { st = 0; const int q = f - SH::F_WE; plane = q & 1; c = (q >> 1) % RR_NKE; blk = (q >> 1) / RR_NKE; }
        else if (f < SH::F_W0XL) { st = 1; const int q = f - SH::F_W0H; plane = 0; c = q % NK0; blk = q / NK0; }
        else if (f < SH::F_W1) { st = 1; const int q = f - SH::F_W0XL; plane = 1; c = NKS + q % NKX; blk = q / NKX; }
        else if (f < SH::F_W0SL) { st = 2; const int q = f - SH::F_W1; plane = q & 1; c = (q >> 1) % NKS; blk = (q >> 1) / NKS; }
        else if (p.pack16) { st = 1; const int q = f - SH::F_W0SL; plane = 1; c = (q % (2 * NKS)) >> 1; blk = 2 * (q / (2 * NKS)) + (q & 1); }   // (layer_rp.hip: [pair][c][block of the pair])
        else { st = 1; const int q = f - SH::F_W0SL; plane = 1; c = q / WB; blk = q % WB; }
        const int row = 32 * blk + l31;
        const float sc = rr_pow2((st == 0 ? Ee : (st == 1 ? E0 : E1)) + 127) * rr_prep_bn(p, st, row);
        float wv[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            int k;
            if (st == 0 && p.pack16) {
                // x_i on slots 0 .. 31 (slot 31: the constant 1 of the node pack = the bias), x_j on 32 .. 63, the edge-level columns from 64
                const int j = 16 * c + 8 * h + s;
                if (j < 32) k = j < p.d_x ? j : (j == 31 ? -2 : -1);
                else if (j < 64) k = j - 32 < p.d_x ? p.d_x + (j - 32) : -1;
                else k = 2 * p.d_x + (j - 64) < p.k_total[0] ? 2 * p.d_x + (j - 64) : -1;
            } else if (st == 0) { k = 16 * c + 8 * h + s; k = k < p.k_total[0] ? k : (k == 16 * RR_NKE - 1 ? -2 : -1); }
            else if (st == 1) {
                if (c < NKS) k = p.d_x + rr_kslot_feature(c, h, s);
                else {
                    const int j = 16 * (c - NKS) + 8 * h + s;
                    if (p.pack16) k = j < p.d_x ? j : (j < p.d_x + 2 ? p.d_x + Wd : -1);          // (the in-degree as a high / low pair)
                    else k = j < p.d_x ? j : (j < p.d_x + 4 ? p.d_x + Wd + (j - p.d_x) : -1);
                }
            } else k = rr_kslot_feature(c, h, s);
            wv[s] = k >= 0 ? p.W[st][(int64_t)row * p.k_total[st] + k] * sc : (k == -2 ? rr_prep_c0(p, 0, row) * rr_pow2(Ee + 127) : 0.f);
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
    // ---- biases: edge stage in accumulator units (times se), the node stages plain --------------------------------------------------
    float *tab = reinterpret_cast<float *>(prep + RR_HDR + SH::F_ALL * 256);
    if (tid < Wd) {
        tab[tid] = rr_prep_c0(p, 0, tid) * rr_pow2(Ee + 127);
        tab[Wd + tid] = rr_prep_c0(p, 1, tid);
        tab[2 * Wd + tid] = rr_prep_c0(p, 2, tid);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
static bool rr_enabled() {
    static const int on = [] { const char *d = getenv("GSN_FUSED_RR"); return d ? atoi(d) : 1; }();
    return on != 0;
}

static bool rr_stage_ok(const gsn_chain_stage &g, int width) {
    if (!g.W || g.n_out != width) return false;
    if (g.act != 0 && g.act != 1) return false;
    if ((g.bn_scale != nullptr) != (g.bn_shift != nullptr) || (g.bn_scale != nullptr) != (g.bn_mean != nullptr)) return false;
    return true;
}

// what both register-resident kernels (this file's and layer_rp.hip's) ask of the stages: every stage 128 wide, d_x + 4 <= 32
int rr_shape_ok(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0, const gsn_chain_stage *node1) {
    if (!rr_enabled() || !edge || !node0 || !node1) return 0;
    const int width = 128;
    if (!rr_stage_ok(*edge, width) || !rr_stage_ok(*node0, width) || !rr_stage_ok(*node1, width)) return 0;
    if (d_x < 4 || (d_x & 3) || d_x + 4 > 32) return 0;
    if (node0->n_blocks != 0 || node1->n_blocks != 0) return 0;
    return 1;
}

int rr_supported(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0, const gsn_chain_stage *node1) {
    if (!rr_shape_ok(edge, d_x, node0, node1)) return 0;
    if (edge->n_blocks < 1 || edge->n_blocks > 6 || !edge->blocks) return 0;
    int64_t ke = 0;
    const void *roles[RR_MAXROLE];
    int nroles = 0;
    for (int b = 0; b < edge->n_blocks; ++b) {
        const gsn_block &bl = edge->blocks[b];
        if (!bl.data || !bl.idx32 || bl.idx || bl.width <= 0 || (bl.width & 3)) return 0;
        if (reinterpret_cast<uintptr_t>(bl.data) & 15) return 0;
        ke += bl.width;
        bool seen = false;
        for (int q = 0; q < nroles; ++q) seen = seen || roles[q] == bl.idx32;
        if (!seen) { if (nroles == RR_MAXROLE) return 0; roles[nroles++] = bl.idx32; }
    }
    if (ke > 16 * RR_NKE - 4) return 0;                // (the last k-slot is the bias column)
    return 1;
}

int64_t rr_prepared_bytes(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0, const gsn_chain_stage *node1) {
    if (!rr_supported(edge, d_x, node0, node1) && !rp_supported(edge, d_x, node0, node1)) return 0;
    return (int64_t)RrShape<4, 2>::PREP_WORDS * 4;
}

int rr_prepare(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0, const gsn_chain_stage *node1, void *prepared, hipStream_t st, bool pack16) {
    RrPrepArgs p{};
    const gsn_chain_stage *gs[3] = {edge, node0, node1};
    int ke = 0;
    for (int b = 0; b < edge->n_blocks; ++b) ke += (int)edge->blocks[b].width;
    const int kt[3] = {ke, (int)(d_x + edge->n_out + 4), (int)node0->n_out};
    for (int s = 0; s < 3; ++s) {
        p.W[s] = gs[s]->W; p.bias[s] = gs[s]->bias; p.bn_mean[s] = gs[s]->bn_mean; p.bn_scale[s] = gs[s]->bn_scale; p.bn_shift[s] = gs[s]->bn_shift;
        p.k_total[s] = kt[s]; p.n_out[s] = (int)gs[s]->n_out; p.act[s] = gs[s]->act;
    }
    p.d_x = (int)d_x; p.pack16 = pack16 ? 1 : 0;
    hipLaunchKernelGGL((layer_rr_prepare_kernel<4, 2>), dim3(1), dim3(1024), 0, st, p, reinterpret_cast<unsigned *>(prepared));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "layer_rr_prepare_kernel: %s", hipGetErrorString(e));
    return GSN_OK;
}

int rr_forward(int64_t n_nodes, int64_t n_edges, const int32_t *seg_ptr, const gsn_chain_stage *edge, const float *x, int64_t d_x,
               const gsn_chain_stage *node0, const gsn_chain_stage *node1, const void *prepared, float *out, hipStream_t st) {
    using SH = RrShape<4, 2>;
    RrArgs a{};
    a.n_nodes = (int)n_nodes; a.n_edges = (int)n_edges; a.seg_ptr = seg_ptr;
    a.x = x; a.d_x = (int)d_x; a.out = out; a.prep = reinterpret_cast<const unsigned *>(prepared);
    // roles: the distinct row-index arrays; unused roles alias the first (valid addresses)
    int nroles = 0;
    int brole[6];
    for (int b = 0; b < edge->n_blocks; ++b) {
        int q = 0;
        for (; q < nroles; ++q) if (a.role_idx[q] == edge->blocks[b].idx32) break;
        if (q == nroles) a.role_idx[nroles++] = edge->blocks[b].idx32;
        brole[b] = q;
    }
    for (int q = nroles; q < RR_MAXROLE; ++q) a.role_idx[q] = a.role_idx[0];
    // slots: 16-byte quad q of the concatenated edge row -> (block, quad in block); slot s of a lane half h holds quad 4 c + 2 h + j (s = 2 c + j)
    int qblock[4 * RR_NKE], qoff[4 * RR_NKE], nq = 0;
    for (int b = 0; b < edge->n_blocks; ++b)
        for (int q = 0; q < (int)edge->blocks[b].width / 4; ++q) { qblock[nq] = b; qoff[nq] = q; ++nq; }
    for (int c = 0; c < RR_NKE; ++c)
        for (int j = 0; j < 2; ++j)
            for (int h = 0; h < 2; ++h) {
                int q = 4 * c + 2 * h + j;
                if (q >= nq) q = 0;                    // columns past K_e: any finite data of the same row (their weights are zero)
                const gsn_block &bl = edge->blocks[qblock[q]];
                if ((uint64_t)bl.width * 4 > 0xffffffffull) return 1;
                RrSlotHalf &sh = a.slot[2 * c + j][h];
                sh.base = reinterpret_cast<unsigned long long>(bl.data) + 16ull * qoff[q];
                sh.stride = (unsigned)(bl.width * 4);
                sh.role = (unsigned)brole[qblock[q]];
                if (n_edges == 0) { sh.base = reinterpret_cast<unsigned long long>(x); sh.stride = 0; sh.role = 0; }   // (the gathers are always issued)
            }
    const int64_t n_tiles = (n_nodes + RR_TN - 1) / RR_TN;
    int64_t gx = 256;
    { const char *d = getenv("GSN_FUSED_GRID"); if (d && atoi(d) > 0) gx = atoi(d); }
    int64_t ranges = gx * RR_NW;
    if (ranges > n_tiles) ranges = n_tiles;
    if (gx > ranges) gx = ranges;
    a.n_ranges = (int)ranges;
    static const bool prof_on = [] { const char *d = getenv("GSN_FUSED_PROF"); return d && atoi(d) != 0; }();
    const void *fn = prof_on ? reinterpret_cast<const void *>(&layer_fused_kernel_rr<4, 2, true>) : reinterpret_cast<const void *>(&layer_fused_kernel_rr<4, 2, false>);
    static DeviceOnce attr_set;
    const int attr_dev = current_device();
    if (!attr_set.done(attr_dev)) {
        hipError_t e0 = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e0 != hipSuccess) return set_error(GSN_E_HIP, "hipFuncSetAttribute(layer_fused_kernel_rr): %s", hipGetErrorString(e0));
        attr_set.mark(attr_dev);
    }
    static_assert(SH::LDS_BYTES <= 160 * 1024, "LDS budget");
    if (getenv("GSN_CHAIN_TRACE")) fprintf(stderr, "gsn chain: layer_fused_kernel_rr<4,2> nodes %d edges %d grid %lld ranges %d\n", a.n_nodes, a.n_edges, (long long)gx, a.n_ranges);
    if (prof_on) {
        unsigned long long *prof = nullptr;
        (void)hipMalloc(&prof, 32 * 8); (void)hipMemsetAsync(prof, 0, 32 * 8, st);
        hipLaunchKernelGGL((layer_fused_kernel_rr<4, 2, true>), dim3((unsigned)gx), dim3(64 * RR_NW), SH::LDS_BYTES, st, a, prof);
        unsigned long long h[32];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost);
        (void)hipFree(prof);
        static int shown = 0;
        if (shown++ % 8 == 7)
            for (int w = 0; w < 2; ++w) {
                const unsigned long long *o = h + 16 * w;
                const double nb = o[6] ? (double)o[6] : 1.0, nt = o[7] ? (double)o[7] : 1.0;
                fprintf(stderr, "rrprof range %s: blocks %llu tiles %llu total %llu cycles | per block: wait+convert %.0f issue %.0f edge %.0f | per tile: stage0 %.0f split %.0f stage1 %.0f | waits per block: gathers %.0f, before advance %.0f; per tile: x rows %.0f\n",
                        w ? "mid" : "0", o[6], o[7], o[8], o[0] / nb, o[1] / nb, o[2] / nb, o[3] / nt, o[4] / nt, o[5] / nt, o[9] / nb, o[11] / nb, o[10] / nt);
            }
    } else {
        hipLaunchKernelGGL((layer_fused_kernel_rr<4, 2, false>), dim3((unsigned)gx), dim3(64 * RR_NW), SH::LDS_BYTES, st, a, (unsigned long long *)nullptr);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "layer_fused_kernel_rr: %s", hipGetErrorString(e));
    return GSN_OK;
}

}  // namespace gsn
