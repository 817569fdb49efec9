// bf16x6 variant of chain_seg.hip (same roles, same row-source / gather / segmented-sum machinery): the matrix products run
// on v_mfma_f32_32x32x16_bf16 with both operands split EXACTLY into three bf16 planes by truncation,
//     x = x_h + x_m + x_l   (8 + 8 + 8 mantissa bits),     x w = sum of the 6 plane products of combined order <= 2
// (hh, hm, mh, mm, hl, lh; the dropped ml, lm, ll terms are < 2^-24 |x w|; accumulation stays fp32 inside the MFMA).
// scripts/micro/bf16x6_check.hip: error vs fp64 2.9e-7 of max|C| at K = 160, the same as an fp32 FMA loop (3.1e-7; bf16x3
// would be 2e-5).  Why: 6 bf16 MFMAs of 8 passes do the work of 8 fp32 MFMAs of 16 passes (2.7x less matrix time), and --
// unlike fp32 MFMAs (profiles/r01_coissue.json) -- bf16 MFMAs let the other waves of the SIMD issue, so the split (4 VALU +
// 1.5 permutes per element, done once per input element while staging it into LDS), the epilogue and the segmented sum
// overlap the matrix phase instead of adding to it.
// LDS input tile: three planes [TBM][KP] of bf16, KP = 16 NK16 + 8 (row pitch an odd multiple of 16 bytes: conflict-free
// ds_read_b128 of a lane's 8 consecutive k).  Weights: three planes of 8 bf16 per k-step per lane in registers.
// Covers float4-gatherable shapes (all block widths multiples of 4 floats); anything else runs chain_seg.hip's fp32 kernel.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "chain_common.h"

namespace gsn {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// exact three-way split by truncation: returns the three 16-bit patterns in the HIGH halves of h, m, l
__device__ __forceinline__ void split3(float x, unsigned &h, unsigned &m, unsigned &l) {
    h = __float_as_uint(x);
    const float r1 = x - __uint_as_float(h & 0xffff0000u);
    m = __float_as_uint(r1);
    l = __float_as_uint(r1 - __uint_as_float(m & 0xffff0000u));
}
// (hi16(a), hi16(b)) -> one register holding two bf16: a in the low half (lower k), b in the high half
__device__ __forceinline__ unsigned pack_hi(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

template <int NK16, int TBM, bool PROF>
__global__ __launch_bounds__(TBM * 12) __attribute__((amdgpu_waves_per_eu(3, 3))) void mlp_chain1_seg_bf16_kernel(ChainArgs a, int py, unsigned long long *prof, int prio) {
    constexpr int GT = TBM * 8;                     // threads of group A
    constexpr int GB = TBM * 4;                     // threads of group B (60 weight registers per lane: three waves per SIMD)
    constexpr int RSTEP = GT / 32;
    constexpr int NROW = TBM / RSTEP;               // = 4
    constexpr bool VEC4 = true;
    constexpr int KIN = NK16 * 16;                  // input columns held in LDS
    constexpr int KP = KIN + 8;                     // bf16 row pitch of a plane
    constexpr int PLANE = TBM * KP / 2;             // floats per plane
    constexpr int PF0_J = (KIN + 31) / 32;          // 32-column groups of the input
    constexpr int NSLOT = 4;
    constexpr int RSS = CMAX_BLOCKS * TBM;          // ring slot: row sources per block
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // buffers as offsets into `lds` (a pointer picked from an array of buffer pointers loses its LDS address space)
    const int in_sz = 3 * PLANE, y_sz = TBM * py;
    auto in_tile = [&](int64_t i) { return lds + (int)(i & 1) * in_sz; };
    auto y_tile = [&](int64_t i) { return lds + 2 * in_sz + (int)(i & 1) * y_sz; };
    int *rsrc = reinterpret_cast<int *>(lds + 2 * in_sz + 2 * y_sz);     // [NSLOT][RSS]
    // exact_flag[b] == ordinal of the tile in input buffer b  <=>  that tile has a non-zero bit in its middle or low plane.
    // One-hot / small-integer inputs (layer 0 of the reference models: atom, bond and identifier encodings) are exact in bf16:
    // their m and l planes are all zero and the three plane products that read them contribute exactly nothing.
    int *exact_flag = rsrc + NSLOT * RSS;                                 // [2]

    const int tid = threadIdx.x;
    const bool grp_b = tid >= GT;                   // wave-uniform
    const int t = grp_b ? tid - GT : tid;
    const int64_t n_tiles = (a.m_rows + TBM - 1) / TBM;
    const int64_t first = blockIdx.x;
    const int64_t n_iter = first < n_tiles ? (n_tiles - first + gridDim.x - 1) / gridDim.x : 0;
    const ChainStage &st = a.st[0];

    for (int i = tid; i < 2 * in_sz + 2 * y_sz; i += GT + GB) lds[i] = 0.f;      // padded columns must hold finite values
    if (tid < 2) exact_flag[tid] = -1;
    __syncthreads();

    if (!grp_b) {
        // =============================================================================================================
        // group A: gathers + MFMAs + activated tile -> LDS.  Wave w8: output columns 32 (w8 & 3) .., tile rows 32 (w8 >> 2) ..
        // =============================================================================================================
        const int lane = t & 63, w8 = t >> 6;
        const int w = w8 & 3, rh = w8 >> 2;
        const int li = lane & 31, lh = lane >> 5;
        const int col = 32 * w + li;
        const bool cok = col < st.n_out;
        const bool active = 32 * w < st.n_out;
        // y = scale * (x W^T + bias - mean) + shift = x (scale W)^T + c0: the BN scale is folded into this lane's weight column
        // and the accumulators start at c0, so the epilogue is one max per element.  fp32 MFMAs and VALU instructions do not
        // overlap on a SIMD (measured: VALU work of ANY wave runs ~4x slower while the SIMD's fp32 MFMA sequence is saturated,
        // and the fp32 matrix peak equals the packed-fp32 vector peak), so every VALU instruction per tile is paid in full.
        const float bias = (cok && st.bias) ? st.bias[col] : 0.f;
        float scale = 1.f, c0 = bias;
        if (cok && st.bn_scale) { scale = st.bn_scale[col]; c0 = (bias - st.bn_mean[col]) * scale + st.bn_shift[col]; }
        u32x4 Bh[NK16], Bm[NK16], Bl[NK16];
#pragma unroll
        for (int s16 = 0; s16 < NK16; ++s16) {
            unsigned h[8], m[8], l[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 16 * s16 + 8 * lh + e;
                const float wv = (k < st.k_total && cok) ? st.W[(int64_t)col * st.k_total + k] * scale : 0.f;
                split3(wv, h[e], m[e], l[e]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                Bh[s16][q] = pack_hi(h[2 * q], h[2 * q + 1]);
                Bm[s16][q] = pack_hi(m[2 * q], m[2 * q + 1]);
                Bl[s16][q] = pack_hi(l[2 * q], l[2 * q + 1]);
            }
        }
        // staging map.  Scalar: thread -> column kc0 (+32j) of rows r0 + RSTEP i.  VEC4 (every block width a multiple of 4 floats,
        // 16-byte aligned): thread -> columns 4 qc .. 4 qc + 3 (+32j) of ONE row: a third of the address arithmetic and a quarter
        // of the load instructions per tile (every VALU instruction is paid in full next to fp32 MFMAs).
        const int kc0 = VEC4 ? 4 * (t & 7) : (t & 31), r0 = VEC4 ? (t >> 3) : (t >> 5);
        ColMap cm0[PF0_J];
#pragma unroll
        for (int j = 0; j < PF0_J; ++j) {
            cm0[j] = col_map(a, 0, kc0 + 32 * j);
            cm0[j].rsoff = cm0[j].rsoff / CBM * TBM;
        }
        // two register sets: the gathers of tile t go to set t & 1, two tiles before the tile is computed (one tile of bf16
        // MFMAs is shorter than a gather's latency)
        float pfs[2][PF0_J][NROW];
        // Row sources: thread t resolves tile row t % TBM of input block t / TBM.  Every load below is unconditional from a
        // valid address and its value stays RAW in a register until the end of the tile (any select / sign extension on a
        // just-loaded value makes the compiler wait for it on the spot: a full memory latency at the top of every tile), and
        // the dependent pair perm[row] -> idx[perm[row]] is split over two tiles: permutation entries run three tiles ahead.
        const int rs_r = t & (TBM - 1), rs_b = t / TBM;
        const bool rs_on = rs_b < a.n_blocks;
        const int32_t *rs_ip = nullptr;
#pragma unroll
        for (int q = 0; q < CMAX_BLOCKS; ++q)
            if (q == rs_b) rs_ip = a.bidx32[q];
        const bool rs_idx = rs_on && rs_ip != nullptr;
        if (!rs_idx) rs_ip = a.seg_target;                    // any readable array of m_rows ints
        const bool has_perm = a.row_perm != nullptr;
        const int32_t *permp = has_perm ? a.row_perm : a.seg_target;
        const int m_rows = (int)a.m_rows, last_row = m_rows - 1;
        const int gstep = (int)gridDim.x * TBM;
        auto clampr = [&](int row) { return row < last_row ? row : last_row; };
        // waves whose threads all sit past the last input block skip the row-source work (wave-uniform branch)
        const bool rs_wave = __builtin_amdgcn_readfirstlane(rs_b) < a.n_blocks;
        int rs_lg = 0, raw_idx = 0, raw_perm = 0;
        auto rs_issue = [&](int row0, int lg_raw) {           // row0: first row of the tile whose sources are resolved now
            const int grow = clampr(row0 + rs_r);
            rs_lg = has_perm ? lg_raw : grow;
            raw_idx = rs_ip[rs_lg];
        };
        auto rs_commit = [&](int *dst, int row0) {
            const bool ok = row0 + rs_r < m_rows;
            if (rs_on) dst[rs_b * TBM + rs_r] = ok ? (rs_idx ? raw_idx : rs_lg) : -1;
        };
        auto prefetch_j = [&](float (*pf0)[NROW], const int *rs, int j) {
            if (VEC4) {
                if (kc0 + 32 * j >= KIN) return;
                const int sr = rs[cm0[j].rsoff + r0];
                const float4 v = *reinterpret_cast<const float4 *>(cm0[j].base + (int64_t)(sr < 0 ? 0 : sr) * cm0[j].bw);
                pf0[j][0] = v.x; pf0[j][1] = v.y; pf0[j][2] = v.z; pf0[j][3] = v.w;
                return;
            }
#pragma unroll
            for (int i = 0; i < NROW; ++i) {
                const int sr = rs[cm0[j].rsoff + r0 + RSTEP * i];
                pf0[j][i] = cm0[j].base[(int64_t)(sr < 0 ? 0 : sr) * cm0[j].bw];
            }
        };
        auto stage_in = [&](float (*pf0)[NROW], float *dst, int64_t ordinal) {
            unsigned low_bits = 0;
#pragma unroll
            for (int j = 0; j < PF0_J; ++j) {
                const int k = kc0 + 32 * j;
                if (k < KIN) {
                    unsigned h[4], m[4], l[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { split3(pf0[j][e], h[e], m[e], l[e]); low_bits |= m[e] | l[e]; }
                    float *p = dst + (r0 * KP + k) / 2;
                    u32x2 vh, vm, vl;
                    vh[0] = pack_hi(h[0], h[1]); vh[1] = pack_hi(h[2], h[3]);
                    vm[0] = pack_hi(m[0], m[1]); vm[1] = pack_hi(m[2], m[3]);
                    vl[0] = pack_hi(l[0], l[1]); vl[1] = pack_hi(l[2], l[3]);
                    *reinterpret_cast<u32x2 *>(p) = vh;
                    *reinterpret_cast<u32x2 *>(p + PLANE) = vm;
                    *reinterpret_cast<u32x2 *>(p + 2 * PLANE) = vl;
                }
            }
            if (low_bits & 0xffff0000u) exact_flag[ordinal & 1] = (int)ordinal;      // (every writer stores the same value)
        };
        {   // row sources of this workgroup's first three tiles; the permutation entry of the fourth in flight
            const int row0 = (int)first * TBM;
            rs_issue(row0, permp[clampr(row0 + rs_r)]);
            rs_commit(rsrc, row0);
            rs_issue(row0 + gstep, permp[clampr(row0 + gstep + rs_r)]);
            rs_commit(rsrc + RSS, row0 + gstep);
            rs_issue(row0 + 2 * gstep, permp[clampr(row0 + 2 * gstep + rs_r)]);
            rs_commit(rsrc + 2 * RSS, row0 + 2 * gstep);
            raw_perm = permp[clampr(row0 + 3 * gstep + rs_r)];
        }
        lds_barrier();
        if (n_iter > 0) {
#pragma unroll
            for (int j = 0; j < PF0_J; ++j) prefetch_j(pfs[0], rsrc, j);
            stage_in(pfs[0], in_tile(0), 0);
#pragma unroll
            for (int j = 0; j < PF0_J; ++j) prefetch_j(pfs[1], rsrc + RSS, j);            // tile 1 -> set 1
        }
        lds_barrier();
        unsigned long long pc[6] = {0, 0, 0, 0, 0, 0};
        auto clk = [&]() -> unsigned long long {
            if (!PROF) return 0;
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long v = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
            return v;
        };
        // pf_load: the set that receives the gathers of tile i + 2 (it held tile i, staged a tile ago); pf_stage: tile i + 1
        auto body = [&](int64_t i, float (*pf_load)[NROW], float (*pf_stage)[NROW]) {
            const unsigned long long t0 = clk();
            unsigned long long t1 = t0, t2 = t0, t3 = t0, t4 = t0;
            if (i < n_iter) {
                const int64_t tile = first + i * gridDim.x;
                const int *rs_next = rsrc + (int)((i + 2) & (NSLOT - 1)) * RSS;
                const int row2 = (int)tile * TBM + 3 * gstep;               // the tile three ahead (row sources)
                if (rs_wave) {
                    rs_issue(row2, raw_perm);
                    raw_perm = permp[clampr(row2 + gstep + rs_r)];
                }
                const float *in = in_tile(i);
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = c0;
                t1 = clk();
#pragma unroll
                for (int j = 0; j < PF0_J; ++j) prefetch_j(pf_load, rs_next, j);   // gathers of tile i + 2: a whole tile to land
                if (active && !(a.dbg & 2)) {
                    const float *ap = in + ((32 * rh + li) * KP + 8 * lh) / 2;
#define GSN_MF(x, y) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), acc, 0, 0, 0)
                    if (exact_flag[i & 1] == (int)i) {              // (wave-uniform; decided once per tile, not between MFMAs)
#pragma unroll
                        for (int s16 = 0; s16 < NK16; ++s16) {
                            const u32x4 ah = *reinterpret_cast<const u32x4 *>(ap + 8 * s16);
                            const u32x4 am = *reinterpret_cast<const u32x4 *>(ap + 8 * s16 + PLANE);
                            const u32x4 al = *reinterpret_cast<const u32x4 *>(ap + 8 * s16 + 2 * PLANE);
                            GSN_MF(al, Bh[s16]); GSN_MF(ah, Bl[s16]); GSN_MF(am, Bm[s16]);     // small terms first
                            GSN_MF(ah, Bm[s16]); GSN_MF(am, Bh[s16]); GSN_MF(ah, Bh[s16]);
                        }
                    } else {                                        // bf16-exact tile: x = x_h, three products
#pragma unroll
                        for (int s16 = 0; s16 < NK16; ++s16) {
                            const u32x4 ah = *reinterpret_cast<const u32x4 *>(ap + 8 * s16);
                            GSN_MF(ah, Bl[s16]); GSN_MF(ah, Bm[s16]); GSN_MF(ah, Bh[s16]);
                        }
                    }
#undef GSN_MF
                }
                t2 = clk();
                if (prio & 2) __builtin_amdgcn_s_setprio(2);
                // activated tile -> Y[i&1]  (group B finished reading it one barrier ago)
                float *lp = y_tile(i) + (32 * rh + 4 * lh) * py + col;
                if (cok) {
                    if (st.act == 1) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) lp[((r & 3) + 8 * (r >> 2)) * py] = fmaxf(acc[r], 0.f);
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) lp[((r & 3) + 8 * (r >> 2)) * py] = acc[r];
                    }
                }
                t3 = clk();
                stage_in(pf_stage, in_tile(i + 1), i + 1);                  // (gathered a tile ago; no stores in this group)
                if (rs_wave) rs_commit(rsrc + (int)((i + 3) & (NSLOT - 1)) * RSS, row2);
                t4 = clk();
            }
            lds_barrier();
            if (prio & 2) __builtin_amdgcn_s_setprio(0);
            if (PROF) {
                const unsigned long long t5 = clk();
                pc[0] += t1 - t0; pc[1] += t2 - t1; pc[2] += t3 - t2; pc[3] += t4 - t3; pc[4] += t5 - t4; pc[5] += 1;
            }
        };
        for (int64_t i = 0; i <= n_iter; i += 2) {
            body(i, pfs[0], pfs[1]);
            if (i + 1 <= n_iter) body(i + 1, pfs[1], pfs[0]);
        }
        if (PROF && prof && lane == 0 && blockIdx.x == 0) {
            unsigned long long *o = prof + w8 * 6;
            for (int q = 0; q < 6; ++q) o[q] = pc[q];
        }
        return;
    }

    // =================================================================================================================
    // group B: segmented sum of the activated tile, one tile behind group A.  Thread -> column c, one range of SEG_ROWS
    // target-sorted rows; a segment inside a range is stored, one that straddles a range boundary is added atomically (its
    // output row was zeroed by gsn_segsum_prepare_hip).  Summation order inside a segment = row order.
    // =================================================================================================================
    // Wave wb of the group owns the 16-row range wb of the tile; a lane owns columns c and c + 64.  The row targets are
    // wave-uniform: they come through the scalar cache (constant address space loads of seg_target), ONE TILE AHEAD (the
    // loads are issued after this tile's LDS reads have returned -- SMEM and LDS share lgkmcnt -- and are consumed a tile
    // later), and the walk over the range is scalar control flow.
    static_assert(GB / 64 * SEG_ROWS == TBM, "one 16-row range per wave");
    typedef const __attribute__((address_space(4))) int cint;
    cint *segc = (cint *)a.seg_target;
    const int c = t & 63;
    const bool on = !(a.dbg & 4);
    const bool cok0 = c < st.n_out && on, cok1 = c + 64 < st.n_out && on;
    const int rb = __builtin_amdgcn_readfirstlane(t >> 6) * SEG_ROWS;
    const int m_rows = (int)a.m_rows;
    int tv[SEG_ROWS], prev_t = -2, next_t = -2;                   // targets of the tile processed in the NEXT iteration
    bool tv_full = false;
    const bool all_cols = st.n_out == 128 && on;
    auto load_targets = [&](int64_t tile) {
        const int g0 = (int)tile * TBM + rb;                      // first row of this wave's range (uniform)
        tv_full = g0 + SEG_ROWS <= m_rows;
        if (g0 + SEG_ROWS <= m_rows) {
#pragma unroll
            for (int r = 0; r < SEG_ROWS; ++r) tv[r] = segc[g0 + r];
        } else {
#pragma unroll
            for (int r = 0; r < SEG_ROWS; ++r) tv[r] = g0 + r < m_rows ? segc[g0 + r] : -1;
        }
        prev_t = g0 > 0 && g0 - 1 < m_rows ? segc[g0 - 1] : -2;
        next_t = g0 + SEG_ROWS < m_rows ? segc[g0 + SEG_ROWS] : -2;
    };
    // The target lines stream from HBM exactly once, so a scalar load of them misses every cache (~3000 cycles).  Each wave
    // therefore touches its range of the tile THREE tiles ahead with one vector load whose result is never used (nor waited
    // for: this group only stores): the scalar loads one tile ahead then hit L2.  `warm` is pinned to one register for the
    // whole loop, because the data lands long after the instruction was issued.
    int warm = 0;
    auto warm_targets = [&](int64_t tile) {
        int g = (int)tile * TBM + rb - 8 + (t & 31);
        g = g < 0 ? 0 : (g < m_rows ? g : m_rows - 1);
        const int *p = a.seg_target + g;
        asm volatile("global_load_dword %0, %1, off" : "+v"(warm) : "v"(p) : "memory");
    };
    if (prio & 1) __builtin_amdgcn_s_setprio(3);
    if (n_iter > 0) load_targets(first);
    warm_targets(first + gridDim.x);
    warm_targets(first + 2 * (int64_t)gridDim.x);
    lds_barrier();
    lds_barrier();
    unsigned long long pb[4] = {0, 0, 0, 0};
    auto clkb = [&]() -> unsigned long long {
        if (!PROF) return 0;
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long v = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
        return v;
    };
    for (int64_t i = 0; i <= n_iter; ++i) {
        const unsigned long long u0 = clkb();
        unsigned long long u1 = u0;
        if (i > 0) {
            const float *yp = y_tile(i - 1) + rb * py + c;
            float y0[SEG_ROWS], y1[SEG_ROWS];
#pragma unroll
            for (int r = 0; r < SEG_ROWS; ++r) { y0[r] = yp[r * py]; y1[r] = yp[r * py + 64]; }
            // The LDS values are consumed (an empty asm the compiler must satisfy with its own lgkmcnt wait) BEFORE the next
            // tile's target loads are issued: SMEM shares lgkmcnt with LDS and returns out of order, so a wait placed after
            // the scalar loads would be lgkmcnt(0), i.e. the scalar loads' full latency in front of the walk.
#pragma unroll
            for (int r = 0; r < SEG_ROWS; ++r) asm volatile("" :: "v"(y0[r]), "v"(y1[r]));
            if (PROF) u1 = clkb();
            int cv[SEG_ROWS];
#pragma unroll
            for (int r = 0; r < SEG_ROWS; ++r) cv[r] = tv[r];
            const int cprev = prev_t, cnext = next_t;
            const bool cfull = tv_full;
            if (i < n_iter) load_targets(first + i * gridDim.x);    // next tile's targets: land under the walk below
            warm_targets(first + (i + 2) * gridDim.x);
            // Scalar walk.  Every instruction and every taken branch of it counts (three waves share the SIMD's issue), so the
            // common case -- all 128 output columns, range inside the matrix -- has no per-store predicates and no checks.
            auto walk = [&](auto fast_tag) {
                constexpr bool FAST = decltype(fast_tag)::value;
                int curt = cv[0];
                bool straddle = curt == cprev;
                float s0 = 0.f, s1 = 0.f;
                auto flush = [&](bool atomic) {
                    if (!FAST && curt < 0) return;
                    float *rowp = a.out + (int64_t)curt * st.n_out;                // uniform base; the lane adds its columns
                    if (atomic) {
                        if (FAST || cok0) atomicAdd(rowp + c, s0);
                        if (FAST || cok1) atomicAdd(rowp + c + 64, s1);
                    } else {
                        if (FAST || cok0) rowp[c] = s0;
                        if (FAST || cok1) rowp[c + 64] = s1;
                    }
                };
#pragma unroll
                for (int r = 0; r < SEG_ROWS; ++r) {
                    if (cv[r] != curt) {
                        flush(straddle);
                        curt = cv[r]; s0 = 0.f; s1 = 0.f; straddle = false;
                    }
                    s0 += y0[r]; s1 += y1[r];
                }
                flush(straddle || curt == cnext);
            };
            if (cfull && all_cols) walk(std::true_type{});
            else if (on) walk(std::false_type{});
        }
        const unsigned long long u2 = clkb();
        lds_barrier();
        if (PROF) { const unsigned long long u3 = clkb(); pb[0] += u1 - u0; pb[1] += u2 - u1; pb[2] += u3 - u2; pb[3] += 1; }
    }
    asm volatile("s_waitcnt vmcnt(0)" :: "v"(warm));
    if (PROF && prof && (t & 63) == 0 && blockIdx.x == 0) {
        unsigned long long *o = prof + (8 + (t >> 6)) * 6;
        for (int q = 0; q < 4; ++q) o[q] = pb[q];
    }
}

template <int NK16, int TBM, bool PROF>
static int launch_bf_impl(const ChainArgs &a, hipStream_t st) {
    const void *fn = reinterpret_cast<const void *>(&mlp_chain1_seg_bf16_kernel<NK16, TBM, PROF>);
    static DeviceOnce attr_set;
    const int attr_dev = current_device();
    if (!attr_set.done(attr_dev)) {
        hipError_t e0 = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e0 != hipSuccess) return set_error(GSN_E_HIP, "hipFuncSetAttribute(mlp_chain1_seg_bf16_kernel): %s", hipGetErrorString(e0));
        attr_set.mark(attr_dev);
    }
    int py = a.st[0].n_out | 1;
    if (py == a.st[0].n_out) py += 2;
    const size_t lds = ((size_t)2 * 3 * (TBM * (NK16 * 16 + 8) / 2) + (size_t)2 * TBM * py + (size_t)4 * CMAX_BLOCKS * TBM + 2) * 4;
    if (lds > 160 * 1024) return 1;
    unsigned long long *prof = nullptr;
    int prio = 0;
    { const char *d = getenv("GSN_SEG_PRIO"); if (d) prio = atoi(d); }
    if (PROF) { (void)hipMalloc(&prof, 2 * 8 * 6 * 8); (void)hipMemset(prof, 0, 2 * 8 * 6 * 8); }
    const int64_t n_tiles = (a.m_rows + TBM - 1) / TBM;
    int64_t gx = 256 * (lds <= 78 * 1024 ? 2 : 1);
    if (gx > n_tiles) gx = n_tiles;
    chain_trace("mlp_chain1_seg_bf16_kernel", a);
    hipLaunchKernelGGL((mlp_chain1_seg_bf16_kernel<NK16, TBM, PROF>), dim3((unsigned)gx), dim3(TBM * 12), lds, st, a, py, prof, prio);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "mlp_chain1_seg_bf16_kernel: %s", hipGetErrorString(e));
    if (PROF) {
        unsigned long long h[2 * 8 * 6];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost);
        (void)hipFree(prof);
        static int shown = 0;
        if (shown++ % 8 == 7)
            for (int w = 0; w < TBM / 8; ++w) {
                const unsigned long long *o = h + w * 6, *q = h + (8 + w) * 6;
                if (o[5]) fprintf(stderr, "segprof(bf16x6) A%d tiles %llu: top %llu mfma %llu ywrite %llu stage %llu barrier %llu | B%d: lds %llu walk %llu barrier %llu (cycles per tile)\n", w, o[5],
                                  o[0] / o[5], o[1] / o[5], o[2] / o[5], o[3] / o[5], o[4] / o[5], w, q[0] / (q[3] ? q[3] : 1), q[1] / (q[3] ? q[3] : 1), q[2] / (q[3] ? q[3] : 1));
            }
    }
    return GSN_OK;
}

template <int TBM>
static int launch_bf_k(const ChainArgs &a, hipStream_t st) {
    const int k = a.st[0].k_total;
    { const char *d = getenv("GSN_SEG_PROF"); if (d && atoi(d) && k > 64) return launch_bf_impl<5, TBM, true>(a, st); }
    if (k <= 48) return launch_bf_impl<3, TBM, false>(a, st);
    if (k <= 64) return launch_bf_impl<4, TBM, false>(a, st);
    return launch_bf_impl<5, TBM, false>(a, st);
}

// Returns GSN_OK after launching, or 1 if this shape is not covered (the caller then tries chain_seg.hip's fp32 kernel).
int launch_chain1_seg_bf16(const ChainArgs &a, int maxch, hipStream_t st) {
    if (a.n_stages != 1 || a.stats || !a.seg_target || maxch != 5) return 1;
    if (a.m_rows > (int64_t)2000000000) return 1;                       // 32-bit row arithmetic
    for (int b = 0; b < a.n_blocks; ++b) {
        if (a.bidx[b] && !a.bidx32[b]) return 1;                        // int64 row indices: chain.hip's kernel
        if ((a.bwidth[b] & 3) || (reinterpret_cast<uintptr_t>(a.bdata[b]) & 15)) return 1;   // float4 gathers only
    }
    int tbm = 64;   // measured at 65 536 ZINC graphs: 0.40 ms (TBM 64, one workgroup per CU) vs 0.56 ms (TBM 32, two)
    { const char *d = getenv("GSN_CHAIN_BF16X6"); if (d) tbm = atoi(d); }
    if (tbm == 64) return launch_bf_k<64>(a, st);
    if (tbm == 32) return launch_bf_k<32>(a, st);
    return 1;
}

}  // namespace gsn
