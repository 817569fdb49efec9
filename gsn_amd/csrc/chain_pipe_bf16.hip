// bf16x6 variant of the stage-pipelined two-stage chain (chain_pipe.hip) -- the node chain of a layer,
//   out = act1(bn1( act0(bn0([blocks] W0^T + b0)) W1^T + b1 )) ,  plain [M][n_out] output.
//
// Same roles as chain_pipe.hip (group A: gathers + stage 0, group B: stage 1 + stores, one tile apart, one barrier per tile),
// but both matrix products run on v_mfma_f32_32x32x16_bf16 with operands split EXACTLY into three bf16 planes by truncation
// (x = x_h + x_m + x_l) and the six plane products of combined order <= 2 (see chain_seg_bf16.hip; error vs fp64 equals an
// fp32 FMA loop's, scripts/micro/bf16x6_check.hip).  6 bf16 MFMAs of 8 passes replace 8 fp32 MFMAs of 16 passes, and bf16
// MFMAs -- unlike fp32 ones (profiles/r01_coissue.json) -- let the SIMD's other wave issue meanwhile, so group A's splitting /
// staging overlaps group B's matrix phase and vice versa.
// Tile = 32 rows (three bf16 planes cost 6 bytes per element: 64-row double-buffered tiles would not fit 160 KiB):
//   IN[2][3][32][KP0] + MID[2][3][32][KP1] bf16 = 114 KiB at K0 = 160, K1 = 128 -> one workgroup per CU.
// Wave w of a group owns output columns 32 w ..; its weights live in registers as three planes of 8 bf16 per 16-k step.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "chain_common.h"

namespace gsn {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// exact three-way split by truncation: the three 16-bit patterns are the HIGH halves of h, m, l
__device__ __forceinline__ void split3p(float x, unsigned &h, unsigned &m, unsigned &l) {
    h = __float_as_uint(x);
    const float r1 = x - __uint_as_float(h & 0xffff0000u);
    m = __float_as_uint(r1);
    l = __float_as_uint(r1 - __uint_as_float(m & 0xffff0000u));
}
__device__ __forceinline__ unsigned pack_hi2(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// weights of one lane (output column `col`, k = 16 s + 8 lh + e) -> three planes of packed bf16, BN scale folded in
template <int NK>
__device__ __forceinline__ void load_weight_planes(const ChainStage &st, int col, bool cok, int lh, float scale, u32x4 *Bh, u32x4 *Bm, u32x4 *Bl) {
#pragma unroll
    for (int s = 0; s < NK; ++s) {
        unsigned h[8], m[8], l[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 16 * s + 8 * lh + e;
            const float wv = (k < st.k_total && cok) ? st.W[(int64_t)col * st.k_total + k] * scale : 0.f;
            split3p(wv, h[e], m[e], l[e]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            Bh[s][q] = pack_hi2(h[2 * q], h[2 * q + 1]);
            Bm[s][q] = pack_hi2(m[2 * q], m[2 * q + 1]);
            Bl[s][q] = pack_hi2(l[2 * q], l[2 * q + 1]);
        }
    }
}

#define GSN_MF(x, y) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), acc, 0, 0, 0)

template <int NK0, int NK1, bool VEC4, bool PROF>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void mlp_chain2_pipe_bf16_kernel(ChainArgs a, unsigned long long *prof) {
    auto clk = [&]() -> unsigned long long {
        if (!PROF) return 0;
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long v = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
        return v;
    };
    unsigned long long pc[6] = {0, 0, 0, 0, 0, 0};
    constexpr int TBM = 32, GT = 256;
    constexpr int K0 = NK0 * 16, KP0 = K0 + 8, PL0 = TBM * KP0 / 2;      // plane sizes in floats
    constexpr int K1 = NK1 * 16, KP1 = K1 + 8, PL1 = TBM * KP1 / 2;
    constexpr int PF0_J = (K0 + 31) / 32;
    constexpr int JB = (PF0_J * 3 + 2) / 5;                                // input column groups staged by group B; the rest by group A
    constexpr int RSTEP = GT / 32, NROW = TBM / RSTEP;                    // 8, 4
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // buffers as offsets into `lds` (a pointer picked from an array of buffer pointers loses its LDS address space)
    auto in_tile = [&](int64_t i) { return lds + (int)(i & 1) * 3 * PL0; };
    auto mid_tile = [&](int64_t i) { return lds + 6 * PL0 + (int)(i & 1) * 3 * PL1; };

    const int tid = threadIdx.x;
    const bool grp_b = tid >= GT;                   // wave-uniform
    const int t = tid & (GT - 1);
    const int lane = t & 63, w = t >> 6;            // column block: output columns 32w .. 32w+31
    const int li = lane & 31, lh = lane >> 5;
    const int64_t n_tiles = (a.m_rows + TBM - 1) / TBM;
    const int64_t first = blockIdx.x;
    const int64_t n_iter = first < n_tiles ? (n_tiles - first + gridDim.x - 1) / gridDim.x : 0;

    for (int i = tid; i < 6 * PL0 + 6 * PL1; i += 512) lds[i] = 0.f;      // padded columns must hold zeros
    __syncthreads();

    if (!grp_b) {
        // =============================================================================================================
        // group A: gathers + split + stage 0
        // =============================================================================================================
        const ChainStage &st = a.st[0];
        const int col = 32 * w + li;
        const bool cok = col < st.n_out;
        const bool active = 32 * w < st.n_out;
        const float bias = (cok && st.bias) ? st.bias[col] : 0.f;
        float scale = 1.f, c0 = bias;
        if (cok && st.bn_scale) { scale = st.bn_scale[col]; c0 = (bias - st.bn_mean[col]) * scale + st.bn_shift[col]; }
        u32x4 Bh[NK0], Bm[NK0], Bl[NK0];
        load_weight_planes<NK0>(st, col, cok, lh, scale, Bh, Bm, Bl);
        // this group's share of the input staging: 32-column groups j >= JB (see group B)
        // staging map.  Scalar: thread -> column kc0 (+32j) of rows r0 + 8 i.  VEC4 (every block width a multiple of 4 floats,
        // 16-byte aligned): thread -> columns kc0 .. kc0 + 3 (+32j) of ONE row: a quarter of the loads and address arithmetic,
        // and the planes are written 4 bf16 at a time.
        const int kc0 = VEC4 ? 4 * (t & 7) : (t & 31), r0 = VEC4 ? (t >> 3) : (t >> 5);
        ColMap cm0[PF0_J];
    #pragma unroll
        for (int j = JB; j < PF0_J; ++j) {
            cm0[j] = col_map(a, 0, kc0 + 32 * j);
        }
        float pf0[PF0_J][NROW];
        // Direct rows only (no row indices, no permutation: the launcher sends anything else to chain_pipe.hip): the source row of
        // tile row r is the row itself, so staging needs no row-source table, and this group -- which has stores in flight --
        // never consumes a value loaded a tile earlier (that would put a wait for the previous tile's stores at the top of
        // every tile).
        const int m_rows = (int)a.m_rows, last_row = m_rows - 1;
        const int gstep = (int)gridDim.x * TBM;
        auto clampr = [&](int row) { return row < last_row ? row : last_row; };
        auto prefetch_j = [&](int row0, int j) {
            if (VEC4) {
                if (kc0 + 32 * j >= K0) return;
                const float4 v = *reinterpret_cast<const float4 *>(cm0[j].base + (int64_t)clampr(row0 + r0) * cm0[j].bw);
                pf0[j][0] = v.x; pf0[j][1] = v.y; pf0[j][2] = v.z; pf0[j][3] = v.w;
                return;
            }
    #pragma unroll
            for (int i = 0; i < NROW; ++i) {
                pf0[j][i] = cm0[j].base[(int64_t)clampr(row0 + r0 + RSTEP * i) * cm0[j].bw];
            }
        };
        // prefetched rows -> three bf16 planes.  Padded columns (k >= K) hold a finite clamped-address value and meet zero weights.
        auto stage_in = [&](float *dst) {
            unsigned short *d16 = reinterpret_cast<unsigned short *>(dst);
    #pragma unroll
            for (int j = JB; j < PF0_J; ++j) {
                if (VEC4) {
                    const int k = kc0 + 32 * j;
                    if (k < K0) {
                        unsigned h[4], m[4], l[4];
    #pragma unroll
                        for (int e = 0; e < 4; ++e) split3p(pf0[j][e], h[e], m[e], l[e]);
                        float *p = dst + (r0 * KP0 + k) / 2;
                        u32x2 vh, vm, vl;
                        vh[0] = pack_hi2(h[0], h[1]); vh[1] = pack_hi2(h[2], h[3]);
                        vm[0] = pack_hi2(m[0], m[1]); vm[1] = pack_hi2(m[2], m[3]);
                        vl[0] = pack_hi2(l[0], l[1]); vl[1] = pack_hi2(l[2], l[3]);
                        *reinterpret_cast<u32x2 *>(p) = vh;
                        *reinterpret_cast<u32x2 *>(p + PL0) = vm;
                        *reinterpret_cast<u32x2 *>(p + 2 * PL0) = vl;
                    }
                } else if (kc0 + 32 * j < K0) {
    #pragma unroll
                    for (int i = 0; i < NROW; ++i) {
                        unsigned h, m, l;
                        split3p(pf0[j][i], h, m, l);
                        const int o = (r0 + RSTEP * i) * KP0 + kc0 + 32 * j;
                        d16[o] = (unsigned short)(h >> 16);
                        d16[o + 2 * PL0] = (unsigned short)(m >> 16);
                        d16[o + 4 * PL0] = (unsigned short)(l >> 16);
                    }
                }
            }
        };
        lds_barrier();
        if (n_iter > 0) {
#pragma unroll
            for (int j = JB; j < PF0_J; ++j) prefetch_j((int)first * TBM, j);
            stage_in(in_tile(0));
        }
        lds_barrier();
        for (int64_t i = 0; i <= n_iter; ++i) {
            const unsigned long long t0 = clk();
            unsigned long long t1 = t0, t2 = t0;
            if (i < n_iter) {
#pragma unroll
                for (int j = JB; j < PF0_J; ++j) prefetch_j((int)(first + i * gridDim.x) * TBM + gstep, j);
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = c0;
                if (active) {
                    const float *ap = in_tile(i) + (li * KP0 + 8 * lh) / 2;
                    // fragments of step s + 1 are read before the MFMAs of step s are issued (LDS latency under the MFMAs)
                    u32x4 ah = *reinterpret_cast<const u32x4 *>(ap), am = *reinterpret_cast<const u32x4 *>(ap + PL0),
                          al = *reinterpret_cast<const u32x4 *>(ap + 2 * PL0);
#pragma unroll
                    for (int s = 0; s < NK0; ++s) {
                        u32x4 nh = ah, nm = am, nl = al;
                        if (s + 1 < NK0) {
                            nh = *reinterpret_cast<const u32x4 *>(ap + 8 * (s + 1));
                            nm = *reinterpret_cast<const u32x4 *>(ap + 8 * (s + 1) + PL0);
                            nl = *reinterpret_cast<const u32x4 *>(ap + 8 * (s + 1) + 2 * PL0);
                        }
                        GSN_MF(al, Bh[s]); GSN_MF(ah, Bl[s]); GSN_MF(am, Bm[s]);     // small terms first
                        GSN_MF(ah, Bm[s]); GSN_MF(am, Bh[s]); GSN_MF(ah, Bh[s]);
                        ah = nh; am = nm; al = nl;
                    }
                }
                if (PROF) { asm volatile("" :: "v"(acc[0])); t1 = clk(); }
                // stage output -> MID[i&1] as three bf16 planes (group B finished reading it one barrier ago)
                if (cok) {
                    unsigned short *m16 = reinterpret_cast<unsigned short *>(mid_tile(i)) + (4 * lh) * KP1 + col;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float y = st.act == 1 ? fmaxf(acc[r], 0.f) : acc[r];
                        unsigned h, m, l;
                        split3p(y, h, m, l);
                        const int o = ((r & 3) + 8 * (r >> 2)) * KP1;
                        m16[o] = (unsigned short)(h >> 16);
                        m16[o + 2 * PL1] = (unsigned short)(m >> 16);
                        m16[o + 4 * PL1] = (unsigned short)(l >> 16);
                    }
                }
                stage_in(in_tile(i + 1));                                   // (waits for this group's gathers; it has no stores)
                t2 = clk();
            }
            lds_barrier();
            if (PROF) { const unsigned long long t3 = clk(); pc[0] += t1 - t0; pc[1] += t2 - t1; pc[2] += t3 - t2; pc[5] += 1; }
        }
        if (PROF && prof && lane == 0 && blockIdx.x == 0) {
            unsigned long long *o = prof + w * 6;
            for (int q = 0; q < 6; ++q) o[q] = pc[q];
        }
        return;
    }

    // =================================================================================================================
    // group B: stage 1 + output, one tile behind group A
    // =================================================================================================================
    const ChainStage &st = a.st[1];
    const int col = 32 * w + li;
    const bool cok = col < st.n_out;
    const bool active = 32 * w < st.n_out;
    const float bias = (cok && st.bias) ? st.bias[col] : 0.f;
    float scale = 1.f, c0 = bias;
    if (cok && st.bn_scale) { scale = st.bn_scale[col]; c0 = (bias - st.bn_mean[col]) * scale + st.bn_shift[col]; }
    u32x4 Bh[NK1], Bm[NK1], Bl[NK1];
    load_weight_planes<NK1>(st, col, cok, lh, scale, Bh, Bm, Bl);
    // This group stages the 32-column groups j < JB of the input tiles (gathers, split into planes), group A the rest: with
    // the split at 3 : 2 the two groups' instruction streams per tile are about even (in-kernel profile, GSN_PIPE_PROF=1).
    // staging map.  Scalar: thread -> column kc0 (+32j) of rows r0 + 8 i.  VEC4 (every block width a multiple of 4 floats,
    // 16-byte aligned): thread -> columns kc0 .. kc0 + 3 (+32j) of ONE row: a quarter of the loads and address arithmetic,
    // and the planes are written 4 bf16 at a time.
    const int kc0 = VEC4 ? 4 * (t & 7) : (t & 31), r0 = VEC4 ? (t >> 3) : (t >> 5);
    ColMap cm0[PF0_J];
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        cm0[j] = col_map(a, 0, kc0 + 32 * j);
    }
    float pf0[PF0_J][NROW];
    // Direct rows only (no row indices, no permutation: the launcher sends anything else to chain_pipe.hip): the source row of
    // tile row r is the row itself, so staging needs no row-source table, and this group -- which has stores in flight --
    // never consumes a value loaded a tile earlier (that would put a wait for the previous tile's stores at the top of
    // every tile).
    const int m_rows = (int)a.m_rows, last_row = m_rows - 1;
    const int gstep = (int)gridDim.x * TBM;
    auto clampr = [&](int row) { return row < last_row ? row : last_row; };
    auto prefetch_j = [&](int row0, int j) {
        if (VEC4) {
            if (kc0 + 32 * j >= K0) return;
            const float4 v = *reinterpret_cast<const float4 *>(cm0[j].base + (int64_t)clampr(row0 + r0) * cm0[j].bw);
            pf0[j][0] = v.x; pf0[j][1] = v.y; pf0[j][2] = v.z; pf0[j][3] = v.w;
            return;
        }
#pragma unroll
        for (int i = 0; i < NROW; ++i) {
            pf0[j][i] = cm0[j].base[(int64_t)clampr(row0 + r0 + RSTEP * i) * cm0[j].bw];
        }
    };
    // prefetched rows -> three bf16 planes.  Padded columns (k >= K) hold a finite clamped-address value and meet zero weights.
    auto stage_in = [&](float *dst) {
        unsigned short *d16 = reinterpret_cast<unsigned short *>(dst);
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            if (VEC4) {
                const int k = kc0 + 32 * j;
                if (k < K0) {
                    unsigned h[4], m[4], l[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) split3p(pf0[j][e], h[e], m[e], l[e]);
                    float *p = dst + (r0 * KP0 + k) / 2;
                    u32x2 vh, vm, vl;
                    vh[0] = pack_hi2(h[0], h[1]); vh[1] = pack_hi2(h[2], h[3]);
                    vm[0] = pack_hi2(m[0], m[1]); vm[1] = pack_hi2(m[2], m[3]);
                    vl[0] = pack_hi2(l[0], l[1]); vl[1] = pack_hi2(l[2], l[3]);
                    *reinterpret_cast<u32x2 *>(p) = vh;
                    *reinterpret_cast<u32x2 *>(p + PL0) = vm;
                    *reinterpret_cast<u32x2 *>(p + 2 * PL0) = vl;
                }
            } else if (kc0 + 32 * j < K0) {
#pragma unroll
                for (int i = 0; i < NROW; ++i) {
                    unsigned h, m, l;
                    split3p(pf0[j][i], h, m, l);
                    const int o = (r0 + RSTEP * i) * KP0 + kc0 + 32 * j;
                    d16[o] = (unsigned short)(h >> 16);
                    d16[o + 2 * PL0] = (unsigned short)(m >> 16);
                    d16[o + 4 * PL0] = (unsigned short)(l >> 16);
                }
            }
        }
    };
    // Schedule of this group per tile:  staged rows of tile i+1 -> LDS (their loads were issued before the previous tile's
    // stores, so the wait counts past those stores instead of draining them)  |  stage-1 MFMAs of tile i-1  |  loads of
    // tile i+2  |  stores of tile i-1.  The staging runs while group A is in its matrix phase and this group's matrix
    // phase follows A's, so the matrix pipe stays busy from the top of the tile until this group's stores.
    lds_barrier();
    if (n_iter > 0) {
#pragma unroll
        for (int j = 0; j < JB; ++j) prefetch_j((int)first * TBM, j);
        stage_in(in_tile(0));
#pragma unroll
        for (int j = 0; j < JB; ++j) prefetch_j((int)first * TBM + gstep, j);
    }
    lds_barrier();
    for (int64_t i = 0; i <= n_iter; ++i) {
        const unsigned long long t0 = clk();
        unsigned long long t1 = t0, t2 = t0, t3 = t0, t4 = t0;
        const int row2 = (int)(first + i * gridDim.x) * TBM + 2 * gstep;       // first row of the tile after next
        if (i < n_iter) stage_in(in_tile(i + 1));
        t1 = clk();
        if (i > 0) {
            const int64_t row0 = (first + (i - 1) * gridDim.x) * TBM;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = c0;
            if (active) {
                const float *ap = mid_tile(i - 1) + (li * KP1 + 8 * lh) / 2;
                u32x4 ah = *reinterpret_cast<const u32x4 *>(ap), am = *reinterpret_cast<const u32x4 *>(ap + PL1),
                      al = *reinterpret_cast<const u32x4 *>(ap + 2 * PL1);
#pragma unroll
                for (int s = 0; s < NK1; ++s) {
                    u32x4 nh = ah, nm = am, nl = al;
                    if (s + 1 < NK1) {
                        nh = *reinterpret_cast<const u32x4 *>(ap + 8 * (s + 1));
                        nm = *reinterpret_cast<const u32x4 *>(ap + 8 * (s + 1) + PL1);
                        nl = *reinterpret_cast<const u32x4 *>(ap + 8 * (s + 1) + 2 * PL1);
                    }
                    GSN_MF(al, Bh[s]); GSN_MF(ah, Bl[s]); GSN_MF(am, Bm[s]);
                    GSN_MF(ah, Bm[s]); GSN_MF(am, Bh[s]); GSN_MF(ah, Bh[s]);
                    ah = nh; am = nm; al = nl;
                }
            }
            if (PROF) { asm volatile("" :: "v"(acc[0])); t2 = clk(); }
            if (i < n_iter) {
#pragma unroll
                for (int j = 0; j < JB; ++j) prefetch_j(row2, j);       // issued BEFORE the stores below
            }
            t3 = clk();
            float *tile_out = a.out + row0 * st.n_out;                  // wave-uniform base
            const int lane_off = (4 * lh) * st.n_out + col;            // 32-bit per-lane offset inside the tile
            const bool full = row0 + TBM <= a.m_rows;
            auto emit = [&](auto actf) {
                if (full) {
                    if (cok) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) tile_out[lane_off + ((r & 3) + 8 * (r >> 2)) * st.n_out] = actf(acc[r]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dr = (r & 3) + 8 * (r >> 2);
                        if (cok && row0 + 4 * lh + dr < a.m_rows) tile_out[lane_off + dr * st.n_out] = actf(acc[r]);
                    }
                }
            };
            if (st.act == 1) emit([](float y) { return y > 0.f ? y : 0.f; });
            else emit([](float y) { return y; });
        } else if (i < n_iter) {
#pragma unroll
            for (int j = 0; j < JB; ++j) prefetch_j(row2, j);
        }
        t4 = clk();
        lds_barrier();
        if (PROF) { const unsigned long long t5 = clk(); pc[0] += t1 - t0; pc[1] += t2 - t1; pc[2] += t3 - t2; pc[3] += t4 - t3; pc[4] += t5 - t4; pc[5] += 1; }
    }
    if (PROF && prof && lane == 0 && blockIdx.x == 0) {
        unsigned long long *o = prof + (4 + w) * 6;
        for (int q = 0; q < 6; ++q) o[q] = pc[q];
    }
}

#undef GSN_MF

template <int NK0, int NK1, bool VEC4, bool PROF = false>
static int launch_pipe_bf_impl(const ChainArgs &a, hipStream_t st) {
    constexpr size_t lds = ((size_t)6 * (32 * (NK0 * 16 + 8) / 2) + (size_t)6 * (32 * (NK1 * 16 + 8) / 2)) * 4;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static DeviceOnce attr_set;
    const int attr_dev = current_device();
    if (!attr_set.done(attr_dev)) {
        hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp_chain2_pipe_bf16_kernel<NK0, NK1, VEC4, PROF>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e0 != hipSuccess) return set_error(GSN_E_HIP, "hipFuncSetAttribute(mlp_chain2_pipe_bf16_kernel): %s", hipGetErrorString(e0));
        attr_set.mark(attr_dev);
    }
    unsigned long long *prof = nullptr;
    if (PROF) { (void)hipMalloc(&prof, 8 * 6 * 8); (void)hipMemset(prof, 0, 8 * 6 * 8); }
    const int64_t n_tiles = (a.m_rows + 31) / 32;
    int64_t gx = 256;
    if (gx > n_tiles) gx = n_tiles;
    chain_trace("mlp_chain2_pipe_bf16_kernel", a);
    hipLaunchKernelGGL((mlp_chain2_pipe_bf16_kernel<NK0, NK1, VEC4, PROF>), dim3((unsigned)gx), dim3(512), lds, st, a, prof);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "mlp_chain2_pipe_bf16_kernel: %s", hipGetErrorString(e));
    if (PROF) {
        unsigned long long h[8 * 6];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost);
        (void)hipFree(prof);
        static int shown = 0;
        if (shown++ % 8 == 7)
            for (int w = 0; w < 4; ++w) {
                const unsigned long long *o = h + w * 6, *q = h + (4 + w) * 6;
                if (o[5] && q[5])
                    fprintf(stderr, "pipeprof(bf16x6) A%d tiles %llu: mfma %llu epilogue %llu barrier %llu | B%d: stage %llu mfma %llu issue %llu stores %llu barrier %llu (cycles per tile)\n",
                            w, o[5], o[0] / o[5], o[1] / o[5], o[2] / o[5], w, q[0] / q[5], q[1] / q[5], q[2] / q[5], q[3] / q[5], q[4] / q[5]);
            }
    }
    return GSN_OK;
}

// Returns GSN_OK after launching, or 1 if this shape is not covered (the caller then uses chain_pipe.hip / chain.hip).
int launch_chain2_pipe_bf16(const ChainArgs &a, int maxch, hipStream_t st) {
    if (a.n_stages != 2 || a.stats || a.seg_target || a.row_perm) return 1;
    for (int b = 0; b < a.n_blocks; ++b)
        if (a.bidx[b] || a.bidx32[b]) return 1;                         // direct rows only
    { const char *d = getenv("GSN_CHAIN_BF16X6"); if (d && atoi(d) == 0) return 1; }
    if (a.m_rows > (int64_t)2000000000) return 1;                       // 32-bit row arithmetic
    const int k1 = a.st[1].k_total;
    bool vec4 = true;                                                   // float4 gathers: stage-0 widths and bases 16-byte aligned
    for (int b = a.st[0].first_block; b < a.st[0].first_block + a.st[0].n_blocks; ++b)
        if ((a.bwidth[b] & 3) || (reinterpret_cast<uintptr_t>(a.bdata[b]) & 15)) vec4 = false;
    { const char *d = getenv("GSN_PIPE_VEC4"); if (d && atoi(d) == 0) vec4 = false; }
    { const char *d = getenv("GSN_PIPE_PROF"); if (d && atoi(d) && vec4 && maxch != 5 && k1 > 64) return launch_pipe_bf_impl<10, 8, true, true>(a, st); }
    if (vec4) {
        if (maxch == 5) return k1 <= 64 ? launch_pipe_bf_impl<5, 4, true>(a, st) : launch_pipe_bf_impl<5, 8, true>(a, st);
        return k1 <= 64 ? launch_pipe_bf_impl<10, 4, true>(a, st) : launch_pipe_bf_impl<10, 8, true>(a, st);
    }
    if (maxch == 5) return k1 <= 64 ? launch_pipe_bf_impl<5, 4, false>(a, st) : launch_pipe_bf_impl<5, 8, false>(a, st);
    return k1 <= 64 ? launch_pipe_bf_impl<10, 4, false>(a, st) : launch_pipe_bf_impl<10, 8, false>(a, st);
}

}  // namespace gsn
