// HP-2 dense stage for gfx950, general shapes: one fused models_misc.mlp layer on the fp32 matrix cores.
//
//   Y = act( (X W^T + bias - mean) * scale + shift )        X rows assembled on the fly from up to 5 blocks
//
// replaces  torch.cat((x_i, x_j, identifiers.., edge_features), -1) -> nn.Linear -> BatchNorm1d -> activation
// (models_misc.py:52-58 driven by GSN_sparse.py:166-171 / GSN_edge_sparse.py:160-165): neither the gathered
// x[edge_index_i] / x[edge_index_j] copies nor the [E, msg_in] concatenation ever exist in HBM.
// This is the any-shape kernel (any K, any n_out via column tiles, elu / tanh epilogues, statistics pass); the common
// small shapes (K <= 160, n_out <= 128, identity / relu) run on the fused weights-in-registers kernel of chain.hip.
//
// Two kernels: linear_fwd_bf16_kernel (default; every fp32 product as six exact bf16 plane products, see further down) and
// linear_fwd_kernel: fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32 (exact f32, 64 FLOP/clk/SIMD = the chip's 157 TF fp32
// peak; gfx950 has no TF32, and a single bf16 product misses the 1e-5 parity tolerance).  Tiling of the fp32 kernel:
//   workgroup = 4 waves = 128 x 128 output tile; each wave owns 64 x 64 = 2 x 2 MFMA tiles (64 accumulator registers);
//   K is walked in 32-wide slices of the CONCATENATED input row; A slices [128][32] and W^T slices [32][128] go through
//   registers into double-buffered LDS tiles with +1 padding (pitch 33 / 129 words: staging writes and the per-lane
//   fragment reads A[i=lane&31][k=lane>>5], B[k=lane>>5][j=lane&31] are all bank-conflict free).  Software pipeline per
//   slice:  write slice c (registers -> LDS) | LDS-only barrier | issue the global loads of slice c+1 | 64 MFMAs on
//   slice c.  With WRES the whole W^T tile is instead loaded into LDS once per (persistent) workgroup.
//   Lessons baked in (profiles/r01_*): loaded values are consumed RAW one step later (a select on a just-loaded value
//   makes the compiler wait for it on the spot and serialises the prefetch), barriers wait on lgkmcnt only
//   (__syncthreads would drain the prefetch), epilogue constants live in registers, and the train-mode BatchNorm
//   statistics pass keeps per-column fp64 partial sums in registers (one atomic per column per workgroup).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "gsn_internal.h"

namespace gsn {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int APITCH = BK + 1, WPITCH = BN + 1;
constexpr int MAX_BLOCKS = 5;
constexpr int NPRE = BM / 8;  // staged elements per thread per slice (A); a W^T slice needs BN / 8 = 16 as well

struct LinArgs {
    int64_t m_rows;
    int n_blocks;
    // the same five blocks as named fields (cb*, cw*): the per-lane block choice of col_map_l among ARRAY elements of the argument
    // block is compiled into a vector load of the pointer from the argument segment -- a dependent load and a wait in front of every
    // slice's fetch; selects among named fields stay in registers
    const float *cb0, *cb1, *cb2, *cb3, *cb4;
    int cw0, cw1, cw2, cw3, cw4;
    const float *bdata[MAX_BLOCKS];
    const int64_t *bidx[MAX_BLOCKS];
    const int32_t *bidx32[MAX_BLOCKS];
    int bwidth[MAX_BLOCKS];
    const float *W, *bias, *bn_mean, *bn_scale, *bn_shift;
    int64_t w_rs, w_cs;        // element strides of W[j][k] (k_total, 1: row-major; 1, ld: a transposed view -- the scalar staging path of the bf16x6 kernel only)
    int k_total, n_out, act;
    const int32_t *row_perm;
    float *out;
    double *stats;
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void lds_barrier_l() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float apply_act(float y, int act) {
    switch (act) {
        case 1: return y > 0.f ? y : 0.f;
        case 2: return y > 0.f ? y : expm1f(y);
        case 3: return tanhf(y);
        default: return y;
    }
}

// which block / column does concatenated column kg belong to (clamped to a valid address when kg >= K: its W row is 0)
struct ColMapL {
    const float *base;
    int bw, rsoff;
};

__device__ __forceinline__ ColMapL col_map_l(const LinArgs &a, int kg) {
    const int p1 = a.cw0, p2 = p1 + a.cw1, p3 = p2 + a.cw2, p4 = p3 + a.cw3;        // first column of blocks 1 .. 4 (uniform; past K when absent)
    if (kg >= a.k_total) kg = 0;                                                     // (a valid address; its W row is 0)
    ColMapL m;
    m.base = a.cb0 + kg; m.bw = a.cw0; m.rsoff = 0;
    if (kg >= p1) { m.base = a.cb1 + (kg - p1); m.bw = a.cw1; m.rsoff = BM; }
    if (kg >= p2) { m.base = a.cb2 + (kg - p2); m.bw = a.cw2; m.rsoff = 2 * BM; }
    if (kg >= p3) { m.base = a.cb3 + (kg - p3); m.bw = a.cw3; m.rsoff = 3 * BM; }
    if (kg >= p4) { m.base = a.cb4 + (kg - p4); m.bw = a.cw4; m.rsoff = 4 * BM; }
    return m;
}

struct RowSrcL {
    int v[3];
};

template <bool STATS, bool WRES>
__global__ __launch_bounds__(256, 2) void linear_fwd_kernel(LinArgs a, int k_pad) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // WRES: Wt [k_pad][WPITCH] resident;  else: Wt [2][BK][WPITCH] double buffer
    float *Wt = lds;
    float *As = Wt + (WRES ? k_pad : 2 * BK) * WPITCH;          // [2][BM][APITCH]
    int *rsrc = reinterpret_cast<int *>(As + 2 * BM * APITCH);  // [2][MAX_BLOCKS][BM]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int kc = tid & 31, r0 = tid >> 5;
    const int n0 = blockIdx.y * BN;
    const int64_t n_tiles = (a.m_rows + BM - 1) / BM;
    const int n_slices = k_pad / BK;

    if (WRES) {
        for (int i = tid; i < k_pad * BN; i += 256) {
            const int j = i / k_pad, k = i - j * k_pad;   // consecutive threads walk k: coalesced rows of W
            float v = 0.f;
            if (k < a.k_total && n0 + j < a.n_out) v = a.W[(int64_t)(n0 + j) * a.k_total + k];
            Wt[k * WPITCH + j] = v;
        }
    }

    // row sources: thread t resolves tile row t&127 of blocks t>>7, (t>>7)+2, (t>>7)+4; its index pointers are picked
    // once and live in VGPRs (looping over the block table per row keeps the whole table in SGPRs and spills it)
    const int rs_r = tid & (BM - 1);
    const int32_t *rp32[3];
    const int64_t *rp64[3];
    bool ron[3];
#pragma unroll
    for (int h = 0; h < 3; ++h) {
        const int b = (tid >> 7) + 2 * h;
        ron[h] = b < a.n_blocks;
        rp32[h] = nullptr; rp64[h] = nullptr;
#pragma unroll
        for (int q = 0; q < MAX_BLOCKS; ++q)
            if (q == b) { rp32[h] = a.bidx32[q]; rp64[h] = a.bidx[q]; }
    }
    auto rs_fetch = [&](int64_t row0, RowSrcL &rs) {
        const int64_t grow = row0 + rs_r;
        const bool ok = grow < a.m_rows;
        int64_t logical = 0;
        if (ok) logical = a.row_perm ? (int64_t)a.row_perm[grow] : grow;
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            int r = 0;   // rows past the end read row 0 (never emitted)
            if (ron[h] && ok) r = rp32[h] ? rp32[h][logical] : (rp64[h] ? (int)rp64[h][logical] : (int)logical);
            rs.v[h] = r;
        }
    };
    auto rs_store = [&](int *dst, const RowSrcL &rs) {
#pragma unroll
        for (int h = 0; h < 3; ++h)
            if (ron[h]) dst[((tid >> 7) + 2 * h) * BM + rs_r] = rs.v[h];
    };

    // epilogue constants of this lane's two output columns:  y = acc * scale + c0
    float e_bias[2], e_scale[2], e_c0[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + li;
        const bool cok = col < a.n_out;
        e_bias[j] = (cok && a.bias) ? a.bias[col] : 0.f;
        e_scale[j] = 1.f; e_c0[j] = e_bias[j];
        if (cok && a.bn_scale) { e_scale[j] = a.bn_scale[col]; e_c0[j] = (e_bias[j] - a.bn_mean[col]) * e_scale[j] + a.bn_shift[col]; }
    }

    double st_sum[2] = {0.0, 0.0}, st_sq[2] = {0.0, 0.0};
    float preA[NPRE], preW[NPRE];

    // global loads of slice c: A columns c*32 + kc for 16 rows; W^T rows (n0 + r0 + 8i) at k = c*32 + kc
    auto fetch = [&](const int *rs, int c) {
        const ColMapL cm = col_map_l(a, c * BK + kc);
        const int *rp = rs + cm.rsoff + r0;
#pragma unroll
        for (int i = 0; i < NPRE; ++i) preA[i] = cm.base[(int64_t)rp[8 * i] * cm.bw];
        if (!WRES) {
            const int kg = c * BK + kc;
            const int kk = kg < a.k_total ? kg : 0;
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                int j = n0 + r0 + 8 * i;
                j = j < a.n_out ? j : 0;
                preW[i] = a.W[(int64_t)j * a.k_total + kk];
            }
        }
    };

    int64_t tile = blockIdx.x;
    {
        RowSrcL r;
        rs_fetch(tile * BM, r);
        rs_store(rsrc, r);
    }
    __syncthreads();
    if (tile < n_tiles) fetch(rsrc, 0);
    int cur_rs = 0, cur_as = 0;
    RowSrcL rs_next;

    for (; tile < n_tiles; tile += gridDim.x) {
        const int64_t row0 = tile * BM;
        const int64_t next_tile = tile + gridDim.x;
        const bool has_next = next_tile < n_tiles;
        rs_fetch(next_tile * BM, rs_next);   // lands while this tile computes

        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        for (int c = 0; c < n_slices; ++c) {
            float *Ab = As + cur_as * (BM * APITCH);
            float *Wb = WRES ? Wt + c * BK * WPITCH : Wt + cur_as * (BK * WPITCH);
#pragma unroll
            for (int i = 0; i < NPRE; ++i) Ab[(r0 + 8 * i) * APITCH + kc] = preA[i];
            if (!WRES) {
                // rows of W^T past K / columns past n_out are zeroed by a per-thread 0/1 factor applied to values that
                // were loaded one slice earlier (no select on fresh loads)
                const float km = (c * BK + kc < a.k_total) ? 1.f : 0.f;
#pragma unroll
                for (int i = 0; i < NPRE; ++i) {
                    const float jm = (n0 + r0 + 8 * i < a.n_out) ? km : 0.f;
                    Wb[kc * WPITCH + r0 + 8 * i] = preW[i] * jm;
                }
            }
            const bool last = c == n_slices - 1;
            if (last && has_next) rs_store(rsrc + (cur_rs ^ 1) * (MAX_BLOCKS * BM), rs_next);
            lds_barrier_l();
            if (!last) fetch(rsrc + cur_rs * (MAX_BLOCKS * BM), c + 1);
            else if (has_next) fetch(rsrc + (cur_rs ^ 1) * (MAX_BLOCKS * BM), 0);

            int ksteps = (a.k_total - c * BK + 1) >> 1;
            ksteps = ksteps > BK / 2 ? BK / 2 : ksteps;
            const float *ap0 = Ab + (wm * 64 + li) * APITCH + lh;
            const float *ap1 = ap0 + 32 * APITCH;
            const float *bp0 = Wb + lh * WPITCH + wn * 64 + li;
            if (ksteps == BK / 2) {
#pragma unroll
                for (int ks = 0; ks < BK / 2; ++ks) {
                    const float a0 = ap0[2 * ks], a1 = ap1[2 * ks];
                    const float b0 = bp0[2 * ks * WPITCH], b1 = bp0[2 * ks * WPITCH + 32];
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                }
            } else {
                for (int ks = 0; ks < ksteps; ++ks) {
                    const float a0 = ap0[2 * ks], a1 = ap1[2 * ks];
                    const float b0 = bp0[2 * ks * WPITCH], b1 = bp0[2 * ks * WPITCH + 32];
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                }
            }
            cur_as ^= 1;
        }
        cur_rs ^= 1;

        // epilogue.  C layout of a 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
        const bool full = (row0 + BM <= a.m_rows) && (n0 + BN <= a.n_out);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + li;
            const bool cok = col < a.n_out;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int64_t rbase = row0 + wm * 64 + i * 32 + 4 * lh;
                float *op = a.out + rbase * a.n_out + col;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    if (!full && (!cok || rbase + dr >= a.m_rows)) continue;
                    if (STATS) {
                        const float h = acc[i][j][r] + e_bias[j];
                        st_sum[j] += (double)h;
                        st_sq[j] += (double)h * (double)h;
                        if (a.out) op[(int64_t)dr * a.n_out] = h;      // statistics AND the raw pre-BN rows in one pass
                    } else {
                        op[(int64_t)dr * a.n_out] = apply_act(fmaf(acc[i][j][r], e_scale[j], e_c0[j]), a.act);
                    }
                }
            }
        }
    }

    if (STATS) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + li;
            double s = st_sum[j], q = st_sq[j];
            s += __shfl_xor(s, 32);
            q += __shfl_xor(q, 32);
            if (lh == 0 && col < a.n_out) {
                atomicAdd(&a.stats[col], s);
                atomicAdd(&a.stats[a.n_out + col], q);
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// bf16x6 twin of linear_fwd_kernel: the same tiling idea on v_mfma_f32_32x32x16_bf16 with both operands split exactly
// into three bf16 planes by truncation while they are staged into LDS, six plane products per fp32 product (see
// chain_seg_bf16.hip / scripts/micro/bf16x6_check.hip: fp32-equivalent error).  Why: fp32 MFMAs block every other wave of
// their SIMD (profiles/r01_coissue.json) and run at 1/16 of the bf16 rate.
//   workgroup = 8 waves (two per SIMD, so that one wave's staging / splitting runs next to the other's MFMAs) = 128 x 128
//   output tile; wave w owns rows 64 (w >> 2) .. + 64 and columns 32 (w & 3) .. + 32 (two accumulator tiles);
//   per 32-wide K slice: A planes [3][128][40] and W planes [3][128 output columns][40] of bf16 (row pitch 80 bytes = an odd
//   multiple of 16: conflict-free ds_read_b128 of a lane's 8 consecutive k), double-buffered: 120 KiB, one workgroup per CU.
// ---------------------------------------------------------------------------------------------------------------------
typedef unsigned u32x4l __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8l __attribute__((ext_vector_type(8)));
constexpr int BKP = BK + 8;                       // bf16 row pitch of a plane
constexpr int BPLANE = BM * BKP / 2;              // floats per plane (BM == BN)
constexpr int NPRE8 = BM / 16;                    // staged elements per thread per slice with 512 threads

__device__ __forceinline__ void split3l(float x, unsigned &h, unsigned &m, unsigned &l) {
    h = __float_as_uint(x);
    const float r1 = x - __uint_as_float(h & 0xffff0000u);
    m = __float_as_uint(r1);
    l = __float_as_uint(r1 - __uint_as_float(m & 0xffff0000u));
}

template <bool STATS, bool VEC4, int D>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void linear_fwd_bf16_kernel(LinArgs a, int k_pad) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // [2 buffers][A planes 3 | W planes 3][BPLANE floats], then the row-source tables [2][MAX_BLOCKS][BM]
    auto a_planes = [&](int buf) { return lds + buf * 6 * BPLANE; };
    auto w_planes = [&](int buf) { return lds + buf * 6 * BPLANE + 3 * BPLANE; };
    int *rsrc = reinterpret_cast<int *>(lds + 12 * BPLANE);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, lh = lane >> 5;
    // staging.  Scalar: column kc, rows / output columns r0 + 16 i (8 each).  VEC4 (block widths, K and all bases multiples
    // of 4 floats): columns kc .. kc + 3 of rows / output columns r0, r0 + 64 -- a quarter of the loads and LDS writes.
    const int kc = VEC4 ? 4 * (tid & 7) : (tid & 31), r0 = VEC4 ? (tid >> 3) : (tid >> 5);
    const int n0 = blockIdx.y * BN;
    const int64_t n_tiles = (a.m_rows + BM - 1) / BM;
    const int n_slices = k_pad / BK;

    for (int i = tid; i < 12 * BPLANE; i += 512) lds[i] = 0.f;

    const int rs_r = tid & (BM - 1);
    const int32_t *rp32[2];
    const int64_t *rp64[2];
    bool ron[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int b = (tid >> 7) + 4 * h;
        ron[h] = b < a.n_blocks;
        rp32[h] = nullptr; rp64[h] = nullptr;
#pragma unroll
        for (int q = 0; q < MAX_BLOCKS; ++q)
            if (q == b) { rp32[h] = a.bidx32[q]; rp64[h] = a.bidx[q]; }
    }
    auto rs_fetch = [&](int64_t row0, int *rs) {
        const int64_t grow = row0 + rs_r;
        const bool ok = grow < a.m_rows;
        int64_t logical = 0;
        if (ok) logical = a.row_perm ? (int64_t)a.row_perm[grow] : grow;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int r = 0;   // rows past the end read row 0 (never emitted)
            if (ron[h] && ok) r = rp32[h] ? rp32[h][logical] : (rp64[h] ? (int)rp64[h][logical] : (int)logical);
            rs[h] = r;
        }
    };
    auto rs_store = [&](int *dst, const int *rs) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (ron[h]) dst[((tid >> 7) + 4 * h) * BM + rs_r] = rs[h];
    };

    const int col = n0 + wn * 32 + li;
    const bool cok = col < a.n_out;
    const float e_bias = (cok && a.bias) ? a.bias[col] : 0.f;
    float e_scale = 1.f, e_c0 = e_bias;
    if (cok && a.bn_scale) { e_scale = a.bn_scale[col]; e_c0 = (e_bias - a.bn_mean[col]) * e_scale + a.bn_shift[col]; }

    double st_sum = 0.0, st_sq = 0.0;
    // D register sets of staged values: the global loads of slice g + D are issued while slice g computes
    float preAs[D][NPRE8], preWs[D][NPRE8];
    auto fetch = [&](float *preA, float *preW, const int *rs, int c) {
        const ColMapL cm = col_map_l(a, c * BK + kc);
        const int *rp = rs + cm.rsoff + r0;
        const int kg = c * BK + kc;
        const int kk = kg < a.k_total ? kg : 0;
        if (VEC4) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float4 v = *reinterpret_cast<const float4 *>(cm.base + (int64_t)rp[64 * i] * cm.bw);
                preA[4 * i] = v.x; preA[4 * i + 1] = v.y; preA[4 * i + 2] = v.z; preA[4 * i + 3] = v.w;
                int j = n0 + r0 + 64 * i;
                j = j < a.n_out ? j : 0;
                const float4 u = *reinterpret_cast<const float4 *>(a.W + (int64_t)j * a.k_total + kk);
                preW[4 * i] = u.x; preW[4 * i + 1] = u.y; preW[4 * i + 2] = u.z; preW[4 * i + 3] = u.w;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NPRE8; ++i) preA[i] = cm.base[(int64_t)rp[16 * i] * cm.bw];
#pragma unroll
        for (int i = 0; i < NPRE8; ++i) {
            int j = n0 + r0 + 16 * i;
            j = j < a.n_out ? j : 0;
            preW[i] = a.W[(int64_t)j * a.w_rs + (int64_t)kk * a.w_cs];
        }
    };
    // registers -> three bf16 planes.  Rows of W^T past K / columns past n_out are zeroed by a 0/1 factor applied to values
    // that were loaded one slice earlier (no select on fresh loads); A columns past K hold a finite clamped-address value.
    auto pack2 = [](unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); };
    auto stage = [&](const float *preA, const float *preW, int buf, int c) {
        const float km = (c * BK + kc < a.k_total) ? 1.f : 0.f;
        if (VEC4) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float jm = (n0 + r0 + 64 * i < a.n_out) ? km : 0.f;
                unsigned h[4], m[4], l[4], wh[4], wm_[4], wl[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { split3l(preA[4 * i + e], h[e], m[e], l[e]); split3l(preW[4 * i + e] * jm, wh[e], wm_[e], wl[e]); }
                const int o = ((r0 + 64 * i) * BKP + kc) / 2;
                float *pa = a_planes(buf) + o, *pw = w_planes(buf) + o;
                typedef unsigned u32x2l __attribute__((ext_vector_type(2)));
                u32x2l v;
                v[0] = pack2(h[0], h[1]); v[1] = pack2(h[2], h[3]); *reinterpret_cast<u32x2l *>(pa) = v;
                v[0] = pack2(m[0], m[1]); v[1] = pack2(m[2], m[3]); *reinterpret_cast<u32x2l *>(pa + BPLANE) = v;
                v[0] = pack2(l[0], l[1]); v[1] = pack2(l[2], l[3]); *reinterpret_cast<u32x2l *>(pa + 2 * BPLANE) = v;
                v[0] = pack2(wh[0], wh[1]); v[1] = pack2(wh[2], wh[3]); *reinterpret_cast<u32x2l *>(pw) = v;
                v[0] = pack2(wm_[0], wm_[1]); v[1] = pack2(wm_[2], wm_[3]); *reinterpret_cast<u32x2l *>(pw + BPLANE) = v;
                v[0] = pack2(wl[0], wl[1]); v[1] = pack2(wl[2], wl[3]); *reinterpret_cast<u32x2l *>(pw + 2 * BPLANE) = v;
            }
            return;
        }
        unsigned short *pa = reinterpret_cast<unsigned short *>(a_planes(buf)) + r0 * BKP + kc;
        unsigned short *pw = reinterpret_cast<unsigned short *>(w_planes(buf)) + r0 * BKP + kc;
#pragma unroll
        for (int i = 0; i < NPRE8; ++i) {
            unsigned h, m, l;
            split3l(preA[i], h, m, l);
            pa[16 * i * BKP] = (unsigned short)(h >> 16);
            pa[16 * i * BKP + 2 * BPLANE] = (unsigned short)(m >> 16);
            pa[16 * i * BKP + 4 * BPLANE] = (unsigned short)(l >> 16);
            const float jm = (n0 + r0 + 16 * i < a.n_out) ? km : 0.f;
            split3l(preW[i] * jm, h, m, l);
            pw[16 * i * BKP] = (unsigned short)(h >> 16);
            pw[16 * i * BKP + 2 * BPLANE] = (unsigned short)(m >> 16);
            pw[16 * i * BKP + 4 * BPLANE] = (unsigned short)(l >> 16);
        }
    };

    // One flat loop over the slices of all tiles of this workgroup (g = running slice number, set g % D holds its staged
    // values): stage slice g | barrier | issue the loads of slice g + D | MFMAs of slice g | (last slice of a tile) epilogue.
    // The row-source table of the next tile is loaded at the start of a tile and stored D slices before its end.
    const int64_t tile0 = blockIdx.x;
    const int64_t n_mine = tile0 < n_tiles ? (n_tiles - tile0 + gridDim.x - 1) / gridDim.x : 0;
    int rs_next[2];
    rs_fetch(tile0 * BM, rs_next);
    rs_store(rsrc, rs_next);
    __syncthreads();
    if (n_mine > 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) fetch(preAs[d], preWs[d], rsrc, d);      // (D <= n_slices: both in the first tile)
    }
    int slot = 0, cur = 0, c = 0;
    int64_t ti = 0;                                                     // ordinal of the current tile among this workgroup's
    f32x16 acc[2];

#define GSN_MFL(acc, x, y) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8l, x), __builtin_bit_cast(bf16x8l, y), acc, 0, 0, 0)
    auto body = [&](float *preA, float *preW) {
        const int64_t tile = tile0 + ti * gridDim.x;
        const bool has_next = ti + 1 < n_mine;
        if (c == 0) {
            rs_fetch((tile + gridDim.x) * BM, rs_next);                 // lands while this tile computes
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        }
        stage(preA, preW, cur, c);
        if (c == n_slices - D && has_next) rs_store(rsrc + (slot ^ 1) * (MAX_BLOCKS * BM), rs_next);
        lds_barrier_l();
        {   // loads of the slice D ahead, into the register set just consumed
            const int c2 = c + D;
            if (c2 < n_slices) fetch(preA, preW, rsrc + slot * (MAX_BLOCKS * BM), c2);
            else if (has_next) fetch(preA, preW, rsrc + (slot ^ 1) * (MAX_BLOCKS * BM), c2 - n_slices);
        }
        const float *ap = a_planes(cur) + ((wm * 64 + li) * BKP + 8 * lh) / 2;
        const float *bp = w_planes(cur) + ((wn * 32 + li) * BKP + 8 * lh) / 2;
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            const u32x4l bh = *reinterpret_cast<const u32x4l *>(bp + 8 * s);
            const u32x4l bm = *reinterpret_cast<const u32x4l *>(bp + 8 * s + BPLANE);
            const u32x4l bl = *reinterpret_cast<const u32x4l *>(bp + 8 * s + 2 * BPLANE);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float *api = ap + i * (32 * BKP / 2);
                const u32x4l ah = *reinterpret_cast<const u32x4l *>(api + 8 * s);
                const u32x4l am = *reinterpret_cast<const u32x4l *>(api + 8 * s + BPLANE);
                const u32x4l al = *reinterpret_cast<const u32x4l *>(api + 8 * s + 2 * BPLANE);
                GSN_MFL(acc[i], al, bh); GSN_MFL(acc[i], ah, bl); GSN_MFL(acc[i], am, bm);     // small terms first
                GSN_MFL(acc[i], ah, bm); GSN_MFL(acc[i], am, bh); GSN_MFL(acc[i], ah, bh);
            }
        }
        cur ^= 1;
        if (++c < n_slices) return;
        // epilogue.  C layout of a 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
        const int64_t row0 = tile * BM;
        const bool full = (row0 + BM <= a.m_rows) && (n0 + BN <= a.n_out);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int64_t rbase = row0 + wm * 64 + i * 32 + 4 * lh;
            float *op = a.out + rbase * a.n_out + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2);
                if (!full && (!cok || rbase + dr >= a.m_rows)) continue;
                if (STATS) {
                    const float h = acc[i][r] + e_bias;
                    st_sum += (double)h;
                    st_sq += (double)h * (double)h;
                    if (a.out) op[(int64_t)dr * a.n_out] = h;          // statistics AND the raw pre-BN rows in one pass
                } else {
                    op[(int64_t)dr * a.n_out] = apply_act(fmaf(acc[i][r], e_scale, e_c0), a.act);
                }
            }
        }
        c = 0; slot ^= 1; ++ti;
    };
    const int64_t total = n_mine * n_slices;
    for (int64_t g = 0; g < total; g += D) {
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (g + d < total) body(preAs[d], preWs[d]);
    }
#undef GSN_MFL

    if (STATS) {
        double s = st_sum, q = st_sq;
        s += __shfl_xor(s, 32);
        q += __shfl_xor(q, 32);
        if (lh == 0 && cok) {
            atomicAdd(&a.stats[col], s);
            atomicAdd(&a.stats[a.n_out + col], q);
        }
    }
}

template <bool STATS, bool VEC4, int D>
static int launch_linear_bf16_impl(const LinArgs &a, int k_pad, int64_t n_tiles, int col_tiles, hipStream_t st) {
    const size_t lds = ((size_t)12 * BPLANE + (size_t)2 * MAX_BLOCKS * BM) * 4;
    static DeviceOnce attr_set;
    const int attr_dev = current_device();
    if (!attr_set.done(attr_dev)) {
        hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_fwd_bf16_kernel<STATS, VEC4, D>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e0 != hipSuccess) return set_error(GSN_E_HIP, "hipFuncSetAttribute(linear_fwd_bf16_kernel): %s", hipGetErrorString(e0));
        attr_set.mark(attr_dev);
    }
    int64_t gx = 256 / col_tiles;   // persistent: one workgroup per CU in total
    if (gx < 1) gx = 1;
    if (gx > n_tiles) gx = n_tiles;
    hipLaunchKernelGGL((linear_fwd_bf16_kernel<STATS, VEC4, D>), dim3((unsigned)gx, (unsigned)col_tiles), dim3(512), lds, st, a, k_pad);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "linear_fwd_bf16_kernel: %s", hipGetErrorString(e));
    return GSN_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The same product on 32-row tiles, for row counts at which the 128-row tiles above leave most of the chip idle (a training step at the
// reference's batch sizes: M = 10^3 .. 10^4 rows is 8 .. 50 tiles of 128 rows on 256 CUs, each a serial chain of K / 32 slices x 48 MFMAs).
// Workgroup = 4 waves = 32 rows x 128 output columns, wave w owns columns 32 w .. + 32 (one accumulator tile); one tile per workgroup; the
// same staging (three bf16 planes per operand, truncation split), the same six plane products in the same order and the same K order, so a
// row's result does not depend on which of the two kernels computed it.  W is re-staged by four times as many workgroups (from L2).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int SBM = 32;
constexpr int SAPLANE = SBM * BKP / 2;            // floats per A plane
constexpr int SBUF = 3 * SAPLANE + 3 * BPLANE;    // floats per buffer: A planes, W planes

// SPLITK: gridDim.z workgroups share an output tile, each takes a contiguous range of the K slices and ADDS its partial product to `out`
// (zeros on entry; float atomics).  For products whose epilogue is the identity (+ bias, added by the first range): the input-gradient products
// gX = gH W of a dense backward at the reference's batch sizes, where one workgroup per tile is a serial chain of K / 32 slices of ~1.1 us each
// on a third of the chip (M = 837 rows: 81 .. 135 workgroups; 22.8 us at K = 600).
template <bool STATS, bool VEC4, bool WT, bool SPLITK = false>
__global__ __launch_bounds__(256) void linear_fwd_bf16_small_kernel(LinArgs a, int k_pad) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    auto a_planes = [&](int buf) { return lds + buf * SBUF; };
    auto w_planes = [&](int buf) { return lds + buf * SBUF + 3 * SAPLANE; };
    int *rsrc = reinterpret_cast<int *>(lds + 2 * SBUF);       // [MAX_BLOCKS][SBM]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wn = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    // scalar staging: column kc of rows r0 + 8 i (4) and of output columns r0 + 8 i (16); VEC4: columns kc .. kc + 3 of row r0 and of output columns r0 + 32 i (4)
    const int kc = VEC4 ? 4 * (tid & 7) : (tid & 31), r0 = VEC4 ? (tid >> 3) : (tid >> 5);
    const int n0 = blockIdx.y * BN;
    const int64_t row0 = (int64_t)blockIdx.x * SBM;
    const int n_slices_all = k_pad / BK;
    const int c_first = SPLITK ? (int)((int64_t)blockIdx.z * n_slices_all / gridDim.z) : 0;
    const int n_slices = SPLITK ? (int)((int64_t)(blockIdx.z + 1) * n_slices_all / gridDim.z) : n_slices_all;      // (one past this range's last slice)

    if (tid < MAX_BLOCKS * SBM) {
        const int b = tid >> 5;
        const int64_t grow = row0 + (tid & 31);
        int r = 0;                      // rows past the end read row 0 (never emitted)
        if (b < a.n_blocks && grow < a.m_rows) {
            const int64_t logical = a.row_perm ? (int64_t)a.row_perm[grow] : grow;
            r = a.bidx32[b] ? a.bidx32[b][logical] : (a.bidx[b] ? (int)a.bidx[b][logical] : (int)logical);
        }
        rsrc[tid] = r;
    }
    __syncthreads();

    const int col = n0 + wn * 32 + li;
    const bool cok = col < a.n_out;
    const float e_bias = (cok && a.bias) ? a.bias[col] : 0.f;
    float e_scale = 1.f, e_c0 = e_bias;
    if (cok && a.bn_scale) { e_scale = a.bn_scale[col]; e_c0 = (e_bias - a.bn_mean[col]) * e_scale + a.bn_shift[col]; }

    // SD register sets of staged values: the loads of slice c + SD are issued while slice c computes (a slice's matrix phase is 12 MFMAs per
    // wave, far shorter than a round trip to L2: one slice ahead left the kernel waiting for its loads in every iteration)
    constexpr int SD = 3;
    float preAs[SD][4], preWs[SD][16];
    // WT (a transposed view of a row-major matrix, w_rs = 1): output column jj of W^T rows 16 k0 .. 16 k0 + 15 -- the lanes walk the contiguous
    // direction, and a thread's sixteen values are the 32 contiguous bytes of a plane row: two 16-byte LDS writes per plane (a thread with
    // every other row k0 + 2 i wrote 48 two-byte pieces per slice -- the slice time of these launches is their instruction count, one wave per SIMD)
    const int jj = tid & 127, k0 = tid >> 7;
    auto fetch = [&](float *preA, float *preW, int c) {
        const ColMapL cm = col_map_l(a, c * BK + kc);
        const int *rp = rsrc + (cm.rsoff / (BM / SBM)) + r0;
        const int kg = c * BK + kc;
        const int kk = kg < a.k_total ? kg : 0;
        if (VEC4) {
            const float4 v = *reinterpret_cast<const float4 *>(cm.base + (int64_t)rp[0] * cm.bw);
            preA[0] = v.x; preA[1] = v.y; preA[2] = v.z; preA[3] = v.w;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int j = n0 + r0 + 32 * i;
                j = j < a.n_out ? j : 0;
                const float4 u = *reinterpret_cast<const float4 *>(a.W + (int64_t)j * a.k_total + kk);
                preW[4 * i] = u.x; preW[4 * i + 1] = u.y; preW[4 * i + 2] = u.z; preW[4 * i + 3] = u.w;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) preA[i] = cm.base[(int64_t)rp[8 * i] * cm.bw];
        if (WT) {
            // 32-bit BYTE offsets from the matrix base (the launcher takes this instantiation only when the view spans < 4 GiB): a load is
            // base (scalar) + offset (one register), the next row one add -- a 64-bit multiply per element was ~100 of a slice's ~400 vector
            // instructions, and with one wave per SIMD a slice's time IS its instruction count
            const unsigned j = n0 + jj < a.n_out ? (unsigned)(n0 + jj) : 0u;
            const int kb = c * BK + 16 * k0;
            const unsigned cs4 = 4u * (unsigned)a.w_cs;
            const char *wb = reinterpret_cast<const char *>(a.W);
            if (kb + 16 <= a.k_total) {                 // (wave-uniform)
                unsigned off = 4u * j * (unsigned)a.w_rs + (unsigned)kb * cs4;
#pragma unroll
                for (int i = 0; i < 16; ++i) { preW[i] = *reinterpret_cast<const float *>(wb + off); off += cs4; }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int kgi = kb + i;
                    preW[i] = *reinterpret_cast<const float *>(wb + (4u * j * (unsigned)a.w_rs + (unsigned)(kgi < a.k_total ? kgi : 0) * cs4));
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int j = n0 + r0 + 8 * i;
            j = j < a.n_out ? j : 0;
            preW[i] = a.W[(int64_t)j * a.w_rs + (int64_t)kk * a.w_cs];
        }
    };
    auto pack2 = [](unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); };
    typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
    auto stage = [&](const float *preA, const float *preW, int buf, int c) {
        const float km = (c * BK + kc < a.k_total) ? 1.f : 0.f;
        if (VEC4) {
            {
                unsigned h[4], m[4], l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) split3l(preA[e], h[e], m[e], l[e]);
                float *pa = a_planes(buf) + (r0 * BKP + kc) / 2;
                u32x2s v;
                v[0] = pack2(h[0], h[1]); v[1] = pack2(h[2], h[3]); *reinterpret_cast<u32x2s *>(pa) = v;
                v[0] = pack2(m[0], m[1]); v[1] = pack2(m[2], m[3]); *reinterpret_cast<u32x2s *>(pa + SAPLANE) = v;
                v[0] = pack2(l[0], l[1]); v[1] = pack2(l[2], l[3]); *reinterpret_cast<u32x2s *>(pa + 2 * SAPLANE) = v;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float jm = (n0 + r0 + 32 * i < a.n_out) ? km : 0.f;
                unsigned h[4], m[4], l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) split3l(preW[4 * i + e] * jm, h[e], m[e], l[e]);
                float *pw = w_planes(buf) + ((r0 + 32 * i) * BKP + kc) / 2;
                u32x2s v;
                v[0] = pack2(h[0], h[1]); v[1] = pack2(h[2], h[3]); *reinterpret_cast<u32x2s *>(pw) = v;
                v[0] = pack2(m[0], m[1]); v[1] = pack2(m[2], m[3]); *reinterpret_cast<u32x2s *>(pw + BPLANE) = v;
                v[0] = pack2(l[0], l[1]); v[1] = pack2(l[2], l[3]); *reinterpret_cast<u32x2s *>(pw + 2 * BPLANE) = v;
            }
            return;
        }
        unsigned short *pa = reinterpret_cast<unsigned short *>(a_planes(buf)) + r0 * BKP + kc;
        unsigned short *pw = reinterpret_cast<unsigned short *>(w_planes(buf)) + r0 * BKP + kc;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned h, m, l;
            split3l(preA[i], h, m, l);
            pa[8 * i * BKP] = (unsigned short)(h >> 16);
            pa[8 * i * BKP + 2 * SAPLANE] = (unsigned short)(m >> 16);
            pa[8 * i * BKP + 4 * SAPLANE] = (unsigned short)(l >> 16);
        }
        if (WT) {
            float *pt = w_planes(buf) + (jj * BKP + 16 * k0) / 2;       // (80-byte rows, 32-byte halves: 16-byte aligned)
            const bool jok = n0 + jj < a.n_out;
            unsigned h[16], m[16], l[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float jm = (jok && c * BK + 16 * k0 + i < a.k_total) ? 1.f : 0.f;
                split3l(preW[i] * jm, h[i], m[i], l[i]);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                u32x4l vh, vm, vl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    vh[e] = pack2(h[8 * q + 2 * e], h[8 * q + 2 * e + 1]);
                    vm[e] = pack2(m[8 * q + 2 * e], m[8 * q + 2 * e + 1]);
                    vl[e] = pack2(l[8 * q + 2 * e], l[8 * q + 2 * e + 1]);
                }
                *reinterpret_cast<u32x4l *>(pt + 4 * q) = vh;
                *reinterpret_cast<u32x4l *>(pt + 4 * q + BPLANE) = vm;
                *reinterpret_cast<u32x4l *>(pt + 4 * q + 2 * BPLANE) = vl;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            unsigned h, m, l;
            const float jm = (n0 + r0 + 8 * i < a.n_out) ? km : 0.f;
            split3l(preW[i] * jm, h, m, l);
            pw[8 * i * BKP] = (unsigned short)(h >> 16);
            pw[8 * i * BKP + 2 * BPLANE] = (unsigned short)(m >> 16);
            pw[8 * i * BKP + 4 * BPLANE] = (unsigned short)(l >> 16);
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int d = 0; d < SD; ++d)
        if (c_first + d < n_slices) fetch(preAs[d], preWs[d], c_first + d);
    int cur = 0;
#define GSN_MFS(x, y) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8l, x), __builtin_bit_cast(bf16x8l, y), acc, 0, 0, 0)
    auto body = [&](float *preA, float *preW, int c) {
        stage(preA, preW, cur, c);
        lds_barrier_l();                        // (the buffer written two slices later is the one read here: a barrier lies between)
        if (c + SD < n_slices) fetch(preA, preW, c + SD);
        const float *ap = a_planes(cur) + (li * BKP + 8 * lh) / 2;
        const float *bp = w_planes(cur) + ((wn * 32 + li) * BKP + 8 * lh) / 2;
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            const u32x4l bh = *reinterpret_cast<const u32x4l *>(bp + 8 * s);
            const u32x4l bm = *reinterpret_cast<const u32x4l *>(bp + 8 * s + BPLANE);
            const u32x4l bl = *reinterpret_cast<const u32x4l *>(bp + 8 * s + 2 * BPLANE);
            const u32x4l ah = *reinterpret_cast<const u32x4l *>(ap + 8 * s);
            const u32x4l am = *reinterpret_cast<const u32x4l *>(ap + 8 * s + SAPLANE);
            const u32x4l al = *reinterpret_cast<const u32x4l *>(ap + 8 * s + 2 * SAPLANE);
            GSN_MFS(al, bh); GSN_MFS(ah, bl); GSN_MFS(am, bm);     // small terms first (the order of the 128-row kernel)
            GSN_MFS(ah, bm); GSN_MFS(am, bh); GSN_MFS(ah, bh);
        }
        cur ^= 1;
    };
    for (int c = c_first; c < n_slices; c += SD) {
#pragma unroll
        for (int d = 0; d < SD; ++d)
            if (c + d < n_slices) body(preAs[d], preWs[d], c + d);
    }
#undef GSN_MFS
    // epilogue.  C layout of a 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    double st_sum = 0.0, st_sq = 0.0;
    const int64_t rbase = row0 + 4 * lh;
    float *op = a.out + rbase * a.n_out + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        if (!cok || rbase + dr >= a.m_rows) continue;
        if (STATS) {
            const float h = acc[r] + e_bias;
            st_sum += (double)h;
            st_sq += (double)h * (double)h;
            if (a.out) op[(int64_t)dr * a.n_out] = h;
        } else if (SPLITK) {
            atomicAdd(op + (int64_t)dr * a.n_out, blockIdx.z == 0 ? acc[r] + e_bias : acc[r]);
        } else {
            op[(int64_t)dr * a.n_out] = apply_act(fmaf(acc[r], e_scale, e_c0), a.act);
        }
    }
    if (STATS) {
        double s = st_sum, q = st_sq;
        s += __shfl_xor(s, 32);
        q += __shfl_xor(q, 32);
        if (lh == 0 && cok) {
            atomicAdd(&a.stats[col], s);
            atomicAdd(&a.stats[a.n_out + col], q);
        }
    }
}

template <bool STATS, bool VEC4, bool WT = false>
static int launch_linear_bf16_small(const LinArgs &a, int k_pad, int col_tiles, hipStream_t st) {
    const size_t lds = ((size_t)2 * SBUF + (size_t)MAX_BLOCKS * SBM) * 4;
    static DeviceOnce attr_set;
    const int attr_dev = current_device();
    if (!attr_set.done(attr_dev)) {
        hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_fwd_bf16_small_kernel<STATS, VEC4, WT>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e0 != hipSuccess) return set_error(GSN_E_HIP, "hipFuncSetAttribute(linear_fwd_bf16_small_kernel): %s", hipGetErrorString(e0));
        attr_set.mark(attr_dev);
    }
    const int64_t gx = (a.m_rows + SBM - 1) / SBM;
    hipLaunchKernelGGL((linear_fwd_bf16_small_kernel<STATS, VEC4, WT>), dim3((unsigned)gx, (unsigned)col_tiles), dim3(256), lds, st, a, k_pad);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "linear_fwd_bf16_small_kernel: %s", hipGetErrorString(e));
    return GSN_OK;
}

template <bool VEC4, bool WT>
static int launch_linear_bf16_splitk(const LinArgs &a, int k_pad, int col_tiles, int splits, hipStream_t st) {
    const size_t lds = ((size_t)2 * SBUF + (size_t)MAX_BLOCKS * SBM) * 4;
    static DeviceOnce attr_set;
    const int attr_dev = current_device();
    if (!attr_set.done(attr_dev)) {
        hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_fwd_bf16_small_kernel<false, VEC4, WT, true>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e0 != hipSuccess) return set_error(GSN_E_HIP, "hipFuncSetAttribute(linear_fwd_bf16_small_kernel): %s", hipGetErrorString(e0));
        attr_set.mark(attr_dev);
    }
    const int64_t gx = (a.m_rows + SBM - 1) / SBM;
    hipLaunchKernelGGL((linear_fwd_bf16_small_kernel<false, VEC4, WT, true>), dim3((unsigned)gx, (unsigned)col_tiles, (unsigned)splits), dim3(256), lds, st, a, k_pad);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "linear_fwd_bf16_small_kernel (split K): %s", hipGetErrorString(e));
    return GSN_OK;
}

template <bool STATS, bool VEC4>
static int launch_linear_bf16(const LinArgs &a, int k_pad, int64_t n_tiles, int col_tiles, hipStream_t st) {
    // (D = 2, loads two slices ahead, was measured at the same speed: the kernel is bound by its lock-step phases --
    // staging, barrier, matrix phase -- not by the gathers' latency; a loader / matrix role split is the next step)
    return launch_linear_bf16_impl<STATS, VEC4, 1>(a, k_pad, n_tiles, col_tiles, st);
}

template <bool STATS, bool WRES>
static int launch_linear(const LinArgs &a, int k_pad, size_t lds, int64_t gx, int col_tiles, hipStream_t st) {
    static DeviceOnce attr_set;
    const int attr_dev = current_device();
    if (!attr_set.done(attr_dev)) {
        hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_fwd_kernel<STATS, WRES>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e0 != hipSuccess) return set_error(GSN_E_HIP, "hipFuncSetAttribute(linear_fwd_kernel): %s", hipGetErrorString(e0));
        attr_set.mark(attr_dev);
    }
    hipLaunchKernelGGL((linear_fwd_kernel<STATS, WRES>), dim3((unsigned)gx, (unsigned)col_tiles), dim3(256), lds, st, a, k_pad);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "linear_fwd_kernel: %s", hipGetErrorString(e));
    return GSN_OK;
}

}  // namespace gsn

using namespace gsn;

static int linear_fwd_impl(int64_t m_rows, int n_blocks, const gsn_block *blocks, const float *W, int64_t w_rs, int64_t w_cs, const float *bias,
                           int64_t n_out, const float *bn_mean, const float *bn_scale, const float *bn_shift, int act,
                           const int32_t *row_perm, float *out, double *stats, void *stream);

extern "C" int gsn_linear_fwd_hip(int64_t m_rows, int n_blocks, const gsn_block *blocks, const float *W, const float *bias,
                                  int64_t n_out, const float *bn_mean, const float *bn_scale, const float *bn_shift, int act,
                                  const int32_t *row_perm, float *out, double *stats, void *stream) {
    return linear_fwd_impl(m_rows, n_blocks, blocks, W, 0, 0, bias, n_out, bn_mean, bn_scale, bn_shift, act, row_perm, out, stats, stream);
}

// W given through element strides (W[j][k] at W + j * w_row_stride + k * w_col_stride): the input-gradient product gX = gH W of a dense stage
// reads the stage's own [n_out][K] weight as its transpose (row stride 1, column stride K) instead of a transposed copy per stage and step.
// The bf16x6 kernel's scalar staging path; GSN_E_UNSUPPORTED when that kernel is switched off.
extern "C" int gsn_linear_fwd_strided_hip(int64_t m_rows, int n_blocks, const gsn_block *blocks, const float *W, int64_t w_row_stride,
                                          int64_t w_col_stride, const float *bias, int64_t n_out, const float *bn_mean, const float *bn_scale,
                                          const float *bn_shift, int act, float *out, double *stats, void *stream) {
    if (w_row_stride < 1 || w_col_stride < 1) return set_error(GSN_E_INVALID, "gsn_linear_fwd_strided_hip: strides must be positive");
    return linear_fwd_impl(m_rows, n_blocks, blocks, W, w_row_stride, w_col_stride, bias, n_out, bn_mean, bn_scale, bn_shift, act, nullptr, out, stats, stream);
}

// Ranges of K slices per output tile (1: no split).  Taken where the 32-row tiles leave most CUs idle and the chain is long: at most two
// workgroups per CU in total, at least three slices per range, at most four ranges.  GSN_LINEAR_SPLITK_RANGES=n forces n ranges (0 or 1: off).
extern "C" int gsn_linear_splitk_plan(int64_t m_rows, int64_t k_total, int64_t n_out) {
    static const int forced = [] { const char *d = getenv("GSN_LINEAR_SPLITK_RANGES"); return d ? atoi(d) : -1; }();
    static const int small_max = [] { const char *d = getenv("GSN_LINEAR_SMALL_MAX"); return d ? atoi(d) : 96; }();
    static const bool bf16x6 = [] { const char *d = getenv("GSN_LINEAR_BF16X6"); return !(d && atoi(d) == 0); }();
    if (forced == 0 || !bf16x6 || m_rows <= 0 || k_total <= 0 || n_out <= 0) return 1;
    const int64_t n_tiles = (m_rows + BM - 1) / BM, col_tiles = (n_out + BN - 1) / BN;
    if (n_tiles * col_tiles > small_max) return 1;
    const int64_t n_slices = (k_total + BK - 1) / BK, wgs = (m_rows + SBM - 1) / SBM * col_tiles;
    int64_t s = forced > 0 ? forced : 4;
    if (forced <= 0) {
        if (s > n_slices / 3) s = n_slices / 3;
        if (s > 512 / wgs) s = 512 / wgs;
    }
    if (s > n_slices) s = n_slices;
    if (s > 16) s = 16;
    return s < 2 ? 1 : (int)s;
}

// out (zeros on entry) += blocks W^T (+ bias): the split-K form of gsn_linear_fwd_strided_hip / gsn_linear_fwd_hip (w_row_stride = 0: row-major W)
// for gsn_linear_splitk_plan(..) > 1 ranges; the sum's order over the ranges varies in the last bits, as the weight gradients' does.
extern "C" int gsn_linear_fwd_splitk_hip(int64_t m_rows, int n_blocks, const gsn_block *blocks, const float *W, int64_t w_row_stride,
                                         int64_t w_col_stride, const float *bias, int64_t n_out, float *out, void *stream) {
    if (n_blocks < 1 || n_blocks > MAX_BLOCKS || !blocks || !W || n_out <= 0 || !out)
        return set_error(GSN_E_INVALID, "gsn_linear_fwd_splitk_hip: need 1..%d input blocks, W, out and n_out > 0", MAX_BLOCKS);
    if (w_row_stride < 0 || w_col_stride < 0 || (w_row_stride == 0) != (w_col_stride == 0))
        return set_error(GSN_E_INVALID, "gsn_linear_fwd_splitk_hip: bad strides");
    if (m_rows <= 0) return GSN_OK;
    const bool strided = w_row_stride != 0;
    LinArgs a{};
    a.m_rows = m_rows; a.n_blocks = n_blocks;
    int k_total = 0;
    for (int b = 0; b < n_blocks; ++b) {
        if (!blocks[b].data || blocks[b].width <= 0 || blocks[b].width > (1 << 20))
            return set_error(GSN_E_INVALID, "gsn_linear_fwd_splitk_hip: block %d has no data or a bad width", b);
        a.bdata[b] = blocks[b].data; a.bidx[b] = blocks[b].idx; a.bidx32[b] = blocks[b].idx32; a.bwidth[b] = (int)blocks[b].width;
        k_total += (int)blocks[b].width;
    }
    a.k_total = k_total; a.n_out = (int)n_out; a.act = 0;
    a.w_rs = strided ? w_row_stride : k_total; a.w_cs = strided ? w_col_stride : 1;
    {
        const float *cb[MAX_BLOCKS]; int cw[MAX_BLOCKS];
        for (int b = 0; b < MAX_BLOCKS; ++b) { cb[b] = b < n_blocks ? a.bdata[b] : a.bdata[0]; cw[b] = b < n_blocks ? a.bwidth[b] : (1 << 27); }
        a.cb0 = cb[0]; a.cb1 = cb[1]; a.cb2 = cb[2]; a.cb3 = cb[3]; a.cb4 = cb[4];
        a.cw0 = cw[0]; a.cw1 = cw[1]; a.cw2 = cw[2]; a.cw3 = cw[3]; a.cw4 = cw[4];
    }
    a.W = W; a.bias = bias; a.out = out;
    const int splits = gsn_linear_splitk_plan(m_rows, k_total, n_out);
    if (splits < 2) return set_error(GSN_E_INVALID, "gsn_linear_fwd_splitk_hip: gsn_linear_splitk_plan gives one range for this shape (take gsn_linear_fwd_hip)");
    const int col_tiles = (int)((n_out + BN - 1) / BN);
    const int k_pad = (k_total + BK - 1) / BK * BK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    bool vec4 = !strided && (k_total & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0;
    for (int b = 0; b < n_blocks; ++b)
        if ((a.bwidth[b] & 3) || (reinterpret_cast<uintptr_t>(a.bdata[b]) & 15)) vec4 = false;
    if (vec4) return launch_linear_bf16_splitk<true, false>(a, k_pad, col_tiles, splits, st);
    if (strided && a.w_rs < a.w_cs && ((n_out - 1) * a.w_rs + (int64_t)(k_total - 1) * a.w_cs) < (int64_t(1) << 30))
        return launch_linear_bf16_splitk<false, true>(a, k_pad, col_tiles, splits, st);
    return launch_linear_bf16_splitk<false, false>(a, k_pad, col_tiles, splits, st);
}

static int linear_fwd_impl(int64_t m_rows, int n_blocks, const gsn_block *blocks, const float *W, int64_t w_rs, int64_t w_cs, const float *bias,
                           int64_t n_out, const float *bn_mean, const float *bn_scale, const float *bn_shift, int act,
                           const int32_t *row_perm, float *out, double *stats, void *stream) {
    if (n_blocks < 1 || n_blocks > MAX_BLOCKS || !blocks || !W || n_out <= 0)
        return set_error(GSN_E_INVALID, "gsn_linear_fwd_hip: need 1..%d input blocks, W and n_out > 0", MAX_BLOCKS);
    if (!out && !stats) return set_error(GSN_E_INVALID, "gsn_linear_fwd_hip: neither out nor stats given");
    if ((bn_scale != nullptr) != (bn_shift != nullptr) || (bn_scale != nullptr) != (bn_mean != nullptr))
        return set_error(GSN_E_INVALID, "gsn_linear_fwd_hip: bn_mean, bn_scale and bn_shift go together");
    if (act < 0 || act > 3) return set_error(GSN_E_INVALID, "gsn_linear_fwd_hip: act must be 0..3");
    if (m_rows <= 0) return GSN_OK;
    const bool strided = w_rs != 0;
    LinArgs a{};
    a.m_rows = m_rows; a.n_blocks = n_blocks;
    int k_total = 0;
    for (int b = 0; b < n_blocks; ++b) {
        if (!blocks[b].data || blocks[b].width <= 0 || blocks[b].width > (1 << 20))
            return set_error(GSN_E_INVALID, "gsn_linear_fwd_hip: block %d has no data or a bad width", b);
        a.bdata[b] = blocks[b].data; a.bidx[b] = blocks[b].idx; a.bidx32[b] = blocks[b].idx32; a.bwidth[b] = (int)blocks[b].width;
        k_total += (int)blocks[b].width;
    }
    a.k_total = k_total; a.n_out = (int)n_out; a.act = act;
    a.w_rs = strided ? w_rs : k_total; a.w_cs = strided ? w_cs : 1;
    {
        const float *cb[MAX_BLOCKS]; int cw[MAX_BLOCKS];
        for (int b = 0; b < MAX_BLOCKS; ++b) { cb[b] = b < n_blocks ? a.bdata[b] : a.bdata[0]; cw[b] = b < n_blocks ? a.bwidth[b] : (1 << 27); }
        a.cb0 = cb[0]; a.cb1 = cb[1]; a.cb2 = cb[2]; a.cb3 = cb[3]; a.cb4 = cb[4];
        a.cw0 = cw[0]; a.cw1 = cw[1]; a.cw2 = cw[2]; a.cw3 = cw[3]; a.cw4 = cw[4];
    }
    a.W = W; a.bias = bias; a.bn_mean = bn_mean; a.bn_scale = bn_scale; a.bn_shift = bn_shift;
    a.row_perm = row_perm; a.out = out; a.stats = stats;
    const int64_t n_tiles = (m_rows + BM - 1) / BM;
    const int col_tiles = (int)((n_out + BN - 1) / BN);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int k_pad = (k_total + BK - 1) / BK * BK;
    {   // default: the bf16x6 kernel (GSN_LINEAR_BF16X6=0 selects the fp32-MFMA kernel below for A/B measurements)
        static const bool bf16x6 = [] { const char *d = getenv("GSN_LINEAR_BF16X6"); return !(d && atoi(d) == 0); }();
        bool vec4 = !strided && (k_total & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0;   // float4 staging of A and W
        for (int b = 0; b < n_blocks; ++b)
            if ((a.bwidth[b] & 3) || (reinterpret_cast<uintptr_t>(a.bdata[b]) & 15)) vec4 = false;
        { const char *d = getenv("GSN_LINEAR_VEC4"); if (d && atoi(d) == 0) vec4 = false; }
        // few 128-row tiles: 32-row tiles on four times as many CUs (GSN_LINEAR_SMALL_MAX = the largest count of 128-row tiles that takes them, 0 = never)
        static const int small_max = [] { const char *d = getenv("GSN_LINEAR_SMALL_MAX"); return d ? atoi(d) : 96; }();
        if (bf16x6 && n_tiles * col_tiles <= small_max) {
            if (vec4) return stats ? launch_linear_bf16_small<true, true>(a, k_pad, col_tiles, st) : launch_linear_bf16_small<false, true>(a, k_pad, col_tiles, st);
            if (strided && w_rs < w_cs && ((n_out - 1) * w_rs + (int64_t)(k_total - 1) * w_cs) < (int64_t(1) << 30))     // a transposed view: staged along its contiguous direction (32-bit byte offsets)
                return stats ? launch_linear_bf16_small<true, false, true>(a, k_pad, col_tiles, st) : launch_linear_bf16_small<false, false, true>(a, k_pad, col_tiles, st);
            return stats ? launch_linear_bf16_small<true, false>(a, k_pad, col_tiles, st) : launch_linear_bf16_small<false, false>(a, k_pad, col_tiles, st);
        }
        if (bf16x6 && vec4) return stats ? launch_linear_bf16<true, true>(a, k_pad, n_tiles, col_tiles, st) : launch_linear_bf16<false, true>(a, k_pad, n_tiles, col_tiles, st);
        if (bf16x6) return stats ? launch_linear_bf16<true, false>(a, k_pad, n_tiles, col_tiles, st) : launch_linear_bf16<false, false>(a, k_pad, n_tiles, col_tiles, st);
    }
    if (strided) return set_error(GSN_E_UNSUPPORTED, "gsn_linear_fwd_strided_hip: the bf16x6 kernel is switched off");
    const size_t common = (size_t)2 * BM * APITCH * 4 + 2 * MAX_BLOCKS * BM * 4;
    const size_t lds_res = (size_t)k_pad * WPITCH * 4 + common;
    const size_t lds_str = (size_t)2 * BK * WPITCH * 4 + common;
    // resident W only when it leaves room for two workgroups per CU (they hide each other's staging); else stream W
    const bool wres = lds_res <= 78 * 1024;
    const size_t lds = wres ? lds_res : lds_str;
    int64_t gx = 512 / col_tiles;   // persistent: ~2 workgroups per CU in total
    if (gx < 1) gx = 1;
    if (gx > n_tiles) gx = n_tiles;
    if (stats) return wres ? launch_linear<true, true>(a, k_pad, lds, gx, col_tiles, st) : launch_linear<true, false>(a, k_pad, lds, gx, col_tiles, st);
    return wres ? launch_linear<false, true>(a, k_pad, lds, gx, col_tiles, st) : launch_linear<false, false>(a, k_pad, lds, gx, col_tiles, st);
}
