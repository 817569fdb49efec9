// Any-width dense stage on the fp16x3 scheme (the matrix arithmetic of layer_fused.hip) for DIRECT rows:
//
//     out[m, :] = act( bn( cat(blocks[0][m], blocks[1][m], ..) W^T + b ) )        (models_misc.py:52-58; node-level stages)
//
// -- the node product of a wide `general` layer (layers._split_edge_stage), its K = 260 node stage, the d = 300 ogb stages,
// jk projections: everything linear_fwd_bf16_kernel (six bf16 plane products, both operands split per K slice by every
// workgroup) handled at 85-100 TF/s fp32-equivalent.  Differences:
//   * two fp16 planes per operand after an exact power-of-two scaling, three plane products per fp32 product: half the matrix
//     work of bf16x6 at the same error (scripts/micro/bf16x6_check.hip);
//   * the weights are split ONCE (gsn_linear_f16x3_prepare_hip: per output column a power-of-two scale from its largest entry,
//     planes [2][n_out][K_pad] of fp16 in a caller buffer that lives as long as the weights do) instead of in every K slice of
//     every row tile: the kernel stages them with plain 8-byte copies;
//   * every input row gets its scale from a pre-pass over the row (lin16_rowscale_kernel: largest magnitude over ALL its
//     columns, so one accumulator serves the whole K loop); the epilogue multiplies the row's and the column's inverse scales
//     back in (exact), then bias / BatchNorm / activation as in linear.hip.
// Tiling as linear_fwd_bf16_kernel: persistent workgroups of 8 waves on 128 x 128 output tiles, 32-wide K slices, planes
// double-buffered in LDS (row pitch 80 bytes: conflict-free ds_read_b128), wave w = rows 64 (w >> 2).., columns 32 (w & 3)..
// Rows with an Inf / NaN come out NaN in all columns (documented deviation of the split-operand kernels).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "gsn_internal.h"

namespace gsn {

namespace {

constexpr int L_BM = 128, L_BN = 128, L_BK = 32;
constexpr int L_BKP = L_BK + 8;                    // fp16 row pitch of a plane
constexpr int L_PLANE = L_BM * L_BKP / 2;          // 32-bit words per plane
constexpr int L_MAXB = 5;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float fl2 __attribute__((ext_vector_type(2)));
typedef unsigned un4 __attribute__((ext_vector_type(4)));
typedef unsigned un2 __attribute__((ext_vector_type(2)));

struct L16Args {
    int64_t m_rows;
    int n_blocks;
    // (named fields, not arrays: a per-lane choice among array elements of the argument block is compiled into a vector LOAD from
    //  the argument segment, and the wait for that pointer drains every prefetch in flight)
    const float *b0, *b1, *b2, *b3, *b4;
    int w0, w1, w2, w3, w4;        // widths (0 past n_blocks)
    const _Float16 *wplanes;       // [2][n_out][k_pad]
    const float *colinv;           // [n_out] inverse column scales
    const float *bias, *bn_mean, *bn_scale, *bn_shift;
    int k_total, k_pad, n_out, act;
    float *rowscale;               // [2][m_rows]: scale, inverse
    float *out;
};

__device__ __forceinline__ void l16_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// power-of-two scale that puts a magnitude with these sign-less float bits into [2^14, 2^15), and its inverse (layer_fused.hip)
__device__ __forceinline__ void l16_scale(unsigned maxbits, float &scale, float &inv) {
    int e = (int)(maxbits >> 23);
    e = e < 15 ? 15 : (e > 254 ? 254 : e);
    scale = __uint_as_float((unsigned)(268 - e) << 23);
    inv = __uint_as_float((unsigned)(e - 14) << 23);
}

__device__ __forceinline__ void l16_split2(fl2 v, unsigned &hi, unsigned &lo) {
    const h16x2 h = __builtin_convertvector(v, h16x2);
    const fl2 r = v - __builtin_convertvector(h, fl2);
    const h16x2 l = __builtin_convertvector(r, h16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}

__device__ __forceinline__ float l16_act(float y, int act) {
    switch (act) {
        case 1: return y > 0.f ? y : 0.f;
        case 2: return y > 0.f ? y : expm1f(y);
        case 3: return tanhf(y);
        default: return y;
    }
}

// concatenated column kg -> (block base + column, block width); clamped to a valid address past K
struct L16Col {
    const float *base;
    int bw;
};
__device__ __forceinline__ L16Col l16_col(const L16Args &a, int kg) {
    const int p1 = a.w0, p2 = p1 + a.w1, p3 = p2 + a.w2, p4 = p3 + a.w3;        // first column of blocks 1 .. 4 (uniform)
    if (kg >= a.k_total) kg = 0;
    L16Col m;
    m.base = a.b0 + kg; m.bw = a.w0;
    if (kg >= p1) { m.base = a.b1 + (kg - p1); m.bw = a.w1; }
    if (kg >= p2) { m.base = a.b2 + (kg - p2); m.bw = a.w2; }
    if (kg >= p3) { m.base = a.b3 + (kg - p3); m.bw = a.w3; }
    if (kg >= p4) { m.base = a.b4 + (kg - p4); m.bw = a.w4; }
    return m;
}

// ---- weights: one wave per output column -----------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void lin16_prepare_kernel(const float *__restrict__ W, int n_out, int k_total, int k_pad, _Float16 *planes,
                                                          float *colinv) {
    const int j = blockIdx.x, lane = threadIdx.x;
    const float *w = W + (int64_t)j * k_total;
    unsigned m = 0;
    for (int k = lane; k < k_total; k += 64) m = max(m, __float_as_uint(w[k]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    float s, inv;
    l16_scale(m, s, inv);
    if (m >= 0x7f800000u) inv = __uint_as_float(0x7fc00000u);        // a non-finite weight: the whole output column is NaN
    _Float16 *ph = planes + (int64_t)j * k_pad, *pl = planes + ((int64_t)n_out + j) * k_pad;
    for (int k = lane; k < k_pad; k += 64) {
        const float v = k < k_total ? w[k] * s : 0.f;
        const _Float16 h = (_Float16)v;
        ph[k] = h;
        pl[k] = (_Float16)(v - (float)h);
    }
    if (lane == 0) colinv[j] = inv;
}

// ---- rows: 8 lanes per row, float4 chunks over the concatenated blocks ---------------------------------------------------------
__global__ __launch_bounds__(256) void lin16_rowscale_kernel(L16Args a) {
    const int64_t row = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
    const int q8 = threadIdx.x & 7;
    unsigned m = 0;
    if (row < a.m_rows) {
        for (int c = q8; 4 * c < a.k_total; c += 8) {
            const L16Col cm = l16_col(a, 4 * c);
            const float4 v = *reinterpret_cast<const float4 *>(cm.base + row * cm.bw);
            m = max(max(m, __float_as_uint(v.x) & 0x7fffffffu), __float_as_uint(v.y) & 0x7fffffffu);
            m = max(max(m, __float_as_uint(v.z) & 0x7fffffffu), __float_as_uint(v.w) & 0x7fffffffu);
        }
    }
    m = max(m, (unsigned)__shfl_xor((int)m, 1));
    m = max(m, (unsigned)__shfl_xor((int)m, 2));
    m = max(m, (unsigned)__shfl_xor((int)m, 4));
    if (row < a.m_rows && q8 == 0) {
        float s, inv;
        l16_scale(m, s, inv);
        if (m >= 0x7f800000u) inv = __uint_as_float(0x7fc00000u);    // Inf / NaN in the row: its outputs become NaN
        a.rowscale[row] = s;
        a.rowscale[a.m_rows + row] = inv;
    }
}

// ---- the product ------------------------------------------------------------------------------------------------------------------
// TWO workgroups per CU (exactly 80 KiB of LDS each, <= 128 registers): the staging phase of one runs under the matrix phase of
// the other -- inside one workgroup the slices are lock-step (stage | barrier | products).
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void linear_f16x3_kernel(L16Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned l16_lds[];
    // [2 buffers][A planes h, l | W planes h, l][L_PLANE words].  The row tables (scale, inverse scale of the tile's 128 rows, two
    // slots) live in the 16 padding bytes behind row r of the first plane: words 16 .. 19 of the row = scale0, inv0, scale1, inv1.
    auto a_planes = [&](int buf) { return l16_lds + buf * 4 * L_PLANE; };
    auto w_planes = [&](int buf) { return l16_lds + buf * 4 * L_PLANE + 2 * L_PLANE; };
    auto rtab = [&](int slot, int which, int r) -> float & { return reinterpret_cast<float *>(l16_lds)[r * (L_BKP / 2) + L_BK / 2 + 2 * slot + which]; };

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, lh = lane >> 5;
    const int kc = 4 * (tid & 7), r0 = tid >> 3;                          // staging: columns kc .. kc + 3 of rows / output columns r0, r0 + 64
    const int n0 = blockIdx.y * L_BN;
    const int64_t n_tiles = (a.m_rows + L_BM - 1) / L_BM;
    const int n_slices = a.k_pad / L_BK;

    for (int i = tid; i < 8 * L_PLANE; i += 512) l16_lds[i] = 0u;

    const int col = n0 + wn * 32 + li;
    const bool cok = col < a.n_out;
    const float e_bias = (cok && a.bias) ? a.bias[col] : 0.f;
    float e_scale = cok ? a.colinv[col] : 0.f, e_c0 = e_bias;
    if (cok && a.bn_scale) { e_c0 = (e_bias - a.bn_mean[col]) * a.bn_scale[col] + a.bn_shift[col]; e_scale *= a.bn_scale[col]; }

    const int64_t tile0 = blockIdx.x;
    const int64_t n_mine = tile0 < n_tiles ? (n_tiles - tile0 + gridDim.x - 1) / gridDim.x : 0;
    // row tables of a tile: thread t < 128 owns row t (scale), 128 <= t < 256 row t - 128 (inverse)
    auto rt_fetch = [&](int64_t row0) -> float {                        // (every thread loads: a clamped address, no branch)
        int64_t r = row0 + (tid & 127);
        r = r < a.m_rows ? r : a.m_rows - 1;
        return a.rowscale[((tid >> 7) & 1) * a.m_rows + r];
    };
    float rt_next = rt_fetch(tile0 * L_BM);
    __syncthreads();                                                     // (the zero fill above also covers the padding)
    if (tid < 256) rtab(0, tid >> 7, tid & 127) = rt_next;
    __syncthreads();

    // TWO register sets of staged values (A, B): the loads of slice c + 2 are issued while slice c computes -- one slice of
    // products does not cover the latency of an HBM / L2 miss, and inside a workgroup the slices are lock-step.  K is padded to a
    // multiple of 64, so a tile has an even number of slices and the sets keep their roles across tiles (compile-time names: a
    // run-time choice between the sets would make every stage wait for all loads in flight).
    float4 preA_a[2], preA_b[2];
    un2 preWh_a[2], preWl_a[2], preWh_b[2], preWl_b[2];
    auto fetch = [&](float4 *pA, un2 *pWh, un2 *pWl, int64_t row0, int c) {
        const L16Col cm = l16_col(a, c * L_BK + kc);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int64_t r = row0 + r0 + 64 * i;
            r = r < a.m_rows ? r : a.m_rows - 1;                        // rows past the end read the last row (never emitted)
            pA[i] = *reinterpret_cast<const float4 *>(cm.base + r * cm.bw);
            int j = n0 + r0 + 64 * i;
            j = j < a.n_out ? j : 0;                                    // (its output column is never emitted)
            const _Float16 *wp = a.wplanes + (int64_t)j * a.k_pad + c * L_BK + kc;
            pWh[i] = *reinterpret_cast<const un2 *>(wp);
            pWl[i] = *reinterpret_cast<const un2 *>(wp + (int64_t)a.n_out * a.k_pad);
        }
    };
    auto stage = [&](const float4 *pA, const un2 *pWh, const un2 *pWl, int buf, int slot) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float s = rtab(slot, 0, r0 + 64 * i);
            unsigned h0, l0, h1, l1;
            l16_split2(fl2{pA[i].x * s, pA[i].y * s}, h0, l0);
            l16_split2(fl2{pA[i].z * s, pA[i].w * s}, h1, l1);
            const int o = ((r0 + 64 * i) * L_BKP + kc) / 2;
            *reinterpret_cast<un2 *>(a_planes(buf) + o) = un2{h0, h1};
            *reinterpret_cast<un2 *>(a_planes(buf) + L_PLANE + o) = un2{l0, l1};
            *reinterpret_cast<un2 *>(w_planes(buf) + o) = pWh[i];
            *reinterpret_cast<un2 *>(w_planes(buf) + L_PLANE + o) = pWl[i];
        }
    };
    f32x16 acc[2];
#define L16_MF(acc, x, y) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, x), __builtin_bit_cast(h16x8, y), acc, 0, 0, 0)
    auto products = [&](int buf) {
        const unsigned *ap = a_planes(buf) + ((wm * 64 + li) * L_BKP + 8 * lh) / 2;
        const unsigned *bp = w_planes(buf) + ((wn * 32 + li) * L_BKP + 8 * lh) / 2;
        un4 bh[2], bl[2], ah[2][2], al[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {                                     // all twelve fragments first: the products then run back to back
            bh[s] = *reinterpret_cast<const un4 *>(bp + 8 * s);
            bl[s] = *reinterpret_cast<const un4 *>(bp + 8 * s + L_PLANE);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[s][i] = *reinterpret_cast<const un4 *>(ap + i * (32 * L_BKP / 2) + 8 * s);
                al[s][i] = *reinterpret_cast<const un4 *>(ap + i * (32 * L_BKP / 2) + 8 * s + L_PLANE);
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            L16_MF(acc[0], al[s][0], bh[s]); L16_MF(acc[1], al[s][1], bh[s]);     // small terms first; the two tiles alternate
            L16_MF(acc[0], ah[s][0], bl[s]); L16_MF(acc[1], ah[s][1], bl[s]);
            L16_MF(acc[0], ah[s][0], bh[s]); L16_MF(acc[1], ah[s][1], bh[s]);
        }
    };
    static_assert(L_BK == 32, "two k-steps per slice");
    if (n_mine > 0) {
        fetch(preA_a, preWh_a, preWl_a, tile0 * L_BM, 0);
        fetch(preA_b, preWh_b, preWl_b, tile0 * L_BM, 1);
    }
    int slot = 0;
    for (int64_t ti = 0; ti < n_mine; ++ti) {
        const int64_t tile = tile0 + ti * gridDim.x;
        const int64_t row0 = tile * L_BM, row_next = (tile + gridDim.x) * L_BM;
        const bool has_next = ti + 1 < n_mine;
        rt_next = rt_fetch(row_next);                                      // lands while this tile computes
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int c = 0; c < n_slices; c += 2) {
            const bool last = c + 2 >= n_slices;
            // what the two sets fetch next: slices c + 2, c + 3 of this tile, or slices 0, 1 of the next one (of this one again when
            // there is none: every slice issues the SAME number of loads, so the waits can be counted -- a conditional load in
            // between turns every wait into "all loads in flight")
            const int64_t row_f = last ? (has_next ? row_next : row0) : row0;
            const int c_f = last ? 0 : c + 2;
            // slice c: set A, buffer 0
            stage(preA_a, preWh_a, preWl_a, 0, slot);
            l16_barrier();
            fetch(preA_a, preWh_a, preWl_a, row_f, c_f);
            products(0);
            // slice c + 1: set B, buffer 1
            stage(preA_b, preWh_b, preWl_b, 1, slot);
            l16_barrier();
            if (last && tid < 256) rtab(slot ^ 1, tid >> 7, tid & 127) = rt_next;   // (read by the next tile after its first barrier)
            fetch(preA_b, preWh_b, preWl_b, row_f, c_f + 1);
            products(1);
        }
        // epilogue.  C layout of a 32x32 tile: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
        const bool full = (row0 + L_BM <= a.m_rows) && (n0 + L_BN <= a.n_out);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rl = wm * 64 + i * 32 + 4 * lh;                    // first of this lane's rows inside the tile
            const int64_t rbase = row0 + rl;
            float *op = a.out + rbase * a.n_out + col;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float ivv[4] = {rtab(slot, 1, rl + 8 * gq), rtab(slot, 1, rl + 8 * gq + 1), rtab(slot, 1, rl + 8 * gq + 2), rtab(slot, 1, rl + 8 * gq + 3)};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int dr = r + 8 * gq;
                    if (!full && (!cok || rbase + dr >= a.m_rows)) continue;
                    op[(int64_t)dr * a.n_out] = l16_act(fmaf(acc[i][4 * gq + r], ivv[r] * e_scale, e_c0), a.act);
                }
            }
        }
        slot ^= 1;
    }
#undef L16_MF
}

}  // namespace

}  // namespace gsn

using namespace gsn;

extern "C" int64_t gsn_linear_f16x3_kpad(int64_t k_total) { return (k_total + 2 * L_BK - 1) / (2 * L_BK) * (2 * L_BK); }

extern "C" int gsn_linear_f16x3_prepare_hip(const float *W, int64_t n_out, int64_t k_total, void *planes, float *col_inv, void *stream) {
    if (!W || !planes || !col_inv || n_out <= 0 || k_total <= 0 || n_out > (1 << 24) || k_total > (1 << 20))
        return set_error(GSN_E_INVALID, "gsn_linear_f16x3_prepare_hip: bad argument");
    const int k_pad = (int)gsn_linear_f16x3_kpad(k_total);
    hipLaunchKernelGGL(lin16_prepare_kernel, dim3((unsigned)n_out), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), W, (int)n_out, (int)k_total,
                       k_pad, reinterpret_cast<_Float16 *>(planes), col_inv);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "lin16_prepare_kernel: %s", hipGetErrorString(e));
    return GSN_OK;
}

extern "C" int gsn_linear_f16x3_fwd_hip(int64_t m_rows, int n_blocks, const gsn_block *blocks, const void *planes, const float *col_inv,
                                        const float *bias, int64_t n_out, const float *bn_mean, const float *bn_scale, const float *bn_shift,
                                        int act, float *row_scratch, float *out, void *stream) {
    if (n_blocks < 1 || n_blocks > L_MAXB || !blocks || !planes || !col_inv || !row_scratch || !out || n_out <= 0)
        return set_error(GSN_E_INVALID, "gsn_linear_f16x3_fwd_hip: need 1..%d input blocks, the weight planes, scratch and out", L_MAXB);
    if ((bn_scale != nullptr) != (bn_shift != nullptr) || (bn_scale != nullptr) != (bn_mean != nullptr))
        return set_error(GSN_E_INVALID, "gsn_linear_f16x3_fwd_hip: bn_mean, bn_scale and bn_shift go together");
    if (act < 0 || act > 3) return set_error(GSN_E_INVALID, "gsn_linear_f16x3_fwd_hip: act must be 0..3");
    if (m_rows <= 0) return GSN_OK;
    L16Args a{};
    a.m_rows = m_rows; a.n_blocks = n_blocks;
    int k_total = 0;
    const float *bd[L_MAXB] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int bw[L_MAXB] = {0, 0, 0, 0, 0};
    for (int b = 0; b < n_blocks; ++b) {
        if (!blocks[b].data || blocks[b].idx || blocks[b].idx32 || blocks[b].width <= 0 || (blocks[b].width & 3) ||
            (reinterpret_cast<uintptr_t>(blocks[b].data) & 15))
            return set_error(GSN_E_UNSUPPORTED, "gsn_linear_f16x3_fwd_hip: block %d: direct rows (no index), width a multiple of 4, 16-byte aligned", b);
        bd[b] = blocks[b].data; bw[b] = (int)blocks[b].width;
        k_total += (int)blocks[b].width;
    }
    for (int b = n_blocks; b < L_MAXB; ++b) { bd[b] = bd[0]; bw[b] = 1 << 28; }      // (never selected: their first column is past K)
    a.b0 = bd[0]; a.b1 = bd[1]; a.b2 = bd[2]; a.b3 = bd[3]; a.b4 = bd[4];
    a.w0 = bw[0]; a.w1 = bw[1]; a.w2 = bw[2]; a.w3 = bw[3]; a.w4 = bw[4];
    a.k_total = k_total; a.k_pad = (int)gsn_linear_f16x3_kpad(k_total); a.n_out = (int)n_out; a.act = act;
    a.wplanes = reinterpret_cast<const _Float16 *>(planes); a.colinv = col_inv;
    a.bias = bias; a.bn_mean = bn_mean; a.bn_scale = bn_scale; a.bn_shift = bn_shift;
    a.rowscale = row_scratch; a.out = out;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(lin16_rowscale_kernel, dim3((unsigned)((m_rows + 31) / 32)), dim3(256), 0, st, a);
    const size_t lds = (size_t)8 * L_PLANE * 4;
    static DeviceOnce attr_set;
    const int attr_dev = current_device();
    if (!attr_set.done(attr_dev)) {
        hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_f16x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e0 != hipSuccess) return set_error(GSN_E_HIP, "hipFuncSetAttribute(linear_f16x3_kernel): %s", hipGetErrorString(e0));
        attr_set.mark(attr_dev);
    }
    const int64_t n_tiles = (m_rows + L_BM - 1) / L_BM;
    const int col_tiles = (int)((n_out + L_BN - 1) / L_BN);
    int64_t gx = 512 / col_tiles;                                            // persistent: two workgroups per CU in total
    if (gx < 1) gx = 1;
    if (gx > n_tiles) gx = n_tiles;
    if (getenv("GSN_CHAIN_TRACE")) {
        int nb = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, linear_f16x3_kernel, 512, lds);
        fprintf(stderr, "gsn linear: linear_f16x3_kernel M %lld K %d N %d grid %lld x %d, %d workgroup(s) per CU\n", (long long)m_rows, k_total, (int)n_out,
                (long long)gx, col_tiles, nb);
    }
    hipLaunchKernelGGL(linear_f16x3_kernel, dim3((unsigned)gx, (unsigned)col_tiles), dim3(512), lds, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "linear_f16x3_kernel: %s", hipGetErrorString(e));
    return GSN_OK;
}
