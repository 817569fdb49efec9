// Backward of a dense models_misc.mlp stage on gfx950 (the adjoint of linear.hip / chain.hip for inputs that are plain
// row-major blocks):   H = X W^T + b ,  Z = (H - mean) * scale + shift ,  Y = act(Z)        (models_misc.py:52-59)
//
//   gZ = gY * act'(Y)                                   (relu / elu / tanh / identity: all derivable from Y)
//   train-mode BatchNorm1d (batch statistics over the M rows):  xh = (H - mean) * invstd
//        g_beta = sum_r gZ ,  g_gamma = sum_r gZ * xh ,  gH = gamma * invstd * (gZ - g_beta / M - xh * g_gamma / M)
//   eval-mode / no BN:  gH = gZ * scale   (train_bn == 2: eval-mode BatchNorm whose gamma / beta want gradients -- the same two
//        column sums with xh from the running statistics, gH = gZ * gamma * invstd)
//   gX = gH W   (gsn_linear_fwd_hip with W^T as the weight) ,   gW = gH^T X ,   gb = sum_r gH
//
// Kernels here: the two elementwise / column-reduction passes of the BN + activation adjoint (HBM-bound, fp64 column
// sums) and the weight-gradient GEMM  gW[n_out][K] = gH^T X  on v_mfma_f32_32x32x2_f32, where the contraction runs over
// the M rows: every MFMA consumes two rows (A[i][k] = gH[row k][col i], B[k][j] = X[row k][col j]), a workgroup owns one
// 128 x 128 tile of gW and one slab of rows, accumulates in registers and adds its partial tile with float atomics.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "gsn_internal.h"

namespace gsn {

__device__ __forceinline__ float act_grad_from_y(float y, int act) {
    switch (act) {
        case 1: return y > 0.f ? 1.f : 0.f;
        case 2: return y > 0.f ? 1.f : y + 1.f;          // elu: d/dz (e^z - 1) = y + 1 for z <= 0
        case 3: return 1.f - y * y;                      // tanh
        default: return 1.f;
    }
}

// the same derivative from the pre-activation value z (the stage output is not read: z is recomputed from the pre-BatchNorm rows with the
// forward pass's own expression, bn_act_kernel in encode.hip)
__device__ __forceinline__ float act_grad_from_z(float z, int act) {
    switch (act) {
        case 1: return z > 0.f ? 1.f : 0.f;
        case 2: return z > 0.f ? 1.f : expm1f(z) + 1.f;
        case 3: { const float t = tanhf(z); return 1.f - t * t; }
        default: return 1.f;
    }
}

// pass 1: column sums of gZ and gZ * xh (fp64 [2][C], caller zero-fills).  grid (blocks_x, ceil(C/64)); a wave covers 64
// adjacent columns of one row, waves stride over rows.
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(int64_t m_rows, int n_cols, const float *gy, const float *y,
                                                                const float *h, const float *mean, const float *invstd,
                                                                int act, double *sums, const float *zscale, const float *zshift) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + lane;
    const bool cok = c < n_cols;
    const float mu = (cok && mean) ? mean[c] : 0.f, is = (cok && invstd) ? invstd[c] : 1.f;
    const float zs = (cok && zscale) ? zscale[c] : 1.f, zb = (cok && zshift) ? zshift[c] : 0.f;
    double s1 = 0.0, s2 = 0.0;
    if (cok) {
        const int64_t step = (int64_t)gridDim.x * 4;
        int64_t r = (int64_t)blockIdx.x * 4 + wave;
        // four rows per pass: their loads are in flight together (the pass is a stream over M x C; one row per iteration left a wave with
        // two loads in flight), summed in the same order
        for (; r + 3 * step < m_rows; r += 4 * step) {
            float g4[4], h4[4], y4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = (r + u * step) * n_cols + c;
                g4[u] = gy[i]; h4[u] = h[i]; y4[u] = y ? y[i] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float gz = g4[u] * (y ? act_grad_from_y(y4[u], act) : act_grad_from_z((h4[u] - mu) * zs + zb, act));
                s1 += (double)gz;
                s2 += (double)gz * (double)((h4[u] - mu) * is);
            }
        }
        for (; r < m_rows; r += step) {
            const int64_t i = r * n_cols + c;
            const float gz = gy[i] * (y ? act_grad_from_y(y[i], act) : act_grad_from_z((h[i] - mu) * zs + zb, act));
            s1 += (double)gz;
            s2 += (double)gz * (double)((h[i] - mu) * is);
        }
    }
    __shared__ double red[2][4][64];
    red[0][wave][lane] = s1;
    red[1][wave][lane] = s2;
    __syncthreads();
    if (wave == 0 && cok) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < 4; ++w) { a += red[0][w][lane]; b += red[1][w][lane]; }
        atomicAdd(&sums[c], a);
        atomicAdd(&sums[n_cols + c], b);
    }
}

// pass 2: gH (may alias gy).  train: coef = gamma * invstd, m1 = sums[0]/M, m2 = sums[1]/M;  otherwise gH = gZ * coef.
// Also the column sums of gH (fp64 [C], caller zero-fills) = the bias gradient.
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(int64_t m_rows, int n_cols, const float *gy, const float *y,
                                                               const float *h, const float *mean, const float *invstd,
                                                               const float *coef, const double *sums, int act, float *gh,
                                                               double *gbias, const float *zshift) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + lane;
    const bool cok = c < n_cols;
    const float mu = (cok && mean) ? mean[c] : 0.f, is = (cok && invstd) ? invstd[c] : 1.f;
    const float cf = (cok && coef) ? coef[c] : 1.f;
    const float zb = (cok && zshift) ? zshift[c] : 0.f;        // (y == null: z = (h - mean) * coef + shift, coef = gamma * invstd)
    const float m1 = (cok && sums) ? (float)(sums[c] / (double)m_rows) : 0.f;
    const float m2 = (cok && sums) ? (float)(sums[n_cols + c] / (double)m_rows) : 0.f;
    double sb = 0.0;
    if (cok) {
        const int64_t step = (int64_t)gridDim.x * 4;
        int64_t r = (int64_t)blockIdx.x * 4 + wave;
        const bool need_h = !y || sums;
        auto one = [&](int64_t i, float gyv, float yv, float hv) {
            const float gz = gyv * (y ? act_grad_from_y(yv, act) : act_grad_from_z((hv - mu) * cf + zb, act));
            float g = gz;
            if (sums) g = gz - m1 - (hv - mu) * is * m2;
            g *= cf;
            gh[i] = g;
            sb += (double)g;
        };
        for (; r + 3 * step < m_rows; r += 4 * step) {      // (four rows' loads in flight, as in the reduce pass)
            float g4[4], h4[4], y4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = (r + u * step) * n_cols + c;
                g4[u] = gy[i]; y4[u] = y ? y[i] : 0.f; h4[u] = need_h ? h[i] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) one((r + u * step) * n_cols + c, g4[u], y4[u], h4[u]);
        }
        for (; r < m_rows; r += step) {
            const int64_t i = r * n_cols + c;
            one(i, gy[i], y ? y[i] : 0.f, need_h ? h[i] : 0.f);
        }
    }
    if (gbias) {
        __shared__ double red[4][64];
        red[wave][lane] = sb;
        __syncthreads();
        if (wave == 0 && cok) atomicAdd(&gbias[c], red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
    }
}

// pass 2 for a stage whose gH is consumed as fp16 PLANES only (r06): the input-gradient product gX = gH W on the fp16x3 kernel and the weight
// gradient gsn_wgrad_f16x3_hip both read the row scratch of gH (inverse row scales + a high and a low half per value, linear_f16.hip's layout) --
// so this pass writes that scratch and no fp32 gH at all: one read of gY and H, one write of 4 bytes per value, instead of apply (read 2, write
// 1) + row pre-pass (read 1, write 1).  Row-oriented like the pre-pass it replaces: 8 lanes per row, float4 chunks, a row's values stay in
// registers between the pass that finds its largest magnitude and the one that writes the planes (rows up to 640 columns).  The per-column
// vectors (mean, invstd, coef, shift, m1, m2) sit in LDS.  Same expression per value as bn_act_bwd_apply_kernel.
// The bias gradient = the column sums of gH: in train mode sum_r gH = coef (S1 - M m1 - m2 sum_r xhat) = 0 up to the rounding of the batch mean
// (the batch statistics absorb any shift of H: the true derivative is zero); on running statistics it is coef * S1.  Written by workgroup 0.
constexpr int BP_NCH = 20;         // float4 chunks per lane: rows up to 8 * 20 * 4 = 640 columns
typedef _Float16 bp_h16x2 __attribute__((ext_vector_type(2)));
typedef float bp_fl2 __attribute__((ext_vector_type(2)));
typedef unsigned bp_un4 __attribute__((ext_vector_type(4)));

// a row's values -> its scale, inverse scale and the two fp16 planes (linear_f16.hip's row scratch: l16_scale, l16_split2 and the lane-pair trade
// of lin16_split_rows_kernel); all 8 lanes of the row call it
__device__ __forceinline__ void bp_write_row(const float4 (&v)[BP_NCH], unsigned m, int q8, bool even, int64_t row, bool on, float *rowinv,
                                             unsigned char *planes, int kp, int nchp) {
    m = max(m, (unsigned)__shfl_xor((int)m, 1));
    m = max(m, (unsigned)__shfl_xor((int)m, 2));
    m = max(m, (unsigned)__shfl_xor((int)m, 4));
    // power-of-two scale that puts the row's largest magnitude into [2^14, 2^15), and its inverse (linear_f16.hip: l16_scale)
    int e = (int)(m >> 23);
    e = e < 15 ? 15 : (e > 254 ? 254 : e);
    const float s = __uint_as_float((unsigned)(268 - e) << 23);
    float inv = __uint_as_float((unsigned)(e - 14) << 23);
    if (m >= 0x7f800000u) inv = __uint_as_float(0x7fc00000u);            // Inf / NaN in the row: whatever reads its planes gets NaN
    if (q8 == 0) rowinv[row] = on ? inv : 0.f;
    unsigned char *prow = planes + row * (int64_t)kp * 4;
#pragma unroll
    for (int i = 0; i < BP_NCH; ++i) {
        const int c = q8 + 8 * i;
        if (c < nchp) {                       // (uniform over the lane pairs that trade halves: c and c ^ 1 lie on the same side of nchp, a multiple of 8)
            auto split2 = [](float x0, float x1, unsigned &hi, unsigned &lo) {
                const bp_h16x2 hv = __builtin_convertvector(bp_fl2{x0, x1}, bp_h16x2);
                const bp_fl2 r = bp_fl2{x0, x1} - __builtin_convertvector(hv, bp_fl2);
                const bp_h16x2 lv = __builtin_convertvector(r, bp_h16x2);
                hi = __builtin_bit_cast(unsigned, hv);
                lo = __builtin_bit_cast(unsigned, lv);
            };
            unsigned h0, l0, h1, l1;
            split2(v[i].x * s, v[i].y * s, h0, l0);
            split2(v[i].z * s, v[i].w * s, h1, l1);
            const unsigned r0 = (unsigned)__shfl_xor((int)(even ? l0 : h0), 1), r1 = (unsigned)__shfl_xor((int)(even ? l1 : h1), 1);
            unsigned char *line = prow + (c >> 3) * 128 + (even ? 0 : 64) + ((c & 7) >> 1) * 16;
            *reinterpret_cast<bp_un4 *>(line) = even ? bp_un4{h0, h1, r0, r1} : bp_un4{r0, r1, l0, l1};
        }
    }
}

struct BwdPlanesArgs {
    int64_t m_rows, m_pad;
    int n_cols, k_pad, act, train;  // train 1: batch statistics, 2: running statistics
    const float *gy, *h, *mean, *invstd, *coef, *shift;
    const double *sums;             // [2][n_cols] of the reduce pass
    float *rowinv;
    unsigned char *planes;
    double *gbias;
};

#ifndef BP_WAVES
#define BP_WAVES 2                 // waves per SIMD the adjoint plane kernel's register allocation aims at (3: spills, 300 -> 388 us at 105 k x 600)
#endif
#ifndef BP_WAVES_FWD
#define BP_WAVES_FWD 3             // ... and the forward one's (2: 152 us, 3: 128 us, 4: 147 us at 105 k x 600)
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BP_WAVES, BP_WAVES))) void bn_act_bwd_planes_kernel(BwdPlanesArgs a) {
    extern __shared__ __attribute__((aligned(16))) float bp_tab[];    // [6][k_pad]: mean | invstd | coef | shift | m1 | m2
    const int kp = a.k_pad;
    for (int c = threadIdx.x; c < kp; c += 256) {
        const bool ok = c < a.n_cols;
        bp_tab[c] = ok ? a.mean[c] : 0.f;
        bp_tab[kp + c] = ok ? a.invstd[c] : 0.f;
        bp_tab[2 * kp + c] = ok ? a.coef[c] : 0.f;                   // (padding columns: coef 0 -> zero planes)
        bp_tab[3 * kp + c] = ok ? a.shift[c] : 0.f;
        bp_tab[4 * kp + c] = (ok && a.train == 1) ? (float)(a.sums[c] / (double)a.m_rows) : 0.f;
        bp_tab[5 * kp + c] = (ok && a.train == 1) ? (float)(a.sums[a.n_cols + c] / (double)a.m_rows) : 0.f;
        if (blockIdx.x == 0 && ok && a.gbias && a.train == 2) a.gbias[c] += (double)a.coef[c] * a.sums[c];
    }
    __syncthreads();
    const int q8 = threadIdx.x & 7;
    const int nch = a.n_cols >> 2, nchp = kp >> 2;
    const bool even = (q8 & 1) == 0;
    const int act = a.act;
    const bool train = a.train == 1;
    for (int64_t tile = blockIdx.x; tile < a.m_pad / 32; tile += gridDim.x) {
        const int64_t row = tile * 32 + (threadIdx.x >> 3);
        const bool on = row < a.m_rows;
        const int64_t rr = on ? row : a.m_rows - 1;
        const float4 *gp = reinterpret_cast<const float4 *>(a.gy + rr * a.n_cols), *hp = reinterpret_cast<const float4 *>(a.h + rr * a.n_cols);
        float4 v[BP_NCH];
        unsigned m = 0;
#pragma unroll
        for (int i = 0; i < BP_NCH; ++i) {
            const int c = q8 + 8 * i;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < nch && on) {
                const float4 g = gp[c], hh = hp[c];
                const float4 mu = *reinterpret_cast<const float4 *>(bp_tab + 4 * c), is = *reinterpret_cast<const float4 *>(bp_tab + kp + 4 * c);
                const float4 cf = *reinterpret_cast<const float4 *>(bp_tab + 2 * kp + 4 * c), zb = *reinterpret_cast<const float4 *>(bp_tab + 3 * kp + 4 * c);
                const float4 m1 = *reinterpret_cast<const float4 *>(bp_tab + 4 * kp + 4 * c), m2 = *reinterpret_cast<const float4 *>(bp_tab + 5 * kp + 4 * c);
                auto one = [&](float gyv, float hv, float mu_, float is_, float cf_, float zb_, float m1_, float m2_) {
                    const float gz = gyv * act_grad_from_z((hv - mu_) * cf_ + zb_, act);
                    float g_ = gz;
                    if (train) g_ = gz - m1_ - (hv - mu_) * is_ * m2_;
                    return g_ * cf_;
                };
                v[i] = make_float4(one(g.x, hh.x, mu.x, is.x, cf.x, zb.x, m1.x, m2.x), one(g.y, hh.y, mu.y, is.y, cf.y, zb.y, m1.y, m2.y),
                                   one(g.z, hh.z, mu.z, is.z, cf.z, zb.z, m1.z, m2.z), one(g.w, hh.w, mu.w, is.w, cf.w, zb.w, m1.w, m2.w));
                m = max(max(m, __float_as_uint(v[i].x) & 0x7fffffffu), __float_as_uint(v[i].y) & 0x7fffffffu);
                m = max(max(m, __float_as_uint(v[i].z) & 0x7fffffffu), __float_as_uint(v[i].w) & 0x7fffffffu);
            }
        }
        bp_write_row(v, m, q8, even, row, on, a.rowinv, a.planes, kp, nchp);
    }
}

// The forward twin: Y = act((H - mean) * scale + shift) (gsn_bn_act_hip's expression) written as the row scratch of Y -- for a BatchNorm stage
// whose output only feeds the next product on the fp16x3 kernel (and, in the backward pass, the plane weight gradient of that product): no
// fp32 Y unless `out` is given, no row pre-pass over it.
struct FwdPlanesArgs {
    int64_t m_rows, m_pad;
    int n_cols, k_pad, act;
    const float *h, *mean, *scale, *shift;
    float *out;                     // fp32 rows too, or null
    float *rowinv;
    unsigned char *planes;
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BP_WAVES_FWD, BP_WAVES_FWD))) void bn_act_planes_kernel(FwdPlanesArgs a) {
    extern __shared__ __attribute__((aligned(16))) float bp_tab[];    // [3][k_pad]: mean | scale | shift
    const int kp = a.k_pad;
    for (int c = threadIdx.x; c < kp; c += 256) {
        const bool ok = c < a.n_cols;
        bp_tab[c] = (ok && a.mean) ? a.mean[c] : 0.f;
        bp_tab[kp + c] = ok ? (a.scale ? a.scale[c] : 1.f) : 0.f;
        bp_tab[2 * kp + c] = (ok && a.shift) ? a.shift[c] : 0.f;
    }
    __syncthreads();
    const int q8 = threadIdx.x & 7;
    const int nch = a.n_cols >> 2, nchp = kp >> 2;
    const bool even = (q8 & 1) == 0;
    const int act = a.act;
    auto fin = [&](float v, float mf, float sc, float sh) {
        float y = (v - mf) * sc + sh;
        switch (act) {
            case 1: y = y > 0.f ? y : 0.f; break;
            case 2: y = y > 0.f ? y : expm1f(y); break;
            case 3: y = tanhf(y); break;
            default: break;
        }
        return y;
    };
    for (int64_t tile = blockIdx.x; tile < a.m_pad / 32; tile += gridDim.x) {
        const int64_t row = tile * 32 + (threadIdx.x >> 3);
        const bool on = row < a.m_rows;
        const int64_t rr = on ? row : a.m_rows - 1;
        const float4 *hp = reinterpret_cast<const float4 *>(a.h + rr * a.n_cols);
        float4 *op = a.out ? reinterpret_cast<float4 *>(a.out + rr * a.n_cols) : nullptr;
        float4 v[BP_NCH];
        unsigned m = 0;
#pragma unroll
        for (int i = 0; i < BP_NCH; ++i) {
            const int c = q8 + 8 * i;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < nch && on) {
                const float4 hh = hp[c];
                const float4 mu = *reinterpret_cast<const float4 *>(bp_tab + 4 * c), sc = *reinterpret_cast<const float4 *>(bp_tab + kp + 4 * c);
                const float4 sh = *reinterpret_cast<const float4 *>(bp_tab + 2 * kp + 4 * c);
                v[i] = make_float4(fin(hh.x, mu.x, sc.x, sh.x), fin(hh.y, mu.y, sc.y, sh.y), fin(hh.z, mu.z, sc.z, sh.z), fin(hh.w, mu.w, sc.w, sh.w));
                if (op) op[c] = v[i];
                m = max(max(m, __float_as_uint(v[i].x) & 0x7fffffffu), __float_as_uint(v[i].y) & 0x7fffffffu);
                m = max(max(m, __float_as_uint(v[i].z) & 0x7fffffffu), __float_as_uint(v[i].w) & 0x7fffffffu);
            }
        }
        bp_write_row(v, m, q8, even, row, on, a.rowinv, a.planes, kp, nchp);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------------------------------------
constexpr int WG_T = 128;          // gW tile edge
constexpr int WG_RB = 16;          // rows per LDS chunk (2 x 2 x 16 x 132 floats = 33 KB of static LDS)
constexpr int WG_PITCH = WG_T + 4; // LDS row pitch: the two row halves of a fragment read land 4 banks apart
constexpr int WG_MAXB = 5;

struct WgradArgs {
    int64_t m_rows, rows_per_wg;
    int n_out, k_total, n_blocks;
    int tn, tk;                    // tiles along n_out / K (wgrad_bf16_kernel's own workgroup map)
    const float *gh;
    const float *bdata[WG_MAXB];
    int bwidth[WG_MAXB];
    const int64_t *bidx[WG_MAXB];  // gathered blocks (wgrad_bf16_kernel): row m of block b = bdata[b][idx[m]] -- the x_i / x_j blocks of an edge stage
    const int32_t *bidx32[WG_MAXB];
    float *gw;
};

typedef float f32x16b __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
    __shared__ float ta[2][WG_RB][WG_PITCH];      // gH rows x 128 output channels of this tile
    __shared__ float tb[2][WG_RB][WG_PITCH];      // X rows  x 128 input columns of this tile
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;            // the wave's 64 x 64 quadrant of the tile
    const int li = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.y * WG_T, k0 = blockIdx.z * WG_T;
    const int64_t r_begin = (int64_t)blockIdx.x * a.rows_per_wg;
    int64_t r_end = r_begin + a.rows_per_wg;
    if (r_end > a.m_rows) r_end = a.m_rows;

    // staging map: thread -> column (tid & 127) of the tile, rows (tid >> 7) + 2 i  (16 rows per thread per chunk)
    const int sc = tid & 127, sr0 = tid >> 7;
    const int ca = n0 + sc;
    const bool ca_ok = ca < a.n_out;
    const int kb = k0 + sc;
    const bool kb_ok = kb < a.k_total;
    const float *xb = a.bdata[0];
    int xw = a.bwidth[0];
    {
        int blk = 0, col = kb_ok ? kb : 0;
#pragma unroll
        for (int b = 0; b < WG_MAXB - 1; ++b)
            if (b < a.n_blocks - 1 && blk == b && col >= a.bwidth[b]) { col -= a.bwidth[b]; blk = b + 1; }
#pragma unroll
        for (int b = 1; b < WG_MAXB; ++b)
            if (blk == b) { xb = a.bdata[b]; xw = a.bwidth[b]; }
        xb += col;
    }
    const float *ga = a.gh + (ca_ok ? ca : 0);

    f32x16b acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float pa[WG_RB / 2], pb[WG_RB / 2];
    auto fetch = [&](int64_t row0) {
#pragma unroll
        for (int i = 0; i < WG_RB / 2; ++i) {
            const int64_t r = row0 + sr0 + 2 * i;
            const int64_t rc = r < r_end ? r : r_begin;            // clamped: masked to zero when stored
            pa[i] = ga[rc * a.n_out];
            pb[i] = xb[rc * xw];
        }
    };
    auto store = [&](int buf, int64_t row0) {
#pragma unroll
        for (int i = 0; i < WG_RB / 2; ++i) {
            const bool ok = row0 + sr0 + 2 * i < r_end;
            ta[buf][sr0 + 2 * i][sc] = (ok && ca_ok) ? pa[i] : 0.f;
            tb[buf][sr0 + 2 * i][sc] = (ok && kb_ok) ? pb[i] : 0.f;
        }
    };
    if (r_begin >= r_end) return;                 // (block-uniform)
    fetch(r_begin);
    store(0, r_begin);
    __syncthreads();
    int buf = 0;
    for (int64_t row0 = r_begin; row0 < r_end; row0 += WG_RB) {
        const bool has_next = row0 + WG_RB < r_end;
        if (has_next) fetch(row0 + WG_RB);
        const float *pa0 = &ta[buf][lh][wm * 64 + li];
        const float *pb0 = &tb[buf][lh][wn * 64 + li];
#pragma unroll
        for (int p = 0; p < WG_RB / 2; ++p) {
            const float a0 = pa0[2 * p * WG_PITCH], a1 = pa0[2 * p * WG_PITCH + 32];
            const float b0 = pb0[2 * p * WG_PITCH], b1 = pb0[2 * p * WG_PITCH + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (has_next) store(buf ^ 1, row0 + WG_RB);
        __syncthreads();
        buf ^= 1;
    }
    // C layout of a 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5); rows = output channels
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kcol = k0 + wn * 64 + j * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nrow = n0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (nrow < a.n_out && kcol < a.k_total) atomicAdd(a.gw + (int64_t)nrow * a.k_total + kcol, acc[i][j][r]);
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradient on the bf16 matrix pipe (r03): the same tile / slab / atomics skeleton, the products as six exact bf16 plane
// products on v_mfma_f32_32x32x16_bf16 (both operands split by truncation into 8 + 8 + 8 mantissa bits; hh, hm, mh, mm, hl, lh --
// the arithmetic of the forward bf16x6 kernels, chain_seg_bf16.hip: error of an fp32 FMA loop, any magnitude, no scales).  The
// contraction runs over ROWS, so a lane's operand (8 consecutive k of one column) is a column segment of the row-major inputs:
// the staging threads own one column and 8 rows, split while the values are in registers and write one 16-byte slot per plane
// into a [column][row-half] LDS layout the MFMA lanes read back with one ds_read_b128 -- the transposition costs nothing.
// 6 MFMAs of 8 passes per 16 rows and tile pair instead of 8 fp32 MFMAs of 16 passes: 2.7x less matrix time; LDS 48 KiB.
// ---------------------------------------------------------------------------------------------------------------------
typedef unsigned wg_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void wg_split3(float x, unsigned &h, unsigned &m, unsigned &l) {
    h = __float_as_uint(x);
    const float r1 = x - __uint_as_float(h & 0xffff0000u);
    m = __float_as_uint(r1);
    l = __float_as_uint(r1 - __uint_as_float(m & 0xffff0000u));
}
__device__ __forceinline__ unsigned wg_pack_hi(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// GATHER: some block of X is read through a row-index array (compiled out otherwise: the index selects and the dependent loads in the
// staging loop cost the plain-row stages of the d = 300 model 2x -- 347 -> 731 us per call at 214 k rows, r04 A/B)
template <bool GATHER>
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(WgradArgs a) {
    __shared__ wg_u32x4 ta[2][3][WG_T][2];        // [buffer][plane][column][row half]: 8 bf16 = rows 8 h .. 8 h + 7 of the chunk
    __shared__ wg_u32x4 tb[2][3][WG_T][2];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int li = lane & 31, lh = lane >> 5;
    // Workgroup -> (row slab, tile): the tn x tk tiles of ONE slab are consecutive workgroups of ONE XCD (the dispatcher deals
    // workgroup ids round-robin over the 8 XCDs), so the slab's rows of gH and X -- read once per tile column / row -- come from
    // that XCD's L2 after the first read instead of from HBM (tn = 5, tk = 3: 1.39 GB of reads -> 0.38 GB per d = 300 stage).
    const int ntile = a.tn * a.tk;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int tile = seq % ntile;
    const int64_t slab = (int64_t)(seq / ntile) * 8 + xcd;
    const int n0 = (tile / a.tk) * WG_T, k0 = (tile % a.tk) * WG_T;
    const int64_t r_begin = slab * a.rows_per_wg;
    int64_t r_end = r_begin + a.rows_per_wg;
    if (r_end > a.m_rows) r_end = a.m_rows;
    if (r_begin >= r_end) return;                 // (block-uniform)

    const int sc = tid & 127, sh = tid >> 7;      // staging: column sc, rows 8 sh .. 8 sh + 7 of a 16-row chunk
    const int ca = n0 + sc;
    const bool ca_ok = ca < a.n_out;
    const int kb = k0 + sc;
    const bool kb_ok = kb < a.k_total;
    const float *xb = a.bdata[0];
    int xw = a.bwidth[0];
    const int64_t *xi = GATHER ? a.bidx[0] : nullptr;
    const int32_t *xi32 = GATHER ? a.bidx32[0] : nullptr;
    {
        int blk = 0, col = kb_ok ? kb : 0;
#pragma unroll
        for (int b = 0; b < WG_MAXB - 1; ++b)
            if (b < a.n_blocks - 1 && blk == b && col >= a.bwidth[b]) { col -= a.bwidth[b]; blk = b + 1; }
#pragma unroll
        for (int b = 1; b < WG_MAXB; ++b)
            if (blk == b) { xb = a.bdata[b]; xw = a.bwidth[b]; if (GATHER) { xi = a.bidx[b]; xi32 = a.bidx32[b]; } }
        xb += col;
    }
    const float *ga = a.gh + (ca_ok ? ca : 0);

    f32x16b acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float pa[8], pb[8];
    // GATHER: the row indices of a chunk are loaded one chunk ahead of its rows (nidx: those of the chunk the next fetch reads), so that a fetch's
    // row loads do not wait for an index load in front of them -- two serial latencies per chunk otherwise, with one chunk in flight
    int nidx[8];                                                   // (row indices fit 32 bits: the blocks' rows are counted in int32 everywhere)
    auto idx_load = [&](int64_t row0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t r = row0 + 8 * sh + i;
            const int64_t rc = r < r_end ? r : r_begin;
            nidx[i] = xi ? (int)xi[rc] : (xi32 ? xi32[rc] : (int)rc);
        }
    };
    if (GATHER) idx_load(r_begin);
    auto fetch = [&](int64_t row0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t r = row0 + 8 * sh + i;
            const int64_t rc = r < r_end ? r : r_begin;            // clamped: masked to zero when stored
            pa[i] = ga[rc * a.n_out];
            // (a gathered block: the row index is the same for the 128 threads of a row half -- one broadcast load)
            pb[i] = xb[(GATHER ? (int64_t)nidx[i] : rc) * xw];
        }
        if (GATHER) idx_load(row0 + 16);                           // (past the slab's end: clamped to r_begin, never used)
    };
    auto store_one = [&](wg_u32x4 (*dst)[WG_T][2], const float (&v)[8], bool col_ok, int64_t row0) {
        unsigned h[8], m[8], l[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool ok = col_ok && (row0 + 8 * sh + i < r_end);
            wg_split3(ok ? v[i] : 0.f, h[i], m[i], l[i]);
        }
        wg_u32x4 ph, pm, pl;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ph[i] = wg_pack_hi(h[2 * i], h[2 * i + 1]);
            pm[i] = wg_pack_hi(m[2 * i], m[2 * i + 1]);
            pl[i] = wg_pack_hi(l[2 * i], l[2 * i + 1]);
        }
        dst[0][sc][sh] = ph;
        dst[1][sc][sh] = pm;
        dst[2][sc][sh] = pl;
    };
    fetch(r_begin);
    store_one(ta[0], pa, ca_ok, r_begin);
    store_one(tb[0], pb, kb_ok, r_begin);
    __syncthreads();
    int buf = 0;
#define WG_MF(x, y, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wg_bf16x8, x), __builtin_bit_cast(wg_bf16x8, y), c, 0, 0, 0)
    for (int64_t row0 = r_begin; row0 < r_end; row0 += 16) {
        const bool has_next = row0 + 16 < r_end;
        if (has_next) fetch(row0 + 16);
        wg_u32x4 fa[2][3], fb[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                fa[i][pl] = ta[buf][pl][wm * 64 + i * 32 + li][lh];
                fb[i][pl] = tb[buf][pl][wn * 64 + i * 32 + li][lh];
            }
        // small terms first (hl, lh, mm, hm, mh), hh last; the four accumulator tiles alternate so that no MFMA waits for its predecessor
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int pa_ = (t == 0) ? 0 : (t == 1) ? 2 : (t == 2) ? 1 : (t == 3) ? 0 : (t == 4) ? 1 : 0;
            const int pb_ = (t == 0) ? 2 : (t == 1) ? 0 : (t == 2) ? 1 : (t == 3) ? 1 : (t == 4) ? 0 : 0;
            WG_MF(fa[0][pa_], fb[0][pb_], acc[0][0]);
            WG_MF(fa[0][pa_], fb[1][pb_], acc[0][1]);
            WG_MF(fa[1][pa_], fb[0][pb_], acc[1][0]);
            WG_MF(fa[1][pa_], fb[1][pb_], acc[1][1]);
        }
        if (has_next) {
            store_one(ta[buf ^ 1], pa, ca_ok, row0 + 16);
            store_one(tb[buf ^ 1], pb, kb_ok, row0 + 16);
        }
        __syncthreads();
        buf ^= 1;
    }
#undef WG_MF
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kcol = k0 + wn * 64 + j * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nrow = n0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (nrow < a.n_out && kcol < a.k_total) atomicAdd(a.gw + (int64_t)nrow * a.k_total + kcol, acc[i][j][r]);
            }
        }
}

// The same kernel for plain rows as a software pipeline INSIDE a wave (r05).  Above, a 16-row step is 24 MFMAs back to back and then ~130 vector
// instructions that split the next step's values into planes: a SIMD does not overlap one wave's matrix stream with another wave's vector
// stream (profiles/r03_issue_mix.txt: the sum, not the maximum), but it does issue ~4 independent vector instructions in the shadow of each of a
// wave's OWN 32x32x16 MFMAs.  Here the rows of step s + 2 are requested at the top of step s, and the values of step s + 1 (requested one step
// earlier: a load's latency is longer than a step's matrix phase) are split between the MFMAs of step s -- one MFMA, five vector instructions,
// by sched_group_barrier -- and written to the other LDS buffer behind them.  Branch-free loop body (rows past the slab's end are clamped
// loads and zero planes), two named register sets, the loop unrolled by two.  Same planes, same products, same order: the same bits.
// SINGLE: X is one block -- every row address is a scalar base (the row) plus a 32-bit lane offset (the column): no vector arithmetic per load
// GATHER (r06): blocks read through a row index (the x_i / x_j blocks of an edge stage, GSN_edge_sparse.py:160-165): the row of a load is the same
// for all lanes of a block, so a gathered block costs one index load per row, requested a step ahead of the rows it addresses (its own register
// set, alternating like the rows') -- wgrad_bf16_kernel<true> keeps one chunk in flight and splits behind its products: 235 -> 194 us at 194 284 x 128 x 272 (scripts/gpu/r6_wgrad_gather.py;
// the same rows assembled first: 153; config-2 training step at 4 096 graphs 5.77 -> 5.59 ms)
template <bool SINGLE, int WG_PIPE_VALU, bool GATHER = false>
__global__ __launch_bounds__(256) void wgrad_bf16_pipe_kernel(WgradArgs a) {
    __shared__ wg_u32x4 ta[2][3][WG_T][2];
    __shared__ wg_u32x4 tb[2][3][WG_T][2];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int ntile = a.tn * a.tk;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int tile = seq % ntile;
    const int64_t slab = (int64_t)(seq / ntile) * 8 + xcd;
    const int n0 = (tile / a.tk) * WG_T, k0 = (tile % a.tk) * WG_T;
    const int64_t r_begin = slab * a.rows_per_wg;
    int64_t r_end = r_begin + a.rows_per_wg;
    if (r_end > a.m_rows) r_end = a.m_rows;
    if (r_begin >= r_end) return;                 // (block-uniform)

    const int sc = tid & 127, sh = __builtin_amdgcn_readfirstlane(tid >> 7);      // (the row half is the same for a whole wave)
    const int ca = n0 + sc;
    const bool ca_ok = ca < a.n_out;
    const int kb = k0 + sc;
    const bool kb_ok = kb < a.k_total;
    const float *xb = a.bdata[0];
    int xw = a.bwidth[0];
    const int64_t *xi = GATHER ? a.bidx[0] : nullptr;
    const int32_t *xi32 = GATHER ? a.bidx32[0] : nullptr;
    {
        int blk = 0, col = kb_ok ? kb : 0;
#pragma unroll
        for (int b = 0; b < WG_MAXB - 1; ++b)
            if (b < a.n_blocks - 1 && blk == b && col >= a.bwidth[b]) { col -= a.bwidth[b]; blk = b + 1; }
#pragma unroll
        for (int b = 1; b < WG_MAXB; ++b)
            if (blk == b) { xb = a.bdata[b]; xw = a.bwidth[b]; if (GATHER) { xi = a.bidx[b]; xi32 = a.bidx32[b]; } }
        xb += col;
    }
    auto idx_load = [&](int (&ix)[8], int64_t row0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t r = row0 + 8 * sh + i;
            const int64_t rc = r < r_end ? r : r_begin;
            ix[i] = xi ? (int)xi[rc] : (xi32 ? xi32[rc] : (int)rc);
        }
    };
    const float *ga = a.gh + (ca_ok ? ca : 0);
    const unsigned ca_u = ca_ok ? (unsigned)ca : 0u, kb_u = kb_ok ? (unsigned)kb : 0u;

    f32x16b acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto fetch = [&](float (&pa)[8], float (&pb)[8], int64_t row0, const int (&ix)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t r = row0 + 8 * sh + i;
            const int64_t rc = r < r_end ? r : r_begin;            // clamped: masked to zero when split  (scalar: row0, sh and i are)
            const float *grow = a.gh + rc * a.n_out;
            pa[i] = grow[ca_u];
            if (SINGLE) {
                const float *xrow = a.bdata[0] + rc * a.bwidth[0];
                pb[i] = xrow[kb_u];
            } else {
                pb[i] = xb[(GATHER ? (int64_t)ix[i] : rc) * xw];
            }
        }
    };
    auto split = [&](const float (&v)[8], bool col_ok, int64_t row0, wg_u32x4 &ph, wg_u32x4 &pm, wg_u32x4 &pl) {
        unsigned h[8], m[8], l[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool ok = col_ok && (row0 + 8 * sh + i < r_end);
            wg_split3(ok ? v[i] : 0.f, h[i], m[i], l[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ph[i] = wg_pack_hi(h[2 * i], h[2 * i + 1]);
            pm[i] = wg_pack_hi(m[2 * i], m[2 * i + 1]);
            pl[i] = wg_pack_hi(l[2 * i], l[2 * i + 1]);
        }
    };
    float pa0[8], pb0[8], pa1[8], pb1[8];
    int ix0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ix1[8] = {0, 0, 0, 0, 0, 0, 0, 0};    // (GATHER: row indices of the rows the NEXT fetch reads / the one after)
    {
        if (GATHER) { idx_load(ix0, r_begin); idx_load(ix1, r_begin + 16); }
        fetch(pa0, pb0, r_begin, ix0);
        fetch(pa1, pb1, r_begin + 16, ix1);
        if (GATHER) idx_load(ix0, r_begin + 32);
        wg_u32x4 gh, gm, gl, xh, xm, xl;
        split(pa0, ca_ok, r_begin, gh, gm, gl);
        split(pb0, kb_ok, r_begin, xh, xm, xl);
        ta[0][0][sc][sh] = gh; ta[0][1][sc][sh] = gm; ta[0][2][sc][sh] = gl;
        tb[0][0][sc][sh] = xh; tb[0][1][sc][sh] = xm; tb[0][2][sc][sh] = xl;
    }
    __syncthreads();
#define WG_MF(x, y, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wg_bf16x8, x), __builtin_bit_cast(wg_bf16x8, y), c, 0, 0, 0)
    // one step: products of the rows staged in `buf`; `nxt` (rows row0 + 16, in registers since the step before) split under them into buf ^ 1;
    // `far` receives the rows row0 + 32
    auto step = [&](int buf, int64_t row0, const float (&npa)[8], const float (&npb)[8], float (&fpa)[8], float (&fpb)[8], const int (&fix)[8], int (&nix)[8]) {
        wg_u32x4 fa[2][3], fb[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                fa[i][pl] = ta[buf][pl][wm * 64 + i * 32 + li][lh];
                fb[i][pl] = tb[buf][pl][wn * 64 + i * 32 + li][lh];
            }
        wg_u32x4 gh, gm, gl, xh, xm, xl;
        split(npa, ca_ok, row0 + 16, gh, gm, gl);
        split(npb, kb_ok, row0 + 16, xh, xm, xl);
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int pa_ = (t == 0) ? 0 : (t == 1) ? 2 : (t == 2) ? 1 : (t == 3) ? 0 : (t == 4) ? 1 : 0;
            const int pb_ = (t == 0) ? 2 : (t == 1) ? 0 : (t == 2) ? 1 : (t == 3) ? 1 : (t == 4) ? 0 : 0;
            WG_MF(fa[0][pa_], fb[0][pb_], acc[0][0]);
            WG_MF(fa[0][pa_], fb[1][pb_], acc[0][1]);
            WG_MF(fa[1][pa_], fb[0][pb_], acc[1][0]);
            WG_MF(fa[1][pa_], fb[1][pb_], acc[1][1]);
        }
        ta[buf ^ 1][0][sc][sh] = gh; ta[buf ^ 1][1][sc][sh] = gm; ta[buf ^ 1][2][sc][sh] = gl;
        tb[buf ^ 1][0][sc][sh] = xh; tb[buf ^ 1][1][sc][sh] = xm; tb[buf ^ 1][2][sc][sh] = xl;
        fetch(fpa, fpb, row0 + 32, fix);
        if (GATHER) idx_load(nix, row0 + 48);       // (the indices of the rows the next step requests)
        // the order asked of the scheduler: the fragment reads, then the far rows' requests, then one MFMA : five vector instructions, the plane
        // writes last
#pragma unroll
        for (int t = 0; t < 24; ++t) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, WG_PIPE_VALU, 0);  // VALU
        }
        __syncthreads();
    };
    int64_t row0 = r_begin;
    for (; row0 < r_end; row0 += 32) {
        step(0, row0, pa1, pb1, pa0, pb0, ix0, ix1);
        if (row0 + 16 >= r_end) break;            // (block-uniform)
        step(1, row0 + 16, pa0, pb0, pa1, pb1, ix1, ix0);
    }
#undef WG_MF
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kcol = k0 + wn * 64 + j * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nrow = n0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (nrow < a.n_out && kcol < a.k_total) atomicAdd(a.gw + (int64_t)nrow * a.k_total + kcol, acc[i][j][r]);
            }
        }
}

// Workgroups along the rows: every workgroup ends with one fp64 atomic per column onto the SAME n_cols addresses, which the L2
// serialises (~20 ns each: 1024 workgroups = 20 us whatever the row count -- the whole run time at the reference's batch sizes,
// E ~ 6 000 rows), so a workgroup takes at least 64 rows (16 per wave).
static int bgrid(int64_t m_rows) {
    int64_t b = (m_rows + 63) / 64;
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

}  // namespace gsn

using namespace gsn;

static int bn_act_bwd_launch(int64_t m_rows, int64_t n_cols, const float *grad_y, const float *y, const float *h, const float *mean,
                             const float *invstd, const float *coef, const float *shift, int train_bn, int act, double *sums, float *grad_h,
                             double *grad_bias, void *stream, const char *who) {
    if (n_cols < 1 || act < 0 || act > 3 || (m_rows > 0 && (!grad_y || !grad_h)) || (train_bn && m_rows > 0 && (!h || !sums)) ||
        (m_rows > 0 && !y && !(train_bn && mean && coef)))
        return set_error(GSN_E_INVALID, "%s: bad arguments", who);
    if (m_rows <= 0) return GSN_OK;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid(bgrid(m_rows), (unsigned)((n_cols + 63) / 64));
    // train_bn == 2: BatchNorm on its RUNNING statistics (eval mode) with gradients wanted for gamma / beta: the same column sums
    // (xhat from the running mean / invstd), but grad_h = gZ * coef -- the statistics are constants of the rows
    if (train_bn)
        hipLaunchKernelGGL(bn_act_bwd_reduce_kernel, grid, dim3(256), 0, s, m_rows, (int)n_cols, grad_y, y, h, mean, invstd, act, sums, coef, shift);
    // (y == null: both train_bn modes read the pre-BatchNorm rows for the activation derivative)
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel, grid, dim3(256), 0, s, m_rows, (int)n_cols, grad_y, y, (train_bn == 1 || !y) ? h : y, mean,
                       invstd, coef, train_bn == 1 ? sums : nullptr, act, grad_h, grad_bias, shift);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "bn_act_bwd kernels: %s", hipGetErrorString(e));
    return GSN_OK;
}

extern "C" int gsn_bn_act_bwd_hip(int64_t m_rows, int64_t n_cols, const float *grad_y, const float *y, const float *h,
                                  const float *mean, const float *invstd, const float *coef, int train_bn, int act,
                                  double *sums, float *grad_h, double *grad_bias, void *stream) {
    if (m_rows > 0 && !y) return set_error(GSN_E_INVALID, "gsn_bn_act_bwd_hip: bad arguments");
    return bn_act_bwd_launch(m_rows, n_cols, grad_y, y, h, mean, invstd, coef, nullptr, train_bn, act, sums, grad_h, grad_bias, stream, "gsn_bn_act_bwd_hip");
}

extern "C" int gsn_bn_act_bwd_from_h_hip(int64_t m_rows, int64_t n_cols, const float *grad_y, const float *h, const float *mean,
                                         const float *invstd, const float *coef, const float *shift, int train_bn, int act, double *sums,
                                         float *grad_h, double *grad_bias, void *stream) {
    if (train_bn != 1 && train_bn != 2) return set_error(GSN_E_INVALID, "gsn_bn_act_bwd_from_h_hip: a BatchNorm stage (train_bn 1 or 2)");
    return bn_act_bwd_launch(m_rows, n_cols, grad_y, nullptr, h, mean, invstd, coef, shift, train_bn, act, sums, grad_h, grad_bias, stream,
                             "gsn_bn_act_bwd_from_h_hip");
}

// gsn_bn_act_bwd_from_h_hip for a stage whose gH is read as fp16 planes only: reduce pass as there, then bn_act_bwd_planes_kernel writes the
// row scratch of gH (gsn_linear_f16x3_scratch_bytes(m_rows, n_cols) bytes: what gsn_linear_f16x3_fwd_presplit_hip and gsn_wgrad_f16x3_hip read)
extern "C" int gsn_bn_act_bwd_planes_hip(int64_t m_rows, int64_t n_cols, const float *grad_y, const float *h, const float *mean, const float *invstd,
                                         const float *coef, const float *shift, int train_bn, int act, double *sums, float *row_scratch,
                                         double *grad_bias, void *stream) {
    if ((train_bn != 1 && train_bn != 2) || n_cols < 4 || (n_cols & 3) || n_cols > 8 * BP_NCH * 4 || act < 0 || act > 3)
        return set_error(GSN_E_UNSUPPORTED, "gsn_bn_act_bwd_planes_hip: a BatchNorm stage (train_bn 1 or 2) of 4 .. %d columns, a multiple of 4", 8 * BP_NCH * 4);
    if (m_rows > 0 && (!grad_y || !h || !mean || !invstd || !coef || !shift || !sums || !row_scratch))
        return set_error(GSN_E_INVALID, "gsn_bn_act_bwd_planes_hip: bad arguments");
    if (m_rows <= 0) return GSN_OK;
    if ((reinterpret_cast<uintptr_t>(grad_y) | reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(row_scratch)) & 15)
        return set_error(GSN_E_INVALID, "gsn_bn_act_bwd_planes_hip: grad_y, h and row_scratch must be 16-byte aligned");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid(bgrid(m_rows), (unsigned)((n_cols + 63) / 64));
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel, grid, dim3(256), 0, s, m_rows, (int)n_cols, grad_y, (const float *)nullptr, h, mean, invstd, act, sums, coef, shift);
    BwdPlanesArgs a{};
    a.m_rows = m_rows; a.m_pad = gsn_linear_f16x3_mpad(m_rows); a.n_cols = (int)n_cols; a.k_pad = (int)gsn_linear_f16x3_kpad(n_cols);
    a.act = act; a.train = train_bn;
    a.gy = grad_y; a.h = h; a.mean = mean; a.invstd = invstd; a.coef = coef; a.shift = shift; a.sums = sums;
    a.rowinv = row_scratch; a.planes = reinterpret_cast<unsigned char *>(row_scratch + a.m_pad); a.gbias = grad_bias;
    int64_t gx = a.m_pad / 32;
    static const int64_t wgs = [] { const char *e = getenv("GSN_BWD_PLANES_WGS"); const int64_t v = e ? atoll(e) : 0; return v > 0 ? v : (int64_t)2048; }();
    if (gx > wgs) gx = wgs;
    hipLaunchKernelGGL(bn_act_bwd_planes_kernel, dim3((unsigned)gx), dim3(256), (size_t)6 * a.k_pad * 4, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "bn_act_bwd_planes kernels: %s", hipGetErrorString(e));
    return GSN_OK;
}

// gsn_bn_act_hip for a stage output that is read as fp16 planes: row_scratch = the row scratch of Y (gsn_linear_f16x3_scratch_bytes(m_rows, n_cols)),
// out = the fp32 rows as well, or null
extern "C" int gsn_bn_act_planes_hip(int64_t m_rows, int64_t n_cols, const float *h, const float *mean, const float *scale, const float *shift, int act,
                                     float *out, float *row_scratch, void *stream) {
    if (n_cols < 4 || (n_cols & 3) || n_cols > 8 * BP_NCH * 4 || act < 0 || act > 3)
        return set_error(GSN_E_UNSUPPORTED, "gsn_bn_act_planes_hip: 4 .. %d columns, a multiple of 4", 8 * BP_NCH * 4);
    if (m_rows > 0 && (!h || !row_scratch)) return set_error(GSN_E_INVALID, "gsn_bn_act_planes_hip: bad arguments");
    if (m_rows <= 0) return GSN_OK;
    if ((reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(row_scratch)) & 15)
        return set_error(GSN_E_INVALID, "gsn_bn_act_planes_hip: h, out and row_scratch must be 16-byte aligned");
    FwdPlanesArgs a{};
    a.m_rows = m_rows; a.m_pad = gsn_linear_f16x3_mpad(m_rows); a.n_cols = (int)n_cols; a.k_pad = (int)gsn_linear_f16x3_kpad(n_cols); a.act = act;
    a.h = h; a.mean = mean; a.scale = scale; a.shift = shift; a.out = out;
    a.rowinv = row_scratch; a.planes = reinterpret_cast<unsigned char *>(row_scratch + a.m_pad);
    int64_t gx = a.m_pad / 32;
    static const int64_t wgs = [] { const char *e = getenv("GSN_FWD_PLANES_WGS"); const int64_t v = e ? atoll(e) : 0; return v > 0 ? v : (int64_t)2048; }();
    if (gx > wgs) gx = wgs;
    hipLaunchKernelGGL(bn_act_planes_kernel, dim3((unsigned)gx), dim3(256), (size_t)3 * a.k_pad * 4, reinterpret_cast<hipStream_t>(stream), a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "bn_act_planes_kernel: %s", hipGetErrorString(e));
    return GSN_OK;
}

// rows per slab of a weight-gradient call (both kernels): see gsn_wgrad_hip.  wg_target > 0: the r05 rule with that many workgroups asked for
namespace gsn {
int64_t gsn_wgrad_slab_rows(int64_t m_rows, int64_t tiles, int64_t wg_target) {
    if (wg_target > 0) {
        const int64_t slabs = (wg_target + tiles - 1) / tiles;
        int64_t rows_per = (m_rows + slabs - 1) / slabs;
        const int64_t min_rows = m_rows >= 65536 ? 256 : 64;
        return rows_per < min_rows ? min_rows : rows_per;
    }
    int64_t rows_per = (m_rows * tiles + 255) / 256;
    rows_per = (rows_per + 63) / 64 * 64;
    rows_per = rows_per < 64 ? 64 : (rows_per > 512 ? 512 : rows_per);
    if ((m_rows + rows_per - 1) / rows_per * tiles > 2048) {
        const int64_t slabs = (2048 + tiles - 1) / tiles;
        rows_per = (m_rows + slabs - 1) / slabs;
    }
    return rows_per;
}
}  // namespace gsn

extern "C" int gsn_wgrad_hip(int64_t m_rows, int64_t n_out, const float *grad_h, int n_blocks, const gsn_block *blocks,
                             float *grad_w, void *stream) {
    if (n_out < 1 || n_blocks < 1 || n_blocks > WG_MAXB || !blocks || !grad_w || (m_rows > 0 && !grad_h))
        return set_error(GSN_E_INVALID, "gsn_wgrad_hip: bad arguments");
    WgradArgs a{};
    a.m_rows = m_rows; a.n_out = (int)n_out; a.gh = grad_h; a.gw = grad_w; a.n_blocks = n_blocks;
    bool gathered = false;
    int k = 0;
    for (int b = 0; b < n_blocks; ++b) {
        if ((!blocks[b].data && m_rows > 0) || blocks[b].width <= 0) return set_error(GSN_E_INVALID, "gsn_wgrad_hip: block %d is empty", b);
        a.bdata[b] = blocks[b].data; a.bwidth[b] = (int)blocks[b].width;
        a.bidx[b] = blocks[b].idx; a.bidx32[b] = blocks[b].idx32;
        gathered = gathered || blocks[b].idx || blocks[b].idx32;
        k += (int)blocks[b].width;
    }
    a.k_total = k;
    if (m_rows <= 0) return GSN_OK;
    const int tn = (int)((n_out + WG_T - 1) / WG_T), tk = (k + WG_T - 1) / WG_T;
    // Slab length.  A workgroup = one 128 x 128 tile of gW over one slab of rows; it ends with 128 x 128 float atomics (~15 us when a few hundred
    // workgroups do it at once) and walks its slab in a serial chain of 16-row steps (~1.4 us each).  Small and mid-size calls (the reference's
    // batch sizes; the virtual node's stages): ONE round of ~256 workgroups -- slabs of ceil(M tiles / 256) rows, 64 .. 512.  r05 used 64-row
    // slabs up to 65 536 rows: 1 410 workgroups in three rounds at M = 6 000 x 300 x 600, 82 us; one round of 360: 41 us (M = 4 096: 61 -> 33 us,
    // profiles/r06_wgrad16_phase.txt).  Large calls: 2 048 workgroups for balance, their slabs are longer than 512 rows by themselves.
    // GSN_WGRAD_WGS=n: n workgroups per call asked for, as in r05 (A/B)
    static const int64_t wg_target = [] { const char *e = getenv("GSN_WGRAD_WGS"); const int64_t v = e ? atoll(e) : 0; return v > 0 ? v : (int64_t)0; }();
    int64_t rows_per = gsn_wgrad_slab_rows(m_rows, (int64_t)tn * tk, wg_target);
    rows_per = (rows_per + WG_RB - 1) / WG_RB * WG_RB;
    a.rows_per_wg = rows_per;
    int64_t slabs = (m_rows + rows_per - 1) / rows_per;
    if (getenv("GSN_CHAIN_TRACE"))
        fprintf(stderr, "gsn wgrad: M %lld N %d K %d blocks %d%s slabs %lld x %lld rows\n", (long long)m_rows, (int)n_out, k, n_blocks, gathered ? " gathered" : "",
                (long long)slabs, (long long)rows_per);
    // GSN_WGRAD_FP32=1: the fp32-MFMA kernel (v_mfma_f32_32x32x2_f32), for A/B runs
    static const bool fp32_kernel = [] { const char *e = getenv("GSN_WGRAD_FP32"); return e && e[0] == '1'; }();
    if (fp32_kernel && !gathered)      // (gathered blocks: the bf16 kernel only)
        hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)slabs, (unsigned)tn, (unsigned)tk), dim3(256), 0,
                           reinterpret_cast<hipStream_t>(stream), a);
    else {
        a.tn = tn; a.tk = tk;
        const int64_t slab_groups = (slabs + 7) / 8;
        // GSN_WGRAD_PIPE: 0 = the kernel above for plain rows too; 4 / 5 / 6 = vector instructions asked for behind each MFMA (default 6; config-4 step on one box: off 19.71, 4: 19.43, 5: 19.22, 6: 19.06-19.21 ms)
        static const int pipe = [] { const char *e = getenv("GSN_WGRAD_PIPE"); const int v = e ? atoi(e) : 6; return v == 1 ? 6 : v; }();
        const dim3 grid((unsigned)(slab_groups * tn * tk * 8));
        hipStream_t st = reinterpret_cast<hipStream_t>(stream);
        if (gathered && pipe) hipLaunchKernelGGL((wgrad_bf16_pipe_kernel<false, 6, true>), grid, dim3(256), 0, st, a);
        else if (gathered) hipLaunchKernelGGL(wgrad_bf16_kernel<true>, grid, dim3(256), 0, st, a);
        else if (pipe == 4 && n_blocks == 1) hipLaunchKernelGGL((wgrad_bf16_pipe_kernel<true, 4>), grid, dim3(256), 0, st, a);
        else if (pipe == 4) hipLaunchKernelGGL((wgrad_bf16_pipe_kernel<false, 4>), grid, dim3(256), 0, st, a);
        else if (pipe == 5 && n_blocks == 1) hipLaunchKernelGGL((wgrad_bf16_pipe_kernel<true, 5>), grid, dim3(256), 0, st, a);
        else if (pipe == 5) hipLaunchKernelGGL((wgrad_bf16_pipe_kernel<false, 5>), grid, dim3(256), 0, st, a);
        else if (pipe && n_blocks == 1) hipLaunchKernelGGL((wgrad_bf16_pipe_kernel<true, 6>), grid, dim3(256), 0, st, a);
        else if (pipe) hipLaunchKernelGGL((wgrad_bf16_pipe_kernel<false, 6>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(wgrad_bf16_kernel<false>, dim3((unsigned)(slab_groups * tn * tk * 8)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "wgrad_kernel: %s", hipGetErrorString(e));
    return GSN_OK;
}
