// The folded first weight of update_fn for the `general` layers with the aggregation pulled in front of msg_fn's last Linear
// (GSN_edge_sparse.py:153-170, GSN_sparse.py:166-171):
//     update_fn.fc[0] applied to cat(x, sum_e (W2 r_e + b2)) = cat(x, S, deg) [W3x | W3a W2 | W3a b2]^T,   S = sum_e r_e
// The three column groups of that weight and their adjoint, each as ONE launch (the training step rebuilds the fold at every step because
// W2, b2 and W3 move; as a composition of tensor ops it was ~18 launches per layer, which at the reference's batch sizes is the step).
// Plain fp32 FMA dot products of length A (forward) and H + 1 / R (adjoint): the matrices are a few hundred rows and columns.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "gsn_internal.h"

namespace gsn {

// out[r] = [ w3[r, :d_x] | w3a[r] W2 | w3a[r] . b2 | pad zeros ]      w3: [R, d_x + A] (ld3), W2: [A, H] (ld2), b2: [A], out: [R, d_x + H + 1 + pad]
__global__ __launch_bounds__(256) void fold_fwd_kernel(int d_x, int A, int H, int pad, const float *__restrict__ w3, int64_t ld3,
                                                       const float *__restrict__ w2, int64_t ld2, const float *__restrict__ b2, float *__restrict__ out) {
    extern __shared__ float row[];      // w3a[r]: A floats
    const int r = blockIdx.x;
    const float *w3r = w3 + (int64_t)r * ld3;
    float *o = out + (int64_t)r * (d_x + H + 1 + pad);
    for (int a = threadIdx.x; a < A; a += 256) row[a] = w3r[d_x + a];
    for (int c = threadIdx.x; c < d_x; c += 256) o[c] = w3r[c];
    if ((int)threadIdx.x < pad) o[d_x + H + 1 + threadIdx.x] = 0.f;
    __syncthreads();
    for (int h = threadIdx.x; h <= H; h += 256) {
        float s0 = 0.f, s1 = 0.f;
        if (h < H) {
            int a = 0;
            for (; a + 7 < A; a += 8) {             // eight loads in flight: the products are a chain of L2 round trips otherwise
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = w2[(int64_t)(a + j) * ld2 + h];
#pragma unroll
                for (int j = 0; j < 8; j += 2) { s0 = fmaf(row[a + j], v[j], s0); s1 = fmaf(row[a + j + 1], v[j + 1], s1); }
            }
            for (; a < A; ++a) s0 = fmaf(row[a], w2[(int64_t)a * ld2 + h], s0);
        } else {
            for (int a = 0; a < A; ++a) s0 = fmaf(row[a], b2[a], s0);
        }
        o[d_x + h] = s0 + s1;
    }
}

// g: [R, d_x + H + 1] (ldg).  Workgroups 0 .. R-1: g_w3[r] = [ g[r, :d_x] | g_fold[r] W2^T + g_b[r] b2 ];  workgroups R .. R+A-1 (a = block - R):
// g_w2[a, h] = sum_r w3a[r, a] g_fold[r, h],  g_b2[a] = sum_r w3a[r, a] g_b[r].
__global__ __launch_bounds__(256) void fold_bwd_kernel(int R, int d_x, int A, int H, const float *__restrict__ g, int64_t ldg, const float *__restrict__ w3,
                                                       int64_t ld3, const float *__restrict__ w2, int64_t ld2, const float *__restrict__ b2,
                                                       float *__restrict__ g_w3, float *__restrict__ g_w2, float *__restrict__ g_b2) {
    extern __shared__ float row[];      // workgroups < R: g[r, d_x : d_x + H + 1];  the others: column a of w3a (R floats)
    if ((int)blockIdx.x < R) {
        const int r = blockIdx.x;
        const float *gr = g + (int64_t)r * ldg;
        float *o = g_w3 + (int64_t)r * (d_x + A);
        for (int h = threadIdx.x; h <= H; h += 256) row[h] = gr[d_x + h];
        for (int c = threadIdx.x; c < d_x; c += 256) o[c] = gr[c];
        __syncthreads();
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const float gb = row[H];
        for (int a0 = 8 * wave; a0 < A; a0 += 32) {      // a wave takes eight rows of W2 at a time: the lanes walk them (coalesced, eight loads in flight), then sum across the wave
            float s[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] = 0.f;
            for (int h = lane; h < H; h += 64) {
                const float gv = row[h];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int a = a0 + j < A ? a0 + j : A - 1;
                    s[j] = fmaf(gv, w2[(int64_t)a * ld2 + h], s[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = s[j];
                for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off);
                if (lane == 0 && a0 + j < A) o[d_x + a0 + j] = fmaf(gb, b2[a0 + j], t);
            }
        }
        return;
    }
    const int a = blockIdx.x - R;
    for (int r = threadIdx.x; r < R; r += 256) row[r] = w3[(int64_t)r * ld3 + d_x + a];
    __syncthreads();
    for (int h = threadIdx.x; h <= H; h += 256) {
        float s0 = 0.f, s1 = 0.f;
        int r = 0;
        for (; r + 7 < R; r += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = g[(int64_t)(r + j) * ldg + d_x + h];
#pragma unroll
            for (int j = 0; j < 8; j += 2) { s0 = fmaf(row[r + j], v[j], s0); s1 = fmaf(row[r + j + 1], v[j + 1], s1); }
        }
        for (; r < R; ++r) s0 = fmaf(row[r], g[(int64_t)r * ldg + d_x + h], s0);
        if (h < H) g_w2[(int64_t)a * H + h] = s0 + s1;
        else if (g_b2) g_b2[a] = s0 + s1;
    }
}

}  // namespace gsn

using namespace gsn;

static bool fold_dims_ok(int64_t R, int64_t d_x, int64_t A, int64_t H) {
    return R >= 1 && d_x >= 0 && A >= 1 && H >= 1 && R <= 8192 && A <= 8192 && H <= 8192 && d_x <= 65536;
}

extern "C" int gsn_fold_weights_fwd_hip(int64_t rows, int64_t d_x, int64_t a_cols, int64_t h_cols, int64_t pad_cols, const float *w3, int64_t ld3, const float *w2,
                                        int64_t ld2, const float *b2, float *out, void *stream) {
    if (!fold_dims_ok(rows, d_x, a_cols, h_cols) || pad_cols < 0 || pad_cols > 64 || !w3 || !w2 || !b2 || !out || ld3 < d_x + a_cols || ld2 < h_cols)
        return set_error(GSN_E_INVALID, "gsn_fold_weights_fwd_hip: bad arguments");
    hipLaunchKernelGGL(fold_fwd_kernel, dim3((unsigned)rows), dim3(256), (size_t)a_cols * sizeof(float), reinterpret_cast<hipStream_t>(stream), (int)d_x,
                       (int)a_cols, (int)h_cols, (int)pad_cols, w3, ld3, w2, ld2, b2, out);
    if (hipGetLastError() != hipSuccess) return set_error(GSN_E_HIP, "gsn_fold_weights_fwd_hip: launch failed");
    return GSN_OK;
}

extern "C" int gsn_fold_weights_bwd_hip(int64_t rows, int64_t d_x, int64_t a_cols, int64_t h_cols, const float *g, int64_t ldg, const float *w3, int64_t ld3,
                                        const float *w2, int64_t ld2, const float *b2, float *g_w3, float *g_w2, float *g_b2, void *stream) {
    if (!fold_dims_ok(rows, d_x, a_cols, h_cols) || !g || !w3 || !w2 || !b2 || !g_w3 || !g_w2 || !g_b2 || ldg < d_x + h_cols + 1 || ld3 < d_x + a_cols ||
        ld2 < h_cols)
        return set_error(GSN_E_INVALID, "gsn_fold_weights_bwd_hip: bad arguments");
    const size_t lds = (size_t)((h_cols + 1 > rows ? h_cols + 1 : rows)) * sizeof(float);
    hipLaunchKernelGGL(fold_bwd_kernel, dim3((unsigned)(rows + a_cols)), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), (int)rows, (int)d_x,
                       (int)a_cols, (int)h_cols, g, ldg, w3, ld3, w2, ld2, b2, g_w3, g_w2, g_b2);
    if (hipGetLastError() != hipSuccess) return set_error(GSN_E_HIP, "gsn_fold_weights_bwd_hip: launch failed");
    return GSN_OK;
}
