// Operand layout of v_mfma_f32_32x32x16_bf16 and accuracy of the exact three-way bf16 split of fp32 operands
// (x = hi + mid + lo by truncation; products of combined order <= 2: hh, hm, mh, mm, hl, lh) against fp64.
// Build: hipcc --offload-arch=gfx950 -O3 bf16x6_check.hip -o bf16x6_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ inline void split3(float x, unsigned short &h, unsigned short &m, unsigned short &l) {
    const unsigned xb = __float_as_uint(x);
    const float fh = __uint_as_float(xb & 0xffff0000u);
    const float r1 = x - fh;
    const unsigned rb = __float_as_uint(r1);
    const float fm = __uint_as_float(rb & 0xffff0000u);
    const float r2 = r1 - fm;
    h = xb >> 16; m = rb >> 16; l = __float_as_uint(r2) >> 16;
}

// one wave: C[32][32] = A[32][K] * B[K][32]^T-layout (B given as [32 cols][K]) ; K multiple of 16
__global__ void k(const float *A, const float *Bm, float *C6, float *C3, float *C1, float *CH, int K, float sa, float sb) {
    const int l = threadIdx.x, li = l & 31, lh = l >> 5;
    f32x16 a6 = {0}, a3 = {0}, a1 = {0}, ah3 = {0};
    for (int k0 = 0; k0 < K; k0 += 16) {
        s16x8 ah, am, al, bh, bm, bl;
        for (int e = 0; e < 8; ++e) {
            unsigned short h, m, lo;
            split3(A[li * K + k0 + 8 * lh + e], h, m, lo); ah[e] = h; am[e] = m; al[e] = lo;
            split3(Bm[li * K + k0 + 8 * lh + e], h, m, lo); bh[e] = h; bm[e] = m; bl[e] = lo;
        }
#define MF(acc, x, y) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), acc, 0, 0, 0)
        // fp16 two-way split (x = h + l, 11 + 11 bits, operands pre-scaled by powers of two): 3 products hh, hl, lh
        {
            f16x8 xh, xl, yh, yl;
            for (int e = 0; e < 8; ++e) {
                const float x = A[li * K + k0 + 8 * lh + e] * sa, y = Bm[li * K + k0 + 8 * lh + e] * sb;
                xh[e] = (_Float16)x; xl[e] = (_Float16)(x - (float)xh[e]);
                yh[e] = (_Float16)y; yl[e] = (_Float16)(y - (float)yh[e]);
            }
            ah3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, yh, ah3, 0, 0, 0);
            ah3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yl, ah3, 0, 0, 0);
            ah3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yh, ah3, 0, 0, 0);
        }
        MF(a1, ah, bh);
        MF(a3, ah, bh); MF(a3, ah, bm); MF(a3, am, bh);
        // small terms first
        MF(a6, al, bh); MF(a6, ah, bl); MF(a6, am, bm); MF(a6, ah, bm); MF(a6, am, bh); MF(a6, ah, bh);
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        C6[row * 32 + li] = a6[r]; C3[row * 32 + li] = a3[r]; C1[row * 32 + li] = a1[r]; CH[row * 32 + li] = ah3[r] / (sa * sb);
    }
}

int main() {
    const int K = 160;
    std::vector<float> A(32 * K), B(32 * K);
    srand(1);
    for (auto &v : A) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
    for (auto &v : B) v = (rand() / (float)RAND_MAX - 0.5f) * 0.5f;
    float *dA, *dB, *d6, *d3, *d1, *dh;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&d6, 4096); hipMalloc(&d3, 4096); hipMalloc(&d1, 4096); hipMalloc(&dh, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    for (int pass = 0; pass < 2; ++pass) {
    const float sa = pass ? 1.f : 1.f, sb = pass ? 256.f : 1.f;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, d6, d3, d1, dh, K, sa, sb);
    std::vector<float> c6(1024), c3(1024), c1(1024), ch(1024);
    hipMemcpy(c6.data(), d6, 4096, hipMemcpyDeviceToHost); hipMemcpy(c3.data(), d3, 4096, hipMemcpyDeviceToHost); hipMemcpy(c1.data(), d1, 4096, hipMemcpyDeviceToHost); hipMemcpy(ch.data(), dh, 4096, hipMemcpyDeviceToHost);
    double e6 = 0, e3 = 0, e1 = 0, ef = 0, mx = 0, eh = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            double ref = 0; float f = 0.f;
            for (int q = 0; q < K; ++q) { ref += (double)A[i * K + q] * (double)B[j * K + q]; f = fmaf(A[i * K + q], B[j * K + q], f); }
            mx = fmax(mx, fabs(ref));
            e6 = fmax(e6, fabs(c6[i * 32 + j] - ref)); e3 = fmax(e3, fabs(c3[i * 32 + j] - ref)); e1 = fmax(e1, fabs(c1[i * 32 + j] - ref));
            ef = fmax(ef, fabs(f - ref)); eh = fmax(eh, fabs(ch[i * 32 + j] - ref));
        }
    printf("{\"K\": %d, \"max_abs_ref\": %.4f, \"max_err_over_max_ref\": {\"bf16x1\": %.3e, \"bf16x3\": %.3e, \"bf16x6\": %.3e, \"fp16x3 (weights scaled by %g)\": %.3e, \"fp32_fma_loop\": %.3e}}\n", K, mx, e1 / mx, e3 / mx, e6 / mx, (double)sb, eh / mx, ef / mx);
    }
    return 0;
}
