// The main loop of the edge stage of csrc/layer_w.hip in isolation (K = 256 + 16 edge rows, 64 edge rows = two 32-row
// sub-blocks per step): ONE wave per SIMD (512 registers), per 16-column chunk 24 MFMAs (2 sub-blocks x 4 feature blocks x 3 plane
// products) that share 8 weight fragments from LDS, with the NEXT chunk's gathered fp32 values converted to scaled fp16 planes
// and the gathers of chunk c + PDG issued in between.  cycles per chunk step vs the matrix pipe's 24 x 32 = 768.
//   build: hipcc --offload-arch=gfx950 -O3 scripts/micro/wide_edge_loop.hip -o scripts/micro/bin/wide_edge_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_h2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a, b}, h2)); }
__device__ __forceinline__ float res_lo_s(float a, float s, unsigned pair) { float r; asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(s), "v"(pair)); return r; }
__device__ __forceinline__ float res_hi_s(float a, float s, unsigned pair) { float r; asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(s), "v"(pair)); return r; }
__device__ __forceinline__ void split2s(float a, float b, float s, unsigned &hi, unsigned &lo) {
    hi = pack_h2(a * s, b * s);
    lo = pack_h2(res_lo_s(a, s, hi), res_hi_s(b, s, hi));
}
#define MFH(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, A), __builtin_bit_cast(h8, B), C, 0, 0, 0)
typedef const __attribute__((address_space(3))) u4 *ldsp;
__device__ __forceinline__ u4 lds_frag(const unsigned (&base)[3], int f) {
    const int byte = f * 1024;
    return *reinterpret_cast<ldsp>(base[byte >> 16] + (unsigned)(byte & 0xffff));
}
typedef const __attribute__((address_space(1))) f4 *gptr;

constexpr int NC = 16;     // x chunks (the 17th, per-edge chunk behaves the same)

template <int PDG, int MIXV>
__global__ __launch_bounds__(256) void k(const float *x, const int *idx, int n_rows, unsigned long long *out, float *sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 136 * 1024 / 4; i += 256) reinterpret_cast<unsigned *>(smem)[i] = 0x3c003c00u;
    __syncthreads();
    unsigned ldsb[3];
    ldsb[0] = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem + 16u * (unsigned)lane;
    ldsb[1] = ldsb[0] + 0x10000u; ldsb[2] = ldsb[0] + 0x20000u;
    asm volatile("" : "+v"(ldsb[1]), "+v"(ldsb[2]));
    f16v sacc[4];
    for (int fb = 0; fb < 4; ++fb) for (int r = 0; r < 16; ++r) sacc[fb][r] = 0.f;
    const int slot = blockIdx.x * 4 + wave;
    unsigned long long t_loop = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(ldsb[0]), "+v"(ldsb[1]), "+v"(ldsb[2]));
        // rows of the two sub-blocks, two roles
        const int e = (slot * iters + it) * 64;
        const char *rowp[2][2];
        float rs[2];
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
#pragma unroll
            for (int role = 0; role < 2; ++role) {
                const int row = idx[role * n_rows + (e + 32 * sb + li) % n_rows];
                rowp[sb][role] = reinterpret_cast<const char *>(x) + (size_t)row * 512 + 32 * lh;
            }
            rs[sb] = __uint_as_float((unsigned)(127 + ((e + sb + li) & 3)) << 23);
        }
        f4 raw[PDG + 1][2][2];
        auto issue = [&](int c) {
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    raw[c % (PDG + 1)][sb][j] = *reinterpret_cast<gptr>((unsigned long long)(rowp[sb][c >> 3] + 64 * (c & 7) + 16 * j));
        };
        u4 Ah[2], Al[2], Nh[2], Nl[2];
        auto convert = [&](int c, u4 (&H)[2], u4 (&L)[2]) {
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                unsigned h[4], l[4];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f4 v = raw[c % (PDG + 1)][sb][j];
                    split2s(v.x, v.y, rs[sb], h[2 * j], l[2 * j]);
                    split2s(v.z, v.w, rs[sb], h[2 * j + 1], l[2 * j + 1]);
                }
                H[sb] = u4{h[0], h[1], h[2], h[3]};
                L[sb] = u4{l[0], l[1], l[2], l[3]};
            }
        };
#pragma unroll
        for (int c = 0; c < PDG; ++c) issue(c);
        f16v acc[2][4];
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int fb = 0; fb < 4; ++fb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[sb][fb][r] = 0.f;
        u4 fr[8], nf[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) fr[q] = lds_frag(ldsb, q);
        convert(0, Ah, Al);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long l0 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (c + PDG < NC) issue(c + PDG);
            if (c + 1 < NC) {
#pragma unroll
                for (int q = 0; q < 8; ++q) nf[q] = lds_frag(ldsb, 8 * (c + 1) + q);
            }
#pragma unroll
            for (int fb = 0; fb < 4; ++fb)
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) {
                    MFH(Al[sb], fr[2 * fb], acc[sb][fb]);
                }
#pragma unroll
            for (int fb = 0; fb < 4; ++fb)
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) {
                    MFH(Ah[sb], fr[2 * fb + 1], acc[sb][fb]);
                }
#pragma unroll
            for (int fb = 0; fb < 4; ++fb)
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) {
                    MFH(Ah[sb], fr[2 * fb], acc[sb][fb]);
                }
            if (c + 1 < NC) convert(c + 1, Nh, Nl);
            if (MIXV > 0) {
#pragma unroll
                for (int q = 0; q < 24; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, MIXV, 0); }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) { Ah[sb] = Nh[sb]; Al[sb] = Nl[sb]; }
#pragma unroll
            for (int q = 0; q < 8; ++q) fr[q] = nf[q];
        }
        __builtin_amdgcn_sched_barrier(0);
        t_loop += __builtin_amdgcn_s_memtime() - l0;
        __builtin_amdgcn_sched_barrier(0);
        // stand-in for the epilogue: fold the accumulators into sacc
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int fb = 0; fb < 4; ++fb)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[fb][r] += acc[sb][fb][r];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int fb = 0; fb < 4; ++fb) for (int r = 0; r < 16; ++r) s += sacc[fb][r];
    if (s == 12345.678f) sink[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 100) { out[0] = t1 - t0; out[1] = t_loop; }
}

template <int PDG, int MIXV>
static void run(const float *x, const int *idx, int n_rows) {
    unsigned long long *d; float *sink;
    (void)hipMalloc(&d, 16); (void)hipMalloc(&sink, 4096);
    const int iters = 40;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k<PDG, MIXV>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<PDG, MIXV>), dim3(256), dim3(256), 150 * 1024, 0, x, idx, n_rows, d, sink, iters);
    (void)hipDeviceSynchronize();
    unsigned long long hh[2] = {0, 0};
    (void)hipMemcpy(hh, d, 16, hipMemcpyDeviceToHost);
    const unsigned long long h = hh[0];
    printf("gathers %d chunk(s) ahead, %d vector instr. per MFMA prescribed: %8.1f cycles / chunk step (24 MFMA = 768), %8.0f / 64-row pair; the chunk loop alone %8.1f / step\n", PDG, MIXV,
           (double)h / iters / NC, (double)h / iters, (double)hh[1] / iters / NC);
    (void)hipFree(d); (void)hipFree(sink);
}

int main() {
    const int n_rows = 1 << 20;
    float *x; int *idx;
    (void)hipMalloc(&x, (size_t)n_rows * 512); (void)hipMalloc(&idx, (size_t)n_rows * 8);
    (void)hipMemset(x, 0, (size_t)n_rows * 512);
    std::vector<int> h(2 * n_rows);
    // target-sorted rows: role 0 = row / 2 (consecutive edges share a target), role 1 = a neighbour within +-12 rows
    for (int i = 0; i < n_rows; ++i) { h[i] = i / 2; int s = i / 2 + ((i * 7) % 25) - 12; h[n_rows + i] = s < 0 ? 0 : (s >= n_rows ? n_rows - 1 : s); }
    (void)hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<2, 0>(x, idx, n_rows); run<3, 0>(x, idx, n_rows); run<4, 0>(x, idx, n_rows);
    run<3, 2>(x, idx, n_rows); run<3, 3>(x, idx, n_rows);
    return 0;
}
