// The weight-fragment stream of csrc/layer_rr.hip in isolation: every wave reads NF fragments (1 KiB, ds_read_b128, lane-linear) per step,
// D steps ahead, and issues NM MFMAs (two accumulators, alternating) per step that consume them.  8 waves per CU (2 per SIMD) like the
// kernel.  cycles per step vs the MFMA pipe time (32 NM x 2 waves per SIMD).
//   build: hipcc --offload-arch=gfx950 -O3 scripts/micro/lds_frag_stream.hip -o scripts/micro/bin/lds_frag_stream
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int NM, int NF, int D, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(unsigned long long *out, float *sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 150 * 1024 / 4; i += 64 * WAVES) reinterpret_cast<unsigned *>(smem)[i] = 0x3c003c00u;
    __syncthreads();
    const u4 *frag = reinterpret_cast<const u4 *>(smem) + lane;
    h8 a;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(0.001f * (lane + i));
    f16v acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    u4 q[D + 1][NF];
    int pos = (threadIdx.x >> 6) * 7;
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int f = 0; f < NF; ++f) { q[d][f] = frag[((pos++) % 150) * 64]; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += D + 1) {
#pragma unroll
        for (int s = 0; s < D + 1; ++s) {
            // read the fragments of step s + D into slot (s + D) % (D + 1), consume slot s
#pragma unroll
            for (int f = 0; f < NF; ++f) q[(s + D) % (D + 1)][f] = frag[((pos++) % 150) * 64];
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const h8 b = __builtin_bit_cast(h8, q[s][m % NF]);
                if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sacc = 0.f;
    for (int i = 0; i < 16; ++i) sacc += acc0[i] + acc1[i];
    if (sacc == 12345.678f) sink[threadIdx.x] = sacc;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int NM, int NF, int D, int WAVES>
static void run() {
    unsigned long long *d; float *sink;
    (void)hipMalloc(&d, 16); (void)hipMalloc(&sink, 4096);
    const int iters = 1200;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k<NM, NF, D, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NM, NF, D, WAVES>), dim3(256), dim3(64 * WAVES), 150 * 1024, 0, d, sink, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h = 0;
    (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%d MFMA + %d fragment reads per step, %d step(s) ahead, %d waves/CU: %7.1f cycles / step   (MFMA pipe alone: %d)\n", NM, NF, D, WAVES,
           (double)h / iters, 32 * NM * (WAVES > 4 ? 2 : 1));
    (void)hipFree(d); (void)hipFree(sink);
}

int main() {
    run<4, 4, 1, 8>(); run<4, 4, 2, 8>(); run<4, 4, 3, 8>();
    run<6, 4, 1, 8>(); run<6, 4, 2, 8>();
    run<2, 2, 1, 8>(); run<2, 2, 3, 8>();
    run<4, 4, 1, 4>(); run<6, 4, 1, 4>();
    return 0;
}
