// Sustained fp32 MFMA rate of the device: every wave issues independent v_mfma_f32_32x32x2_f32 back to back from
// registers (no memory traffic).  Build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
static void run(int wgs_per_cu, int iters) {
    float *out;
    const int grid = 256 * wgs_per_cu;
    hipMalloc(&out, sizeof(float) * grid * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<CHAINS>, dim3(grid), dim3(256), 0, 0, out, iters / 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<CHAINS>, dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 /*waves*/ * iters * 8.0 * CHAINS * 4096.0;
    printf("{\"chains_per_wave\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"TFLOPs\": %.1f}\n", CHAINS, wgs_per_cu, ms, flop / ms / 1e9);
    hipFree(out);
}

int main() {
    run<1>(1, 20000);
    run<2>(1, 20000);
    run<4>(1, 10000);
    run<2>(2, 10000);
    run<2>(4, 10000);
    run<2>(1, 200000);   // ~1 s: sustained clocks
    return 0;
}
