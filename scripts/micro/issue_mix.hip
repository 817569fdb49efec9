// How do one or two waves of a SIMD issue mixes of 16-bit MFMAs and plain vector instructions?  (gfx950; the questions behind
// csrc/layer_rr.hip's structure.)  Every test: `waves` waves per SIMD run the same loop (or, "split", the first wave of a SIMD only
// MFMAs and the second only vector instructions); s_memtime around the loop of wave 0; cycles per loop iteration.
//   build: hipcc --offload-arch=gfx950 -O3 scripts/micro/issue_mix.hip -o /tmp/issue_mix     run: /tmp/issue_mix
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define MF(acc) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0)
#define VF(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2))

// mode: number of vector instructions per MFMA (0 = MFMAs only; -1 = vector instructions only; -2 = dependent vector chain only)
template <int MODE, bool SPLIT>
__global__ __launch_bounds__(512) void k(unsigned long long *out, float *sink, int iters) {
    const int wave = threadIdx.x >> 6;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * i); }
    f16v acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.5f + i;
    const float c1 = 0.999f, c2 = 0.001f;
    const bool do_m = SPLIT ? (wave < 4) : (MODE >= 0), do_v = SPLIT ? (wave >= 4) : (MODE != 0);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (SPLIT) {
            if (do_m) { MF(acc0); MF(acc1); MF(acc0); MF(acc1); }
            if (do_v) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { VF(v[0]); VF(v[1]); VF(v[2]); VF(v[3]); VF(v[4]); VF(v[5]); VF(v[6]); VF(v[7]); }
            }
        } else if (MODE == -2) {
#pragma unroll
            for (int q = 0; q < 32; ++q) VF(v[0]);
        } else if (MODE == -1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { VF(v[0]); VF(v[1]); VF(v[2]); VF(v[3]); VF(v[4]); VF(v[5]); VF(v[6]); VF(v[7]); }
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (m & 1) MF(acc1); else MF(acc0);
#pragma unroll
                for (int q = 0; q < MODE; ++q) VF(v[q & 7]);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) sink[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (threadIdx.x == 256 && blockIdx.x == 0) out[1] = t1 - t0;
}

template <int MODE, bool SPLIT>
static void run(const char *name, int threads, int per_iter_m, int per_iter_v) {
    unsigned long long *d;
    float *sink;
    hipMalloc(&d, 16);
    hipMalloc(&sink, 4096);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE, SPLIT>), dim3(256), dim3(threads), 0, 0, d, sink, iters);
    hipDeviceSynchronize();
    unsigned long long h[2];
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-44s waves/SIMD %d: %7.1f cycles / iteration (wave 0)", name, threads / 256, (double)h[0] / iters);
    if (threads > 256) printf("  %7.1f (wave 4)", (double)h[1] / iters);
    printf("   [%d MFMA, %d VALU per iteration and wave]\n", per_iter_m, per_iter_v);
    hipFree(d);
    hipFree(sink);
}

int main() {
    for (int threads : {256, 512}) {
        run<0, false>("4 MFMA (two accumulators, alternating)", threads, 4, 0);
        run<-1, false>("32 independent v_fma (8 chains)", threads, 0, 32);
        run<-2, false>("32 dependent v_fma (1 chain)", threads, 0, 32);
        run<2, false>("4 x (MFMA + 2 v_fma)", threads, 4, 8);
        run<4, false>("4 x (MFMA + 4 v_fma)", threads, 4, 16);
        run<6, false>("4 x (MFMA + 6 v_fma)", threads, 4, 24);
        run<8, false>("4 x (MFMA + 8 v_fma)", threads, 4, 32);
        run<12, false>("4 x (MFMA + 12 v_fma)", threads, 4, 48);
    }
    run<0, true>("split: wave 0-3 4 MFMA | wave 4-7 32 v_fma", 512, 4, 32);
    return 0;
}
