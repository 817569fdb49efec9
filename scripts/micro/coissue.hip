// What can a SIMD issue from OTHER waves while its matrix pipe is saturated with fp32 MFMAs?
// Workgroup = 12 waves: waves 0-7 (two per SIMD) run dependent v_mfma_f32_32x32x2_f32 chains; waves 8-11 (one per SIMD) run
// 64 x REPS instructions of one kind (VALU add chain / SALU add chain / scalar compare+branch / LDS read) and time them with
// s_memtime.  Printed: cycles per instruction of the probe waves with and without the MFMA waves, and the MFMA waves' own time.
// Build: hipcc --offload-arch=gfx950 -O3 coissue.hip -o coissue ; run: ./coissue
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))

template <int MODE, int CHAINS, bool BF16>
__global__ __launch_bounds__(768) void probe(float *out, unsigned long long *cyc, int mfma_iters, int reps, int prio, int first, int mfma_waves) {
    __shared__ float sh[1024];
    const int w0 = threadIdx.x >> 6;
    const int w = first ? (w0 < 4 ? w0 + 8 : w0 - 4) : w0;   // first: the probe waves are the OLDEST waves of the workgroup
    sh[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (w < 8) {
        if (w >= mfma_waves) return;
        f32x16 acc[CHAINS];
        for (int c = 0; c < CHAINS; ++c)
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        const float a = threadIdx.x * 1e-3f, b = 1.f + threadIdx.x * 1e-4f;
        bf16x8 a8, b8;
        for (int q = 0; q < 8; ++q) { a8[q] = (__bf16)(a + q); b8[q] = (__bf16)(b - q); }
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < mfma_iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8 / CHAINS; ++u)
#pragma unroll
                for (int c = 0; c < CHAINS; ++c) {
                    if (BF16) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[c], 0, 0, 0);
                    else acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
                }
        }
        float s = 0.f;
        for (int c = 0; c < CHAINS; ++c)
            for (int r = 0; r < 16; ++r) s += acc[c][r];
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        out[blockIdx.x * 768 + threadIdx.x] = s;
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[w] = t1 - t0;
        return;
    }
    if (prio) __builtin_amdgcn_s_setprio(3);
    float v = threadIdx.x;
    int sc = __builtin_amdgcn_readfirstlane(reps);
    unsigned lds_addr = (threadIdx.x & 255) * 4;
    // let the MFMA waves get going
    for (int i = 0; i < 200; ++i) asm volatile("s_nop 15");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < reps; ++i) {
        if (MODE == 0) asm volatile(R64("v_add_f32 %0, %0, %0\n") : "+v"(v));
        if (MODE == 1) asm volatile(R64("s_add_u32 %0, %0, 1\n") : "+s"(sc)::"scc");
        if (MODE == 2) asm volatile(R64("s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 1f\n s_add_u32 %0, %0, 1\n 1:\n") : "+s"(sc)::"scc");
        if (MODE == 3) asm volatile(R64("ds_read_b32 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : "=v"(v) : "v"(lds_addr) : "memory");
        if (MODE == 4) asm volatile(R64("v_mov_b32 %0, %0\n s_add_u32 %1, %1, 1\n") : "+v"(v), "+s"(sc)::"scc");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 768 + threadIdx.x] = v + sc;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) cyc[w] = t1 - t0;
}

template <int MODE, int CHAINS, bool BF16 = false>
static void run(const char *name, int per_rep, int prio, int first, int mfma_waves) {
    float *out;
    unsigned long long *cyc, h[12];
    hipMalloc(&out, 256 * 768 * 4);
    hipMalloc(&cyc, 12 * 8);
    const int reps = 200;
    double res[2], mf[2];
    for (int with = 0; with < 2; ++with) {
        hipMemset(cyc, 0, 96);
        hipLaunchKernelGGL((probe<MODE, CHAINS, BF16>), dim3(256), dim3(768), 0, 0, out, cyc, with ? 4000 : 0, reps, prio, first, mfma_waves);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, 96, hipMemcpyDeviceToHost);
        res[with] = (double)h[8] / (reps * per_rep);
        mf[with] = (double)h[0];
    }
    printf("{\"probe\": \"%s\", \"mfma_waves_per_simd\": %d, \"chains_per_mfma_wave\": %d, \"mfma\": \"%s\", \"probe_waves_oldest\": %d, \"prio\": %d, \"cycles_per_instr_alone\": %.2f, \"cycles_per_instr_under_mfma\": %.2f, \"mfma_wave_cycles_per_mfma\": %.1f}\n",
           name, mfma_waves / 4, CHAINS, BF16 ? "32x32x16_bf16" : "32x32x2_f32", first, prio, res[0], res[1], mf[1] / (4000.0 * 8));
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0, 1>("v_add_f32 (dependent)", 64, 0, 0, 8);
    run<0, 1>("v_add_f32 (dependent)", 64, 0, 0, 4);
    run<0, 4>("v_add_f32 (dependent)", 64, 0, 0, 4);
    run<1, 2>("s_add_u32 (dependent)", 64, 0, 0, 4);
    run<3, 2>("ds_read_b32 x64 + wait", 64, 0, 0, 8);
    run<0, 1, true>("v_add_f32 (dependent)", 64, 0, 0, 8);
    run<0, 1, true>("v_add_f32 (dependent)", 64, 0, 0, 4);
    run<0, 2, true>("v_add_f32 (dependent)", 64, 0, 0, 4);
    run<0, 2, true>("v_add_f32 (dependent)", 64, 1, 0, 8);
    run<1, 2, true>("s_add_u32 (dependent)", 64, 0, 0, 8);
    run<3, 2, true>("ds_read_b32 x64 + wait", 64, 0, 0, 8);
    return 0;
}
