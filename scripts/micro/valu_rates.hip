// Issue cost of the VALU / LDS instructions the staging code of layer_fused.hip is made of, and of MFMA chains with and
// without other instructions between dependent MFMAs.  One wave per SIMD (256 threads), s_memtime around an unrolled body.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define REP16(x) x x x x x x x x x x x x x x x x
#define TIMED(NAME, IDX, N, BODY)                                           \
    {                                                                       \
        __builtin_amdgcn_s_barrier();                                       \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();               \
        asm volatile("s_waitcnt lgkmcnt(0)");                               \
        for (int it = 0; it < 64; ++it) { BODY }                            \
        asm volatile("s_nop 7\n s_nop 7\n s_waitcnt lgkmcnt(0)" ::: "memory"); \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();               \
        if (threadIdx.x == 0) out[IDX] = (float)(t1 - t0) / (64.0f * N);    \
    }

__global__ __launch_bounds__(256) void k(float *out, float *sink, int waves_active) {
    __shared__ float lds[4096];
    if ((int)(threadIdx.x >> 6) >= waves_active) return;
    float a = threadIdx.x * 0.37f + 1.f, b = 0.11f * threadIdx.x, c = 2.f, d = 3.f;
    unsigned u = threadIdx.x * 2654435761u, v = u ^ 0x5555u;
    f2 p = {a, b}, q = {c, d};
    h2 hh;
    lds[threadIdx.x] = a;
    TIMED("v_and", 0, 16, REP16(asm volatile("v_and_b32 %0, 0x7fffffff, %0" : "+v"(u));))
    TIMED("v_max3_u32", 1, 16, REP16(asm volatile("v_max3_u32 %0, %0, %1, %1" : "+v"(u) : "v"(v));))
    TIMED("v_cvt_pk_f16_f32", 2, 16, REP16(asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u) : "v"(a), "v"(b));))
    TIMED("v_cvt_f32_f16", 3, 16, REP16(asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a) : "v"(u));))
    TIMED("v_cvt_f32_f16_sdwa", 4, 16, REP16(asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(b) : "v"(u));))
    TIMED("v_pk_add_f32", 5, 16, REP16(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(q));))
    TIMED("v_pk_mul_f32", 6, 16, REP16(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(q));))
    TIMED("v_mul_f32", 7, 16, REP16(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(c));))
    TIMED("v_fma_f32", 8, 16, REP16(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(c));))
    TIMED("v_max_u32_dpp", 9, 16, REP16(asm volatile("s_nop 1\n v_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(u));))
    TIMED("v_or3", 10, 16, REP16(asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(u) : "v"(v));))
    unsigned addr = (threadIdx.x & 63) * 16;
    float4 r4;
    TIMED("ds_read_b128 (indep)", 11, 16, REP16(asm volatile("ds_read_b128 %0, %1" : "=v"(r4) : "v"(addr));))
    TIMED("ds_write_b64", 12, 16, REP16(asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(p));))
    TIMED("ds_write_b32", 13, 16, REP16(asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(a));))
    // independent-operand versions (no dependency between consecutive instructions)
    float a0 = a, a1 = b, a2 = c, a3 = d;
    TIMED("v_cvt_pk indep x4", 14, 16, REP16(asm volatile("v_cvt_pk_f16_f32 %0, %4, %5\n v_cvt_pk_f16_f32 %1, %4, %5\n v_cvt_pk_f16_f32 %2, %4, %5\n v_cvt_pk_f16_f32 %3, %4, %5" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(a), "v"(b));))
    // MFMA chains
    f16x8 fa, fb;
    for (int e = 0; e < 8; ++e) { fa[e] = (_Float16)(0.01f * (threadIdx.x + e)); fb[e] = (_Float16)(0.02f * e); }
    f32x16 acc0 = {0}, acc1 = {0};
    TIMED("mfma same acc back-to-back", 15, 16, REP16(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc0, 0, 0, 0);))
    TIMED("mfma alternating 2 acc", 16, 16, REP16(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc1, 0, 0, 0);))
    TIMED("mfma same acc + 1 valu between", 17, 16, REP16(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc0, 0, 0, 0); asm volatile("v_and_b32 %0, 0x7fffffff, %0" : "+v"(u));))
    TIMED("mfma same acc + ds_read between", 18, 16, REP16(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc0, 0, 0, 0); asm volatile("ds_read_b128 %0, %1" : "=v"(r4) : "v"(addr));))
    TIMED("mfma 2 acc + ds_read between", 19, 16, REP16(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc0, 0, 0, 0); asm volatile("ds_read_b128 %0, %1" : "=v"(r4) : "v"(addr)); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc1, 0, 0, 0); asm volatile("ds_read_b128 %0, %1" : "=v"(r4) : "v"(addr));))
    TIMED("mfma 3-chain then 1 valu", 20, 16, REP16(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc0, 0, 0, 0); acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc0, 0, 0, 0); acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc0, 0, 0, 0); asm volatile("v_and_b32 %0, 0x7fffffff, %0" : "+v"(u));))
    sink[threadIdx.x] = a + b + p.x + q.y + (float)u + r4.x + a0 + a1 + a2 + a3 + acc0[0] + acc1[3] + (float)hh.x;
}

int main() {
    float *out, *sink;
    hipMalloc(&out, 64 * 4); hipMalloc(&sink, 4096);
    const char *names[] = {"v_and_b32", "v_max3_u32", "v_cvt_pk_f16_f32 (dependent dst only)", "v_cvt_f32_f16", "v_cvt_f32_f16_sdwa", "v_pk_add_f32", "v_pk_mul_f32", "v_mul_f32",
                           "v_fma_f32", "s_nop1 + v_max_u32_dpp", "v_or3_b32", "ds_read_b128", "ds_write_b64", "ds_write_b32", "v_cvt_pk_f16_f32 x4 independent (per group of 4)",
                           "mfma same acc back-to-back", "mfma alternating 2 acc (per pair)", "mfma same acc + 1 valu", "mfma same acc + ds_read_b128", "mfma 2 acc, ds_read after each (per pair)", "3 mfma same acc + 1 valu (per group)"};
    for (int waves = 1; waves <= 4; waves += 3) {
        hipMemset(out, 0, 256);
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, out, sink, waves);
        hipDeviceSynchronize();
        std::vector<float> h(64);
        hipMemcpy(h.data(), out, 256, hipMemcpyDeviceToHost);
        printf("{\"waves_in_workgroup\": %d, \"cycles_per_instruction\": {", waves);
        for (int i = 0; i < 21; ++i) printf("%s\"%s\": %.2f", i ? ", " : "", names[i], h[i]);
        printf("}}\n");
    }
    return 0;
}
