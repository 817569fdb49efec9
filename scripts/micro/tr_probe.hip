// r06 probe: what ds_read_b64_tr_b16 (__builtin_amdgcn_ds_read_tr16_b64_v4f16) returns.  LDS halfs hold their own index; every lane supplies an
// address; the four halfs each lane gets back are printed as LDS indices.  Case A: lane l supplies base + 8 l bytes (64 lanes x 8 B contiguous).
// Case B: lane l supplies the address of row (l & 15) >> 2, column quad (l & 3) of a [4][16] block with a row pitch of 64 halfs, block l >> 4.
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 fp16x4_t;
typedef __attribute__((address_space(3))) fp16x4_t *lds4_t;

__global__ void probe(int mode, float *out) {
    __shared__ __attribute__((aligned(16))) _Float16 buf[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) buf[i] = (_Float16)(float)(i & 2047);
    __syncthreads();
    const int l = threadIdx.x;
    int idx;                                   // index (in halfs) of the 4 contiguous halfs this lane points at
    if (mode == 0) idx = 4 * l;
    else idx = (l >> 4) * 256 + (((l & 15) >> 2) * 64) + (l & 3) * 4;
    fp16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds4_t)(buf + idx));
    for (int j = 0; j < 4; ++j) out[mode * 256 + l * 4 + j] = (float)v[j];
}

int main() {
    float *d, h[512];
    hipMalloc(&d, sizeof(h));
    for (int mode = 0; mode < 2; ++mode) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, mode, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int mode = 0; mode < 2; ++mode) {
        printf("mode %d (lane: the four LDS half indices it received)\n", mode);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %4.0f %4.0f %4.0f %4.0f%s", l, h[mode * 256 + l * 4], h[mode * 256 + l * 4 + 1], h[mode * 256 + l * 4 + 2], h[mode * 256 + l * 4 + 3], (l & 3) == 3 ? "\n" : "");
    }
    return 0;
}
