// How fast can one CU store?  One workgroup of 512 threads per CU writes 256 x 256 fp32 "output tiles" (256 KiB each, rows of a
// [M][768] matrix) in the shapes a matrix kernel's epilogue can choose:
//   dword : 4-byte stores, a wave instruction = 2 rows x 128 B
//   x4    : 16-byte stores, a wave instruction = 2 rows x 512 B
//   x4nt  : the same with the nontemporal hint
// M large (HBM: 1.2 GB) or small (the same 8 tiles over and over: L2 / MALL).  Build: hipcc --offload-arch=gfx950 -O3 store_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned un4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void stores(float *out, int n_out, long tiles_total, int iters) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int it = 0; it < iters; ++it) {
        const long tile = ((long)it * gridDim.x + blockIdx.x) % tiles_total;
        float *base = out + tile * 256 * n_out + (blockIdx.x % 3) * 256;      // column tile by workgroup
        if (MODE == 0) {
            for (int q = 0; q < 128; ++q) {            // wave: rows 32 wave + .., 2 rows per instruction, 8 column groups of 32
                const int row = 32 * wave + 2 * (q >> 3) + (lane >> 5), col = 32 * (q & 7) + (lane & 31);
                base[(long)row * n_out + col] = (float)q;
            }
        } else {
            for (int q = 0; q < 32; ++q) {
                const int row = 32 * wave + 2 * (q >> 1) + (lane >> 5), col = 128 * (q & 1) + 4 * (lane & 31);
                un4 v = {(unsigned)q, 1u, 2u, 3u};
                un4 *p = reinterpret_cast<un4 *>(base + (long)row * n_out + col);
                if (MODE == 2) __builtin_nontemporal_store(v, p);
                else *p = v;
            }
        }
    }
}

int main() {
    const int n_out = 768;
    const long tiles_big = 1536;                          // 1536 x 256 rows x 768 x 4 B = 1.2 GB
    float *out; hipMalloc(&out, (size_t)tiles_big * 256 * n_out * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 40;
    const char *names[3] = {"dword", "x4", "x4 nontemporal"};
    for (long tiles : {tiles_big, 80L}) {
        for (int mode = 0; mode < 3; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL((stores<0>), dim3(240), dim3(512), 0, 0, out, n_out, tiles, iters);
                else if (mode == 1) hipLaunchKernelGGL((stores<1>), dim3(240), dim3(512), 0, 0, out, n_out, tiles, iters);
                else hipLaunchKernelGGL((stores<2>), dim3(240), dim3(512), 0, 0, out, n_out, tiles, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double bytes = 240.0 * iters * 256 * 256 * 4;
                if (rep) printf("{\"target\": \"%s\", \"stores\": \"%s\", \"TB_per_s\": %.2f, \"bytes_per_clk_per_CU_at_2.4GHz\": %.1f}\n",
                                tiles == tiles_big ? "1.2 GB (HBM)" : "63 MB (cache)", names[mode], bytes / ms / 1e9, bytes / ms / 1e9 * 1e12 / 240 / 2.4e9);
            }
        }
    }
    return 0;
}
