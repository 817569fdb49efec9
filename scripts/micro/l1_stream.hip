// How fast can one CU pull L2-resident data through its vector L1?  One persistent workgroup of 512 threads per CU reads the same
// 64 KiB "slice" shapes the fp16x3 linear kernel reads (16-byte loads), in two lane -> address patterns:
//   half : 16 lanes = 16 rows, 4 lane groups = the four 16-byte pieces of a 64-byte row segment (row pitch 640 B): every wave
//          instruction touches 16 half cache lines
//   full : 8 lanes = one 128-byte line, 8 lines per wave instruction
// and with 1 or 2 such workgroups per CU.  Build: hipcc --offload-arch=gfx950 -O3 l1_stream.hip -o l1_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned un4 __attribute__((ext_vector_type(4)));

template <int PATTERN, int DEPTH>
__global__ __launch_bounds__(512) void stream(const unsigned char *base, size_t region, int iters, unsigned *sink) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the workgroups of an XCD walk the same 1 MiB window (L2-resident after the first pass), 64 KiB per iteration
    const unsigned char *win = base + (size_t)(blockIdx.x % 8) * region;
    un4 acc = {0, 0, 0, 0};
    size_t off[8];
    for (int i = 0; i < 8; ++i) {
        if (PATTERN == 0) {       // half lines: row = 16 wave + (lane & 15) + 128 i', piece = lane >> 4; 8 loads = 4 "planes" x 2 row halves
            const int row = 16 * wave + (lane & 15) + 128 * (i & 1), plane = i >> 1;
            off[i] = (size_t)plane * (256 * 640) + (size_t)row * 640 + 16 * (lane >> 4);
        } else {                  // full lines: 8 lanes per 128-B line
            const int line = 8 * wave + (lane >> 3) + 64 * i;
            off[i] = (size_t)line * 128 + 16 * (lane & 7);
        }
    }
    for (int it = 0; it < iters; ++it) {
        const size_t step = PATTERN == 0 ? (size_t)(it % 10) * 64 : (size_t)(it % 10) * 65536;
        un4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const un4 *>(win + off[i] + step);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += v[i];
        if (DEPTH == 1) __syncthreads();
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345u) sink[0] = 1;
}

int main() {
    const size_t region = 1u << 20;                       // 1 MiB per workgroup window (pattern 0 spans 4 x 160 KiB + steps)
    unsigned char *buf; unsigned *sink;
    hipMalloc(&buf, 480 * region + (1 << 20)); hipMalloc(&sink, 4);
    hipMemset(buf, 1, 480 * region + (1 << 20));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int wgs : {256, 512}) {
        for (int pat = 0; pat < 2; ++pat) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (pat == 0) hipLaunchKernelGGL((stream<0, 0>), dim3(wgs), dim3(512), 0, 0, buf, region, iters, sink);
                else hipLaunchKernelGGL((stream<1, 0>), dim3(wgs), dim3(512), 0, 0, buf, region, iters, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double bytes = (double)wgs * iters * 65536.0;
                if (rep) printf("{\"workgroups\": %d, \"pattern\": \"%s\", \"TB_per_s\": %.2f, \"bytes_per_clk_per_CU_at_2.4GHz\": %.1f}\n", wgs, pat ? "full lines" : "half lines",
                                bytes / ms / 1e9, bytes / ms / 1e9 * 1e12 / 256 / 2.4e9);
            }
        }
    }
    return 0;
}
