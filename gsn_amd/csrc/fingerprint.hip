// Content fingerprint of a set of device tensors, for the host-side caches that are keyed on PyTorch's version counters (prepared weight
// fragments, folded weights, eval-mode BatchNorm vectors: gsn_amd/layers.py).  A write through `.data` does not move a version counter; the
// layer enqueues this kernel behind its forward (one launch, a few hundred KB of reads), the 64-bit result travels to pinned host memory
// asynchronously and is compared at the next forward -- no host synchronisation anywhere.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "gsn_internal.h"

namespace gsn {

__device__ __forceinline__ unsigned long long fp_mix(unsigned long long x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

// meta: [0, n) base pointers, [n, 2n) sizes in 4-byte words.  grid (blocks, n).  The sum over (tensor, position, word) of a mixed 64-bit
// value: independent of the order in which workgroups finish.
__global__ __launch_bounds__(256) void fingerprint_kernel(int n, const int64_t *meta, unsigned long long *acc) {
    const int t = blockIdx.y;
    const uint32_t *p = reinterpret_cast<const uint32_t *>(meta[t]);
    const int64_t words = meta[n + t];
    unsigned long long s = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (int64_t)gridDim.x * 256)
        s += fp_mix(((unsigned long long)p[i] << 32) ^ (unsigned long long)(i * 0x9E3779B1u + (unsigned)t * 0x85EBCA77u + 1u));
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(acc, s);
}

}  // namespace gsn

using namespace gsn;

extern "C" int gsn_fingerprint_hip(int n_tensors, const int64_t *meta, int64_t max_words, unsigned long long *acc, unsigned long long *host_out,
                                   void *stream) {
    if (n_tensors < 1 || n_tensors > 65535 || !meta || !acc || max_words < 0) return set_error(GSN_E_INVALID, "gsn_fingerprint_hip: bad arguments");
    int64_t bx = (max_words + 2047) / 2048;      // ~8 words per thread
    bx = bx < 1 ? 1 : (bx > 64 ? 64 : bx);
    hipLaunchKernelGGL(fingerprint_kernel, dim3((unsigned)bx, (unsigned)n_tensors), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n_tensors, meta, acc);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "fingerprint_kernel: %s", hipGetErrorString(e));
    if (host_out) {      // pinned host memory: the copy is asynchronous, ordered behind the kernel on the stream
        e = hipMemcpyAsync(host_out, acc, sizeof(unsigned long long), hipMemcpyDeviceToHost, reinterpret_cast<hipStream_t>(stream));
        if (e != hipSuccess) return set_error(GSN_E_HIP, "gsn_fingerprint_hip: copy to the host: %s", hipGetErrorString(e));
    }
    return GSN_OK;
}
