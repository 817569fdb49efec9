// Identifier encoding on the device: multi-hot encoding of integer count columns.
//
// Mirrors utils_graph_learning.one_hot_encoder.forward (utils_graph_learning.py:170-187), the DiscreteEmbedding
// ('one_hot_encoder') the reference applies to `data.identifiers` before every GSN layer
// (models_graph_classification.py:222): column c of the int64 input becomes n_classes[c] floats with a single 1.
// HBM-bound: 8*C bytes in, 4*sum(n_classes) bytes out per row.  With `clamp` values above the last class are mapped to
// the last class (synthetic benchmarks have no dataset-level category table; the reference recodes counts to dense
// category indices on the host first, utils_encoding.py:37-59).
#include <hip/hip_runtime.h>

#include "gsn_internal.h"

namespace gsn {

constexpr int OH_MAX_COLS = 64;

struct OneHotArgs {
    int64_t m_rows;
    int n_cols, width, clamp;
    int cls_ptr[OH_MAX_COLS + 1];  // prefix sums of n_classes
    const int64_t *values;
    float *out;
};

__global__ __launch_bounds__(256) void one_hot_kernel(OneHotArgs a) {
    __shared__ short col_of[1024], cls_of[1024];
    for (int w = threadIdx.x; w < a.width; w += 256) {
        int c = 0;
        while (c + 1 < a.n_cols && a.cls_ptr[c + 1] <= w) ++c;
        col_of[w] = (short)c;
        cls_of[w] = (short)(w - a.cls_ptr[c]);
    }
    __syncthreads();
    const int64_t total = a.m_rows * a.width;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / a.width;
        const int w = (int)(i - row * a.width);
        const int c = col_of[w];
        int64_t v = a.values[row * a.n_cols + c];
        const int ncls = a.cls_ptr[c + 1] - a.cls_ptr[c];
        if (a.clamp) v = v < 0 ? 0 : (v >= ncls ? ncls - 1 : v);
        a.out[i] = (v == cls_of[w]) ? 1.f : 0.f;
    }
}

}  // namespace gsn

using namespace gsn;

extern "C" int gsn_one_hot_hip(int64_t m_rows, int n_cols, const int64_t *values, const int32_t *n_classes, int clamp,
                               float *out, void *stream) {
    if (n_cols < 1 || n_cols > OH_MAX_COLS || !values || !n_classes || !out)
        return set_error(GSN_E_INVALID, "gsn_one_hot_hip: need 1..%d columns and non-null pointers", OH_MAX_COLS);
    OneHotArgs a{};
    a.m_rows = m_rows; a.n_cols = n_cols; a.clamp = clamp; a.values = values; a.out = out;
    a.cls_ptr[0] = 0;
    for (int c = 0; c < n_cols; ++c) {
        if (n_classes[c] < 1) return set_error(GSN_E_INVALID, "gsn_one_hot_hip: n_classes[%d] < 1", c);
        a.cls_ptr[c + 1] = a.cls_ptr[c] + n_classes[c];
    }
    a.width = a.cls_ptr[n_cols];
    if (a.width > 1024) return set_error(GSN_E_UNSUPPORTED, "gsn_one_hot_hip: encoded width %d > 1024", a.width);
    if (m_rows <= 0) return GSN_OK;
    int64_t blocks = (m_rows * a.width + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(one_hot_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "one_hot_kernel: %s", hipGetErrorString(e));
    return GSN_OK;
}
