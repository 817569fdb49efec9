// Identifier encoding on the device: multi-hot encoding of integer count columns.
//
// Mirrors utils_graph_learning.one_hot_encoder.forward (utils_graph_learning.py:170-187), the DiscreteEmbedding
// ('one_hot_encoder') the reference applies to `data.identifiers` before every GSN layer
// (models_graph_classification.py:222): column c of the int64 input becomes n_classes[c] floats with a single 1.
// HBM-bound: 8*C bytes in, 4*sum(n_classes) bytes out per row.  With `clamp` values above the last class are mapped to
// the last class (synthetic benchmarks have no dataset-level category table; the reference recodes counts to dense
// category indices on the host first, utils_encoding.py:37-59).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

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

// Row-per-thread variant for the common narrow case (<= 4 identifier columns, encoded width a multiple of 4 and <= 64, output
// 16-byte aligned): the row's values are read once, the floats leave as float4 stores (the element-per-thread kernel above
// pays a 64-bit division, two LDS look-ups and an 8-byte load per output float).
template <int NC>
__global__ __launch_bounds__(256) void one_hot_rows_kernel(OneHotArgs a) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= a.m_rows) return;
    int hot[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        int64_t v = a.values[row * NC + c];
        const int ncls = a.cls_ptr[c + 1] - a.cls_ptr[c];
        if (a.clamp) v = v < 0 ? 0 : (v >= ncls ? ncls - 1 : v);
        hot[c] = (v >= 0 && v < ncls) ? a.cls_ptr[c] + (int)v : -1;      // position of the 1 in the encoded row, or none
    }
    float4 *dst = reinterpret_cast<float4 *>(a.out + row * a.width);
    for (int k = 0; k < a.width; k += 4) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int d = hot[c] - k;
            o.x = d == 0 ? 1.f : o.x; o.y = d == 1 ? 1.f : o.y; o.z = d == 2 ? 1.f : o.z; o.w = d == 3 ? 1.f : o.w;
        }
        dst[k >> 2] = o;
    }
}

// The same encoding written straight into an exact fp16 row pack (include/gsn_abi.h, HP-2 packs; csrc/layer_rp.hip reads them): one thread
// per row, the 1s of the row as a bit mask over the pack's columns, the segment [col0, col0 + width) leaves as 32-bit words (16-bit
// stores at an odd boundary); `one_col` >= 0: that column = 1.0 (the node pack's bias column).  Columns outside are not touched.
struct OneHotPackArgs {
    int64_t m_rows;
    int n_cols, width, clamp;
    int cls_ptr[OH_MAX_COLS + 1];
    const int64_t *values;
    uint16_t *dst;
    int stride, col0, one_col;
    int32_t *status;            // OR 1: a code outside its column's classes (no clamp); may be null
    int q_first, q_count;       // the 8-column groups of a row this call writes: all of them (whole_row), or those its segment / 1.0 column touch
    int whole_row;              // the call owns every column of the pack (col0 == 0 and a 1.0 column: a node pack): full 16-byte stores
};

__global__ __launch_bounds__(256) void one_hot_pack16_kernel(OneHotPackArgs a) {
    // one thread per (row, group of 8 pack columns = 16 bytes): a group that lies inside the segment -- or anywhere in a pack this call owns
    // entirely (a node pack: its other columns are zero by contract) -- leaves as ONE 16-byte store, neighbouring threads write
    // neighbouring groups of a row
    // (only the groups the call writes get a thread: a 4-column segment inside a 32-column pack is ONE group per row, not four)
    const int Q = a.q_count;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.m_rows * Q) return;
    const int64_t row = i / Q;
    const int q = a.q_first + (int)(i - row * Q);
    unsigned long long hot = 0;
    for (int c = 0; c < a.n_cols; ++c) {
        int64_t v = a.values[row * a.n_cols + c];
        const int ncls = a.cls_ptr[c + 1] - a.cls_ptr[c];
        if (a.clamp) v = v < 0 ? 0 : (v >= ncls ? ncls - 1 : v);
        if (v >= 0 && v < ncls) hot |= 1ull << (a.col0 + a.cls_ptr[c] + (int)v);
        else if (a.status && q == a.q_first) atomicOr(a.status, 1);
    }
    if (a.one_col >= 0) hot |= 1ull << a.one_col;
    uint16_t *d = a.dst + row * a.stride;
    const int g0 = 8 * q, seg_lo = a.col0, seg_hi = a.col0 + a.width;
    const int lo = g0 > seg_lo ? g0 : seg_lo, hi = g0 + 8 < seg_hi ? g0 + 8 : seg_hi;
    auto word = [&](int k) { return ((hot >> k) & 1 ? 0x3C00u : 0u) | ((hot >> (k + 1)) & 1 ? 0x3C000000u : 0u); };
    if (a.whole_row || (lo == g0 && hi == g0 + 8)) {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<u4 *>(d + g0) = u4{word(g0), word(g0 + 2), word(g0 + 4), word(g0 + 6)};
        return;
    }
    int k = lo;
    if (k < hi && (k & 1)) { d[k] = (hot >> k) & 1 ? 0x3C00 : 0; ++k; }
    for (; k + 2 <= hi; k += 2) *reinterpret_cast<unsigned *>(d + k) = word(k);
    if (k < hi) d[k] = (hot >> k) & 1 ? 0x3C00 : 0;
    if (a.one_col >= g0 && a.one_col < g0 + 8) d[a.one_col] = 0x3C00;
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
    if (n_cols <= 4 && (a.width & 3) == 0 && a.width <= 64 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        const dim3 grid((unsigned)((m_rows + 255) / 256)), block(256);
        hipStream_t st = reinterpret_cast<hipStream_t>(stream);
        if (n_cols == 1) hipLaunchKernelGGL(one_hot_rows_kernel<1>, grid, block, 0, st, a);
        else if (n_cols == 2) hipLaunchKernelGGL(one_hot_rows_kernel<2>, grid, block, 0, st, a);
        else if (n_cols == 3) hipLaunchKernelGGL(one_hot_rows_kernel<3>, grid, block, 0, st, a);
        else hipLaunchKernelGGL(one_hot_rows_kernel<4>, grid, block, 0, st, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return set_error(GSN_E_HIP, "one_hot_rows_kernel: %s", hipGetErrorString(e));
        return GSN_OK;
    }
    int64_t blocks = (m_rows * a.width + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(one_hot_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "one_hot_kernel: %s", hipGetErrorString(e));
    return GSN_OK;
}

extern "C" int gsn_one_hot_pack16_hip(int64_t m_rows, int n_cols, const int64_t *values, const int32_t *n_classes, int clamp,
                                      uint16_t *dst, int64_t dst_stride, int64_t col0, int64_t one_col, int32_t *status, void *stream) {
    if (n_cols < 1 || n_cols > OH_MAX_COLS || !values || !n_classes || !dst || dst_stride < 1 || dst_stride > 64 || (dst_stride & 1) || col0 < 0)
        return set_error(GSN_E_INVALID, "gsn_one_hot_pack16_hip: need 1..%d columns, non-null pointers and an even pack width <= 64", OH_MAX_COLS);
    OneHotPackArgs a{};
    a.m_rows = m_rows; a.n_cols = n_cols; a.clamp = clamp; a.values = values; a.dst = dst;
    a.stride = (int)dst_stride; a.col0 = (int)col0; a.one_col = one_col < 0 ? -1 : (int)one_col; a.status = status;
    a.whole_row = (col0 == 0 && one_col >= 0 && (dst_stride & 7) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) ? 1 : 0;
    a.cls_ptr[0] = 0;
    for (int c = 0; c < n_cols; ++c) {
        if (n_classes[c] < 1) return set_error(GSN_E_INVALID, "gsn_one_hot_pack16_hip: n_classes[%d] < 1", c);
        a.cls_ptr[c + 1] = a.cls_ptr[c] + n_classes[c];
    }
    a.width = a.cls_ptr[n_cols];
    if (col0 + a.width > dst_stride || one_col >= dst_stride || (one_col >= col0 && one_col < col0 + a.width))
        return set_error(GSN_E_INVALID, "gsn_one_hot_pack16_hip: %d encoded columns at %lld (+ the 1.0 column %lld) do not fit a %lld-column pack", a.width,
                         (long long)col0, (long long)one_col, (long long)dst_stride);
    if (m_rows <= 0) return GSN_OK;
    if ((dst_stride & 7) || (reinterpret_cast<uintptr_t>(dst) & 15)) return set_error(GSN_E_INVALID, "gsn_one_hot_pack16_hip: pack rows must be multiples of 16 bytes, 16-byte aligned");
    a.q_first = 0; a.q_count = (int)(dst_stride >> 3);
    if (!a.whole_row) {
        int lo = (int)col0, hi = (int)col0 + a.width - 1;
        if (a.one_col >= 0) { lo = a.one_col < lo ? a.one_col : lo; hi = a.one_col > hi ? a.one_col : hi; }
        a.q_first = lo >> 3; a.q_count = (hi >> 3) - a.q_first + 1;
    }
    hipLaunchKernelGGL(one_hot_pack16_kernel, dim3((unsigned)((m_rows * a.q_count + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "one_hot_pack16_kernel: %s", hipGetErrorString(e));
    return GSN_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Dataset-level dense recoding of integer columns (utils_encoding.one_hot_unique, utils_encoding.py:37-59): every column
// of the concatenated identifier matrix is replaced by the rank of its value among the column's distinct values
// (np.unique(..., return_inverse=True)).  Counts are small non-negative integers, so instead of a sort the values index
// a presence table:  range pass (min/max per column) -> mark -> exclusive scan of the table -> gather.
// HBM-bound and run once per dataset.
// ---------------------------------------------------------------------------------------------------------------------
namespace gsn {

__global__ __launch_bounds__(256) void column_range_init_kernel(int n_cols, int64_t *mn, int64_t *mx) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < n_cols) {
        mn[c] = INT64_MAX;
        mx[c] = INT64_MIN;
    }
}

// grid (blocks_x, n_cols): every block reduces a strided slice of one column, one atomic pair per block
__global__ __launch_bounds__(256) void column_range_kernel(int64_t m_rows, int n_cols, const int64_t *values, int64_t *mn,
                                                           int64_t *mx) {
    const int c = blockIdx.y;
    int64_t lo = INT64_MAX, hi = INT64_MIN;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < m_rows; r += (int64_t)gridDim.x * 256) {
        const int64_t v = values[r * n_cols + c];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
    }
    __shared__ int64_t slo[256], shi[256];
    slo[threadIdx.x] = lo;
    shi[threadIdx.x] = hi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            slo[threadIdx.x] = slo[threadIdx.x + s] < slo[threadIdx.x] ? slo[threadIdx.x + s] : slo[threadIdx.x];
            shi[threadIdx.x] = shi[threadIdx.x + s] > shi[threadIdx.x] ? shi[threadIdx.x + s] : shi[threadIdx.x];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicMin((long long *)&mn[c], (long long)slo[0]);
        atomicMax((long long *)&mx[c], (long long)shi[0]);
    }
}

__global__ __launch_bounds__(256) void rank_mark_kernel(int64_t m_rows, int n_cols, const int64_t *values,
                                                        const int64_t *col_min, const int64_t *col_base, int32_t *table) {
    const int64_t total = m_rows * n_cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % n_cols);
        table[col_base[c] + (values[i] - col_min[c])] = 1;
    }
}

// one workgroup per column: exclusive prefix sum of the 0/1 presence flags in place, number of distinct values to d_out
__global__ __launch_bounds__(1024) void rank_scan_kernel(const int64_t *col_base, int32_t *table, int64_t *d_out) {
    const int c = blockIdx.x;
    int32_t *t = table + col_base[c];
    const int64_t len = col_base[c + 1] - col_base[c];
    __shared__ int32_t wsum[16];
    __shared__ int32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t base = 0; base < len; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const int32_t v = i < len ? t[i] : 0;
        int32_t x = v;                                   // inclusive scan inside the wave
        for (int d = 1; d < 64; d <<= 1) {
            const int32_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int32_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        const int32_t carry = carry_s;
        if (i < len) t[i] = carry + woff + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) d_out[c] = carry_s;
}

__global__ __launch_bounds__(256) void rank_gather_kernel(int64_t m_rows, int n_cols, const int64_t *values,
                                                          const int64_t *col_min, const int64_t *col_base,
                                                          const int32_t *table, int64_t *codes) {
    const int64_t total = m_rows * n_cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % n_cols);
        codes[i] = table[col_base[c] + (values[i] - col_min[c])];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Embedding of several categorical columns (utils_graph_learning.multi_embedding, :134-167; also the shape of ogb's
// AtomEncoder / BondEncoder): out[r] = concat_c T_c[code[r][c]]  or  sum_c T_c[code[r][c]].
// meta (device, int64): [0..C) table base pointers, [C..2C) table row counts.  One thread per output float; codes are
// broadcast within a row, table rows are read coalesced.  status (device int32) is raised to GSN_ST_BAD_INDEX on a code
// outside its table (the reference's nn.Embedding raises IndexError) and the row's values come out NaN.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_fwd_kernel(int64_t m_rows, int n_cols, int d, int concat, const int64_t *codes,
                                                        const int64_t *meta, float *out, int32_t *status) {
    const int width = concat ? n_cols * d : d;
    const int64_t total = m_rows * width;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / width;
        const int w = (int)(i - r * width);
        float acc = 0.f;
        if (concat) {
            const int c = w / d, j = w - c * d;
            const int64_t code = codes[r * n_cols + c];
            if (code < 0 || code >= meta[n_cols + c]) { atomicMax(status, GSN_ST_BAD_INDEX); out[i] = __builtin_nanf(""); continue; }
            acc = reinterpret_cast<const float *>(meta[c])[code * d + j];
        } else {
            // (eight columns at a time: their codes, then their table elements, are in flight together; summed in column order)
            for (int c0 = 0; c0 < n_cols; c0 += 8) {
                int64_t code[8];
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) code[u] = c0 + u < n_cols ? codes[r * n_cols + c0 + u] : 0;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    v[u] = 0.f;
                    if (c0 + u < n_cols) {
                        if (code[u] < 0 || code[u] >= meta[n_cols + c0 + u]) { atomicMax(status, GSN_ST_BAD_INDEX); v[u] = __builtin_nanf(""); }
                        else v[u] = reinterpret_cast<const float *>(meta[c0 + u])[code[u] * d + w];
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (c0 + u < n_cols) acc += v[u];
            }
        }
        out[i] = acc;
    }
}

constexpr int EMB_DCH = 64;           // columns of the embedding handled by one workgroup
constexpr int EMB_ROWS = 2048;        // input rows per workgroup at most (the launcher shrinks it until the grid fills the chip: EmbArgs::rows_per_wg)

// adjoint for tables too large for the LDS path below: gT_c[code[r][c]] += g_out[r][...] with global fp32 atomics
// (collisions are rare on large tables); meta holds the GRADIENT table pointers
__global__ __launch_bounds__(256) void embed_bwd_kernel(int64_t m_rows, int n_cols, int d, int concat, const int64_t *codes,
                                                        const int64_t *meta, const float *gout, int only_col) {
    const int64_t total = m_rows * d;
    const int gw = concat ? n_cols * d : d;
    const int c = only_col;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / d;
        const int j = (int)(i - r * d);
        const int64_t code = codes[r * n_cols + c];
        if (code < 0 || code >= meta[n_cols + c]) continue;
        atomicAdd(reinterpret_cast<float *>(meta[c]) + code * d + j, gout[r * gw + (concat ? c * d + j : j)]);
    }
}

// y = act((h - mean) * scale + shift) per column: the BatchNorm + activation of a stage whose pre-BN rows were
// materialised together with their batch statistics (train mode, shapes outside the fused chain).  In place allowed.
__global__ __launch_bounds__(256) void bn_act_kernel(int64_t total, int n_cols, const float *h, const float *mean,
                                                     const float *scale, const float *shift, int act, float *out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % n_cols);
        float y = (h[i] - (mean ? mean[c] : 0.f)) * (scale ? scale[c] : 1.f) + (shift ? shift[c] : 0.f);
        switch (act) {
            case 1: y = y > 0.f ? y : 0.f; break;
            case 2: y = y > 0.f ? y : expm1f(y); break;
            case 3: y = tanhf(y); break;
            default: break;
        }
        out[i] = y;
    }
}

// Small-table path of both directions (sum of table rows over all columns <= EMB_LDS_SUM_ROWS): a workgroup owns
// (row chunk, 64-wide slice of the embedding) for ALL code columns: the table slices [sum rows][64] live in LDS, a wave
// covers the 64 slice columns of one input row, eight rows per step so that their loads are in flight together.
//   forward : out[r] = sum_c / concat_c  T_c[code_c][slice]                (tables copied to LDS once per workgroup)
//   backward: gT_c[code_c][slice] += g[r]  in LDS (a wave's lanes never collide), one global atomic per entry at the end
constexpr int EMB_LDS_SUM_ROWS = 448;      // 448 x 64 floats = 112 KiB
constexpr int EMB_MAXC = 16;

struct EmbArgs {
    int64_t m_rows;
    int n_cols, d, concat;
    const int64_t *codes;
    const int64_t *meta;          // table pointers, then row counts (device)
    int row_off[EMB_MAXC + 1];    // prefix sums of the row counts (host copy)
    const float *gout;            // backward
    float *out;                   // forward
    int32_t *status;              // forward
    int rows_per_wg;              // input rows per workgroup (multiple of 32, <= EMB_ROWS)
    float *gflat;                 // backward, flat variant: the gradient tables live in ONE allocation, table c at gflat + goff[c]
    int64_t goff[EMB_MAXC];       // (no device pointer array: nothing to copy per step, nothing for a graph replay to re-read)
    int vec4;                     // forward, whole rows: d a multiple of 4 and out 16-byte aligned -> 16-byte stores
    int pipe;                     // forward, vec4: codes of the next 8-row group requested ahead, one coalesced load per group (GSN_EMBED_PIPE=0: off)
};

// NSUB: 64-column sub-slices per workgroup (slice = 64 NSUB columns of the embedding).  When the tables are small enough a workgroup
// takes WHOLE rows (NSUB = ceil(d / 64)): consecutive rows of the output are then written as one contiguous run instead of five
// 256-byte pieces per row by five workgroups at different times (d = 300: pieces that straddle cache lines).
template <bool BWD, int NSUB>
__global__ __launch_bounds__(256) void embed_lds_kernel(EmbArgs a) {
    constexpr int DCH = EMB_DCH * NSUB;
    extern __shared__ __attribute__((aligned(16))) float tab[];      // [sum rows][DCH]  (BWD: one copy per wave)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwv = blockDim.x >> 6;
    const int j0 = blockIdx.y * DCH;
    const int total_rows = a.row_off[a.n_cols];
    // BWD: every wave accumulates into its OWN copy with plain read-modify-write (its lanes own distinct columns and a
    // wave executes in order).  LDS float atomics were measured at ~0.7 us per instruction here.
    float *mine = tab + (BWD ? wave * total_rows * DCH : 0);
    if (BWD) {
        for (int i = threadIdx.x; i < nwv * total_rows * DCH; i += blockDim.x) tab[i] = 0.f;
    } else if (NSUB > 1 && a.pipe) {
        // whole rows, d a multiple of four (a.pipe implies both): the tables' pointers and row offsets go to LDS first (ONE round trip for all
        // columns instead of one per column in front of that column's copy), then one flat loop of 16-byte copies over every (table row, chunk)
        __shared__ const float *tps[EMB_MAXC];
        __shared__ int roff[EMB_MAXC + 1];
        if (threadIdx.x <= (unsigned)a.n_cols) roff[threadIdx.x] = a.row_off[threadIdx.x];
        if (threadIdx.x < (unsigned)a.n_cols) tps[threadIdx.x] = reinterpret_cast<const float *>(a.meta[threadIdx.x]);
        __syncthreads();
        constexpr int CH = DCH / 4;
        const int d4 = a.d >> 2;
        for (int i = threadIdx.x; i < total_rows * CH; i += blockDim.x) {
            const int trow = i / CH, ch = i - trow * CH;
            int c = 0;
            for (int q = 1; q < a.n_cols; ++q) c = trow >= roff[q] ? q : c;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ch < d4) {
                const float *src = tps[c] + (int64_t)(trow - roff[c]) * a.d + 4 * ch;
                if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) v = *reinterpret_cast<const float4 *>(src);
                else v = make_float4(src[0], src[1], src[2], src[3]);        // (a table that is a view at an odd offset of a flat parameter buffer)
            }
            *reinterpret_cast<float4 *>(tab + trow * DCH + 4 * ch) = v;
        }
    } else {
        for (int c = 0; c < a.n_cols; ++c) {
            const float *t = reinterpret_cast<const float *>(a.meta[c]);
            const int rows_c = a.row_off[c + 1] - a.row_off[c];
            for (int i = threadIdx.x; i < rows_c * DCH; i += blockDim.x) {
                const int row = i / DCH, jj = j0 + (i - row * DCH);
                tab[(a.row_off[c] + row) * DCH + (i - row * DCH)] = jj < a.d ? t[(int64_t)row * a.d + jj] : 0.f;
            }
        }
    }
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_wg;
    const int64_t r1 = r0 + a.rows_per_wg < a.m_rows ? r0 + a.rows_per_wg : a.m_rows;
    const int gw = a.concat ? a.n_cols * a.d : a.d;
    if (!BWD && NSUB > 1 && a.vec4 && a.pipe) {
        // r05: whole rows, 16-byte stores (below), with the codes of an 8-row group read by ONE coalesced request (slot s = u C + c on lane s, and on
        // lane s - 64 as its second value: C <= 16) one group AHEAD, and handed to the wave with v_readlane.  The loop below waits for eight
        // broadcast loads per code column -- C dependent round trips per 8 rows: 4 x 7 of them per workgroup with both edge encoders of an ogb layer
        // in one call, which is what the 155 us per 214 500 x 300 rows were made of (the r04 lane-distributed variant had no prefetch and 3-4 columns).
        constexpr int NV = (DCH / 4 + 63) / 64;
        const int C = a.n_cols;
        auto load_codes = [&](int64_t rb, int &c0, int &c1) {
            c0 = c1 = -1;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int sl = lane + 64 * k;
                if (sl < 8 * C) {
                    const int u = sl / C, c = sl - u * C;
                    const int64_t row = rb + u < r1 ? rb + u : r1 - 1;
                    const int64_t v = a.codes[row * C + c];
                    const int lo = a.row_off[c], rows_c = a.row_off[c + 1] - lo;
                    const bool ok = v >= 0 && v < rows_c;
                    if (!ok) atomicMax(a.status, GSN_ST_BAD_INDEX);
                    const int t = ok ? lo + (int)v : -1;          // row of the concatenated LDS table, or none
                    if (k == 0) c0 = t; else c1 = t;
                }
            }
        };
        int cur0 = -1, cur1 = -1;
        int64_t rb = r0 + wave * 8;
        if (rb < r1) load_codes(rb, cur0, cur1);
        for (; rb < r1; rb += 8 * nwv) {
            int nx0 = -1, nx1 = -1;
            if (rb + 8 * nwv < r1) load_codes(rb + 8 * nwv, nx0, nx1);
            float4 acc[8][NV];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int pv = 0; pv < NV; ++pv) acc[u][pv] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int c = 0; c < C; ++c) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int sl = u * C + c;                       // (uniform)
                    const int t = sl < 64 ? __builtin_amdgcn_readlane(cur0, sl) : __builtin_amdgcn_readlane(cur1, sl - 64);
                    const bool ok = t >= 0;
                    const float *trow = tab + (ok ? t : 0) * DCH;
#pragma unroll
                    for (int pv = 0; pv < NV; ++pv) {
                        const int j = 4 * (64 * pv + lane);
                        if (j < DCH) {
                            float4 v = *reinterpret_cast<const float4 *>(trow + j);
                            if (!ok) v = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
                            acc[u][pv].x += v.x; acc[u][pv].y += v.y; acc[u][pv].z += v.z; acc[u][pv].w += v.w;
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int pv = 0; pv < NV; ++pv) {
                    const int j = 4 * (64 * pv + lane);
                    if (rb + u < r1 && j < a.d) *reinterpret_cast<float4 *>(a.out + (rb + u) * gw + j) = acc[u][pv];
                }
            cur0 = nx0; cur1 = nx1;
        }
    } else
    if (j0 + lane < a.d) {
        for (int64_t rb = r0 + wave * 8; rb < r1; rb += 8 * nwv) {
            if (BWD && !a.concat) {
                float g[8][NSUB];
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int sb = 0; sb < NSUB; ++sb) {
                        const int j = j0 + EMB_DCH * sb + lane;
                        g[u][sb] = j < a.d ? a.gout[(rb + u < r1 ? rb + u : r1 - 1) * gw + j] : 0.f;
                    }
                for (int c = 0; c < a.n_cols; ++c) {
                    const int rows_c = a.row_off[c + 1] - a.row_off[c];
                    int64_t code[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) code[u] = a.codes[(rb + u < r1 ? rb + u : r1 - 1) * a.n_cols + c];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (rb + u < r1 && code[u] >= 0 && code[u] < rows_c) {
#pragma unroll
                            for (int sb = 0; sb < NSUB; ++sb) mine[(a.row_off[c] + (int)code[u]) * DCH + EMB_DCH * sb + lane] += g[u][sb];
                        }
                }
            } else if (BWD) {
                for (int c = 0; c < a.n_cols; ++c) {
                    const int rows_c = a.row_off[c + 1] - a.row_off[c];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (rb + u >= r1) break;
                        const int64_t code = a.codes[(rb + u) * a.n_cols + c];
                        if (code >= 0 && code < rows_c) {
#pragma unroll
                            for (int sb = 0; sb < NSUB; ++sb) {
                                const int j = j0 + EMB_DCH * sb + lane;
                                if (j < a.d) mine[(a.row_off[c] + (int)code) * DCH + EMB_DCH * sb + lane] += a.gout[(rb + u) * gw + c * a.d + j];
                            }
                        }
                    }
                }
            } else if (!BWD && NSUB > 1 && a.vec4) {
                // whole rows, width a multiple of 4, 16-byte aligned output: a lane owns FOUR consecutive columns -- 64 lanes x 16 bytes per
                // store instruction instead of 64 x 4 (a CU retires about one store instruction per ~65 cycles whatever its width: the 4-byte
                // version spent 5 instructions per 1 200-byte row, 130 us per 214 500 rows by that rate alone; measured 115)
                constexpr int NV = (DCH / 4 + 63) / 64;                // passes of 64 float4 over a row (d = 300: 64 + 11 lanes)
                float4 acc[8][NV];
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int pv = 0; pv < NV; ++pv) acc[u][pv] = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int c = 0; c < a.n_cols; ++c) {
                    const int rows_c = a.row_off[c + 1] - a.row_off[c];
                    int64_t code[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) code[u] = a.codes[(rb + u < r1 ? rb + u : r1 - 1) * a.n_cols + c];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const bool ok = code[u] >= 0 && code[u] < rows_c;
                        if (!ok) atomicMax(a.status, GSN_ST_BAD_INDEX);
                        const float *trow = tab + (a.row_off[c] + (ok ? (int)code[u] : 0)) * DCH;
#pragma unroll
                        for (int pv = 0; pv < NV; ++pv) {
                            const int j = 4 * (64 * pv + lane);
                            if (j < DCH) {
                                float4 v = *reinterpret_cast<const float4 *>(trow + j);
                                if (!ok) v = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
                                acc[u][pv].x += v.x; acc[u][pv].y += v.y; acc[u][pv].z += v.z; acc[u][pv].w += v.w;
                            }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int pv = 0; pv < NV; ++pv) {
                        const int j = 4 * (64 * pv + lane);
                        if (rb + u < r1 && j < a.d) *reinterpret_cast<float4 *>(a.out + (rb + u) * gw + j) = acc[u][pv];
                    }
            } else {
                float acc[8][NSUB];
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int sb = 0; sb < NSUB; ++sb) acc[u][sb] = 0.f;
                for (int c = 0; c < a.n_cols; ++c) {
                    const int rows_c = a.row_off[c + 1] - a.row_off[c];
                    int64_t code[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) code[u] = a.codes[(rb + u < r1 ? rb + u : r1 - 1) * a.n_cols + c];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const bool ok = code[u] >= 0 && code[u] < rows_c;
                        if (!ok) atomicMax(a.status, GSN_ST_BAD_INDEX);
#pragma unroll
                        for (int sb = 0; sb < NSUB; ++sb) {
                            const int j = j0 + EMB_DCH * sb + lane;
                            const float v = ok ? tab[(a.row_off[c] + (int)code[u]) * DCH + EMB_DCH * sb + lane] : __builtin_nanf("");
                            if (a.concat) { if (rb + u < r1 && j < a.d) a.out[(rb + u) * gw + c * a.d + j] = v; }
                            else acc[u][sb] += v;
                        }
                    }
                }
                if (!a.concat) {
#pragma unroll
                    for (int u = 0; u < 8; ++u)
#pragma unroll
                        for (int sb = 0; sb < NSUB; ++sb) {
                            const int j = j0 + EMB_DCH * sb + lane;
                            if (rb + u < r1 && j < a.d) a.out[(rb + u) * gw + j] = acc[u][sb];
                        }
                }
            }
        }
    }
    if (BWD) {
        __syncthreads();
        for (int c = 0; c < a.n_cols; ++c) {
            float *t = a.gflat ? a.gflat + a.goff[c] : reinterpret_cast<float *>(a.meta[c]);
            const int rows_c = a.row_off[c + 1] - a.row_off[c];
            for (int i = threadIdx.x; i < rows_c * DCH; i += blockDim.x) {
                const int row = i / DCH, jj = j0 + (i - row * DCH);
                float v = 0.f;
                for (int wv = 0; wv < nwv; ++wv) v += tab[(wv * total_rows + a.row_off[c] + row) * DCH + (i - row * DCH)];
                if (jj < a.d && v != 0.f) atomicAdd(t + (int64_t)row * a.d + jj, v);
            }
        }
    }
}

template <bool BWD, int NSUB>
static int launch_embed_lds_n(EmbArgs &a, int nwv, hipStream_t s) {
    constexpr int DCH = EMB_DCH * NSUB;
    const size_t lds = (size_t)(BWD ? nwv : 1) * a.row_off[a.n_cols] * DCH * sizeof(float);
    static DeviceOnce attr_set;
    const int attr_dev = current_device();
    if (!attr_set.done(attr_dev)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&embed_lds_kernel<BWD, NSUB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                EMB_LDS_SUM_ROWS * EMB_DCH * (int)sizeof(float)) != hipSuccess)
            return set_error(GSN_E_HIP, "embed_lds_kernel: cannot raise the LDS limit");
        attr_set.mark(attr_dev);
    }
    // rows per workgroup: 2048 at most (the table slices are copied to / flushed from LDS once per workgroup), fewer until there are ~8
    // workgroups per CU -- at 2048 a molhiv-sized batch (214 k edge rows, d = 300) gave every CU two workgroups and 0.9 TB/s
    const int64_t n_slices = (a.d + DCH - 1) / DCH;
    int64_t rpw = (a.m_rows * n_slices + 2047) / 2048;
    rpw = (rpw + 31) / 32 * 32;
    rpw = rpw < 128 ? 128 : (rpw > EMB_ROWS ? EMB_ROWS : rpw);
    a.rows_per_wg = (int)rpw;
    const dim3 grid((unsigned)((a.m_rows + rpw - 1) / rpw), (unsigned)n_slices);
    static const bool no_vec4 = getenv("GSN_EMBED_NOVEC4") != nullptr;        // (A/B)
    a.vec4 = (!BWD && NSUB > 1 && !a.concat && a.d % 4 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 && !no_vec4) ? 1 : 0;
    static const bool no_pipe = [] { const char *e = getenv("GSN_EMBED_PIPE"); return e && atoi(e) == 0; }();
    a.pipe = (a.vec4 && a.n_cols <= 16 && !no_pipe) ? 1 : 0;
    hipLaunchKernelGGL((embed_lds_kernel<BWD, NSUB>), grid, dim3(64 * nwv), lds, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "embed_lds_kernel: %s", hipGetErrorString(e));
    return GSN_OK;
}

template <bool BWD>
static int launch_embed_lds(EmbArgs &a, const int64_t *table_rows, hipStream_t s) {
    a.row_off[0] = 0;
    for (int c = 0; c < a.n_cols; ++c) a.row_off[c + 1] = a.row_off[c] + (int)table_rows[c];
    const int rows = a.row_off[a.n_cols];
    // backward keeps one table copy per wave: 4 waves while that fits, else a single wave per workgroup
    // whole rows per workgroup (summed embeddings of width <= 320) while the tables (x the wave copies of the backward pass) stay under
    // 48 KiB of LDS: three workgroups and more per CU
    const int nsub = (a.d + EMB_DCH - 1) / EMB_DCH;
    const int64_t one_copy = (int64_t)rows * nsub * EMB_DCH * 4;
    int wv = 4;                                                  // (backward: four wave copies, or two, of the tables)
    if (BWD && 4 * one_copy > 48 * 1024) wv = 2;
    const bool whole = !a.concat && nsub >= 2 && nsub <= 5 && (BWD ? wv : 1) * one_copy <= 48 * 1024;
    if (whole) {
        switch (nsub) {
            case 2: return launch_embed_lds_n<BWD, 2>(a, wv, s);
            case 3: return launch_embed_lds_n<BWD, 3>(a, wv, s);
            case 4: return launch_embed_lds_n<BWD, 4>(a, wv, s);
            default: return launch_embed_lds_n<BWD, 5>(a, wv, s);
        }
    }
    const int nwv = (!BWD || 4 * rows <= EMB_LDS_SUM_ROWS) ? 4 : 1;
    return launch_embed_lds_n<BWD, 1>(a, nwv, s);
}

static bool embed_lds_fits(int n_cols, const int64_t *table_rows) {
    if (!table_rows || n_cols > EMB_MAXC) return false;
    int64_t sum = 0;
    for (int c = 0; c < n_cols; ++c) sum += table_rows[c];
    return sum <= EMB_LDS_SUM_ROWS;
}

static int grid_for(int64_t items) {
    int64_t b = (items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 256 * 16 ? 256 * 16 : b));
}

}  // namespace gsn

#define GSN_LAUNCH_CHECK(name)                                                                  \
    do {                                                                                        \
        hipError_t e_ = hipGetLastError();                                                      \
        if (e_ != hipSuccess) return set_error(GSN_E_HIP, name ": %s", hipGetErrorString(e_)); \
    } while (0)

extern "C" int gsn_column_range_hip(int64_t m_rows, int n_cols, const int64_t *values, int64_t *col_min, int64_t *col_max,
                                    void *stream) {
    if (n_cols < 1 || n_cols > 65535 || !col_min || !col_max || (m_rows > 0 && !values))
        return set_error(GSN_E_INVALID, "gsn_column_range_hip: bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(column_range_init_kernel, dim3((n_cols + 255) / 256), dim3(256), 0, s, n_cols, col_min, col_max);
    if (m_rows > 0) {
        int bx = grid_for(m_rows);
        if (bx > 1024) bx = 1024;
        hipLaunchKernelGGL(column_range_kernel, dim3(bx, n_cols), dim3(256), 0, s, m_rows, n_cols, values, col_min, col_max);
    }
    GSN_LAUNCH_CHECK("column_range_kernel");
    return GSN_OK;
}

extern "C" int gsn_column_ranks_hip(int64_t m_rows, int n_cols, const int64_t *values, const int64_t *col_min,
                                    const int64_t *col_base, int64_t table_elems, int32_t *table, int64_t *codes,
                                    int64_t *n_distinct, void *stream) {
    if (n_cols < 1 || n_cols > 65535 || !col_min || !col_base || !n_distinct || table_elems < 0 ||
        (table_elems > 0 && !table) || (m_rows > 0 && (!values || !codes)))
        return set_error(GSN_E_INVALID, "gsn_column_ranks_hip: bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (table_elems > 0 && hipMemsetAsync(table, 0, (size_t)table_elems * sizeof(int32_t), s) != hipSuccess)
        return set_error(GSN_E_HIP, "gsn_column_ranks_hip: memset failed");
    if (m_rows > 0)
        hipLaunchKernelGGL(rank_mark_kernel, dim3(grid_for(m_rows * n_cols)), dim3(256), 0, s, m_rows, n_cols, values, col_min,
                           col_base, table);
    hipLaunchKernelGGL(rank_scan_kernel, dim3(n_cols), dim3(1024), 0, s, col_base, table, n_distinct);
    if (m_rows > 0)
        hipLaunchKernelGGL(rank_gather_kernel, dim3(grid_for(m_rows * n_cols)), dim3(256), 0, s, m_rows, n_cols, values,
                           col_min, col_base, table, codes);
    GSN_LAUNCH_CHECK("rank kernels");
    return GSN_OK;
}

extern "C" int gsn_embed_fwd_hip(int64_t m_rows, int n_cols, int d, int concat, const int64_t *codes, const int64_t *meta,
                                 const int64_t *table_rows, float *out, int32_t *status, void *stream) {
    if (n_cols < 1 || d < 1 || !meta || !status || (m_rows > 0 && (!codes || !out)))
        return set_error(GSN_E_INVALID, "gsn_embed_fwd_hip: bad arguments");
    if (m_rows <= 0) return GSN_OK;
    // (few rows -- the reference's batch sizes: 800 .. 6 000 rows -- gather straight from the L2-resident tables: the LDS kernel first copies
    //  every table slice into every workgroup, ~40 us per call whatever the row count)
    static const int64_t lds_min_rows = [] { const char *e = getenv("GSN_EMBED_LDS_MIN_ROWS"); return e ? atoll(e) : (int64_t)8192; }();
    if (m_rows > lds_min_rows && embed_lds_fits(n_cols, table_rows)) {
        EmbArgs a{};
        a.m_rows = m_rows; a.n_cols = n_cols; a.d = d; a.concat = concat; a.codes = codes; a.meta = meta; a.out = out; a.status = status;
        return launch_embed_lds<false>(a, table_rows, reinterpret_cast<hipStream_t>(stream));
    }
    const int64_t total = m_rows * (concat ? (int64_t)n_cols * d : d);
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), m_rows,
                       n_cols, d, concat, codes, meta, out, status);
    GSN_LAUNCH_CHECK("embed_fwd_kernel");
    return GSN_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Summed-embedding backward on the matrix pipe (r03).  gT_c[r][:] = sum over the rows m with code[m][c] = r of g[m][:]  is the product
// OneHot(codes)^T g: the one-hot operand is made in registers from the codes (1.0 is exact in bf16: ONE plane), g is split exactly into
// three bf16 planes while it is staged (the weight-gradient kernel's skeleton, backward.hip: contraction over the rows, operands
// transposed for free in the staging registers), three products per tile pair, fp32 accumulation.  The concatenated table rows of all
// code columns are the output rows; tiles of 128 table rows x 128 embedding columns, row slabs per workgroup, float atomics at the end.
// Why: the LDS-accumulating kernel above keeps one table copy per wave -- tables of more than ~100 rows leave room for ONE wave per
// workgroup (identifier codes of the ogb model: 298 us per call at 214 k x 300); this one reads g once per 128 table rows at memory speed.
// ---------------------------------------------------------------------------------------------------------------------
namespace gsn {
typedef float emb_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned emb_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 emb_bf16x8 __attribute__((ext_vector_type(8)));
constexpr int EMBM_T = 128;

struct EmbMArgs {
    int64_t m_rows, rows_per_wg;
    int n_cols, d, rtot, tn, tk;
    const int64_t *codes;
    const int64_t *meta;           // gradient table pointers (device)
    int row_off[EMB_MAXC + 1];
    const float *gout;
    float *gflat;                  // flat variant (see EmbArgs): table c at gflat + goff[c]
    int64_t goff[EMB_MAXC];
};

__device__ __forceinline__ void embm_split3(float x, unsigned &h, unsigned &m, unsigned &l) {
    h = __float_as_uint(x);
    const float r1 = x - __uint_as_float(h & 0xffff0000u);
    m = __float_as_uint(r1);
    l = __float_as_uint(r1 - __uint_as_float(m & 0xffff0000u));
}
__device__ __forceinline__ unsigned embm_pack_hi(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

__global__ __launch_bounds__(256) void embed_bwd_mfma_kernel(EmbMArgs a) {
    __shared__ emb_u32x4 tb[2][3][EMBM_T][2];           // [buffer][plane][embedding column][row half]: 8 bf16 = rows 8 h .. 8 h + 7 of the chunk
    __shared__ emb_u32x4 tc[2][EMB_MAXC][2];            // [buffer][code column][row half]: 8 int16 codes of those rows (0xffff = none)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int ntile = a.tn * a.tk;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int tile = seq % ntile;
    const int64_t slab = (int64_t)(seq / ntile) * 8 + xcd;
    const int n0 = (tile / a.tk) * EMBM_T, k0 = (tile % a.tk) * EMBM_T;
    const int64_t r_begin = slab * a.rows_per_wg;
    int64_t r_end = r_begin + a.rows_per_wg;
    if (r_end > a.m_rows) r_end = a.m_rows;
    if (r_begin >= r_end) return;                 // (block-uniform)

    // this lane's two output rows (concatenated table rows): which code column they belong to and the code that selects them
    int my_col[2], my_code[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int i = n0 + wm * 64 + t * 32 + li;
        int c = 0;
        for (int q = 1; q < a.n_cols; ++q) c = i >= a.row_off[q] ? q : c;
        my_col[t] = c;
        my_code[t] = i < a.rtot ? i - a.row_off[c] : 0xfffe;       // (rows past the tables match nothing: codes are < 0xfffe)
    }
    const int sc = tid & 127, sh = tid >> 7;      // staging: embedding column sc, rows 8 sh .. 8 sh + 7 of a 16-row chunk
    const int kb = k0 + sc;
    const bool kb_ok = kb < a.d;
    const float *xb = a.gout + (kb_ok ? kb : 0);
    // codes: thread (column c = tid >> 4, row = tid & 15) for tid < 16 n_cols
    const int cc = tid >> 4, cr = tid & 15;
    const bool c_ok = cc < a.n_cols;
    const int rows_cc = c_ok ? a.row_off[cc + 1] - a.row_off[cc] : 0;

    emb_f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float pb[8];
    int64_t pcode = 0;
    auto fetch = [&](int64_t row0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t r = row0 + 8 * sh + i;
            const int64_t rc = r < r_end ? r : r_begin;
            pb[i] = xb[rc * a.d];
        }
        if (c_ok) { const int64_t r = row0 + cr; pcode = a.codes[(r < r_end ? r : r_begin) * a.n_cols + cc]; }
    };
    auto store = [&](int buf, int64_t row0) {
        unsigned h[8], m[8], l[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool ok = kb_ok && (row0 + 8 * sh + i < r_end);
            embm_split3(ok ? pb[i] : 0.f, h[i], m[i], l[i]);
        }
        emb_u32x4 ph, pm, pl;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ph[i] = embm_pack_hi(h[2 * i], h[2 * i + 1]);
            pm[i] = embm_pack_hi(m[2 * i], m[2 * i + 1]);
            pl[i] = embm_pack_hi(l[2 * i], l[2 * i + 1]);
        }
        tb[buf][0][sc][sh] = ph;
        tb[buf][1][sc][sh] = pm;
        tb[buf][2][sc][sh] = pl;
        if (c_ok) {
            const bool ok = row0 + cr < r_end && pcode >= 0 && pcode < rows_cc;
            reinterpret_cast<unsigned short *>(&tc[buf][cc][0])[cr] = ok ? (unsigned short)pcode : (unsigned short)0xffffu;
        }
    };
    fetch(r_begin);
    store(0, r_begin);
    __syncthreads();
    int buf = 0;
#define EMBM_MF(x, y, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(emb_bf16x8, x), __builtin_bit_cast(emb_bf16x8, y), c, 0, 0, 0)
    for (int64_t row0 = r_begin; row0 < r_end; row0 += 16) {
        const bool has_next = row0 + 16 < r_end;
        if (has_next) fetch(row0 + 16);
        // the one-hot operand of this lane's two output rows: 1.0 (bf16 0x3f80) where the row's code selects it
        emb_u32x4 fa[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const emb_u32x4 cd = tc[buf][my_col[t]][lh];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned lo = cd[q] & 0xffffu, hi = cd[q] >> 16;
                fa[t][q] = ((int)lo == my_code[t] ? 0x3f80u : 0u) | ((int)hi == my_code[t] ? 0x3f800000u : 0u);
            }
        }
        emb_u32x4 fb[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fb[i][pl] = tb[buf][pl][wn * 64 + i * 32 + li][lh];
#pragma unroll
        for (int pl = 2; pl >= 0; --pl) {            // low plane first
            EMBM_MF(fa[0], fb[0][pl], acc[0][0]);
            EMBM_MF(fa[0], fb[1][pl], acc[0][1]);
            EMBM_MF(fa[1], fb[0][pl], acc[1][0]);
            EMBM_MF(fa[1], fb[1][pl], acc[1][1]);
        }
        if (has_next) store(buf ^ 1, row0 + 16);
        __syncthreads();
        buf ^= 1;
    }
#undef EMBM_MF
    // C layout of a 32x32 tile: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5); rows = concatenated table rows
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nrow = n0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (nrow >= a.rtot) continue;
            int c = 0, roff = 0;
            int64_t go = a.goff[0];
            for (int q = 1; q < a.n_cols; ++q) {        // (uniform q: scalar loads of the argument block, no per-lane indexing of it)
                const bool in = nrow >= a.row_off[q];
                c = in ? q : c; roff = in ? a.row_off[q] : roff; go = in ? a.goff[q] : go;
            }
            float *t = (a.gflat ? a.gflat + go : reinterpret_cast<float *>(a.meta[c])) + (int64_t)(nrow - roff) * a.d;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int kcol = k0 + wn * 64 + j * 32 + li;
                const float v = acc[i][j][r];
                if (kcol < a.d && v != 0.f) atomicAdd(t + kcol, v);
            }
        }
}
}  // namespace gsn

namespace gsn {
static bool embed_bwd_mfma_fits(int64_t m_rows, int n_cols, int concat, const int64_t *table_rows, int64_t *rtot_out) {
    static const bool lds_only = [] { const char *e = getenv("GSN_EMBED_BWD_LDS"); return e && e[0] == '1'; }();
    int64_t rtot = 0;
    bool small_tables = n_cols <= EMB_MAXC;
    for (int c = 0; small_tables && c < n_cols; ++c) { rtot += table_rows[c]; small_tables = table_rows[c] > 0 && table_rows[c] < 0xfffe; }
    if (rtot_out) *rtot_out = rtot;
    return !lds_only && !concat && small_tables && rtot <= 4096 && m_rows >= 256;
}
}  // namespace gsn

static int embed_bwd_impl(int64_t m_rows, int n_cols, int d, int concat, const int64_t *codes, const int64_t *grad_meta, float *grad_flat,
                          const int64_t *table_offsets, const int64_t *table_rows, const float *grad_out, void *stream);

extern "C" int gsn_embed_bwd_hip(int64_t m_rows, int n_cols, int d, int concat, const int64_t *codes, const int64_t *grad_meta,
                                 const int64_t *table_rows, const float *grad_out, void *stream) {
    if (!grad_meta) return set_error(GSN_E_INVALID, "gsn_embed_bwd_hip: bad arguments");
    return embed_bwd_impl(m_rows, n_cols, d, concat, codes, grad_meta, nullptr, nullptr, table_rows, grad_out, stream);
}

// 1 when gsn_embed_bwd_flat_hip handles the shape (the kernels that take their table addresses as launch arguments: <= 16 code columns and
// tables that fit LDS slices or the matrix-pipe product), else 0
extern "C" int gsn_embed_bwd_flat_supported(int64_t m_rows, int n_cols, int concat, const int64_t *table_rows) {
    if (n_cols < 1 || n_cols > EMB_MAXC || !table_rows) return 0;
    return (embed_bwd_mfma_fits(m_rows, n_cols, concat, table_rows, nullptr) || embed_lds_fits(n_cols, table_rows)) ? 1 : 0;
}

// The same accumulation with the gradient tables inside ONE caller-zeroed allocation: table c at grad_flat + table_offsets[c] floats
// (table_offsets: HOST array [C]).  The addresses travel as launch arguments -- no device pointer array, hence no host-to-device copy per
// call (eager) and no memcpy node per gradient table set in a captured training step (gsn_amd.graphs).
extern "C" int gsn_embed_bwd_flat_hip(int64_t m_rows, int n_cols, int d, int concat, const int64_t *codes, float *grad_flat,
                                      const int64_t *table_offsets, const int64_t *table_rows, const float *grad_out, void *stream) {
    if (!grad_flat || !table_offsets || !table_rows) return set_error(GSN_E_INVALID, "gsn_embed_bwd_flat_hip: bad arguments");
    if (!gsn_embed_bwd_flat_supported(m_rows > 0 ? m_rows : 1, n_cols, concat, table_rows))
        return set_error(GSN_E_UNSUPPORTED, "gsn_embed_bwd_flat_hip: tables outside the flat variant (ask gsn_embed_bwd_flat_supported)");
    for (int c = 0; c < n_cols; ++c)
        if (table_offsets[c] < 0) return set_error(GSN_E_INVALID, "gsn_embed_bwd_flat_hip: negative table offset");
    return embed_bwd_impl(m_rows, n_cols, d, concat, codes, nullptr, grad_flat, table_offsets, table_rows, grad_out, stream);
}

static int embed_bwd_impl(int64_t m_rows, int n_cols, int d, int concat, const int64_t *codes, const int64_t *grad_meta, float *grad_flat,
                          const int64_t *table_offsets, const int64_t *table_rows, const float *grad_out, void *stream) {
    if (n_cols < 1 || d < 1 || !table_rows || (m_rows > 0 && (!codes || !grad_out)))
        return set_error(GSN_E_INVALID, "gsn_embed_bwd_hip: bad arguments");
    if (m_rows <= 0) return GSN_OK;
    {   // summed embeddings of enough rows: the one-hot product on the matrix pipe (GSN_EMBED_BWD_LDS=1: the LDS-accumulating kernel)
        int64_t rtot = 0;
        if (embed_bwd_mfma_fits(m_rows, n_cols, concat, table_rows, &rtot)) {
            EmbMArgs a{};
            a.m_rows = m_rows; a.n_cols = n_cols; a.d = d; a.rtot = (int)rtot; a.codes = codes; a.meta = grad_meta; a.gout = grad_out;
            a.gflat = grad_flat;
            for (int c = 0; c < n_cols && grad_flat; ++c) a.goff[c] = table_offsets[c];
            a.row_off[0] = 0;
            for (int c = 0; c < n_cols; ++c) a.row_off[c + 1] = a.row_off[c] + (int)table_rows[c];
            a.tn = (int)((rtot + EMBM_T - 1) / EMBM_T); a.tk = (d + EMBM_T - 1) / EMBM_T;
            const int nt = a.tn * a.tk;
            int64_t slabs = (2048 + nt - 1) / nt;
            int64_t rows_per = (m_rows + slabs - 1) / slabs;
            const int64_t min_rows = m_rows >= 65536 ? 256 : 64;      // (small batches: bound by the serial chain of 16-row steps, backward.hip)
            if (rows_per < min_rows) rows_per = min_rows;
            rows_per = (rows_per + 15) / 16 * 16;
            a.rows_per_wg = rows_per;
            slabs = (m_rows + rows_per - 1) / rows_per;
            const int64_t groups = (slabs + 7) / 8;
            hipLaunchKernelGGL(embed_bwd_mfma_kernel, dim3((unsigned)(groups * nt * 8)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
            GSN_LAUNCH_CHECK("embed_bwd_mfma_kernel");
            return GSN_OK;
        }
    }
    if (embed_lds_fits(n_cols, table_rows)) {
        EmbArgs a{};
        a.m_rows = m_rows; a.n_cols = n_cols; a.d = d; a.concat = concat; a.codes = codes; a.meta = grad_meta; a.gout = grad_out;
        a.gflat = grad_flat;
        for (int c = 0; c < n_cols && grad_flat; ++c) a.goff[c] = table_offsets[c];
        return launch_embed_lds<true>(a, table_rows, reinterpret_cast<hipStream_t>(stream));
    }
    if (!grad_meta) return set_error(GSN_E_UNSUPPORTED, "gsn_embed_bwd_flat_hip: tables outside the flat variant");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    for (int c = 0; c < n_cols; ++c)
        hipLaunchKernelGGL(embed_bwd_kernel, dim3(grid_for(m_rows * d)), dim3(256), 0, s, m_rows, n_cols, d, concat, codes, grad_meta,
                           grad_out, c);
    GSN_LAUNCH_CHECK("embed_bwd kernels");
    return GSN_OK;
}

namespace gsn {
// the same pass with a lane per column and four rows' loads in flight per wave (the element-per-thread kernel above pays a 64-bit modulo per
// float and keeps one load in flight: 5.3 TB/s at 105 k x 600).  grid (row blocks, ceil(C / 64)).  Same expression, same values.
__global__ __launch_bounds__(256) void bn_act_cols_kernel(int64_t m_rows, int n_cols, const float *h, const float *mean, const float *scale,
                                                          const float *shift, int act, float *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + lane;
    if (c >= n_cols) return;
    const float mf = mean ? mean[c] : 0.f, sc = scale ? scale[c] : 1.f, sh = shift ? shift[c] : 0.f;
    auto fin = [&](float v) {
        float y = (v - mf) * sc + sh;
        switch (act) {
            case 1: y = y > 0.f ? y : 0.f; break;
            case 2: y = y > 0.f ? y : expm1f(y); break;
            case 3: y = tanhf(y); break;
            default: break;
        }
        return y;
    };
    const int64_t step = (int64_t)gridDim.x * 4;
    int64_t r = (int64_t)blockIdx.x * 4 + wave;
    for (; r + 3 * step < m_rows; r += 4 * step) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = h[(r + u * step) * n_cols + c];
#pragma unroll
        for (int u = 0; u < 4; ++u) out[(r + u * step) * n_cols + c] = fin(v[u]);
    }
    for (; r < m_rows; r += step) out[r * n_cols + c] = fin(h[r * n_cols + c]);
}
}  // namespace gsn

extern "C" int gsn_bn_act_hip(int64_t m_rows, int64_t n_cols, const float *h, const float *mean, const float *scale,
                              const float *shift, int act, float *out, void *stream) {
    if (n_cols < 1 || act < 0 || act > 3 || (m_rows > 0 && (!h || !out))) return set_error(GSN_E_INVALID, "gsn_bn_act_hip: bad arguments");
    if (m_rows <= 0) return GSN_OK;
    if (m_rows >= 64) {
        int64_t bx = (m_rows + 63) / 64;
        bx = bx > 2048 ? 2048 : bx;
        hipLaunchKernelGGL(bn_act_cols_kernel, dim3((unsigned)bx, (unsigned)((n_cols + 63) / 64)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                           m_rows, (int)n_cols, h, mean, scale, shift, act, out);
        GSN_LAUNCH_CHECK("bn_act_cols_kernel");
        return GSN_OK;
    }
    hipLaunchKernelGGL(bn_act_kernel, dim3(grid_for(m_rows * n_cols)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       m_rows * n_cols, (int)n_cols, h, mean, scale, shift, act, out);
    GSN_LAUNCH_CHECK("bn_act_kernel");
    return GSN_OK;
}

// per-column sum and sum of squares of materialised rows, added to stats[2][n_cols] (fp64): a workgroup takes a band of rows, a thread
// the columns tid, tid + 256, ... (consecutive threads read consecutive floats), one fp64 atomic per column and workgroup at the end
namespace gsn {
constexpr int CS_ROWS = 128;
__global__ __launch_bounds__(256) void column_stats_kernel(int64_t m_rows, int n_cols, const float *h, double *stats) {
    const int64_t r0 = (int64_t)blockIdx.x * CS_ROWS;
    const int64_t r1 = r0 + CS_ROWS < m_rows ? r0 + CS_ROWS : m_rows;
    for (int c = threadIdx.x; c < n_cols; c += 256) {
        double s = 0.0, q = 0.0;
        const float *p = h + r0 * n_cols + c;
        for (int64_t r = r0; r < r1; ++r, p += n_cols) {
            const double v = (double)*p;
            s += v; q += v * v;
        }
        atomicAdd(stats + c, s);
        atomicAdd(stats + n_cols + c, q);
    }
}
}  // namespace gsn

extern "C" int gsn_column_stats_hip(int64_t m_rows, int64_t n_cols, const float *h, double *stats, void *stream) {
    if (n_cols < 1 || n_cols > (1 << 20) || !stats || (m_rows > 0 && !h)) return set_error(GSN_E_INVALID, "gsn_column_stats_hip: bad arguments");
    if (m_rows <= 0) return GSN_OK;
    const int64_t blocks = (m_rows + gsn::CS_ROWS - 1) / gsn::CS_ROWS;
    if (blocks > 0x7fffffff) return set_error(GSN_E_UNSUPPORTED, "gsn_column_stats_hip: too many rows");
    hipLaunchKernelGGL(gsn::column_stats_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), m_rows, (int)n_cols, h, stats);
    GSN_LAUNCH_CHECK("column_stats_kernel");
    return GSN_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// torch.cat((x[idx_0], x[idx_1], ids, e), -1) materialised (device): the training path of the `general` layers keeps the
// assembled edge rows for the weight gradient (the inference path never builds them: chain.hip gathers on the fly).
// ---------------------------------------------------------------------------------------------------------------------
namespace gsn {
constexpr int GC_MAXB = 6;
struct GatherCatArgs {
    int64_t m_rows;
    int n_blocks, k_total;
    const float *data[GC_MAXB];
    const int64_t *idx[GC_MAXB];
    const int32_t *idx32[GC_MAXB];
    int width[GC_MAXB], off[GC_MAXB + 1];
    float *out;
};

__global__ __launch_bounds__(256) void gather_cat_kernel(GatherCatArgs a) {
    const int64_t total = a.m_rows * a.k_total;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / a.k_total;
        const int k = (int)(i - r * a.k_total);
        int b = 0;
#pragma unroll
        for (int q = 1; q < GC_MAXB; ++q)
            if (q < a.n_blocks && k >= a.off[q]) b = q;
        const int64_t src = a.idx32[b] ? (int64_t)a.idx32[b][r] : (a.idx[b] ? a.idx[b][r] : r);
        a.out[i] = a.data[b][src * a.width[b] + (k - a.off[b])];
    }
}
}  // namespace gsn

// out[r] = x[r] + table[idx[r]]: the virtual node's embedding added to every vertex of its graph
// (models_graph_classification_ogb_original.py:236: x + vn_embedding[data.batch]) -- gather and add in one pass; a row whose index
// is outside the table comes out NaN (the reference raises an index error).
namespace gsn {
template <int VEC>
__global__ __launch_bounds__(256) void add_gathered_kernel(int64_t n_rows, int d, const float *__restrict__ x, const float *__restrict__ table,
                                                           const int64_t *__restrict__ idx, int64_t n_table, float *__restrict__ out) {
    const int per_row = d / VEC;
    const int64_t total = n_rows * per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / per_row;
        const int c = (int)(i - r * per_row) * VEC;
        const int64_t g = idx[r];
        const bool ok = g >= 0 && g < n_table;
        if (VEC == 4) {
            const float4 a = *reinterpret_cast<const float4 *>(x + r * d + c);
            float4 b = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
            if (ok) b = *reinterpret_cast<const float4 *>(table + g * d + c);
            *reinterpret_cast<float4 *>(out + r * d + c) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        } else {
            out[r * d + c] = x[r * d + c] + (ok ? table[g * d + c] : __builtin_nanf(""));
        }
    }
}
}  // namespace gsn

extern "C" int gsn_add_gathered_hip(int64_t n_rows, int64_t d, const float *x, const float *table, const int64_t *idx, int64_t n_table,
                                    float *out, void *stream) {
    if (d < 1 || n_table < 0 || (n_rows > 0 && (!x || !table || !idx || !out))) return set_error(GSN_E_INVALID, "gsn_add_gathered_hip: bad arguments");
    if (n_rows <= 0) return GSN_OK;
    const bool v4 = d % 4 == 0 && (((uintptr_t)x | (uintptr_t)table | (uintptr_t)out) % 16 == 0);
    const int64_t total = n_rows * (v4 ? d / 4 : d);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (v4) hipLaunchKernelGGL((gsn::add_gathered_kernel<4>), dim3((unsigned)blocks), dim3(256), 0, s, n_rows, (int)d, x, table, idx, n_table, out);
    else hipLaunchKernelGGL((gsn::add_gathered_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, s, n_rows, (int)d, x, table, idx, n_table, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "add_gathered_kernel: %s", hipGetErrorString(e));
    return GSN_OK;
}

extern "C" int gsn_gather_cat_hip(int64_t m_rows, int n_blocks, const gsn_block *blocks, float *out, void *stream) {
    if (n_blocks < 1 || n_blocks > GC_MAXB || !blocks || (m_rows > 0 && !out)) return set_error(GSN_E_INVALID, "gsn_gather_cat_hip: bad arguments");
    GatherCatArgs a{};
    a.m_rows = m_rows; a.n_blocks = n_blocks; a.out = out; a.off[0] = 0;
    for (int b = 0; b < n_blocks; ++b) {
        // (zero rows: per-edge blocks of an edge-less batch are empty tensors without a pointer)
        if ((!blocks[b].data && m_rows > 0) || blocks[b].width <= 0) return set_error(GSN_E_INVALID, "gsn_gather_cat_hip: block %d is empty", b);
        a.data[b] = blocks[b].data; a.idx[b] = blocks[b].idx; a.idx32[b] = blocks[b].idx32; a.width[b] = (int)blocks[b].width;
        a.off[b + 1] = a.off[b] + a.width[b];
    }
    a.k_total = a.off[n_blocks];
    if (m_rows <= 0) return GSN_OK;
    hipLaunchKernelGGL(gather_cat_kernel, dim3(grid_for(m_rows * a.k_total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    GSN_LAUNCH_CHECK("gather_cat_kernel");
    return GSN_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// BatchNorm1d bookkeeping of a train-mode stage in ONE launch (nn.BatchNorm1d semantics): from the fp64 column sums of the
// pre-BN rows -> batch mean, biased variance, invstd, the epilogue vectors scale = gamma * invstd and shift = beta, and the
// running statistics update (momentum, unbiased variance).  Replaces ~10 tiny PyTorch launches per BatchNorm stage.
// ---------------------------------------------------------------------------------------------------------------------
namespace gsn {
__global__ __launch_bounds__(256) void bn_finalize_kernel(int n_cols, double m_rows, double eps, double momentum, const double *stats,
                                                          const float *gamma, const float *beta, float *running_mean,
                                                          float *running_var, float *mean, float *invstd, float *scale, float *shift,
                                                          int64_t *num_batches_tracked) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n_cols) return;
    if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;       // (nn.BatchNorm1d's counter: one tensor op per call otherwise)
    const double mu = stats[c] / m_rows;
    double var = stats[n_cols + c] / m_rows - mu * mu;
    var = var > 0.0 ? var : 0.0;
    const float is = (float)(1.0 / sqrt(var + eps));
    mean[c] = (float)mu;
    invstd[c] = is;
    scale[c] = gamma ? is * gamma[c] : is;
    shift[c] = beta ? beta[c] : 0.f;
    if (running_mean) {
        const double unbiased = var * (m_rows / (m_rows > 1.0 ? m_rows - 1.0 : 1.0));
        running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * (double)(float)mu);
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * (double)(float)unbiased);
    }
}
}  // namespace gsn

namespace gsn {
// bn_finalize_kernel and bn_act_kernel in ONE launch (the train-mode BatchNorm of a materialised stage at the reference's batch sizes, where a
// launch costs more than the pass): a lane owns a column, computes its mean / invstd / scale / shift from the fp64 sums with bn_finalize_kernel's
// expressions (every workgroup the same values), the first row block writes the vectors and the running statistics, then the rows are
// normalised and activated with bn_act_kernel's expression.  grid (row blocks, ceil(C / 64)).
__global__ __launch_bounds__(256) void bn_finalize_act_kernel(int n_cols, int64_t m_rows_i, double eps, double momentum, const double *stats,
                                                              const float *gamma, const float *beta, float *running_mean, float *running_var,
                                                              float *mean, float *invstd, float *scale, float *shift,
                                                              int64_t *num_batches_tracked, const float *h, int act, float *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + lane;
    if (c >= n_cols) return;
    const double m_rows = (double)m_rows_i;
    const double mu = stats[c] / m_rows;
    double var = stats[n_cols + c] / m_rows - mu * mu;
    var = var > 0.0 ? var : 0.0;
    const float is = (float)(1.0 / sqrt(var + eps));
    const float mf = (float)mu, sc = gamma ? is * gamma[c] : is, sh = beta ? beta[c] : 0.f;
    if (blockIdx.x == 0 && wave == 0) {
        if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
        mean[c] = mf; invstd[c] = is; scale[c] = sc; shift[c] = sh;
        if (running_mean) {
            const double unbiased = var * (m_rows / (m_rows > 1.0 ? m_rows - 1.0 : 1.0));
            running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * (double)(float)mu);
            running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * (double)(float)unbiased);
        }
    }
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < m_rows_i; r += (int64_t)gridDim.x * 4) {
        const int64_t i = r * n_cols + c;
        float y = (h[i] - mf) * sc + sh;
        switch (act) {
            case 1: y = y > 0.f ? y : 0.f; break;
            case 2: y = y > 0.f ? y : expm1f(y); break;
            case 3: y = tanhf(y); break;
            default: break;
        }
        out[i] = y;
    }
}
}  // namespace gsn

extern "C" int gsn_bn_finalize_act_hip(int64_t n_cols, int64_t m_rows, double eps, double momentum, const double *stats, const float *gamma,
                                       const float *beta, float *running_mean, float *running_var, float *mean, float *invstd, float *scale,
                                       float *shift, int64_t *num_batches_tracked, const float *h, int act, float *out, void *stream) {
    if (n_cols < 1 || m_rows < 1 || !stats || !mean || !invstd || !scale || !shift || !h || !out || act < 0 || act > 3 ||
        ((running_mean != nullptr) != (running_var != nullptr)))
        return set_error(GSN_E_INVALID, "gsn_bn_finalize_act_hip: bad arguments");
    int64_t bx = (m_rows + 15) / 16;
    bx = bx < 1 ? 1 : (bx > 1024 ? 1024 : bx);
    hipLaunchKernelGGL(bn_finalize_act_kernel, dim3((unsigned)bx, (unsigned)((n_cols + 63) / 64)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       (int)n_cols, m_rows, eps, momentum, stats, gamma, beta, running_mean, running_var, mean, invstd, scale, shift,
                       num_batches_tracked, h, act, out);
    GSN_LAUNCH_CHECK("bn_finalize_act_kernel");
    return GSN_OK;
}

extern "C" int gsn_bn_finalize_hip(int64_t n_cols, int64_t m_rows, double eps, double momentum, const double *stats,
                                   const float *gamma, const float *beta, float *running_mean, float *running_var, float *mean,
                                   float *invstd, float *scale, float *shift, void *stream) {
    return gsn_bn_finalize_count_hip(n_cols, m_rows, eps, momentum, stats, gamma, beta, running_mean, running_var, mean, invstd, scale, shift,
                                     nullptr, stream);
}

extern "C" int gsn_bn_finalize_count_hip(int64_t n_cols, int64_t m_rows, double eps, double momentum, const double *stats,
                                         const float *gamma, const float *beta, float *running_mean, float *running_var, float *mean,
                                         float *invstd, float *scale, float *shift, int64_t *num_batches_tracked, void *stream) {
    if (n_cols < 1 || m_rows < 1 || !stats || !mean || !invstd || !scale || !shift || ((running_mean != nullptr) != (running_var != nullptr)))
        return set_error(GSN_E_INVALID, "gsn_bn_finalize_hip: bad arguments");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((n_cols + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       (int)n_cols, (double)m_rows, eps, momentum, stats, gamma, beta, running_mean, running_var, mean, invstd, scale, shift,
                       num_batches_tracked);
    GSN_LAUNCH_CHECK("bn_finalize_kernel");
    return GSN_OK;
}
