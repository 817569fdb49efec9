// First Linear of an edge MLP over one-hot encoded inputs, as a weight-row gather.
//
// When every input block of msg_fn's first Linear is the one-hot encoding of an integer code column (atom types, bond
// types, dense-recoded substructure counts: DiscreteEmbedding('one_hot_encoder'), utils_graph_learning.py:78-81), the
// product [one_hot(c_0) | one_hot(c_1) | ...] W^T is the sum of one column of W per code:
//     h[e][:] = sum_s  WT[w_off_s + code_s(e)][:]                      WT = W^T, [K][n_out] row-major
// so neither the dense [E, K] one-hot matrix nor the E x K x n_out product exists.  The kernel walks the edges in
// target-sorted order, applies bias / BatchNorm / activation and sums the rows of each target on the fly (rows of one
// target are consecutive, so a running register sum replaces the scatter-add).
//
// Bound: LDS bandwidth -- n_slots * n_out * 4 B of LDS reads per edge (WT lives in LDS), no HBM traffic beyond the codes
// (8 B per code) and the [N, n_out] result.  Integer gathers + fp32 adds; the only difference to the dense formulation
// is the summation order of the <= 16 non-zero terms of each dot product.
#include <hip/hip_runtime.h>

#include "gsn_internal.h"

namespace gsn {

struct CodeStageArgs {
    int64_t n_rows;
    int n_slots, n_out, k_total, act, rows_per_wg;
    gsn_code_slot slots[GSN_MAX_CODE_SLOTS];
    const float *wt, *bias, *bn_mean, *bn_scale, *bn_shift;
    const int32_t *seg_target;
    float *out;
    double *stats;
    int32_t *status;
};

__device__ __forceinline__ float code_act(float v, int act) {
    switch (act) {
        case 1: return v > 0.f ? v : 0.f;
        case 2: return v > 0.f ? v : expm1f(v);
        case 3: return tanhf(v);
        default: return v;
    }
}

// One wave serves one row at a time, every lane CPL adjacent output columns (vector LDS reads); the 4 waves of a
// workgroup walk their own contiguous shares of the workgroup's rows.  NSLOT = number of gathers per row, padded to a
// multiple of 4 (padding slots read an all-zero row appended to WT in LDS), so the row body is branch-free: NSLOT
// independent LDS reads.  The kernel is VALU-issue bound (a wave64 instruction occupies the SIMD for 4 cycles), hence
// the wide lanes: instructions per row do not grow with CPL.
template <int CPL>
struct FVec {
    float v[CPL];
};

template <int CPL, int NSLOT, bool STATS, bool FANCY>
__global__ __launch_bounds__(256) void code_stage_kernel(CodeStageArgs a) {
    extern __shared__ __attribute__((aligned(16))) float wt[];   // [k_total + 1][pitch], last row zero
    __shared__ gsn_code_slot sl[GSN_MAX_CODE_SLOTS];
    constexpr int NP = (NSLOT + 7) / 8;                    // index registers per 8-row batch
    typedef float vec_t __attribute__((ext_vector_type(CPL)));
    const int tid = threadIdx.x, lane = tid & 63;
    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pitch = (a.n_out + 3) & ~3;
    const int j0 = lane * CPL;                             // first column of this lane
    const int jb = j0 < pitch ? j0 : 0;                    // LDS column of it (lanes beyond the row read column 0)
    for (int i = tid; i < (a.k_total + 1) * pitch; i += 256) {
        const int r = i / pitch, c = i - r * pitch;
        wt[i] = (r < a.k_total && c < a.n_out) ? a.wt[r * a.n_out + c] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < GSN_MAX_CODE_SLOTS; ++s)
        if (tid == s && s < a.n_slots) sl[s] = a.slots[s];
    float bias[CPL], scale[CPL], c0[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const int j = j0 + c;
        const bool ok = j < a.n_out;
        bias[c] = (ok && a.bias) ? a.bias[j] : 0.f;
        const float mean = (ok && a.bn_mean) ? a.bn_mean[j] : 0.f;
        scale[c] = (ok && a.bn_scale) ? a.bn_scale[j] : 1.f;
        c0[c] = (bias[c] - mean) * scale[c] + ((ok && a.bn_shift) ? a.bn_shift[j] : 0.f);
    }
    __syncthreads();
    const int64_t rpg = a.rows_per_wg / 4;
    const int64_t q0 = (int64_t)blockIdx.x * a.rows_per_wg + (int64_t)g * rpg;
    int64_t q1 = q0 + rpg;
    if (q1 > a.n_rows) q1 = a.n_rows;
    if (q0 >= q1) return;
    bool first_shared = false, last_shared = false;
    int cur = 0;
    if (!STATS) {
        cur = __builtin_amdgcn_readfirstlane(a.seg_target[q0]);
        first_shared = q0 > 0 && a.seg_target[q0 - 1] == cur;
        last_shared = q1 < a.n_rows && a.seg_target[q1] == a.seg_target[q1 - 1];
    }
    const bool vec_store = (a.n_out % CPL) == 0;           // rows of `out` keep vector alignment
    bool first = true;
    float acc[CPL];
    double s1[CPL], s2[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) { acc[c] = 0.f; s1[c] = 0.0; s2[c] = 0.0; }
    const int n_slots = a.n_slots;
    const int my_slot = lane & 7;
    const float act_floor = a.act == 1 ? 0.f : -INFINITY;   // identity / relu as one max
    // row offsets are kept in FLOATS and every access goes through `wt` itself: a char* view of the LDS array loses its
    // address space and turns the gathers into FLAT loads
    const int zero_off = a.k_total * pitch;

    auto flush = [&](int target, bool atomic) {
        float *o = a.out + (int64_t)target * a.n_out + j0;
        if (atomic) {
#pragma unroll
            for (int c = 0; c < CPL; ++c)
                if (j0 + c < a.n_out) atomicAdd(o + c, acc[c]);
        } else if (vec_store) {
            if (j0 < a.n_out) {
                vec_t v;
#pragma unroll
                for (int c = 0; c < CPL; ++c) v[c] = acc[c];
                *reinterpret_cast<vec_t *>(o) = v;
            }
        } else {
#pragma unroll
            for (int c = 0; c < CPL; ++c)
                if (j0 + c < a.n_out) o[c] = acc[c];
        }
    };

    constexpr int NB = 8;                                  // 8-row batches per block: 64 rows of gathers in flight
    for (int64_t q = q0; q < q1; q += 8 * NB) {
        // lanes (t = lane>>3, s = lane&7) of batch u fetch the LDS byte offset of the weight row of slot s + 8p of row
        // q + 8u + t
        int kv[NB][NP];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int64_t qr = q + 8 * u + (lane >> 3);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int s = my_slot + 8 * p;
                int off = zero_off;
                if (qr < q1 && s < n_slots) {
                    const gsn_code_slot d = sl[s];
                    const int64_t r = d.idx ? (int64_t)d.idx[qr] : qr;
                    int64_t code = d.codes[r * d.stride + d.col];
                    if (code < 0 || code >= d.n_classes) {
                        if (d.clamp) code = code < 0 ? 0 : d.n_classes - 1;
                        else { atomicMax(a.status, GSN_ST_BAD_INDEX); code = 0; }
                    }
                    off = (d.w_off + (int)code) * pitch;
                }
                kv[u][p] = off;
            }
        }
        int tg = 0;
        if (!STATS && q + lane < q1) tg = a.seg_target[q + lane];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            if (q + 8 * u >= q1) break;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (q + 8 * u + t >= q1) break;
                vec_t hv[NSLOT];
#pragma unroll
                for (int s = 0; s < NSLOT; ++s) {
                    const int base = __builtin_amdgcn_readlane(kv[u][s >> 3], t * 8 + (s & 7));
                    hv[s] = *reinterpret_cast<const vec_t *>(&wt[base + jb]);
                }
                vec_t h = hv[0];
#pragma unroll
                for (int s = 1; s < NSLOT; ++s) h += hv[s];
                if (!STATS) {
                    const int tgt = __builtin_amdgcn_readlane(tg, u * 8 + t);
                    if (tgt != cur) {
                        flush(cur, first && first_shared);
                        first = false;
#pragma unroll
                        for (int c = 0; c < CPL; ++c) acc[c] = 0.f;
                        cur = tgt;
                    }
#pragma unroll
                    for (int c = 0; c < CPL; ++c) {
                        const float z = fmaf(h[c], scale[c], c0[c]);
                        acc[c] += FANCY ? code_act(z, a.act) : fmaxf(z, act_floor);
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < CPL; ++c) {
                        const double hb = (double)(h[c] + bias[c]);
                        s1[c] += hb;
                        s2[c] += hb * hb;
                    }
                }
            }
        }
    }
    if (STATS) {
#pragma unroll
        for (int c = 0; c < CPL; ++c)
            if (j0 + c < a.n_out) {
                atomicAdd(a.stats + j0 + c, s1[c]);
                atomicAdd(a.stats + a.n_out + j0 + c, s2[c]);
            }
    } else {
        flush(cur, last_shared || (first && first_shared));
    }
}

template <int CPL, int NSLOT>
static void launch_code_stage2(const CodeStageArgs &a, bool stats, int grid, size_t lds, hipStream_t s) {
    if (stats) hipLaunchKernelGGL((code_stage_kernel<CPL, NSLOT, true, false>), dim3(grid), dim3(256), lds, s, a);
    else if (a.act >= 2) hipLaunchKernelGGL((code_stage_kernel<CPL, NSLOT, false, true>), dim3(grid), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((code_stage_kernel<CPL, NSLOT, false, false>), dim3(grid), dim3(256), lds, s, a);
}

template <int CPL>
static void launch_code_stage(const CodeStageArgs &a, bool stats, int grid, size_t lds, hipStream_t s) {
    const int ns = (a.n_slots + 3) / 4 * 4;
    if (ns == 4) launch_code_stage2<CPL, 4>(a, stats, grid, lds, s);
    else if (ns == 8) launch_code_stage2<CPL, 8>(a, stats, grid, lds, s);
    else if (ns == 12) launch_code_stage2<CPL, 12>(a, stats, grid, lds, s);
    else launch_code_stage2<CPL, 16>(a, stats, grid, lds, s);
}

}  // namespace gsn

using namespace gsn;

extern "C" int gsn_code_stage_supported(int n_slots, int64_t k_total, int64_t n_out) {
    return n_slots >= 1 && n_slots <= GSN_MAX_CODE_SLOTS && n_out >= 1 && n_out <= 256 && k_total >= 1 &&
           (k_total + 1) * ((n_out + 3) / 4 * 4) * 4 <= 64 * 1024;
}

extern "C" int gsn_code_stage_fwd_hip(int64_t m_rows, int n_slots, const gsn_code_slot *slots, const float *WT, int64_t k_total,
                                      const float *bias, int64_t n_out, const float *bn_mean, const float *bn_scale,
                                      const float *bn_shift, int act, const int32_t *seg_target, float *out, double *stats,
                                      int32_t *status, void *stream) {
    if (!gsn_code_stage_supported(n_slots, k_total, n_out))
        return set_error(GSN_E_UNSUPPORTED, "gsn_code_stage_fwd_hip: needs 1..%d slots, n_out <= 256 and K*n_out*4 <= 64 KiB",
                         GSN_MAX_CODE_SLOTS);
    if (!slots || !WT || !status || act < 0 || act > 3 || (!stats && (!seg_target || !out)))
        return set_error(GSN_E_INVALID, "gsn_code_stage_fwd_hip: bad arguments");
    if (m_rows <= 0) return GSN_OK;
    CodeStageArgs a{};
    a.n_rows = m_rows; a.n_slots = n_slots; a.n_out = (int)n_out; a.k_total = (int)k_total; a.act = act;
    for (int s = 0; s < n_slots; ++s) {
        a.slots[s] = slots[s];
        if (!slots[s].codes || slots[s].n_classes < 1 || slots[s].w_off < 0 || slots[s].w_off + slots[s].n_classes > k_total)
            return set_error(GSN_E_INVALID, "gsn_code_stage_fwd_hip: slot %d is inconsistent with K = %lld", s, (long long)k_total);
    }
    a.wt = WT; a.bias = bias; a.bn_mean = bn_mean; a.bn_scale = bn_scale; a.bn_shift = bn_shift;
    a.seg_target = seg_target; a.out = out; a.stats = stats; a.status = status;
    const int cpl = n_out <= 64 ? 1 : (n_out <= 128 ? 2 : 4);
    const int64_t unit = (int64_t)GSN_SEG_RANGE_ROWS * 4;            // the 4 waves start on prepared boundaries
    int64_t rpw = (m_rows + 4095) / 4096;
    rpw = (rpw + unit - 1) / unit * unit;
    if (rpw < 8 * unit) rpw = 8 * unit;
    a.rows_per_wg = (int)rpw;
    const int grid = (int)((m_rows + rpw - 1) / rpw);
    const size_t lds = (size_t)(k_total + 1) * ((n_out + 3) / 4 * 4) * sizeof(float);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (cpl == 1) launch_code_stage<1>(a, stats != nullptr, grid, lds, s);
    else if (cpl == 2) launch_code_stage<2>(a, stats != nullptr, grid, lds, s);
    else launch_code_stage<4>(a, stats != nullptr, grid, lds, s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "code_stage_kernel: %s", hipGetErrorString(e));
    return GSN_OK;
}
