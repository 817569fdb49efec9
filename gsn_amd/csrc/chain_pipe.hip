// Stage-pipelined variant of the two-stage fused chain (chain.hip) for the node chain of a layer,
//   out = act1(bn1( act0(bn0([blocks] W0^T + b0)) W1^T + b1 )) ,  plain [M][n_out] output.
//
// chain.hip runs both stages in the same four waves: 144 weight registers per lane leave room for ONE wave per SIMD, so
// nothing overlaps that wave's staging / epilogue work (ablation: 0.22 ms of non-MFMA work + 0.90 ms of MFMA phase at 65 536
// graphs).  Here a workgroup has two groups of four waves and every SIMD hosts one wave of each:
//   group A (waves 0-3): stage-0 weights in registers (<= 80); per tile  gather next tile (under its own MFMAs) ->
//                        stage-0 MFMAs on IN[t&1] -> bias/BN/activation -> MID[t&1] -> next tile's rows to IN[(t+1)&1];
//   group B (waves 4-7): stage-1 weights in registers (<= 64); per tile  stage-1 MFMAs on MID[(t-1)&1] -> epilogue -> HBM.
// B works one tile behind A; one workgroup barrier per tile hands MID over.  While a wave of one group runs its epilogue or
// staging, the other group's wave on the same SIMD keeps the matrix pipe busy.  Each wave needs < 256 registers.
// LDS: IN[2] + MID[2] + row-source ring = 154 KiB at K0 = 160 -> one workgroup per CU.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "chain_common.h"

namespace gsn {

template <int CH0, int CH1>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void mlp_chain2_pipe_kernel(ChainArgs a, int pin, int pmid) {
    constexpr int GT = 256;                         // threads per group
    constexpr int RSTEP = GT / 32;
    constexpr int NROW = CBM / RSTEP;
    constexpr int NP = 2;
    constexpr int PF0_J = (CH0 * CHK + 31) / 32;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // buffers as offsets into `lds` (a pointer picked from an array of buffer pointers loses its LDS address space)
    const int in_sz = CBM * pin, mid_sz = CBM * pmid;
    auto in_tile = [&](int64_t i) { return lds + (int)(i & 1) * in_sz; };
    auto mid_tile = [&](int64_t i) { return lds + 2 * in_sz + (int)(i & 1) * mid_sz; };
    int *rsrc = reinterpret_cast<int *>(lds + 2 * in_sz + 2 * mid_sz);     // [3][RS_STRIDE]

    const int tid = threadIdx.x;
    const bool grp_b = tid >= GT;                   // wave-uniform
    const int t = tid & (GT - 1);
    const int lane = t & 63, w = t >> 6;            // column block: output columns 32w .. 32w+31
    const int li = lane & 31, lh = lane >> 5;
    const int64_t n_tiles = (a.m_rows + CBM - 1) / CBM;
    const int64_t first = blockIdx.x;
    const int64_t n_iter = first < n_tiles ? (n_tiles - first + gridDim.x - 1) / gridDim.x : 0;

    for (int i = tid; i < 2 * in_sz + 2 * mid_sz; i += 512) lds[i] = 0.f;      // padded columns must hold finite values
    __syncthreads();

    if (!grp_b) {
        // =============================================================================================================
        // group A: gathers + stage 0
        // =============================================================================================================
        const ChainStage &st = a.st[0];
        const int nch = (st.k_total + CHK - 1) / CHK;
        const int col = 32 * w + li;
        const bool cok = col < st.n_out;
        const bool active = 32 * w < st.n_out;
        float B0[CH0][8];
#pragma unroll
        for (int ch = 0; ch < CH0; ++ch)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k = ch * CHK + 2 * q + lh;
                B0[ch][q] = (k < st.k_total && cok) ? st.W[(int64_t)col * st.k_total + k] : 0.f;
            }
        const float bias = (cok && st.bias) ? st.bias[col] : 0.f;
        float scale = 1.f, c0 = bias;
        if (cok && st.bn_scale) { scale = st.bn_scale[col]; c0 = (bias - st.bn_mean[col]) * scale + st.bn_shift[col]; }
        const int kc0 = t & 31, r0 = t >> 5;
        ColMap cm0[PF0_J];
#pragma unroll
        for (int j = 0; j < PF0_J; ++j) cm0[j] = col_map(a, 0, kc0 + 32 * j);
        float pf0[PF0_J][NROW];
        RowSrcC<NP> rsn;
        const RowSrcThread<NP> rst = rs_thread<NP>(a, t);
        auto prefetch_j = [&](const int *rs, int j) {
#pragma unroll
            for (int i = 0; i < NROW; ++i) {
                const int sr = rs[cm0[j].rsoff + r0 + RSTEP * i];
                pf0[j][i] = cm0[j].base[(int64_t)(sr < 0 ? 0 : sr) * cm0[j].bw];
            }
        };
        auto stage_in = [&](float *dst) {
#pragma unroll
            for (int j = 0; j < PF0_J; ++j)
#pragma unroll
                for (int i = 0; i < NROW; ++i) dst[(r0 + RSTEP * i) * pin + kc0 + 32 * j] = pf0[j][i];
        };
        {
            RowSrcC<NP> r;
            rs_fetch(rst, first * CBM, r);
            rs_store(rsrc, rst, r);
            rs_fetch(rst, (first + gridDim.x) * CBM, r);
            rs_store(rsrc + RS_STRIDE, rst, r);
        }
        lds_barrier();
        if (n_iter > 0) {
#pragma unroll
            for (int j = 0; j < PF0_J; ++j) prefetch_j(rsrc, j);
            stage_in(in_tile(0));
        }
        lds_barrier();
        int slot = 0;
        for (int64_t i = 0; i <= n_iter; ++i) {
            if (i < n_iter) {
                const int64_t tile = first + i * gridDim.x;
                const int slot_n = slot == 2 ? 0 : slot + 1, slot_nn = slot_n == 2 ? 0 : slot_n + 1;
                const int *rs_next = rsrc + slot_n * RS_STRIDE;
                rs_fetch(rst, (tile + 2 * (int64_t)gridDim.x) * CBM, rsn);
                const float *in = in_tile(i);
                f32x16 acc[2];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[h][r] = 0.f;
                if (active) {
                    const float *ap = in + li * pin + lh;
#pragma unroll
                    for (int ch = 0; ch < CH0; ++ch) {
                        if (ch < nch) {
                            if (ch < PF0_J) prefetch_j(rs_next, ch);       // next tile's gathers under this chunk's MFMAs
#pragma unroll
                            for (int q = 0; q < 8; ++q)
#pragma unroll
                                for (int h = 0; h < 2; ++h)
                                    acc[h] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[h * 32 * pin + ch * CHK + 2 * q], B0[ch][q], acc[h], 0, 0, 0);
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < PF0_J; ++j) prefetch_j(rs_next, j);
                }
                // stage output -> MID[i&1]  (group B finished reading it one barrier ago)
                float *lp = mid_tile(i) + (4 * lh) * pmid + col;
                if (cok) {
                    if (st.act == 1) {
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int r = 0; r < 16; ++r) { const float y = fmaf(acc[h][r], scale, c0); lp[(h * 32 + (r & 3) + 8 * (r >> 2)) * pmid] = y > 0.f ? y : 0.f; }
                    } else {
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int r = 0; r < 16; ++r) lp[(h * 32 + (r & 3) + 8 * (r >> 2)) * pmid] = fmaf(acc[h][r], scale, c0);
                    }
                }
                stage_in(in_tile(i + 1));                                   // (waits for the gathers; no stores in this group)
                rs_store(rsrc + slot_nn * RS_STRIDE, rst, rsn);
                slot = slot_n;
            }
            lds_barrier();
        }
        return;
    }

    // =================================================================================================================
    // group B: stage 1 + output, one tile behind group A
    // =================================================================================================================
    const ChainStage &st = a.st[1];
    const int nch = (st.k_total + CHK - 1) / CHK;
    const int col = 32 * w + li;
    const bool cok = col < st.n_out;
    const bool active = 32 * w < st.n_out;
    float B1[CH1][8];
#pragma unroll
    for (int ch = 0; ch < CH1; ++ch)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = ch * CHK + 2 * q + lh;
            B1[ch][q] = (k < st.k_total && cok) ? st.W[(int64_t)col * st.k_total + k] : 0.f;
        }
    const float bias = (cok && st.bias) ? st.bias[col] : 0.f;
    float scale = 1.f, c0 = bias;
    if (cok && st.bn_scale) { scale = st.bn_scale[col]; c0 = (bias - st.bn_mean[col]) * scale + st.bn_shift[col]; }
    lds_barrier();
    lds_barrier();
    for (int64_t i = 0; i <= n_iter; ++i) {
        if (i > 0) {
            const int64_t row0 = (first + (i - 1) * gridDim.x) * CBM;
            const float *in = mid_tile(i - 1);
            f32x16 acc[2];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[h][r] = 0.f;
            if (active) {
                const float *ap = in + li * pmid + lh;
#pragma unroll
                for (int ch = 0; ch < CH1; ++ch) {
                    if (ch < nch) {
#pragma unroll
                        for (int q = 0; q < 8; ++q)
#pragma unroll
                            for (int h = 0; h < 2; ++h)
                                acc[h] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[h * 32 * pmid + ch * CHK + 2 * q], B1[ch][q], acc[h], 0, 0, 0);
                    }
                }
            }
            float *tile_out = a.out + row0 * st.n_out;                  // wave-uniform base
            const int lane_off = (4 * lh) * st.n_out + col;            // 32-bit per-lane offset inside the tile
            const bool full = row0 + CBM <= a.m_rows;
            auto emit = [&](auto actf) {
                if (full) {
                    if (cok) {
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int r = 0; r < 16; ++r) tile_out[lane_off + (h * 32 + (r & 3) + 8 * (r >> 2)) * st.n_out] = actf(fmaf(acc[h][r], scale, c0));
                    }
                } else {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int dr = h * 32 + (r & 3) + 8 * (r >> 2);
                            if (cok && row0 + 4 * lh + dr < a.m_rows) tile_out[lane_off + dr * st.n_out] = actf(fmaf(acc[h][r], scale, c0));
                        }
                }
            };
            if (st.act == 1) emit([](float y) { return y > 0.f ? y : 0.f; });
            else emit([](float y) { return y; });
        }
        lds_barrier();
    }
}

template <int CH0, int CH1>
static int launch_pipe_impl(const ChainArgs &a, int pin, int pmid, size_t lds, hipStream_t st) {
    static DeviceOnce attr_set;
    const int attr_dev = current_device();
    if (!attr_set.done(attr_dev)) {
        hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp_chain2_pipe_kernel<CH0, CH1>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e0 != hipSuccess) return set_error(GSN_E_HIP, "hipFuncSetAttribute(mlp_chain2_pipe_kernel): %s", hipGetErrorString(e0));
        attr_set.mark(attr_dev);
    }
    const int64_t n_tiles = (a.m_rows + CBM - 1) / CBM;
    int64_t gx = 256;
    if (gx > n_tiles) gx = n_tiles;
    chain_trace("mlp_chain2_pipe_kernel", a);
    hipLaunchKernelGGL((mlp_chain2_pipe_kernel<CH0, CH1>), dim3((unsigned)gx), dim3(512), lds, st, a, pin, pmid);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "mlp_chain2_pipe_kernel: %s", hipGetErrorString(e));
    return GSN_OK;
}

// Returns GSN_OK after launching, or 1 if this shape is not covered (the caller then uses chain.hip's kernel).
int launch_chain2_pipe(const ChainArgs &a, int maxch, hipStream_t st) {
    if (a.n_stages != 2 || a.stats || a.seg_target) return 1;
    { const char *d = getenv("GSN_CHAIN_PIPE"); if (d && atoi(d) == 0) return 1; }
    const int pin = ((maxch * CHK + 31) / 32 * 32) | 1;
    // (the matrix loop reads whole CHK-column chunks: a pitch below the padded width lets the last row of a tile read past the
    //  tile -- into the row-source table behind the second MID tile, whose -1 entries are NaNs as floats: NaN x 0 weights)
    const int k1_pad = (a.st[1].k_total + CHK - 1) / CHK * CHK;
    int pmid = k1_pad | 1;
    if (pmid == k1_pad) pmid += 2;
    const size_t lds = ((size_t)2 * CBM * pin + (size_t)2 * CBM * pmid + (size_t)3 * RS_STRIDE) * 4;
    if (lds > 160 * 1024) return 1;
    if (maxch == 5) return launch_pipe_impl<5, 8>(a, pin, pmid, lds, st);
    return launch_pipe_impl<10, 8>(a, pin, pmid, lds, st);
}

}  // namespace gsn
