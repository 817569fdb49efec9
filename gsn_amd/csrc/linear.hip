// HP-2 dense stage for gfx950: one fused models_misc.mlp layer on the fp32 matrix cores.
//
//   Y = act( (X W^T + bias - mean) * scale + shift )        X rows assembled on the fly from up to 4 blocks
//
// replaces  torch.cat((x_i, x_j, identifiers.., edge_features), -1) -> nn.Linear -> BatchNorm1d -> activation
// (models_misc.py:52-58 driven by GSN_sparse.py:166-171 / GSN_edge_sparse.py:160-165): neither the gathered
// x[edge_index_i] / x[edge_index_j] copies nor the [E, msg_in] concatenation ever exist in HBM.
//
// fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32 (exact f32, 64 FLOP/clk/SIMD = the chip's 157 TF fp32 peak;
// gfx950 has no TF32, and the 1e-5 parity tolerance rules out bf16).  Tiling for 64-wide waves:
//   workgroup = 4 waves = 128 x 128 output tile; each wave owns 64 x 64 = 2 x 2 MFMA tiles (64 accumulator VGPRs);
//   K is walked in chunks of <= 32 that never straddle two input blocks; per chunk the A tile [128][32] and the
//   W^T tile [32][128] are staged in LDS with +1 padding (row pitch 33 / 129 words) so that both the staging writes
//   and the per-lane fragment reads (A[i=lane&31][k=lane>>5], B[k=lane>>5][j=lane&31]) are bank-conflict free.
//   Workgroups are persistent over row tiles (grid-stride) so the train-mode BatchNorm statistics pass can keep
//   per-column partial sums in registers (fp64) and issue one atomic per column per workgroup.
#include <hip/hip_runtime.h>

#include "gsn_internal.h"

namespace gsn {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int APITCH = BK + 1, WPITCH = BN + 1;
constexpr int MAX_BLOCKS = 5, MAX_CHUNKS = 24;

struct LinArgs {
    int64_t m_rows;
    int n_blocks, n_chunks;
    const float *bdata[MAX_BLOCKS];
    const int64_t *bidx[MAX_BLOCKS];
    const int32_t *bidx32[MAX_BLOCKS];
    int bwidth[MAX_BLOCKS];
    // chunk c: block, column offset inside the block, length (<= BK), k offset inside a row of W
    unsigned char cblock[MAX_CHUNKS], clen[MAX_CHUNKS];
    short ccol[MAX_CHUNKS], cwk[MAX_CHUNKS];
    const float *W, *bias, *bn_mean, *bn_scale, *bn_shift;
    int k_total, n_out, act;
    const int32_t *row_perm;
    float *out;
    double *stats;
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float apply_act(float y, int act) {
    switch (act) {
        case 1: return y > 0.f ? y : 0.f;
        case 2: return y > 0.f ? y : expm1f(y);
        case 3: return tanhf(y);
        default: return y;
    }
}

__global__ __launch_bounds__(256) void linear_fwd_stream_kernel(LinArgs a) {
    __shared__ float As[BM * APITCH];
    __shared__ float Ws[BK * WPITCH];
    __shared__ int64_t rowsrc[MAX_BLOCKS][BM];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.y * BN;
    const int64_t n_tiles = (a.m_rows + BM - 1) / BM;

    double st_sum[2] = {0.0, 0.0}, st_sq[2] = {0.0, 0.0};

    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t row0 = tile * BM;
        // resolve the source row of every tile row for every block once
        if (tid < BM) {
            const int64_t grow = row0 + tid;
            int64_t logical = 0;
            if (grow < a.m_rows) logical = a.row_perm ? (int64_t)a.row_perm[grow] : grow;
#pragma unroll
            for (int b = 0; b < MAX_BLOCKS; ++b)
                if (b < a.n_blocks) rowsrc[b][tid] = (grow < a.m_rows) ? (a.bidx32[b] ? (int64_t)a.bidx32[b][logical] : (a.bidx[b] ? a.bidx[b][logical] : logical)) : -1;
        }
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        __syncthreads();

        for (int c = 0; c < a.n_chunks; ++c) {
            const int blk = a.cblock[c], len = a.clen[c], coff = a.ccol[c], wk = a.cwk[c];
            const float *bd = a.bdata[blk];
            const int bw = a.bwidth[blk];
            // stage A chunk: lanes along k (contiguous 128 B per row), 8 rows per pass
            {
                const int kc = tid & 31, r0 = tid >> 5;
#pragma unroll
                for (int i = 0; i < BM / 8; ++i) {
                    const int r = r0 + 8 * i;
                    const int64_t sr = rowsrc[blk][r];
                    float v = 0.f;
                    if (kc < len && sr >= 0) v = bd[sr * bw + coff + kc];
                    As[r * APITCH + kc] = v;
                }
                // stage W^T chunk: Ws[k][j] = W[n0+j][wk+k]
#pragma unroll
                for (int i = 0; i < BN / 8; ++i) {
                    const int j = r0 + 8 * i;
                    float v = 0.f;
                    if (kc < len && n0 + j < a.n_out) v = a.W[(int64_t)(n0 + j) * a.k_total + wk + kc];
                    Ws[kc * WPITCH + j] = v;
                }
            }
            __syncthreads();
            const int ksteps = (len + 1) >> 1;
            const float *ap0 = As + (wm * 64 + li) * APITCH + lh;
            const float *ap1 = ap0 + 32 * APITCH;
            const float *bp0 = Ws + lh * WPITCH + wn * 64 + li;
            for (int ks = 0; ks < ksteps; ++ks) {
                const float a0 = ap0[2 * ks], a1 = ap1[2 * ks];
                const float b0 = bp0[2 * ks * WPITCH], b1 = bp0[2 * ks * WPITCH + 32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
            __syncthreads();
        }

        // epilogue.  C layout of a 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + li;
            const bool cok = col < a.n_out;
            const float bias = (cok && a.bias) ? a.bias[col] : 0.f;
            float mean = 0.f, scale = 1.f, shift = 0.f;
            if (cok && a.bn_scale) { mean = a.bn_mean[col]; scale = a.bn_scale[col]; shift = a.bn_shift[col]; }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = row0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (!cok || row >= a.m_rows) continue;
                    const float h = acc[i][j][r] + bias;
                    if (a.stats) {
                        st_sum[j] += (double)h;
                        st_sq[j] += (double)h * (double)h;
                    } else {
                        float y = h;
                        if (a.bn_scale) y = (h - mean) * scale + shift;
                        a.out[row * a.n_out + col] = apply_act(y, a.act);
                    }
                }
            }
        }
        __syncthreads();  // rowsrc is rewritten by the next tile
    }

    if (a.stats) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + li;
            double s = st_sum[j], q = st_sq[j];
            s += __shfl_xor(s, 32);
            q += __shfl_xor(q, 32);
            if (lh == 0 && col < a.n_out) {
                atomicAdd(&a.stats[col], s);
                atomicAdd(&a.stats[a.n_out + col], q);
            }
        }
    }
}


// ----------------------------------------------------------------------------------------------------------------
// Main kernel: W^T resident in LDS for the whole (persistent) workgroup, A staged in 32-wide slices of the
// CONCATENATED input row (a slice may straddle input blocks), software-pipelined:
//     write slice c (registers -> LDS buffer c&1) | barrier | issue global loads of slice c+1 into registers |
//     16 k-steps x 4 MFMA on slice c
// so HBM/L2 latency of the gathers hides under the 4096-cycle MFMA phase (one wave per SIMD, 4 independent
// accumulators keep the matrix pipe issuing back to back), one barrier per slice, and the gather indices of the NEXT
// row tile are fetched while the current tile computes.  Used when K_pad*129*4 + 38 KB fits the 160 KB LDS (K <= 224).
// ----------------------------------------------------------------------------------------------------------------
constexpr int NPRE = BM / 8;  // staged elements per thread per slice

struct RowSrc {
    int v[MAX_BLOCKS];
};

__device__ __forceinline__ void rowsrc_fetch(const LinArgs &a, int64_t row0, int tid, RowSrc &rs) {
    if (tid < BM) {
        const int64_t grow = row0 + tid;
        const bool ok = grow < a.m_rows;
        int64_t logical = 0;
        if (ok) logical = a.row_perm ? (int64_t)a.row_perm[grow] : grow;
#pragma unroll
        for (int b = 0; b < MAX_BLOCKS; ++b) {
            int r = -1;
            if (b < a.n_blocks && ok) r = a.bidx32[b] ? a.bidx32[b][logical] : (a.bidx[b] ? (int)a.bidx[b][logical] : (int)logical);
            rs.v[b] = r;
        }
    }
}

__device__ __forceinline__ void rowsrc_store(int *dst /*[MAX_BLOCKS][BM]*/, int tid, const RowSrc &rs) {
    if (tid < BM) {
#pragma unroll
        for (int b = 0; b < MAX_BLOCKS; ++b) dst[b * BM + tid] = rs.v[b];
    }
}

// global loads of slice c (columns c*32 .. c*32+31 of the concatenated row) for 16 rows of this thread.
// Loads are unconditional (row / column clamped into the block, result masked) so the 16 of them issue back to back.
__device__ __forceinline__ void slice_fetch(const LinArgs &a, const int *rsrc, int c, int tid, float (&pre)[NPRE]) {
    const int kc = tid & 31, r0 = tid >> 5;
    const int kg = c * BK + kc;
    int blk = 0, col = kg;
#pragma unroll
    for (int b = 0; b < MAX_BLOCKS - 1; ++b) {
        if (b < a.n_blocks - 1 && blk == b && col >= a.bwidth[b]) { col -= a.bwidth[b]; blk = b + 1; }
    }
    const bool kok = kg < a.k_total;
    const float *bd = a.bdata[0];
    int bw = a.bwidth[0];
#pragma unroll
    for (int b = 1; b < MAX_BLOCKS; ++b)
        if (blk == b) { bd = a.bdata[b]; bw = a.bwidth[b]; }
    col = kok ? col : 0;
    const float *bcol = bd + col;
    const int *rs = rsrc + blk * BM + r0;
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
        const int sr = rs[8 * i];
        pre[i] = bcol[(int64_t)(sr < 0 ? 0 : sr) * bw];   // raw; masked when written to LDS (a select here would force
    }                                                      // an immediate vmcnt wait and serialise the prefetch)
}

template <bool STATS>
__global__ __launch_bounds__(256) void linear_fwd_kernel(LinArgs a, int k_pad) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Wt = lds;                                   // [k_pad][WPITCH]
    float *As = Wt + k_pad * WPITCH;                   // [2][BM][APITCH]
    int *rsrc = reinterpret_cast<int *>(As + 2 * BM * APITCH);  // [2][MAX_BLOCKS][BM]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.y * BN;
    const int64_t n_tiles = (a.m_rows + BM - 1) / BM;
    const int n_slices = k_pad / BK;

    // W^T resident: Wt[k][j] = W[n0+j][k], zero padded
    for (int i = tid; i < k_pad * BN; i += 256) {
        const int j = i / k_pad, k = i - j * k_pad;   // consecutive threads walk k: coalesced rows of W
        float v = 0.f;
        if (k < a.k_total && n0 + j < a.n_out) v = a.W[(int64_t)(n0 + j) * a.k_total + k];
        Wt[k * WPITCH + j] = v;
    }

    double st_sum[2] = {0.0, 0.0}, st_sq[2] = {0.0, 0.0};
    float pre[NPRE];
    RowSrc rs_next;
    int64_t tile = blockIdx.x;
    if (tile < n_tiles) {
        RowSrc rs0;
        rowsrc_fetch(a, tile * BM, tid, rs0);
        rowsrc_store(rsrc, tid, rs0);
    }
    __syncthreads();
    if (tile < n_tiles) slice_fetch(a, rsrc, 0, tid, pre);
    int cur_rs = 0, cur_as = 0;

    for (; tile < n_tiles; tile += gridDim.x) {
        const int64_t row0 = tile * BM;
        const int64_t next_tile = tile + gridDim.x;
        const bool has_next = next_tile < n_tiles;
        if (has_next) rowsrc_fetch(a, next_tile * BM, tid, rs_next);   // lands while this tile computes

        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        for (int c = 0; c < n_slices; ++c) {
            float *Ab = As + cur_as * (BM * APITCH);
            {
                const int kc = tid & 31, r0 = tid >> 5;
#pragma unroll
                for (int i = 0; i < NPRE; ++i) {
                    const bool ok = (c * BK + kc < a.k_total) && (row0 + r0 + 8 * i < a.m_rows);
                    Ab[(r0 + 8 * i) * APITCH + kc] = ok ? pre[i] : 0.f;
                }
            }
            const bool last = c == n_slices - 1;
            if (last && has_next) rowsrc_store(rsrc + (cur_rs ^ 1) * (MAX_BLOCKS * BM), tid, rs_next);
            __syncthreads();
            if (!last) slice_fetch(a, rsrc + cur_rs * (MAX_BLOCKS * BM), c + 1, tid, pre);
            else if (has_next) slice_fetch(a, rsrc + (cur_rs ^ 1) * (MAX_BLOCKS * BM), 0, tid, pre);

            int ksteps = (a.k_total - c * BK + 1) >> 1;
            ksteps = ksteps > BK / 2 ? BK / 2 : ksteps;
            const float *ap0 = Ab + (wm * 64 + li) * APITCH + lh;
            const float *ap1 = ap0 + 32 * APITCH;
            const float *bp0 = Wt + (c * BK + lh) * WPITCH + wn * 64 + li;
            if (ksteps == BK / 2) {
                // full slice: fully unrolled so the scheduler hoists the LDS operand reads of later k-steps above the
                // MFMAs of earlier ones (4 independent accumulators keep the matrix pipe issuing back to back)
#pragma unroll
                for (int ks = 0; ks < BK / 2; ++ks) {
                    const float a0 = ap0[2 * ks], a1 = ap1[2 * ks];
                    const float b0 = bp0[2 * ks * WPITCH], b1 = bp0[2 * ks * WPITCH + 32];
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                }
            } else {
                for (int ks = 0; ks < ksteps; ++ks) {
                    const float a0 = ap0[2 * ks], a1 = ap1[2 * ks];
                    const float b0 = bp0[2 * ks * WPITCH], b1 = bp0[2 * ks * WPITCH + 32];
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                }
            }
            cur_as ^= 1;
        }
        cur_rs ^= 1;

        // epilogue.  C layout of a 32x32 tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
        const bool full = (row0 + BM <= a.m_rows) && (n0 + BN <= a.n_out);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + li;
            const bool cok = col < a.n_out;
            const float bias = (cok && a.bias) ? a.bias[col] : 0.f;
            float mean = 0.f, scale = 1.f, shift = 0.f;
            if (cok && a.bn_scale) { mean = a.bn_mean[col]; scale = a.bn_scale[col]; shift = a.bn_shift[col]; }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int64_t rbase = row0 + wm * 64 + i * 32 + 4 * lh;
                float *op = a.out + rbase * a.n_out + col;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    if (!full && (!cok || rbase + dr >= a.m_rows)) continue;
                    const float h = acc[i][j][r] + bias;
                    if (STATS) {
                        st_sum[j] += (double)h;
                        st_sq[j] += (double)h * (double)h;
                    } else {
                        const float y = a.bn_scale ? (h - mean) * scale + shift : h;
                        op[(int64_t)dr * a.n_out] = apply_act(y, a.act);
                    }
                }
            }
        }
    }

    if (STATS) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + li;
            double s = st_sum[j], q = st_sq[j];
            s += __shfl_xor(s, 32);
            q += __shfl_xor(q, 32);
            if (lh == 0 && col < a.n_out) {
                atomicAdd(&a.stats[col], s);
                atomicAdd(&a.stats[a.n_out + col], q);
            }
        }
    }
}

}  // namespace gsn

using namespace gsn;

extern "C" int gsn_linear_fwd_hip(int64_t m_rows, int n_blocks, const gsn_block *blocks, const float *W, const float *bias,
                                  int64_t n_out, const float *bn_mean, const float *bn_scale, const float *bn_shift, int act,
                                  const int32_t *row_perm, float *out, double *stats, void *stream) {
    if (n_blocks < 1 || n_blocks > MAX_BLOCKS || !blocks || !W || n_out <= 0)
        return set_error(GSN_E_INVALID, "gsn_linear_fwd_hip: need 1..%d input blocks, W and n_out > 0", MAX_BLOCKS);
    if (!out && !stats) return set_error(GSN_E_INVALID, "gsn_linear_fwd_hip: neither out nor stats given");
    if ((bn_scale != nullptr) != (bn_shift != nullptr) || (bn_scale != nullptr) != (bn_mean != nullptr))
        return set_error(GSN_E_INVALID, "gsn_linear_fwd_hip: bn_mean, bn_scale and bn_shift go together");
    if (act < 0 || act > 3) return set_error(GSN_E_INVALID, "gsn_linear_fwd_hip: act must be 0..3");
    if (m_rows <= 0) return GSN_OK;
    LinArgs a{};
    a.m_rows = m_rows; a.n_blocks = n_blocks;
    int k_total = 0, nc = 0;
    for (int b = 0; b < n_blocks; ++b) {
        if (!blocks[b].data || blocks[b].width <= 0 || blocks[b].width > 32767)
            return set_error(GSN_E_INVALID, "gsn_linear_fwd_hip: block %d has no data or a bad width", b);
        a.bdata[b] = blocks[b].data; a.bidx[b] = blocks[b].idx; a.bidx32[b] = blocks[b].idx32; a.bwidth[b] = (int)blocks[b].width;
        for (int off = 0; off < (int)blocks[b].width; off += BK) {
            if (nc >= MAX_CHUNKS) return set_error(GSN_E_UNSUPPORTED, "gsn_linear_fwd_hip: input wider than %d chunks of %d", MAX_CHUNKS, BK);
            const int len = (int)blocks[b].width - off < BK ? (int)blocks[b].width - off : BK;
            a.cblock[nc] = (unsigned char)b; a.clen[nc] = (unsigned char)len; a.ccol[nc] = (short)off; a.cwk[nc] = (short)(k_total + off);
            ++nc;
        }
        k_total += (int)blocks[b].width;
    }
    a.n_chunks = nc; a.k_total = k_total; a.n_out = (int)n_out; a.act = act;
    a.W = W; a.bias = bias; a.bn_mean = bn_mean; a.bn_scale = bn_scale; a.bn_shift = bn_shift;
    a.row_perm = row_perm; a.out = out; a.stats = stats;
    const int64_t n_tiles = (m_rows + BM - 1) / BM;
    const int col_tiles = (int)((n_out + BN - 1) / BN);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int k_pad = (k_total + BK - 1) / BK * BK;
    const size_t lds = (size_t)k_pad * WPITCH * 4 + 2 * BM * APITCH * 4 + 2 * MAX_BLOCKS * BM * 4;
    if (lds <= 160 * 1024) {
        // persistent: one workgroup per CU (LDS-bound), grid-stride over row tiles
        int64_t gx = n_tiles < 256 ? n_tiles : 256;
        if (col_tiles > 1) gx = n_tiles < 128 ? n_tiles : 128;
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_fwd_kernel<false>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e0 == hipSuccess)
                e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_fwd_kernel<true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e0 != hipSuccess) return set_error(GSN_E_HIP, "hipFuncSetAttribute(linear_fwd_kernel): %s", hipGetErrorString(e0));
            attr_set = true;
        }
        if (stats) hipLaunchKernelGGL(linear_fwd_kernel<true>, dim3((unsigned)gx, (unsigned)col_tiles), dim3(256), lds, st, a, k_pad);
        else hipLaunchKernelGGL(linear_fwd_kernel<false>, dim3((unsigned)gx, (unsigned)col_tiles), dim3(256), lds, st, a, k_pad);
    } else {
        int64_t gx = n_tiles < 1024 ? n_tiles : 1024;
        hipLaunchKernelGGL(linear_fwd_stream_kernel, dim3((unsigned)gx, (unsigned)col_tiles), dim3(256), 0, st, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "linear_fwd_kernel: %s", hipGetErrorString(e));
    return GSN_OK;
}
