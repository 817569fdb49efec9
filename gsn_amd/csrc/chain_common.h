// Pieces of the fused MLP-chain kernel (chain.hip) kept in a header: argument block, row-source tables, column maps,
// LDS-only barrier.
#pragma once

#include <hip/hip_runtime.h>

#include "gsn_internal.h"

namespace gsn {

constexpr int CBM = 64;         // rows per tile
constexpr int CHK = 16;         // k per register chunk (8 k-steps of 2)
constexpr int CMAX_BLOCKS = 6;  // input blocks over all stages
constexpr int CMAX_STAGES = 2;  // 3 stages x 80 weight registers per lane would spill

struct ChainStage {
    const float *W, *bias, *bn_mean, *bn_scale, *bn_shift;
    int k_total, k_hbm, n_out, act;
    int first_block, n_blocks;
};

struct ChainArgs {
    int64_t m_rows;
    int n_stages, n_blocks;
    const float *bdata[CMAX_BLOCKS];
    const int64_t *bidx[CMAX_BLOCKS];
    const int32_t *bidx32[CMAX_BLOCKS];
    int bwidth[CMAX_BLOCKS];
    ChainStage st[CMAX_STAGES];
    const int32_t *row_perm;
    const int32_t *seg_target;  // [m_rows] target segment of every tile-space row (rows sorted by target) or null
    float *out;                 // [m_rows][n_out], or [n_seg][n_out] segment sums when seg_target is given
    double *stats;              // statistics of the LAST stage's pre-BN values instead of an output
    int pitch;                  // LDS row pitch in floats (odd)
    int dbg;                    // ablation switches for profiling (env GSN_CHAIN_DBG): 2 no MFMA, 4 no output, 8 no stores of the fused scatter-add
};

constexpr int RS_STRIDE = (CMAX_BLOCKS + 1) * CBM + 2;  // per slot: row sources per block, row targets, prev / next target

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Workgroup barrier that orders LDS only.  __syncthreads() is fence + barrier and drains vmcnt(0) first, i.e. it would
// wait for the just-issued prefetch loads of the next tile and for the epilogue's global stores at every stage
// boundary (measured: SQ_WAIT_ANY 37-48 % of wave cycles).  The tile buffers are LDS, so lgkmcnt(0) is all that is
// needed; registers fed by global loads are still guarded by the compiler's own counted vmcnt waits at their first use.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float chain_act(float y, int act) {
    switch (act) {
        case 1: return y > 0.f ? y : 0.f;
        case 2: return y > 0.f ? y : expm1f(y);
        case 3: return tanhf(y);
        default: return y;
    }
}

// which block / column of stage `s` does concatenated column kc belong to
struct ColMap {
    const float *base;  // bdata[blk] + col
    int bw;             // row stride of that block
    int rsoff;          // blk * CBM: offset into the row-source table
    bool ok;
};

__device__ __forceinline__ ColMap col_map(const ChainArgs &a, int s, int kc) {
    ColMap m;
    const ChainStage &st = a.st[s];
    int blk = st.first_block, col = kc;
    m.ok = kc < st.k_hbm;
#pragma unroll
    for (int b = 0; b < CMAX_BLOCKS - 1; ++b) {
        if (b >= st.first_block && b < st.first_block + st.n_blocks - 1 && blk == b && col >= a.bwidth[b]) { col -= a.bwidth[b]; blk = b + 1; }
    }
    if (!m.ok) { blk = st.n_blocks > 0 ? st.first_block : 0; col = 0; }
    const float *bd = a.bdata[0];
    int bw = a.bwidth[0];
#pragma unroll
    for (int b = 1; b < CMAX_BLOCKS; ++b)
        if (blk == b) { bd = a.bdata[b]; bw = a.bwidth[b]; }
    m.base = bd + col;
    m.bw = bw;
    m.rsoff = blk * CBM;
    return m;
}

// Row sources (which row of each input block feeds tile row r) are resolved by all threads of the workgroup: thread t owns
// tile row t&63 of block t>>6 (and of block (t>>6)+4 in a 4-wave workgroup); its index pointers are chosen once, before
// the tile loop, and live in VGPRs -- looping over the block table per row instead keeps ~40 kernarg pointers in SGPRs
// and spills them to VGPR lanes.
template <int NP>
struct RowSrcThread {
    const int32_t *p32[NP];
    const int64_t *p64[NP];
    bool on[NP];       // block exists
    const int32_t *perm, *seg;
    int64_t m_rows;
    int b[NP], r;
};
template <int NP>
struct RowSrcC {
    int v[NP];
    int tg, edge;  // (threads of block 0 only) target segment of the row; row 0 / 63: target of the row before / after the tile
};

template <int NP>
__device__ __forceinline__ RowSrcThread<NP> rs_thread(const ChainArgs &a, int tid) {
    RowSrcThread<NP> t;
    t.r = tid & (CBM - 1);
    t.perm = a.row_perm; t.seg = a.seg_target; t.m_rows = a.m_rows;
#pragma unroll
    for (int h = 0; h < NP; ++h) {
        const int b = (tid >> 6) + 4 * h;
        t.b[h] = b;
        t.on[h] = b < a.n_blocks;
        t.p32[h] = nullptr; t.p64[h] = nullptr;
#pragma unroll
        for (int q = 0; q < CMAX_BLOCKS; ++q)
            if (q == b) { t.p32[h] = a.bidx32[q]; t.p64[h] = a.bidx[q]; }
    }
    return t;
}

template <int NP>
__device__ __forceinline__ void rs_fetch(const RowSrcThread<NP> &t, int64_t row0, RowSrcC<NP> &rs) {
    const int64_t grow = row0 + t.r;
    const bool ok = grow < t.m_rows;
    int64_t logical = 0;
    if (ok) logical = t.perm ? (int64_t)t.perm[grow] : grow;
#pragma unroll
    for (int h = 0; h < NP; ++h) {
        int r = -1;
        if (t.on[h] && ok) r = t.p32[h] ? t.p32[h][logical] : (t.p64[h] ? (int)t.p64[h][logical] : (int)logical);
        rs.v[h] = r;
    }
    rs.tg = -1; rs.edge = -2;
    if (t.seg && t.b[0] == 0) {
        if (ok) rs.tg = t.seg[grow];
        if (t.r == 0 && row0 > 0 && row0 - 1 < t.m_rows) rs.edge = t.seg[row0 - 1];
        if (t.r == CBM - 1 && row0 + CBM < t.m_rows) rs.edge = t.seg[row0 + CBM];
    }
}

template <int NP>
__device__ __forceinline__ void rs_store(int *dst, const RowSrcThread<NP> &t, const RowSrcC<NP> &rs) {
#pragma unroll
    for (int h = 0; h < NP; ++h)
        if (t.on[h]) dst[t.b[h] * CBM + t.r] = rs.v[h];
    if (t.b[0] == 0) {
        dst[CMAX_BLOCKS * CBM + t.r] = rs.tg;
        if (t.r == 0) dst[(CMAX_BLOCKS + 1) * CBM] = rs.edge;
        if (t.r == CBM - 1) dst[(CMAX_BLOCKS + 1) * CBM + 1] = rs.edge;
    }
}

// chain_pipe.hip: stage-pipelined launch of two-stage chains with plain output; returns 1 when the shape is not covered
int launch_chain2_pipe(const ChainArgs &a, int maxch, hipStream_t st);
// chain_pipe_bf16.hip: the same on bf16 MFMAs with exactly split operands (6 plane products); returns 1 when not covered
int launch_chain2_pipe_bf16(const ChainArgs &a, int maxch, hipStream_t st);
// chain_seg.hip: role-pipelined launch of single-stage chains with the fused scatter-add; returns 1 when not covered
int launch_chain1_seg(const ChainArgs &a, int maxch, hipStream_t st);
// chain_seg_bf16.hip: the same on bf16 MFMAs with exactly split operands (6 plane products); returns 1 when not covered
int launch_chain1_seg_bf16(const ChainArgs &a, int maxch, hipStream_t st);

// GSN_CHAIN_TRACE=1: one stderr line per chain launch naming the kernel variant (used by the parity tests' coverage check)
void chain_trace(const char *kernel, const ChainArgs &a);

constexpr int SEG_ROWS = GSN_SEG_RANGE_ROWS;  // rows per reduction range of the segmented-sum epilogue

}  // namespace gsn
