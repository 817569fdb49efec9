// Shared by the register-resident layer kernels layer_rr.hip (fp32 rows) and layer_rp.hip (exact fp16 row packs): shapes of the prepared
// buffer, the tile iterator.  Not part of the ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "layer_rr_inl.h"

namespace gsn {

constexpr int RR_TN = 32;          // nodes per tile
constexpr int RR_TE = 64;          // a tile's in-edges: a whole number of 64-row chunks where the degrees allow it
constexpr int RR_BE = 32;          // edge rows per block (the unit of the edge stage)
constexpr int RR_NKE = 5;          // 16-column chunks of the edge rows (K_e <= 80)
constexpr int RR_NSLOT = 2 * RR_NKE;   // 16-byte loads per lane and edge block
constexpr int RR_MAXROLE = 3;      // distinct row-index arrays of the edge blocks (sorted target, sorted source, perm)
constexpr int RR_HDR = 32;         // header words of the prepared buffer
constexpr unsigned RR_MAGIC = 0x52523031u;

// header of the prepared buffer (words)
enum { RRH_MAGIC = 0, RRH_EE = 1, RRH_E0 = 2, RRH_E1 = 3, RRH_EMIN = 4, RRH_BAD = 5, RRH_ACT = 6, RRH_PACK = 7 };   // RRH_PACK: 1 = the k-slot order of layer_rp.hip

template <int WB, int NKX>
struct RrShape {
    static constexpr int NKS = 2 * WB;                       // chunks of the S part of node stage 0 / of node stage 1's input
    static constexpr int NK0 = NKS + NKX;
    static constexpr int F_WE = 0;                           // fragment indices (1 KiB each): edge stage [fb][c][plane]
    static constexpr int F_W0H = F_WE + WB * RR_NKE * 2;     // node stage 0, high plane [fbo][c], c < NK0
    static constexpr int F_W0XL = F_W0H + WB * NK0;          // node stage 0, low plane of the [x | deg] chunks [fbo][cq]
    static constexpr int F_W1 = F_W0XL + WB * NKX;           // node stage 1 [fb][c][plane]
    static constexpr int F_LDS = F_W1 + WB * NKS * 2;        // fragments held in LDS
    static constexpr int F_W0SL = F_LDS;                     // node stage 0, low plane of the S chunks, in the order of use [c][fbo]: streamed from L2
    static constexpr int F_ALL = F_W0SL + NKS * WB;
    static constexpr int TAB_WORDS = 3 * 32 * WB;            // c0 of the three stages
    static constexpr int LDS_BYTES = F_LDS * 1024 + TAB_WORDS * 4 + RR_NSLOT * 2 * 16;
    static constexpr int PREP_WORDS = RR_HDR + F_ALL * 256 + TAB_WORDS;
};

// ------------------------------------------------------------------------------------------------------------------------------
// tile iterator of one wave (the one of layer_fused.hip, wave-local): tiles of <= 32 nodes whose in-edges are a whole number of
// 64-row chunks where possible, handed out as BLOCKS of <= 32 edge rows (the unit of the edge stage and of the gather pipeline)
struct RrDesc {
    int m0, e0, pk;                         // pk: valid | first << 1 | last << 2 | nn << 6 | ne << 12
    __device__ __forceinline__ int valid() const { return pk & 1; }
    __device__ __forceinline__ int first() const { return (pk >> 1) & 1; }
    __device__ __forceinline__ int last() const { return (pk >> 2) & 1; }
    __device__ __forceinline__ int nn() const { return (pk >> 6) & 63; }
    __device__ __forceinline__ int ne() const { return (pk >> 12) & 63; }
};

struct RrIter {
    const int32_t *seg;
    int n_nodes, m_next, m_end;
    int m0, nn, eb, ee, ec, pending;
    int win;                                // lane l: seg_ptr[m_next + l] (window of the NEXT tile, fetched when the current one is formed)
};

__device__ __forceinline__ void rr_iter_load(RrIter &it, int lane) {
    int idx = it.m_next + lane;
    idx = idx < it.n_nodes ? idx : it.n_nodes;
    it.win = it.seg[idx];
}

__device__ __forceinline__ RrDesc rr_iter_next(RrIter &it, int lane) {
    RrDesc d; d.m0 = 0; d.e0 = 0; d.pk = 0;
    if (!(it.pending || it.ec < it.ee)) {
        if (it.m_next >= it.m_end) return d;
        int nmax = it.m_end - it.m_next;
        nmax = nmax < RR_TN ? nmax : RR_TN;
        const int w0 = __builtin_amdgcn_readfirstlane(it.win);
        const int cnt = it.win - w0;
        const int ne_all = __builtin_amdgcn_readlane(cnt, nmax);
        int nn = nmax;
        if (ne_all > RR_TE) {
            const int cap = ne_all / RR_TE * RR_TE;
            const unsigned long long ok = __ballot(lane <= nmax && cnt <= cap);
            nn = __popcll(ok) - 1;
            nn = nn < 1 ? 1 : nn;
        }
        nn = __builtin_amdgcn_readfirstlane(nn);
        it.m0 = it.m_next; it.nn = nn; it.eb = w0; it.ee = __builtin_amdgcn_readlane(it.win, nn);
        it.ec = it.eb; it.pending = 1;
        it.m_next += nn;
        rr_iter_load(it, lane);
    }
    d.m0 = it.m0; d.e0 = it.ec;
    const int left = it.ee - it.ec;
    const int ne = left < RR_BE ? left : RR_BE;
    d.pk = 1 | ((it.ec == it.eb) << 1) | ((it.ec + RR_BE >= it.ee) << 2) | (it.nn << 6) | (ne << 12);
    it.ec += RR_BE; it.pending = 0;
    return d;
}


#ifndef RR_NW
#define RR_NW 8            // waves per workgroup (one workgroup per CU): 8 = two per SIMD, 256 registers each
#endif

}  // namespace gsn
