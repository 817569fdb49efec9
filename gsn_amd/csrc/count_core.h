// Per-lane rooted subgraph search: the arithmetic core of the counting kernel.
//
// Shared between the HIP kernel (count.hip) and a host-side logic harness used ONLY by tests
// (tests/host_harness.cpp compiles this header with g++ to check plans + search against the oracle on CPU;
// it is not a product fallback -- the product path launches the HIP kernel or fails).
//
// A "lane" owns one rooted search at a time:
//   * the target graph is a bit matrix A[n][W] of 64-bit words (LDS on device),
//   * the partial map f is packed 8 bits per level into one 64-bit register (n <= 256),
//   * a frame per level holds the not-yet-tried candidates of that level (LDS stack on device, lane-interleaved),
//   * one call to lane_step() pops one candidate, and either descends one level or -- at the last-but-one
//     level -- adds popcount(candidates of the last level) to the lane's counter (the last level is never
//     enumerated one by one).
// Everything is integer / bit arithmetic; results are exact.
#pragma once

#include <stdint.h>

#include "gsn_internal.h"

#if defined(__HIPCC__)
#define GSN_HD __host__ __device__ __forceinline__
#else
#define GSN_HD inline
#endif

namespace gsn {

GSN_HD int popc64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}
GSN_HD int ctz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffsll((unsigned long long)x) - 1;
#else
    return __builtin_ctzll(x);
#endif
}

// bits strictly below position p, restricted to word wi
GSN_HD uint64_t below_word(int p, int wi) {
    const int pw = p >> 6;
    return wi < pw ? ~0ull : (wi == pw ? ((1ull << (p & 63)) - 1ull) : 0ull);
}

template <int W>
struct Bits {
    uint64_t w[W];
};

template <int W>
GSN_HD int popc(const Bits<W> &b) {
    int c = 0;
#pragma unroll
    for (int i = 0; i < W; ++i) c += popc64(b.w[i]);
    return c;
}

// Candidate set of level l given the packed partial map fvec (levels 0..l-1 assigned).
//   desc = adj_mask | nonadj_mask<<8 | gt_mask<<16 | lt_mask<<24  (bit j <-> earlier level j)
//   adj    : candidate must be a neighbour of f_j          (pattern edge)
//   nonadj : candidate must NOT be a neighbour of f_j      (induced matching, pattern non-edge)
//   gt/lt  : symmetry breaking, candidate id must be > / < f_j
// Every earlier f_j is excluded (injectivity).
template <int W>
GSN_HD void candidates(Bits<W> &C, uint32_t desc, int l, uint64_t fvec, const uint64_t *A, const uint64_t *valid) {
#pragma unroll
    for (int w = 0; w < W; ++w) C.w[w] = valid[w];
#pragma unroll
    for (int j = 0; j < GSN_KMAX - 1; ++j) {
        if (j < l) {
            const int fj = (int)((fvec >> (8 * j)) & 0xffu);
            const bool a = (desc >> j) & 1u, na = (desc >> (8 + j)) & 1u;
            const bool gt = (desc >> (16 + j)) & 1u, lt = (desc >> (24 + j)) & 1u;
            const uint64_t *row = A + fj * W;
#pragma unroll
            for (int w = 0; w < W; ++w) {
                uint64_t m = ~0ull;
                if (a | na) {
                    const uint64_t r = row[w];
                    m = a ? r : ~r;
                }
                const uint64_t bl = below_word(fj, w);
                const uint64_t self = (w == (fj >> 6)) ? (1ull << (fj & 63)) : 0ull;
                m &= ~self;
                if (gt) m &= ~(bl | self);
                if (lt) m &= bl;
                C.w[w] &= m;
            }
        }
    }
}

struct Lane {
    int l;          // level whose frame is being consumed; < 0: no search in progress
    int k, nfix;
    uint64_t fvec;  // partial map, 8 bits per level
    uint64_t cnt;   // matches found so far for the current task (accumulates over the task's plans)
    const uint32_t *plan;
};

// frame addressing: word w of level l of lane `tid` lives at stack[((l * W + w) * sstride) + tid]
template <int W>
GSN_HD void frame_store(uint64_t *stack, int sstride, int tid, int l, const Bits<W> &b) {
#pragma unroll
    for (int w = 0; w < W; ++w) stack[(l * W + w) * sstride + tid] = b.w[w];
}
template <int W>
GSN_HD void frame_load(const uint64_t *stack, int sstride, int tid, int l, Bits<W> &b) {
#pragma unroll
    for (int w = 0; w < W; ++w) b.w[w] = stack[(l * W + w) * sstride + tid];
}

// Start the rooted search of `plan` with the root levels already in fvec.  May finish immediately (s.l < 0).
template <int W>
GSN_HD void lane_begin(Lane &s, const uint32_t *plan, uint64_t fvec_roots, const uint64_t *A, const uint64_t *valid,
                       uint64_t *stack, int sstride, int tid) {
    const uint32_t h = plan[0];
    s.k = (int)(h & 0xffu);
    s.nfix = (int)((h >> 8) & 0xffu);
    s.plan = plan;
    s.fvec = fvec_roots;
    s.l = -1;
    if (s.nfix == s.k) { s.cnt += 1; return; }
    Bits<W> C;
    candidates<W>(C, plan[2 + s.nfix], s.nfix, s.fvec, A, valid);
    if (s.nfix == s.k - 1) { s.cnt += (uint64_t)popc<W>(C); return; }
    s.l = s.nfix;
    frame_store<W>(stack, sstride, tid, s.l, C);
}

// One search step.  Precondition: s.l >= 0.
template <int W>
GSN_HD void lane_step(Lane &s, const uint64_t *A, const uint64_t *valid, uint64_t *stack, int sstride, int tid) {
    Bits<W> M;
    frame_load<W>(stack, sstride, tid, s.l, M);
    int v = -1;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        if (v < 0 && M.w[w]) {
            v = w * 64 + ctz64(M.w[w]);
            M.w[w] &= M.w[w] - 1ull;
        }
    }
    if (v < 0) {  // level exhausted: backtrack
        s.l -= 1;
        if (s.l < s.nfix) s.l = -1;
        return;
    }
    frame_store<W>(stack, sstride, tid, s.l, M);
    s.fvec = (s.fvec & ~(0xffull << (8 * s.l))) | ((uint64_t)v << (8 * s.l));
    const int nl = s.l + 1;
    Bits<W> C;
    candidates<W>(C, s.plan[2 + nl], nl, s.fvec, A, valid);
    if (nl == s.k - 1) {
        s.cnt += (uint64_t)popc<W>(C);
    } else {
        s.l = nl;
        frame_store<W>(stack, sstride, tid, nl, C);
    }
}

}  // namespace gsn
