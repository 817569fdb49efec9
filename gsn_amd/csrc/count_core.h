// Per-lane rooted subgraph search: the arithmetic core of the counting kernel.
//
// Shared between the HIP kernel (count.hip) and a host-side logic harness used ONLY by tests
// (tests/host_harness.cpp compiles this header with g++ to check plans + search against the oracle on CPU;
// it is not a product fallback -- the product path launches the HIP kernel or fails).
//
// A "lane" owns one rooted search at a time:
//   * the target graph is a bit matrix A[n][W] of 64-bit words (LDS on device),
//   * the partial map f is packed 8 bits per level into one 64-bit register (n <= 256; 16 bits / two registers above),
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

// bit planes of the vertices' degrees inside each d-core: plane p of core d has bit v set iff bit p of |N(v) & core_d| is set (v in core_d);
// layout [CORE_MAX + 1][DEG_PLANES][W] words (n <= 768 < 2^10).  sum_{v in S} deg(v) = sum_p 2^p popc(S & plane_p).
constexpr int DEG_PLANES = 10;

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

// Packed partial map: the image of level l sits at bits [VB*l, VB*l + VB) with VB = 8 for graphs of <= 256 vertices
// (W <= 4: one 64-bit register for k <= 8) and VB = 16 above (two registers).
template <int W>
struct FVec {
    static constexpr int VB = W <= 4 ? 8 : 16;
    uint64_t lo, hi;   // hi is dead code for VB == 8
};
template <int W>
GSN_HD int fv_get(const FVec<W> &f, int l) {
    if (FVec<W>::VB == 8) return (int)((f.lo >> (8 * l)) & 0xffu);
    const uint64_t w = l < 4 ? f.lo : f.hi;
    return (int)((w >> (16 * (l & 3))) & 0xffffu);
}
template <int W>
GSN_HD void fv_set(FVec<W> &f, int l, int v) {
    if (FVec<W>::VB == 8) {
        f.lo = (f.lo & ~(0xffull << (8 * l))) | ((uint64_t)v << (8 * l));
    } else {
        const int sh = 16 * (l & 3);
        if (l < 4) f.lo = (f.lo & ~(0xffffull << sh)) | ((uint64_t)v << sh);
        else f.hi = (f.hi & ~(0xffffull << sh)) | ((uint64_t)v << sh);
    }
}
template <int W>
GSN_HD FVec<W> fv_roots(int a, int b) {
    FVec<W> f;
    f.lo = (uint64_t)a | ((uint64_t)b << FVec<W>::VB);
    f.hi = 0;
    return f;
}

// Candidate set of level l given the packed partial map fvec (levels 0..l-1 assigned) and the set `used` of their images.
//   desc = adj_mask | nonadj_mask<<8 | gt_mask<<16 | lt_mask<<24  (bit j <-> earlier level j)
//   adj    : candidate must be a neighbour of f_j          (pattern edge)
//   nonadj : candidate must NOT be a neighbour of f_j      (induced matching, pattern non-edge)
//   gt/lt  : symmetry breaking, candidate id must be > / < f_j
// Only the levels named in desc are visited (a cycle or path level has one or two), everything else is covered by
// `valid & ~used` -- the kernel is VALU-issue bound, so the per-step instruction count is what matters.
// `ball` = j | r<<3 (r = 0: none) with `balls` = the r-hop balls of the target graph, radius 2 at balls[v*W], radius 3 at
// balls[(ball_n + v)*W] (nullptr: pruning off -- it never changes the result, only the work).
// DIR (digraph patterns and targets): A holds the OUT-neighbour rows, A_in the IN-neighbour rows, and desc_in =
// in_adj_mask | in_nonadj_mask<<8 names the earlier levels whose image the candidate must (not) have an arc TO.
template <int W, bool DIR = false>
GSN_HD void candidates(Bits<W> &C, uint32_t desc, uint32_t ball, const FVec<W> &fvec, const Bits<W> &used, const uint64_t *A,
                       const uint64_t *valid, const uint64_t *balls, int ball_n, uint32_t desc_in = 0, const uint64_t *A_in = nullptr) {
#pragma unroll
    for (int w = 0; w < W; ++w) C.w[w] = valid[w] & ~used.w[w];
    if (balls && (ball >> 3)) {
        const int fj = fv_get<W>(fvec, (int)(ball & 7u));
        const uint64_t *row = balls + (size_t)(((ball >> 3) == 3 ? ball_n : 0) + fj) * W;
#pragma unroll
        for (int w = 0; w < W; ++w) C.w[w] &= row[w];
    }
    uint32_t m = desc & 0xffu;
    while (m) {
        const int j = ctz64(m);
        m &= m - 1u;
        const uint64_t *row = A + fv_get<W>(fvec, j) * W;
#pragma unroll
        for (int w = 0; w < W; ++w) C.w[w] &= row[w];
    }
    m = (desc >> 8) & 0xffu;
    while (m) {
        const int j = ctz64(m);
        m &= m - 1u;
        const uint64_t *row = A + fv_get<W>(fvec, j) * W;
#pragma unroll
        for (int w = 0; w < W; ++w) C.w[w] &= ~row[w];
    }
    if (DIR) {
        m = desc_in & 0xffu;
        while (m) {
            const int j = ctz64(m);
            m &= m - 1u;
            const uint64_t *row = A_in + fv_get<W>(fvec, j) * W;
#pragma unroll
            for (int w = 0; w < W; ++w) C.w[w] &= row[w];
        }
        m = (desc_in >> 8) & 0xffu;
        while (m) {
            const int j = ctz64(m);
            m &= m - 1u;
            const uint64_t *row = A_in + fv_get<W>(fvec, j) * W;
#pragma unroll
            for (int w = 0; w < W; ++w) C.w[w] &= ~row[w];
        }
    }
    m = desc >> 16;
    while (m) {
        const int jj = ctz64(m);
        m &= m - 1u;
        const bool lt = jj >= 8;
        const int fj = fv_get<W>(fvec, jj & 7);
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const uint64_t bl = below_word(fj, w);                       // ids < fj
            C.w[w] &= lt ? bl : ~(bl | ((w == (fj >> 6)) ? (1ull << (fj & 63)) : 0ull));
        }
    }
}

// r-hop balls of vertex v from the adjacency bit matrix: ball2 = v + N(v) + N(N(v)), ball3 = one more hop (needs every
// ball2 first).  One call per vertex; the caller parallelises over v.
template <int W>
GSN_HD void ball_expand(const uint64_t *A, const uint64_t *from /* [n][W] or nullptr: the adjacency itself */, int v, uint64_t *out) {
    Bits<W> acc;
#pragma unroll
    for (int w = 0; w < W; ++w) acc.w[w] = (from ? from[v * W + w] : A[v * W + w]) | ((w == (v >> 6)) ? (1ull << (v & 63)) : 0ull);
#pragma unroll
    for (int w = 0; w < W; ++w) {
        uint64_t m = A[v * W + w];
        while (m) {
            const int u = w * 64 + ctz64(m);
            m &= m - 1ull;
#pragma unroll
            for (int x = 0; x < W; ++x) acc.w[x] |= from ? from[u * W + x] : A[u * W + x];
        }
    }
#pragma unroll
    for (int w = 0; w < W; ++w) out[v * W + w] = acc.w[w];
}

// One peeling round of the d-core: does vertex v (a member of `core`) keep at least d neighbours inside `core`?
template <int W>
GSN_HD bool core_keeps(const uint64_t *A, const uint64_t *core, int v, int d) {
    int c = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) c += popc64(A[v * W + w] & core[w]);
    return c >= d;
}

// degree planes of ONE vertex v of core d (caller: every v of every needed core; planes zeroed before).  `orfn(word_ptr, bit)` sets a bit.
template <int W, class OrFn>
GSN_HD void deg_planes_vertex(const uint64_t *A, const uint64_t *core, int v, uint64_t *planes /* of this core */, OrFn orfn) {
    if (!((core[v >> 6] >> (v & 63)) & 1ull)) return;
    int deg = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) deg += popc64(A[v * W + w] & core[w]);
    for (int p = 0; p < DEG_PLANES; ++p)
        if ((deg >> p) & 1) orfn(planes + p * W + (v >> 6), 1ull << (v & 63));
}

// The last two levels without a closed form, in ONE tight loop instead of one lane_step per image of level k - 2: the last level's candidate
// set minus everything that names level k - 2 is made once (`base`), then every d in C1 costs an AND with its adjacency row (or its
// complement: induced non-edge), the order mask when the two levels are ordered, and a popcount.
// Where the loop pays: graphs of more than 64 vertices (W >= 2), whose candidate sets are long and whose generic step (frame load / store,
// candidates() from scratch) costs several times an iteration of the loop: ER G(128,1000), edge mode 45.7 -> 57.3 k graphs/s.  On
// molecules (W == 1, sets of one or two vertices) the loop only makes a wave's steps uneven -- the headline's counting kernel went from
// 0.249 to 0.271 ms with it -- so there level k - 2 keeps its frame and is stepped through like the levels above.
template <int W>
struct TailLoop { static constexpr bool on = W >= 2; };

template <int W, bool DIR>
GSN_HD uint64_t tail_loop(const Bits<W> &C1, const uint32_t *plan, int k, const FVec<W> &fvec, const Bits<W> &used, const uint64_t *A,
                          const uint64_t *valid, const uint64_t *balls, int ball_n, const uint64_t *A_in) {
    const int a = k - 2;
    const uint32_t bit = 1u << a, d1 = plan[2 + k - 1], d1_in = DIR ? plan[PLAN_STRIDE_WORDS + k - 1] : 0u;
    uint32_t ball = (plan[2 + GSN_KMAX + ((k - 1) >> 2)] >> (8 * ((k - 1) & 3))) & 0xffu;
    if ((ball >> 3) && (int)(ball & 7u) == a) ball = 0;          // (a distance bound never changes the result: dropped when it names level k - 2)
    const bool adj = (d1 & bit) != 0, non = ((d1 >> 8) & bit) != 0, gt = ((d1 >> 16) & bit) != 0, lt = ((d1 >> 24) & bit) != 0;
    const bool adj_in = DIR && (d1_in & bit) != 0, non_in = DIR && ((d1_in >> 8) & bit) != 0;
    Bits<W> base;
    candidates<W, DIR>(base, d1 & ~(bit | (bit << 8) | (bit << 16) | (bit << 24)), ball, fvec, used, A, valid, balls, ball_n,
                       d1_in & ~(bit | (bit << 8)), A_in);
    uint64_t cnt = 0;
#pragma unroll
    for (int wi = 0; wi < W; ++wi) {
        uint64_t m = C1.w[wi];
        while (m) {
            const int d = wi * 64 + ctz64(m);
            m &= m - 1ull;
            const uint64_t *row = A + d * W;
#pragma unroll
            for (int w = 0; w < W; ++w) {
                uint64_t c = base.w[w];
                if (adj) c &= row[w];
                if (non) c &= ~row[w];
                if (DIR) {
                    const uint64_t rin = A_in[d * W + w];
                    if (adj_in) c &= rin;
                    if (non_in) c &= ~rin;
                }
                const uint64_t self = (w == (d >> 6)) ? (1ull << (d & 63)) : 0ull;
                const uint64_t bl = below_word(d, w);
                if (gt) c &= ~(bl | self);
                if (lt) c &= bl;
                c &= ~self;
                cnt += (uint64_t)popc64(c);
            }
        }
    }
    return cnt;
}

template <int W>
struct Lane {
    int l;          // level whose frame is being consumed; < 0: no search in progress
    int k, nfix;
    FVec<W> fvec;   // partial map, VB bits per level
    uint64_t cnt;   // matches found so far for the current task (accumulates over the task's plans)
    Bits<W> used;   // images of levels 0 .. l-1
    const uint32_t *plan;
    const uint64_t *balls;   // distance-pruning tables of the target graph or nullptr
    int ball_n;              // vertex capacity of one table
    const uint64_t *degp;    // degree bit planes of the cores (plans with a chain tail) or nullptr
    int loop;                // W == 1: the last two levels in tail_loop all the same (dense small graphs: the launcher decides)
#if defined(COUNT_PROF_STEP) && defined(__HIPCC__)
    unsigned long long prof[12];   // wave-level: [0..2] tail_loop cycles / visits / active lanes, [3] its iterations summed over lanes, [4] max iterations per visit summed,
                                   // [5..7] tail_pairs cycles / visits / lanes, [8..10] rest of lane_step (candidates + frames + climb) cycles / visits / lanes
#endif
};
#if defined(COUNT_PROF_STEP) && defined(__HIP_DEVICE_COMPILE__)
#define CP_T0() const unsigned long long cp_t0 = __builtin_amdgcn_s_memtime()
#define CP_ADD(S, I) do { (S).prof[I] += __builtin_amdgcn_s_memtime() - cp_t0; (S).prof[(I) + 1] += 1; (S).prof[(I) + 2] += (unsigned long long)__popcll(__ballot(1)); } while (0)
#else
#define CP_T0()
#define CP_ADD(S, I)
#endif

// core index of a plan: images must lie in the min-degree(H) core of the target; cores 0..CORE_MAX are tabulated
// (core d for d > CORE_MAX uses core CORE_MAX, a superset)
constexpr int CORE_MAX = 4;
GSN_HD int plan_core(const uint32_t *plan) {
    const int d = (int)((plan[1] >> 20) & 0xfu);
    return d > CORE_MAX ? CORE_MAX : d;
}
GSN_HD uint32_t plan_ball(const uint32_t *plan, int l) { return (plan[2 + GSN_KMAX + (l >> 2)] >> (8 * (l & 3))) & 0xffu; }
// closed form of the last two levels (patterns.cpp: plan_tail_mode): 0 none, 1 independent candidate sets, 2 twins, 3 chain
GSN_HD int plan_tail(const uint32_t *plan) { return (int)((plan[1] >> 28) & 3u); }
// the level at which the closed form replaces the search: k - 2, or k - r for a run of r twin levels (mode 2, r = 2 .. 5)
GSN_HD int plan_tail_level(const uint32_t *plan, int k) { return ((plan[1] >> 28) & 3u) == 2u ? k - 2 - (int)(plan[1] >> 30) : k - 2; }

// Number of ways to place the last two levels given C1 = the candidates of level k - 2 (levels 0 .. k - 3 are in fvec / used):
// independent sets: |C1| |C2| - |C1 & C2| with C2 = the candidates of level k - 1 (whose constraints do not name level k - 2);
// twins: C(|C1|, 2).
template <int W, bool DIR>
GSN_HD uint64_t tail_pairs(int mode, const Bits<W> &C1, const uint32_t *plan, int k, const FVec<W> &fvec, const Bits<W> &used, const uint64_t *A,
                           const uint64_t *valid, const uint64_t *balls, int ball_n, const uint64_t *A_in, const uint64_t *degp) {
    const uint64_t n1 = (uint64_t)popc<W>(C1);
    if (mode == 2) {                                    // C(n1, r): r twin levels
        const int r = 2 + (int)(plan[1] >> 30);
        if (n1 < (uint64_t)r) return 0;
        uint64_t c = 1;
        for (int i = 0; i < r; ++i) c = c * (n1 - (uint64_t)i) / (uint64_t)(i + 1);     // (exact at every step: i + 1 consecutive integers)
        return c;
    }
    if (mode == 3) {
        // chain: sum over d in C1 of |N(d) & core \ placed| -- d itself is no neighbour of d, the placed images f_0 .. f_{k-3} are
        const uint64_t *pl = degp + (size_t)plan_core(plan) * DEG_PLANES * W;
        uint64_t s = 0;
#pragma unroll
        for (int p = 0; p < DEG_PLANES; ++p) {
            int c = 0;
#pragma unroll
            for (int w = 0; w < W; ++w) c += popc64(C1.w[w] & pl[p * W + w]);
            s += (uint64_t)c << p;
        }
        for (int j = 0; j < k - 2; ++j) {
            const uint64_t *row = A + fv_get<W>(fvec, j) * W;
            int c = 0;
#pragma unroll
            for (int w = 0; w < W; ++w) c += popc64(C1.w[w] & row[w]);
            s -= (uint64_t)c;
        }
        return s;
    }
    Bits<W> C2;
    candidates<W, DIR>(C2, plan[2 + k - 1], plan_ball(plan, k - 1), fvec, used, A, valid, balls, ball_n,
                       DIR ? plan[PLAN_STRIDE_WORDS + k - 1] : 0u, A_in);
    int both = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) both += popc64(C1.w[w] & C2.w[w]);
    return n1 * (uint64_t)popc<W>(C2) - (uint64_t)both;
}

template <int W>
GSN_HD void bit_set(Bits<W> &b, int v) {
#pragma unroll
    for (int w = 0; w < W; ++w) b.w[w] |= (w == (v >> 6)) ? (1ull << (v & 63)) : 0ull;
}
template <int W>
GSN_HD void bit_clear(Bits<W> &b, int v) {
#pragma unroll
    for (int w = 0; w < W; ++w) b.w[w] &= ~((w == (v >> 6)) ? (1ull << (v & 63)) : 0ull);
}

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
// TAIL = false compiles the closed forms and the tail loop of the last two levels out (the kernel instantiation of the molecule
// workloads, whose plans have none and whose graphs are below 65 vertices: their code and registers cost the headline kernel 7 %)
template <int W, bool DIR = false, bool TAIL = true>
GSN_HD void lane_begin(Lane<W> &s, const uint32_t *plan, const FVec<W> &fvec_roots, const uint64_t *A, const uint64_t *valid,
                       uint64_t *stack, int sstride, int tid, const uint64_t *A_in = nullptr) {
    const uint32_t h = plan[0];
    s.k = (int)(h & 0xffu);
    s.nfix = (int)((h >> 8) & 0xffu);
    s.plan = plan;
    s.fvec = fvec_roots;
    s.l = -1;
#pragma unroll
    for (int w = 0; w < W; ++w) s.used.w[w] = 0ull;
    bit_set<W>(s.used, fv_get<W>(fvec_roots, 0));
    if (s.nfix > 1) bit_set<W>(s.used, fv_get<W>(fvec_roots, 1));
    {   // a root outside the plan's core cannot be the image of anything
        const int ra = fv_get<W>(fvec_roots, 0);
        bool in = (valid[ra >> 6] >> (ra & 63)) & 1ull;
        if (s.nfix > 1) { const int rb = fv_get<W>(fvec_roots, 1); in = in && ((valid[rb >> 6] >> (rb & 63)) & 1ull); }
        if (!in) return;
    }
    if (s.nfix == s.k) { s.cnt += 1; return; }
    Bits<W> C;
    candidates<W, DIR>(C, plan[2 + s.nfix], plan_ball(plan, s.nfix), s.fvec, s.used, A, valid, s.balls, s.ball_n,
                       DIR ? plan[PLAN_STRIDE_WORDS + s.nfix] : 0u, A_in);
    if (s.nfix == s.k - 1) { s.cnt += (uint64_t)popc<W>(C); return; }
    if (TAIL && plan_tail(plan) == 2 && s.nfix == plan_tail_level(plan, s.k)) {            // a run of twin levels: C(|C|, r)
        s.cnt += tail_pairs<W, DIR>(2, C, plan, s.k, s.fvec, s.used, A, valid, s.balls, s.ball_n, A_in, s.degp);
        return;
    }
    if (TAIL && s.nfix == s.k - 2 && (TailLoop<W>::on || s.loop || plan_tail(plan))) {     // the last two levels: closed form or one tight loop, no frame
        const int tm = plan_tail(plan);
        s.cnt += tm ? tail_pairs<W, DIR>(tm, C, plan, s.k, s.fvec, s.used, A, valid, s.balls, s.ball_n, A_in, s.degp)
                    : tail_loop<W, DIR>(C, plan, s.k, s.fvec, s.used, A, valid, s.balls, s.ball_n, A_in);
        return;
    }
    bool cempty = true;
#pragma unroll
    for (int w = 0; w < W; ++w) cempty = cempty && (C.w[w] == 0ull);
    if (cempty) return;  // invariant: the frame of the current level is never empty when lane_step runs
    s.l = s.nfix;
    frame_store<W>(stack, sstride, tid, s.l, C);
}

// One search step.  Precondition: s.l >= 0.  Pops one candidate of the current level; at the last-but-one level the
// whole last level is counted by popcount.  A level that runs empty backtracks in the same step (no wasted iteration).
template <int W, bool DIR = false, bool TAIL = true>
GSN_HD void lane_step(Lane<W> &s, const uint64_t *A, const uint64_t *valid, uint64_t *stack, int sstride, int tid,
                      const uint64_t *A_in = nullptr) {
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
    bool empty = true;
#pragma unroll
    for (int w = 0; w < W; ++w) empty = empty && (M.w[w] == 0ull);
    // v >= 0 always: a frame is only ever stored non-empty or left through the `empty` path below
    const int l = s.l;
    fv_set<W>(s.fvec, l, v);
    const int nl = l + 1;
    Bits<W> used2 = s.used;
    bit_set<W>(used2, v);
    Bits<W> C;
    candidates<W, DIR>(C, s.plan[2 + nl], plan_ball(s.plan, nl), s.fvec, used2, A, valid, s.balls, s.ball_n,
                       DIR ? s.plan[PLAN_STRIDE_WORDS + nl] : 0u, A_in);
    bool descend = false;
    if (nl == s.k - 1) {
        s.cnt += (uint64_t)popc<W>(C);
    } else if (TAIL && plan_tail(s.plan) == 2 && nl == plan_tail_level(s.plan, s.k)) {
        s.cnt += tail_pairs<W, DIR>(2, C, s.plan, s.k, s.fvec, used2, A, valid, s.balls, s.ball_n, A_in, s.degp);      // r twin levels: C(|C|, r)
    } else if (TAIL && nl == s.k - 2 && (TailLoop<W>::on || s.loop || plan_tail(s.plan))) {
        // the last two levels: in closed form where the plan allows it (level k - 2 is not enumerated), else in one tight loop over its images
        const int tm = plan_tail(s.plan);
        if (tm) {
            CP_T0();
            s.cnt += tail_pairs<W, DIR>(tm, C, s.plan, s.k, s.fvec, used2, A, valid, s.balls, s.ball_n, A_in, s.degp);
            CP_ADD(s, 5);
        } else {
            CP_T0();
            s.cnt += tail_loop<W, DIR>(C, s.plan, s.k, s.fvec, used2, A, valid, s.balls, s.ball_n, A_in);
            CP_ADD(s, 0);
#if defined(COUNT_PROF_STEP) && defined(__HIP_DEVICE_COMPILE__)
            {   // (inside a divergent branch: ballots over the active lanes, no shuffles)
                const int it = popc<W>(C);
                unsigned long long alive = __ballot(1), sm = 0;
                int mx = 0;
                for (int b = 10; b >= 0; --b) {
                    const unsigned long long hb = __ballot((it >> b) & 1);
                    sm += (unsigned long long)__popcll(hb) << b;
                    if (hb & alive) { mx |= 1 << b; alive &= hb; }
                }
                s.prof[3] += sm; s.prof[4] += (unsigned long long)mx;
            }
#endif
        }
    } else {
        bool cempty = true;
#pragma unroll
        for (int w = 0; w < W; ++w) cempty = cempty && (C.w[w] == 0ull);
        descend = !cempty;
    }
    if (descend) {
        frame_store<W>(stack, sstride, tid, l, M);   // possibly empty: the climb below finds it so on the way back
        s.used = used2;
        s.l = nl;
        frame_store<W>(stack, sstride, tid, nl, C);
        return;
    }
    if (!empty) {
        frame_store<W>(stack, sstride, tid, l, M);
        return;
    }
    // this level is exhausted: climb to the nearest level that still has candidates
    int cl = l - 1;
    while (cl >= s.nfix) {
        bit_clear<W>(s.used, fv_get<W>(s.fvec, cl));
        Bits<W> P;
        frame_load<W>(stack, sstride, tid, cl, P);
        bool pe = true;
#pragma unroll
        for (int w = 0; w < W; ++w) pe = pe && (P.w[w] == 0ull);
        if (!pe) break;
        --cl;
    }
    s.l = cl >= s.nfix ? cl : -1;
}

}  // namespace gsn
