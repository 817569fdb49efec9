// TEST-ONLY logic harness (not product code, never shipped in libgsn_hip.so).
// Compiles the kernel's per-lane search core (gsn_amd/csrc/count_core.h) and the plan compiler
// (gsn_amd/csrc/patterns.cpp) for the HOST and runs every (column,row) task sequentially, so that plan
// compilation + symmetry breaking + the search itself can be checked against the oracle on a machine
// without a GPU.  The HIP kernel's own setup phases are covered by the -m gpu tests.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../gsn_amd/csrc/count_core.h"

namespace gsn {
int set_error(int code, const char *fmt, ...) {
    va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr);
    return code;
}
}
using namespace gsn;

template <int W, bool DIR>
static int run(const uint32_t *plan, int64_t n, int64_t E, const int64_t *src, const int64_t *dst, int64_t *out) {
    const int mode = (int)plan[1], n_cols = (int)plan[4], kmax = (int)plan[5], plans_off = (int)plan[7];
    const int stride = plan_stride(plan[6]);
    std::vector<uint64_t> A_in(DIR ? (size_t)(n ? n : 1) * W : 1, 0);
    const uint32_t *col_ptr = plan + PLAN_HEADER_WORDS, *plans = plan + plans_off;
    std::vector<uint64_t> A((size_t)(n ? n : 1) * W, 0), stack((size_t)kmax * W, 0);
    int n_active = 0;
    for (int64_t c = 0; c < E; ++c) {
        int u = (int)src[c], v = (int)dst[c];
        n_active = std::max(n_active, std::max(u, v) + 1);
        if (u == v) continue;
        A[(size_t)u * W + (v >> 6)] |= 1ull << (v & 63);
        (DIR ? A_in : A)[(size_t)v * W + (u >> 6)] |= 1ull << (u & 63);
    }
    uint64_t valid[W];
    for (int w = 0; w < W; ++w) valid[w] = below_word(n_active, w);
    const int nb = (int)(n ? n : 1);
    std::vector<uint64_t> balls((size_t)2 * nb * W, 0);           // radius 2, then radius 3
    for (int v = 0; v < (int)n; ++v) ball_expand<W>(A.data(), nullptr, v, balls.data());
    for (int v = 0; v < (int)n; ++v) ball_expand<W>(A.data(), balls.data(), v, balls.data() + (size_t)nb * W);
    const bool prune = getenv("GSN_HARNESS_NO_PRUNE") == nullptr && !DIR;
    // d-cores, d = 0 .. CORE_MAX (plan_core): candidate universe of a plan
    std::vector<uint64_t> cores((size_t)(CORE_MAX + 1) * W, 0);
    for (int d = 0; d <= CORE_MAX; ++d) {
        uint64_t *core = cores.data() + (size_t)d * W;
        for (int w = 0; w < W; ++w) core[w] = valid[w];
        for (bool changed = prune && d > 0; changed;) {
            changed = false;
            std::vector<int> drop;
            for (int v = 0; v < (int)n; ++v)
                if (((core[v >> 6] >> (v & 63)) & 1ull) && !core_keeps<W>(A.data(), core, v, d)) drop.push_back(v);
            for (int v : drop) { core[v >> 6] &= ~(1ull << (v & 63)); changed = true; }
        }
    }
    // degree bit planes of every core (count_core.h: tail_pairs, mode 3)
    std::vector<uint64_t> degp((size_t)(CORE_MAX + 1) * DEG_PLANES * W, 0);
    for (int d = 0; d <= CORE_MAX; ++d)
        for (int v = 0; v < (int)n; ++v)
            deg_planes_vertex<W>(A.data(), cores.data() + (size_t)d * W, v, degp.data() + (size_t)d * DEG_PLANES * W,
                                 [](uint64_t *wp, uint64_t bit) { *wp |= bit; });
    std::vector<int64_t> last((size_t)(n * n ? n * n : 1), -1);
    for (int64_t c = 0; c < E; ++c) last[(size_t)src[c] * n + dst[c]] = c;
    const int64_t rows = mode == GSN_MODE_EDGE ? E : n;
    int status = 0;
    for (int col = 0; col < n_cols; ++col)
        for (int64_t row = 0; row < rows; ++row) {
            Lane<W> s; s.l = -1; s.cnt = 0; s.k = 0; s.nfix = 0; s.fvec = fv_roots<W>(0, 0); s.plan = plans;
            s.balls = prune ? balls.data() : nullptr; s.ball_n = nb; s.degp = degp.data(); s.loop = getenv("GSN_HARNESS_TAIL_LOOP") ? 1 : 0;
            for (int w = 0; w < W; ++w) s.used.w[w] = 0;
            FVec<W> roots = fv_roots<W>(0, 0); bool live = true, rev_missing = false;
            if (mode == GSN_MODE_EDGE) {
                int u = (int)src[row], v = (int)dst[row];
                live = u != v && last[(size_t)u * n + v] == row;
                rev_missing = u != v && last[(size_t)v * n + u] < 0;
                roots = fv_roots<W>(u, v);
            } else {
                live = row < n_active; roots = fv_roots<W>((int)row, 0);
            }
            if (live)
                for (uint32_t p = col_ptr[col]; p < col_ptr[col + 1]; ++p) {
                    const uint64_t *pv = cores.data() + (size_t)plan_core(plans + p * stride) * W;
                    lane_begin<W, DIR>(s, plans + p * stride, roots, A.data(), pv, stack.data(), 1, 0, A_in.data());
                    while (s.l >= 0) lane_step<W, DIR>(s, A.data(), pv, stack.data(), 1, 0, A_in.data());
                }
            out[row * n_cols + col] = (int64_t)s.cnt;
            if (rev_missing && s.cnt) status = 1;
        }
    return status;
}

template <bool DIR>
static int run_w(const uint32_t *plan, int64_t n, int64_t E, const int64_t *src, const int64_t *dst, int64_t *out) {
    if (n <= 64) return run<1, DIR>(plan, n, E, src, dst, out);
    if (n <= 128) return run<2, DIR>(plan, n, E, src, dst, out);
    if (n <= 256) return run<4, DIR>(plan, n, E, src, dst, out);
    if (n <= 512) return run<8, DIR>(plan, n, E, src, dst, out);
    if (n <= 768) return run<12, DIR>(plan, n, E, src, dst, out);
    return -1;
}

extern "C" int harness_count(const uint32_t *plan, int64_t n, int64_t E, const int64_t *src, const int64_t *dst, int64_t *out) {
    return (plan[6] & 2u) ? run_w<true>(plan, n, E, src, dst, out) : run_w<false>(plan, n, E, src, dst, out);
}
