// Host-side pattern analysis and counting-plan compiler (C ABI: gsn_pattern_orbits, gsn_count_plan_build).
//
// Product code.  Mirrors the *results* of utils_graph_processing.automorphism_orbits (:10-56) and
// induced_edge_automorphism_orbits (:58-100) of the reference, computed here by a direct permutation search over the
// <= 8 pattern vertices (no graph-tool), and compiles each pattern into rooted search plans for the HIP kernel:
//
//   vertex mode: counts[v, o] = #occurrences of H that contain v at a position of vertex orbit o
//                             = #maps f with f(rep_o) = v that satisfy the symmetry-breaking constraints of Stab(rep_o)
//   edge   mode: counts[(u,v), c] = sum over directed-edge orbits Omega (under Aut(H)) inside edge class c of
//                               #maps f with f(a)=u, f(b)=v, (a,b)=rep(Omega), satisfying the constraints of Stab(a,b)
//
// Both identities follow from orbit-stabiliser (DESIGN.md "Counting identity"); every quantity is an exact integer, so
// the result is bit-identical to "enumerate all |Aut| maps per occurrence, then divide by |Aut|" (what the reference does).
#include "gsn_internal.h"

#include <algorithm>
#include <array>
#include <cstring>
#include <vector>

namespace gsn {

struct Pattern {
    int k = 0;
    bool directed = false;                       // main.py --directed: the rows of the edge list are arcs
    uint16_t adj[GSN_KMAX] = {0};                // adj[i] bit j: edge {i,j} (both rows) or, directed, arc i -> j
    std::vector<std::array<uint8_t, GSN_KMAX>> aut;  // all automorphisms sigma: sigma[i] = image of i
    int vorbit[GSN_KMAX] = {0};
    int n_vorbits = 0;
    std::vector<std::pair<int, int>> arcs;       // sorted directed edges
    std::vector<int> arc_class;                  // reference's "induced" edge orbit id per arc
    int n_classes = 0;
};

static bool has(const Pattern &P, int i, int j) { return (P.adj[i] >> j) & 1; }
static bool linked(const Pattern &P, int i, int j) { return has(P, i, j) || has(P, j, i); }   // adjacent in the underlying graph

// all edge- (arc-) preserving bijections V(H)->V(H)  (= non-induced self-monomorphisms, which for equal edge counts are automorphisms)
static void enum_aut(const Pattern &P, int l, std::array<uint8_t, GSN_KMAX> &sigma, unsigned used,
                     std::vector<std::array<uint8_t, GSN_KMAX>> &out) {
    if (l == P.k) { out.push_back(sigma); return; }
    for (int v = 0; v < P.k; ++v) {
        if ((used >> v) & 1) continue;
        bool ok = true;
        for (int j = 0; j < l && ok; ++j)
            if ((has(P, l, j) && !has(P, v, sigma[j])) || (has(P, j, l) && !has(P, sigma[j], v))) ok = false;
        if (!ok) continue;
        sigma[l] = (uint8_t)v;
        enum_aut(P, l + 1, sigma, used | (1u << v), out);
    }
}

// flags: bit 0 = directed_orbits (edge classes keep the order of the two vertex orbits), bit 1 = directed (digraph pattern)
static int analyse(int64_t n_edges, const int64_t *edges, int flags, Pattern &P) {
    const int directed_orbits = flags & 1;
    P.directed = (flags & 2) != 0;
    int64_t mx = -1;
    for (int64_t i = 0; i < 2 * n_edges; ++i) {
        if (edges[i] < 0) return set_error(GSN_E_INVALID, "pattern vertex id < 0");
        mx = std::max(mx, edges[i]);
    }
    if (mx + 1 > GSN_KMAX) return set_error(GSN_E_UNSUPPORTED, "pattern has %lld vertices; this build handles k <= %d", (long long)(mx + 1), GSN_KMAX);
    if (mx < 0) return set_error(GSN_E_INVALID, "empty pattern edge list");
    P.k = (int)mx + 1;
    for (int64_t i = 0; i < n_edges; ++i) {
        int u = (int)edges[2 * i], v = (int)edges[2 * i + 1];
        if (u == v) continue;  // gt.stats.remove_self_loops
        P.adj[u] |= (uint16_t)(1u << v);
        if (!P.directed) P.adj[v] |= (uint16_t)(1u << u);
    }
    std::array<uint8_t, GSN_KMAX> sigma{};
    enum_aut(P, 0, sigma, 0, P.aut);
    // vertex orbits: orbit id = rank of the orbit's smallest vertex (np.unique(..., return_inverse) on the per-vertex minima)
    int minrep[GSN_KMAX];
    for (int v = 0; v < P.k; ++v) minrep[v] = v;
    for (auto &s : P.aut)
        for (int i = 0; i < P.k; ++i) minrep[s[i]] = std::min(minrep[s[i]], i);
    std::vector<int> reps(minrep, minrep + P.k);
    std::sort(reps.begin(), reps.end());
    reps.erase(std::unique(reps.begin(), reps.end()), reps.end());
    P.n_vorbits = (int)reps.size();
    for (int v = 0; v < P.k; ++v) P.vorbit[v] = (int)(std::lower_bound(reps.begin(), reps.end(), minrep[v]) - reps.begin());
    // sorted directed edge list + first-seen class numbering
    std::vector<std::pair<int, int>> keys;
    for (int u = 0; u < P.k; ++u)
        for (int v = 0; v < P.k; ++v) {
            if (!has(P, u, v)) continue;
            P.arcs.push_back({u, v});
            int a = P.vorbit[u], b = P.vorbit[v];
            if (!directed_orbits && a > b) std::swap(a, b);
            auto it = std::find(keys.begin(), keys.end(), std::make_pair(a, b));
            if (it == keys.end()) { keys.push_back({a, b}); it = keys.end() - 1; }
            P.arc_class.push_back((int)(it - keys.begin()));
        }
    P.n_classes = (int)keys.size();
    return GSN_OK;
}

// ---- plan compilation -----------------------------------------------------------------------------------------

struct Plan {
    int k, n_fixed, out_col, pattern, root_a, root_b, min_degree;
    uint32_t level[GSN_KMAX];
    uint32_t level_in[GSN_KMAX];     // directed patterns: the constraints through arcs that LEAVE the level's vertex
    uint8_t ball[GSN_KMAX];
};

// Closed form for the LAST TWO levels (count_core.h, lane_step): when the last level's constraints do not name the level before it, the
// two candidate sets C1, C2 are both known once the levels above are placed, and the number of (u, v), u in C1, v in C2, u != v, is
// |C1| |C2| - |C1 & C2| -- the last-but-one level need not be enumerated.  Two pendant vertices of a pattern (a star's leaves, the two ends
// of a path rooted in its middle) end a matching order this way.  When the only link is ONE symmetry-breaking order constraint and the
// two levels are otherwise constrained alike (twins under the root's stabiliser), the count is C(|C1|, 2).
// When the last level is a pendant vertex hanging off the level before it and carries no other constraint (a path's far end, the tail of a
// tadpole), its candidate set for the image d of level k - 2 is N(d) within the plan's core minus the images placed so far, so the
// placements of both levels are  sum_{d in C1} deg_core(d)  -  sum_{j placed} |N(f_j) & C1|  -- a weighted popcount of C1 over the bit
// planes of the core degrees (count_core.h: DEG_PLANES, built once per graph) and one AND + popcount per placed level.
//   0 = none, 1 = independent, 2 = twins, 3 = chain
// are levels a and b = a + 1 twins: linked by exactly one order constraint, otherwise constrained alike (level b may omit bounds on the link's
// side that level a has: v > u > f_j makes them hold)?  Returns +1 (b > a), -1 (b < a) or 0.
static int twin_link(const Plan &pl, bool directed, int a) {
    const int b = a + 1;
    const uint32_t bit = 1u << a, d1 = pl.level[b], d2 = pl.level[a];
    const bool ref_adj = (d1 & bit) != 0, ref_non = ((d1 >> 8) & bit) != 0, ref_gt = ((d1 >> 16) & bit) != 0, ref_lt = ((d1 >> 24) & bit) != 0;
    const bool ball_ref = (pl.ball[b] >> 3) != 0 && (pl.ball[b] & 7) == a;
    const bool in_ref = directed && ((pl.level_in[b] & bit) != 0 || ((pl.level_in[b] >> 8) & bit) != 0);
    if (ref_adj || ref_non || ball_ref || in_ref || ref_gt == ref_lt) return 0;
    const uint32_t d1c = d1 & ~((bit << 16) | (bit << 24));
    const bool same_adj = (d1c & 0xffffu) == (d2 & 0xffffu);
    const uint32_t gt1 = (d1c >> 16) & 0xffu, lt1 = d1c >> 24, gt2 = (d2 >> 16) & 0xffu, lt2 = d2 >> 24;
    const bool order_ok = ref_gt ? ((gt1 & ~gt2) == 0 && lt1 == lt2) : ((lt1 & ~lt2) == 0 && gt1 == gt2);
    if (!(same_adj && order_ok && pl.ball[b] == pl.ball[a] && (!directed || pl.level_in[b] == pl.level_in[a]))) return 0;
    return ref_gt ? 1 : -1;
}

// *twin_run (mode 2 only): how many of the last levels are twins in a row (2 .. 5) -- the leaves of a star: the images of r twin levels are
// the r-subsets of the candidate set of the first of them, C(|C1|, r), and none of the r levels is enumerated.
static int plan_tail_mode(const Plan &pl, bool directed, int *twin_run) {
    *twin_run = 2;
    if (pl.k - pl.n_fixed < 2) return 0;
    const int a = pl.k - 2, b = pl.k - 1;
    const uint32_t bit = 1u << a, d1 = pl.level[b];
    if (!directed && (d1 & 0xffu) == bit && ((d1 >> 8) & 0xffu) == 0 && (d1 >> 16) == 0) return 3;
    const bool ref_adj = (d1 & bit) != 0, ref_non = ((d1 >> 8) & bit) != 0, ref_gt = ((d1 >> 16) & bit) != 0, ref_lt = ((d1 >> 24) & bit) != 0;
    const bool ball_ref = (pl.ball[b] >> 3) != 0 && (pl.ball[b] & 7) == a;
    const bool in_ref = directed && ((pl.level_in[b] & bit) != 0 || ((pl.level_in[b] >> 8) & bit) != 0);
    if (ref_adj || ref_non || ball_ref || in_ref) return 0;
    if (!ref_gt && !ref_lt) return 1;
    if (const int dir = twin_link(pl, directed, a)) {
        // a run of twins: extend downwards while the level below is linked the same way
        int first = a;
        while (first - 1 >= pl.n_fixed && pl.k - (first - 1) <= 5 && twin_link(pl, directed, first - 1) == dir) --first;
        *twin_run = pl.k - first;
        return 2;
    }
    return 0;
}

// Matching order: fixed roots first, then greedily the unplaced vertex with most placed neighbours (ties: higher degree,
// lower id) so candidate sets shrink as early as possible.
static void matching_order(const Pattern &P, const int *fixed, int n_fixed, int *order) {
    bool placed[GSN_KMAX] = {false};
    for (int i = 0; i < n_fixed; ++i) { order[i] = fixed[i]; placed[fixed[i]] = true; }
    for (int l = n_fixed; l < P.k; ++l) {
        int best = -1, best_conn = -1, best_deg = -1;
        for (int v = 0; v < P.k; ++v) {
            if (placed[v]) continue;
            int conn = 0;
            for (int j = 0; j < l; ++j) conn += linked(P, v, order[j]);
            int deg = 0;
            for (int u = 0; u < P.k; ++u) deg += (u != v && linked(P, v, u)) ? 1 : 0;
            if (conn > best_conn || (conn == best_conn && deg > best_deg)) { best = v; best_conn = conn; best_deg = deg; }
        }
        order[l] = best; placed[best] = true;
    }
}

// Symmetry-breaking constraints (Grochow & Kellis 2007) for the subgroup A of Aut(H) fixing the roots pointwise:
// repeatedly take the earliest-ordered vertex v moved by A, require f(v) < f(u) for every other u in v's A-orbit, and
// descend to the stabiliser of v.  Exactly one map of every A-class satisfies all constraints.
static void symmetry_constraints(const Pattern &P, const int *fixed, int n_fixed, const int *order,
                                 std::vector<std::pair<int, int>> &less /* f(first) < f(second) */) {
    std::vector<std::array<uint8_t, GSN_KMAX>> A;
    for (auto &s : P.aut) {
        bool ok = true;
        for (int i = 0; i < n_fixed; ++i) ok = ok && s[fixed[i]] == fixed[i];
        if (ok) A.push_back(s);
    }
    while (A.size() > 1) {
        int v = -1;
        for (int l = 0; l < P.k && v < 0; ++l)
            for (auto &s : A)
                if (s[order[l]] != order[l]) { v = order[l]; break; }
        if (v < 0) break;
        bool in_orbit[GSN_KMAX] = {false};
        for (auto &s : A) in_orbit[s[v]] = true;
        for (int u = 0; u < P.k; ++u)
            if (u != v && in_orbit[u]) less.push_back({v, u});
        std::vector<std::array<uint8_t, GSN_KMAX>> S;
        for (auto &s : A)
            if (s[v] == v) S.push_back(s);
        A.swap(S);
    }
}

static Plan make_plan(const Pattern &P, int pattern_id, const int *fixed, int n_fixed, int out_col, bool induced) {
    Plan pl{};
    pl.k = P.k; pl.n_fixed = n_fixed; pl.out_col = out_col; pl.pattern = pattern_id;
    pl.root_a = fixed[0]; pl.root_b = n_fixed > 1 ? fixed[1] : 0;
    // every image has at least min-degree(H) neighbours among the images, so all images lie in that core of the target
    pl.min_degree = GSN_KMAX;
    for (int v = 0; v < P.k; ++v) {
        int dv = 0;
        for (int u = 0; u < P.k; ++u) dv += (u != v && has(P, v, u)) ? 1 : 0;
        pl.min_degree = std::min(pl.min_degree, dv);
    }
    if (P.directed) pl.min_degree = 0;   // (core and distance pruning are defined on undirected targets only)
    int order[GSN_KMAX], pos[GSN_KMAX];
    matching_order(P, fixed, n_fixed, order);
    for (int l = 0; l < P.k; ++l) pos[order[l]] = l;
    std::vector<std::pair<int, int>> less;
    symmetry_constraints(P, fixed, n_fixed, order, less);
    for (int l = 0; l < P.k; ++l) {
        uint32_t adj = 0, nonadj = 0, gt = 0, lt = 0;
        // bit j of adj: the image must be an (out-)neighbour of f_j -- pattern edge, or arc order[j] -> order[l]
        uint32_t adj_in = 0, nonadj_in = 0;
        for (int j = 0; j < l; ++j) {
            if (has(P, order[j], order[l])) adj |= 1u << j;
            else if (induced) nonadj |= 1u << j;
            if (P.directed) {        // arc order[l] -> order[j]: the image must be an in-neighbour of f_j
                if (has(P, order[l], order[j])) adj_in |= 1u << j;
                else if (induced) nonadj_in |= 1u << j;
            }
        }
        pl.level_in[l] = adj_in | (nonadj_in << 8);
        for (auto &c : less) {
            // f(c.first) < f(c.second)
            if (c.second == order[l] && pos[c.first] < l) gt |= 1u << pos[c.first];   // f_l > f_j
            if (c.first == order[l] && pos[c.second] < l) lt |= 1u << pos[c.second];  // f_l < f_j
        }
        pl.level[l] = adj | (nonadj << 8) | (gt << 16) | (lt << 24);
    }
    // Distance pruning: a subgraph isomorphism maps a path of the pattern onto a path of the target, so
    // dist_G(f(a), f(b)) <= dist_H(a, b).  Per level keep the tightest such bound of radius 2 or 3 against an earlier level
    // (radius 1 is the adjacency mask itself); the kernel intersects the candidates with the r-hop ball of that image.
    int dist[GSN_KMAX][GSN_KMAX];
    for (int a = 0; a < P.k; ++a)
        for (int b = 0; b < P.k; ++b) dist[a][b] = a == b ? 0 : (has(P, a, b) ? 1 : 99);
    for (int m = 0; m < P.k; ++m)
        for (int a = 0; a < P.k; ++a)
            for (int b = 0; b < P.k; ++b) dist[a][b] = std::min(dist[a][b], dist[a][m] + dist[m][b]);
    for (int l = 0; l < GSN_KMAX; ++l) pl.ball[l] = 0;
    for (int l = n_fixed; l < P.k && !P.directed; ++l) {
        int best_j = -1, best_r = 99;
        for (int j = 0; j < l; ++j) {
            const int d = dist[order[l]][order[j]];
            if (d >= 2 && d <= 3 && d < best_r) { best_r = d; best_j = j; }
        }
        if (best_j >= 0) pl.ball[l] = (uint8_t)(best_j | (best_r << 3));
    }
    return pl;
}

// orbit representatives of the directed edges under Aut(H)
static void arc_orbits(const Pattern &P, std::vector<int> &rep_arc /* indices into P.arcs */) {
    std::vector<int> orbit_of(P.arcs.size(), -1);
    for (size_t i = 0; i < P.arcs.size(); ++i) {
        if (orbit_of[i] >= 0) continue;
        rep_arc.push_back((int)i);
        for (auto &s : P.aut) {
            std::pair<int, int> img{s[P.arcs[i].first], s[P.arcs[i].second]};
            size_t j = std::lower_bound(P.arcs.begin(), P.arcs.end(), img) - P.arcs.begin();
            orbit_of[j] = (int)rep_arc.size() - 1;
        }
    }
}

}  // namespace gsn

using namespace gsn;

extern "C" int gsn_pattern_orbits(int64_t n_edges, const int64_t *edges, int directed_orbits, int64_t *out_k,
                                  int64_t *out_vertex_orbit, int64_t *out_n_vertex_orbits, int64_t *out_arcs,
                                  int64_t *out_arc_orbit, int64_t *out_n_arcs, int64_t *out_n_edge_orbits,
                                  int64_t *out_aut_count) {
    if (!edges || n_edges <= 0) return set_error(GSN_E_INVALID, "gsn_pattern_orbits: no edges");
    Pattern P;
    int rc = analyse(n_edges, edges, directed_orbits, P);
    if (rc) return rc;
    if (out_k) *out_k = P.k;
    if (out_vertex_orbit) for (int v = 0; v < P.k; ++v) out_vertex_orbit[v] = P.vorbit[v];
    if (out_n_vertex_orbits) *out_n_vertex_orbits = P.n_vorbits;
    for (size_t i = 0; i < P.arcs.size(); ++i) {
        if (out_arcs) { out_arcs[2 * i] = P.arcs[i].first; out_arcs[2 * i + 1] = P.arcs[i].second; }
        if (out_arc_orbit) out_arc_orbit[i] = P.arc_class[i];
    }
    if (out_n_arcs) *out_n_arcs = (int64_t)P.arcs.size();
    if (out_n_edge_orbits) *out_n_edge_orbits = P.n_classes;
    if (out_aut_count) *out_aut_count = (int64_t)P.aut.size();
    return GSN_OK;
}

// ---- vertex orbits of a graph with up to 64 vertices (line graphs of patterns: utils_graph_processing.py:205-231) ----------
namespace {
struct OrbitSearch {
    int n;
    uint64_t adj[64];
    int deg[64];
    int order[64];      // assignment order: the forced vertex first, then breadth-first from it, then the rest
    int img[64];
    // extend sigma over order[l..]; sigma must map edges to edges and non-edges to non-edges among assigned vertices
    bool extend(int l, uint64_t used) {
        if (l == n) return true;
        const int x = order[l];
        for (int y = 0; y < n; ++y) {
            if (((used >> y) & 1) || deg[y] != deg[x]) continue;
            bool ok = true;
            for (int j = 0; j < l && ok; ++j) {
                const int a = order[j];
                if (((adj[x] >> a) & 1) != ((adj[y] >> img[a]) & 1)) ok = false;
            }
            if (!ok) continue;
            img[x] = y;
            if (extend(l + 1, used | (1ull << y))) return true;
        }
        return false;
    }
    bool maps(int u, int v) {   // is there an automorphism with sigma(u) = v
        if (deg[u] != deg[v]) return false;
        bool seen[64] = {false};
        int cnt = 0;
        auto bfs = [&](int s) {
            int head = cnt;
            order[cnt++] = s; seen[s] = true;
            while (head < cnt) {
                const int a = order[head++];
                for (int b = 0; b < n; ++b)
                    if (((adj[a] >> b) & 1) && !seen[b]) { seen[b] = true; order[cnt++] = b; }
            }
        };
        bfs(u);
        for (int s = 0; s < n; ++s)
            if (!seen[s]) bfs(s);
        img[u] = v;
        return extend(1, 1ull << v);
    }
};
}  // namespace

extern "C" int gsn_graph_vertex_orbits(int64_t n_vertices, int64_t n_edges, const int64_t *edges, int64_t *out_orbit,
                                       int64_t *out_n_orbits) {
    if (n_vertices < 0 || n_vertices > 64) return set_error(GSN_E_UNSUPPORTED, "gsn_graph_vertex_orbits: %lld vertices; this build handles <= 64", (long long)n_vertices);
    if (n_edges > 0 && !edges) return set_error(GSN_E_INVALID, "gsn_graph_vertex_orbits: no edge array");
    OrbitSearch S;
    S.n = (int)n_vertices;
    for (int v = 0; v < S.n; ++v) S.adj[v] = 0;
    for (int64_t i = 0; i < n_edges; ++i) {
        const int64_t u = edges[2 * i], v = edges[2 * i + 1];
        if (u < 0 || v < 0 || u >= n_vertices || v >= n_vertices) return set_error(GSN_E_INVALID, "gsn_graph_vertex_orbits: vertex id out of range");
        if (u == v) continue;
        S.adj[u] |= 1ull << v;
        S.adj[v] |= 1ull << u;
    }
    for (int v = 0; v < S.n; ++v) S.deg[v] = __builtin_popcountll(S.adj[v]);
    // minrep[v] = smallest vertex of v's orbit: vertices in ascending order try the earlier orbit minima only
    int minrep[64];
    std::vector<int> reps;
    for (int v = 0; v < S.n; ++v) {
        minrep[v] = v;
        for (int r : reps)
            if (S.maps(r, v)) { minrep[v] = r; break; }
        if (minrep[v] == v) reps.push_back(v);
    }
    if (out_orbit)
        for (int v = 0; v < S.n; ++v) out_orbit[v] = (int64_t)(std::lower_bound(reps.begin(), reps.end(), minrep[v]) - reps.begin());
    if (out_n_orbits) *out_n_orbits = (int64_t)reps.size();
    return GSN_OK;
}

extern "C" int gsn_count_plan_build(int mode, int induced, int directed_orbits, int64_t n_patterns, const int64_t *pat_ptr,
                                    const int64_t *pat_edges, uint32_t *plan, int64_t capacity, int64_t *out_words,
                                    int64_t *out_n_cols) {
    if (mode != GSN_MODE_VERTEX && mode != GSN_MODE_EDGE) return set_error(GSN_E_INVALID, "mode must be 0 (vertex) or 1 (edge)");
    if (n_patterns <= 0 || !pat_ptr || !pat_edges) return set_error(GSN_E_INVALID, "no patterns");
    const bool directed = (directed_orbits & 2) != 0;
    if (directed && mode == GSN_MODE_EDGE)
        return set_error(GSN_E_UNSUPPORTED, "directed patterns: vertex counts only (the reference's directed edge counter fails on an unbound name, utils_graph_processing.py:146 vs :164)");
    const int stride = plan_stride((uint32_t)directed_orbits);
    std::vector<Plan> plans;
    int col0 = 0, kmax = 0;
    for (int64_t p = 0; p < n_patterns; ++p) {
        Pattern P;
        int rc = analyse(pat_ptr[p + 1] - pat_ptr[p], pat_edges + 2 * pat_ptr[p], directed_orbits, P);
        if (rc) return rc;
        kmax = std::max(kmax, P.k);
        if (mode == GSN_MODE_VERTEX) {
            for (int o = 0; o < P.n_vorbits; ++o) {
                int rep = 0;
                while (P.vorbit[rep] != o) ++rep;  // smallest vertex of the orbit
                plans.push_back(make_plan(P, (int)p, &rep, 1, col0 + o, induced != 0));
            }
            col0 += P.n_vorbits;
        } else {
            std::vector<int> reps;
            arc_orbits(P, reps);
            for (int r : reps) {
                int fixed[2] = {P.arcs[r].first, P.arcs[r].second};
                plans.push_back(make_plan(P, (int)p, fixed, 2, col0 + P.arc_class[r], induced != 0));
            }
            col0 += P.n_classes;
        }
    }
    // group the plans of one output column together: a kernel task is (column, row) and runs that column's plans
    std::stable_sort(plans.begin(), plans.end(), [](const Plan &a, const Plan &b) { return a.out_col < b.out_col; });
    const int64_t plans_off = PLAN_HEADER_WORDS + (col0 + 1) + col0;
    int64_t words = plans_off + (int64_t)plans.size() * stride;
    if (out_words) *out_words = words;
    if (out_n_cols) *out_n_cols = col0;
    if (!plan) return GSN_OK;
    if (capacity < words) return set_error(GSN_E_NOSPACE, "plan buffer too small: need %lld words", (long long)words);
    plan[0] = PLAN_MAGIC; plan[1] = (uint32_t)mode; plan[2] = (uint32_t)(induced != 0); plan[3] = (uint32_t)plans.size();
    plan[4] = (uint32_t)col0; plan[5] = (uint32_t)kmax; plan[6] = (uint32_t)((directed_orbits & 1) | (directed ? 2 : 0)); plan[7] = (uint32_t)plans_off;
    {   // col_ptr[c] .. col_ptr[c+1]: plan indices of column c
        uint32_t *cp = plan + PLAN_HEADER_WORDS;
        size_t i = 0;
        for (int c = 0; c <= col0; ++c) {
            while (i < plans.size() && plans[i].out_col < c) ++i;
            cp[c] = (uint32_t)i;
        }
    }
    {   // col_order: the columns by falling estimated search cost.  The kernel hands out the (column, row) cells of a graph in this
        // column order, so the lanes that go idle at the end of the pool are left with the short searches.  Estimate per plan:
        // product over the enumerated levels (the last one is counted by popcount) of a branching factor -- the whole graph for
        // a level without an adjacency constraint, an assumed degree shrinking with every further adjacency / order constraint.
        std::vector<double> cost((size_t)col0, 0.0);
        for (const Plan &pl : plans) {
            double c = 1.0;
            for (int l = pl.n_fixed; l + 1 < pl.k; ++l) {
                const uint32_t d = pl.level[l], din = pl.level_in[l];
                const int n_adj = __builtin_popcount(d & 0xffu) + __builtin_popcount(din & 0xffu);
                const int n_ord = __builtin_popcount(d >> 16);
                double b = n_adj == 0 ? 24.0 : 4.0 / n_adj;
                if (n_ord) b *= 0.5;
                c *= b < 1.0 ? 1.0 : b;
            }
            cost[(size_t)pl.out_col] += c;
        }
        std::vector<int> order((size_t)col0);
        for (int c = 0; c < col0; ++c) order[(size_t)c] = c;
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return cost[(size_t)x] > cost[(size_t)y]; });
        uint32_t *co = plan + PLAN_HEADER_WORDS + (col0 + 1);
        for (int c = 0; c < col0; ++c) co[c] = (uint32_t)order[(size_t)c];
    }
    for (size_t i = 0; i < plans.size(); ++i) {
        uint32_t *w = plan + plans_off + i * stride;
        const Plan &pl = plans[i];
        w[0] = (uint32_t)pl.k | ((uint32_t)pl.n_fixed << 8) | ((uint32_t)pl.out_col << 16);
        int twin_run = 2;
        const int tail_mode = plan_tail_mode(pl, directed != 0, &twin_run);
        w[1] = (uint32_t)pl.pattern | ((uint32_t)pl.root_a << 16) | ((uint32_t)pl.min_degree << 20) | ((uint32_t)pl.root_b << 24) |
               ((uint32_t)tail_mode << 28) | ((uint32_t)(twin_run - 2) << 30);
        for (int l = 0; l < GSN_KMAX; ++l) w[2 + l] = pl.level[l];
        for (int l = 0; l < PLAN_BALL_WORDS; ++l) w[2 + GSN_KMAX + l] = 0;
        for (int l = 0; l < GSN_KMAX; ++l) w[2 + GSN_KMAX + l / 4] |= (uint32_t)pl.ball[l] << (8 * (l % 4));
        if (directed)
            for (int l = 0; l < GSN_KMAX; ++l) w[PLAN_STRIDE_WORDS + l] = pl.level_in[l];
    }
    return GSN_OK;
}
