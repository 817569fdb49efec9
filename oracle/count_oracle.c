/*
 * ORACLE -- test infrastructure, NOT product code.
 *
 * Plain-C CPU restatement of the reference's substructure counting path (HP-1).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product path
 * (gsn_amd/) never does.  Parity pin: checked against tests/golden/{orbits,counts,counts2ids}.npz,
 * which were produced by the reference's own Python (tests/golden/make_golden.py) over networkx's VF2
 * because graph-tool cannot be installed here ("reference semantics over an independent VF2").
 *
 * Deliberately literal: every injective edge-preserving map f : V(H) -> V(G) is enumerated one by
 * one (|Aut(H)| maps per occurrence, no symmetry breaking, no counting shortcuts), each map bumps
 * the same cells the reference's Python loop bumps, and the totals are divided by |Aut(H)| at the
 * end -- so it is independent of every shortcut the HIP kernel takes.
 *
 * Reference lines followed (all in /root/reference):
 *   oracle_automorphism_orbits        utils_graph_processing.py:10-56
 *   oracle_induced_edge_orbits        utils_graph_processing.py:58-100
 *   oracle_vertex_counts              utils_graph_processing.py:103-131
 *   oracle_edge_counts                utils_graph_processing.py:134-179
 *   oracle_counts2ids                 utils_ids.py:7-29
 *
 * Third-party piece restated: graph_tool.topology.subgraph_isomorphism(sub, g, induced, subgraph=True)
 * (graph-tool, version unpinned in the reference README.md:39) = all subgraph monomorphisms
 * (induced=False, Boost vf2_subgraph_mono) or all induced-subgraph isomorphisms (induced=True,
 * Boost vf2_subgraph_iso) of the simple undirected graphs obtained after remove_self_loops /
 * remove_parallel_edges.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define OR_KMAX 16

typedef struct {
    int n;          /* vertices 0..n-1 (n = max id + 1 over all columns, as gt.Graph.add_edge_list does) */
    int words;      /* 64-bit words per adjacency row */
    uint64_t *adj;  /* n * words; bit v of row u = edge {u,v} (undirected: both rows) or arc u -> v (directed) */
    int directed;   /* gt.Graph(directed=...) -- main.py --directed (utils_graph_processing.py:14-16 / :108-110) */
} or_graph;

static int or_graph_init(or_graph *g, int n) {
    g->n = n;
    g->directed = 0;
    g->words = (n + 63) / 64;
    if (g->words == 0) g->words = 1;
    g->adj = (uint64_t *)calloc((size_t)(n > 0 ? n : 1) * g->words, sizeof(uint64_t));
    return g->adj ? 0 : -1;
}
static void or_graph_free(or_graph *g) { free(g->adj); g->adj = NULL; }
static inline void or_set(or_graph *g, int u, int v) { g->adj[(size_t)u * g->words + (v >> 6)] |= 1ull << (v & 63); }
static inline int or_has(const or_graph *g, int u, int v) { return (int)((g->adj[(size_t)u * g->words + (v >> 6)] >> (v & 63)) & 1); }

/* simple graph from an edge list: self loops dropped, parallel edges merged
 * (utils_graph_processing.py:16-19 / :110-113 / :150-153); directed: every row (u, v) is the arc u -> v and only
 * repeated arcs of the same direction are parallel */
static int or_build_d(or_graph *g, const int64_t *src, const int64_t *dst, int64_t m, int directed) {
    int64_t mx = -1;
    for (int64_t i = 0; i < m; ++i) { if (src[i] > mx) mx = src[i]; if (dst[i] > mx) mx = dst[i]; }
    if (or_graph_init(g, (int)(mx + 1))) return -1;
    g->directed = directed;
    for (int64_t i = 0; i < m; ++i) {
        if (src[i] == dst[i]) continue;
        or_set(g, (int)src[i], (int)dst[i]);
        if (!directed) or_set(g, (int)dst[i], (int)src[i]);
    }
    return 0;
}
static int or_build(or_graph *g, const int64_t *src, const int64_t *dst, int64_t m) { return or_build_d(g, src, dst, m, 0); }

/* ---- generic "all maps" enumerator: pattern vertices assigned in order 0..k-1 ------------------ */
typedef void (*or_visit)(const int *f, void *ctx);

typedef struct {
    const or_graph *H, *G;
    int k, induced;
    int order[OR_KMAX];   /* matching order: a connected order of the pattern vertices (VF2 also grows connected) */
    int f[OR_KMAX];       /* f[pattern vertex] = target vertex */
    uint8_t *used;
    or_visit visit;
    void *ctx;
    int64_t n_maps;
} or_enum;

static int or_feasible(const or_enum *e, int l, int v) {
    int p = e->order[l];
    for (int j = 0; j < l; ++j) {
        int q = e->order[j];
        int he = or_has(e->H, p, q), ge = or_has(e->G, v, e->f[q]);
        if (he && !ge) return 0;               /* pattern edge must be a target edge            */
        if (e->induced && !he && ge) return 0; /* induced: pattern non-edge must be a non-edge  */
        if (e->H->directed) {                  /* digraphs: the same two rules for the arc q -> p */
            he = or_has(e->H, q, p); ge = or_has(e->G, e->f[q], v);
            if (he && !ge) return 0;
            if (e->induced && !he && ge) return 0;
        }
    }
    return 1;
}

static void or_rec(or_enum *e, int l) {
    if (l == e->k) { e->n_maps++; e->visit(e->f, e->ctx); return; }
    int p = e->order[l];
    /* candidate generation: neighbours of the image of an already-matched pattern neighbour, else all */
    int anchor = -1;
    for (int j = 0; j < l; ++j) if (or_has(e->H, e->order[j], p)) { anchor = e->f[e->order[j]]; break; }   /* (arc q -> p: out-neighbours of f(q)) */
    if (anchor >= 0) {
        const uint64_t *row = e->G->adj + (size_t)anchor * e->G->words;
        for (int w = 0; w < e->G->words; ++w) {
            uint64_t bits = row[w];
            while (bits) {
                int v = w * 64 + __builtin_ctzll(bits);
                bits &= bits - 1;
                if (e->used[v] || !or_feasible(e, l, v)) continue;
                e->used[v] = 1; e->f[p] = v;
                or_rec(e, l + 1);
                e->used[v] = 0;
            }
        }
    } else {
        for (int v = 0; v < e->G->n; ++v) {
            if (e->used[v] || !or_feasible(e, l, v)) continue;
            e->used[v] = 1; e->f[p] = v;
            or_rec(e, l + 1);
            e->used[v] = 0;
        }
    }
}

static int64_t or_enumerate(const or_graph *H, const or_graph *G, int induced, or_visit visit, void *ctx) {
    or_enum e;
    e.H = H; e.G = G; e.k = H->n; e.induced = induced; e.visit = visit; e.ctx = ctx; e.n_maps = 0;
    if (H->n > OR_KMAX || H->n > G->n) return 0;
    /* connected matching order: lowest-numbered unplaced vertex adjacent to a placed one, else lowest unplaced */
    int placed[OR_KMAX] = {0};
    for (int l = 0; l < H->n; ++l) {
        int pick = -1;
        for (int p = 0; p < H->n && pick < 0; ++p) {
            if (placed[p]) continue;
            for (int j = 0; j < l; ++j) if (or_has(H, p, e.order[j]) || or_has(H, e.order[j], p)) { pick = p; break; }
        }
        if (pick < 0) for (int p = 0; p < H->n; ++p) if (!placed[p]) { pick = p; break; }
        e.order[l] = pick; placed[pick] = 1;
    }
    e.used = (uint8_t *)calloc((size_t)G->n + 1, 1);
    or_rec(&e, 0);
    free(e.used);
    return e.n_maps;
}

/* ---- automorphism_orbits (utils_graph_processing.py:10-56) ------------------------------------- */
typedef struct { int k; int memb[OR_KMAX]; } or_orbit_ctx;

static void or_orbit_visit(const int *f, void *ctx_) {
    /* :29-32  for original, vertex in enumerate(aut): orbit_membership[vertex] = min(original, orbit_membership[vertex]) */
    or_orbit_ctx *c = (or_orbit_ctx *)ctx_;
    for (int original = 0; original < c->k; ++original) {
        int vertex = f[original];
        if (original < c->memb[vertex]) c->memb[vertex] = original;
    }
}

/* out_membership[k]: contiguous orbit id per pattern vertex (np.unique(..., return_inverse), :40);
 * returns k (number of pattern vertices) or <0 */
static int or_automorphism_orbits(const int64_t *edges /* [m][2] */, int64_t m, int directed, int64_t *out_membership,
                                  int64_t *out_n_orbits, int64_t *out_aut_count) {
    or_graph H;
    int64_t *s = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m ? m : 1)), *d = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m ? m : 1));
    for (int64_t i = 0; i < m; ++i) { s[i] = edges[2 * i]; d[i] = edges[2 * i + 1]; }
    if (or_build_d(&H, s, d, m, directed)) { free(s); free(d); return -1; }
    free(s); free(d);
    if (H.n > OR_KMAX) { or_graph_free(&H); return -2; }
    or_orbit_ctx c; c.k = H.n;
    for (int v = 0; v < H.n; ++v) c.memb[v] = v;                     /* :24-26 */
    int64_t aut = or_enumerate(&H, &H, 0, or_orbit_visit, &c);       /* :22 induced=False self-match */
    /* :40 contiguous ids = rank of each distinct value */
    int n_orb = 0;
    for (int v = 0; v < H.n; ++v) {
        int rank = 0, seen_before = 0;
        for (int u = 0; u < H.n; ++u) {
            /* count distinct values smaller than memb[v] */
            if (c.memb[u] < c.memb[v]) {
                int dup = 0;
                for (int w = 0; w < u; ++w) if (c.memb[w] == c.memb[u]) dup = 1;
                if (!dup) rank++;
            }
        }
        (void)seen_before;
        out_membership[v] = rank;
        if (rank + 1 > n_orb) n_orb = rank + 1;
    }
    *out_n_orbits = n_orb;
    *out_aut_count = aut;
    int k = H.n;
    or_graph_free(&H);
    return k;
}
int oracle_automorphism_orbits(const int64_t *edges, int64_t m, int64_t *out_membership, int64_t *out_n_orbits, int64_t *out_aut_count) {
    return or_automorphism_orbits(edges, m, 0, out_membership, out_n_orbits, out_aut_count);
}
/* directed=True (:14): the pattern is the digraph whose arcs are the rows of edge_list */
int oracle_automorphism_orbits_directed(const int64_t *edges, int64_t m, int64_t *out_membership, int64_t *out_n_orbits, int64_t *out_aut_count) {
    return or_automorphism_orbits(edges, m, 1, out_membership, out_n_orbits, out_aut_count);
}

/* sorted bidirectional edge list of the simple pattern graph: to_undirected(get_edges()) = concat both
 * directions + coalesce (sorted by (u,v)) -- utils_graph_processing.py:73-74 and :146-147.
 * out_list [2m'][2]; returns number of directed edges */
static int or_sorted_arcs(const or_graph *H, int64_t *out_list) {
    int c = 0;
    for (int u = 0; u < H->n; ++u)
        for (int v = 0; v < H->n; ++v)
            if (or_has(H, u, v)) { out_list[2 * c] = u; out_list[2 * c + 1] = v; c++; }
    return c;
}

/* induced_edge_automorphism_orbits (utils_graph_processing.py:58-100).
 * out_arcs [<=k*k][2], out_arc_membership[<=k*k]; returns number of directed edges */
int oracle_induced_edge_orbits(const int64_t *edges, int64_t m, int directed_orbits, int64_t *out_arcs,
                               int64_t *out_arc_membership, int64_t *out_n_orbits, int64_t *out_aut_count) {
    int64_t vmemb[OR_KMAX], n_vorb;
    int k = oracle_automorphism_orbits(edges, m, vmemb, &n_vorb, out_aut_count);   /* :65-67 */
    if (k < 0) return k;
    or_graph H;
    int64_t *s = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m ? m : 1)), *d = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m ? m : 1));
    for (int64_t i = 0; i < m; ++i) { s[i] = edges[2 * i]; d[i] = edges[2 * i + 1]; }
    or_build(&H, s, d, m); free(s); free(d);
    int na = or_sorted_arcs(&H, out_arcs);
    /* :77-94 first-seen numbering of the key {orbit(u),orbit(v)} (ordered pair iff directed_orbits) */
    int64_t keys_a[OR_KMAX * OR_KMAX], keys_b[OR_KMAX * OR_KMAX];
    int n_keys = 0;
    for (int i = 0; i < na; ++i) {
        int64_t a = vmemb[out_arcs[2 * i]], b = vmemb[out_arcs[2 * i + 1]];
        if (!directed_orbits && a > b) { int64_t t = a; a = b; b = t; }
        int idx = -1;
        for (int q = 0; q < n_keys; ++q) if (keys_a[q] == a && keys_b[q] == b) { idx = q; break; }
        if (idx < 0) { idx = n_keys; keys_a[n_keys] = a; keys_b[n_keys] = b; n_keys++; }
        out_arc_membership[i] = idx;
    }
    *out_n_orbits = n_keys;
    or_graph_free(&H);
    return na;
}

/* ---- vertex counts (utils_graph_processing.py:103-131) ------------------------------------------ */
typedef struct { int k; int n_orb; const int64_t *memb; int64_t *counts; } or_vc_ctx;

static void or_vc_visit(const int *f, void *ctx_) {
    or_vc_ctx *c = (or_vc_ctx *)ctx_;
    for (int i = 0; i < c->k; ++i) c->counts[(size_t)f[i] * c->n_orb + c->memb[i]] += 1;   /* :124-126 */
}

/* edge_index given as two int64 rows (src[E], dst[E]); out [num_nodes][n_orbits] int64.
 * returns 0, or <0 on error (-3: a count is not divisible by aut_count -> restatement bug) */
static int or_vertex_counts(const int64_t *src, const int64_t *dst, int64_t E, int64_t num_nodes, const int64_t *pat_edges,
                            int64_t pat_m, int induced, int directed, int64_t *out, int64_t out_stride) {
    int64_t memb[OR_KMAX], n_orb, aut;
    int k = or_automorphism_orbits(pat_edges, pat_m, directed, memb, &n_orb, &aut);
    if (k < 0) return k;
    or_graph H, G;
    int64_t *s = (int64_t *)malloc(sizeof(int64_t) * (size_t)(pat_m ? pat_m : 1)), *d = (int64_t *)malloc(sizeof(int64_t) * (size_t)(pat_m ? pat_m : 1));
    for (int64_t i = 0; i < pat_m; ++i) { s[i] = pat_edges[2 * i]; d[i] = pat_edges[2 * i + 1]; }
    or_build_d(&H, s, d, pat_m, directed); free(s); free(d);
    or_build_d(&G, src, dst, E, directed);                                           /* :110-113 */
    int64_t rows = num_nodes > G.n ? num_nodes : G.n;
    int64_t *cnt = (int64_t *)calloc((size_t)(rows ? rows : 1) * n_orb, sizeof(int64_t)); /* :122 */
    or_vc_ctx c; c.k = k; c.n_orb = (int)n_orb; c.memb = memb; c.counts = cnt;
    or_enumerate(&H, &G, induced, or_vc_visit, &c);                                  /* :116, :123 */
    int rc = 0;
    for (int64_t r = 0; r < num_nodes; ++r)
        for (int64_t o = 0; o < n_orb; ++o) {
            int64_t x = cnt[r * n_orb + o];
            if (x % aut) rc = -3;
            out[r * out_stride + o] = x / aut;                                       /* :127 */
        }
    free(cnt); or_graph_free(&H); or_graph_free(&G);
    return rc;
}
int oracle_vertex_counts(const int64_t *src, const int64_t *dst, int64_t E, int64_t num_nodes,
                         const int64_t *pat_edges, int64_t pat_m, int induced, int64_t *out, int64_t out_stride) {
    return or_vertex_counts(src, dst, E, num_nodes, pat_edges, pat_m, induced, 0, out, out_stride);
}
/* directed=True (:108): pattern and target are digraphs; a match preserves arcs (induced: and non-arcs, per direction) */
int oracle_vertex_counts_directed(const int64_t *src, const int64_t *dst, int64_t E, int64_t num_nodes,
                                  const int64_t *pat_edges, int64_t pat_m, int induced, int64_t *out, int64_t out_stride) {
    return or_vertex_counts(src, dst, E, num_nodes, pat_edges, pat_m, induced, 1, out, out_stride);
}

/* ---- edge counts (utils_graph_processing.py:134-179) --------------------------------------------- */
typedef struct {
    int n_arcs, n_orb, n;
    const int64_t *arcs, *amemb;
    const int64_t *edge_dict; /* n*n -> column or -1 */
    int64_t *counts;
    int key_error;
} or_ec_ctx;

static void or_ec_visit(const int *f, void *ctx_) {
    or_ec_ctx *c = (or_ec_ctx *)ctx_;
    for (int i = 0; i < c->n_arcs; ++i) {                                            /* :164 */
        int u = f[c->arcs[2 * i]], v = f[c->arcs[2 * i + 1]];                        /* :172 */
        int64_t col = c->edge_dict[(size_t)u * c->n + v];
        if (col < 0) { c->key_error = 1; continue; }                                 /* KeyError in the reference (:173) */
        c->counts[(size_t)col * c->n_orb + c->amemb[i]] += 1;                        /* :173 */
    }
}

/* out [E][n_edge_orbits]; returns 0, -3 (divisibility), -4 (KeyError: a match uses a direction that is not a column) */
int oracle_edge_counts(const int64_t *src, const int64_t *dst, int64_t E, const int64_t *pat_edges, int64_t pat_m,
                       int induced, int directed_orbits, int64_t *out, int64_t out_stride) {
    int64_t arcs[2 * OR_KMAX * OR_KMAX], amemb[OR_KMAX * OR_KMAX], n_orb, aut;
    int na = oracle_induced_edge_orbits(pat_edges, pat_m, directed_orbits, arcs, amemb, &n_orb, &aut);
    if (na < 0) return na;
    or_graph H, G;
    int64_t *s = (int64_t *)malloc(sizeof(int64_t) * (size_t)(pat_m ? pat_m : 1)), *d = (int64_t *)malloc(sizeof(int64_t) * (size_t)(pat_m ? pat_m : 1));
    for (int64_t i = 0; i < pat_m; ++i) { s[i] = pat_edges[2 * i]; d[i] = pat_edges[2 * i + 1]; }
    or_build(&H, s, d, pat_m); free(s); free(d);
    or_build(&G, src, dst, E);                                                       /* :150-153 */
    size_t nn = (size_t)(G.n ? G.n : 1);
    int64_t *dict = (int64_t *)malloc(sizeof(int64_t) * nn * nn);
    for (size_t i = 0; i < nn * nn; ++i) dict[i] = -1;
    for (int64_t i = 0; i < E; ++i) dict[(size_t)src[i] * nn + dst[i]] = i;          /* :142-144 last duplicate wins */
    int64_t *cnt = (int64_t *)calloc((size_t)(E ? E : 1) * n_orb, sizeof(int64_t));  /* :159 */
    or_ec_ctx c; c.n_arcs = na; c.n_orb = (int)n_orb; c.n = (int)nn; c.arcs = arcs; c.amemb = amemb;
    c.edge_dict = dict; c.counts = cnt; c.key_error = 0;
    or_enumerate(&H, &G, induced, or_ec_visit, &c);                                  /* :156, :161 */
    int rc = c.key_error ? -4 : 0;
    for (int64_t r = 0; r < E; ++r)
        for (int64_t o = 0; o < n_orb; ++o) {
            int64_t x = cnt[r * n_orb + o];
            if (x % aut && rc == 0) rc = -3;
            out[r * out_stride + o] = x / aut;                                       /* :175 */
        }
    free(cnt); free(dict); or_graph_free(&H); or_graph_free(&G);
    return rc;
}

/* ---- subgraph_counts2ids over a batch of graphs (utils_ids.py:7-29) -------------------------------
 * Self loops are assumed already removed by the caller when mode == edge (utils_ids.py:11-15 does it before
 * calling count_fn, and the identifiers' rows refer to the stripped columns); vertex mode tolerates them.
 * Graph g owns columns [edge_ptr[g], edge_ptr[g+1]) of the graph-LOCAL edge_index (src,dst) and
 * rows node_ptr[g]..node_ptr[g+1] (vertex mode) or its columns (edge mode) of `out` [rows_total][cols_total].
 * Patterns: pat_ptr[P+1] into pat_edges [.,2].  Column order = pattern order then orbit id (:19-25).
 * n_threads > 1 uses OpenMP over graphs (the reference's joblib-over-graphs, utils_data_gen.py:60-70). */
int oracle_counts2ids(int mode /*0 vertex, 1 edge*/, int induced, int directed_orbits, int64_t n_graphs,
                      const int64_t *node_ptr, const int64_t *edge_ptr, const int64_t *src, const int64_t *dst,
                      int64_t n_patterns, const int64_t *pat_ptr, const int64_t *pat_edges,
                      int64_t *out, int64_t cols_total, int n_threads) {
    /* bit 1 of directed_orbits = main.py --directed (digraph patterns and targets; vertex mode only: the reference's
     * directed edge counter dies on an unbound name, utils_graph_processing.py:146 vs :164) */
    const int directed = (directed_orbits >> 1) & 1;
    directed_orbits &= 1;
    if (directed && mode != 0) return -6;
    int64_t *col_off = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_patterns + 1));
    col_off[0] = 0;
    for (int64_t p = 0; p < n_patterns; ++p) {
        int64_t n_orb, aut;
        const int64_t *pe = pat_edges + 2 * pat_ptr[p];
        int64_t pm = pat_ptr[p + 1] - pat_ptr[p];
        if (mode == 0) { int64_t memb[OR_KMAX]; if (or_automorphism_orbits(pe, pm, directed, memb, &n_orb, &aut) < 0) { free(col_off); return -2; } }
        else { int64_t arcs[2 * OR_KMAX * OR_KMAX], am[OR_KMAX * OR_KMAX]; if (oracle_induced_edge_orbits(pe, pm, directed_orbits, arcs, am, &n_orb, &aut) < 0) { free(col_off); return -2; } }
        col_off[p + 1] = col_off[p] + n_orb;
    }
    if (col_off[n_patterns] != cols_total) { free(col_off); return -5; }
    int rc_all = 0;
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (int64_t g = 0; g < n_graphs; ++g) {
        int64_t e0 = edge_ptr[g], E = edge_ptr[g + 1] - e0;
        int64_t nn = node_ptr[g + 1] - node_ptr[g];
        for (int64_t p = 0; p < n_patterns; ++p) {
            const int64_t *pe = pat_edges + 2 * pat_ptr[p];
            int64_t pm = pat_ptr[p + 1] - pat_ptr[p];
            int rc;
            if (mode == 0)
                rc = or_vertex_counts(src + e0, dst + e0, E, nn, pe, pm, induced, directed,
                                      out + node_ptr[g] * cols_total + col_off[p], cols_total);
            else
                rc = oracle_edge_counts(src + e0, dst + e0, E, pe, pm, induced, directed_orbits,
                                        out + e0 * cols_total + col_off[p], cols_total);
            if (rc) {
#ifdef _OPENMP
#pragma omp critical
#endif
                rc_all = rc;
            }
        }
    }
    free(col_off);
    return rc_all;
}
