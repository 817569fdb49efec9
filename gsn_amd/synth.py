"""Seeded synthetic graph generators + tiny readers for the data files the path is measured on.

Shapes follow SURVEY.md §8(d): ZINC-shape molecules (~23 nodes / ~50 directed edges, max degree 4,
5-/6-rings), molhiv-shape, Erdos-Renyi G(n, m), and the two real inputs that ship with the
reference as *data* (graph6 files, the IMDB-BINARY txt).  All generators are ours; nothing here is
taken from the reference.  Graphs are returned as ``(num_nodes, edge_index int64 [2, E])`` with both
directions present and, unless stated, columns sorted by (row, col) the way ``nonzero()`` on an
adjacency matrix yields them (that is how the reference's ZINC loader produces ``edge_mat``,
utils_data_prep.py:150-156).
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "zinc_shape_graph", "zinc_shape_batch", "er_graph", "parse_graph6", "read_graph6_file",
    "read_tu_txt", "undirected_to_edge_index", "Batch", "collate",
]


def undirected_to_edge_index(n: int, und_edges, sort: bool = True) -> np.ndarray:
    """Both directions of every undirected edge; sorted by (row, col) if ``sort``."""
    und = np.asarray(list(und_edges), dtype=np.int64).reshape(-1, 2)
    if und.shape[0] == 0:
        return np.zeros((2, 0), dtype=np.int64)
    ei = np.concatenate([und, und[:, ::-1]], axis=0)
    if sort:
        key = ei[:, 0] * max(n, 1) + ei[:, 1]
        ei = ei[np.argsort(key, kind="stable")]
    return np.ascontiguousarray(ei.T)


def zinc_shape_graph(rng: np.random.Generator, mean_n: float = 23.2, sd_n: float = 4.5,
                     n_min: int = 9, n_max: int = 38, ring_rate: float = 1.6, max_deg: int = 4):
    """Random tree with max degree ``max_deg`` plus Poisson(ring_rate) ring closures forming 5-/6-rings."""
    n = int(np.clip(round(rng.normal(mean_n, sd_n)), n_min, n_max))
    deg = np.zeros(n, dtype=np.int64)
    parent = np.full(n, -1, dtype=np.int64)
    adj = [set() for _ in range(n)]
    for v in range(1, n):
        # attach to a random earlier vertex that still has capacity; bias to recent ones -> chains
        lo = max(0, v - 6)
        cands = [u for u in range(lo, v) if deg[u] < max_deg - 1] or [u for u in range(v) if deg[u] < max_deg]
        u = int(cands[rng.integers(len(cands))])
        parent[v] = u
        adj[u].add(v); adj[v].add(u)
        deg[u] += 1; deg[v] += 1
    n_rings = int(rng.poisson(ring_rate))
    for _ in range(n_rings):
        # close a ring of length 5 or 6: pick v, walk L-1 tree/graph steps without revisiting, connect the ends
        L = 6 if rng.random() < 0.7 else 5
        for _attempt in range(8):
            v = int(rng.integers(n))
            path = [v]
            ok = True
            for _s in range(L - 1):
                nxt = [w for w in adj[path[-1]] if w not in path]
                if not nxt:
                    ok = False
                    break
                path.append(int(nxt[rng.integers(len(nxt))]))
            if not ok:
                continue
            a, b = path[0], path[-1]
            if b in adj[a] or deg[a] >= max_deg or deg[b] >= max_deg:
                continue
            adj[a].add(b); adj[b].add(a)
            deg[a] += 1; deg[b] += 1
            break
    und = [(u, v) for u in range(n) for v in adj[u] if u < v]
    return n, undirected_to_edge_index(n, und)


class Batch:
    """Disjoint union of graphs, the way PyG's collate builds it (SURVEY.md §8a footnote):
    node-offset ``edge_index``, ``node_ptr`` / ``edge_ptr`` (CSR over graphs) and ``batch`` vector."""

    def __init__(self, node_ptr, edge_ptr, edge_index, **extra):
        self.node_ptr = np.asarray(node_ptr, dtype=np.int64)
        self.edge_ptr = np.asarray(edge_ptr, dtype=np.int64)
        self.edge_index = np.ascontiguousarray(edge_index, dtype=np.int64)
        self.num_graphs = len(self.node_ptr) - 1
        self.num_nodes = int(self.node_ptr[-1])
        self.num_edges = int(self.edge_ptr[-1])
        for k, v in extra.items():
            setattr(self, k, v)

    @property
    def batch(self) -> np.ndarray:
        return np.repeat(np.arange(self.num_graphs, dtype=np.int64), np.diff(self.node_ptr))

    def graph(self, g: int):
        """(num_nodes, graph-local edge_index) of graph ``g``."""
        lo, hi = self.edge_ptr[g], self.edge_ptr[g + 1]
        return int(self.node_ptr[g + 1] - self.node_ptr[g]), self.edge_index[:, lo:hi] - self.node_ptr[g]


def collate(graphs) -> Batch:
    """graphs: iterable of (num_nodes, edge_index[2,E] graph-local)."""
    node_ptr = [0]
    edge_ptr = [0]
    eis = []
    for n, ei in graphs:
        eis.append(np.asarray(ei, dtype=np.int64) + node_ptr[-1])
        node_ptr.append(node_ptr[-1] + int(n))
        edge_ptr.append(edge_ptr[-1] + ei.shape[1])
    edge_index = np.concatenate(eis, axis=1) if eis else np.zeros((2, 0), dtype=np.int64)
    return Batch(node_ptr, edge_ptr, edge_index)


def zinc_shape_batch(num_graphs: int, seed: int = 0, n_atom_types: int = 28, **kw) -> Batch:
    """ZINC-shaped batch with integer atom types in [0,28) and bond types in {1,2,3} (p=.75,.2,.05),
    symmetric per undirected edge."""
    rng = np.random.default_rng(seed)
    graphs = [zinc_shape_graph(rng, **kw) for _ in range(num_graphs)]
    b = collate(graphs)
    b.atom_type = rng.integers(0, n_atom_types, size=b.num_nodes, dtype=np.int64)
    # bond type must agree on (u,v) and (v,u): hash the unordered pair
    u, v = b.edge_index
    lo, hi = np.minimum(u, v), np.maximum(u, v)
    h = (lo * 2654435761 + hi * 40503 + seed * 97) % 1000
    b.bond_type = np.where(h < 750, 1, np.where(h < 950, 2, 3)).astype(np.int64)
    return b


def er_graph(n: int, m: int, seed: int):
    """Uniform random m-edge simple graph on n vertices, G(n, m)."""
    rng = np.random.default_rng(seed)
    total = n * (n - 1) // 2
    m = min(m, total)
    pick = rng.choice(total, size=m, replace=False)
    # unrank pair index -> (i, j), i < j
    iu, ju = np.triu_indices(n, k=1)
    und = np.stack([iu[pick], ju[pick]], axis=1)
    return n, undirected_to_edge_index(n, und)


# ---------------------------------------------------------------------------------------------
# readers for the data files that ship with the reference (format readers, written from the
# public format descriptions: graph6 = http://users.cecs.anu.edu.au/~bdm/data/formats.txt)
# ---------------------------------------------------------------------------------------------

def parse_graph6(line: bytes):
    """One graph6 record -> (n, list of undirected edges (i<j)) in the format's own order
    (column-major over the upper triangle: (0,1),(0,2),(1,2),(0,3),...)."""
    line = line.strip()
    if line.startswith(b">>graph6<<"):
        line = line[10:]
    data = [c - 63 for c in line]
    if data[0] <= 62:
        n, data = data[0], data[1:]
    elif data[1] <= 62:
        n = (data[1] << 12) + (data[2] << 6) + data[3]
        data = data[4:]
    else:
        raise ValueError("graph6: n too large for this reader")
    bits = []
    for d in data:
        bits.extend((d >> s) & 1 for s in (5, 4, 3, 2, 1, 0))
    edges = []
    k = 0
    for j in range(1, n):
        for i in range(j):
            if bits[k]:
                edges.append((i, j))
            k += 1
    return n, edges


def read_graph6_file(path: str):
    out = []
    with open(path, "rb") as f:
        for line in f:
            if line.strip():
                out.append(parse_graph6(line))
    return out


def read_tu_txt(path: str):
    """Reader for the 'powerful-gnns' TU txt format (n_graphs; per graph 'n label'; per node
    'tag deg nb...').  Returns a list of (n, label, node_tags, und_edges) with undirected edges in
    first-seen order (i, j) as the adjacency rows list them."""
    out = []
    with open(path, "r") as f:
        n_g = int(f.readline().strip())
        for _ in range(n_g):
            n, label = (int(w) for w in f.readline().split())
            seen = {}
            tags = []
            for j in range(n):
                row = f.readline().split()
                tags.append(int(row[0]))
                d = int(row[1])
                for w in row[2:2 + d]:
                    k = int(w)
                    key = (min(j, k), max(j, k))
                    if key not in seen:
                        seen[key] = (j, k)
            out.append((n, label, tags, list(seen.values())))
    return out
