// The one-launch `general` layer for WIDE node rows (d_x = 128, edge rows cat(x_i, x_j, ids.., e) of K = 256 + (<= 16) columns: layers
// 1 .. L-1 of every d = 128 reference model, models_graph_classification.py:147-155 over MPNN_edge_sparse.py:110-151 /
// GSN_edge_sparse.py:82-170) on GRAPH-ALIGNED tiles, with the node part of the edge stage computed once per NODE:
//
//     r_e   = act_e( bn_e( [x_i | x_j | z_e] W1^T + b1 ) )  =  act_e( P_i[t(e)] + P_j[s(e)] + z_e Wz^T )       per edge
//     P_i   = x Wi'^T + c0,   P_j = x Wj'^T                                                                    per NODE
//     S_v   = sum_{e -> v} r_e
//     h_v   = act_0( bn_0( [x_v | S_v | deg_v] W0'^T + b0 ) ),    out_v = act_1( bn_1( h_v W1'^T + b1' ) )
//
// (Wi', Wj', Wz the column blocks of bn_scale . W1: the algebra of gsn_edge_split_sum_hip, inside one launch.)  layer_w.hip multiplies
// every EDGE row by all 272 columns: 17 x 4 x 3 plane products per 32 edge rows, ~2.05 edge rows per node -- 204 of its 357 GF per 65 536
// molecules are products of x_i / x_j that are the same for every edge of a node.  Here they are 2 x 8 x 4 x 3 products per 32 NODES.
// A batch is a disjoint union, so the sources of a node's in-edges lie in its own graph:
//   * a WORKGROUP (four waves, one per SIMD, 512 registers each) takes runs of WHOLE graphs of <= 128 nodes together (ZINC: ~5.5 graphs,
//     ~90 % of the rows used), wave w the rows 32 w .. 32 w + 31 of the run.  Every wave multiplies its 32 x rows (two fp16 planes, in
//     registers for the whole tile: they are the B operand of three products) by [Wj | Wi | W0x]: P_j^T goes to LDS as fp32 rows (66 KiB:
//     the one thing the four waves share), P_i^T and the x part of node stage 0 stay in accumulator registers.
//   * edge phase, no matrix work on E rows except the <= 16 per-edge columns: the accumulator layout gives lane (t, h) 64 features of
//     NODE t -- so a lane owns a target, walks its in-edges (molecules: <= 4 rounds per tile), and per round the z_e Wz^T term of all 32
//     lanes' current edges is ONE 12-product MFMA step whose output column t is exactly what lane t needs; P_j[s] is 16 ds_read_b128 of the
//     source's row; relu and the per-target sum are plain register adds in the S^T layout node stage 0 reads.  No incidence product, no
//     bf16x3 split of activated rows (960 instructions per 64 x 128 values in layer_w.hip), no atomics.
//   * all weight fragments (320 KiB per tile) stream from L2 once per WORKGROUP: a cyclic schedule of 15 steps through a three-slot ring
//     of 24 KiB, filled by global_load_lds (no staging registers, no ds_write), two steps ahead; one barrier per step.
// Same prepared buffer as layer_w.hip (w_prepare: its fragments are in this kernel's operand layouts already), same fp16x3 arithmetic
// (two fp16 planes per operand after exact power-of-two row / matrix scaling, three plane products, fp32 accumulation), same
// non-finite semantics (an Inf / NaN in a node row, an edge row or a hidden row makes exactly the output rows that see it NaN).
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "chain_common.h"
#include "layer_g.h"
#include "layer_rr_inl.h"
#include "layer_w.h"

// diagnostic builds (RR_VARIANT_SRC=layer_g scripts/rr_variant.sh NAME -DG_ABL_...): one kind of work switched off, results are garbage
#ifdef G_ABL_NOBAR
#define G_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define G_BARRIER() lds_barrier()
#endif

// issue order inside one fenced group of products: NM x (one MFMA, then -- while there are any -- one LDS read)
#ifdef G_NOMIX
#define G_MIX(NM, ND)
#define G_MIXV(NM, ND, NV)
#else
#define G_MIXV(NM, ND, NV) _Pragma("unroll") for (int mix_q = 0; mix_q < (NM); ++mix_q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); \
    if (mix_q < (ND)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, (NV), 0); }
#define G_MIX(NM, ND) _Pragma("unroll") for (int mix_q = 0; mix_q < (NM); ++mix_q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); \
    if (mix_q < (ND)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#endif

namespace gsn {

constexpr int G_ROWS = G_MAX_GRAPH_NODES;          // nodes per workgroup tile
constexpr int G_PITCH = 528;                       // bytes per P_j row in LDS: [half][feature block][4-group][4] floats + 16 (bank spread)
constexpr int G_SLOT = 24 * 1024;                  // ring slot: 24 fragments
constexpr int G_NSTEP = 15;                        // steps of the cyclic weight schedule: 8 (products of x) + 4 (S part of node stage 0) + 3 (node stage 1)
constexpr int G_PRE = 4;                           // in-edges per target whose indices and per-edge columns are requested a tile ahead
constexpr int GL_PJ = 0;                           // P_j rows of the tile's 128 nodes; behind the edge phase: the waves' output rows on their way out
constexpr int GL_BAD = GL_PJ + G_ROWS * G_PITCH;   // 128 ints: the node's row holds an Inf / NaN
constexpr int GL_C0 = GL_BAD + 4 * G_ROWS;         // the edge stage's folded bias in the accumulator order [half][fb][group][4]
constexpr int GL_C1 = GL_C0 + 512;                 // node stage 1's folded bias
constexpr int GL_WZ = GL_C1 + 512;                 // 8 fragments of the per-edge chunk + 4 of node stage 0's (c0, deg) product
constexpr int GL_RING = GL_WZ + 12 * 1024;
constexpr int GL_TOTAL = GL_RING + 3 * G_SLOT;
static_assert(GL_TOTAL <= 160 * 1024, "LDS budget");

struct GArgs {
    int n_nodes, n_edges, n_graphs, n_wg;
    const int32_t *seg_ptr;
    const int32_t *src;               // source node of every edge row (target-sorted order)
    const int32_t *eidx;              // row of the per-edge blocks of every edge row (perm), or = src when unused
    const long long *node_ptr;
    WQuad zq[4];                      // the four 16-byte quads of the per-edge chunk: role 0 = row of the target, 1 = of the source, 2 = eidx
    const float *x;
    float *out;
    const unsigned *prep;
};

// loads of step k's fragments per wave; X(c): other loads a wave issues in step c IN FRONT of its request of step c + 2 (step 0: the
// targets' segment bounds and the next window, 2: their first in-edges' indices, 5: those edges' per-edge columns, 8 .. 14: one chunk of
// the next tile's x rows each -- a CU moves ~10 bytes per cycle to and from memory when all of them stream: 64 KiB of x rows requested at
// once cost the tile 5 000 cycles)
__device__ __forceinline__ constexpr int g_step_loads(int k) { return (k < 8 || k == 12 || k == 13) ? 6 : 4; }
__device__ __forceinline__ constexpr int g_extra(int c) { return c == 0 ? 3 : ((c == 2 || c == 5) ? 2 * G_PRE : (c >= 8 ? 2 : 0)); }
// a wave's loads complete in order: at the end of step c its share of step c + 1 (requested at the top of step c - 1) has landed when at
// most the loads issued behind that request are outstanding (stores are not counted: they may complete in any order relative to loads,
// and a bound below the number of younger LOADS holds whatever they do)
__device__ __forceinline__ constexpr int g_wait(int c) { return g_extra(c) + g_step_loads((c + 2) % G_NSTEP); }
__device__ __forceinline__ void g_vmcnt(int n) {       // (n is a constant after unrolling: one of the schedule's counts)
    switch (n) {
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

struct GMeta {                        // a lane's target in a tile: node, segment bounds, its first in-edges
    int node, pt, pt1;
    int srcv[G_PRE], eiv[G_PRE];
};

template <bool PROF, bool HASZ>
__global__ __launch_bounds__(256) void layer_fused_kernel_g(GArgs a, unsigned long long *prof) {
    using SH = WShape;
    constexpr int WB = SH::WB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    auto clk = [&]() -> unsigned {
        if (!PROF) return 0u;
        __builtin_amdgcn_sched_barrier(0);
        const unsigned v = (unsigned)__builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
        return v;
    };
    unsigned pe[4] = {0, 0, 0, 0};                      // edge phase: in front of the rounds, rounds (cycles), rounds (count)
    unsigned pd[6] = {0, 0, 0, 0, 0, 0};                // inside the product steps: other loads + output rows, stream request, first reads + products, wait, barrier
    unsigned pc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // top, -, product steps, P_j write, edge phase, stage 0, between, stage 1, stores, tiles

    const int tid = threadIdx.x;
    const int lane0 = tid & 63;
    int wave_v = tid >> 6;
    int wave = __builtin_amdgcn_readfirstlane(wave_v);
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef const __attribute__((address_space(1))) rr_f4 *gf4_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    typedef __attribute__((address_space(3))) rr_f4 *ldsf4_t;
    typedef __attribute__((address_space(3))) float *ldsf_t;
    typedef __attribute__((address_space(3))) int *ldsi_t;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;

    // ---- prologue: the per-edge chunk's fragments, node stage 0's (c0, deg) fragments and two folded biases -> LDS ----------------------
    const int Ee = (int)a.prep[WH_EE], E0 = (int)a.prep[WH_E0], E1 = (int)a.prep[WH_E1], e_min = (int)a.prep[WH_EMIN];
    const bool w_bad = a.prep[WH_BAD] != 0;
    const unsigned acts = a.prep[WH_ACT];
    {
        const float *tab0 = reinterpret_cast<const float *>(a.prep + W_HDR + SH::F_ALL * 256);
        const rr_u4 *fr = reinterpret_cast<const rr_u4 *>(a.prep + W_HDR);
        rr_u4 *dst = reinterpret_cast<rr_u4 *>(smem + GL_WZ);
        for (int i = tid; i < 12 * 64; i += 256) {
            const int f = i >> 6;
            dst[i] = fr[(f < 8 ? SH::F_WE + f : SH::F_WB + (f - 8)) * 64 + (i & 63)];
        }
        if (tid < 128) {
            const int h = tid >> 6, fb = (tid >> 4) & 3, g = (tid >> 2) & 3, j = tid & 3;
            reinterpret_cast<float *>(smem + GL_C0)[tid] = tab0[32 * fb + 8 * g + 4 * h + j];          // (matrix units: c0 2^Ee)
            reinterpret_cast<float *>(smem + GL_C1)[tid] = tab0[3 * 32 * WB + tid];
        }
    }
    // the edge stage's folded bias starts the P_i accumulators (matrix units x the row scale): the row exponent is kept high enough for
    // that product to stay finite -- a row that small against the bias has nothing to lose
    int ex_min = 15;
    {
        const float *tab0 = reinterpret_cast<const float *>(a.prep + W_HDR + SH::F_ALL * 256);
        unsigned m = __float_as_uint(tab0[lane0]) & 0x7fffffffu;
        m = max(m, __float_as_uint(tab0[64 + lane0]) & 0x7fffffffu);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
        const int f = (int)(__builtin_amdgcn_readfirstlane(m) >> 23);
        ex_min = max(15, min(f - 111, 254));
    }
    const int lo_e = (acts & 1) ? 0 : INT_MIN, lo_0 = (acts & 2) ? 0 : INT_MIN;
    const float lo_1 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint((acts & 4) ? 0.f : -INFINITY)));

    // ---- the weight stream: step k's fragments -> ring slot k % 3, this wave's share ------------------------------------------------------
    // One request = one global_load_lds_dwordx4 (1 KiB: 16 bytes per lane, destination = a wave-uniform LDS address in M0 + 16 x lane),
    // address = a wave-uniform base in scalar registers + the lane's 16 bytes as a 32-bit offset.  As an asm statement: the builtin costs
    // two 64-bit vector adds per request (the compiler does not select the scalar-base form), and while a request the compiler knows of
    // is in flight it waits for EVERY ordinary load with vmcnt(0).  The waits for the stream are counted by hand (g_wait) either way.
    const unsigned char *const gfrag = reinterpret_cast<const unsigned char *>(a.prep + W_HDR);
    unsigned lane16 = 16u * (unsigned)lane0;
    auto dma1 = [&](int src_frag, int slot, int dst_frag) {
#ifdef G_ABL_NODMA
        return;
#endif
        const unsigned char *ub = gfrag + (size_t)src_frag * 1024;
        const unsigned ldst = lds0 + (unsigned)(GL_RING + slot * G_SLOT + dst_frag * 1024);
        unsigned keep;
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(lane16), "s"(ub), "s"(ldst) : "memory");
    };
    // n (2 .. 4) CONSECUTIVE fragments -> consecutive ring fragments: one M0 write, the further requests through the instruction offset
    // (it moves the memory address and the LDS address alike)
    auto dman = [&](int src_frag, int slot, int dst_frag, int n) {
#ifdef G_ABL_NODMA
        return;
#endif
        const unsigned char *ub = gfrag + (size_t)src_frag * 1024;
        const unsigned ldst = lds0 + (unsigned)(GL_RING + slot * G_SLOT + dst_frag * 1024);
        unsigned keep;
        if (n == 2)
            asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                         "global_load_lds_dwordx4 %1, %2 offset:1024\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(lane16), "s"(ub), "s"(ldst) : "memory");
        else if (n == 3)
            asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                         "global_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(lane16), "s"(ub), "s"(ldst) : "memory");
        else
            asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                         "global_load_lds_dwordx4 %1, %2 offset:1024\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                         "global_load_lds_dwordx4 %1, %2 offset:3072\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(lane16), "s"(ub), "s"(ldst) : "memory");
    };
    // all requests of step k in as few M0 writes as their placement allows (same loads, same order as dma_one(k, 0 ..))
    auto dma_grouped = [&](int k) {
        const int slot = k % 3;
        if (k < 8) {
            const int runs[3] = {SH::F_WE + 8 * (1 + W_NXC + k), SH::F_WE + 8 * (1 + k), SH::F_W0 + 8 * (SH::NKS + k)};
#pragma unroll
            for (int r = 0; r < 3; ++r) dman(runs[r] + 2 * wave, slot, 8 * r + 2 * wave, 2);
        } else if (k < 12) {
            dman(SH::F_W0 + 16 * (k - 8) + 4 * wave, slot, 4 * wave, 4);
        } else {
            const int kk = k - 12;
            const int per = kk < 2 ? 6 : 4;
            dman(SH::F_W1 + 24 * kk + per * wave, slot, per * wave, per == 6 ? 3 : 4);
            if (per == 6) dman(SH::F_W1 + 24 * kk + per * wave + 3, slot, per * wave + 3, 3);
        }
    };
    // the j-th request of step k (j < g_step_loads(k))
    auto dma_one = [&](int k, int j) {
        const int slot = k % 3;
        if (k < 8) {                                   // chunk k of x against [Wj | Wi | W0x]: two fragments of each of the three runs
            const int runs[3] = {SH::F_WE + 8 * (1 + W_NXC + k), SH::F_WE + 8 * (1 + k), SH::F_W0 + 8 * (SH::NKS + k)};
            const int r = j >> 1, jj = j & 1;
            dma1(runs[r] + 2 * wave + jj, slot, 8 * r + 2 * wave + jj);
        } else if (k < 12) {                           // chunks 2 kk, 2 kk + 1 of the S part of node stage 0
            const int kk = k - 8;
            dma1(SH::F_W0 + 16 * kk + 4 * wave + j, slot, 4 * wave + j);
        } else {                                       // chunks 3 kk .. of node stage 1 (3, 3, 2)
            const int kk = k - 12;
            const int per = kk < 2 ? 6 : 4;
            dma1(SH::F_W1 + 24 * kk + per * wave + j, slot, per * wave + j);
        }
    };
    auto dma = [&](int k) {
#ifndef G_DMA_SINGLE
        dma_grouped(k);
        return;
#endif
#pragma unroll
        for (int j = 0; j < 6; ++j)
            if (j < g_step_loads(k)) dma_one(k, j);
    };

    // ---- this workgroup's graphs; tiles = runs of whole graphs with <= 128 nodes together ------------------------------------------------
    const int g_begin = (int)((long long)a.n_graphs * blockIdx.x / a.n_wg);
    const int g_end = (int)((long long)a.n_graphs * (blockIdx.x + 1) / a.n_wg);
    auto window = [&](int g0, int lane) {              // lane l: node_ptr[g0 + l]
        int idx = g0 + lane;
        idx = idx < g_end ? idx : g_end;
        return reinterpret_cast<const int *>(a.node_ptr)[2 * (size_t)idx];      // (node counts fit 31 bits: the low word)
    };
    auto tile_of = [&](int win, int g0, int lane, int &n0, int &nn) -> int {     // -> graphs taken (0: none left)
        n0 = __builtin_amdgcn_readfirstlane(win);
        nn = 0;
        if (g0 >= g_end) return 0;
        int k = __popcll(__ballot(lane >= 1 && g0 + lane <= g_end && win - n0 <= G_ROWS));
        k = k < 1 ? 1 : k;                             // (a graph above 128 nodes: refused by the host; never loops forever)
        nn = __builtin_amdgcn_readlane(win, k) - n0;
        nn = nn > G_ROWS ? G_ROWS : nn;
        return k;
    };
    auto rows_of = [&](int nn, int w) { const int r = nn - 32 * w; return r < 0 ? 0 : (r > 32 ? 32 : r); };
    auto meta_pt = [&](int n0, int nn, int w, int li, GMeta &m) {
        const int nnw = rows_of(nn, w);
        int node = n0 + 32 * w + (li < nnw ? li : 0);
        node = node < a.n_nodes ? node : a.n_nodes - 1;
        m.node = node;
        m.pt = a.seg_ptr[node]; m.pt1 = a.seg_ptr[node + 1];
    };
    auto meta_idx = [&](int nn, int w, int li, GMeta &m) {
        if (li >= rows_of(nn, w)) { m.pt = 0; m.pt1 = 0; }
#pragma unroll
        for (int it = 0; it < G_PRE; ++it) {
            const int e = it < m.pt1 - m.pt ? m.pt + it : 0;
            m.srcv[it] = a.src[e];
            m.eiv[it] = a.eidx[e];
        }
    };
    auto zload = [&](int lh, int node, int sg, int eg, int j) {
        const WQuad &qa = a.zq[j], &qb = a.zq[2 + j];
        const unsigned long long base = lh ? qb.base : qa.base;
        const unsigned stride = lh ? qb.stride : qa.stride, role = lh ? qb.role : qa.role;
        const unsigned row = role == 0u ? (unsigned)node : (role == 1u ? (unsigned)sg : (unsigned)eg);
        return *reinterpret_cast<gf4_t>(base + (unsigned long long)row * stride);
    };
    auto xload = [&](int n0, int nn, int w, int li, int lh, rr_f4 (&v)[W_NXC][2], int c0, int nc) {     // chunks c0 .. c0 + nc - 1 of the lane's row
        const int nnw = rows_of(nn, w);
        int xrow = n0 + 32 * w + (li < nnw ? li : (nnw > 0 ? nnw - 1 : 0));
        xrow = xrow < a.n_nodes ? xrow : a.n_nodes - 1;
        const rr_f4 *xp = reinterpret_cast<const rr_f4 *>(a.x + (size_t)xrow * W_DX + 8 * lh);
#pragma unroll
        for (int c = 0; c < W_NXC; ++c)
            if (c >= c0 && c < c0 + nc) { v[c][0] = xp[4 * c]; v[c][1] = xp[4 * c + 1]; }
    };

    int g_nxt = g_begin;
    int t_n0, t_nn;
    rr_f4 xraw[W_NXC][2];
    int winN;
    {
        const int win = window(g_begin, lane0);
        g_nxt += tile_of(win, g_begin, lane0, t_n0, t_nn);
        winN = window(g_nxt, lane0);
        xload(t_n0, t_nn, wave, lane0 & 31, lane0 >> 5, xraw, 0, W_NXC);
    }
    // output rows of the tile before: pairs i0 .. i0 + n - 1 of this wave's 32 rows go from LDS to memory
    int p_r0 = 0, p_nnw = 0;
    auto out_rows = [&](int w, int li, int lh, int i0, int n) {
        // (the descriptor from values the compiler KNOWS to be wave-uniform: it wraps every store in a readfirstlane loop otherwise)
        const int r0c = __builtin_amdgcn_readfirstlane(p_r0 < a.n_nodes ? p_r0 : 0), nnwu = __builtin_amdgcn_readfirstlane(p_nnw);
        const __amdgpu_buffer_rsrc_t orow = __builtin_amdgcn_make_buffer_rsrc(a.out + (int64_t)r0c * (32 * WB), 0, nnwu * (32 * WB * 4), 0x00020000);
        int voff = (lh * 32 * WB + 4 * li) * 4;
        asm volatile("" : "+v"(voff));
        const unsigned rrow = lds0 + GL_PJ + (unsigned)(32 * w) * G_PITCH + 16u * (unsigned)li + (unsigned)lh * G_PITCH;
        rr_f4 ov[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i >= i0 && i < i0 + n) ov[i] = *reinterpret_cast<ldsf4_t>(rrow + (unsigned)(2 * i * G_PITCH));
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i >= i0 && i < i0 + n) {
#ifdef G_ABL_NOSTORE
                if (ov[i].x == 12345.678f)
#endif
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(rr_u4, ov[i]), orow, voff, 2 * i * (32 * WB * 4), 0);
            }
    };
    dma(0);
    dma(1);
    __syncthreads();                                   // (drains the two steps' loads as well)
    const unsigned t_start = clk();

    while (true) {
        const unsigned q0 = clk();
        // (everything made from the lane / wave index is made per tile: hoisted out of the loop it is spilled, and a reload waits for
        //  every load in flight)
        int lane = lane0;
        asm volatile("" : "+v"(lane16), "+v"(wave_v), "+v"(lane));
        wave = __builtin_amdgcn_readfirstlane(wave_v);
        const int li = lane & 31, lh = lane >> 5;
        const unsigned ring_lane = lds0 + GL_RING + 16u * (unsigned)lane;
#ifdef G_ABL_NOFRAG
        auto frag = [&](int slot, int f) { return rr_u4{ring_lane, (unsigned)f, (unsigned)slot, 0x3c003c00u}; };
#else
        auto frag = [&](int slot, int f) { return *reinterpret_cast<rr_ldsp>(ring_lane + (unsigned)(slot * G_SLOT + f * 1024)); };
#endif
        const int n0 = t_n0, nn = t_nn;
        const int r0 = n0 + 32 * wave, nnw = rows_of(nn, wave);
        // ---- the next tile (its window came in under the last one); the window behind it ------------------------------------------------
        int n_n0, n_nn;
        const int k_n = tile_of(winN, g_nxt, lane, n_n0, n_nn);
        const bool more = k_n > 0;
        g_nxt += k_n;
        GMeta M;
        rr_f4 zr[G_PRE][2];

        // ---- x rows of this wave's 32 nodes as two fp16 planes in operand layout (lane (t, h): columns 16 c + 8 h .. + 8 of row t) -----------
        rr_u4 xh[W_NXC], xl[W_NXC];
        bool bad_x;
        float facx, rs_x;                              // accumulator of a product of x (units rs_x 2^E) -> true value: x 2^-E further
        int e_x;
        {
            unsigned m = 0;
#pragma unroll
            for (int c = 0; c < W_NXC; ++c)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    m = max(max(m, __float_as_uint(xraw[c][j].x) & 0x7fffffffu), __float_as_uint(xraw[c][j].y) & 0x7fffffffu);
                    m = max(max(m, __float_as_uint(xraw[c][j].z) & 0x7fffffffu), __float_as_uint(xraw[c][j].w) & 0x7fffffffu);
                }
            m = rr_xhalf_max(m);
            int e = (int)(m >> 23);
            bad_x = e >= 255;
            e = e < ex_min ? ex_min : (e > 254 ? 254 : e);
            e_x = e;
            const float rs = __uint_as_float((unsigned)(268 - e) << 23);
            rs_x = rs;
#pragma unroll
            for (int c = 0; c < W_NXC; ++c) {
                unsigned h[4], l[4];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    rr_split2s(xraw[c][j].x, xraw[c][j].y, rs, h[2 * j], l[2 * j]);
                    rr_split2s(xraw[c][j].z, xraw[c][j].w, rs, h[2 * j + 1], l[2 * j + 1]);
                }
                xh[c] = rr_u4{h[0], h[1], h[2], h[3]};
                xl[c] = rr_u4{l[0], l[1], l[2], l[3]};
            }
            facx = bad_x ? __uint_as_float(0x7fc00000u) : rr_pow2(e_x - 14 - Ee);
        }
        RR_SB();                                       // (the planes are made here: sunk into the product steps the raw rows stay alive beside them)
        const unsigned q1 = clk();

        // =====================================================================================================================================
        // steps 0 .. 7: chunk c of the x rows against [Wj | Wi | W0x]:  P_j^T, P_i^T, Hx^T  (feature x node accumulators)
        // =====================================================================================================================================
        f32x16 pj[WB], pi[WB], hx[WB];
        {
            const unsigned c0p = lds0 + GL_C0 + 256u * (unsigned)lh;
#pragma unroll
            for (int fb = 0; fb < WB; ++fb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const rr_f4 c0 = *reinterpret_cast<ldsf4_t>(c0p + 64u * fb + 16u * g);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { pj[fb][4 * g + j] = 0.f; pi[fb][4 * g + j] = c0[j] * rs_x; hx[fb][4 * g + j] = 0.f; }
                }
        }
#pragma unroll
        for (int c = 0; c < W_NXC; ++c) {
            // this tile's targets: segment bounds (and the window behind the next tile), two steps later their first in-edges' indices, three
            // steps later those edges' columns -- each IN FRONT of the step's request of the weight stream: the compiler waits for a loaded
            // value with vmcnt(0) while stream loads are in flight, and here the youngest of those is a step old
            const unsigned d0 = clk();
            if (c == 0) { meta_pt(n0, nn, wave, li, M); winN = window(g_nxt, lane); }
            if (c == 2) meta_idx(nn, wave, li, M);
            if (c == 5) {
#pragma unroll
                for (int it = 0; it < G_PRE; ++it)
#pragma unroll
                    for (int j = 0; j < 2; ++j) zr[it][j] = HASZ ? zload(lh, M.node, M.srcv[it], M.eiv[it], j) : rr_f4{0.f, 0.f, 0.f, 0.f};
            }
            RR_SB();
            const unsigned d1 = clk();
            const unsigned d2 = d1;
            // The step in issue order -- a wave issues in order, and what stands between two MFMAs runs under the first one: every group
            // of twelve products carries the next group's eight fragment reads, two requests of the stream (step c + 2) and, in the
            // first two groups, one pair of the previous tile's output rows (read from LDS in group 0, stored in group 1).
            const int slot = c % 3, kq = (c + 2) % G_NSTEP;
            rr_u4 f[8], fn[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) f[q] = frag(slot, q);
#ifndef G_NOHEAD
            // the step's first eight fragment reads are in flight while the requests of step c + 2 issue: their issue time (~60 cycles
            // each between products) runs under the reads' latency instead of between two products
            RR_SB();
            dma(kq);
            RR_SB();
#endif
            rr_f4 ov[2];
            const int r0p = __builtin_amdgcn_readfirstlane(p_r0 < a.n_nodes ? p_r0 : 0), nnwp = __builtin_amdgcn_readfirstlane(p_nnw);
            const __amdgpu_buffer_rsrc_t orow = __builtin_amdgcn_make_buffer_rsrc(a.out + (int64_t)r0p * (32 * WB), 0, nnwp * (32 * WB * 4), 0x00020000);
            int voff = (lh * 32 * WB + 4 * li) * 4;
            asm volatile("" : "+v"(voff));
            const unsigned rrow = lds0 + GL_PJ + (unsigned)(32 * wave) * G_PITCH + 16u * (unsigned)li + (unsigned)lh * G_PITCH;
#pragma unroll
            for (int grp = 0; grp < 3; ++grp) {
                f32x16 (&acc)[WB] = grp == 0 ? pj : (grp == 1 ? pi : hx);
#pragma unroll
                for (int th = 0; th < 3; ++th) {
                    // four products, the reads that go with them (3, 3, 2 of the next group's eight)
                    if (grp < 2) {
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            if (q >= 3 * th && q < (th == 2 ? 8 : 3 * th + 3)) fn[q] = frag(slot, 8 * (grp + 1) + q);
                    }
                    if (grp == 0 && th < 2) ov[th] = *reinterpret_cast<ldsf4_t>(rrow + (unsigned)(2 * (2 * c + th) * G_PITCH));
#pragma unroll
                    for (int fb = 0; fb < WB; ++fb) {
                        if (th == 0) RR_MFH(f[2 * fb], xl[c], acc[fb]);
                        if (th == 1) RR_MFH(f[2 * fb + 1], xh[c], acc[fb]);
                        if (th == 2) RR_MFH(f[2 * fb], xh[c], acc[fb]);
                    }
                    G_MIX(4, grp < 2 ? 4 : 0)
                    RR_SB();
#ifdef G_NOHEAD
                    if (th < 2 && 2 * grp + th < g_step_loads(kq)) dma_one(kq, 2 * grp + th);
#endif
                    if (grp == 1 && th < 2) {
#ifdef G_ABL_NOSTORE
                        if (ov[th].x == 12345.678f)
#endif
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(rr_u4, ov[th]), orow, voff, 2 * (2 * c + th) * (32 * WB * 4), 0);
                    }
                    RR_SB();
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) f[q] = fn[q];
            }
            if (c == W_NXC - 1) {
                // ---- P_j^T -> LDS rows (true units), the row's bad flag ----------------------------------------------------------------------
                const unsigned q2 = clk();
                const unsigned rowp = lds0 + GL_PJ + (unsigned)(32 * wave + li) * G_PITCH + 256u * (unsigned)lh;
#pragma unroll
                for (int fb = 0; fb < WB; ++fb)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<ldsf4_t>(rowp + 64u * fb + 16u * g) =
                            rr_f4{pj[fb][4 * g] * facx, pj[fb][4 * g + 1] * facx, pj[fb][4 * g + 2] * facx, pj[fb][4 * g + 3] * facx};
                if (lh == 0) *reinterpret_cast<ldsi_t>(lds0 + GL_BAD + 4u * (unsigned)(32 * wave + li)) = bad_x ? 1 : 0;
                if (PROF) pc[3] += clk() - q2;
            }
            const unsigned d3 = clk();
            g_vmcnt(g_wait(c));
            const unsigned d4 = clk();
            G_BARRIER();
            if (PROF) { const unsigned d5 = clk(); pd[0] += d1 - d0; pd[1] += d2 - d1; pd[2] += d3 - d2; pd[3] += d4 - d3; pd[4] += d5 - d4; }
        }
        const unsigned q3 = clk();

        // =====================================================================================================================================
        // edge phase: lane (t, h) walks the in-edges of target t; S^T in the accumulator layout (64 features of node t per lane)
        // =====================================================================================================================================
        if (HASZ) {                                    // (the per-edge columns are waited for HERE, in front of the next request of the stream)
#pragma unroll
            for (int it = 0; it < G_PRE; ++it)
#pragma unroll
                for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(zr[it][j]));
        }
        RR_SB();
        xload(n_n0, n_nn, wave, li, lh, xraw, 0, 1);   // (the next tile's x rows, a chunk per step from here on)
        dma(10);                                       // (step 8's request of step 10: in flight under the edge phase)
        f32x16 S[WB];
        unsigned badt = 0;
        const int deg = M.pt1 - M.pt;
        {
            // P_i in true units (the folded bias came with the accumulators' start values)
#pragma unroll
            for (int fb = 0; fb < WB; ++fb)
#pragma unroll
                for (int r = 0; r < 16; ++r) pi[fb][r] *= facx;
#pragma unroll
            for (int fb = 0; fb < WB; ++fb)
#pragma unroll
                for (int r = 0; r < 16; ++r) S[fb][r] = 0.f;
            const unsigned wzp = lds0 + GL_WZ + 16u * (unsigned)lane;
            if (PROF) pe[0] += clk() - q3;
            // one round: the it-th in-edge of every target.  An idle lane (fewer in-edges) adds 0 x (values of its own row: finite unless
            // the row is marked anyway)
            auto edge_round = [&](bool act, int s_glob, rr_f4 z0, rr_f4 z1) {
                int sl = s_glob - n0;
                sl = act ? sl : 32 * wave + li;
                sl = sl < 0 ? 0 : (sl > G_ROWS - 1 ? G_ROWS - 1 : sl);
                const int bad_s = *reinterpret_cast<ldsi_t>(lds0 + GL_BAD + 4u * (unsigned)sl);
                const unsigned rowp = lds0 + GL_PJ + (unsigned)sl * G_PITCH + 256u * (unsigned)lh;
                const float mk = act ? 1.f : 0.f;
                const rr_f2 mk2 = rr_f2{mk, mk};
                float invz = 0.f;
                bool bad_z = false;
                rr_u4 zh_ = rr_u4{0u, 0u, 0u, 0u}, zl_ = zh_;
                if (HASZ) {
                    unsigned m = max(max(__float_as_uint(z0.x) & 0x7fffffffu, __float_as_uint(z0.y) & 0x7fffffffu), max(__float_as_uint(z0.z) & 0x7fffffffu, __float_as_uint(z0.w) & 0x7fffffffu));
                    m = max(m, max(max(__float_as_uint(z1.x) & 0x7fffffffu, __float_as_uint(z1.y) & 0x7fffffffu), max(__float_as_uint(z1.z) & 0x7fffffffu, __float_as_uint(z1.w) & 0x7fffffffu)));
                    m = act ? m : 0u;
                    m = rr_xhalf_max(m);
                    int e = (int)(m >> 23);
                    bad_z = e >= 255;
                    e = e < 15 ? 15 : (e > 254 ? 254 : e);
                    const float rs = __uint_as_float((unsigned)(268 - e) << 23);
                    invz = rr_pow2(e - 14 - Ee);
                    unsigned h[4], l[4];
                    rr_split2s(z0.x, z0.y, rs, h[0], l[0]);
                    rr_split2s(z0.z, z0.w, rs, h[1], l[1]);
                    rr_split2s(z1.x, z1.y, rs, h[2], l[2]);
                    rr_split2s(z1.z, z1.w, rs, h[3], l[3]);
                    rr_u4 zh = rr_u4{h[0], h[1], h[2], h[3]}, zl = rr_u4{l[0], l[1], l[2], l[3]};
                    if (!act || bad_z) { zh = rr_u4{0u, 0u, 0u, 0u}; zl = zh; }
                    zh_ = zh; zl_ = zl;
                }
                const rr_f2 iz2 = rr_f2{invz, invz};
                // two feature blocks at a time: the source's P_j values and the fragments are requested first (their latency runs under the
                // six products), then the vector work
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    rr_f4 pv[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) pv[i] = *reinterpret_cast<ldsf4_t>(rowp + 128u * hf + 16u * i);
                    f32x16 q[2];
                    if (HASZ) {
                        rr_u4 wz[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) wz[i] = *reinterpret_cast<rr_ldsp>(wzp + 1024u * (4 * hf + i));
#pragma unroll
                        for (int b = 0; b < 2; ++b)
#pragma unroll
                            for (int r = 0; r < 16; ++r) q[b][r] = 0.f;
#pragma unroll
                        for (int b = 0; b < 2; ++b) RR_MFH(wz[2 * b], zl_, q[b]);
#pragma unroll
                        for (int b = 0; b < 2; ++b) RR_MFH(wz[2 * b + 1], zh_, q[b]);
#pragma unroll
                        for (int b = 0; b < 2; ++b) RR_MFH(wz[2 * b], zh_, q[b]);
                    }
                    RR_SB();
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const int fb = 2 * hf + b;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
#pragma unroll
                            for (int j = 0; j < 4; j += 2) {
                                const int r = 4 * g + j;
                                rr_f2 v = rr_f2{pi[fb][r], pi[fb][r + 1]};
                                if (HASZ) v = rr_f2{q[b][r], q[b][r + 1]} * iz2 + v;
                                v += rr_f2{pv[4 * b + g][j], pv[4 * b + g][j + 1]};
                                const rr_f2 y = rr_f2{rr_imax(v[0], lo_e), rr_imax(v[1], lo_e)} * mk2 + rr_f2{S[fb][r], S[fb][r + 1]};
                                S[fb][r] = y[0]; S[fb][r + 1] = y[1];
                            }
                        }
                    }
                    RR_SB();
                }
                if (act && (bad_s || bad_z)) badt = 1;
            };
#ifdef G_ABL_NOEDGE
            const int deg_r = 0;
#else
            const int deg_r = deg;
#endif
#pragma unroll
            for (int it = 0; it < G_PRE; ++it)
                if (__builtin_amdgcn_ballot_w64(it < deg_r) != 0ull) {
                    const unsigned e0 = clk();
                    edge_round(it < deg, M.srcv[it], zr[it][0], zr[it][1]);
                    if (PROF) { pe[1] += clk() - e0; pe[2] += 1; }
                }
            for (int it = G_PRE; __builtin_amdgcn_ballot_w64(it < deg_r) != 0ull; ++it) {        // hubs: on demand
                const bool act = it < deg;
                const int e = act ? M.pt + it : 0;
                const int sg = a.src[e], eg = a.eidx[e];
                rr_f4 z[2] = {rr_f4{0.f, 0.f, 0.f, 0.f}, rr_f4{0.f, 0.f, 0.f, 0.f}};
                if (HASZ) { z[0] = zload(lh, M.node, sg, eg, 0); z[1] = zload(lh, M.node, sg, eg, 1); }
                edge_round(act, sg, z[0], z[1]);
            }
        }
        const unsigned q4 = clk();
        // =====================================================================================================================================
        // node stage 0, S part (transposed):  Hs^T = W0s S^T + (c0 + deg w_deg);  h = act_0(Hs + Hx)
        // =====================================================================================================================================
        float ms = 0.f;
#pragma unroll
        for (int fb = 0; fb < WB; ++fb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) ms = fmaxf(fmaxf(fabsf(S[fb][r]), fabsf(S[fb][r + 1])), ms);
        const float degf = (float)deg;
        const unsigned msb = __float_as_uint(ms);
        unsigned fld = (unsigned)max(max((int)(msb >> 23), (int)(__float_as_uint(degf) >> 23)), 0);
        fld = rr_xhalf_max(fld);
        bool badrow = msb >= 0x7f800000u || bad_x || badt != 0 || w_bad;
        badrow = rr_xhalf_or(badrow ? 1u : 0u) != 0;
        int e_t = (int)fld;
        e_t = e_t < e_min ? e_min : (e_t > 254 ? 254 : e_t);
        const float rs0 = __uint_as_float((unsigned)(268 - e_t) << 23);
        f32x16 hacc[WB];
        {
            // accumulators start at (c0 + deg w_deg) 2^E0 rs0: one bf16 product per feature block (layer_w.hip)
            const float dh = __uint_as_float(__float_as_uint(degf) & 0xffff0000u), dl = degf - dh;
            const unsigned rsb = __float_as_uint(rs0) >> 16, dhb = __float_as_uint(dh * rs0) >> 16, dlb = __float_as_uint(dl * rs0) >> 16;
            const rr_u4 bv = lh ? rr_u4{dlb, 0u, 0u, 0u} : rr_u4{rsb | (rsb << 16), rsb | (dhb << 16), dhb | (dhb << 16), dlb | (dlb << 16)};
            const unsigned wbp = lds0 + GL_WZ + 8192u + 16u * (unsigned)lane;
#pragma unroll
            for (int fbo = 0; fbo < WB; ++fbo) {
                const rr_u4 bfr = *reinterpret_cast<rr_ldsp>(wbp + 1024u * fbo);
#pragma unroll
                for (int r = 0; r < 16; ++r) hacc[fbo][r] = 0.f;
                RR_MFB(bfr, bv, hacc[fbo]);
            }
        }
        {
            // eight chunks of S in four steps of two; chunk c + 1's planes and fragments are made / read between chunk c's products
            unsigned ph[4], pl[4], nph[4], npl[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) rr_split2s(S[0][2 * q], S[0][2 * q + 1], rs0, ph[q], pl[q]);
            rr_u4 f[8], fn[8];
#pragma unroll
            for (int c = 0; c < SH::NKS; ++c) {
                const int kk = c >> 1, cc = c & 1, k = 8 + kk, slot = k % 3;
                if (cc == 0) {
#ifdef G_NOHEAD
                    if (kk > 0) { xload(n_n0, n_nn, wave, li, lh, xraw, kk, 1); dma((k + 2) % G_NSTEP); }
#endif
#pragma unroll
                    for (int q = 0; q < 8; ++q) f[q] = frag(slot, q);
#pragma unroll
                    for (int q = 0; q < 8; ++q) fn[q] = frag(slot, 8 + q);
#ifndef G_NOHEAD
                    RR_SB();
                    if (kk > 0) { xload(n_n0, n_nn, wave, li, lh, xraw, kk, 1); dma((k + 2) % G_NSTEP); }
                    RR_SB();
#endif
                }
                const rr_u4 bh = rr_u4{ph[0], ph[1], ph[2], ph[3]}, bl = rr_u4{pl[0], pl[1], pl[2], pl[3]};
#pragma unroll
                for (int fbo = 0; fbo < WB; ++fbo) RR_MFH(f[2 * fbo], bl, hacc[fbo]);
#pragma unroll
                for (int fbo = 0; fbo < WB; ++fbo) RR_MFH(f[2 * fbo + 1], bh, hacc[fbo]);
#pragma unroll
                for (int fbo = 0; fbo < WB; ++fbo) RR_MFH(f[2 * fbo], bh, hacc[fbo]);
                if (c + 1 < SH::NKS) {
                    const int c1 = c + 1, fb1 = c1 >> 1, cc1 = c1 & 1;
#pragma unroll
                    for (int q = 0; q < 4; ++q) rr_split2s(S[fb1][8 * cc1 + 2 * q], S[fb1][8 * cc1 + 2 * q + 1], rs0, nph[q], npl[q]);
                }
                G_MIXV(12, cc == 0 ? 8 : 0, 2)
                RR_SB();
#pragma unroll
                for (int q = 0; q < 4; ++q) { ph[q] = nph[q]; pl[q] = npl[q]; }
#pragma unroll
                for (int q = 0; q < 8; ++q) f[q] = fn[q];
                if (cc == 1) {
                    g_vmcnt(g_wait(8 + kk));
                    G_BARRIER();
                }
            }
        }
        const unsigned q5 = clk();
        // ---- between the stages: h = act_0(Hs 2^(e_t - 141 - E0) + Hx 2^(e_x - 141 - E0)) in true units, its row scale -------------------------
        float f2, inv2;
        bool anybad;
        {
            const float fa = rr_pow2(e_t - 14 - E0);
            const float fxh = bad_x ? __uint_as_float(0x7fc00000u) : rr_pow2(e_x - 14 - E0);
            float m2 = 0.f;
#pragma unroll
            for (int fbo = 0; fbo < WB; ++fbo)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float h = rr_imax(fmaf(hx[fbo][r], fxh, hacc[fbo][r] * fa), lo_0);
                    hacc[fbo][r] = h;
                    m2 = fmaxf(m2, fabsf(h));
                }
            const unsigned m2b = rr_xhalf_max(__float_as_uint(m2));
            int e2 = (int)(m2b >> 23);
            e2 = e2 < 15 ? 15 : (e2 > 254 ? 254 : e2);
            f2 = __uint_as_float((unsigned)(268 - e2) << 23);
            inv2 = rr_pow2(e2 - 14 - E1);
            if (m2b >= 0x7f800000u || badrow) { f2 = __uint_as_float(0x7fc00000u); inv2 = f2; badrow = true; }
            anybad = __builtin_amdgcn_ballot_w64(badrow) != 0ull;
        }
        const unsigned q6 = clk();
        // =====================================================================================================================================
        // node stage 1:  OUT = H W1^T  (A = the H^T tiles as operand fragments)
        // =====================================================================================================================================
        f32x16 oacc[WB];
#pragma unroll
        for (int fb = 0; fb < WB; ++fb)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[fb][r] = 0.f;
        {
            // eight chunks of H in three steps of 3, 3, 2
            unsigned ph[4], pl[4], nph[4], npl[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) rr_split2s(hacc[0][2 * q], hacc[0][2 * q + 1], f2, ph[q], pl[q]);
            rr_u4 f[8], fn[8];
#pragma unroll
            for (int c = 0; c < SH::NKS; ++c) {
                const int kk = c / 3, cc = c % 3, k = 12 + kk, slot = k % 3;
                const bool last_of_step = cc == 2 || c == SH::NKS - 1;
                if (cc == 0) {
#ifdef G_NOHEAD
                    xload(n_n0, n_nn, wave, li, lh, xraw, 4 + kk, 1);
                    dma((k + 2) % G_NSTEP);
#endif
#pragma unroll
                    for (int q = 0; q < 8; ++q) f[q] = frag(slot, q);
#ifndef G_NOHEAD
                    RR_SB();
                    xload(n_n0, n_nn, wave, li, lh, xraw, 4 + kk, 1);
                    dma((k + 2) % G_NSTEP);
                    RR_SB();
#endif
                }
                if (!last_of_step) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) fn[q] = frag(slot, 8 * (cc + 1) + q);
                }
                const rr_u4 bh = rr_u4{ph[0], ph[1], ph[2], ph[3]}, bl = rr_u4{pl[0], pl[1], pl[2], pl[3]};
#pragma unroll
                for (int fb = 0; fb < WB; ++fb) RR_MFH(bl, f[2 * fb], oacc[fb]);
#pragma unroll
                for (int fb = 0; fb < WB; ++fb) RR_MFH(bh, f[2 * fb + 1], oacc[fb]);
#pragma unroll
                for (int fb = 0; fb < WB; ++fb) RR_MFH(bh, f[2 * fb], oacc[fb]);
                if (c + 1 < SH::NKS) {
                    const int c1 = c + 1, fb1 = c1 >> 1, cc1 = c1 & 1;
#pragma unroll
                    for (int q = 0; q < 4; ++q) rr_split2s(hacc[fb1][8 * cc1 + 2 * q], hacc[fb1][8 * cc1 + 2 * q + 1], f2, nph[q], npl[q]);
                }
                G_MIXV(12, last_of_step ? 0 : 8, 2)
                RR_SB();
#pragma unroll
                for (int q = 0; q < 4; ++q) { ph[q] = nph[q]; pl[q] = npl[q]; }
#pragma unroll
                for (int q = 0; q < 8; ++q) f[q] = fn[q];
                if (last_of_step) {
                    g_vmcnt(g_wait(12 + kk));
                    G_BARRIER();
                }
            }
        }
        const unsigned q7 = clk();
        // ---- output rows: through this wave's (now idle) P_j rows in LDS, so that they leave as whole 512-byte rows, 16 bytes per lane
        //      (row = accumulator register: 64 dword stores per lane otherwise, and they queue behind each other) ----------------------------
        {
            float cb1[WB];
#pragma unroll
            for (int fb = 0; fb < WB; ++fb) cb1[fb] = *reinterpret_cast<ldsf_t>(lds0 + GL_C1 + 4u * (unsigned)(32 * fb + li));
            float invr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) invr[r] = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * rr_crow(r, lh), __float_as_int(inv2)));
            const unsigned wrow = lds0 + GL_PJ + (unsigned)(32 * wave) * G_PITCH;
            const unsigned wcol = wrow + 4u * (unsigned)li + (unsigned)(4 * lh) * G_PITCH;
#pragma unroll
            for (int fb = 0; fb < WB; ++fb) {
                const float cb = cb1[fb];
                auto put = [&](auto nanrows) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float y = fmaxf(fmaf(oacc[fb][r], invr[r], cb), lo_1);
                        if (decltype(nanrows)::value) y = invr[r] != invr[r] ? invr[r] : y;     // (the max drops a NaN)
                        *reinterpret_cast<ldsf_t>(wcol + (unsigned)(((r & 3) + 8 * (r >> 2)) * G_PITCH + 128 * fb)) = y;
                    }
                };
                if (anybad) put(std::true_type{}); else put(std::false_type{});
            }
            xload(n_n0, n_nn, wave, li, lh, xraw, 7, 1);
            // the rows stay there: they leave two per product step of the NEXT tile (out_rows), behind the wave's own reads of them -- a burst of
            // sixteen 1 KiB stores per wave queues at ~10 bytes per cycle and CU (6 800 cycles per tile when issued here)
            p_r0 = r0; p_nnw = nnw;
        }
        if (PROF) {
            const unsigned q8 = clk();
            pc[0] += q1 - q0; pc[2] += q3 - q1; pc[4] += q4 - q3; pc[5] += q5 - q4; pc[6] += q6 - q5; pc[7] += q7 - q6; pc[8] += q8 - q7; pc[9] += 1;
        }
        if (!more) break;
        t_n0 = n_n0; t_nn = n_nn;
    }
    out_rows(wave, lane0 & 31, lane0 >> 5, 0, 16);        // (the last tile's output rows)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the ring's last two prefetched steps land before the LDS is given back)
    if (PROF && prof && lane0 == 0 && blockIdx.x == gridDim.x / 2) {
        unsigned long long *o = prof + 16 * (tid >> 6);
        for (int q = 0; q < 10; ++q) o[q] = pc[q];
        o[10] = clk() - t_start;
        for (int q = 0; q < 5; ++q) o[11 + q] = pd[q];
        if (tid >> 6) { o[11] = pe[0]; o[12] = pe[1]; o[13] = pe[2]; }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
static bool g_enabled() {
    static const int on = [] { const char *d = getenv("GSN_FUSED_G"); return d ? atoi(d) : 1; }();
    return on != 0;
}

int g_supported(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0, const gsn_chain_stage *node1) {
    return g_enabled() && w_supported(edge, d_x, node0, node1);
}

int g_forward(int64_t n_nodes, int64_t n_edges, const int32_t *seg_ptr, const gsn_chain_stage *edge, const float *x, int64_t d_x,
              const gsn_chain_stage *node0, const gsn_chain_stage *node1, const void *prepared, int64_t n_graphs, const int64_t *node_ptr,
              int64_t max_nodes, float *out, hipStream_t st) {
    (void)d_x; (void)node0; (void)node1;
    if (max_nodes > G_MAX_GRAPH_NODES)
        return set_error(GSN_E_UNSUPPORTED, "gsn_layer_fused_fwd_graphs_hip: a graph of %lld nodes (the graph-aligned kernel takes <= %d)", (long long)max_nodes, G_MAX_GRAPH_NODES);
    if (edge->blocks[0].data != x) return set_error(GSN_E_UNSUPPORTED, "gsn_layer_fused_fwd_graphs_hip: the first two edge blocks must be x itself");
    if (n_graphs <= 0 || !node_ptr) return set_error(GSN_E_INVALID, "gsn_layer_fused_fwd_graphs_hip: no graph boundaries");
    GArgs a{};
    a.n_nodes = (int)n_nodes; a.n_edges = (int)n_edges; a.n_graphs = (int)n_graphs; a.seg_ptr = seg_ptr;
    a.node_ptr = reinterpret_cast<const long long *>(node_ptr);
    a.x = x; a.out = out; a.prep = reinterpret_cast<const unsigned *>(prepared);
    const int32_t *ridx[3] = {edge->blocks[0].idx32, edge->blocks[1].idx32, nullptr};
    int nq = 0;
    for (int b = 2; b < edge->n_blocks; ++b) {
        const gsn_block &bl = edge->blocks[b];
        unsigned role;
        if (bl.idx32 == ridx[0]) role = 0;
        else if (bl.idx32 == ridx[1]) role = 1;
        else { ridx[2] = bl.idx32; role = 2; }
        for (int q = 0; q < (int)bl.width / 4; ++q) {
            a.zq[nq].base = reinterpret_cast<unsigned long long>(bl.data) + 16ull * q;
            a.zq[nq].stride = (unsigned)(bl.width * 4);
            a.zq[nq].role = role;
            ++nq;
        }
    }
    const bool has_z = nq > 0 && n_edges > 0;
    a.src = ridx[1];
    a.eidx = ridx[2] ? ridx[2] : ridx[1];
    for (int q = nq; q < 4; ++q) {                        // columns past K: finite data of the same rows (their weights are zero)
        if (nq > 0) a.zq[q] = a.zq[0];
        else { a.zq[q].base = reinterpret_cast<unsigned long long>(x); a.zq[q].stride = 0; a.zq[q].role = 0; }
    }
    if (n_edges == 0) {                                   // (no edge row is ever addressed: every load of the index arrays lands on seg_ptr[0])
        a.src = seg_ptr; a.eidx = seg_ptr;
        for (int q = 0; q < 4; ++q) { a.zq[q].base = reinterpret_cast<unsigned long long>(x); a.zq[q].stride = 0; a.zq[q].role = 0; }
    }
    int64_t gx = 256;
    { const char *d = getenv("GSN_FUSED_GRID"); if (d && atoi(d) > 0) gx = atoi(d); }
    if (gx > n_graphs) gx = n_graphs;
    a.n_wg = (int)gx;
    static const bool prof_on = [] { const char *d = getenv("GSN_FUSED_PROF"); return d && atoi(d) != 0; }();
    static DeviceOnce attr_set;
    const int attr_dev = current_device();
    if (!attr_set.done(attr_dev)) {
        hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(&layer_fused_kernel_g<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e0 == hipSuccess) e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(&layer_fused_kernel_g<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e0 == hipSuccess) e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(&layer_fused_kernel_g<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e0 != hipSuccess) return set_error(GSN_E_HIP, "hipFuncSetAttribute(layer_fused_kernel_g): %s", hipGetErrorString(e0));
        attr_set.mark(attr_dev);
    }
    if (getenv("GSN_CHAIN_TRACE")) fprintf(stderr, "gsn chain: layer_fused_kernel_g nodes %d edges %d graphs %d grid %lld\n", a.n_nodes, a.n_edges, a.n_graphs, (long long)gx);
    if (prof_on && has_z) {
        unsigned long long *prof = nullptr;
        (void)hipMalloc(&prof, 64 * 8); (void)hipMemsetAsync(prof, 0, 64 * 8, st);
        hipLaunchKernelGGL((layer_fused_kernel_g<true, true>), dim3((unsigned)gx), dim3(256), GL_TOTAL, st, a, prof);
        unsigned long long h[64];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost);
        (void)hipFree(prof);
        static int shown = 0;
        if (shown++ % 8 == 7)
            for (int w = 0; w < 4; ++w) {
                const unsigned long long *o = h + 16 * w;
                const double nt = o[9] ? (double)o[9] : 1.0;
                if (w) fprintf(stderr, "gprof wave %d edge phase per tile: in front of the rounds %.0f, rounds %.0f cycles in %.2f rounds\n", w, o[11] / nt, o[12] / nt, o[13] / nt);
                else fprintf(stderr, "gprof wave %d product steps per tile: other loads + output rows %.0f stream request %.0f reads + products %.0f wait %.0f barrier %.0f\n", w, o[11] / nt, o[12] / nt, o[13] / nt, o[14] / nt, o[15] / nt);
                fprintf(stderr, "gprof wave %d: tiles %llu total %llu cycles | per tile: top+x %.0f products %.0f (P_j write %.0f) edges %.0f stage0 %.0f between %.0f stage1 %.0f stores %.0f\n",
                        w, o[9], o[10], o[0] / nt, o[2] / nt, o[3] / nt, o[4] / nt, o[5] / nt, o[6] / nt, o[7] / nt, o[8] / nt);
            }
    } else if (has_z) {
        hipLaunchKernelGGL((layer_fused_kernel_g<false, true>), dim3((unsigned)gx), dim3(256), GL_TOTAL, st, a, (unsigned long long *)nullptr);
    } else {
        hipLaunchKernelGGL((layer_fused_kernel_g<false, false>), dim3((unsigned)gx), dim3(256), GL_TOTAL, st, a, (unsigned long long *)nullptr);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "layer_fused_kernel_g: %s", hipGetErrorString(e));
    return GSN_OK;
}

}  // namespace gsn
