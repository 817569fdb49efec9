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

namespace gsn {

constexpr int G_ROWS = G_MAX_GRAPH_NODES;          // nodes per workgroup tile
constexpr int G_PITCH = 528;                       // bytes per P_j row in LDS: [half][feature block][4-group][4] floats + 16 (bank spread)
constexpr int G_SLOT = 24 * 1024;                  // ring slot: 24 fragments
constexpr int G_NSTEP = 15;                        // steps of the cyclic weight schedule: 8 (products of x) + 4 (S part of node stage 0) + 3 (node stage 1)
constexpr int G_PRE = 4;                           // in-edges per target whose indices and per-edge columns are requested at the top of a tile
constexpr int GL_PJ = 0;
constexpr int GL_BAD = GL_PJ + G_ROWS * G_PITCH;   // 128 ints: the node's row holds an Inf / NaN
constexpr int GL_C0 = GL_BAD + 4 * G_ROWS;         // the edge stage's folded bias in the accumulator order [half][fb][group][4]
constexpr int GL_WZ = GL_C0 + 512;                 // 8 fragments of the per-edge chunk + 4 of node stage 0's (c0, deg) product
constexpr int GL_RING = GL_WZ + 12 * 1024;
constexpr int GL_TOTAL = GL_RING + 3 * G_SLOT;
static_assert(GL_TOTAL <= 160 * 1024, "LDS budget");
static_assert(GL_RING % 1024 == 0 || true, "");

struct GArgs {
    int n_nodes, n_edges, n_graphs, n_wg;
    const int32_t *seg_ptr;
    const int32_t *src;               // source node of every edge row (target-sorted order)
    const int32_t *eidx;              // row of the per-edge blocks of every edge row (perm), or = src when unused
    const long long *node_ptr;
    WQuad zq[4];                      // the four 16-byte quads of the per-edge chunk: role 0 = row of the target, 1 = of the source, 2 = eidx
    int has_z;
    const float *x;
    float *out;
    const unsigned *prep;
};

__device__ __forceinline__ constexpr int g_step_loads(int k) { return (k < 8 || k == 12 || k == 13) ? 6 : 4; }

template <bool PROF>
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
    unsigned pc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // top, x planes, product steps, P_j write, edge phase, stage 0, between, stage 1, stores, tiles

    const int tid = threadIdx.x;
    const int lane0 = tid & 63;
    int wave_v = tid >> 6;
    int wave = __builtin_amdgcn_readfirstlane(wave_v);
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    typedef __attribute__((address_space(3))) rr_f4 *ldsf4_t;
    typedef __attribute__((address_space(3))) int *ldsi_t;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;

    // ---- prologue: the per-edge chunk's fragments, node stage 0's (c0, deg) fragments and the edge stage's folded bias -> LDS ---------
    const int Ee = (int)a.prep[WH_EE], E0 = (int)a.prep[WH_E0], E1 = (int)a.prep[WH_E1], e_min = (int)a.prep[WH_EMIN];
    const bool w_bad = a.prep[WH_BAD] != 0;
    const unsigned acts = a.prep[WH_ACT];
    const float *tab0 = reinterpret_cast<const float *>(a.prep + W_HDR + SH::F_ALL * 256);
    {
        const rr_u4 *fr = reinterpret_cast<const rr_u4 *>(a.prep + W_HDR);
        rr_u4 *dst = reinterpret_cast<rr_u4 *>(smem + GL_WZ);
        for (int i = tid; i < 12 * 64; i += 256) {
            const int f = i >> 6;
            dst[i] = fr[(f < 8 ? SH::F_WE + f : SH::F_WB + (f - 8)) * 64 + (i & 63)];
        }
        if (tid < 128) {
            const int h = tid >> 6, fb = (tid >> 4) & 3, g = (tid >> 2) & 3, j = tid & 3;
            reinterpret_cast<float *>(smem + GL_C0)[tid] = tab0[32 * fb + 8 * g + 4 * h + j] * rr_pow2(127 - Ee);
        }
    }
    const int lo_e = (acts & 1) ? 0 : INT_MIN, lo_0 = (acts & 2) ? 0 : INT_MIN;
    const float lo_1 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint((acts & 4) ? 0.f : -INFINITY)));

    // ---- the weight stream: step k's fragments -> ring slot k % 3, this wave's share ------------------------------------------------------
    // (address = a wave-uniform base in scalar registers + the lane's 16 bytes as a 32-bit offset: one address register for every load)
    const unsigned char *const gfrag = reinterpret_cast<const unsigned char *>(a.prep + W_HDR);
    unsigned lane16 = 16u * (unsigned)lane0;
    auto dma1 = [&](int src_frag, int slot, int dst_frag) {
        __builtin_amdgcn_global_load_lds((gptr_t)(gfrag + (size_t)src_frag * 1024 + lane16), (lptr_t)(smem + GL_RING + slot * G_SLOT + dst_frag * 1024), 16, 0, 0);
    };
    auto dma = [&](int k) {
        const int slot = k % 3;
        if (k < 8) {                                   // chunk k of x against [Wj | Wi | W0x]
            const int runs[3] = {SH::F_WE + 8 * (1 + W_NXC + k), SH::F_WE + 8 * (1 + k), SH::F_W0 + 8 * (SH::NKS + k)};
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int j = 0; j < 2; ++j) dma1(runs[r] + 2 * wave + j, slot, 8 * r + 2 * wave + j);
        } else if (k < 12) {                           // chunks 2 kk, 2 kk + 1 of the S part of node stage 0
            const int kk = k - 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) dma1(SH::F_W0 + 16 * kk + 4 * wave + j, slot, 4 * wave + j);
        } else {                                       // chunks 3 kk .. of node stage 1 (3, 3, 2)
            const int kk = k - 12;
            const int per = kk < 2 ? 6 : 4;
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (j < per) dma1(SH::F_W1 + 24 * kk + per * wave + j, slot, per * wave + j);
        }
    };
    dma(0);
    dma(1);
    __syncthreads();                                   // (drains the two steps' loads as well)

    // ---- this workgroup's graphs ---------------------------------------------------------------------------------------------------------
    int g_cur = (int)((long long)a.n_graphs * blockIdx.x / a.n_wg);
    const int g_end = (int)((long long)a.n_graphs * (blockIdx.x + 1) / a.n_wg);
    auto window = [&](int g0, int lane) {
        int idx = g0 + lane;
        idx = idx < g_end ? idx : g_end;
        return (int)a.node_ptr[idx];
    };
    int win = window(g_cur, lane0);
    const unsigned t_start = clk();

    while (g_cur < g_end) {
        const unsigned q0 = clk();
        // (everything made from the lane / wave index is made per tile: hoisted out of the loop it is spilled, and a reload waits for
        //  every load in flight)
        int lane = lane0;
        asm volatile("" : "+v"(lane16), "+v"(wave_v), "+v"(lane));
        wave = __builtin_amdgcn_readfirstlane(wave_v);
        const int li = lane & 31, lh = lane >> 5;
        const unsigned ring_lane = lds0 + GL_RING + 16u * (unsigned)lane;
        auto frag = [&](int slot, int f) { return *reinterpret_cast<rr_ldsp>(ring_lane + (unsigned)(slot * G_SLOT + f * 1024)); };
        // ---- the tile: whole graphs g_cur .. g_cur + k - 1 with <= 128 nodes together -----------------------------------------------------
        const int n0 = __builtin_amdgcn_readfirstlane(win);
        int k_g = __popcll(__ballot(lane >= 1 && g_cur + lane <= g_end && win - n0 <= G_ROWS));
        k_g = k_g < 1 ? 1 : k_g;                       // (a graph above 128 nodes: refused by the host; never loops forever)
        const int n1 = __builtin_amdgcn_readlane(win, k_g);
        int nn = n1 - n0;
        nn = nn > G_ROWS ? G_ROWS : nn;
        g_cur += k_g;
        win = window(g_cur, lane);                     // (the next tile's window: in flight under this tile)
        const int r0 = n0 + 32 * wave;
        int nnw = nn - 32 * wave;
        nnw = nnw < 0 ? 0 : (nnw > 32 ? 32 : nnw);

        // ---- this lane's target: segment bounds, the first G_PRE in-edges' sources / per-edge rows -------------------------------------------
        const bool tvalid = li < nnw;
        int node = r0 + (tvalid ? li : 0);
        node = node < a.n_nodes ? node : a.n_nodes - 1;
        int pt = a.seg_ptr[node], pt1 = a.seg_ptr[node + 1];
        if (!tvalid) { pt = 0; pt1 = 0; }
        const int deg = pt1 - pt;
        int srcv[G_PRE], eiv[G_PRE];
#pragma unroll
        for (int it = 0; it < G_PRE; ++it) {
            const int e = it < deg ? pt + it : 0;
            srcv[it] = a.src[e];
            eiv[it] = a.eidx[e];
        }
        // ---- x rows of this wave's 32 nodes as two fp16 planes in operand layout (lane (t, h): columns 16 c + 8 h .. + 8 of row t) -----------
        rr_u4 xh[W_NXC], xl[W_NXC];
        bool bad_x;
        float facx;                                    // accumulator of a product of x (units rs_x 2^E) -> true value: x 2^-E further
        int e_x;
        {
            int xrow = r0 + (tvalid ? li : (nnw > 0 ? nnw - 1 : 0));
            xrow = xrow < a.n_nodes ? xrow : a.n_nodes - 1;
            const rr_f4 *xp = reinterpret_cast<const rr_f4 *>(a.x + (size_t)xrow * W_DX + 8 * lh);
            rr_f4 v[W_NXC][2];
#pragma unroll
            for (int c = 0; c < W_NXC; ++c) { v[c][0] = xp[4 * c]; v[c][1] = xp[4 * c + 1]; }
            unsigned m = 0;
#pragma unroll
            for (int c = 0; c < W_NXC; ++c)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    m = max(max(m, __float_as_uint(v[c][j].x) & 0x7fffffffu), __float_as_uint(v[c][j].y) & 0x7fffffffu);
                    m = max(max(m, __float_as_uint(v[c][j].z) & 0x7fffffffu), __float_as_uint(v[c][j].w) & 0x7fffffffu);
                }
            m = rr_xhalf_max(m);
            int e = (int)(m >> 23);
            bad_x = e >= 255;
            e = e < 15 ? 15 : (e > 254 ? 254 : e);
            e_x = e;
            const float rs = __uint_as_float((unsigned)(268 - e) << 23);
#pragma unroll
            for (int c = 0; c < W_NXC; ++c) {
                unsigned h[4], l[4];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    rr_split2s(v[c][j].x, v[c][j].y, rs, h[2 * j], l[2 * j]);
                    rr_split2s(v[c][j].z, v[c][j].w, rs, h[2 * j + 1], l[2 * j + 1]);
                }
                xh[c] = rr_u4{h[0], h[1], h[2], h[3]};
                xl[c] = rr_u4{l[0], l[1], l[2], l[3]};
            }
        }
        const unsigned q1 = clk();

        // =====================================================================================================================================
        // steps 0 .. 7: chunk c of the x rows against [Wj | Wi | W0x]:  P_j^T, P_i^T, Hx^T  (feature x node accumulators)
        // =====================================================================================================================================
        f32x16 pj[WB], pi[WB], hx[WB];
#pragma unroll
        for (int fb = 0; fb < WB; ++fb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { pj[fb][r] = 0.f; pi[fb][r] = 0.f; hx[fb][r] = 0.f; }
        rr_f4 zr[G_PRE][2];
#pragma unroll
        for (int c = 0; c < W_NXC; ++c) {
            dma((c + 2) % G_NSTEP);
            const int slot = c % 3;
#pragma unroll
            for (int grp = 0; grp < 3; ++grp) {
                rr_u4 f[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) f[q] = frag(slot, 8 * grp + q);
                f32x16 (&acc)[WB] = grp == 0 ? pj : (grp == 1 ? pi : hx);
#pragma unroll
                for (int fb = 0; fb < WB; ++fb) RR_MFH(f[2 * fb], xl[c], acc[fb]);
#pragma unroll
                for (int fb = 0; fb < WB; ++fb) RR_MFH(f[2 * fb + 1], xh[c], acc[fb]);
#pragma unroll
                for (int fb = 0; fb < WB; ++fb) RR_MFH(f[2 * fb], xh[c], acc[fb]);
            }
            int extra = 0;
            if (c == 1) {
                // the per-edge columns of the first G_PRE in-edges of every target: behind this step's products, consumed after step 7
#pragma unroll
                for (int it = 0; it < G_PRE; ++it)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const WQuad &qa = a.zq[j], &qb = a.zq[2 + j];
                        const unsigned long long base = lh ? qb.base : qa.base;
                        const unsigned stride = lh ? qb.stride : qa.stride, role = lh ? qb.role : qa.role;
                        const unsigned row = role == 0u ? (unsigned)node : (role == 1u ? (unsigned)srcv[it] : (unsigned)eiv[it]);
                        zr[it][j] = *reinterpret_cast<const __attribute__((address_space(1))) rr_f4 *>(base + (unsigned long long)row * stride);
                    }
                extra = 2 * G_PRE;
            }
            if (c == W_NXC - 1) {
                // ---- P_j^T -> LDS rows (true units), the row's bad flag ----------------------------------------------------------------------
                const unsigned q2 = clk();
                facx = bad_x ? __uint_as_float(0x7fc00000u) : rr_pow2(e_x - 14 - Ee);
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
            // this wave's share of step c + 1 has landed (all but the newest loads: step c + 2's, and the gathers issued behind them)
            if (g_step_loads((c + 2) % G_NSTEP) + extra == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
            else if (g_step_loads((c + 2) % G_NSTEP) == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            lds_barrier();
        }
        const unsigned q3 = clk();

        // =====================================================================================================================================
        // edge phase: lane (t, h) walks the in-edges of target t; S^T in the accumulator layout (64 features of node t per lane)
        // =====================================================================================================================================
        dma(10);                                       // (step 8's loads of step 10: in flight under the edge phase)
        f32x16 S[WB];
        unsigned badt = 0;
        {
            // P_i in true units with the folded bias
            const unsigned c0p = lds0 + GL_C0 + 256u * (unsigned)lh;
#pragma unroll
            for (int fb = 0; fb < WB; ++fb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const rr_f4 c0 = *reinterpret_cast<ldsf4_t>(c0p + 64u * fb + 16u * g);
#pragma unroll
                    for (int j = 0; j < 4; ++j) pi[fb][4 * g + j] = fmaf(pi[fb][4 * g + j], facx, c0[j]);
                }
#pragma unroll
            for (int fb = 0; fb < WB; ++fb)
#pragma unroll
                for (int r = 0; r < 16; ++r) S[fb][r] = 0.f;
            const unsigned wzp = lds0 + GL_WZ + 16u * (unsigned)lane;
            auto edge_round = [&](bool act, int s_glob, rr_f4 z0, rr_f4 z1) {
                int sl = s_glob - n0;
                sl = sl < 0 ? 0 : (sl > G_ROWS - 1 ? G_ROWS - 1 : sl);
                const int bad_s = *reinterpret_cast<ldsi_t>(lds0 + GL_BAD + 4u * (unsigned)sl);
                f32x16 q[WB];
                float invz = 0.f;
                bool bad_z = false;
                if (a.has_z) {
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
                    rr_u4 wz[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) wz[i] = *reinterpret_cast<rr_ldsp>(wzp + 1024u * i);
#pragma unroll
                    for (int fb = 0; fb < WB; ++fb) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) q[fb][r] = 0.f;
                    }
#pragma unroll
                    for (int fb = 0; fb < WB; ++fb) RR_MFH(wz[2 * fb], zl, q[fb]);
#pragma unroll
                    for (int fb = 0; fb < WB; ++fb) RR_MFH(wz[2 * fb + 1], zh, q[fb]);
#pragma unroll
                    for (int fb = 0; fb < WB; ++fb) RR_MFH(wz[2 * fb], zh, q[fb]);
                }
                if (act) {
                    const unsigned rowp = lds0 + GL_PJ + (unsigned)sl * G_PITCH + 256u * (unsigned)lh;
#pragma unroll
                    for (int fb = 0; fb < WB; ++fb)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const rr_f4 pv = *reinterpret_cast<ldsf4_t>(rowp + 64u * fb + 16u * g);
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int r = 4 * g + j;
                                float v = a.has_z ? fmaf(q[fb][r], invz, pi[fb][r]) : pi[fb][r];
                                v += pv[j];
                                S[fb][r] += rr_imax(v, lo_e);
                            }
                        }
                    if (bad_s || bad_z) badt = 1;
                }
            };
            const int maxdeg = [&]() {
                int d = deg;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) d = max(d, __shfl_xor(d, o, 64));
                return __builtin_amdgcn_readfirstlane(d);
            }();
#pragma unroll
            for (int it = 0; it < G_PRE; ++it)
                if (it < maxdeg) edge_round(it < deg, srcv[it], zr[it][0], zr[it][1]);
            for (int it = G_PRE; it < maxdeg; ++it) {          // hubs: on demand
                const bool act = it < deg;
                const int e = act ? pt + it : 0;
                const int sg = a.src[e], eg = a.eidx[e];
                rr_f4 z[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const WQuad &qa = a.zq[j], &qb = a.zq[2 + j];
                    const unsigned long long base = lh ? qb.base : qa.base;
                    const unsigned stride = lh ? qb.stride : qa.stride, role = lh ? qb.role : qa.role;
                    const unsigned row = role == 0u ? (unsigned)node : (role == 1u ? (unsigned)sg : (unsigned)eg);
                    z[j] = *reinterpret_cast<const __attribute__((address_space(1))) rr_f4 *>(base + (unsigned long long)row * stride);
                }
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
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int k = 8 + kk, slot = k % 3;
            if (kk > 0) dma((k + 2) % G_NSTEP);
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int c = 2 * kk + cc, fb1 = c >> 1, cc1 = c & 1;
                unsigned ph[4], pl[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) rr_split2s(S[fb1][8 * cc1 + 2 * q], S[fb1][8 * cc1 + 2 * q + 1], rs0, ph[q], pl[q]);
                const rr_u4 bh = rr_u4{ph[0], ph[1], ph[2], ph[3]}, bl = rr_u4{pl[0], pl[1], pl[2], pl[3]};
                rr_u4 f[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) f[q] = frag(slot, 8 * cc + q);
#pragma unroll
                for (int fbo = 0; fbo < WB; ++fbo) RR_MFH(f[2 * fbo], bl, hacc[fbo]);
#pragma unroll
                for (int fbo = 0; fbo < WB; ++fbo) RR_MFH(f[2 * fbo + 1], bh, hacc[fbo]);
#pragma unroll
                for (int fbo = 0; fbo < WB; ++fbo) RR_MFH(f[2 * fbo], bh, hacc[fbo]);
            }
            if (g_step_loads((k + 2) % G_NSTEP) == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            lds_barrier();
        }
        const unsigned q5 = clk();
        // ---- between the stages: h = act_0(Hs 2^(e_t - 141 - E0) + Hx facx 2^(Ee - E0)) in true units, its row scale -------------------------
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
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            const int k = 12 + kk, slot = k % 3;
            dma((k + 2) % G_NSTEP);
            const int nc = kk < 2 ? 3 : 2;
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                if (cc >= nc) continue;
                const int c = 3 * kk + cc, fb1 = c >> 1, cc1 = c & 1;
                unsigned ph[4], pl[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) rr_split2s(hacc[fb1][8 * cc1 + 2 * q], hacc[fb1][8 * cc1 + 2 * q + 1], f2, ph[q], pl[q]);
                const rr_u4 bh = rr_u4{ph[0], ph[1], ph[2], ph[3]}, bl = rr_u4{pl[0], pl[1], pl[2], pl[3]};
                rr_u4 f[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) f[q] = frag(slot, 8 * cc + q);
#pragma unroll
                for (int fb = 0; fb < WB; ++fb) RR_MFH(bl, f[2 * fb], oacc[fb]);
#pragma unroll
                for (int fb = 0; fb < WB; ++fb) RR_MFH(bh, f[2 * fb + 1], oacc[fb]);
#pragma unroll
                for (int fb = 0; fb < WB; ++fb) RR_MFH(bh, f[2 * fb], oacc[fb]);
            }
            if (g_step_loads((k + 2) % G_NSTEP) == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            lds_barrier();
        }
        const unsigned q7 = clk();
        // ---- rows leave as 128-byte row segments (layer_w.hip) ----------------------------------------------------------------------------
        {
            float cb1[WB];
#pragma unroll
            for (int fb = 0; fb < WB; ++fb) cb1[fb] = tab0[3 * 32 * WB + 32 * fb + li];
            float invr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) invr[r] = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * rr_crow(r, lh), __float_as_int(inv2)));
            int voff_lane = (4 * lh * 32 * WB + li) * 4;
            asm volatile("" : "+v"(voff_lane));         // (one offset register + a scalar constant per store: 64 hoisted offsets spill otherwise)
            const int r0c = r0 < a.n_nodes ? r0 : 0;
            const __amdgpu_buffer_rsrc_t orow = __builtin_amdgcn_make_buffer_rsrc(a.out + (int64_t)r0c * (32 * WB), 0, nnw * (32 * WB * 4), 0x00020000);
#pragma unroll
            for (int fb = 0; fb < WB; ++fb) {
                const float cb = cb1[fb];
                auto put = [&](auto nanrows) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float y = fmaxf(fmaf(oacc[fb][r], invr[r], cb), lo_1);
                        if (decltype(nanrows)::value) y = invr[r] != invr[r] ? invr[r] : y;     // (the max drops a NaN)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), orow, voff_lane, ((r & 3) + 8 * (r >> 2)) * (32 * WB * 4) + 32 * fb * 4, 0);
                    }
                };
                if (anybad) put(std::true_type{}); else put(std::false_type{});
            }
        }
        if (PROF) {
            const unsigned q8 = clk();
            pc[0] += q1 - q0; pc[2] += q3 - q1; pc[4] += q4 - q3; pc[5] += q5 - q4; pc[6] += q6 - q5; pc[7] += q7 - q6; pc[8] += q8 - q7; pc[9] += 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the ring's last two prefetched steps land before the LDS is given back)
    if (PROF && prof && lane0 == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2) && wave == 0) {
        unsigned long long *o = prof + (blockIdx.x == 0 ? 0 : 16);
        for (int q = 0; q < 10; ++q) o[q] = pc[q];
        o[10] = clk() - t_start;
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
    a.has_z = nq > 0 && n_edges > 0;
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
        hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(&layer_fused_kernel_g<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e0 == hipSuccess) e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(&layer_fused_kernel_g<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e0 != hipSuccess) return set_error(GSN_E_HIP, "hipFuncSetAttribute(layer_fused_kernel_g): %s", hipGetErrorString(e0));
        attr_set.mark(attr_dev);
    }
    if (getenv("GSN_CHAIN_TRACE")) fprintf(stderr, "gsn chain: layer_fused_kernel_g nodes %d edges %d graphs %d grid %lld\n", a.n_nodes, a.n_edges, a.n_graphs, (long long)gx);
    if (prof_on) {
        unsigned long long *prof = nullptr;
        (void)hipMalloc(&prof, 32 * 8); (void)hipMemsetAsync(prof, 0, 32 * 8, st);
        hipLaunchKernelGGL((layer_fused_kernel_g<true>), dim3((unsigned)gx), dim3(256), GL_TOTAL, st, a, prof);
        unsigned long long h[32];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost);
        (void)hipFree(prof);
        static int shown = 0;
        if (shown++ % 8 == 7)
            for (int w = 0; w < 2; ++w) {
                const unsigned long long *o = h + 16 * w;
                const double nt = o[9] ? (double)o[9] : 1.0;
                fprintf(stderr, "gprof wg %s: tiles %llu total %llu cycles | per tile: top+x %.0f products %.0f (P_j write %.0f) edges %.0f stage0 %.0f between %.0f stage1 %.0f stores %.0f\n",
                        w ? "mid" : "0", o[9], o[10], o[0] / nt, o[2] / nt, o[3] / nt, o[4] / nt, o[5] / nt, o[6] / nt, o[7] / nt, o[8] / nt);
            }
    } else {
        hipLaunchKernelGGL((layer_fused_kernel_g<false>), dim3((unsigned)gx), dim3(256), GL_TOTAL, st, a, (unsigned long long *)nullptr);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "layer_fused_kernel_g: %s", hipGetErrorString(e));
    return GSN_OK;
}

}  // namespace gsn
