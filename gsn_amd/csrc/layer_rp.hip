// The one-launch `general` layer on EXACT fp16 ROW PACKS (GSN_edge_sparse.py:82-170 / GSN_sparse.py:93-176 with id_scope='local',
// MPNN_edge_sparse.py:110-151; layer 0 of every reference model: its inputs are one-hot / small-integer encodings,
// utils_graph_learning.py:78-88, :170-187).  Same operation, same data flow and the same fp16 plane arithmetic as layer_rr.hip:
//
//     r_e  = act_e( bn_e( cat(x_i, x_j, ids.., e) W1^T + b1 ) )              per edge
//     S_v  = sum_{e -> v} r_e                                                per node      (torch.sparse.sum, :136-139)
//     h_v  = act_0( bn_0( [x_v | S_v | deg_v] W0'^T + b0 ) ),   out_v = act_1( bn_1( h_v W1'^T + b1' ) )
//
// What layer_rr.hip spends on its INPUT rows is what this kernel does not do.  There a wave gathers fp32 rows (ten 16-byte loads per
// lane and block through a table of base / stride / index role), converts them to fp16, tests whether the conversion was exact
// (~140 vector instructions per block), and keeps a second code path for rows that are not.  Here the producers of the rows -- the
// count-encode kernel, the one-hot kernels, gsn_pack16_rows_hip -- have already written them as fp16, exact by construction, in the
// operand layout:
//
//   node pack  fp16 [n_nodes][32]   columns 0 .. d_x-1 = x, d_x .. 30 = 0, column 31 = 1.0
//   edge pack  fp16 [n_rows][16]    the edge-level blocks (identifiers, edge features) concatenated, zero padded
//
// so the concatenated edge row cat(x_i, x_j, edge-level) is 80 fp16 values = FIVE 16-byte loads per lane that ARE the operand
// fragments (lane (e, h) of chunk c: bytes 32 c + 16 h of the row), addressed through two buffer resources with the row index shifted
// into the offset (3 vector instructions per block instead of ~100); the constant-1 column 31 of x_i carries the folded bias of the
// edge stage.  Node stage 0 reads its x rows from the same pack (two loads, scaled inside a v_fma_mixlo/hi_f16 pair: the exact rows have
// no low plane, one product set less), the in-degree rides k-slots d_x and d_x + 1 as a high / low pair.  Everything that is in
// registers afterwards is what layer_rr.hip has there too.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "chain_common.h"
#include "layer_rr.h"
#include "layer_rr_inl.h"
#include "layer_rr_core.h"

namespace gsn {

struct RpArgs {
    int n_nodes, n_edges;
    const int32_t *seg_ptr;
    const int32_t *role_idx[3];        // per sorted edge row: the node row of x_i, the node row of x_j, the row of the edge pack
    const uint16_t *node16, *edge16;
    unsigned node_bytes, edge_bytes;   // extents of the two packs (range-checked buffer loads)
    int edge_shift;                    // log2 of the edge pack's row bytes (5; 6 when there is no edge pack and edge16 aliases node16)
    int d_x;
    float *out;
    const unsigned *prep;
    int n_ranges;
    int n_split;                       // nodes of the older half of the waves (0: equal ranges)
};

struct RpIdx { int r[3]; int pt, pt1; };

__device__ __forceinline__ void rp_idx_load(const RpArgs &a, const RrDesc &d, int li, RpIdx &ix) {
    const int ne = d.ne();
    if (d.valid() && ne > 0) {
        const int e = d.e0 + (li < ne ? li : ne - 1);
#pragma unroll
        for (int q = 0; q < 3; ++q) ix.r[q] = a.role_idx[q][e];
    }
    if (d.valid()) {
        int t = d.m0 + li;
        t = t < a.n_nodes ? t : a.n_nodes - 1;          // (lanes past nn are masked when used)
        ix.pt = a.seg_ptr[t];
        ix.pt1 = a.seg_ptr[t + 1];
    }
}

#define RP_LOAD(RS, VOFF, IMM) __builtin_bit_cast(rr_u4, __builtin_amdgcn_raw_buffer_load_b128(RS, (int)((VOFF) + (IMM)), 0, 0))

// fp16 half of a packed pair times an fp32 scale -> fp16, one instruction per value (the product is formed in fp32: no intermediate overflow)
__device__ __forceinline__ unsigned rp_scale_pair(unsigned pair, float s) {
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pair), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(pair), "v"(s));
    return r;
}
// (a s) as an fp16 high part in bits 15:0 and its fp16 residual in bits 31:16
__device__ __forceinline__ unsigned rp_hi_lo_word(float a, float s) {
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%0 op_sel_hi:[0,0,1]" : "+v"(r) : "v"(a), "v"(s));
    return r;
}
__device__ __forceinline__ unsigned rp_pk_max_u16(unsigned a, unsigned b) { unsigned r; asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

#ifndef RP_DEFER
#define RP_DEFER 0         // 1: the second feature pair's epilogue of a block waits for the next block's first products (not at a tile's end): +40 live registers, spills
#endif
#ifndef RP_GATHER_AT
#define RP_GATHER_AT 3        // chunk step of node stage 1's first pair at which the next tile's first gathers are issued
#endif
#ifndef RP_STORE_AUX
#define RP_STORE_AUX 2       // cache policy bits of the output stores (2: nt; measured 0.5495 -> 0.533 ms, with RP_PD 8: 0.518)
#endif
#ifndef RP_PD
#define RP_PD 8              // fragments of the low-plane weight stream in flight (layer_rr.hip: 4)
#endif
#ifndef RP_STAGGER
#define RP_STAGGER 0          // start delay of the second wave of each SIMD, units of 64 x 64 = 4096 cycles
#endif
#ifndef RP_S1T
#define RP_S1T 0             // 1: node stage 1 transposed (a lane holds four consecutive features of its own target: 16-byte stores, 32-byte pieces of a row)
#endif
#ifndef RP_DRAIN
#define RP_DRAIN 0           // 1: everything in flight is waited for in front of a tile's first output store
#endif
#ifndef RP_S1E
#define RP_S1E 1           // node stage 1: the first pair's output rows leave under the second pair's products
#endif

template <int WB, int NKX, bool PROF>
__global__ __launch_bounds__(64 * RR_NW) __attribute__((amdgpu_waves_per_eu(RR_NW / 4, RR_NW / 4))) void layer_fused_kernel_rp(RpArgs a, unsigned long long *prof) {
    auto clk = [&]() -> unsigned {
        if (!PROF) return 0u;
        __builtin_amdgcn_sched_barrier(0);
        const unsigned v = (unsigned)__builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
        return v;
    };
    unsigned pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // block top + issue, edge, exposed edge epilogue, stage 0, between the stages, stage 1, blocks, tiles
    using SH = RrShape<WB, NKX>;
    constexpr int NKS = SH::NKS, NK0 = SH::NK0;
    static_assert(WB == 4 && NKX == 2, "two pairs of feature blocks; the node pack is two chunks wide");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane0 = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li0 = lane0 & 31, lh0 = lane0 >> 5;
    unsigned ldsb[3];
    ldsb[0] = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem + 16u * (unsigned)lane0;
    ldsb[1] = ldsb[0] + 0x10000u; ldsb[2] = ldsb[0] + 0x20000u;
    asm volatile("" : "+v"(ldsb[1]), "+v"(ldsb[2]));          // (opaque: see rr_lds_frag)

    // ---- prologue: weights -> LDS ---------------------------------------------------------------------------------------------
    {
        const rr_u4 *src = reinterpret_cast<const rr_u4 *>(a.prep + RR_HDR);
        rr_u4 *dst = reinterpret_cast<rr_u4 *>(smem);
        for (int i = tid; i < SH::F_LDS * 64; i += 64 * RR_NW) dst[i] = src[i];
        const float *tsrc = reinterpret_cast<const float *>(a.prep + RR_HDR + SH::F_ALL * 256);
        float *tdst = reinterpret_cast<float *>(smem + SH::F_LDS * 1024);
        for (int i = tid; i < SH::TAB_WORDS; i += 64 * RR_NW) tdst[i] = tsrc[i];
    }
    __syncthreads();
    // (wave-uniform header words: scalar registers -- read as they stand they are ONE vector load of four, kept in vector registers for the
    //  whole kernel and spilled)
    auto hdr = [&](int i) { return (int)__builtin_amdgcn_readfirstlane(a.prep[i]); };
    const int Ee = hdr(RRH_EE), E0 = hdr(RRH_E0), E1 = hdr(RRH_E1), e_min = hdr(RRH_EMIN);
    const bool w_bad = hdr(RRH_BAD) != 0;
    if (a.prep[RRH_PACK] != 1u) __builtin_trap();              // (prepared for layer_rr.hip: another k-slot order)
    const unsigned acts = (unsigned)hdr(RRH_ACT);
    // activations as an INTEGER max on the bit patterns (relu: against 0, identity: against INT_MIN): visible to the compiler's hazard
    // recogniser and scheduler, unlike an asm v_max_f32; a NaN of positive sign passes, the rows that must come out NaN are forced below
    const int lo_e = __builtin_amdgcn_readfirstlane((acts & 1) ? 0 : (int)0x80000000), lo_0 = __builtin_amdgcn_readfirstlane((acts & 2) ? 0 : (int)0x80000000);
    const int lo_1 = __builtin_amdgcn_readfirstlane((acts & 4) ? 0 : (int)0x80000000);
    const __amdgpu_buffer_rsrc_t wstream = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(a.prep) + RR_HDR + SH::F_W0SL * 256, 0, NKS * WB * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_n = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(a.node16), 0, (int)a.node_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_e = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(a.edge16), 0, (int)a.edge_bytes, 0x00020000);
    const int esh = a.edge_shift;
    // where the in-degree goes: k-slots d_x (high part) and d_x + 1 (residual) of the node pack's row = one word of one chunk, one lane half
    const int deg_c = a.d_x >> 4, deg_h = (a.d_x >> 3) & 1, deg_w = (a.d_x >> 1) & 3;

    // ---- this wave's node range -----------------------------------------------------------------------------------------------
    const int range = wave * (int)gridDim.x + (int)blockIdx.x;
    if (range >= a.n_ranges) return;
    const unsigned long long wave_t0 = PROF ? __builtin_amdgcn_s_memrealtime() : 0ull;
#if RP_STAGGER > 0
    // the second wave of every SIMD starts a fraction of a tile late: both waves run the same tile loop on tiles of (nearly) the same
    // shape, so a phase offset set here persists -- one wave's matrix segments beside the other's conversions instead of beside its
    // matrix segments (MI355X_MICROARCH.md, two waves per SIMD: the matrix pipe is per SIMD)
    if (wave >= RR_NW / 2) {
#pragma unroll 1
        for (int i = 0; i < RP_STAGGER; ++i) __builtin_amdgcn_s_sleep(64);
    }
#endif
    RrIter it;
    it.seg = a.seg_ptr; it.n_nodes = a.n_nodes;
    // The two waves of a SIMD are not served alike: the older one (waves 0 .. NW/2 - 1 of the workgroup) wins the issue arbitration and
    // finishes an equal share ~28 % earlier (profiles/r05_layer_rp_issue.txt: first wave done at 308 us, last at 430) -- and the younger
    // one, then alone on its SIMD, runs at the pair's combined rate: the SIMD's instruction issue is what is full, so cutting the node
    // ranges unevenly (the first a.n_split nodes to the older half of the waves; GSN_RP_OLD_SHARE, default 0.6) only trims the tail:
    // 0.519-0.529 -> 0.514 ms on one box, both waves ending within 15 % of each other instead of 28 %.
    {
        const int half_n = a.n_ranges >> 1;
        if (a.n_split > 0 && half_n > 0 && (a.n_ranges & 1) == 0 && (int)gridDim.x * (RR_NW / 2) == half_n) {
            const int young = wave >= RR_NW / 2 ? 1 : 0;
            const int idx = range - young * half_n;
            const int64_t base = young ? a.n_split : 0, cnt = young ? a.n_nodes - a.n_split : a.n_split;
            it.m_next = (int)(base + cnt * idx / half_n);
            it.m_end = (int)(base + cnt * (idx + 1) / half_n);
        } else {
            it.m_next = (int)((int64_t)a.n_nodes * range / a.n_ranges);
            it.m_end = (int)((int64_t)a.n_nodes * (range + 1) / a.n_ranges);
        }
    }
    it.m0 = 0; it.nn = 0; it.eb = 0; it.ee = 0; it.ec = 0; it.pending = 0; it.win = 0;
    rr_iter_load(it, lane0);
    RrDesc cur = rr_iter_next(it, lane0);
    RpIdx ixc, ixn;
#pragma unroll
    for (int q = 0; q < 3; ++q) { ixc.r[q] = 0; ixn.r[q] = 0; }
    ixc.pt = ixc.pt1 = ixn.pt = ixn.pt1 = 0;
    rp_idx_load(a, cur, li0, ixc);
    RrDesc nxt = rr_iter_next(it, lane0);
    rp_idx_load(a, nxt, li0, ixn);
    rr_u4 g[RR_NKE];                                   // the gathered rows of the CURRENT block (in flight at the top of the loop)
    auto gather = [&](const int (&r)[3], int lh) {
        const unsigned vt = ((unsigned)r[0] << 6) | ((unsigned)lh << 4), vs = ((unsigned)r[1] << 6) | ((unsigned)lh << 4);
        const unsigned ve = ((unsigned)r[2] << esh) | ((unsigned)lh << 4);
        g[0] = RP_LOAD(rs_n, vt, 0); g[1] = RP_LOAD(rs_n, vt, 32);
        g[2] = RP_LOAD(rs_n, vs, 0); g[3] = RP_LOAD(rs_n, vs, 32);
        g[4] = RP_LOAD(rs_e, ve, 0);
    };
    gather(ixc.r, lh0);
    int pt = ixc.pt, pt1 = ixc.pt1;

    const unsigned t_start = clk();
    while (cur.valid()) {
        // =============================================================================================================================
        // one TILE: its blocks (edge stage + per-node sums), then the two node stages
        // =============================================================================================================================
        asm volatile("" : "+v"(ldsb[0]), "+v"(ldsb[1]), "+v"(ldsb[2]));
        int li = li0, lh = lh0;
        asm volatile("" : "+v"(li), "+v"(lh));
        const int lane = li + 32 * lh;
        const float *tab = reinterpret_cast<const float *>(rr_lds_generic(ldsb[0] - 16u * (unsigned)lane + SH::F_LDS * 1024));
        const int m0 = cur.m0, nn = cur.nn();
        // the x rows of the tile's nodes in the operand layout of node stage 0 (lane (t, h): bytes 32 c + 16 h of row t): requested in the
        // tile's last block, in front of its last epilogue (rows this tile's edge blocks have just gathered: cache hits)
        rr_u4 X[NKX];                                  // (requested with EVERY block -- an assignment under `last` makes a loop-carried value of it)
        auto load_x = [&]() {
            const unsigned vx = ((unsigned)(m0 + (li < nn ? li : nn - 1)) << 6) | ((unsigned)lh << 4);
#pragma unroll
            for (int c = 0; c < NKX; ++c) X[c] = RP_LOAD(rs_n, vx, 32 * c);
        };
        f32x16 sacc[WB];                               // S^T tiles: row = feature in block, column = target
#pragma unroll
        for (int fb = 0; fb < WB; ++fb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[fb][r] = 0.f;
        // the second feature pair of a block that is not the tile's last waits here: its activation / plane split / incidence products run
        // under the first products of the NEXT block
        f32x16 pa0, pa1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { pa0[r] = 0.f; pa1[r] = 0.f; }
        rr_u4 Mp[2] = {rr_u4{0, 0, 0, 0}, rr_u4{0, 0, 0, 0}};
        bool deferred = false;
        int tpt = 0, tpt1 = 0;
        int rnext[3] = {0, 0, 0};
        bool last;
        do {
            const unsigned t0 = clk();
            const int ne = cur.ne();
            if (li >= nn) { pt = 0; pt1 = 0; }
            tpt = pt; tpt1 = pt1;
            // ---- the gathered rows ARE the operand fragments; the next block's gathers take their place in flight ----------------------------
            rr_u4 Ah[RR_NKE];
#pragma unroll
            for (int c = 0; c < RR_NKE; ++c) Ah[c] = g[c];
            last = cur.last() != 0;
            // the next block's gathers: inside a tile right here (they fly under this block's edge stage); across a tile boundary they
            // would hold 20 registers through both node stages -- issued in node stage 1, once the hidden rows' registers are free
            if (!last) gather(ixn.r, lh);
#pragma unroll
            for (int q = 0; q < 3; ++q) rnext[q] = ixn.r[q];
            const int npt = ixn.pt, npt1 = ixn.pt1;
            const RrDesc nn2 = rr_iter_next(it, lane);
            rp_idx_load(a, nn2, li, ixn);
            const unsigned t1 = clk();
            unsigned t2 = t1;
            // =========================================================================================================================
            // edge stage + per-node sums of this block: two plane products (the rows are exact), Y'' = se Y bounded by 2^15, two fp16 planes
            // into the incidence product.  Feature blocks in pairs = two accumulator chains issued alternately; weight fragments one step
            // ahead.  A wave issues in order and only vector work BETWEEN two of its own MFMAs runs under them: the activation + plane
            // split of a pair (80 vector instructions) and its 8 incidence products are spread over the 20 products of the NEXT pair.
            // =========================================================================================================================
            if (ne > 0) {
                const unsigned bm = rr_edge_mask(pt, pt1, cur.e0);
                rr_u4 M[2];
                rr_incidence(bm, lh, M);
                auto frag4 = [&](int fp, int c, rr_u4 &h0, rr_u4 &l0, rr_u4 &h1, rr_u4 &l1) {
                    h0 = rr_lds_frag(ldsb, SH::F_WE + 2 * (fp * RR_NKE + c)); l0 = rr_lds_frag(ldsb, SH::F_WE + 2 * (fp * RR_NKE + c) + 1);
                    h1 = rr_lds_frag(ldsb, SH::F_WE + 2 * ((fp + 1) * RR_NKE + c)); l1 = rr_lds_frag(ldsb, SH::F_WE + 2 * ((fp + 1) * RR_NKE + c) + 1);
                };
                // half an accumulator tile's epilogue: rows 8 h .. + 8 of the register index -> four words of each plane
                auto ep_half = [&](const f32x16 &acc, int h, unsigned (&yh)[8], unsigned (&yl)[8]) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int r = 8 * h + 4 * q;
                        rr_split4(rr_imax(acc[r], lo_e), rr_imax(acc[r + 1], lo_e), rr_imax(acc[r + 2], lo_e), rr_imax(acc[r + 3], lo_e),
                                  yh[4 * h + 2 * q], yl[4 * h + 2 * q], yh[4 * h + 2 * q + 1], yl[4 * h + 2 * q + 1]);
                    }
                };
                auto incid = [&](f32x16 &sv, const unsigned (&yh)[8], const unsigned (&yl)[8], const rr_u4 (&MM)[2], int h) {
                    const rr_u4 pl = rr_u4{yl[4 * h], yl[4 * h + 1], yl[4 * h + 2], yl[4 * h + 3]};
                    const rr_u4 ph = rr_u4{yh[4 * h], yh[4 * h + 1], yh[4 * h + 2], yh[4 * h + 3]};
                    RR_MFH(pl, MM[h], sv);
                    RR_MFH(ph, MM[h], sv);
                };
                // step c (0 .. 4) of a pair's epilogue: 0: tile 0 rows 0-7 | 1: tile 0 rows 8-15 + its first incidence pair | 2: tile 1 rows 0-7 +
                // tile 0's second pair | 3: tile 1 rows 8-15 + tile 1's first pair | 4: tile 1's second pair
                unsigned yh0[8], yl0[8], yh1[8], yl1[8];
                auto ep_step = [&](int c, const f32x16 &A0, const f32x16 &A1, f32x16 &S0, f32x16 &S1, const rr_u4 (&MM)[2]) {
                    if (c == 0) ep_half(A0, 0, yh0, yl0);
                    if (c == 1) { ep_half(A0, 1, yh0, yl0); incid(S0, yh0, yl0, MM, 0); }
                    if (c == 2) { ep_half(A1, 0, yh1, yl1); incid(S0, yh0, yl0, MM, 1); }
                    if (c == 3) { ep_half(A1, 1, yh1, yl1); incid(S1, yh1, yl1, MM, 0); }
                    if (c == 4) incid(S1, yh1, yl1, MM, 1);
                };
                auto products = [&](int fp, f32x16 &acc0, f32x16 &acc1, auto &&under) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
                    rr_u4 f0h, f0l, f1h, f1l;
                    frag4(fp, 0, f0h, f0l, f1h, f1l);
#pragma unroll
                    for (int c = 0; c < RR_NKE; ++c) {
                        rr_u4 n0h = f0h, n0l = f0l, n1h = f1h, n1l = f1l;
                        if (c + 1 < RR_NKE) frag4(fp, c + 1, n0h, n0l, n1h, n1l);
                        RR_MFH(Ah[c], f0l, acc0);
                        RR_MFH(Ah[c], f1l, acc1);
                        under(c);
                        RR_MFH(Ah[c], f0h, acc0);
                        RR_MFH(Ah[c], f1h, acc1);
                        RR_SB();
                        f0h = n0h; f0l = n0l; f1h = n1h; f1l = n1l;
                    }
                };
                f32x16 qa0, qa1, ra0, ra1;
                if (deferred) products(0, qa0, qa1, [&](int c) { ep_step(c, pa0, pa1, sacc[2], sacc[3], Mp); });
                else products(0, qa0, qa1, [&](int) {});
                products(2, ra0, ra1, [&](int c) { ep_step(c, qa0, qa1, sacc[0], sacc[1], M); });
                t2 = clk();
                if (!last && RP_DEFER) {
                    pa0 = ra0; pa1 = ra1; Mp[0] = M[0]; Mp[1] = M[1];
                    deferred = true;
                } else {
                    // the tile's last block: its second pair's epilogue behind its own products
                    load_x();
#pragma unroll
                    for (int c = 0; c < RR_NKE; ++c) ep_step(c, ra0, ra1, sacc[2], sacc[3], M);
                    RR_SB();
                    deferred = false;
                }
            }
            else load_x();
            const unsigned t3 = clk();
            if (PROF) { pc[0] += t1 - t0; pc[1] += t2 - t1; pc[2] += t3 - t2; pc[6] += 1; }
            cur = nxt; nxt = nn2;
            pt = npt; pt1 = npt1;
        } while (!last);
        // (the x rows exist HERE: left alone the compiler sinks their loads to their use, where nothing hides them)
#pragma unroll
        for (int c = 0; c < NKX; ++c) asm volatile("" : "+v"(X[c]));
        const unsigned t3 = clk();
        // =========================================================================================================================
        // node stage 0 (transposed): H^T = W0 [S | x | deg]^T
        // =========================================================================================================================
        f32x16 hacc[WB];
        // ---- row scale from max(|S|, |x|, deg) in true units (sacc = 2 se S) ------------------------------------------------
        float ms = 0.f;
#pragma unroll
        for (int fb = 0; fb < WB; ++fb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) ms = fmaxf(fmaxf(fabsf(sacc[fb][r]), fabsf(sacc[fb][r + 1])), ms);
        // the constant-1 column (31: chunk 1, upper lane half, last slot) is the edge stage's; then the largest |x| of the lane's 16 values
        X[NKX - 1][3] = lh ? X[NKX - 1][3] & 0xffffu : X[NKX - 1][3];
        unsigned m16 = 0;
#pragma unroll
        for (int c = 0; c < NKX; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) m16 = rp_pk_max_u16(m16, X[c][q] & 0x7fff7fffu);
        m16 = max(m16 & 0xffffu, m16 >> 16);
        const unsigned fld_x = m16 ? (m16 >> 10) + 112u : 0u;       // fp32 exponent field of an fp16 magnitude (a denormal: rounded up)
        const float degf = (float)(tpt1 - tpt);
        const unsigned msb = __float_as_uint(ms);
        int es = (int)(msb >> 23) - 1 - Ee;                       // exponent field of the largest |S| in true units
        es = msb == 0 ? 0 : es;
        unsigned fld = (unsigned)max(max(es, (int)max(fld_x, __float_as_uint(degf) >> 23)), 0);
        fld = rr_xhalf_max(fld);
        bool badrow = msb >= 0x7f800000u || m16 >= 0x7c00u || w_bad;
        badrow = rr_xhalf_or(badrow ? 1u : 0u) != 0;
        int e_t = (int)fld;                                        // exponent field the row scale of target li is made from
        e_t = e_t < e_min ? e_min : (e_t > 254 ? 254 : e_t);
        const float rs = __uint_as_float((unsigned)(268 - e_t) << 23);     // 2^(141 - e_t)
        float fs = rr_pow2(267 - e_t - Ee);                                // rs / (2 se): sacc -> scaled planes
        const float sc = rr_pow2(E0 + 268 - e_t);                          // s0 rs: the bias in accumulator units
        if (badrow) fs = __uint_as_float(0x7fc00000u);
        // ---- the [x | deg] fragments: the pack's values times the row scale (exact: a power of two), the degree as a high / low pair ----
        rr_u4 Xh[NKX];
#pragma unroll
        for (int c = 0; c < NKX; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) Xh[c][q] = rp_scale_pair(X[c][q], rs);
        {
            const unsigned dw = rp_hi_lo_word(degf, rs);
#pragma unroll
            for (int c = 0; c < NKX; ++c)
#pragma unroll
                for (int q = 0; q < 4; q += 2) Xh[c][q] = (c == deg_c && q == deg_w && lh == deg_h) ? dw : Xh[c][q];
        }
        // ---- accumulators start at the bias (row = hidden feature: per register; column = target: this lane's scale) -----
        {
            const rr_f2 sc2 = rr_f2{sc, sc};
#pragma unroll
            for (int fbo = 0; fbo < WB; ++fbo)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 cv = *reinterpret_cast<const float4 *>(tab + 32 * WB + 32 * fbo + 8 * i + 4 * lh);
                    const rr_f2 p0 = rr_f2{cv.x, cv.y} * sc2, p1 = rr_f2{cv.z, cv.w} * sc2;
                    hacc[fbo][4 * i] = p0[0]; hacc[fbo][4 * i + 1] = p0[1]; hacc[fbo][4 * i + 2] = p1[0]; hacc[fbo][4 * i + 3] = p1[1];
                }
        }
        // ---- [x | deg] part first: exact rows, two plane products ------------------------------------------------------------
        {
            auto fr = [&](int i, rr_u4 &h0, rr_u4 &l0, rr_u4 &h1, rr_u4 &l1) {          // step i = (cq, pair)
                const int cq = i / (WB / 2), fa = 2 * (i % (WB / 2));
                h0 = rr_lds_frag(ldsb, SH::F_W0H + fa * NK0 + NKS + cq); l0 = rr_lds_frag(ldsb, SH::F_W0XL + fa * NKX + cq);
                h1 = rr_lds_frag(ldsb, SH::F_W0H + (fa + 1) * NK0 + NKS + cq); l1 = rr_lds_frag(ldsb, SH::F_W0XL + (fa + 1) * NKX + cq);
            };
            rr_u4 a0h, a0l, a1h, a1l;
            fr(0, a0h, a0l, a1h, a1l);
#pragma unroll
            for (int i = 0; i < NKX * (WB / 2); ++i) {
                const int cq = i / (WB / 2), fa = 2 * (i % (WB / 2));
                rr_u4 n0h = a0h, n0l = a0l, n1h = a1h, n1l = a1l;
                if (i + 1 < NKX * (WB / 2)) fr(i + 1, n0h, n0l, n1h, n1l);
                RR_MFH(a0l, Xh[cq], hacc[fa]);
                RR_MFH(a1l, Xh[cq], hacc[fa + 1]);
                RR_MFH(a0h, Xh[cq], hacc[fa]);
                RR_MFH(a1h, Xh[cq], hacc[fa + 1]);
                RR_SB();
                a0h = n0h; a0l = n0l; a1h = n1h; a1l = n1l;
            }
        }
        // ---- S part, one PAIR of hidden feature blocks after the other (two accumulator chains issued alternately).  The S^T tiles become
        //      operand fragments 16 features at a time: pass 1 makes the planes of chunk c + 1 under the products of chunk c and keeps them
        //      (they take the registers of the sums they are made from); pass 2 has its operands ready, so the activation and the row maximum
        //      of pass 1's finished accumulators run under its products.  The weights' low planes arrive from L2 in the order of use, PD
        //      fragments ahead; the high planes one step ahead from LDS.
#ifdef RR_ABL_NOSTREAM
#define RP_STREAM(I) rr_u4{(unsigned)(I), (unsigned)lane, 0x3c003c00u, 0u}
#elif defined(RR_ABL_STREAM1)       // (ablation: every stream load hits the same two fragments -- L1 hits instead of the L2 path; results wrong)
#define RP_STREAM(I) __builtin_bit_cast(rr_u4, __builtin_amdgcn_raw_buffer_load_b128(wstream, 16 * lane, ((I) & 1) * 1024, 0))
#else
#define RP_STREAM(I) __builtin_bit_cast(rr_u4, __builtin_amdgcn_raw_buffer_load_b128(wstream, 16 * lane, (I) * 1024, 0))
#endif
        constexpr int PD = RP_PD, NSL = NKS * WB;
        static_assert(PD % 2 == 0, "the stream is consumed two fragments per step");
        float m2 = 0.f;
        {
            rr_u4 ql[PD];
#pragma unroll
            for (int i = 0; i < PD; ++i) ql[i] = RP_STREAM(i);
            unsigned ph[NKS][4], pl[NKS][4];
            auto ssplit = [&](int c) {
                const int fb = c >> 1, cc = c & 1;
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    rr_split4s(sacc[fb][8 * cc + 4 * q], sacc[fb][8 * cc + 4 * q + 1], sacc[fb][8 * cc + 4 * q + 2], sacc[fb][8 * cc + 4 * q + 3], fs,
                               ph[c][2 * q], pl[c][2 * q], ph[c][2 * q + 1], pl[c][2 * q + 1]);
            };
            ssplit(0);
#pragma unroll
            for (int pass = 0; pass < WB / 2; ++pass) {
                const int fa = 2 * pass;
                rr_u4 a0h = rr_lds_frag(ldsb, SH::F_W0H + fa * NK0), a1h = rr_lds_frag(ldsb, SH::F_W0H + (fa + 1) * NK0);
#pragma unroll
                for (int c = 0; c < NKS; ++c) {
                    const int i = pass * 2 * NKS + 2 * c;          // stream position of (pass, c, first block of the pair); the second follows
                    const rr_u4 bh = rr_u4{ph[c][0], ph[c][1], ph[c][2], ph[c][3]}, bl = rr_u4{pl[c][0], pl[c][1], pl[c][2], pl[c][3]};
                    rr_u4 n0h = a0h, n1h = a1h;
                    if (c + 1 < NKS) { n0h = rr_lds_frag(ldsb, SH::F_W0H + fa * NK0 + c + 1); n1h = rr_lds_frag(ldsb, SH::F_W0H + (fa + 1) * NK0 + c + 1); }
                    const rr_u4 a0l = ql[i % PD], a1l = ql[(i + 1) % PD];
                    RR_MFH(a0h, bl, hacc[fa]);
                    RR_MFH(a1h, bl, hacc[fa + 1]);
                    RR_MFH(a0l, bh, hacc[fa]);
                    RR_MFH(a1l, bh, hacc[fa + 1]);
                    RR_MFH(a0h, bh, hacc[fa]);
                    RR_MFH(a1h, bh, hacc[fa + 1]);
                    if (i + PD < NSL) {
                        ql[i % PD] = RP_STREAM(i + PD);
                        ql[(i + 1) % PD] = RP_STREAM(i + 1 + PD);
                    }
                    if (pass == 0 && c + 1 < NKS) ssplit(c + 1);
                    if (pass > 0) {
                        // activation + row maximum of the previous pass's accumulators, 2 registers of each tile per step
#pragma unroll
                        for (int u = 0; u < 2; ++u)
#pragma unroll
                            for (int r = 2 * c; r < 2 * c + 2; ++r) {
                                hacc[fa - 2 + u][r] = rr_imax(hacc[fa - 2 + u][r], lo_0);
                            }
                        m2 = fmaxf(fmaxf(fabsf(hacc[fa - 2][2 * c]), fabsf(hacc[fa - 2][2 * c + 1])), m2);
                        m2 = fmaxf(fmaxf(fabsf(hacc[fa - 1][2 * c]), fabsf(hacc[fa - 1][2 * c + 1])), m2);
                    }
                    RR_SB();
                    a0h = n0h; a1h = n1h;
                }
            }
        }
        const unsigned t4 = clk();
        // =====================================================================================================================
        // node stage 1: OUT = H W1^T, rows leave as 128-byte row segments
        // =====================================================================================================================
#pragma unroll
        for (int fbo = WB - 2; fbo < WB; ++fbo)
#pragma unroll
            for (int r = 0; r < 16; ++r) hacc[fbo][r] = rr_imax(hacc[fbo][r], lo_0);
#pragma unroll
        for (int fbo = WB - 2; fbo < WB; ++fbo)
#pragma unroll
            for (int r = 0; r < 16; r += 2) m2 = fmaxf(fmaxf(fabsf(hacc[fbo][r]), fabsf(hacc[fbo][r + 1])), m2);
        const unsigned m2b = rr_xhalf_max(__float_as_uint(m2));
        int e2 = (int)(m2b >> 23) + (e_t - 141 - E0);                // exponent field of the largest |h| in true units
        e2 = e2 < 15 ? 15 : (e2 > 254 ? 254 : e2);
        float f2 = rr_pow2(e_t - e2 - E0 + 127);                     // accumulator units -> scaled planes
        float inv2 = rr_pow2(e2 - 14 - E1);                          // 1 / (row scale x matrix scale of stage 1)
        if (m2b >= 0x7f800000u || badrow) { f2 = __uint_as_float(0x7fc00000u); inv2 = f2; badrow = true; }
        const bool anybad = __builtin_amdgcn_ballot_w64(badrow) != 0ull;
        rr_u4 Hh[NKS], Hl[NKS];
        auto hsplit = [&](int c) {
            const int fbo = c >> 1, cc = c & 1;
            unsigned h[4], l[4];
#pragma unroll
            for (int q = 0; q < 2; ++q)
                rr_split4s(hacc[fbo][8 * cc + 4 * q], hacc[fbo][8 * cc + 4 * q + 1], hacc[fbo][8 * cc + 4 * q + 2], hacc[fbo][8 * cc + 4 * q + 3], f2,
                           h[2 * q], l[2 * q], h[2 * q + 1], l[2 * q + 1]);
            Hh[c] = rr_u4{h[0], h[1], h[2], h[3]};
            Hl[c] = rr_u4{l[0], l[1], l[2], l[3]};
        };
        hsplit(0);
        const unsigned t5 = clk();
#if RP_S1T
        // OUT^T[f][t] = W1[f][:] H^T[:][t]: the weight fragments as the A operand, the hidden planes as B (the two operand layouts are the
        // same, so the planes made from the H^T accumulators serve as either).  The C layout then gives a lane FOUR CONSECUTIVE features
        // of ITS OWN target per register quad (features 8 j + 4 h .. + 4 of the block, registers 4 j .. + 4): the row scale is the lane's own
        // (no permutes), a quad leaves as one 16-byte store -- 32 stores per tile instead of 128 dwords (with OUT = H W1^T a lane held one
        // feature of 16 targets).  Rows past nn are dropped by the buffer's range check.
        const int voff_lane = (li * (32 * WB) + 4 * lh) * 4;        // byte offset of (row li, column 4 h) in the tile's output rows
        const __amdgpu_buffer_rsrc_t orow = __builtin_amdgcn_make_buffer_rsrc(a.out + (int64_t)m0 * (32 * WB), 0, nn * (32 * WB * 4), 0x00020000);
        const rr_f2 inv22 = rr_f2{inv2, inv2};
        // one register quad of a 32 x 32 output tile: un-scale, bias, activation, store
        auto put4 = [&](const f32x16 &o, int fbn, int j) {
            int lic = 4 * lh;                                        // (the bias reads are loop-invariant to the compiler, which hoists and spills them: address made here)
            asm volatile("" : "+v"(lic));
            const float4 cv = *reinterpret_cast<const float4 *>(tab + 2 * 32 * WB + 32 * fbn + 8 * j + lic);
            const rr_f2 t0 = __builtin_elementwise_fma(rr_f2{o[4 * j], o[4 * j + 1]}, inv22, rr_f2{cv.x, cv.y});
            const rr_f2 t1 = __builtin_elementwise_fma(rr_f2{o[4 * j + 2], o[4 * j + 3]}, inv22, rr_f2{cv.z, cv.w});
            const rr_u4 y = rr_u4{(unsigned)max(__float_as_int(t0[0]), lo_1), (unsigned)max(__float_as_int(t0[1]), lo_1),
                                  (unsigned)max(__float_as_int(t1[0]), lo_1), (unsigned)max(__float_as_int(t1[1]), lo_1)};
#ifdef RR_ABL_NOSTORE
            asm volatile("" :: "v"(y));
#else
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, y), orow, voff_lane, (32 * fbn + 8 * j) * 4, RP_STORE_AUX);
#endif
        };
        f32x16 po0, po1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { po0[r] = 0.f; po1[r] = 0.f; }
        {
#pragma unroll
            for (int fp = 0; fp < WB; fp += 2) {
                // two output feature blocks = two accumulator chains issued alternately; weight fragments one step ahead; under the products of
                // the first pair the planes of the next 16 hidden features, under those of the second the first pair's output rows
                f32x16 o0, o1;
#pragma unroll
                for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
                rr_u4 b0h = rr_lds_frag(ldsb, SH::F_W1 + 2 * (fp * NKS)), b0l = rr_lds_frag(ldsb, SH::F_W1 + 2 * (fp * NKS) + 1);
                rr_u4 b1h = rr_lds_frag(ldsb, SH::F_W1 + 2 * ((fp + 1) * NKS)), b1l = rr_lds_frag(ldsb, SH::F_W1 + 2 * ((fp + 1) * NKS) + 1);
                // Everything this wave has in flight must have landed BEFORE its first output store is issued: loads and stores leave one
                // counter out of order, so the first wait for a load behind a store is a wait for every store (the compiler emits vmcnt(0)) --
                // measured: 2 900 of a tile's 22 600 cycles at the top of the next tile.  After this point nothing is waited for until the
                // next block's top, one edge stage later, when the stores are long done.
                if (RP_DRAIN && fp == (RP_S1E ? 2 : 0)) __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0), gfx9 encoding
#pragma unroll
                for (int c = 0; c < NKS; ++c) {
                    rr_u4 n0h = b0h, n0l = b0l, n1h = b1h, n1l = b1l;
                    if (c + 1 < NKS) {
                        n0h = rr_lds_frag(ldsb, SH::F_W1 + 2 * (fp * NKS + c + 1)); n0l = rr_lds_frag(ldsb, SH::F_W1 + 2 * (fp * NKS + c + 1) + 1);
                        n1h = rr_lds_frag(ldsb, SH::F_W1 + 2 * ((fp + 1) * NKS + c + 1)); n1l = rr_lds_frag(ldsb, SH::F_W1 + 2 * ((fp + 1) * NKS + c + 1) + 1);
                    }
                    RR_MFH(b0h, Hl[c], o0);
                    RR_MFH(b1h, Hl[c], o1);
                    RR_MFH(b0l, Hh[c], o0);
                    RR_MFH(b1l, Hh[c], o1);
                    RR_MFH(b0h, Hh[c], o0);
                    RR_MFH(b1h, Hh[c], o1);
                    if (fp == 0 && c + 1 < NKS) hsplit(c + 1);
                    if (fp == 0 && c == RP_GATHER_AT) gather(rnext, lh);      // the next tile's first block
                    if (fp > 0 && RP_S1E) { if (c < 4) put4(po0, fp - 2, c); else put4(po1, fp - 1, c - 4); }
                    RR_SB();
                    b0h = n0h; b0l = n0l; b1h = n1h; b1l = n1l;
                }
                if (fp + 2 < WB && RP_S1E) { po0 = o0; po1 = o1; continue; }
                // the last pair (or every pair without the overlap): behind its own products
#pragma unroll
                for (int j = 0; j < 4; ++j) put4(o0, fp, j);
#pragma unroll
                for (int j = 0; j < 4; ++j) put4(o1, fp + 1, j);
                RR_SB();
            }
        }
        // rows that must come out NaN (a non-finite value reached them; the integer maximum above may have dropped a NaN of negative sign):
        // written again, behind the tile's stores.  Rare: a branch around it.
        if (anybad) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (badrow) {
                const rr_u4 qn = rr_u4{0x7fc00000u, 0x7fc00000u, 0x7fc00000u, 0x7fc00000u};
#pragma unroll
                for (int q = 0; q < 4 * WB; ++q)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, qn), orow, voff_lane, 8 * q * 4, 0);
            }
        }
#else
        // OUT[t][f] = H[t][:] W1^T: the hidden planes as the A operand; the C layout gives 32 lanes 32 consecutive floats of one row: every
        // store instruction writes two full 128-byte row segments (the transposed product, RP_S1T, needs a quarter of the store
        // instructions but writes 32-byte pieces of 32 rows each: slower, profiles/r04_layer_rp_variants.txt)
        float invr[16];                                             // (the row scale travels to the accumulator's REGISTER index; made under the last products of the first pair)
        const int voff_lane = (4 * lh * 32 * WB + li) * 4;          // byte offset of (row 4 h, column li) in an output tile
        const __amdgpu_buffer_rsrc_t orow = __builtin_amdgcn_make_buffer_rsrc(a.out + (int64_t)m0 * (32 * WB), 0, nn * (32 * WB * 4), 0x00020000);
        // four output rows (registers 4 j .. + 4) of one 32 x 32 tile: un-scale, bias, activation, store
        auto put4 = [&](const f32x16 &o, int fbn, float cb, int j) {
            const rr_f2 cb2 = rr_f2{cb, cb};
#ifdef RR_ABL_STORE4      // (ablation, results in the wrong places: the four values as ONE 16-byte store, every instruction two full 512-byte rows)
            {
                unsigned v4[4];
#pragma unroll
                for (int r = 4 * j; r < 4 * j + 4; r += 2) {
                    const rr_f2 t = __builtin_elementwise_fma(rr_f2{o[r], o[r + 1]}, rr_f2{invr[r], invr[r + 1]}, cb2);
                    v4[r - 4 * j] = (unsigned)max(__float_as_int(t[0]), lo_1); v4[r - 4 * j + 1] = (unsigned)max(__float_as_int(t[1]), lo_1);
                }
                typedef unsigned u4s __attribute__((__vector_size__(4 * sizeof(unsigned))));
                __builtin_amdgcn_raw_buffer_store_b128(u4s{v4[0], v4[1], v4[2], v4[3]}, orow, (lh * 32 * WB + 4 * li) * 4, (2 * (4 * fbn + j)) * (32 * WB * 4), RP_STORE_AUX);
                return;
            }
#endif
#pragma unroll
            for (int r = 4 * j; r < 4 * j + 4; r += 2) {
                const rr_f2 t = __builtin_elementwise_fma(rr_f2{o[r], o[r + 1]}, rr_f2{invr[r], invr[r + 1]}, cb2);
#pragma unroll
                for (int u = 0; u < 2; ++u)
#ifdef RR_ABL_NOSTORE
                    asm volatile("" :: "v"(max(__float_as_int(t[u]), lo_1)));
#else
                    __builtin_amdgcn_raw_buffer_store_b32((unsigned)max(__float_as_int(t[u]), lo_1), orow, voff_lane,
                                                          (((r + u) & 3) + 8 * ((r + u) >> 2)) * (32 * WB * 4) + 32 * fbn * 4, RP_STORE_AUX);    // (row / block offset: scalar)
#endif
            }
        };
        f32x16 po0, po1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { po0[r] = 0.f; po1[r] = 0.f; }
        {
#pragma unroll
            for (int fp = 0; fp < WB; fp += 2) {
                // two output feature blocks = two accumulator chains issued alternately; weight fragments one step ahead; under the products of
                // the first pair the planes of the next 16 hidden features, under those of the second the first pair's output rows
                f32x16 o0, o1;
#pragma unroll
                for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
                rr_u4 b0h = rr_lds_frag(ldsb, SH::F_W1 + 2 * (fp * NKS)), b0l = rr_lds_frag(ldsb, SH::F_W1 + 2 * (fp * NKS) + 1);
                rr_u4 b1h = rr_lds_frag(ldsb, SH::F_W1 + 2 * ((fp + 1) * NKS)), b1l = rr_lds_frag(ldsb, SH::F_W1 + 2 * ((fp + 1) * NKS) + 1);
                if (RP_DRAIN && fp == (RP_S1E ? 2 : 0)) __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0), gfx9 encoding (see RP_DRAIN)
                // (the bias reads are loop-invariant to the compiler, which hoists them in front of the stage and spills them: address made here)
                int lic = li;
                asm volatile("" : "+v"(lic));
                const float cbp0 = fp > 0 ? tab[2 * 32 * WB + 32 * (fp - 2) + lic] : 0.f, cbp1 = fp > 0 ? tab[2 * 32 * WB + 32 * (fp - 1) + lic] : 0.f;
#pragma unroll
                for (int c = 0; c < NKS; ++c) {
                    rr_u4 n0h = b0h, n0l = b0l, n1h = b1h, n1l = b1l;
                    if (c + 1 < NKS) {
                        n0h = rr_lds_frag(ldsb, SH::F_W1 + 2 * (fp * NKS + c + 1)); n0l = rr_lds_frag(ldsb, SH::F_W1 + 2 * (fp * NKS + c + 1) + 1);
                        n1h = rr_lds_frag(ldsb, SH::F_W1 + 2 * ((fp + 1) * NKS + c + 1)); n1l = rr_lds_frag(ldsb, SH::F_W1 + 2 * ((fp + 1) * NKS + c + 1) + 1);
                    }
                    RR_MFH(Hl[c], b0h, o0);
                    RR_MFH(Hl[c], b1h, o1);
                    RR_MFH(Hh[c], b0l, o0);
                    RR_MFH(Hh[c], b1l, o1);
                    RR_MFH(Hh[c], b0h, o0);
                    RR_MFH(Hh[c], b1h, o1);
                    if (fp == 0 && c + 1 < NKS) hsplit(c + 1);
                    if (fp == 0 && c == RP_GATHER_AT) gather(rnext, lh);      // the next tile's first block
                    if (fp == 0 && c + 1 == NKS) {
                        int inv2l = __float_as_int(inv2);                    // (opaque here: the permutes would otherwise be issued in front of the stage and their results spilled)
                        asm volatile("" : "+v"(inv2l));
#pragma unroll
                        for (int r = 0; r < 16; ++r) invr[r] = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * rr_crow(r, lh), inv2l));
                    }
                    if (fp > 0 && RP_S1E) { if (c < 4) put4(po0, fp - 2, cbp0, c); else put4(po1, fp - 1, cbp1, c - 4); }
                    RR_SB();
                    b0h = n0h; b0l = n0l; b1h = n1h; b1l = n1l;
                }
                if (fp + 2 < WB && RP_S1E) { po0 = o0; po1 = o1; continue; }
                // the last pair (or every pair without the overlap): behind its own products
                int lid = li;
                asm volatile("" : "+v"(lid));
                const float cb0 = tab[2 * 32 * WB + 32 * fp + lid], cb1 = tab[2 * 32 * WB + 32 * (fp + 1) + lid];
#pragma unroll
                for (int j = 0; j < 4; ++j) put4(o0, fp, cb0, j);
#pragma unroll
                for (int j = 0; j < 4; ++j) put4(o1, fp + 1, cb1, j);
                RR_SB();
            }
        }
        // rows that must come out NaN (a non-finite value reached them: their inverse scale is NaN; the integer maximum above may have
        // dropped a NaN of negative sign): written again, behind the tile's stores.  Rare: a branch around it.
        if (anybad) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (invr[r] != invr[r]) {
#pragma unroll
                    for (int fb = 0; fb < WB; ++fb)
                        __builtin_amdgcn_raw_buffer_store_b32(0x7fc00000u, orow, voff_lane, ((r & 3) + 8 * (r >> 2)) * (32 * WB * 4) + 32 * fb * 4, 0);
                }
        }
#endif
        if (PROF) { const unsigned t6 = clk(); pc[7] += 1; pc[3] += t4 - t3; pc[4] += t5 - t4; pc[5] += t6 - t5; }
    }
    if (PROF && prof && lane0 == 0 && (range == 0 || range == a.n_ranges / 2)) {
        unsigned long long *o = prof + (range == 0 ? 0 : 16);
        for (int q = 0; q < 8; ++q) o[q] = pc[q];
        o[8] = clk() - t_start;
    }
    if (PROF && prof && lane0 == 0) {                  // every wave's life on the constant 100 MHz clock: how evenly the ranges end
        prof[32 + 2 * range] = wave_t0;
        prof[32 + 2 * range + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// fp32 rows -> a column range of an fp16 pack, checking that every value is exact in fp16 and < 2 in magnitude (what the edge stage's
// weight scale is made for).  One thread per (row, group of four columns).
__global__ void pack16_rows_kernel(const float *src, int64_t rows, int width, uint16_t *dst, int dst_stride, int col0, int one_col, int *status) {
    const int64_t gq = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int wq = width > 0 ? (width + 3) >> 2 : 1;
    const int64_t row = gq / wq;
    const int q = (int)(gq % wq);
    if (row >= rows) return;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = 4 * q + j;
        if (c >= width) break;
        const float v = src[row * width + c];
        const _Float16 h = (_Float16)v;
        bad = bad || !((float)h == v) || !(fabsf(v) < 2.f);
        dst[row * dst_stride + col0 + c] = __builtin_bit_cast(uint16_t, h);
    }
    if (q == 0 && one_col >= 0) dst[row * dst_stride + one_col] = 0x3c00;
    if (bad) atomicOr(status, 1);
}

static int rp_blocks(const gsn_chain_stage *edge, int64_t d_x, const float *x, int *edge_cols) {
    // blocks: x through one index, x through another, then the edge-level blocks through one common index; <= 16 columns of those
    if (edge->n_blocks < 2 || edge->n_blocks > 6 || !edge->blocks) return 0;
    const gsn_block *b = edge->blocks;
    if (b[0].width != d_x || b[1].width != d_x || !b[0].idx32 || !b[1].idx32 || b[0].idx || b[1].idx) return 0;
    if (x && (b[0].data != x || b[1].data != x)) return 0;
    if (b[0].data != b[1].data) return 0;
    int64_t cols = 0;
    for (int i = 2; i < edge->n_blocks; ++i) {
        if (!b[i].idx32 || b[i].idx || b[i].idx32 != b[2].idx32 || b[i].width <= 0) return 0;
        cols += b[i].width;
    }
    if (cols > 16) return 0;
    *edge_cols = (int)cols;
    return 1;
}

int rp_supported(const gsn_chain_stage *edge, int64_t d_x, const gsn_chain_stage *node0, const gsn_chain_stage *node1) {
    if (!rr_shape_ok(edge, d_x, node0, node1)) return 0;
    int cols = 0;
    return rp_blocks(edge, d_x, nullptr, &cols);
}

int rp_forward(int64_t n_nodes, int64_t n_edges, const int32_t *seg_ptr, const gsn_chain_stage *edge, const float *x, int64_t d_x,
               const gsn_chain_stage *node0, const gsn_chain_stage *node1, const void *prepared, const gsn_pack16 *pack, int64_t edge_rows,
               float *out, hipStream_t st) {
    using SH = RrShape<4, 2>;
    int cols = 0;
    if (!rp_blocks(edge, d_x, x, &cols)) return set_error(GSN_E_INVALID, "gsn_layer_fused_fwd_pack16_hip: the edge stage is not cat(x[i], x[j], edge-level blocks through one index)");
    if (!pack || !pack->node_rows || (cols > 0 && !pack->edge_rows)) return set_error(GSN_E_INVALID, "gsn_layer_fused_fwd_pack16_hip: null pack");
    if ((reinterpret_cast<uintptr_t>(pack->node_rows) | reinterpret_cast<uintptr_t>(pack->edge_rows)) & 15) return set_error(GSN_E_INVALID, "gsn_layer_fused_fwd_pack16_hip: packs must be 16-byte aligned");
    if (n_nodes * 64 > 0x7fffffffll || edge_rows * 32 > 0x7fffffffll) return 1;        // (32-bit buffer offsets: the fp32 kernel takes the call)
    if (reinterpret_cast<const unsigned *>(prepared) == nullptr) return set_error(GSN_E_INVALID, "gsn_layer_fused_fwd_pack16_hip: null prepared");
    RpArgs a{};
    a.n_nodes = (int)n_nodes; a.n_edges = (int)n_edges; a.seg_ptr = seg_ptr;
    a.role_idx[0] = edge->blocks[0].idx32; a.role_idx[1] = edge->blocks[1].idx32;
    a.role_idx[2] = cols > 0 ? edge->blocks[2].idx32 : edge->blocks[0].idx32;
    a.node16 = pack->node_rows; a.node_bytes = (unsigned)(n_nodes * 64);
    if (cols > 0) { a.edge16 = pack->edge_rows; a.edge_bytes = (unsigned)(edge_rows * 32); a.edge_shift = 5; }
    else { a.edge16 = pack->node_rows; a.edge_bytes = a.node_bytes; a.edge_shift = 6; }      // (finite values under zero weights)
    a.d_x = (int)d_x; a.out = out; a.prep = reinterpret_cast<const unsigned *>(prepared);
    const int64_t n_tiles = (n_nodes + RR_TN - 1) / RR_TN;
    int64_t gx = 256;
    { const char *d = getenv("GSN_FUSED_GRID"); if (d && atoi(d) > 0) gx = atoi(d); }
    int64_t ranges = gx * RR_NW;
    if (ranges > n_tiles) ranges = n_tiles;
    if (gx > ranges) gx = ranges;
    a.n_ranges = (int)ranges;
    {
        static const double share = [] { const char *d = getenv("GSN_RP_OLD_SHARE"); return d ? atof(d) : 0.6; }();
        a.n_split = (ranges == gx * RR_NW && share > 0.0 && share < 1.0) ? (int)((double)n_nodes * share) : 0;
    }
    static const bool prof_on = [] { const char *d = getenv("GSN_FUSED_PROF"); return d && atoi(d) != 0; }();
    const void *fn = prof_on ? reinterpret_cast<const void *>(&layer_fused_kernel_rp<4, 2, true>) : reinterpret_cast<const void *>(&layer_fused_kernel_rp<4, 2, false>);
    static DeviceOnce attr_set;
    const int attr_dev = current_device();
    if (!attr_set.done(attr_dev)) {
        hipError_t e0 = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e0 != hipSuccess) return set_error(GSN_E_HIP, "hipFuncSetAttribute(layer_fused_kernel_rp): %s", hipGetErrorString(e0));
        attr_set.mark(attr_dev);
    }
    if (getenv("GSN_CHAIN_TRACE")) fprintf(stderr, "gsn chain: layer_fused_kernel_rp<4,2> nodes %d edges %d grid %lld ranges %d\n", a.n_nodes, a.n_edges, (long long)gx, a.n_ranges);
    if (prof_on) {
        unsigned long long *prof = nullptr;
        const size_t pn = 32 + 2 * (size_t)a.n_ranges;
        (void)hipMalloc(&prof, pn * 8); (void)hipMemsetAsync(prof, 0, pn * 8, st);
        hipLaunchKernelGGL((layer_fused_kernel_rp<4, 2, true>), dim3((unsigned)gx), dim3(64 * RR_NW), SH::LDS_BYTES, st, a, prof);
        std::vector<unsigned long long> hv(pn);
        unsigned long long *h = hv.data();
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h, prof, pn * 8, hipMemcpyDeviceToHost);
        (void)hipFree(prof);
        static int shown = 0;
        if (shown % 8 == 7) {
            unsigned long long t_lo = ~0ull, t_hi = 0, e_lo = ~0ull;
            double life = 0.0, lmin = 1e30, lmax = 0.0;
            for (int r = 0; r < a.n_ranges; ++r) {
                const unsigned long long s0 = h[32 + 2 * r], s1 = h[32 + 2 * r + 1];
                t_lo = s0 < t_lo ? s0 : t_lo; t_hi = s1 > t_hi ? s1 : t_hi; e_lo = s1 < e_lo ? s1 : e_lo;
                const double l = (double)(s1 - s0);
                life += l; lmin = l < lmin ? l : lmin; lmax = l > lmax ? l : lmax;
            }
            fprintf(stderr, "rpprof waves %d: first start -> last end %.1f us | wave life mean %.1f min %.1f max %.1f us | first end after %.1f us | mean life / span %.3f\n",
                    a.n_ranges, (t_hi - t_lo) / 100.0, life / a.n_ranges / 100.0, lmin / 100.0, lmax / 100.0, (e_lo - t_lo) / 100.0,
                    life / a.n_ranges / (double)(t_hi - t_lo));
        }
        if (shown++ % 8 == 7)
            for (int w = 0; w < 2; ++w) {
                const unsigned long long *o = h + 16 * w;
                const double nb = o[6] ? (double)o[6] : 1.0, nt = o[7] ? (double)o[7] : 1.0;
                fprintf(stderr, "rpprof range %s: blocks %llu tiles %llu total %llu cycles | per block: top %.0f edge %.0f | per tile: exposed edge epilogue %.0f stage0 %.0f split %.0f stage1 %.0f\n",
                        w ? "mid" : "0", o[6], o[7], o[8], o[0] / nb, o[1] / nb, o[2] / nt, o[3] / nt, o[4] / nt, o[5] / nt);
            }
    } else {
        hipLaunchKernelGGL((layer_fused_kernel_rp<4, 2, false>), dim3((unsigned)gx), dim3(64 * RR_NW), SH::LDS_BYTES, st, a, (unsigned long long *)nullptr);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "layer_fused_kernel_rp: %s", hipGetErrorString(e));
    return GSN_OK;
}

}  // namespace gsn

using namespace gsn;

extern "C" int gsn_pack16_rows_hip(const float *src, int64_t rows, int64_t width, uint16_t *dst, int64_t dst_stride, int64_t col0,
                                   int64_t one_col, int32_t *status, void *stream) {
    if (rows < 0 || width < 0 || dst_stride <= 0 || col0 < 0 || col0 + width > dst_stride || one_col >= dst_stride)
        return set_error(GSN_E_INVALID, "gsn_pack16_rows_hip: rows %lld width %lld stride %lld col0 %lld one_col %lld", (long long)rows, (long long)width,
                         (long long)dst_stride, (long long)col0, (long long)one_col);
    if (rows == 0 || (width == 0 && one_col < 0)) return GSN_OK;
    if (!dst || !status || (width > 0 && !src)) return set_error(GSN_E_INVALID, "gsn_pack16_rows_hip: null pointer");
    const int wq = (int)((width + 3) >> 2) > 0 ? (int)((width + 3) >> 2) : 1;
    const int64_t total = rows * wq;
    hipLaunchKernelGGL(pack16_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, rows, (int)width, dst,
                       (int)dst_stride, (int)col0, (int)(one_col < 0 ? -1 : one_col), status);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "pack16_rows_kernel: %s", hipGetErrorString(e));
    return GSN_OK;
}
