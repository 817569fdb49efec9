// Weight gradient of a dense stage from the fp16 planes its neighbours already made (r06):
//
//     gW[n_out][K] += gH^T X            (the adjoint of models_misc.py:52-58's Linear; torch.nn.Linear's weight.grad)
//
// backward.hip's wgrad_bf16_pipe_kernel reads gH and X as fp32 and splits every value into three bf16 planes in EVERY column tile that
// needs it (~6.5 vector instructions per value, six plane products): at the d = 300 ogb stages (105 k rows, 300 <-> 600 columns) it is
// bound by that vector stream at 0.22 of the matrix pipe.  But both operands exist as fp16 planes already: the forward product of the stage
// split X (gsn_linear_f16x3_fwd_hip's row pre-pass: per row a power-of-two scale, a high and a low half per value) and the input-gradient
// product gX = gH W split gH the same way.  This kernel multiplies those planes: three fp16 plane products per fp32 product (hh, hl, lh),
// per value two byte permutes and a packed multiply.
//
// Row scales.  The planes of row r hold g_r * sg_r and x_r * sx_r; the term of row r needs the factor 1 / (sg_r sx_r), which differs from
// row to row INSIDE the reduction.  A workgroup reduces a slab of rows: it takes E = max over its rows of the exponent of 1 / (sg_r sx_r)
// and multiplies the planes of row r by 2^(e_r - E) <= 1, half of the exponent on each operand (exact in fp16 while a value stays above 2^-14
// in the high plane and 2^-24 in the low one: rows far below the slab's largest lose low-plane bits that are far below that largest term, rows
// 2^-48 below it vanish: absolute error as in an fp32 sum of the same terms), accumulates in fp32 and multiplies its tile by 2^E before the
// atomic adds.  No pass over gH or X, no workspace.
//
// Data flow per 16-row step of a 128 x 128 tile (4 waves): wave w takes K slice w of both operands, one instruction = two whole 128-byte
// lines (rows 2i and 2i + 1 of the slice: 32 high halfs | 32 low halfs), eight instructions per operand -- lanes 0..31 end up with the
// EVEN rows of a column pair, lanes 32..63 with the odd rows, lanes 16..31 / 48..63 with the low plane.  The order of the 16 rows inside
// the reduction index of the products is free as long as both operands agree: k-group 0 = even rows, k-group 1 = odd rows, so a lane's
// eight values of a column ARE one 16-byte operand fragment and go to LDS with one write; rows two steps ahead are in flight.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "gsn_internal.h"

namespace gsn {

namespace {

constexpr int WF_T = 128;          // gW tile edge
constexpr int WF_RB = 16;          // rows per step
constexpr int WF_MAX_ROWS = 3072;  // rows per slab (the scale tables live in LDS: both kernels stay under 64 KiB of dynamic LDS, 42 / 61 KiB)

typedef float wf_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 wf_h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 wf_h16x2 __attribute__((ext_vector_type(2)));
typedef unsigned wf_un4 __attribute__((ext_vector_type(4)));

struct WfArgs {
    int64_t m_rows, rows_per_wg;
    int n_out, k_total;            // gW [n_out][k_total]
    int g_slices, x_slices;        // 32-column slices of the two plane sets (row pitch = slices * 128 bytes)
    int tn, tk;                    // tiles along n_out / K
    const float *g_inv, *x_inv;    // inverse row scales (powers of two; NaN for a row with a non-finite value)
    const unsigned char *g_planes, *x_planes;
    float *gw;
    unsigned long long *prof;      // diagnostic build (dbg & 16): cycles of workgroup 0, wave 0: prologue | loop | barrier waits | epilogue | steps
    int dbg;                       // diagnostic build: 1 no atomics, 2 no products, 4 no loads behind the first two steps, 8 no staging
};

#ifndef WF_WAVES
#define WF_WAVES 2                 // waves per SIMD the register allocation aims at (A/B builds: 3 spills 14-33 registers)
#endif
template <int VALU_PER_MFMA, bool DBG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WF_WAVES, WF_WAVES))) void wgrad_f16x3_kernel(WfArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wf_smem[];
    // LDS (no static __shared__ in front of it: the 16-byte accesses need the base aligned).  Per operand and buffer: [plane][k-group] blocks
    // of 128 fragments of 16 bytes, even columns first, then (128 bytes of padding further) the odd columns -- a lane holds columns 2c and
    // 2c + 1: eight neighbouring lanes write 128 contiguous bytes (ds_write_b128 works in groups of 8 lanes over 32 banks), and a fragment
    // read of 32 consecutive columns touches every bank once per 16-lane group (parity stride = 128 mod 256 bytes).
    constexpr int WF_PAR = 64 * 16 + 128, WF_BLK = 2 * WF_PAR, WF_OPB = 4 * WF_BLK;      // parity stride, [plane][k-group] block, operand buffer
    constexpr int WF_TAB = 4 * WF_OPB;                                                   // (ta: buffers 0, 1 | tb: buffers 0, 1) then the tables
    unsigned char *const ta = wf_smem, *const tb = wf_smem + 2 * WF_OPB;
    unsigned short *ctab = reinterpret_cast<unsigned short *>(wf_smem + WF_TAB + 16);
    int &s_emax = *reinterpret_cast<int *>(wf_smem + WF_TAB);
    auto frag = [&](unsigned char *t, int buf, int plane, int kg, int col) -> wf_un4 & {
        return *reinterpret_cast<wf_un4 *>(t + buf * WF_OPB + (plane * 2 + kg) * WF_BLK + (col & 1) * WF_PAR + (col >> 1) * 16);
    };

    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int ntile = a.tn * a.tk;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int tile = seq % ntile;
    const int64_t slab = (int64_t)(seq / ntile) * 8 + xcd;
    const int n0 = (tile / a.tk) * WF_T, k0 = (tile % a.tk) * WF_T;
    const int64_t r_begin = slab * a.rows_per_wg;
    int64_t r_end = r_begin + a.rows_per_wg;
    if (r_end > a.m_rows) r_end = a.m_rows;
    if (r_begin >= r_end) return;                 // (block-uniform)
    const int n_rows = (int)(r_end - r_begin);
    const int n_rows16 = (n_rows + 47) / 48 * 48;        // (the loop below runs three steps per trip, no exits in between: rows past the slab get zero scales)
    // (the tables have 16 zero entries behind the slab: the last step stages one block too many, unconditionally)
    unsigned short *xtab = ctab + n_rows16 + 16, *raw = xtab + n_rows16 + 16;

    // ---- the slab's scale table -------------------------------------------------------------------------------------------------------
    if (tid == 0) s_emax = 0;
    __syncthreads();
    {
        int mymax = 0;
        for (int r = tid; r < n_rows16; r += 256) {
            int e = 0;                            // 0: the row adds nothing (past the end, or a zero scale)
            if (r < n_rows) {
                const unsigned fg = (__float_as_uint(a.g_inv[r_begin + r]) >> 23) & 255u, fx = (__float_as_uint(a.x_inv[r_begin + r]) >> 23) & 255u;
                if (fg == 255u || fx == 255u) e = 0x7fff;                     // a non-finite row: NaN into every product it takes part in
                else if (fg != 0u && fx != 0u) { e = (int)(fg + fx); mymax = max(mymax, e); }
            }
            raw[r] = (unsigned short)e;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mymax = max(mymax, __shfl_xor(mymax, o));
        if (lane == 0 && mymax > 0) atomicMax(&s_emax, mymax);
    }
    __syncthreads();
    const int emax = s_emax;
    if (tid < 16) { ctab[n_rows16 + tid] = 0; xtab[n_rows16 + tid] = 0; }
    // 2^-d as a half (d >= 0): exact down to 2^-24, zero below
    auto pow2h = [](int d) -> unsigned short { return d <= 14 ? (unsigned short)((15 - d) << 10) : (d <= 24 ? (unsigned short)(1u << (24 - d)) : (unsigned short)0); };
    for (int r = tid; r < n_rows16; r += 256) {
        const int e = raw[r];
        unsigned short cg = 0, cx = 0;
        if (e == 0x7fff) { cg = 0x7e00; cx = 0x3c00; }
        else if (e != 0) {
            // the row's factor 2^-(emax - e), half of the exponent on each operand: a value 2^-j below its row's largest stays exact in both planes
            // while j + d / 2 <= 17, instead of j + d with the factor on one side
            const int d = emax - e, dg = d >> 1;
            cg = pow2h(dg); cx = pow2h(d - dg);
        }
        // inside a 16-row block: the even rows first, then the odd rows -- a lane's eight rows are one 16-byte read
        const int j = r & 15, at = (r & ~15) + (j & 1) * 8 + (j >> 1);
        ctab[at] = cg;
        xtab[at] = cx;
    }
    // (a slab whose rows are all zero still walks the loop: its planes are zeros -- no special case)

    // ---- lanes -> lines ----------------------------------------------------------------------------------------------------------------
    const int gs = min(n0 / 32 + w, a.g_slices - 1), xs = min(k0 / 32 + w, a.x_slices - 1);       // (slices past the operand: a valid one again, its columns are never written)
    const unsigned g_pitch = (unsigned)a.g_slices * 128u, x_pitch = (unsigned)a.x_slices * 128u;
    const unsigned lofs = (unsigned)(lane & 31) * 4u;
    const unsigned g_odd = lofs + (lh ? g_pitch : 0u), x_odd = lofs + (lh ? x_pitch : 0u);      // (lanes 32..63: the odd row of a pair)
    const unsigned char *gbase = a.g_planes + (int64_t)gs * 128, *xbase = a.x_planes + (int64_t)xs * 128;
    const int dbg = DBG ? a.dbg : 0;
    const bool prof = DBG && (dbg & 16);
    unsigned long long q_start = prof ? clock64() : 0, q_bar = 0, q_loop0 = 0, q_loop1 = 0, q_ph[5] = {0, 0, 0, 0, 0};

    // (no bound checks: a row scratch holds at least 128 rows of zero planes behind its last row, and the scale table zeroes rows past the slab)
    const int64_t g2 = 2 * (int64_t)g_pitch, x2 = 2 * (int64_t)x_pitch;
    auto fetch = [&](unsigned (&pg)[8], unsigned (&px)[8], int64_t row0) {
        const unsigned char *gr = gbase + row0 * g_pitch, *xr = xbase + row0 * x_pitch;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            pg[i] = *reinterpret_cast<const unsigned *>(gr + g_odd);
            px[i] = *reinterpret_cast<const unsigned *>(xr + x_odd);
            gr += g2; xr += x2;
        }
    };
    // eight rows of a column pair -> the two columns' fragments (k index = row order); gH's with the rows' scale factors
    const int pl = (lane >> 4) & 1, cp = w * 32 + 2 * (lane & 15);
    auto pk_mul = [](unsigned v, unsigned c) { return __builtin_bit_cast(unsigned, __builtin_bit_cast(wf_h16x2, v) * __builtin_bit_cast(wf_h16x2, c)); };
    auto stage = [&](int buf, const unsigned (&pg)[8], const unsigned (&px)[8], int64_t row0) {
        const wf_un4 cs = *reinterpret_cast<const wf_un4 *>(ctab + (row0 - r_begin) + lh * 8);
        const wf_un4 ds = *reinterpret_cast<const wf_un4 *>(xtab + (row0 - r_begin) + lh * 8);
        constexpr unsigned LO = 0x05040100u, HI = 0x07060302u;
        const wf_un4 g0 = {pk_mul(__builtin_amdgcn_perm(pg[1], pg[0], LO), cs.x), pk_mul(__builtin_amdgcn_perm(pg[3], pg[2], LO), cs.y),
                           pk_mul(__builtin_amdgcn_perm(pg[5], pg[4], LO), cs.z), pk_mul(__builtin_amdgcn_perm(pg[7], pg[6], LO), cs.w)};
        const wf_un4 g1 = {pk_mul(__builtin_amdgcn_perm(pg[1], pg[0], HI), cs.x), pk_mul(__builtin_amdgcn_perm(pg[3], pg[2], HI), cs.y),
                           pk_mul(__builtin_amdgcn_perm(pg[5], pg[4], HI), cs.z), pk_mul(__builtin_amdgcn_perm(pg[7], pg[6], HI), cs.w)};
        const wf_un4 x0 = {pk_mul(__builtin_amdgcn_perm(px[1], px[0], LO), ds.x), pk_mul(__builtin_amdgcn_perm(px[3], px[2], LO), ds.y),
                           pk_mul(__builtin_amdgcn_perm(px[5], px[4], LO), ds.z), pk_mul(__builtin_amdgcn_perm(px[7], px[6], LO), ds.w)};
        const wf_un4 x1 = {pk_mul(__builtin_amdgcn_perm(px[1], px[0], HI), ds.x), pk_mul(__builtin_amdgcn_perm(px[3], px[2], HI), ds.y),
                           pk_mul(__builtin_amdgcn_perm(px[5], px[4], HI), ds.z), pk_mul(__builtin_amdgcn_perm(px[7], px[6], HI), ds.w)};
        frag(ta, buf, pl, lh, cp) = g0;
        frag(ta, buf, pl, lh, cp + 1) = g1;
        frag(tb, buf, pl, lh, cp) = x0;
        frag(tb, buf, pl, lh, cp + 1) = x1;
    };

    wf_f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    unsigned pg0[8], px0[8], pg1[8], px1[8], pg2[8], px2[8];
    fetch(pg0, px0, r_begin);
    fetch(pg1, px1, r_begin + 16);
    fetch(pg2, px2, r_begin + 32);
    __syncthreads();                              // (the scale table)
    stage(0, pg0, px0, r_begin);
    __syncthreads();
#define WF_MF(x, y, c) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wf_h16x8, x), __builtin_bit_cast(wf_h16x8, y), c, 0, 0, 0)
    // one step: products of the rows staged in `buf`; the rows row0 + 16 (`nxt`, requested two steps ago) staged into buf ^ 1 under them; `far`
    // (staged in the step before) receives the rows row0 + 48 -- three register sets, two steps of memory latency covered
    auto step = [&](int buf, int64_t row0, const unsigned (&npg)[8], const unsigned (&npx)[8], unsigned (&fpg)[8], unsigned (&fpx)[8]) {
        // (the far rows are requested FIRST: the wait in front of the staging below then leaves them in flight)
        const unsigned long long p0 = prof ? clock64() : 0;
        if (!(dbg & 4)) fetch(fpg, fpx, row0 + 48);
        const unsigned long long p1 = prof ? clock64() : 0;
        if (prof) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");          // (the rows about to be staged)
        const unsigned long long p2 = prof ? clock64() : 0;
        wf_un4 fa[2][2], fb[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                fa[i][p] = frag(ta, buf, p, lh, wm * 64 + i * 32 + li);
                fb[i][p] = frag(tb, buf, p, lh, wn * 64 + i * 32 + li);
            }
        if (prof) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long p3 = prof ? clock64() : 0;
        if (!(dbg & 8)) stage(buf ^ 1, npg, npx, row0 + 16);      // (past the slab: zero scales, a buffer nobody reads -- no branch, so that the
                                                                  //  compiler's vmcnt bookkeeping stays exact: two sets of loads stay in flight)
        if (prof) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long p4 = prof ? clock64() : 0;
        if (!(dbg & 2))
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int pa_ = t == 2 ? 0 : (t == 0 ? 1 : 0), pb_ = t == 2 ? 0 : (t == 0 ? 0 : 1);      // lh, hl, hh
            WF_MF(fa[0][pa_], fb[0][pb_], acc[0][0]);
            WF_MF(fa[0][pa_], fb[1][pb_], acc[0][1]);
            WF_MF(fa[1][pa_], fb[0][pb_], acc[1][0]);
            WF_MF(fa[1][pa_], fb[1][pb_], acc[1][1]);
        }
        if (!prof) {
#pragma unroll
            for (int t = 0; t < 12; ++t) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);              // MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_MFMA, 0);  // VALU
            }
        }
        if (prof) { const unsigned long long p5 = clock64(); q_ph[0] += p1 - p0; q_ph[1] += p2 - p1; q_ph[2] += p3 - p2; q_ph[3] += p4 - p3; q_ph[4] += p5 - p4; }
        const unsigned long long qb0 = prof ? clock64() : 0;
        __syncthreads();
        if (prof) q_bar += clock64() - qb0;
    };
    {
        // three steps per trip and no exit inside a trip: one basic block, so that the compiler's vmcnt bookkeeping is exact and two sets of loads
        // stay in flight across the back edge (rows past the slab: zero scales; past the operand: the scratch's zero rows)
        int64_t row0 = r_begin;
        int buf = 0;
        if (prof) q_loop0 = clock64();
        for (int it = n_rows16 / 48; it > 0; --it, row0 += 48) {
            step(buf, row0, pg1, px1, pg0, px0);
            step(buf ^ 1, row0 + 16, pg2, px2, pg1, px1);
            step(buf, row0 + 32, pg0, px0, pg2, px2);
            buf ^= 1;                             // (an odd number of steps per trip: the buffers swap roles from trip to trip)
        }
        if (prof) q_loop1 = clock64();
    }
#undef WF_MF
    // 2^(emax - 254) in two exact factors (either may leave the normal range alone, not both)
    const int ex = emax - 254;
    const int e1 = ex < -126 ? -126 : (ex > 127 ? 127 : ex), e2r = ex - e1, e2 = e2r < -126 ? -126 : (e2r > 127 ? 127 : e2r);
    const float f1 = __uint_as_float((unsigned)(e1 + 127) << 23), f2 = __uint_as_float((unsigned)(e2 + 127) << 23);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kcol = k0 + wn * 64 + j * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nrow = n0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (nrow < a.n_out && kcol < a.k_total && !(dbg & 1)) atomicAdd(a.gw + (int64_t)nrow * a.k_total + kcol, acc[i][j][r] * f1 * f2);
            }
        }
    if (prof && blockIdx.x == 0 && threadIdx.x == 0 && a.prof) {
        a.prof[0] = q_loop0 - q_start; a.prof[1] = q_loop1 - q_loop0; a.prof[2] = q_bar; a.prof[3] = clock64() - q_loop1; a.prof[4] = (unsigned long long)(n_rows16 / 16);
        for (int i = 0; i < 5; ++i) a.prof[5 + i] = q_ph[i];
    }
}

// ---- the same product without staging registers (second kernel of r06) --------------------------------------------------------------------
// global_load_lds_dwordx4 moves whole lines straight into LDS and ds_read_b64_tr_b16 reads the operand fragments TRANSPOSED out of that image:
// no byte permutes, no ds_write pass, no registers holding rows in flight (the kernel above keeps three sets of 16).
//   * one DMA instruction = 8 rows x one K slice x both planes = 8 whole 128-byte lines -> 1 KiB of LDS, lane q's 16 bytes at position q: eight
//     neighbouring lanes fetch one line (the address unit sees whole lines), rows at a pitch of 128 bytes.  Rows 2, 3, 6, 7 of an image swap
//     their high and low halves (the lane fetches piece p ^ 4): without it the four rows of a transposing read fall on two sets of banks.
//   * ds_read_b64_tr_b16 (scripts/micro/tr_probe.hip): in a group of 16 lanes, lane i points at 4 contiguous halfs = row i >> 2, columns
//     4 (i & 3) .. + 3 of a [4][16] block (any row pitch) and lane c receives column c of it.  MFMA lane l wants column l & 31, rows
//     8 (l >> 5) .. + 7: two reads (rows 0-3, 4-7 of the lane's 8-row image), lanes 0-15 / 16-31 on the two column blocks, lanes 32-63 on the
//     slice's second image.  The 32 lanes served together touch 4 rows x 64 bytes on 64 different banks.
//   * the row factors are applied to the FRAGMENTS (rows run along the fragment: the four packed factors are the same for every lane of a half wave).
// Three LDS buffers of 16 KiB: the lines of steps s + 1 and s + 2 are in flight while step s is multiplied; one counted wait + barrier per step.
typedef __fp16 wf_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) wf_fp16x4 *wf_lds4_t;
typedef unsigned wf_un2 __attribute__((ext_vector_type(2)));

template <bool DBG>
__global__ __launch_bounds__(256) void wgrad_f16x3_dma_kernel(WfArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wf_smem[];
    constexpr int WD_OP = 8192, WD_BUF = 2 * WD_OP, WD_TAB = 3 * WD_BUF;     // operand image of a step, buffer (gH | X), tables behind the three buffers
    unsigned short *ctab = reinterpret_cast<unsigned short *>(wf_smem + WD_TAB + 16);
    int &s_emax = *reinterpret_cast<int *>(wf_smem + WD_TAB);
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int ntile = a.tn * a.tk;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int tile = seq % ntile;
    const int64_t slab = (int64_t)(seq / ntile) * 8 + xcd;
    const int n0 = (tile / a.tk) * WF_T, k0 = (tile % a.tk) * WF_T;
    const int64_t r_begin = slab * a.rows_per_wg;
    int64_t r_end = r_begin + a.rows_per_wg;
    if (r_end > a.m_rows) r_end = a.m_rows;
    if (r_begin >= r_end) return;                 // (block-uniform)
    const int n_rows = (int)(r_end - r_begin);
    const int n_rows16 = (n_rows + WF_RB - 1) / WF_RB * WF_RB;
    unsigned short *xtab = ctab + n_rows16 + 16, *raw = reinterpret_cast<unsigned short *>(wf_smem);      // (raw exponents: in the buffers, before any line lands)

    // ---- the slab's scale tables (as above, rows in natural order) ----------------------------------------------------------------------
    if (tid == 0) s_emax = 0;
    __syncthreads();
    {
        int mymax = 0;
        for (int r = tid; r < n_rows16; r += 256) {
            int e = 0;
            if (r < n_rows) {
                const unsigned fg = (__float_as_uint(a.g_inv[r_begin + r]) >> 23) & 255u, fx = (__float_as_uint(a.x_inv[r_begin + r]) >> 23) & 255u;
                if (fg == 255u || fx == 255u) e = 0x7fff;
                else if (fg != 0u && fx != 0u) { e = (int)(fg + fx); mymax = max(mymax, e); }
            }
            raw[r] = (unsigned short)e;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mymax = max(mymax, __shfl_xor(mymax, o));
        if (lane == 0 && mymax > 0) atomicMax(&s_emax, mymax);
    }
    __syncthreads();
    const int emax = s_emax;
    if (tid < 16) { ctab[n_rows16 + tid] = 0; xtab[n_rows16 + tid] = 0; }
    auto pow2h = [](int d) -> unsigned short { return d <= 14 ? (unsigned short)((15 - d) << 10) : (d <= 24 ? (unsigned short)(1u << (24 - d)) : (unsigned short)0); };
    for (int r = tid; r < n_rows16; r += 256) {
        const int e = raw[r];
        unsigned short cg = 0, cx = 0;
        if (e == 0x7fff) { cg = 0x7e00; cx = 0x3c00; }
        else if (e != 0) {
            const int d = emax - e, dg = d >> 1;
            cg = pow2h(dg); cx = pow2h(d - dg);
        }
        ctab[r] = cg;
        xtab[r] = cx;
    }

    // ---- lanes -> source pieces (DMA) and -> fragment addresses (transposing reads) ---------------------------------------------------------
    const int gs = min(n0 / 32 + w, a.g_slices - 1), xs = min(k0 / 32 + w, a.x_slices - 1);
    const unsigned g_pitch = (unsigned)a.g_slices * 128u, x_pitch = (unsigned)a.x_slices * 128u;
    unsigned g_src, x_src;                        // byte offset of this lane's piece from the first row of an 8-row image
    {
        const int row = lane >> 3, piece = (lane & 7) ^ (((row >> 1) & 1) << 2);
        g_src = (unsigned)row * g_pitch + (unsigned)piece * 16u;
        x_src = (unsigned)row * x_pitch + (unsigned)piece * 16u;
    }
    const unsigned char *gbase = a.g_planes + (int64_t)gs * 128, *xbase = a.x_planes + (int64_t)xs * 128;
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    const int dbg = DBG ? a.dbg : 0;
    auto fetch = [&](int buf, int64_t row0) {     // this wave's slice of both operands: 2 x 2 images of 8 rows
        unsigned char *const dg = wf_smem + buf * WD_BUF + w * 2048, *const dx = dg + WD_OP;
        const unsigned char *gr = gbase + row0 * g_pitch, *xr = xbase + row0 * x_pitch;
        __builtin_amdgcn_global_load_lds((gptr_t)(gr + g_src), (lptr_t)dg, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(gr + 8 * (int64_t)g_pitch + g_src), (lptr_t)(dg + 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(xr + x_src), (lptr_t)dx, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(xr + 8 * (int64_t)x_pitch + x_src), (lptr_t)(dx + 1024), 16, 0, 0);
    };
    // fragment of 32 columns (K slice sl of the tile) x this lane's 8 rows: image lh of the slice; the lane points at row (lane & 15) >> 2 of the
    // read's four rows, column quad lane & 3 of column block (lane >> 4) & 1, plane p: piece 4 p + 2 cb + (quad >> 1), swapped halves on rows 2, 3
    const int t_row = (lane & 15) >> 2, t_cb = (lane >> 4) & 1, t_q = lane & 3;
    unsigned fr[2][2];                            // [plane][read] byte offset inside the slice's 2 KiB
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int r = 4 * g + t_row, piece = (4 * p + 2 * t_cb + (t_q >> 1)) ^ (((r >> 1) & 1) << 2);
            fr[p][g] = (unsigned)(lh * 1024 + r * 128 + piece * 16 + (t_q & 1) * 8);
        }
    auto frag = [&](const unsigned char *op, int sl, int p, wf_un4 c) -> wf_un4 {
        const unsigned char *q = op + sl * 2048;
        const wf_fp16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((wf_lds4_t)(q + fr[p][0]));
        const wf_fp16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((wf_lds4_t)(q + fr[p][1]));
        const wf_un2 u0 = __builtin_bit_cast(wf_un2, v0), u1 = __builtin_bit_cast(wf_un2, v1);
        auto pk_mul = [](unsigned v, unsigned k) { return __builtin_bit_cast(unsigned, __builtin_bit_cast(wf_h16x2, v) * __builtin_bit_cast(wf_h16x2, k)); };
        return wf_un4{pk_mul(u0.x, c.x), pk_mul(u0.y, c.y), pk_mul(u1.x, c.z), pk_mul(u1.y, c.w)};
    };

    wf_f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    __syncthreads();                              // (the raw exponents lived in the buffers)
    fetch(0, r_begin);
    fetch(1, r_begin + 16);
    int buf = 0;
#define WF_MF(x, y, c) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wf_h16x8, x), __builtin_bit_cast(wf_h16x8, y), c, 0, 0, 0)
    for (int64_t row0 = r_begin; row0 < r_end; row0 += 16) {
        // the lines of this step have landed (this wave's share -- all but the newest four requests; the barrier makes it everybody's) and nobody
        // reads the buffer of the step before any more: it takes the lines of the step after the next
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __syncthreads();
        const int far = buf >= 1 ? buf - 1 : 2;   // (buf + 2) % 3
        if (!(dbg & 4)) fetch(far, row0 + 32);    // (past the slab: rows nobody multiplies; past the operand: the scratch's zero rows)
        const unsigned char *og = wf_smem + buf * WD_BUF, *ox = og + WD_OP;
        const wf_un4 cs = *reinterpret_cast<const wf_un4 *>(ctab + (row0 - r_begin) + lh * 8);
        const wf_un4 ds = *reinterpret_cast<const wf_un4 *>(xtab + (row0 - r_begin) + lh * 8);
        wf_un4 fa[2][2], fb[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                fa[i][p] = frag(og, wm * 2 + i, p, cs);
                fb[i][p] = frag(ox, wn * 2 + i, p, ds);
            }
        if (!(dbg & 2))
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int pa_ = t == 2 ? 0 : (t == 0 ? 1 : 0), pb_ = t == 2 ? 0 : (t == 0 ? 0 : 1);      // lh, hl, hh
            WF_MF(fa[0][pa_], fb[0][pb_], acc[0][0]);
            WF_MF(fa[0][pa_], fb[1][pb_], acc[0][1]);
            WF_MF(fa[1][pa_], fb[0][pb_], acc[1][0]);
            WF_MF(fa[1][pa_], fb[1][pb_], acc[1][1]);
        }
        buf = buf == 2 ? 0 : buf + 1;
    }
#undef WF_MF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the last requests: its lines land in an LDS that is about to be released)
    const int ex = emax - 254;
    const int e1 = ex < -126 ? -126 : (ex > 127 ? 127 : ex), e2r = ex - e1, e2 = e2r < -126 ? -126 : (e2r > 127 ? 127 : e2r);
    const float f1 = __uint_as_float((unsigned)(e1 + 127) << 23), f2 = __uint_as_float((unsigned)(e2 + 127) << 23);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kcol = k0 + wn * 64 + j * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nrow = n0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (nrow < a.n_out && kcol < a.k_total && !(dbg & 1)) atomicAdd(a.gw + (int64_t)nrow * a.k_total + kcol, acc[i][j][r] * f1 * f2);
            }
        }
}

}  // namespace

}  // namespace gsn

using namespace gsn;

extern "C" int gsn_wgrad_f16x3_hip(int64_t m_rows, int64_t n_out, int64_t k_total, const void *g_scratch, const void *x_scratch, float *grad_w,
                                   void *stream) {
    if (n_out < 1 || k_total < 1 || !grad_w || (m_rows > 0 && (!g_scratch || !x_scratch)))
        return set_error(GSN_E_INVALID, "gsn_wgrad_f16x3_hip: bad arguments");
    if (m_rows <= 0) return GSN_OK;
    if (((reinterpret_cast<uintptr_t>(g_scratch) | reinterpret_cast<uintptr_t>(x_scratch)) & 15) != 0)
        return set_error(GSN_E_INVALID, "gsn_wgrad_f16x3_hip: the row scratches are 16-byte aligned (gsn_linear_f16x3_fwd_hip's row_scratch)");
    WfArgs a{};
    a.m_rows = m_rows; a.n_out = (int)n_out; a.k_total = (int)k_total; a.gw = grad_w;
    const int64_t m_pad = gsn_linear_f16x3_mpad(m_rows);                   // (gsn_linear_f16x3_scratch_bytes: [m_pad] inverse scales | [m_pad] rows of planes)
    a.g_slices = (int)(gsn_linear_f16x3_kpad(n_out) / 32); a.x_slices = (int)(gsn_linear_f16x3_kpad(k_total) / 32);
    if (m_rows * (int64_t)a.g_slices * 128 >= ((int64_t)1 << 40) || a.g_slices * 128 >= (1 << 24) || a.x_slices * 128 >= (1 << 24))
        return set_error(GSN_E_UNSUPPORTED, "gsn_wgrad_f16x3_hip: rows of 16 MiB and more are not supported");
    a.g_inv = reinterpret_cast<const float *>(g_scratch); a.x_inv = reinterpret_cast<const float *>(x_scratch);
    a.g_planes = reinterpret_cast<const unsigned char *>(a.g_inv + m_pad); a.x_planes = reinterpret_cast<const unsigned char *>(a.x_inv + m_pad);
    const int tn = (int)((n_out + WF_T - 1) / WF_T), tk = (int)((k_total + WF_T - 1) / WF_T);
    a.tn = tn; a.tk = tk;
    // slabs as gsn_wgrad_hip takes them (one round of ~256 workgroups for small and mid-size calls, 2 048 workgroups for large ones)
    static const int64_t wg_target = [] { const char *e = getenv("GSN_WGRAD16_WGS"); const int64_t v = e ? atoll(e) : 0; return v > 0 ? v : (int64_t)0; }();
    int64_t rows_per = gsn_wgrad_slab_rows(m_rows, (int64_t)tn * tk, wg_target);
    if (rows_per > WF_MAX_ROWS) rows_per = WF_MAX_ROWS;
    rows_per = (rows_per + 47) / 48 * 48;                                  // (the kernel's loop: three 16-row steps per trip)
    a.rows_per_wg = rows_per;
    const int64_t slabs = (m_rows + rows_per - 1) / rows_per;
    const int64_t slab_groups = (slabs + 7) / 8;
    if (slab_groups * tn * tk * 8 >= ((int64_t)1 << 31)) return set_error(GSN_E_UNSUPPORTED, "gsn_wgrad_f16x3_hip: too many workgroups");
    if (getenv("GSN_CHAIN_TRACE"))
        fprintf(stderr, "gsn wgrad: wgrad_f16x3_kernel M %lld N %d K %d slabs %lld x %lld rows\n", (long long)m_rows, (int)n_out, (int)k_total, (long long)slabs,
                (long long)rows_per);
    const dim3 grid((unsigned)(slab_groups * tn * tk * 8));
    const size_t lds = (size_t)4 * 4 * 2 * (64 * 16 + 128) + 16 + (size_t)(rows_per + 16) * 6;      // fragments | emax | the two scale tables, raw exponents
    static const int valu = [] { const char *e = getenv("GSN_WGRAD16_VALU"); const int v = e ? atoi(e) : 3; return v; }();
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    a.dbg = getenv("GSN_WGRAD16_DBG") ? atoi(getenv("GSN_WGRAD16_DBG")) : 0;
    // GSN_WGRAD16_DMA=1: the LDS-DMA + transposing-read kernel (read per call: tests switch it).  Measured on one box at 105 083 x 300 x 600:
    // 240-248 us against 218-227 us for the register-staged kernel -- fewer vector and LDS instructions (SQ_INSTS_VALU 4.5e7 -> 2.9e7,
    // SQ_LDS_IDX_ACTIVE 2.7e7 -> 1.6e7), three workgroups per CU instead of two, the matrix pipe mostly hidden (no products: -35 us), but 74 us of
    // waiting for lines that the staged kernel hides behind its own vector work; both move the same 1.5 GB from L2 to the CUs, which bounds
    // either at ~150 us.  Not the default.
    const char *dma_env = getenv("GSN_WGRAD16_DMA");
    const bool dma = dma_env && dma_env[0] == '1';
    if (dma) {
        const size_t lds_dma = (size_t)3 * 16384 + 16 + (size_t)(rows_per + 16) * 4;      // three buffers | emax | the two scale tables
        if (a.dbg) hipLaunchKernelGGL((wgrad_f16x3_dma_kernel<true>), grid, dim3(256), lds_dma, st, a);
        else hipLaunchKernelGGL((wgrad_f16x3_dma_kernel<false>), grid, dim3(256), lds_dma, st, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return set_error(GSN_E_HIP, "wgrad_f16x3_dma_kernel: %s", hipGetErrorString(e));
        return GSN_OK;
    }
    a.dbg = getenv("GSN_WGRAD16_DBG") ? atoi(getenv("GSN_WGRAD16_DBG")) : 0;
    a.prof = nullptr;
    if ((a.dbg & 16) && hipMalloc(reinterpret_cast<void **>(&a.prof), 128) != hipSuccess) a.prof = nullptr;
    if (a.dbg) {
        hipLaunchKernelGGL((wgrad_f16x3_kernel<3, true>), grid, dim3(256), lds, st, a);
        if (a.prof) {
            unsigned long long h[16];
            (void)hipStreamSynchronize(st);
            (void)hipMemcpy(h, a.prof, 128, hipMemcpyDeviceToHost);
            (void)hipFree(a.prof);
            fprintf(stderr, "wgrad16prof: workgroup 0 wave 0: prologue %llu | loop %llu (%llu steps: %.0f per step, of it waiting at the barrier %.0f) | epilogue %llu cycles (100 MHz clock)\n",
                    h[0], h[1], h[4], (double)h[1] / (double)(h[4] ? h[4] : 1), (double)h[2] / (double)(h[4] ? h[4] : 1), h[3]);
            const double n = (double)(h[4] ? h[4] : 1);
            fprintf(stderr, "wgrad16prof: per step: issue of the far loads %.0f | wait for the rows to stage %.0f | fragment reads %.0f | staging %.0f | products (issue) %.0f\n",
                    h[5] / n, h[6] / n, h[7] / n, h[8] / n, h[9] / n);
        }
    }
    else if (valu == 2) hipLaunchKernelGGL((wgrad_f16x3_kernel<2, false>), grid, dim3(256), lds, st, a);
    else if (valu == 4) hipLaunchKernelGGL((wgrad_f16x3_kernel<4, false>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((wgrad_f16x3_kernel<3, false>), grid, dim3(256), lds, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "wgrad_f16x3_kernel: %s", hipGetErrorString(e));
    return GSN_OK;
}
