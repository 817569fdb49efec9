// Any-width dense stage on the fp16x3 scheme (the matrix arithmetic of layer_fused.hip) for DIRECT rows:
//
//     out[m, :] = act( bn( cat(blocks[0][m], blocks[1][m], ..) W^T + b ) )        (models_misc.py:52-58; node-level stages)
//
// -- the node product of a wide `general` layer (layers._split_edge_stage), its K = 260 node stage, the d = 300 ogb stages,
// jk projections: everything linear_fwd_bf16_kernel (six bf16 plane products, both operands split per K slice by every
// workgroup) handled at 85-100 TF/s fp32-equivalent.  Differences:
//   * two fp16 planes per operand after an exact power-of-two scaling, three plane products per fp32 product: half the matrix
//     work of bf16x6 at the same error (scripts/micro/bf16x6_check.hip);
//   * BOTH operands reach the matrix kernel as fp16 planes: the weights are split once per weight version
//     (gsn_linear_f16x3_prepare_hip: per output column a power-of-two scale from its largest entry), the rows once per call by a
//     pre-pass (lin16_split_rows_kernel: per row a scale from its largest magnitude over ALL its columns, so one accumulator
//     serves the whole K loop) -- not per K slice by every column tile; the epilogue multiplies the row's and the column's
//     inverse scales back in (exact), then bias / BatchNorm / activation as in linear.hip;
//   * plane layout in memory [row][K slice][high | low][32 halfs]: the 128 bytes one (row, 32-wide K slice) needs are ONE cache
//     line.  scripts/micro/l1_stream.hip: a CU pulls 51 bytes per clock through its vector L1 when every wave instruction reads
//     whole lines, 16 when it reads 64-byte halves of 16 lines (what [plane][row][K] gives) -- and that rate, not the matrix
//     pipe, is what bounds a kernel of this shape.
// Rows with an Inf / NaN come out NaN in all columns (documented deviation of the split-operand kernels).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "gsn_internal.h"

namespace gsn {

namespace {

constexpr int L_BK = 32;                           // K slice
constexpr int L_LINE = 4 * L_BK;                   // bytes of one (row, slice): 32 high halfs, 32 low halfs
constexpr int L_MAXB = 5;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float fl2 __attribute__((ext_vector_type(2)));
typedef unsigned un4 __attribute__((ext_vector_type(4)));
typedef unsigned un2 __attribute__((ext_vector_type(2)));

struct L16Args {
    int64_t m_rows;
    int n_blocks;
    // (named fields, not arrays: a per-lane choice among array elements of the argument block is compiled into a vector LOAD from
    //  the argument segment, and the wait for that pointer drains every prefetch in flight)
    const float *b0, *b1, *b2, *b3, *b4;
    int w0, w1, w2, w3, w4;        // widths (0 past n_blocks)
    const unsigned char *wplanes;  // [n_out][k_pad / 32][2][32] halfs
    const float *colinv;           // [n_out] inverse column scales
    const float *bias, *bn_mean, *bn_scale, *bn_shift;
    int k_total, k_pad, n_out, act;
    float *out;
    double *stats;                 // train-mode stage: [2][n_out] fp64 column sums / sums of squares of the rows written (added to), or null
    float *rowinv;                 // [m_pad] inverse row scales (0 past m_rows)
    unsigned char *aplanes;        // [m_rows][k_pad / 32][2][32] halfs
    int64_t m_pad;
    int col_tiles, groups;         // column tiles of the output, row groups per XCD
    int dbg;                       // diagnostic build: 1 no stores, 2 no products, 4 no loads, 8 no LDS writes
    unsigned long long *prof;      // diagnostic build: cycles per phase of workgroup 0, wave 0
};

__device__ __forceinline__ void l16_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// power-of-two scale that puts a magnitude with these sign-less float bits into [2^14, 2^15), and its inverse (layer_fused.hip)
__device__ __forceinline__ void l16_scale(unsigned maxbits, float &scale, float &inv) {
    int e = (int)(maxbits >> 23);
    e = e < 15 ? 15 : (e > 254 ? 254 : e);
    scale = __uint_as_float((unsigned)(268 - e) << 23);
    inv = __uint_as_float((unsigned)(e - 14) << 23);
}

__device__ __forceinline__ void l16_split2(fl2 v, unsigned &hi, unsigned &lo) {
    const h16x2 h = __builtin_convertvector(v, h16x2);
    const fl2 r = v - __builtin_convertvector(h, fl2);
    const h16x2 l = __builtin_convertvector(r, h16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}

__device__ __forceinline__ float l16_act(float y, int act) {
    switch (act) {
        case 1: return y > 0.f ? y : 0.f;
        case 2: return y > 0.f ? y : expm1f(y);
        case 3: return tanhf(y);
        default: return y;
    }
}

__device__ __noinline__ float l16_act_slow(float y, int act) { return l16_act(y, act); }

// concatenated column kg -> (block base + column, block width); clamped to a valid address past K
struct L16Col {
    const float *base;
    int bw;
};
__device__ __forceinline__ L16Col l16_col(const L16Args &a, int kg) {
    const int p1 = a.w0, p2 = p1 + a.w1, p3 = p2 + a.w2, p4 = p3 + a.w3;        // first column of blocks 1 .. 4 (uniform)
    if (kg >= a.k_total) kg = 0;
    L16Col m;
    m.base = a.b0 + kg; m.bw = a.w0;
    if (kg >= p1) { m.base = a.b1 + (kg - p1); m.bw = a.w1; }
    if (kg >= p2) { m.base = a.b2 + (kg - p2); m.bw = a.w2; }
    if (kg >= p3) { m.base = a.b3 + (kg - p3); m.bw = a.w3; }
    if (kg >= p4) { m.base = a.b4 + (kg - p4); m.bw = a.w4; }
    return m;
}

// ---- weights: one wave per output column -----------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void lin16_prepare_kernel(const float *__restrict__ W, int n_out, int k_total, int k_pad, _Float16 *planes,
                                                          float *colinv, int64_t w_rs, int64_t w_cs) {
    const int j = blockIdx.x, lane = threadIdx.x;
    const float *w = W + (int64_t)j * w_rs;      // (element k of output column j at w[k * w_cs]: row-major, or a transposed view)
    unsigned m = 0;
    for (int k = lane; k < k_total; k += 64) m = max(m, __float_as_uint(w[k * w_cs]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    float s, inv;
    l16_scale(m, s, inv);
    if (m >= 0x7f800000u) inv = __uint_as_float(0x7fc00000u);        // a non-finite weight: the whole output column is NaN
    _Float16 *row = planes + (int64_t)j * k_pad * 2;
    for (int k = lane; k < k_pad; k += 64) {
        const float v = k < k_total ? w[k * w_cs] * s : 0.f;
        const _Float16 h = (_Float16)v;
        _Float16 *line = row + (k >> 5) * 64 + (k & 31);
        line[0] = h;
        line[32] = (_Float16)(v - (float)h);
    }
    if (lane == 0) colinv[j] = inv;
}

// ---- rows split ONCE per call: scale + two fp16 planes -----------------------------------------------------------------------------
// 8 lanes per row, float4 chunks over the concatenated blocks.  The chunks of a row up to K = 320 stay in registers between the
// pass that finds the row's largest magnitude and the pass that writes its planes; wider rows are read again (from L2).
constexpr int L_RC = 10;
__global__ __launch_bounds__(256) void lin16_split_rows_kernel(L16Args a) {
    const int64_t row = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);     // < m_pad
    const int q8 = threadIdx.x & 7;
    const bool on = row < a.m_rows;
    const int64_t rr = on ? row : a.m_rows - 1;
    const int nch = a.k_total >> 2, nchp = a.k_pad >> 2;
    float4 v[L_RC];
    unsigned m = 0;
    auto amax = [&](const float4 &x) {
        m = max(max(m, __float_as_uint(x.x) & 0x7fffffffu), __float_as_uint(x.y) & 0x7fffffffu);
        m = max(max(m, __float_as_uint(x.z) & 0x7fffffffu), __float_as_uint(x.w) & 0x7fffffffu);
    };
#pragma unroll
    for (int i = 0; i < L_RC; ++i) {
        const int c = q8 + 8 * i;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < nch) {
            const L16Col cm = l16_col(a, 4 * c);
            v[i] = *reinterpret_cast<const float4 *>(cm.base + rr * cm.bw);
            amax(v[i]);
        }
    }
    for (int c = q8 + 8 * L_RC; c < nch; c += 8) {
        const L16Col cm = l16_col(a, 4 * c);
        amax(*reinterpret_cast<const float4 *>(cm.base + rr * cm.bw));
    }
    m = max(m, (unsigned)__shfl_xor((int)m, 1));
    m = max(m, (unsigned)__shfl_xor((int)m, 2));
    m = max(m, (unsigned)__shfl_xor((int)m, 4));
    float s, inv;
    l16_scale(m, s, inv);
    if (m >= 0x7f800000u) inv = __uint_as_float(0x7fc00000u);            // Inf / NaN in the row: its outputs become NaN
    if (q8 == 0) a.rowinv[row] = on ? inv : 0.f;
    // (rows past m_rows, up to m_pad: ZERO planes -- gsn_wgrad_f16x3_hip walks 16-row steps and requests rows ahead without a bound check)
    unsigned char *prow = a.aplanes + row * a.k_pad * 4;                 // (4 bytes per column: a high and a low half)
    // chunk c = columns 4c .. 4c + 3 of slice c >> 3.  Lane pairs trade halves so that every lane stores 16 bytes: the even lane
    // the high halfs of both chunks, the odd lane the low halfs -- the 8 lanes of a row write one whole line with one instruction
    const bool even = (q8 & 1) == 0;
    auto put = [&](int c, const float4 &xin) {
        const float4 x = on ? xin : make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned h0, l0, h1, l1;
        l16_split2(fl2{x.x * s, x.y * s}, h0, l0);
        l16_split2(fl2{x.z * s, x.w * s}, h1, l1);
        const unsigned r0 = (unsigned)__shfl_xor((int)(even ? l0 : h0), 1), r1 = (unsigned)__shfl_xor((int)(even ? l1 : h1), 1);
        unsigned char *line = prow + (c >> 3) * L_LINE + (even ? 0 : 64) + ((c & 7) >> 1) * 16;
        *reinterpret_cast<un4 *>(line) = even ? un4{h0, h1, r0, r1} : un4{r0, r1, l0, l1};
    };
#pragma unroll
    for (int i = 0; i < L_RC; ++i) {
        const int c = q8 + 8 * i;
        if (c < nchp) put(c, v[i]);                                        // (zero past K: the padding columns)
    }
    for (int c = q8 + 8 * L_RC; c < nchp; c += 8) {
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < nch) {
            const L16Col cm = l16_col(a, 4 * c);
            x = *reinterpret_cast<const float4 *>(cm.base + rr * cm.bw);
        }
        put(c, x);
    }
}

// r05, rows given as ONE block (every chunk of a row is its base + a constant: no per-chunk block selection, no per-chunk address registers):
// grid-stride over the 32-row tiles with the NEXT tile's chunks requested before this tile is split (two named register sets): a
// workgroup that lived for one tile spent a memory round trip with nothing else to do, and 3 284 one-tile workgroups over 2 048 slots left the
// second round 60 % full.  GSN_L16_SPLIT_WGS (default 1024; 0: one tile per workgroup, the r02 launch).
__global__ __launch_bounds__(256) void lin16_split_rows1_kernel(L16Args a) {
    const int q8 = threadIdx.x & 7;
    const int nch = a.k_total >> 2, nchp = a.k_pad >> 2;
    const int64_t tiles = a.m_pad / 32;
    auto load = [&](int64_t tile, float4 (&v)[L_RC]) {
        const int64_t row = tile * 32 + (threadIdx.x >> 3);
        const int64_t rr = row < a.m_rows ? row : a.m_rows - 1;
        const float4 *rp = reinterpret_cast<const float4 *>(a.b0 + rr * a.w0);
#pragma unroll
        for (int i = 0; i < L_RC; ++i) {
            const int c = q8 + 8 * i;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < nch) v[i] = rp[c];
        }
    };
    auto process = [&](int64_t tile, const float4 (&v)[L_RC]) {
        const int64_t row = tile * 32 + (threadIdx.x >> 3);     // < m_pad
        const bool on = row < a.m_rows;
        const int64_t rr = on ? row : a.m_rows - 1;
        unsigned m = 0;
        auto amax = [&](const float4 &x) {
            m = max(max(m, __float_as_uint(x.x) & 0x7fffffffu), __float_as_uint(x.y) & 0x7fffffffu);
            m = max(max(m, __float_as_uint(x.z) & 0x7fffffffu), __float_as_uint(x.w) & 0x7fffffffu);
        };
#pragma unroll
        for (int i = 0; i < L_RC; ++i) amax(v[i]);              // (chunks past K hold zeros)
        const float4 *rp = reinterpret_cast<const float4 *>(a.b0 + rr * a.w0);
        for (int c = q8 + 8 * L_RC; c < nch; c += 8) amax(rp[c]);
        m = max(m, (unsigned)__shfl_xor((int)m, 1));
        m = max(m, (unsigned)__shfl_xor((int)m, 2));
        m = max(m, (unsigned)__shfl_xor((int)m, 4));
        float s, inv;
        l16_scale(m, s, inv);
        if (m >= 0x7f800000u) inv = __uint_as_float(0x7fc00000u);            // Inf / NaN in the row: its outputs become NaN
        if (q8 == 0) a.rowinv[row] = on ? inv : 0.f;
        // (rows past m_rows, up to m_pad, get ZERO planes: gsn_wgrad_f16x3_hip requests rows ahead without a bound check)
        unsigned char *prow = a.aplanes + row * a.k_pad * 4;                 // (4 bytes per column: a high and a low half)
        // chunk c = columns 4c .. 4c + 3 of slice c >> 3.  Lane pairs trade halves so that every lane stores 16 bytes: the even lane
        // the high halfs of both chunks, the odd lane the low halfs -- the 8 lanes of a row write one whole line with one instruction
        const bool even = (q8 & 1) == 0;
        auto put = [&](int c, const float4 &xin) {
            const float4 x = on ? xin : make_float4(0.f, 0.f, 0.f, 0.f);
            unsigned h0, l0, h1, l1;
            l16_split2(fl2{x.x * s, x.y * s}, h0, l0);
            l16_split2(fl2{x.z * s, x.w * s}, h1, l1);
            const unsigned r0 = (unsigned)__shfl_xor((int)(even ? l0 : h0), 1), r1 = (unsigned)__shfl_xor((int)(even ? l1 : h1), 1);
            unsigned char *line = prow + (c >> 3) * L_LINE + (even ? 0 : 64) + ((c & 7) >> 1) * 16;
            *reinterpret_cast<un4 *>(line) = even ? un4{h0, h1, r0, r1} : un4{r0, r1, l0, l1};
        };
#pragma unroll
        for (int i = 0; i < L_RC; ++i) {
            const int c = q8 + 8 * i;
            if (c < nchp) put(c, v[i]);                                        // (zero past K: the padding columns)
        }
        for (int c = q8 + 8 * L_RC; c < nchp; c += 8) {
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < nch) x = rp[c];
            put(c, x);
        }
    };
    float4 vc[L_RC], vn[L_RC];
    const int64_t step = gridDim.x;
    int64_t t = blockIdx.x;
    if (t >= tiles) return;
    load(t, vc);
    for (; t < tiles; t += step) {
        const bool more = t + step < tiles;
        if (more) load(t + step, vn);
        process(t, vc);
#pragma unroll
        for (int i = 0; i < L_RC; ++i) vc[i] = vn[i];
    }
}

// ---- the product on pre-split rows ---------------------------------------------------------------------------------------------------
// A plain fp16 matrix kernel with three plane products per k-step: both operands arrive as fp16 planes and are staged with 16-byte
// copies (no arithmetic between the load and LDS).  2 WM waves on a (64 WM) x (64 NJ) output tile, wave w = rows 64 (w >> 1)..,
// columns 32 NJ (w & 1)..; the lines of two K slices in LDS (row pitch 144 bytes = 64 high + 64 low + 16: 16-byte fragment reads
// and 128-byte row writes are both conflict-free), the next TWO slices travel through two named register sets while the
// current one is multiplied; ONE barrier per slice.  Instantiated as <2, 2>: 128 x 128 tiles, 4 waves, two workgroups per CU whose staging and
// matrix phases interleave (256 x 256 tiles with 8 waves, one per CU, leave no registers for the second set of loads in flight
// and measured 3-5 % slower).
// Workgroup -> (XCD, column tile, row group): the column tiles of one row tile run on the same XCD at the same time, its rows
// leave HBM once.  The output tile leaves in 16-byte stores through a buffer resource per tile (rows past M and columns past N
// are dropped by the range check).
constexpr int P_PITCH = L_LINE + 16;               // bytes per row in LDS
template <int WM, int NJ, bool PROF, bool VEC>
__global__ __launch_bounds__(128 * WM) __attribute__((amdgpu_waves_per_eu(2, 2))) void linear_f16x3_planes_kernel(L16Args a) {
    const int dbg = PROF ? a.dbg : 0;
    constexpr int P_BM = 64 * WM, BN = 64 * NJ;                            // output rows / columns per tile
    constexpr int NT = 128 * WM;                                           // threads
    constexpr int RP = 16 * WM;                                            // rows one staging pass of the workgroup covers
    constexpr int NWP = BN / RP;                                           // passes over the weight rows
    constexpr int A_RG = P_BM * P_PITCH, W_RG = BN * P_PITCH, BUF = A_RG + W_RG;
    extern __shared__ __attribute__((aligned(16))) unsigned l16_lds[];
    unsigned char *const lds = reinterpret_cast<unsigned char *>(l16_lds);
    auto a_rows = [&](int buf) { return lds + buf * BUF; };
    auto w_rows = [&](int buf) { return lds + buf * BUF + A_RG; };
    // inverse row scales of a tile: two slots in the padding behind row r of buffer 0 (bytes 128 .. 135 of the row)
    auto rtab = [&](int slot, int r) -> float & { return *reinterpret_cast<float *>(lds + r * P_PITCH + L_LINE + 4 * slot); };
    auto clk = [&]() -> unsigned long long {
        if (!PROF) return 0ull;
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long v = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
        return v;
    };
    unsigned long long pq[8] = {0, 0, 0, 0, 0, 0, 0, 0};

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform, and the compiler should know)
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    // staging: 8 lanes = the 128-byte line of one (row, slice); a wave instruction moves 8 whole lines.  Thread -> piece sp of rows
    // (output columns) sr, sr + RP, ..
    const int sr = 8 * wave + (lane >> 3), sp = lane & 7;
    const int xcd = blockIdx.x & 7, slot_in_xcd = blockIdx.x >> 3;
    const int ct = slot_in_xcd % a.col_tiles, grp = slot_in_xcd / a.col_tiles;
    const int n0 = ct * BN;
    const int64_t n_tiles = (a.m_rows + P_BM - 1) / P_BM;
    const int64_t tile0 = (int64_t)xcd * a.groups + grp, tstride = 8 * (int64_t)a.groups;
    const int64_t n_mine = tile0 < n_tiles ? (n_tiles - tile0 + tstride - 1) / tstride : 0;
    const int n_slices = a.k_pad / L_BK;
    const int row_bytes = n_slices * L_LINE;                              // one row of planes in memory

    // per-column epilogue constants and per-lane store offsets live in LDS behind the slice buffers: held in registers across the K
    // loop they are spilled, and a spill reload in the epilogue waits for every store issued before it
    float *const ecol = reinterpret_cast<float *>(lds + 2 * BUF);         // [BN] scale | [BN] constant
    int *const vtab = reinterpret_cast<int *>(lds + 2 * BUF + 8 * BN);    // [2][NJ][64] byte offsets of the 16-byte stores | [2][NJ][32] of the 4-byte ones
    for (int cidx = tid; cidx < BN; cidx += NT) {
        const int col = n0 + cidx;
        const bool cok = col < a.n_out;
        const float e_bias = (cok && a.bias) ? a.bias[col] : 0.f;
        float es = cok ? a.colinv[col] : 0.f, ec = e_bias;
        if (cok && a.bn_scale) { ec = (e_bias - a.bn_mean[col]) * a.bn_scale[col] + a.bn_shift[col]; es *= a.bn_scale[col]; }
        ecol[cidx] = es; ecol[BN + cidx] = ec;
    }
    const int rstride = a.n_out * 4;                                      // bytes per output row
    for (int q = tid; q < 2 * NJ * 64; q += NT) {                         // piece read k of lane l in a wave with column half w: (row idx / (8 NJ) of the piece, 4 columns)
        const int l = q & 63, k = (q >> 6) % NJ, w = q / (64 * NJ);
        const int idx = 64 * k + l, c4 = idx % (8 * NJ), col = n0 + w * 32 * NJ + 4 * c4;
        vtab[q] = col < a.n_out ? ((idx / (8 * NJ)) - (64 * k) / (8 * NJ)) * rstride + col * 4 : 0x7f000000;
    }
    for (int q = tid; q < 2 * NJ * 32; q += NT) {                         // 4-byte stores: column j of lane li; past the tile's extent when the column does not exist
        const int l = q & 31, j = (q >> 5) % NJ, w = q / (32 * NJ);
        const int col = n0 + w * 32 * NJ + 32 * j + l;
        vtab[2 * NJ * 64 + q] = col < a.n_out ? col * 4 : 0x7f000000;
    }
    unsigned wofs[NWP];                                                     // weight rows of this workgroup's column tile (fixed): byte offsets (the planes of W stay below 4 GiB)
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
        int j = n0 + sr + RP * i;
        j = j < a.n_out ? j : 0;                                            // (its output column is never emitted)
        wofs[i] = (unsigned)j * (unsigned)row_bytes + 16u * sp;
    }
    auto rt_fetch = [&](int64_t row0) -> float {                          // (every thread loads: rowinv is padded to whole tiles)
        const int64_t r = row0 + (tid & (P_BM - 1));
        return a.rowinv[r < a.m_pad ? r : a.m_pad - 1];
    };

    // TWO named register sets: the loads of slice c + 2 are issued while slice c is multiplied (one slice of products does not
    // cover a load's latency when the other workgroup of the CU keeps the memory pipeline busy).  Every slice issues the same
    // number of loads, so the waits are counted.
    un4 pA_a[4], pW_a[NWP], pA_b[4], pW_b[NWP];
    auto fetch = [&](un4 (&pA)[4], un4 (&pW)[NWP], const unsigned char *abase, const unsigned *aofs, int c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) pA[i] = *reinterpret_cast<const un4 *>(abase + aofs[i] + c * L_LINE);
#pragma unroll
        for (int i = 0; i < NWP; ++i) pW[i] = *reinterpret_cast<const un4 *>(a.wplanes + wofs[i] + c * L_LINE);
    };
    auto stage = [&](const un4 (&pA)[4], const un4 (&pW)[NWP], int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<un4 *>(a_rows(buf) + (sr + RP * i) * P_PITCH + 16 * sp) = pA[i];
#pragma unroll
        for (int i = 0; i < NWP; ++i) *reinterpret_cast<un4 *>(w_rows(buf) + (sr + RP * i) * P_PITCH + 16 * sp) = pW[i];
    };
    f32x16 acc[2][NJ];
#define L16_MF(acc, x, y) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, x), __builtin_bit_cast(h16x8, y), acc, 0, 0, 0)
    auto products = [&](int buf) {
        const unsigned char *ap = a_rows(buf) + (wm * 64 + li) * P_PITCH + 16 * lh, *bp = w_rows(buf) + (wn * 32 * NJ + li) * P_PITCH + 16 * lh;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            un4 ah[2], al[2], bh[NJ], bl[NJ];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const un4 *>(ap + i * (32 * P_PITCH) + 32 * s);
                al[i] = *reinterpret_cast<const un4 *>(ap + i * (32 * P_PITCH) + 32 * s + 64);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                bh[j] = *reinterpret_cast<const un4 *>(bp + j * (32 * P_PITCH) + 32 * s);
                bl[j] = *reinterpret_cast<const un4 *>(bp + j * (32 * P_PITCH) + 32 * s + 64);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) L16_MF(acc[i][j], al[i], bh[j]);                      // small terms first
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) L16_MF(acc[i][j], ah[i], bl[j]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) L16_MF(acc[i][j], ah[i], bh[j]);
        }
    };
    // rows of a tile: byte offsets from the tile's first row (rows past the end read the last row; never emitted)
    auto row_ofs = [&](int64_t row0, unsigned *aofs) {
        const int64_t left = a.m_rows - row0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r = sr + RP * i;
            r = r < left ? r : (int)(left > 0 ? left - 1 : 0);
            aofs[i] = (unsigned)r * (unsigned)row_bytes + 16u * sp;
        }
    };
    auto tile_base = [&](int64_t row0) { return a.aplanes + (row0 < a.m_rows ? row0 : 0) * row_bytes; };
    unsigned aofs[4], aofs_next[4];
    const unsigned char *abase, *abase_next;
    int64_t ti = 0;                                                        // this workgroup's tile counter
    int c = 0, slot = 0;                                                   // slice inside the tile; row-scale slot of the tile
    float rt_next;
    // what a tile needs besides its slices: the rows of the NEXT tile (its first two slices are fetched during this tile's last
    // two) and that tile's inverse row scales; when there is no next tile the current one is read again (every slice issues the
    // same loads)
    auto begin_tile = [&]() {
        const int64_t row0 = (tile0 + ti * tstride) * P_BM, row_next = ti + 1 < n_mine ? row0 + tstride * P_BM : row0;
        rt_next = rt_fetch(row_next);
        row_ofs(row_next, aofs_next);
        abase_next = tile_base(row_next);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    // One slice: stage its register set, barrier, refill the set with the slice two ahead, multiply; after a tile's last slice its
    // epilogue.  Slices alternate between the two register sets / LDS buffers across tile boundaries too (a tile may have an odd
    // number of slices), so the epilogue is compiled behind both halves of the loop below -- the sets keep compile-time names.
    auto step = [&](un4 (&pA)[4], un4 (&pW)[NWP], const int bufx) -> bool {
        const unsigned long long q0 = clk();
        if (!(dbg & 8)) stage(pA, pW, bufx);
        if (c + 1 >= n_slices && tid < P_BM) rtab(slot ^ 1, tid) = rt_next;           // (read by the next tile's epilogue)
        const unsigned long long q1 = clk();
        l16_barrier();
        const unsigned long long q2 = clk();
        const int c2 = c + 2;
        const bool wrap = c2 >= n_slices;
        if (!(dbg & 4)) fetch(pA, pW, wrap ? abase_next : abase, wrap ? aofs_next : aofs, wrap ? c2 - n_slices : c2);
        const unsigned long long q3 = clk();
        if (!(dbg & 2)) products(bufx);
        if (PROF) { const unsigned long long q4 = clk(); pq[0] += q3 - q2; pq[1] += q4 - q3; pq[2] += q1 - q0; pq[3] += q2 - q1; pq[5] += 1; }
        if (++c < n_slices) return false;
        const unsigned long long qe0 = clk();
        const int64_t row0 = (tile0 + ti * tstride) * P_BM;
        // epilogue.  C layout of a 32x32 tile: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).  A CU retires
        // about one store instruction per ~65 cycles whatever its width (measured: 4-byte stores of a 256 x 256 tile took longer
        // than its whole K loop), so the tile leaves in 16-byte stores: every wave turns 8 rows x 32 NJ columns at a time through a
        // private piece of the plane buffer that was multiplied last (free until the next tile's second slice is staged).
        const int64_t rows_left = a.m_rows - row0;
        const int nrows = rows_left < P_BM ? (int)rows_left : P_BM;
        const __amdgpu_buffer_rsrc_t orow = __builtin_amdgcn_make_buffer_rsrc(a.out + row0 * a.n_out, 0, nrows * rstride, 0x00020000);
        const float act_lo = a.act == 1 ? 0.f : -INFINITY;
        constexpr int SP = 32 * NJ * 4 + 16;                               // bytes per row of a wave's 8-row piece
        unsigned char *const scr = w_rows(bufx ^ 1) + wave * (8 * SP);
        float es[NJ], ec[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) { es[j] = ecol[wn * 32 * NJ + 32 * j + li]; ec[j] = ecol[BN + wn * 32 * NJ + 32 * j + li]; }
        auto emit = [&](auto simple) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int rl = wm * 64 + i * 32 + 8 * gq;            // first tile row of this piece
                    const float ivv[4] = {rtab(slot, rl + 4 * lh), rtab(slot, rl + 4 * lh + 1), rtab(slot, rl + 4 * lh + 2), rtab(slot, rl + 4 * lh + 3)};
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            float y = fmaf(acc[i][j][4 * gq + r], ivv[r] * es[j], ec[j]);
                            if (decltype(simple)::value) y = y < act_lo ? act_lo : y;           // (a NaN stays a NaN)
                            else y = l16_act_slow(y, a.act);
                            *reinterpret_cast<float *>(scr + (r + 4 * lh) * SP + (li + 32 * j) * 4) = y;
                        }
                    if ((dbg & 1) && acc[i][0][4 * gq] != 12345.f) continue;
#pragma unroll
                    for (int k = 0; k < NJ; ++k) {
                        const int idx = 64 * k + lane, row = idx / (8 * NJ), c4 = idx % (8 * NJ);
                        const un4 v = *reinterpret_cast<const un4 *>(scr + row * SP + 16 * c4);
                        // (the row offset goes into the vector offset, not into soffset: with an SGPR soffset the compiler treats a
                        //  16-byte buffer store as free of the "store data overwritten by the next VALU write" hazard -- on this part it
                        //  is not: v_mul wrote the third dword's register right behind the store and a few lanes stored the new value)
                        __builtin_amdgcn_raw_buffer_store_b128(v, orow, vtab[(wn * NJ + k) * 64 + lane] + (rl + (64 * k) / (8 * NJ)) * rstride, 0, 0);
                        asm volatile("s_nop 1" ::: "memory");
                    }
                }
            }
        };
        if (VEC) {
            if (a.act <= 1) emit(std::true_type{});
            else emit(std::false_type{});
        } else {
#pragma unroll 1
            for (int i = 0; i < 2; ++i) {
                const int rl = wm * 64 + i * 32 + 4 * lh;                // first of this lane's rows inside the tile
#pragma unroll 1
                for (int gq = 0; gq < 4; ++gq) {
                    const float ivv[4] = {rtab(slot, rl + 8 * gq), rtab(slot, rl + 8 * gq + 1), rtab(slot, rl + 8 * gq + 2), rtab(slot, rl + 8 * gq + 3)};
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            const float y = l16_act_slow(fmaf(i ? acc[1][j][4 * gq + r] : acc[0][j][4 * gq + r], ivv[r] * es[j], ec[j]), a.act);
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), orow, vtab[2 * NJ * 64 + (wn * NJ + j) * 32 + li] + 4 * lh * rstride, (wm * 64 + i * 32 + 8 * gq + r) * rstride, 0);
                        }
                }
            }
        }
        l16_barrier();                                                     // (the pieces are overwritten by the next slice's staging)
        if (PROF) { pq[4] += clk() - qe0; pq[6] += 1; }
        c = 0; slot ^= 1; ++ti;
#pragma unroll
        for (int i = 0; i < 4; ++i) aofs[i] = aofs_next[i];
        abase = abase_next;
        if (ti >= n_mine) return true;
        begin_tile();
        return false;
    };
    if (n_mine > 0) {                                                      // (K is padded to two slices at least)
        row_ofs(tile0 * P_BM, aofs);
        abase = tile_base(tile0 * P_BM);
        fetch(pA_a, pW_a, abase, aofs, 0);
        fetch(pA_b, pW_b, abase, aofs, 1);
        if (tid < P_BM) rtab(0, tid) = rt_fetch(tile0 * P_BM);
        begin_tile();
        for (;;) {
            if (step(pA_a, pW_a, 0)) break;
            if (step(pA_b, pW_b, 1)) break;
        }
    }
    if (PROF && a.prof && blockIdx.x == 0 && tid == 0)
        for (int q = 0; q < 8; ++q) a.prof[q] = pq[q];
#undef L16_MF
}


// ---- the same product with LDS-DMA staging -------------------------------------------------------------------------------------------
// global_load_lds_dwordx4 moves a slice's lines straight into LDS (destination: a wave-uniform base + 16 bytes per lane, so the LDS
// image of a wave instruction is its 8 lines back to back: row pitch 128 bytes, no padding): no staging registers, no ds_write pass,
// no wait between a load and its LDS write.  Bank conflicts of the fragment reads are avoided by a swizzle instead of a pad: the
// 16-byte piece p of row r sits at slot p ^ ((r >> 1) & 7) -- applied on the SOURCE side (lane -> which piece it fetches) and on
// the fragment reads.  Two slice buffers; slice c + 1 is in flight while slice c is multiplied; one barrier per slice.
constexpr int D_BM = 128, D_NT = 256;
// NJ: 32-column blocks per wave (the tile is 128 rows x 64 NJ columns, waves 2 x 2).  NJ = 2: 128 x 128, two workgroups per CU.  NJ = 5 (r06):
// 128 x 320 for the d = 300 ogb stages -- n_out = 300 is ONE column tile instead of three (384 columns of matrix work -> 320, the rows' planes
// read once instead of three times), n_out = 600 two instead of five; 160 accumulator registers per wave and 112 KiB of slice buffers: one
// workgroup per CU, which LDS-DMA staging (no vector work between a load and its use) leaves able to keep the matrix pipe busy.
// STATS (a train-mode BatchNorm stage, models_misc.py:52-58 with bn in train mode): the rows written are the pre-BN rows h = x W^T + b, and
// their column sums / sums of squares go to a.stats from the SAME values on their way out, in fp64 registers across the workgroup's tiles (as
// linear_fwd_bf16_kernel<STATS> does), one atomic pair per column and workgroup at the end.  Rows past m_rows (inverse scale 0: they would count as `bias`) are masked.
template <int NJ, bool PROF, bool VEC, bool STATS = false>
__global__ __launch_bounds__(D_NT) __attribute__((amdgpu_waves_per_eu(NJ > 2 ? 1 : 2, NJ > 2 ? 1 : 2))) void linear_f16x3_dma_kernel(L16Args a) {
    constexpr int D_BN = 64 * NJ, D_BUF = (D_BM + D_BN) * L_LINE;          // columns per tile, bytes per slice buffer
    constexpr int NWI = 2 * NJ;                                              // DMA instructions per wave and slice for the weights' lines (8 rows each)
    double st_sum[NJ] = {}, st_sq[NJ] = {};
    const int dbg = PROF ? a.dbg : 0;
    extern __shared__ __attribute__((aligned(16))) unsigned l16_lds[];
    unsigned char *const lds = reinterpret_cast<unsigned char *>(l16_lds);
    float *const rtabp = reinterpret_cast<float *>(lds + 2 * D_BUF);      // [2][D_BM] inverse row scales of the current / next tile
    float *const ecol = rtabp + 2 * D_BM;                                  // [D_BN] scale | [D_BN] constant
    int *const vtab = reinterpret_cast<int *>(ecol + 2 * D_BN);            // [2][NJ][64] offsets of the 16-byte stores | [2][NJ][32] of the 4-byte ones
    auto clk = [&]() -> unsigned long long {
        if (!PROF) return 0ull;
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long v = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
        return v;
    };
    unsigned long long pq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int xcd = blockIdx.x & 7, slot_in_xcd = blockIdx.x >> 3;
    const int ct = slot_in_xcd % a.col_tiles, grp = slot_in_xcd / a.col_tiles;
    const int n0 = ct * D_BN;
    const int64_t n_tiles = (a.m_rows + D_BM - 1) / D_BM;
    const int64_t tile0 = (int64_t)xcd * a.groups + grp, tstride = 8 * (int64_t)a.groups;
    const int64_t n_mine = tile0 < n_tiles ? (n_tiles - tile0 + tstride - 1) / tstride : 0;
    const int n_slices = a.k_pad / L_BK;
    const int row_bytes = n_slices * L_LINE;
    const int rstride = a.n_out * 4;

    for (int cidx = tid; cidx < D_BN; cidx += D_NT) {
        const int col = n0 + cidx;
        const bool cok = col < a.n_out;
        const float e_bias = (cok && a.bias) ? a.bias[col] : 0.f;
        float es = cok ? a.colinv[col] : 0.f, ec = e_bias;
        if (cok && a.bn_scale) { ec = (e_bias - a.bn_mean[col]) * a.bn_scale[col] + a.bn_shift[col]; es *= a.bn_scale[col]; }
        ecol[cidx] = es; ecol[D_BN + cidx] = ec;
    }
    for (int q = tid; q < 2 * NJ * 64; q += D_NT) {
        const int l = q & 63, k = (q >> 6) % NJ, w = q / (64 * NJ);
        const int idx = 64 * k + l, c4 = idx % (8 * NJ), col = n0 + w * 32 * NJ + 4 * c4;
        vtab[q] = col < a.n_out ? ((idx / (8 * NJ)) - (64 * k) / (8 * NJ)) * rstride + col * 4 : 0x7f000000;
    }
    for (int q = tid; q < 2 * NJ * 32; q += D_NT) {
        const int l = q & 31, j = (q >> 5) % NJ, w = q / (32 * NJ);
        const int col = n0 + w * 32 * NJ + 32 * j + l;
        vtab[2 * NJ * 64 + q] = col < a.n_out ? col * 4 : 0x7f000000;
    }
    // staging: wave w moves rows (output columns) 16 NJ w + 8 i .. + 7 with instruction i; lane -> row 8 i + (lane >> 3), slot lane & 7,
    // i.e. piece (lane & 7) ^ swizzle(row)
    const int s_r8 = lane >> 3, s_q = lane & 7;
    unsigned wofs[NWI];
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
        const int r = 8 * NWI * wave + 8 * i + s_r8;
        int j = n0 + r;
        j = j < a.n_out ? j : 0;
        wofs[i] = (unsigned)j * (unsigned)row_bytes + 16u * (unsigned)(s_q ^ ((r >> 1) & 7));
    }
    auto row_ofs = [&](int64_t row0, unsigned *aofs) {
        const int64_t left = a.m_rows - row0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 32 * wave + 8 * i + s_r8;
            const int rc = r < left ? r : (int)(left > 0 ? left - 1 : 0);
            aofs[i] = (unsigned)rc * (unsigned)row_bytes + 16u * (unsigned)(s_q ^ ((r >> 1) & 7));
        }
    };
    auto tile_base = [&](int64_t row0) { return a.aplanes + (row0 < a.m_rows ? row0 : 0) * row_bytes; };
    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    auto fetch_rows = [&](const unsigned char *abase, const unsigned *aofs, int c, int buf) {
        unsigned char *const da = lds + buf * D_BUF + (32 * wave) * L_LINE;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(abase + aofs[i] + c * L_LINE), (lptr_t)(da + 8 * i * L_LINE), 16, 0, 0);
    };
    auto fetch_weights = [&](int c, int buf) {
        unsigned char *const dw = lds + buf * D_BUF + (D_BM + 8 * NWI * wave) * L_LINE;
#pragma unroll
        for (int i = 0; i < NWI; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(a.wplanes + wofs[i] + c * L_LINE), (lptr_t)(dw + 8 * i * L_LINE), 16, 0, 0);
    };
    auto fetch = [&](const unsigned char *abase, const unsigned *aofs, int c, int buf) { fetch_rows(abase, aofs, c, buf); fetch_weights(c, buf); };
    // fragment reads: piece 4 plane + 2 s + lh of row (.. + li): slot = piece ^ ((li >> 1) & 7) (the tile offsets of a row are multiples of 16)
    const int swz = (li >> 1) & 7;
    int fo[2][2];                                                           // [plane][k-step] byte offset inside the row
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int st = 0; st < 2; ++st) fo[pl][st] = 16 * ((4 * pl + 2 * st + lh) ^ swz);
    f32x16 acc[2][NJ];
#define L16_MF(acc, x, y) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, x), __builtin_bit_cast(h16x8, y), acc, 0, 0, 0)
    auto products = [&](int buf, auto between) {
        const unsigned char *ap = lds + buf * D_BUF + (wm * 64 + li) * L_LINE, *bp = lds + buf * D_BUF + (D_BM + wn * 32 * NJ + li) * L_LINE;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            if (st == 1) between();
            un4 ah[2], al[2], bh[NJ], bl[NJ];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const un4 *>(ap + i * (32 * L_LINE) + fo[0][st]);
                al[i] = *reinterpret_cast<const un4 *>(ap + i * (32 * L_LINE) + fo[1][st]);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                bh[j] = *reinterpret_cast<const un4 *>(bp + j * (32 * L_LINE) + fo[0][st]);
                bl[j] = *reinterpret_cast<const un4 *>(bp + j * (32 * L_LINE) + fo[1][st]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) L16_MF(acc[i][j], al[i], bh[j]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) L16_MF(acc[i][j], ah[i], bl[j]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) L16_MF(acc[i][j], ah[i], bh[j]);
        }
    };
    if (n_mine > 0) {
        unsigned aofs[4], aofs_next[4];
        const unsigned char *abase = tile_base(tile0 * D_BM), *abase_next = abase;
        row_ofs(tile0 * D_BM, aofs);
        fetch(abase, aofs, 0, 0);
        if (tid < D_BM) { const int64_t r = tile0 * D_BM + tid; rtabp[tid] = a.rowinv[r < a.m_pad ? r : a.m_pad - 1]; }
        int buf = 0, slot = 0;
        for (int64_t ti = 0; ti < n_mine; ++ti) {
            const int64_t row0 = (tile0 + ti * tstride) * D_BM, row_next = ti + 1 < n_mine ? row0 + tstride * D_BM : row0;
            float rt_next;                                                  // (every wave issues this load: the counted wait below relies on it)
            { const int64_t r = row_next + (tid & (D_BM - 1)); rt_next = a.rowinv[r < a.m_pad ? r : a.m_pad - 1]; }
            row_ofs(row_next, aofs_next);
            abase_next = tile_base(row_next);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            for (int c = 0; c < n_slices; ++c) {
                const unsigned long long q0 = clk();
                // slice c has landed (this wave's share; the barrier makes that true for all of them) and nobody reads the other buffer
                // (first slice of a later tile: the 16 output stores of the previous tile and the row-scale load below were issued
                //  AFTER this slice's loads -- memory operations complete in order, so "all but the newest 17" covers the slice
                //  without waiting for the stores)
                if (VEC && !PROF && c == 0 && ti > 0) {
                    if (NJ == 2) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");         // (8 NJ stores + the row-scale load)
                    else asm volatile("s_waitcnt vmcnt(41)" ::: "memory");
                }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                l16_barrier();
                // (the next tile's row scales go to LDS here, where nothing is in flight: consumed at the end of the tile, the
                //  compiler's wait for this load would also wait for the next tile's first slice)
                if (c == 1 && tid < D_BM) rtabp[(slot ^ 1) * D_BM + tid] = rt_next;
                const unsigned long long q1 = clk();
                const bool wrap = c + 1 >= n_slices;
                // the next slice's rows are issued in front of the products, its weights between the two k-steps: the wave's matrix
                // chain starts after four DMA instructions instead of eight (their issue waits for the memory pipeline)
                const unsigned char *const nb = wrap ? abase_next : abase;
                const unsigned *const no = wrap ? aofs_next : aofs;
                const int nc = wrap ? 0 : c + 1;
                if (!(dbg & 4)) fetch_rows(nb, no, nc, buf ^ 1);
                const unsigned long long q2 = clk();
                if (!(dbg & 2)) products(buf, [&]() { if (!(dbg & 4)) fetch_weights(nc, buf ^ 1); });
                else if (!(dbg & 4)) fetch_weights(nc, buf ^ 1);
                buf ^= 1;
                if (PROF) { const unsigned long long q3 = clk(); pq[0] += q2 - q1; pq[1] += q3 - q2; pq[3] += q1 - q0; pq[5] += 1; }
            }
            const unsigned long long qe0 = clk();
            l16_barrier();                                                 // every wave is done with the last slice's buffer: it is the epilogue's scratch
            const int64_t rows_left = a.m_rows - row0;
            const int nrows = rows_left < D_BM ? (int)rows_left : D_BM;
            const __amdgpu_buffer_rsrc_t orow = __builtin_amdgcn_make_buffer_rsrc(a.out + row0 * a.n_out, 0, nrows * rstride, 0x00020000);
            const float act_lo = a.act == 1 ? 0.f : -INFINITY;
            constexpr int SP = 32 * NJ * 4 + 16;
            unsigned char *const scr = lds + (buf ^ 1) * D_BUF + wave * (8 * SP);
            const float *const rt = rtabp + slot * D_BM;
            float es[NJ], ec[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) { es[j] = ecol[wn * 32 * NJ + 32 * j + li]; ec[j] = ecol[D_BN + wn * 32 * NJ + 32 * j + li]; }
            auto emit = [&](auto simple) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int rl = wm * 64 + i * 32 + 8 * gq;
                        const float ivv[4] = {rt[rl + 4 * lh], rt[rl + 4 * lh + 1], rt[rl + 4 * lh + 2], rt[rl + 4 * lh + 3]};
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int j = 0; j < NJ; ++j) {
                                float y = fmaf(acc[i][j][4 * gq + r], ivv[r] * es[j], ec[j]);
                                if (decltype(simple)::value) y = y < act_lo ? act_lo : y;       // (a NaN stays a NaN)
                                else y = l16_act_slow(y, a.act);
                                *reinterpret_cast<float *>(scr + (r + 4 * lh) * SP + (li + 32 * j) * 4) = y;
                                if (STATS) {
                                    const double ym = rl + 4 * lh + r < nrows ? (double)y : 0.0;
                                    st_sum[j] += ym;
                                    st_sq[j] = fma(ym, ym, st_sq[j]);
                                }
                            }
                        if ((dbg & 1) && acc[i][0][4 * gq] != 12345.f) continue;
                        if (VEC) {
#pragma unroll
                            for (int k = 0; k < NJ; ++k) {
                                const int idx = 64 * k + lane, row = idx / (8 * NJ), c4 = idx % (8 * NJ);
                                const un4 v = *reinterpret_cast<const un4 *>(scr + row * SP + 16 * c4);
                                // (row offset in the vector offset, s_nop behind the store: see linear_f16x3_planes_kernel)
                                __builtin_amdgcn_raw_buffer_store_b128(v, orow, vtab[(wn * NJ + k) * 64 + lane] + (rl + (64 * k) / (8 * NJ)) * rstride, 0, 0);
                                asm volatile("s_nop 1" ::: "memory");
                            }
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
#pragma unroll
                                for (int j = 0; j < NJ; ++j) {
                                    const unsigned y = *reinterpret_cast<const unsigned *>(scr + (r + 4 * lh) * SP + (li + 32 * j) * 4);
                                    __builtin_amdgcn_raw_buffer_store_b32(y, orow, vtab[2 * NJ * 64 + (wn * NJ + j) * 32 + li] + (rl + r + 4 * lh) * rstride, 0, 0);
                                }
                        }
                    }
                }
            };
            if (a.act <= 1) emit(std::true_type{});
            else emit(std::false_type{});
            if (PROF) { pq[4] += clk() - qe0; pq[6] += 1; }
            slot ^= 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) aofs[i] = aofs_next[i];
            abase = abase_next;
        }
    }
    if (STATS) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            double sv = st_sum[j], qv = st_sq[j];
            sv += __shfl_xor(sv, 32);
            qv += __shfl_xor(qv, 32);
            const int col = n0 + wn * 32 * NJ + 32 * j + li;
            if (lh == 0 && col < a.n_out && n_mine > 0) {
                atomicAdd(a.stats + col, sv);
                atomicAdd(a.stats + a.n_out + col, qv);
            }
        }
    }
    if (PROF && a.prof && blockIdx.x == 0 && tid == 0)
        for (int q = 0; q < 8; ++q) a.prof[q] = pq[q];
#undef L16_MF
}

}  // namespace

}  // namespace gsn

using namespace gsn;

extern "C" int64_t gsn_linear_f16x3_kpad(int64_t k_total) {
    const int64_t k = (k_total + L_BK - 1) / L_BK * L_BK;
    return k < 2 * L_BK ? 2 * L_BK : k;                                   // (the kernel keeps two slices in flight)
}

// rows of a row scratch: m_rows and at least 128 more (zero planes, zero inverse scales), in whole 256-row tiles
extern "C" int64_t gsn_linear_f16x3_mpad(int64_t m_rows) { return m_rows <= 0 ? 0 : (m_rows + 128 + 255) / 256 * 256; }

extern "C" int64_t gsn_linear_f16x3_scratch_bytes(int64_t m_rows, int64_t k_total) {
    if (m_rows <= 0 || k_total <= 0) return 0;
    const int64_t m_pad = gsn_linear_f16x3_mpad(m_rows), k_pad = gsn_linear_f16x3_kpad(k_total);
    return m_pad * 4 + m_pad * k_pad * 4;                                 // inverse row scales | the rows' planes (zero past m_rows)
}

static int l16_prepare(const float *W, int64_t n_out, int64_t k_total, int64_t w_rs, int64_t w_cs, void *planes, float *col_inv, void *stream) {
    if (!W || !planes || !col_inv || n_out <= 0 || k_total <= 0 || n_out > (1 << 24) || k_total > (1 << 20) || w_rs < 1 || w_cs < 1)
        return set_error(GSN_E_INVALID, "gsn_linear_f16x3_prepare_hip: bad argument");
    const int k_pad = (int)gsn_linear_f16x3_kpad(k_total);
    if (n_out * (int64_t)k_pad * 4 >= ((int64_t)1 << 31))
        return set_error(GSN_E_UNSUPPORTED, "gsn_linear_f16x3_prepare_hip: weight planes of 2 GiB and more are not supported");
    hipLaunchKernelGGL(lin16_prepare_kernel, dim3((unsigned)n_out), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), W, (int)n_out, (int)k_total,
                       k_pad, reinterpret_cast<_Float16 *>(planes), col_inv, w_rs, w_cs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "lin16_prepare_kernel: %s", hipGetErrorString(e));
    return GSN_OK;
}

extern "C" int gsn_linear_f16x3_prepare_hip(const float *W, int64_t n_out, int64_t k_total, void *planes, float *col_inv, void *stream) {
    return l16_prepare(W, n_out, k_total, k_total, 1, planes, col_inv, stream);
}

// the same from a weight given through element strides (a transposed view: row stride 1, column stride = the leading dimension)
extern "C" int gsn_linear_f16x3_prepare_strided_hip(const float *W, int64_t n_out, int64_t k_total, int64_t w_row_stride, int64_t w_col_stride,
                                                    void *planes, float *col_inv, void *stream) {
    return l16_prepare(W, n_out, k_total, w_row_stride, w_col_stride, planes, col_inv, stream);
}

// the rows of a call: blocks -> L16Args (widths, K, the scratch's two parts), then the pre-pass that fills the scratch
static int l16_rows(int64_t m_rows, int n_blocks, const gsn_block *blocks, float *row_scratch, L16Args &a, const char *who) {
    if (n_blocks < 1 || n_blocks > L_MAXB || !blocks || !row_scratch)
        return set_error(GSN_E_INVALID, "%s: need 1..%d input blocks and the row scratch", who, L_MAXB);
    if ((reinterpret_cast<uintptr_t>(row_scratch) & 15) != 0) return set_error(GSN_E_INVALID, "%s: row_scratch must be 16-byte aligned", who);
    a.m_rows = m_rows; a.n_blocks = n_blocks;
    int k_total = 0;
    const float *bd[L_MAXB] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int bw[L_MAXB] = {0, 0, 0, 0, 0};
    for (int b = 0; b < n_blocks; ++b) {
        if (!blocks[b].data || blocks[b].idx || blocks[b].idx32 || blocks[b].width <= 0 || (blocks[b].width & 3) ||
            (reinterpret_cast<uintptr_t>(blocks[b].data) & 15))
            return set_error(GSN_E_UNSUPPORTED, "%s: block %d: direct rows (no index), width a multiple of 4, 16-byte aligned", who, b);
        bd[b] = blocks[b].data; bw[b] = (int)blocks[b].width;
        k_total += (int)blocks[b].width;
    }
    for (int b = n_blocks; b < L_MAXB; ++b) { bd[b] = bd[0]; bw[b] = 1 << 28; }      // (never selected: their first column is past K)
    a.b0 = bd[0]; a.b1 = bd[1]; a.b2 = bd[2]; a.b3 = bd[3]; a.b4 = bd[4];
    a.w0 = bw[0]; a.w1 = bw[1]; a.w2 = bw[2]; a.w3 = bw[3]; a.w4 = bw[4];
    a.k_total = k_total; a.k_pad = (int)gsn_linear_f16x3_kpad(k_total);
    // scratch: [m_pad] inverse row scales | [m_pad][k_pad / 32] lines of the rows' planes
    a.m_pad = gsn_linear_f16x3_mpad(m_rows);
    a.rowinv = row_scratch;
    a.aplanes = reinterpret_cast<unsigned char *>(row_scratch + a.m_pad);
    return GSN_OK;
}

static void l16_split_launch(const L16Args &a, hipStream_t st) {
    // one block of 16-byte aligned rows: the grid-stride kernel (GSN_L16_SPLIT_WGS workgroups, default 1024; 0: the one-tile-per-workgroup kernel)
    static const int64_t split_wgs = [] { const char *e = getenv("GSN_L16_SPLIT_WGS"); return e ? (int64_t)atoll(e) : (int64_t)1024; }();
    int64_t gx = a.m_pad / 32;
    if (split_wgs > 0 && a.n_blocks == 1 && (a.w0 & 3) == 0) {
        if (gx > split_wgs) gx = split_wgs;
        hipLaunchKernelGGL(lin16_split_rows1_kernel, dim3((unsigned)gx), dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL(lin16_split_rows_kernel, dim3((unsigned)gx), dim3(256), 0, st, a);
    }
}

static int l16_fwd(int64_t m_rows, int n_blocks, const gsn_block *blocks, const void *planes, const float *col_inv,
                   const float *bias, int64_t n_out, const float *bn_mean, const float *bn_scale, const float *bn_shift,
                   int act, float *row_scratch, float *out, double *stats, void *stream, int rows_are_split) {
    if (!planes || !col_inv || !out || n_out <= 0)
        return set_error(GSN_E_INVALID, "gsn_linear_f16x3_fwd_hip: need the weight planes and out");
    if ((bn_scale != nullptr) != (bn_shift != nullptr) || (bn_scale != nullptr) != (bn_mean != nullptr))
        return set_error(GSN_E_INVALID, "gsn_linear_f16x3_fwd_hip: bn_mean, bn_scale and bn_shift go together");
    if (act < 0 || act > 3) return set_error(GSN_E_INVALID, "gsn_linear_f16x3_fwd_hip: act must be 0..3");
    if (m_rows <= 0) return GSN_OK;
    L16Args a{};
    const int rc = l16_rows(m_rows, n_blocks, blocks, row_scratch, a, "gsn_linear_f16x3_fwd_hip");
    if (rc != GSN_OK) return rc;
    const int k_total = a.k_total;
    a.n_out = (int)n_out; a.act = act;
    if (n_out * (int64_t)a.k_pad * 4 >= ((int64_t)1 << 31) || n_out >= (1 << 24))
        return set_error(GSN_E_UNSUPPORTED, "gsn_linear_f16x3_fwd_hip: weight planes of 2 GiB and more are not supported");
    a.wplanes = reinterpret_cast<const unsigned char *>(planes); a.colinv = col_inv;
    a.bias = bias; a.bn_mean = bn_mean; a.bn_scale = bn_scale; a.bn_shift = bn_shift;
    a.out = out; a.stats = stats;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!rows_are_split) l16_split_launch(a, st);
    // 128 x 128 tiles, two workgroups of 4 waves per CU (256 x 256 tiles with 8 waves measured 3-5 % slower: no registers left
    // for the second set of loads in flight).  GSN_L16_WIDE=1 (r06, measured, NOT the default): 128 x 320 tiles, one workgroup per CU, where
    // that wastes fewer columns and the product is large enough to fill the chip that way (the d = 300 ogb stages: n_out = 300 is one column
    // tile instead of three, 600 two instead of five: 40-60 % less L2 -> CU traffic, 17 % less matrix work at n_out = 300) -- 105 083 rows:
    // 300 -> 600 194 vs 188 us, 600 -> 300 178 vs 183 us, 196 608 x 300 -> 600 295 vs 318 us (scripts/gpu/r6_l16_wide.py; config-4 step 16.68 vs
    // 16.45 ms): neither the column waste nor the traffic is what bounds this kernel.
    const int wm = 2;
    const char *wide_env = getenv("GSN_L16_WIDE");                          // (read per call: a test switches it)
    const bool wide_on = wide_env && wide_env[0] == '1';
    const int64_t cols128 = (n_out + 127) / 128 * 128, cols320 = (n_out + 319) / 320 * 320;
    const bool vec_ok = n_out % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && !getenv("GSN_L16_NOVEC");
    const bool wide = wide_on && vec_ok && n_out > 256 && cols320 <= cols128 && (m_rows + 127) / 128 * (cols320 / 320) >= 512 && !getenv("GSN_L16_REGSTAGE") &&
                      !getenv("GSN_L16_PROF") && !getenv("GSN_L16_DBG");
    const int nj = wide ? 5 : 2;
    const int bm = 64 * wm, bn = 64 * nj;
    const int64_t n_tiles = (m_rows + bm - 1) / bm;
    const int col_tiles = (int)((n_out + bn - 1) / bn);
    const int slots = wide ? 32 : 64;                                       // workgroups per XCD
    int groups = slots / col_tiles;
    if (groups < 1) groups = 1;
    const int64_t need = (n_tiles + 7) / 8;
    if (groups > need) groups = (int)need;
    a.col_tiles = col_tiles; a.groups = groups;
    a.dbg = getenv("GSN_L16_DBG") ? atoi(getenv("GSN_L16_DBG")) : 0;
    static const bool want_prof = getenv("GSN_L16_PROF") != nullptr;
    a.prof = nullptr;
    if (want_prof && hipMalloc(reinterpret_cast<void **>(&a.prof), 64) != hipSuccess) a.prof = nullptr;
    size_t lds = (size_t)2 * (bm + bn) * P_PITCH + 8 * bn + 4 * (2 * nj * 64 + 2 * nj * 32);   // slice buffers | epilogue tables
    if (getenv("GSN_CHAIN_TRACE"))
        fprintf(stderr, "gsn linear: linear_f16x3_kernel (%d x %d tiles) M %lld K %d N %d grid 8 x %d x %d\n", bm, bn, (long long)m_rows, a.k_total, (int)n_out,
                groups, col_tiles);
    const dim3 grid((unsigned)(8 * groups * col_tiles));
    const bool vec = n_out % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && !getenv("GSN_L16_NOVEC");      // 16-byte output stores
    if (stats && (!vec || act != 0 || bn_scale))
        return set_error(GSN_E_UNSUPPORTED, "gsn_linear_f16x3_fwd_stats_hip: statistics go with plain pre-BN rows (act 0, no bn vectors), n_out a multiple of 4, out 16-byte aligned");
    static DeviceOnce attr_set[9];
    const int attr_dev = current_device();
    hipError_t e0 = hipSuccess;
    auto launch = [&](auto kern, int which) {
        if (!attr_set[which].done(attr_dev)) {
            e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e0 != hipSuccess) return;
            attr_set[which].mark(attr_dev);
        }
        hipLaunchKernelGGL(kern, grid, dim3(128 * wm), lds, st, a);
    };
    static const bool reg_stage = getenv("GSN_L16_REGSTAGE") != nullptr;  // (A/B: slices staged through registers)
    const size_t lds_dma = (size_t)2 * (D_BM + bn) * L_LINE + 4 * (2 * D_BM + 2 * bn) + 4 * (2 * nj * 64 + 2 * nj * 32);   // slice buffers | row scales, column tables | store offsets
    if (stats && wide) {
        lds = lds_dma;
        launch(linear_f16x3_dma_kernel<5, false, true, true>, 8);
    } else if (stats) {
        lds = lds_dma;
        launch(linear_f16x3_dma_kernel<2, false, true, true>, 6);
    } else if (reg_stage) {
        if (want_prof || a.dbg) launch(linear_f16x3_planes_kernel<2, 2, true, true>, 0);       // diagnostic build
        else if (!vec) launch(linear_f16x3_planes_kernel<2, 2, false, false>, 1);
        else launch(linear_f16x3_planes_kernel<2, 2, false, true>, 2);
    } else {
        lds = lds_dma;
        if (wide) launch(linear_f16x3_dma_kernel<5, false, true>, 7);
        else if (want_prof || a.dbg) launch(linear_f16x3_dma_kernel<2, true, true>, 3);
        else if (!vec) launch(linear_f16x3_dma_kernel<2, false, false>, 4);
        else launch(linear_f16x3_dma_kernel<2, false, true>, 5);
    }
    if (e0 != hipSuccess) return set_error(GSN_E_HIP, "hipFuncSetAttribute(linear_f16x3_kernel): %s", hipGetErrorString(e0));
    if (a.prof) {
        unsigned long long h[8];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h, a.prof, 64, hipMemcpyDeviceToHost);
        (void)hipFree(a.prof);
        const double n = h[5] ? (double)h[5] : 1.0, nt = h[6] ? (double)h[6] : 1.0;
        fprintf(stderr, "l16prof: per slice: fetch issue %.0f products %.0f stage(+load wait) %.0f barrier %.0f | epilogue %.0f per tile (%llu slices, %llu tiles)\n",
                h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / nt, h[5], h[6]);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "linear_f16x3_kernel: %s", hipGetErrorString(e));
    return GSN_OK;
}

extern "C" int gsn_linear_f16x3_fwd_hip(int64_t m_rows, int n_blocks, const gsn_block *blocks, const void *planes, const float *col_inv,
                                        const float *bias, int64_t n_out, const float *bn_mean, const float *bn_scale, const float *bn_shift,
                                        int act, float *row_scratch, float *out, void *stream) {
    return l16_fwd(m_rows, n_blocks, blocks, planes, col_inv, bias, n_out, bn_mean, bn_scale, bn_shift, act, row_scratch, out, nullptr, stream, 0);
}

// train-mode BatchNorm stage: out = the pre-BN rows x W^T + b, stats[2][n_out] (fp64, ADDED to: the caller zeroes it) their column sums and
// sums of squares, from the same launch (linear_fwd_bf16_kernel<STATS>'s contract, at half its matrix work)
extern "C" int gsn_linear_f16x3_fwd_stats_hip(int64_t m_rows, int n_blocks, const gsn_block *blocks, const void *planes, const float *col_inv,
                                              const float *bias, int64_t n_out, float *row_scratch, float *out, double *stats, void *stream) {
    if (!stats) return set_error(GSN_E_INVALID, "gsn_linear_f16x3_fwd_stats_hip: stats is null");
    return l16_fwd(m_rows, n_blocks, blocks, planes, col_inv, bias, n_out, nullptr, nullptr, nullptr, 0, row_scratch, out, stats, stream, 0);
}

// the rows' pre-pass by itself: row_scratch as gsn_linear_f16x3_fwd_hip leaves it (inverse row scales | the two fp16 planes) -- for
// gsn_wgrad_f16x3_hip when no product over these rows runs beside it, and for products over rows split earlier (`rows_are_split`)
extern "C" int gsn_linear_f16x3_split_rows_hip(int64_t m_rows, int n_blocks, const gsn_block *blocks, float *row_scratch, void *stream) {
    if (m_rows <= 0) return GSN_OK;
    L16Args a{};
    const int rc = l16_rows(m_rows, n_blocks, blocks, row_scratch, a, "gsn_linear_f16x3_split_rows_hip");
    if (rc != GSN_OK) return rc;
    l16_split_launch(a, reinterpret_cast<hipStream_t>(stream));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "lin16_split_rows_kernel: %s", hipGetErrorString(e));
    return GSN_OK;
}

extern "C" int gsn_linear_f16x3_fwd_presplit_hip(int64_t m_rows, int n_blocks, const gsn_block *blocks, const void *planes, const float *col_inv,
                                                 const float *bias, int64_t n_out, const float *bn_mean, const float *bn_scale, const float *bn_shift,
                                                 int act, float *row_scratch, float *out, void *stream) {
    return l16_fwd(m_rows, n_blocks, blocks, planes, col_inv, bias, n_out, bn_mean, bn_scale, bn_shift, act, row_scratch, out, nullptr, stream, 1);
}

extern "C" int gsn_linear_f16x3_fwd_stats_presplit_hip(int64_t m_rows, int n_blocks, const gsn_block *blocks, const void *planes, const float *col_inv,
                                                       const float *bias, int64_t n_out, float *row_scratch, float *out, double *stats, void *stream) {
    if (!stats) return set_error(GSN_E_INVALID, "gsn_linear_f16x3_fwd_stats_presplit_hip: stats is null");
    return l16_fwd(m_rows, n_blocks, blocks, planes, col_inv, bias, n_out, nullptr, nullptr, nullptr, 0, row_scratch, out, stats, stream, 1);
}
