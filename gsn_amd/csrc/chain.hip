// HP-2 fused MLP chain for gfx950: one or two models_misc.mlp stages on one row tile without leaving the chip.
//
//   stage s:  Y_s = act_s( (X_s W_s^T + b_s - mean_s) * scale_s + shift_s ),   X_s = [ HBM blocks of stage s | Y_{s-1} ]
//
// e.g. update_fn of every layer:  h = relu(bn(X W3^T + b3)) ; out = h W4^T + b4  (models_misc.py:52-58), whose [N,128]
// intermediate never touches HBM (for msg_kind='general' X = [x | S | deg] with the deferred last Linear of msg_fn folded
// into W3, layers.py); and the single edge stage  r = relu(bn(cat(x_i, x_j, id, e) W1^T + b1)).
//
// Why this shape (measured on the previous kernel, profiles/r01_*): with K ~ N_out ~ 128 one fp32 stage has only ~32 flop
// per HBM byte, so un-fused stages wait on memory (SQ_WAIT_ANY 55 %).  Here
//   * the WEIGHTS LIVE IN REGISTERS: wave w owns output columns [32w, 32w+32) of every stage and keeps its B fragments
//     (lane (li,lh): W_s[32w+li][2q+lh], ceil(K/2) floats per stage) in VGPRs for the whole persistent kernel -- no LDS or
//     L2 traffic for weights at all, and LDS is free for activations;
//   * a tile is 64 rows x full K in LDS (pitch odd -> conflict-free fragment reads), so there is one barrier per stage,
//     not per K-slice; the workgroup has 8 waves = 2 per SIMD: waves w and w+4 own the same 32 output columns and the two
//     32-row halves of the tile (one v_mfma_f32_32x32x2_f32 chain each, back-to-back issue = its 64-cycle latency), so
//     the per-wave staging / epilogue work -- which is VALU-issue bound and cannot overlap the wave's own MFMAs --
//     halves, and two waves issue it concurrently on every SIMD;
//   * the HBM inputs of the NEXT tile (all stages) are prefetched into registers while the current tile computes
//     (>= 5k MFMA cycles of cover), gather indices two tiles ahead;
//   * stage outputs are written by the epilogue straight into the next stage's LDS input tile.
// fp32 in / fp32 accumulate, exact f32 MFMA (157.3 TF peak).  Limits: K_s <= 16*MAXCH, n_out_s <= 128, HBM part of a
// stage s >= 1 at most 64 columns; anything else goes through linear.hip.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "chain_common.h"

namespace gsn {


// CH0 / CH1 = register chunks (16 k each) of the two stages' weight fragments; WPE = waves per SIMD the kernel is
// compiled for; NW = waves per workgroup:
//   NW = 8 (single-stage chains): waves w and w+4 own the same 32 output columns and the two 32-row halves of the tile,
//           so the VALU-issue-bound staging / epilogue work per wave halves and 2-4 waves share every SIMD;
//   NW = 4 (two-stage chains): one wave per SIMD with both row halves (two independent accumulator chains) -- 144+
//           weight registers per lane leave no room for a second wave (it spills and the MFMA phase collapses).
template <int NST, int CH0, int CH1, bool STATS, int WPE, bool SEG, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void mlp_chain_kernel(ChainArgs a) {
    constexpr int CT = 64 * NW;                     // threads per workgroup
    constexpr int NACC = NW == 4 ? 2 : 1;           // row halves handled by one wave
    constexpr int RSTEP = CT / 32;                  // tile rows covered by one staging pass
    constexpr int NROW = CBM / RSTEP;               // staging passes (rows per thread)
    constexpr int NP = NW == 4 ? 2 : 1;             // row-source passes (blocks per thread)
    constexpr int PF0_J = (CH0 * CHK + 31) / 32;    // 32-column groups of the stage-0 input
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int pitch = a.pitch;
    float *buf0 = lds;
    float *buf1 = lds + CBM * pitch;
    int *rsrc = reinterpret_cast<int *>(lds + 2 * CBM * pitch);  // [3][RS_STRIDE] ring: tiles t, t+1, t+2

    const int tid = threadIdx.x;
    const int lane = tid & 63, w8 = tid >> 6;
    const int w = w8 & 3;                  // column block: output columns 32w .. 32w+31
    const int rh = NW == 8 ? (w8 >> 2) : 0;  // row half of the tile owned by this wave (NW == 8)
    const int li = lane & 31, lh = lane >> 5;
    const int kc0 = tid & 31, r0 = tid >> 5;   // staging: column kc0 (+32j), rows r0 + RSTEP * i
    const int64_t n_tiles = (a.m_rows + CBM - 1) / CBM;

    // ---- weights -> registers (once) ----------------------------------------------------------------------------
    float B0[CH0][8], B1[CH1 > 0 ? CH1 : 1][8];
    int nch[2] = {0, 0};
    {
        const ChainStage &st = a.st[0];
        nch[0] = (st.k_total + CHK - 1) / CHK;
        const int col = 32 * w + li;
#pragma unroll
        for (int ch = 0; ch < CH0; ++ch)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k = ch * CHK + 2 * q + lh;
                float v = 0.f;
                if (k < st.k_total && col < st.n_out) v = st.W[(int64_t)col * st.k_total + k];
                B0[ch][q] = v;
            }
    }
    if (NST > 1) {
        const ChainStage &st = a.st[1];
        nch[1] = (st.k_total + CHK - 1) / CHK;
        const int col = 32 * w + li;
#pragma unroll
        for (int ch = 0; ch < CH1; ++ch)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k = ch * CHK + 2 * q + lh;
                float v = 0.f;
                if (k < st.k_total && col < st.n_out) v = st.W[(int64_t)col * st.k_total + k];
                B1[ch][q] = v;
            }
    }

    // ---- per-thread column maps of the stage-0 input ------------------------------------------------------------
    ColMap cm0[PF0_J];
#pragma unroll
    for (int j = 0; j < PF0_J; ++j) cm0[j] = col_map(a, 0, kc0 + 32 * j);

    // zero both activation tiles once: padded columns must hold finite values (their weights are zero)
    for (int i = tid; i < 2 * CBM * pitch; i += CT) lds[i] = 0.f;

    // per-stage epilogue constants of this lane's output column (loaded once: a load inside the tile loop would put a
    // vmcnt(0) in front of every use and drain the prefetch / the stores)
    float e_bias[NST], e_scale[NST], e_c0[NST];   // y = acc * scale + c0,  c0 = (bias - mean) * scale + shift
#pragma unroll
    for (int s = 0; s < NST; ++s) {
        const ChainStage &st = a.st[s];
        const int col = 32 * w + li;
        const bool cok = col < st.n_out;
        e_bias[s] = (cok && st.bias) ? st.bias[col] : 0.f;
        e_scale[s] = 1.f; e_c0[s] = e_bias[s];
        if (cok && st.bn_scale) { e_scale[s] = st.bn_scale[col]; e_c0[s] = (e_bias[s] - st.bn_mean[col]) * e_scale[s] + st.bn_shift[col]; }
    }

    double st_sum = 0.0, st_sq = 0.0;
    float pf0[PF0_J][NROW];
    RowSrcC<NP> rsn;
    const RowSrcThread<NP> rst = rs_thread<NP>(a, tid);

    // NOTE: the loaded values are kept RAW in the prefetch registers; masking (padded columns, rows past the end) is
    // applied when they are written to LDS.  Selecting on the value right here would make the compiler wait for every
    // load immediately after issuing it (vmcnt countdown) and serialise the prefetch.
    auto prefetch_j = [&](const int *rs, int j) {   // j must be a compile-time constant at the call site (register index)
#pragma unroll
        for (int i = 0; i < NROW; ++i) {
            const int sr = rs[cm0[j].rsoff + r0 + RSTEP * i];
            pf0[j][i] = cm0[j].base[(int64_t)(sr < 0 ? 0 : sr) * cm0[j].bw];
        }
    };
    // prefetched rows -> LDS.  No masking needed: a padded column (k >= K) holds a finite clamped-address value and meets a
    // zero weight; a row past the end holds row 0's values and is never emitted (store guard / target -1 / stats guard).
    auto stage_in_j = [&](float *dst, int j) {
#pragma unroll
        for (int i = 0; i < NROW; ++i) dst[(r0 + RSTEP * i) * pitch + kc0 + 32 * j] = pf0[j][i];
    };
    auto prefetch = [&](const int *rs) {
#pragma unroll
        for (int j = 0; j < PF0_J; ++j) prefetch_j(rs, j);
    };
    auto stage_in = [&](float *dst) {
#pragma unroll
        for (int j = 0; j < PF0_J; ++j) stage_in_j(dst, j);
    };

    // ---- prologue: row sources of the first two tiles; the first tile's inputs go to LDS synchronously -------------------
    int64_t tile = blockIdx.x;
    {
        RowSrcC<NP> r;
        rs_fetch(rst, tile * CBM, r);
        rs_store(rsrc, rst, r);
        rs_fetch(rst, (tile + gridDim.x) * CBM, r);
        rs_store(rsrc + RS_STRIDE, rst, r);
    }
    __syncthreads();
    if (tile < n_tiles) prefetch(rsrc);
    if (tile < n_tiles) stage_in(buf0);
    __syncthreads();
    int slot = 0;      // row-source table slot of the CURRENT tile; (slot+1)%3: next tile; (slot+2)%3: tile after next
    int cur = 0;       // NST == 1: which buffer holds the current tile's input (they alternate); NST == 2: always buf0

    // Tile pipeline (per wave):   issue loads of tile t+1  |  MFMA(t)  |  loads(t+1) -> LDS  |  stores / atomics of tile t
    // The only vmcnt wait sits AFTER the MFMA phase and BEFORE this tile's stores are issued, so it never waits for a
    // store: the stores of tile t complete under the MFMAs of tile t+1 (vmcnt is in-order and the compiler cannot count
    // the data-dependent stores of the segmented epilogue, i.e. a wait placed after them would drain them).
    for (; tile < n_tiles; tile += gridDim.x) {
        const int64_t row0 = tile * CBM;
        const bool has_next = tile + gridDim.x < n_tiles;
        float *in0 = (NST == 1 && cur) ? buf1 : buf0;         // stage-0 input of this tile
        float *other = (NST == 1 && cur) ? buf0 : buf1;       // NST==1: next tile's input; NST==2: stage-1 input
        const int slot_n = slot == 2 ? 0 : slot + 1, slot_nn = slot_n == 2 ? 0 : slot_n + 1;
        const int *rs_next_tab = rsrc + slot_n * RS_STRIDE;
        rs_fetch(rst, (tile + 2 * (int64_t)gridDim.x) * CBM, rsn);         // row sources two tiles ahead

        float *in = in0;
#pragma unroll
        for (int s = 0; s < NST; ++s) {
            const ChainStage &st = a.st[s];
            const bool active = 32 * w < st.n_out;  // wave-uniform
            f32x16 acc[NACC];
#pragma unroll
            for (int h = 0; h < NACC; ++h)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[h][r] = 0.f;
            if (active && !(a.dbg & 2)) {
                const float *ap = in + (32 * rh + li) * pitch + lh;
                if (s == 0) {
#pragma unroll
                    for (int ch = 0; ch < CH0; ++ch) {
                        if (ch < nch[0]) {
                            // the next tile's global loads (address arithmetic + issue) ride in the shadow of this chunk's
                            // MFMAs instead of standing in front of the whole MFMA phase (one wave per SIMD: nothing else
                            // would fill the matrix pipe meanwhile)
                            // Unconditional (row sources past the end read row 0): without a branch the chunk is one
                            // basic block and the scheduler interleaves the address arithmetic with the MFMAs.
                            if (ch < PF0_J) prefetch_j(rs_next_tab, ch);
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
#pragma unroll
                                for (int h = 0; h < NACC; ++h)
                                    acc[h] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[h * 32 * pitch + ch * CHK + 2 * q], B0[ch][q], acc[h], 0, 0, 0);
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int ch = 0; ch < CH1; ++ch) {
                        if (ch < nch[1]) {
                            if (ch < PF0_J && has_next) stage_in_j(buf0, ch);   // next tile's rows -> buf0 (free since the mid barrier)
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
#pragma unroll
                                for (int h = 0; h < NACC; ++h)
                                    acc[h] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[h * 32 * pitch + ch * CHK + 2 * q], B1[ch][q], acc[h], 0, 0, 0);
                            }
                        }
                    }
                }
            } else if (s == 0) {
                // a wave without output columns (n_out <= 96: wave 3, n_out <= 64: waves 2 and 3) multiplies nothing but still owns
                // its share of the next tile's rows: without these loads its rows of the next tile kept what the buffer held --
                // the rows of the tile this workgroup handled before (wrong output rows from the second tile of a workgroup on,
                // i.e. past 128 x gridDim rows)
#pragma unroll
                for (int ch = 0; ch < CH0; ++ch)
                    if (ch < nch[0] && ch < PF0_J) prefetch_j(rs_next_tab, ch);
            }
            // epilogue.  C layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).  Straight-line for full tiles
            // (activation switch hoisted out of the 32-element loops) so the stores / LDS writes issue back to back.
            const int col = 32 * w + li;
            const bool cok = col < st.n_out;
            const float bias = e_bias[s], scale = e_scale[s], c0 = e_c0[s];
            const bool last = s == NST - 1;
            auto value = [&](int h, int r) { return fmaf(acc[h][r], scale, c0); };
            if (!last) {
                // stage output -> the next stage's LDS input tile
                float *lp = buf1 + (32 * rh + 4 * lh) * pitch + col;
                if (cok) {
                    if (st.act == 1) {
#pragma unroll
                        for (int h = 0; h < NACC; ++h)
#pragma unroll
                            for (int r = 0; r < 16; ++r) { const float y = value(h, r); lp[(h * 32 + (r & 3) + 8 * (r >> 2)) * pitch] = y > 0.f ? y : 0.f; }
                    } else {
#pragma unroll
                        for (int h = 0; h < NACC; ++h)
#pragma unroll
                            for (int r = 0; r < 16; ++r) lp[(h * 32 + (r & 3) + 8 * (r >> 2)) * pitch] = value(h, r);
                    }
                }
                lds_barrier();
                in = buf1;
                continue;
            }
            // ---- last stage ------------------------------------------------------------------------------------------
            // where the next tile's input goes: NST==1 -> the other buffer; NST==2 -> buf0 (free since the barrier above)
            float *next_in = (NST == 1) ? other : buf0;
            // next tile's prefetched rows -> LDS: everything (single-stage kernels) or what the stage-1 MFMA loop above
            // did not already interleave (two-stage kernels: column groups >= nch[1], or a wave without output columns)
            auto finish_stage_in = [&]() {
                if (!has_next) return;
                if (NST == 1) { stage_in(next_in); return; }
#pragma unroll
                for (int j = 0; j < PF0_J; ++j)
                    if (!(active && j < CH1 && j < nch[1])) stage_in_j(buf0, j);
            };
            if (STATS) {
                finish_stage_in();
                rs_store(rsrc + slot_nn * RS_STRIDE, rst, rsn);   // row sources of tile t+2
#pragma unroll
                for (int hh = 0; hh < NACC; ++hh)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t row = row0 + 32 * (rh + hh) + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const float h = acc[hh][r] + bias;
                        if (cok && row < a.m_rows) { st_sum += (double)h; st_sq += (double)h * (double)h; }
                    }
                lds_barrier();
            } else if (SEG) {
                // segmented-sum epilogue (the scatter-add of the layer, fused): activated tile -> LDS (over this stage's
                // own input tile, once every wave is done reading it), then every thread reduces one column over 32
                // consecutive rows, whose targets are sorted; a segment that lies inside the 32-row range is stored, one
                // that straddles a range boundary is added atomically (its output row was zeroed by
                // gsn_segsum_prepare_hip).  Summation order inside a segment = row order.
                lds_barrier();                                   // all waves finished the MFMAs on `in`
                float *Y = in + (32 * rh + 4 * lh) * pitch + col;
                if (cok) {
                    if (st.act == 1) {
#pragma unroll
                        for (int h = 0; h < NACC; ++h)
#pragma unroll
                            for (int r = 0; r < 16; ++r) { const float y = value(h, r); Y[(h * 32 + (r & 3) + 8 * (r >> 2)) * pitch] = y > 0.f ? y : 0.f; }
                    } else {
#pragma unroll
                        for (int h = 0; h < NACC; ++h)
#pragma unroll
                            for (int r = 0; r < 16; ++r) Y[(h * 32 + (r & 3) + 8 * (r >> 2)) * pitch] = value(h, r);
                    }
                }
                finish_stage_in();                    // (the one vmcnt wait of the tile: before any store)
                rs_store(rsrc + slot_nn * RS_STRIDE, rst, rsn);  // row sources of tile t+2
                lds_barrier();
                // thread -> column c, NRANGE consecutive ranges of SEG_ROWS rows
                constexpr int NRANGE = (CBM * 128 / CT) / SEG_ROWS;
                const int c = tid & 127;
                const int *tg = rsrc + slot * RS_STRIDE + CMAX_BLOCKS * CBM;   // this tile's targets (slot is recycled two tiles later)
                if (c < st.n_out && !(a.dbg & 4)) {
                    float *op = a.out + c;
#pragma unroll
                    for (int g = 0; g < NRANGE; ++g) {
                        const int rb = ((tid >> 7) * NRANGE + g) * SEG_ROWS;
                        const float *yp = in + rb * pitch + c;
                        const int prev_t = rb > 0 ? tg[rb - 1] : tg[CBM];
                        const int next_t = rb + SEG_ROWS < CBM ? tg[rb + SEG_ROWS] : tg[CBM + 1];
                        int curt = tg[rb];
                        bool straddle = curt == prev_t;
                        float sum = 0.f;
#pragma unroll
                        for (int r = 0; r < SEG_ROWS; ++r) {
                            const int t = tg[rb + r];
                            if (t != curt) {
                                if (curt >= 0 && !(a.dbg & 8)) {
                                    if (straddle) atomicAdd(op + (int64_t)curt * st.n_out, sum);
                                    else op[(int64_t)curt * st.n_out] = sum;
                                }
                                curt = t; sum = 0.f; straddle = false;
                            }
                            sum += yp[r * pitch];
                        }
                        if (curt >= 0 && !(a.dbg & 8)) {
                            if (straddle || curt == next_t) atomicAdd(op + (int64_t)curt * st.n_out, sum);
                            else op[(int64_t)curt * st.n_out] = sum;
                        }
                    }
                }
                // next tile's MFMAs read `next_in` (complete before the barrier above); its own pre-Y barrier orders the
                // reduction above against the following writes into `in`
            } else {
                finish_stage_in();                    // (the one vmcnt wait of the tile: before any store)
                rs_store(rsrc + slot_nn * RS_STRIDE, rst, rsn);  // row sources of tile t+2
                float *tile_out = a.out + row0 * st.n_out;                       // wave-uniform base (SGPR pair)
                const int lane_off = (32 * rh + 4 * lh) * st.n_out + col;        // 32-bit per-lane offset inside the tile
                const bool full = row0 + CBM <= a.m_rows;
                auto emit = [&](auto actf) {
                    if (a.dbg & 4) {
                    } else if (full) {
                        if (cok) {
#pragma unroll
                            for (int h = 0; h < NACC; ++h)
#pragma unroll
                                for (int r = 0; r < 16; ++r) tile_out[lane_off + (h * 32 + (r & 3) + 8 * (r >> 2)) * st.n_out] = actf(value(h, r));
                        }
                    } else {
#pragma unroll
                        for (int h = 0; h < NACC; ++h)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int dr = h * 32 + (r & 3) + 8 * (r >> 2);
                                if (cok && row0 + 32 * rh + 4 * lh + dr < a.m_rows) tile_out[lane_off + dr * st.n_out] = actf(value(h, r));
                            }
                    }
                };
                switch (st.act) {
                    case 1: emit([](float y) { return y > 0.f ? y : 0.f; }); break;
                    default: emit([](float y) { return y; }); break;
                }
                lds_barrier();   // next tile's input tile + row-source table complete
            }
        }
        slot = slot_n;
        cur ^= 1;
    }

    if (STATS) {
        const int col = 32 * w + li;
        double s = st_sum, q = st_sq;
        s += __shfl_xor(s, 32);
        q += __shfl_xor(q, 32);
        const int n_out = a.st[NST - 1].n_out;
        if (lh == 0 && col < n_out) {
            atomicAdd(&a.stats[col], s);
            atomicAdd(&a.stats[n_out + col], q);
        }
    }
}

void chain_trace(const char *kernel, const ChainArgs &a) {
    static const bool on = [] { const char *d = getenv("GSN_CHAIN_TRACE"); return d && atoi(d) != 0; }();
    if (on) fprintf(stderr, "gsn_chain_launch %s m_rows=%lld k0=%d n_out=%d stages=%d\n", kernel, (long long)a.m_rows, a.st[0].k_total, a.st[a.n_stages - 1].n_out, a.n_stages);
}

template <int NST, int CH0, int CH1, bool STATS, bool SEG>
static int launch_chain_impl(const ChainArgs &a, hipStream_t st) {
    constexpr bool SMALL = (NST == 1 && CH0 == 5);
    constexpr int NW = NST == 1 ? 8 : 4;
    constexpr int WPE = NST == 1 ? (SMALL ? 4 : 2) : 1;
    const void *fn = reinterpret_cast<const void *>(&mlp_chain_kernel<NST, CH0, CH1, STATS, WPE, SEG, NW>);
    const size_t lds = (size_t)2 * CBM * a.pitch * 4 + 3 * RS_STRIDE * 4;
    static DeviceOnce attr_set;
    const int attr_dev = current_device();
    if (!attr_set.done(attr_dev)) {
        hipError_t e0 = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e0 != hipSuccess) return set_error(GSN_E_HIP, "hipFuncSetAttribute(mlp_chain_kernel): %s", hipGetErrorString(e0));
        attr_set.mark(attr_dev);
    }
    const int64_t n_tiles = (a.m_rows + CBM - 1) / CBM;
    int per_cu = (SMALL && lds <= 76 * 1024) ? 2 : 1;
    { const char *d = getenv("GSN_CHAIN_PERCU"); if (d) per_cu = atoi(d); }
    int64_t gx = 256 * per_cu;
    if (gx > n_tiles) gx = n_tiles;
    chain_trace(STATS ? "mlp_chain_kernel(stats)" : (SEG ? "mlp_chain_kernel(seg)" : "mlp_chain_kernel"), a);
    hipLaunchKernelGGL((mlp_chain_kernel<NST, CH0, CH1, STATS, WPE, SEG, NW>), dim3((unsigned)gx), dim3(64 * NW), lds, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "mlp_chain_kernel: %s", hipGetErrorString(e));
    return GSN_OK;
}

template <int NST, int CH0, int CH1>
static int launch_chain(const ChainArgs &a, hipStream_t st) {
    if (a.stats) return launch_chain_impl<NST, CH0, CH1, true, false>(a, st);
    if (a.seg_target) return launch_chain_impl<NST, CH0, CH1, false, true>(a, st);
    return launch_chain_impl<NST, CH0, CH1, false, false>(a, st);
}

}  // namespace gsn

using namespace gsn;

extern "C" int gsn_mlp_chain_supported(int n_stages, const gsn_chain_stage *stages) {
    if (n_stages < 1 || n_stages > CMAX_STAGES || !stages) return 0;
    int nb = 0;
    for (int s = 0; s < n_stages; ++s) {
        int k_hbm = 0;
        for (int b = 0; b < stages[s].n_blocks; ++b) k_hbm += (int)stages[s].blocks[b].width;
        nb += stages[s].n_blocks;
        const int k_total = k_hbm + (s > 0 ? (int)stages[s - 1].n_out : 0);
        if (k_total > 160 || k_total < 1) return 0;
        if (stages[s].n_out > 128 || stages[s].n_out < 1) return 0;
        // elu / tanh epilogues call into the device math library; a call inside this kernel makes the register allocator
        // spill the weight fragments around it, so those activations go through gsn_linear_fwd_hip instead
        if (stages[s].act != 0 && stages[s].act != 1) return 0;
        if (s == 0 && stages[s].n_blocks < 1) return 0;
        if (s >= 1 && k_hbm > 0) return 0;   // later stages take only the previous stage's output
    }
    return nb <= CMAX_BLOCKS ? 1 : 0;
}

extern "C" int gsn_mlp_chain_fwd_hip(int64_t m_rows, int n_stages, const gsn_chain_stage *stages, const int32_t *row_perm,
                                     const int32_t *seg_target, float *out, double *stats, void *stream) {
    if (!gsn_mlp_chain_supported(n_stages, stages))
        return set_error(GSN_E_UNSUPPORTED, "gsn_mlp_chain_fwd_hip: shape outside the fused kernel (K<=160, n_out<=128, <=6 blocks); use gsn_linear_fwd_hip");
    if (!out && !stats) return set_error(GSN_E_INVALID, "gsn_mlp_chain_fwd_hip: neither out nor stats given");
    if (m_rows <= 0) return GSN_OK;
    ChainArgs a{};
    a.m_rows = m_rows; a.n_stages = n_stages; a.row_perm = row_perm; a.out = out; a.stats = stats;
    a.seg_target = stats ? nullptr : seg_target;
    { const char *d = getenv("GSN_CHAIN_DBG"); a.dbg = d ? atoi(d) : 0; }
    int nb = 0, kmax = 0;
    for (int s = 0; s < n_stages; ++s) {
        const gsn_chain_stage &g = stages[s];
        ChainStage &c = a.st[s];
        if (!g.W) return set_error(GSN_E_INVALID, "gsn_mlp_chain_fwd_hip: stage %d has no weight", s);
        if ((g.bn_scale != nullptr) != (g.bn_shift != nullptr) || (g.bn_scale != nullptr) != (g.bn_mean != nullptr))
            return set_error(GSN_E_INVALID, "gsn_mlp_chain_fwd_hip: bn_mean, bn_scale and bn_shift go together");
        if (g.act < 0 || g.act > 1) return set_error(GSN_E_INVALID, "gsn_mlp_chain_fwd_hip: act must be 0 (identity) or 1 (relu)");
        c.W = g.W; c.bias = g.bias; c.bn_mean = g.bn_mean; c.bn_scale = g.bn_scale; c.bn_shift = g.bn_shift;
        c.n_out = (int)g.n_out; c.act = g.act; c.first_block = nb; c.n_blocks = g.n_blocks;
        int k_hbm = 0;
        for (int b = 0; b < g.n_blocks; ++b) {
            if (!g.blocks[b].data || g.blocks[b].width <= 0) return set_error(GSN_E_INVALID, "gsn_mlp_chain_fwd_hip: stage %d block %d is empty", s, b);
            a.bdata[nb] = g.blocks[b].data; a.bidx[nb] = g.blocks[b].idx; a.bidx32[nb] = g.blocks[b].idx32; a.bwidth[nb] = (int)g.blocks[b].width;
            k_hbm += (int)g.blocks[b].width;
            ++nb;
        }
        c.k_hbm = k_hbm;
        c.k_total = k_hbm + (s > 0 ? (int)stages[s - 1].n_out : 0);
        if (s == 0) kmax = c.k_total;
    }
    a.n_blocks = nb;
    const int maxch = kmax <= 80 ? 5 : 10;
    // LDS row pitch: the staging writes cover whole 32-column groups of stage 0; a later stage's input tile holds the
    // previous stage's n_out columns; the segmented-sum epilogue stages n_out_last columns.  +1 / odd: conflict-free.
    int cols = (maxch * CHK + 31) / 32 * 32;
    for (int s = 1; s < n_stages; ++s) {                       // (whole CHK-column chunks are read: see chain_pipe.hip)
        const int kp = (a.st[s].k_total + CHK - 1) / CHK * CHK;
        cols = kp > cols ? kp : cols;
    }
    if (a.seg_target && a.st[n_stages - 1].n_out > cols) cols = a.st[n_stages - 1].n_out;
    a.pitch = cols | 1;
    if (a.pitch == cols) a.pitch += 2;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (n_stages == 2 && a.st[1].k_total > 8 * CHK)
        return set_error(GSN_E_UNSUPPORTED, "gsn_mlp_chain_fwd_hip: second stage wider than 128 inputs");
    if (n_stages == 2) {   // plain two-stage chains: stage-pipelined kernel (chain_pipe.hip) where it covers the shape
        int rc = launch_chain2_pipe_bf16(a, maxch, st);
        if (rc != 1) return rc;
        rc = launch_chain2_pipe(a, maxch, st);
        if (rc != 1) return rc;
    } else if (a.seg_target) {   // edge stage with the fused scatter-add: role-pipelined kernel (chain_seg.hip)
        int rc = launch_chain1_seg_bf16(a, maxch, st);
        if (rc != 1) return rc;
        rc = launch_chain1_seg(a, maxch, st);
        if (rc != 1) return rc;
    }
    if (maxch == 5) {
        if (n_stages == 1) return launch_chain<1, 5, 0>(a, st);
        return launch_chain<2, 5, 8>(a, st);
    }
    if (n_stages == 1) return launch_chain<1, 10, 0>(a, st);
    return launch_chain<2, 10, 8>(a, st);
}
