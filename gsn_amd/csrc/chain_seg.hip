// Role-split variant of the single-stage fused chain with the scatter-add epilogue (chain.hip, SEG) -- the edge stage of a
// `general` layer:   S[t] = sum over the target-sorted rows e -> t of  act(bn([blocks]_e W^T + b)).
//
// What shapes this kernel (scripts/micro/coissue.hip, profiles/r01_coissue.json): while a SIMD's matrix pipe is busy with
// v_mfma_f32_32x32x2_f32, NO other wave on that SIMD gets an instruction issued -- VALU, SALU and LDS alike, whatever its
// s_setprio (with bf16 MFMAs the same probe runs at full speed).  fp32 MFMA time and every other instruction on the SIMD
// therefore ADD UP; nothing "hides under the MFMAs" except memory latency.  Hence: (1) as few non-MFMA instructions per tile
// as possible on every wave, (2) loads issued early and consumed late, (3) no dependent load pairs inside a tile.
//   group A (TBM/8 waves): row sources of tile t+2 (raw loads, resolved at the end of the tile; the permutation entry a
//                          tile earlier than the index entry that needs it) | gathers of tile t+1 (float4 where the block
//                          widths allow) issued in front of the MFMAs on IN[t&1] | max(acc, 0) -> Y[t&1] (the BN scale is
//                          folded into the weight registers, the accumulators start at the folded bias) | rows of tile t+1 ->
//                          IN[(t+1)&1];
//   group B (TBM/8 waves): segmented sum of Y[(t-1)&1]: the row targets come through the scalar cache (they are wave-uniform),
//                          the walk over a 16-row range is scalar control flow; a lane executes 16 LDS reads, 16 adds and
//                          its stores / atomics per tile.
// B works one tile behind A; ONE workgroup barrier per tile hands Y over.  TBM = 32: 8 waves, 61 KiB of LDS, two workgroups
// per CU.  NKS = k-steps (MFMAs per 32 x 32 tile, 2 k each) the kernel is compiled for: K = 72 runs 36 steps instead of the
// 40 of five whole 16-k chunks (a run-time bound inside the MFMA sequence would put branches between the MFMAs).
// GSN_SEG_PROF=1 prints the per-phase cycle counts (s_memtime) of workgroup 0.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "chain_common.h"

namespace gsn {

template <int NKS, int TBM, bool PROF, bool VEC4>
__global__ __launch_bounds__(TBM * 16) __attribute__((amdgpu_waves_per_eu(4, 4))) void mlp_chain1_seg_kernel(ChainArgs a, int pin, int py, unsigned long long *prof, int prio) {
    constexpr int GT = TBM * 8;                     // threads per group
    constexpr int RSTEP = GT / 32;
    constexpr int NROW = TBM / RSTEP;               // = 4
    constexpr int PF0_J = (NKS * 2 + 31) / 32;      // 32-column groups of the input
    constexpr int NSLOT = 4;
    constexpr int RSS = CMAX_BLOCKS * TBM;          // ring slot: row sources per block
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // buffers as offsets into `lds` (a pointer picked from an array of buffer pointers loses its LDS address space)
    const int in_sz = TBM * pin, y_sz = TBM * py;
    auto in_tile = [&](int64_t i) { return lds + (int)(i & 1) * in_sz; };
    auto y_tile = [&](int64_t i) { return lds + 2 * in_sz + (int)(i & 1) * y_sz; };
    int *rsrc = reinterpret_cast<int *>(lds + 2 * in_sz + 2 * y_sz);     // [NSLOT][RSS]

    const int tid = threadIdx.x;
    const bool grp_b = tid >= GT;                   // wave-uniform
    const int t = tid & (GT - 1);
    const int64_t n_tiles = (a.m_rows + TBM - 1) / TBM;
    const int64_t first = blockIdx.x;
    const int64_t n_iter = first < n_tiles ? (n_tiles - first + gridDim.x - 1) / gridDim.x : 0;
    const ChainStage &st = a.st[0];

    for (int i = tid; i < 2 * in_sz + 2 * y_sz; i += 2 * GT) lds[i] = 0.f;      // padded columns must hold finite values
    __syncthreads();

    if (!grp_b) {
        // =============================================================================================================
        // group A: gathers + MFMAs + activated tile -> LDS.  Wave w8: output columns 32 (w8 & 3) .., tile rows 32 (w8 >> 2) ..
        // =============================================================================================================
        const int lane = t & 63, w8 = t >> 6;
        const int w = w8 & 3, rh = w8 >> 2;
        const int li = lane & 31, lh = lane >> 5;
        const int col = 32 * w + li;
        const bool cok = col < st.n_out;
        const bool active = 32 * w < st.n_out;
        // y = scale * (x W^T + bias - mean) + shift = x (scale W)^T + c0: the BN scale is folded into this lane's weight column
        // and the accumulators start at c0, so the epilogue is one max per element.  fp32 MFMAs and VALU instructions do not
        // overlap on a SIMD (measured: VALU work of ANY wave runs ~4x slower while the SIMD's fp32 MFMA sequence is saturated,
        // and the fp32 matrix peak equals the packed-fp32 vector peak), so every VALU instruction per tile is paid in full.
        const float bias = (cok && st.bias) ? st.bias[col] : 0.f;
        float scale = 1.f, c0 = bias;
        if (cok && st.bn_scale) { scale = st.bn_scale[col]; c0 = (bias - st.bn_mean[col]) * scale + st.bn_shift[col]; }
        float B0[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int k = 2 * ks + lh;
            B0[ks] = (k < st.k_total && cok) ? st.W[(int64_t)col * st.k_total + k] * scale : 0.f;
        }
        // staging map.  Scalar: thread -> column kc0 (+32j) of rows r0 + RSTEP i.  VEC4 (every block width a multiple of 4 floats,
        // 16-byte aligned): thread -> columns 4 qc .. 4 qc + 3 (+32j) of ONE row: a third of the address arithmetic and a quarter
        // of the load instructions per tile (every VALU instruction is paid in full next to fp32 MFMAs).
        const int kc0 = VEC4 ? 4 * (t & 7) : (t & 31), r0 = VEC4 ? (t >> 3) : (t >> 5);
        ColMap cm0[PF0_J];
#pragma unroll
        for (int j = 0; j < PF0_J; ++j) {
            cm0[j] = col_map(a, 0, kc0 + 32 * j);
            cm0[j].rsoff = cm0[j].rsoff / CBM * TBM;
        }
        float pf0[PF0_J][NROW];
        // Row sources: thread t resolves tile row t % TBM of input block t / TBM.  Every load below is unconditional from a
        // valid address and its value stays RAW in a register until the end of the tile (any select / sign extension on a
        // just-loaded value makes the compiler wait for it on the spot: a full memory latency at the top of every tile), and
        // the dependent pair perm[row] -> idx[perm[row]] is split over two tiles: permutation entries run three tiles ahead.
        const int rs_r = t & (TBM - 1), rs_b = t / TBM;
        const bool rs_on = rs_b < a.n_blocks;
        const int32_t *rs_ip = nullptr;
#pragma unroll
        for (int q = 0; q < CMAX_BLOCKS; ++q)
            if (q == rs_b) rs_ip = a.bidx32[q];
        const bool rs_idx = rs_on && rs_ip != nullptr;
        if (!rs_idx) rs_ip = a.seg_target;                    // any readable array of m_rows ints
        const bool has_perm = a.row_perm != nullptr;
        const int32_t *permp = has_perm ? a.row_perm : a.seg_target;
        const int m_rows = (int)a.m_rows, last_row = m_rows - 1;
        const int gstep = (int)gridDim.x * TBM;
        auto clampr = [&](int row) { return row < last_row ? row : last_row; };
        // waves whose threads all sit past the last input block skip the row-source work (wave-uniform branch)
        const bool rs_wave = __builtin_amdgcn_readfirstlane(rs_b) < a.n_blocks;
        int rs_lg = 0, raw_idx = 0, raw_perm = 0;
        auto rs_issue = [&](int row0, int lg_raw) {           // row0: first row of the tile whose sources are resolved now
            const int grow = clampr(row0 + rs_r);
            rs_lg = has_perm ? lg_raw : grow;
            raw_idx = rs_ip[rs_lg];
        };
        auto rs_commit = [&](int *dst, int row0) {
            const bool ok = row0 + rs_r < m_rows;
            if (rs_on) dst[rs_b * TBM + rs_r] = ok ? (rs_idx ? raw_idx : rs_lg) : -1;
        };
        auto prefetch_j = [&](const int *rs, int j) {
            if (VEC4) {
                const int sr = rs[cm0[j].rsoff + r0];
                const float4 v = *reinterpret_cast<const float4 *>(cm0[j].base + (int64_t)(sr < 0 ? 0 : sr) * cm0[j].bw);
                pf0[j][0] = v.x; pf0[j][1] = v.y; pf0[j][2] = v.z; pf0[j][3] = v.w;
                return;
            }
#pragma unroll
            for (int i = 0; i < NROW; ++i) {
                const int sr = rs[cm0[j].rsoff + r0 + RSTEP * i];
                pf0[j][i] = cm0[j].base[(int64_t)(sr < 0 ? 0 : sr) * cm0[j].bw];
            }
        };
        auto stage_in = [&](float *dst) {
#pragma unroll
            for (int j = 0; j < PF0_J; ++j)
#pragma unroll
                for (int i = 0; i < NROW; ++i) {
                    if (VEC4) dst[r0 * pin + kc0 + 32 * j + i] = pf0[j][i];
                    else dst[(r0 + RSTEP * i) * pin + kc0 + 32 * j] = pf0[j][i];
                }
        };
        {
            const int row0 = (int)first * TBM;
            rs_issue(row0, permp[clampr(row0 + rs_r)]);
            rs_commit(rsrc, row0);
            rs_issue(row0 + gstep, permp[clampr(row0 + gstep + rs_r)]);
            rs_commit(rsrc + RSS, row0 + gstep);
            raw_perm = permp[clampr(row0 + 2 * gstep + rs_r)];
        }
        lds_barrier();
        if (n_iter > 0) {
#pragma unroll
            for (int j = 0; j < PF0_J; ++j) prefetch_j(rsrc, j);
            stage_in(in_tile(0));
        }
        lds_barrier();
        unsigned long long pc[6] = {0, 0, 0, 0, 0, 0};
        auto clk = [&]() -> unsigned long long {
            if (!PROF) return 0;
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long v = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
            return v;
        };
        for (int64_t i = 0; i <= n_iter; ++i) {
            const unsigned long long t0 = clk();
            unsigned long long t1 = t0, t2 = t0, t3 = t0, t4 = t0;
            if (i < n_iter) {
                const int64_t tile = first + i * gridDim.x;
                const int *rs_next = rsrc + (int)((i + 1) & (NSLOT - 1)) * RSS;
                const int row2 = (int)tile * TBM + 2 * gstep;               // the tile after next
                if (rs_wave) {
                    rs_issue(row2, raw_perm);
                    raw_perm = permp[clampr(row2 + gstep + rs_r)];
                }
                const float *in = in_tile(i);
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = c0;
                t1 = clk();
                if (active && !(a.dbg & 2)) {
                    const float *ap = in + (32 * rh + li) * pin + lh;
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        // next tile's gathers under these MFMAs (VEC4: three loads per thread, all issued up front to give them
                        // the whole matrix phase to land)
                        if (VEC4) {
                            if (ks == 0) {
#pragma unroll
                                for (int j = 0; j < PF0_J; ++j) prefetch_j(rs_next, j);
                            }
                        } else if (ks % 8 == 0 && ks / 8 < PF0_J) prefetch_j(rs_next, ks / 8);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * ks], B0[ks], acc, 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < PF0_J; ++j) prefetch_j(rs_next, j);
                }
                t2 = clk();
                if (prio & 2) __builtin_amdgcn_s_setprio(2);
                // activated tile -> Y[i&1]  (group B finished reading it one barrier ago)
                float *lp = y_tile(i) + (32 * rh + 4 * lh) * py + col;
                if (cok) {
                    if (st.act == 1) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) lp[((r & 3) + 8 * (r >> 2)) * py] = fmaxf(acc[r], 0.f);
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) lp[((r & 3) + 8 * (r >> 2)) * py] = acc[r];
                    }
                }
                t3 = clk();
                stage_in(in_tile(i + 1));                                   // (waits for the gathers; no stores in this group)
                if (rs_wave) rs_commit(rsrc + (int)((i + 2) & (NSLOT - 1)) * RSS, row2);
                t4 = clk();
            }
            lds_barrier();
            if (prio & 2) __builtin_amdgcn_s_setprio(0);
            if (PROF) {
                const unsigned long long t5 = clk();
                pc[0] += t1 - t0; pc[1] += t2 - t1; pc[2] += t3 - t2; pc[3] += t4 - t3; pc[4] += t5 - t4; pc[5] += 1;
            }
        }
        if (PROF && prof && lane == 0 && blockIdx.x == 0) {
            unsigned long long *o = prof + w8 * 6;
            for (int q = 0; q < 6; ++q) o[q] = pc[q];
        }
        return;
    }

    // =================================================================================================================
    // group B: segmented sum of the activated tile, one tile behind group A.  Thread -> column c, one range of SEG_ROWS
    // target-sorted rows; a segment inside a range is stored, one that straddles a range boundary is added atomically (its
    // output row was zeroed by gsn_segsum_prepare_hip).  Summation order inside a segment = row order.
    // =================================================================================================================
    // The row targets are wave-uniform, so they come through the scalar cache (constant address space loads of seg_target)
    // and the walk over the range is scalar control flow: per tile a lane executes 16 adds, the LDS reads and its stores.
    static_assert(TBM * 128 / GT == SEG_ROWS, "one range per thread");
    typedef const __attribute__((address_space(4))) int cint;
    cint *segc = (cint *)a.seg_target;
    const int c = t & 127;
    const bool cok = c < st.n_out && !(a.dbg & 4);
    const int rb = __builtin_amdgcn_readfirstlane(t >> 7) * SEG_ROWS;
    const int m_rows = (int)a.m_rows;
    if (prio & 1) __builtin_amdgcn_s_setprio(3);       // few instructions per tile: never wait behind the matrix waves' issue
    lds_barrier();
    lds_barrier();
    unsigned long long pb[4] = {0, 0, 0, 0};
    auto clkb = [&]() -> unsigned long long {
        if (!PROF) return 0;
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long v = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
        return v;
    };
    for (int64_t i = 0; i <= n_iter; ++i) {
        const unsigned long long u0 = clkb();
        unsigned long long u1 = u0;
        if (i > 0 && cok) {
            const int g0 = (int)(first + (i - 1) * gridDim.x) * TBM + rb;      // first row of this wave's range (uniform)
            const float *yp = y_tile(i - 1) + rb * py + c;
            int tv[SEG_ROWS];
            if (g0 + SEG_ROWS <= m_rows) {
#pragma unroll
                for (int r = 0; r < SEG_ROWS; ++r) tv[r] = segc[g0 + r];
            } else {
#pragma unroll
                for (int r = 0; r < SEG_ROWS; ++r) tv[r] = g0 + r < m_rows ? segc[g0 + r] : -1;
            }
            const int prev_t = g0 > 0 && g0 - 1 < m_rows ? segc[g0 - 1] : -2;
            const int next_t = g0 + SEG_ROWS < m_rows ? segc[g0 + SEG_ROWS] : -2;
            float yv[SEG_ROWS];
#pragma unroll
            for (int r = 0; r < SEG_ROWS; ++r) yv[r] = yp[r * py];
            if (PROF) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); u1 = clkb(); }
            int curt = tv[0];
            bool straddle = curt == prev_t;
            float sum = 0.f;
            auto flush = [&](bool atomic) {
                if (curt < 0 || (a.dbg & 8)) return;
                float *rowp = a.out + (int64_t)curt * st.n_out;              // uniform base; the lane adds its column
                if (atomic) atomicAdd(rowp + c, sum);
                else rowp[c] = sum;
            };
#pragma unroll
            for (int r = 0; r < SEG_ROWS; ++r) {
                const int tt = tv[r];
                if (tt != curt) {
                    flush(straddle);
                    curt = tt; sum = 0.f; straddle = false;
                }
                sum += yv[r];
            }
            flush(straddle || curt == next_t);
        }
        const unsigned long long u2 = clkb();
        lds_barrier();
        if (PROF) { const unsigned long long u3 = clkb(); pb[0] += u1 - u0; pb[1] += u2 - u1; pb[2] += u3 - u2; pb[3] += 1; }
    }
    if (PROF && prof && (t & 63) == 0 && blockIdx.x == 0) {
        unsigned long long *o = prof + (8 + (t >> 6)) * 6;
        for (int q = 0; q < 4; ++q) o[q] = pb[q];
    }
}

template <int NKS, int TBM, bool PROF, bool VEC4>
static int launch_seg_impl(const ChainArgs &a, hipStream_t st) {
    const void *fn = reinterpret_cast<const void *>(&mlp_chain1_seg_kernel<NKS, TBM, PROF, VEC4>);
    static DeviceOnce attr_set;
    const int attr_dev = current_device();
    if (!attr_set.done(attr_dev)) {
        hipError_t e0 = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e0 != hipSuccess) return set_error(GSN_E_HIP, "hipFuncSetAttribute(mlp_chain1_seg_kernel): %s", hipGetErrorString(e0));
        attr_set.mark(attr_dev);
    }
    const int pin = (((NKS * 2 + 31) / 32) * 32) | 1;
    int py = a.st[0].n_out | 1;
    if (py == a.st[0].n_out) py += 2;
    const size_t lds = ((size_t)2 * TBM * pin + (size_t)2 * TBM * py + (size_t)4 * CMAX_BLOCKS * TBM) * 4;
    if (lds > 160 * 1024) return 1;
    unsigned long long *prof = nullptr;
    int prio = 3;
    { const char *d = getenv("GSN_SEG_PRIO"); if (d) prio = atoi(d); }
    if (PROF) { (void)hipMalloc(&prof, 2 * 8 * 6 * 8); (void)hipMemset(prof, 0, 2 * 8 * 6 * 8); }
    const int64_t n_tiles = (a.m_rows + TBM - 1) / TBM;
    int64_t gx = 256 * (lds <= 78 * 1024 ? 2 : 1);
    if (gx > n_tiles) gx = n_tiles;
    chain_trace("mlp_chain1_seg_kernel", a);
    hipLaunchKernelGGL((mlp_chain1_seg_kernel<NKS, TBM, PROF, VEC4>), dim3((unsigned)gx), dim3(TBM * 16), lds, st, a, pin, py, prof, prio);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(GSN_E_HIP, "mlp_chain1_seg_kernel: %s", hipGetErrorString(e));
    if (PROF) {
        unsigned long long h[2 * 8 * 6];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost);
        (void)hipFree(prof);
        static int shown = 0;
        if (shown++ % 8 == 7)
            for (int w = 0; w < TBM / 8; ++w) {
                const unsigned long long *o = h + w * 6, *q = h + (8 + w) * 6;
                if (o[5]) fprintf(stderr, "segprof A%d tiles %llu: top %llu mfma %llu ywrite %llu stage %llu barrier %llu | B%d: lds %llu walk %llu barrier %llu (cycles per tile)\n", w, o[5],
                                  o[0] / o[5], o[1] / o[5], o[2] / o[5], o[3] / o[5], o[4] / o[5], w, q[0] / q[3], q[1] / q[3], q[2] / q[3]);
            }
    }
    return GSN_OK;
}

template <int TBM, bool VEC4>
static int launch_seg_k(const ChainArgs &a, hipStream_t st) {
    const int nks = (a.st[0].k_total + 1) / 2;
    { const char *d = getenv("GSN_SEG_PROF"); if (d && atoi(d) && nks == 36) return launch_seg_impl<36, TBM, true, VEC4>(a, st); }
    if (nks <= 24) return launch_seg_impl<24, TBM, false, VEC4>(a, st);
    if (nks <= 32) return launch_seg_impl<32, TBM, false, VEC4>(a, st);
    if (nks <= 36) return launch_seg_impl<36, TBM, false, VEC4>(a, st);
    return 1;   // K = 73..80: 40 weight registers per lane spill at 128 registers; chain.hip's kernel covers it
}

// Returns GSN_OK after launching, or 1 if this shape is not covered (the caller then uses chain.hip's kernel).
int launch_chain1_seg(const ChainArgs &a, int maxch, hipStream_t st) {
    if (a.n_stages != 1 || a.stats || !a.seg_target || maxch != 5) return 1;
    if (a.m_rows > (int64_t)2000000000) return 1;                       // 32-bit row arithmetic
    for (int b = 0; b < a.n_blocks; ++b)
        if (a.bidx[b] && !a.bidx32[b]) return 1;                        // int64 row indices: chain.hip's kernel
    int tbm = 32;
    { const char *d = getenv("GSN_CHAIN_SEGPIPE"); if (d) tbm = atoi(d); }
    bool vec4 = true;                                                   // float4 gathers: widths and bases 16-byte aligned
    for (int b = 0; b < a.n_blocks; ++b)
        if ((a.bwidth[b] & 3) || (reinterpret_cast<uintptr_t>(a.bdata[b]) & 15)) vec4 = false;
    { const char *d = getenv("GSN_SEG_VEC4"); if (d && atoi(d) == 0) vec4 = false; }
    if (tbm == 64) return vec4 ? launch_seg_k<64, true>(a, st) : launch_seg_k<64, false>(a, st);
    if (tbm == 32) return vec4 ? launch_seg_k<32, true>(a, st) : launch_seg_k<32, false>(a, st);
    return 1;
}

}  // namespace gsn
