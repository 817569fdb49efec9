// Device helpers shared by the register-resident layer kernels (layer_rr.hip, layer_w.hip): MFMA fragment index maps, fp16 / bf16
// plane splits, the LDS fragment read, the incidence operand.  Not part of the ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "chain_common.h"

namespace gsn {

typedef _Float16 rr_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 rr_h2 __attribute__((ext_vector_type(2)));
typedef __bf16 rr_b8 __attribute__((ext_vector_type(8)));
typedef __bf16 rr_b2 __attribute__((ext_vector_type(2)));
typedef float rr_f2 __attribute__((ext_vector_type(2)));
typedef unsigned rr_u4 __attribute__((ext_vector_type(4)));
typedef float rr_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int rr_crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// feature held by k-slot s (lane half h) of chunk c of an operand that is made from accumulator tiles of 32 features
__device__ __host__ __forceinline__ int rr_kslot_feature(int c, int h, int s) { return 32 * (c >> 1) + 16 * (c & 1) + 8 * (s >> 2) + 4 * h + (s & 3); }

__device__ __forceinline__ unsigned rr_pack_h2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(rr_f2{a, b}, rr_h2)); }
__device__ __forceinline__ rr_f2 rr_unpack_h2(unsigned p) { return __builtin_convertvector(__builtin_bit_cast(rr_h2, p), rr_f2); }
__device__ __forceinline__ unsigned rr_pack_b2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(rr_f2{a, b}, rr_b2)); }

// fp32 - (one half of a packed fp16 pair), one instruction: the compiler converts the pair back (two v_cvt) and subtracts packed
__device__ __forceinline__ float rr_res_lo(float a, unsigned pair) { float r; asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(pair)); return r; }
__device__ __forceinline__ float rr_res_hi(float a, unsigned pair) { float r; asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(pair)); return r; }
// a * s - half of the pair (one rounding)
__device__ __forceinline__ float rr_res_lo_s(float a, float s, unsigned pair) { float r; asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(s), "v"(pair)); return r; }
__device__ __forceinline__ float rr_res_hi_s(float a, float s, unsigned pair) { float r; asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(s), "v"(pair)); return r; }
// max(x, lo) with lo in a scalar register, one instruction (fmaxf / fmed3 on an MFMA result get a canonicalising v_max in front)
__device__ __forceinline__ float rr_max(float x, float lo) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "s"(lo)); return r; }

// The helpers above are opaque to the compiler's hazard recogniser: it pads an MFMA result -> VALU read with the wait states the
// hardware needs (it has no interlock there) only for instructions it knows.  Where such a helper is the FIRST reader of an
// accumulator, this goes between the last product and the read (8-pass MFMA: 12 states; fences keep both sides in place).
__device__ __forceinline__ void rr_mfma_settle() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 1" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// fp32 fma -> one half of a packed fp16 pair, one instruction each (v_fma_mixlo_f16 writes bits 15:0, v_fma_mixhi_f16 bits 31:16, the
// other half is kept): a pair of planes of (a s, b s) is FOUR instructions -- the scale rides the fma, the residual a s - hi is formed
// and rounded in one go -- against two multiplies, a packed convert, two v_fma_mix_f32 and another packed convert (r03).  Same roundings,
// same bits (a s is exact: s is a power of two).
__device__ __forceinline__ unsigned rr_mix_pack_s(float a, float b, float s) {
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(r) : "v"(b), "v"(s));
    return r;
}
__device__ __forceinline__ unsigned rr_mix_res_s(float a, float b, float s, unsigned pair) {
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(s), "v"(pair));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(r) : "v"(b), "v"(s), "v"(pair));
    return r;
}
__device__ __forceinline__ unsigned rr_mix_res(float a, float b, unsigned pair) {
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(pair));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(r) : "v"(b), "v"(pair));
    return r;
}
// (a, b) -> packed fp16 high parts and low parts (round to nearest)
__device__ __forceinline__ void rr_split2(float a, float b, unsigned &hi, unsigned &lo) {
    hi = rr_pack_h2(a, b);
#ifdef RR_SPLIT_R03
    lo = rr_pack_h2(rr_res_lo(a, hi), rr_res_hi(b, hi));
#else
    lo = rr_mix_res(a, b, hi);
#endif
}
// the same of (a s, b s)
__device__ __forceinline__ void rr_split2s(float a, float b, float s, unsigned &hi, unsigned &lo) {
#ifdef RR_SPLIT_R03
    hi = rr_pack_h2(a * s, b * s);
    lo = rr_pack_h2(rr_res_lo_s(a, s, hi), rr_res_hi_s(b, s, hi));
#else
    hi = rr_mix_pack_s(a, b, s);
    lo = rr_mix_res_s(a, b, s, hi);
#endif
}
// ---- the same splits two pairs at a time, compiler-visible except for the four residual instructions (layer_rp.hip).  The hazard
// recogniser pads every inline-asm result that the NEXT instruction reads with an s_nop (it cannot know whether the asm wrote a
// partial register), and it does not see an MFMA result -> asm read hazard at all: here the first readers of an accumulator are plain
// C (integer max, packed multiply), the asm statements come in an order in which no two neighbours depend on each other.
__device__ __forceinline__ float rr_imax(float x, int lo_i) { return __int_as_float(max(__float_as_int(x), lo_i)); }   // relu: lo_i = 0, identity: INT_MIN
__device__ __forceinline__ void rr_res4(float a0, float b0, float a1, float b1, unsigned hi0, unsigned hi1, unsigned &lo0, unsigned &lo1) {
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(lo0) : "v"(a0), "v"(hi0));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(lo1) : "v"(a1), "v"(hi1));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo0) : "v"(b0), "v"(hi0));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo1) : "v"(b1), "v"(hi1));
}
__device__ __forceinline__ void rr_split4(float a0, float b0, float a1, float b1, unsigned &hi0, unsigned &lo0, unsigned &hi1, unsigned &lo1) {
    hi0 = rr_pack_h2(a0, b0); hi1 = rr_pack_h2(a1, b1);
    rr_res4(a0, b0, a1, b1, hi0, hi1, lo0, lo1);
}
__device__ __forceinline__ void rr_split4s(float a0, float b0, float a1, float b1, float s, unsigned &hi0, unsigned &lo0, unsigned &hi1, unsigned &lo1) {
    const rr_f2 s2 = rr_f2{s, s};
    const rr_f2 p0 = rr_f2{a0, b0} * s2, p1 = rr_f2{a1, b1} * s2;           // (exact: s is a power of two)
    hi0 = rr_pack_h2(p0[0], p0[1]); hi1 = rr_pack_h2(p1[0], p1[1]);
    rr_res4(p0[0], p0[1], p1[0], p1[1], hi0, hi1, lo0, lo1);
}
// the same, also returning the OR of the residuals' bits (zero: both exact in fp16)
__device__ __forceinline__ void rr_split2r(float a, float b, unsigned &hi, unsigned &res) {
    hi = rr_pack_h2(a, b);
    res |= __float_as_uint(rr_res_lo(a, hi)) | __float_as_uint(rr_res_hi(b, hi));
}
// (a, b) -> three packed bf16 planes, a + b exactly (8 + 8 + 8 bits, round to nearest each)
__device__ __forceinline__ void rr_split3b(float a, float b, unsigned &p1, unsigned &p2, unsigned &p3) {
    p1 = rr_pack_b2(a, b);
    const float ra = a - __uint_as_float(p1 << 16), rb = b - __uint_as_float(p1 & 0xffff0000u);
    p2 = rr_pack_b2(ra, rb);
    const float rra = ra - __uint_as_float(p2 << 16), rrb = rb - __uint_as_float(p2 & 0xffff0000u);
    p3 = rr_pack_b2(rra, rrb);
}

__device__ __forceinline__ unsigned rr_xhalf_max(unsigned v) {     // max over the two lanes l, l ^ 32
    const auto sw = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return max((unsigned)sw[0], (unsigned)sw[1]);
}
__device__ __forceinline__ unsigned rr_xhalf_or(unsigned v) {
    const auto sw = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (unsigned)sw[0] | (unsigned)sw[1];
}
__device__ __forceinline__ float rr_pow2(int field) {               // 2^(field - 127), field clamped to a normal float
    field = field < 1 ? 1 : (field > 254 ? 254 : field);
    return __uint_as_float((unsigned)field << 23);
}

// one prepared fragment from LDS: lane l reads 16 bytes at fragment * 1024 + 16 l.  Three base registers 64 KiB apart and a 16-bit
// immediate offset -- left to itself the compiler makes one address register per fragment beyond the first 64 KiB, hoists the
// 90 of them out of the tile loop and spills them (their reloads wait for every load in flight).
typedef const __attribute__((address_space(3))) rr_u4 *rr_ldsp;
__device__ __forceinline__ const void *rr_lds_generic(unsigned lds_addr) { return (const void *)reinterpret_cast<const __attribute__((address_space(3))) unsigned char *>(lds_addr); }
__device__ __forceinline__ rr_u4 rr_lds_frag(const unsigned (&base)[3], int f) {
    const int byte = f * 1024;
#ifdef RR_ABL_NOLDS
    return rr_u4{base[0], (unsigned)f, base[1], 0x3c003c00u};
#else
    return *reinterpret_cast<rr_ldsp>(base[byte >> 16] + (unsigned)(byte & 0xffff));
#endif
}

// scheduling fence between the unrolled groups of products (the compiler otherwise hoists every fragment read of a stage to its top)
#ifdef RR_NOSB
#define RR_SB()
#else
#define RR_SB() __builtin_amdgcn_sched_barrier(0)
#endif
#ifndef RR_PD
#define RR_PD 4
#endif
// prescribes the issue order inside one fenced group: NM x (one MFMA, then NV vector instructions).  A wave issues in order, and an
// MFMA behind an MFMA waits 32 cycles for the pipe: only vector work placed BETWEEN two MFMAs runs under the first one
// (scripts/micro/issue_mix.hip: MFMA + 4 v_fma = 34 cycles, + 8 = 52; a second wave's vector stream beside an MFMA stream: both ~1.6x slower).
#ifdef RR_NOMIX
#define RR_MIX(NM, NV)
#else
#define RR_MIX(NM, NV) _Pragma("unroll") for (int mix_q = 0; mix_q < (NM); ++mix_q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, (NV), 0); }
#endif
// diagnostic builds (scripts/rr_variant.sh): RR_ABL_NOMFMA / NOLDS / NOSTREAM / NOGATHER / NOSTORE switch one kind of work off (results are
// then garbage) to see what the kernel's time is sensitive to
#ifdef RR_ABL_NOMFMA
#define RR_MFH(A, B, C) asm volatile("" :: "v"(A), "v"(B))
#else
#define RR_MFH(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(rr_h8, A), __builtin_bit_cast(rr_h8, B), C, 0, 0, 0)
#endif
#define RR_MFB(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(rr_b8, A), __builtin_bit_cast(rr_b8, B), C, 0, 0, 0)

// the incidence operand of one 32-row edge block: lane (t, h), k-slot s of chunk cc <-> block row 16 cc + 8 (s >> 2) + 4 h + (s & 3);
// 2.0 (0x4000 as fp16 and as bf16: one bit) where that row is an in-edge of target t.  `bm` = the lane's rows as a bit mask.
__device__ __forceinline__ unsigned rr_edge_mask(int pt, int pt1, int ebase) {
    int lo = pt - ebase, hi = pt1 - ebase;
    lo = lo < 0 ? 0 : (lo > 32 ? 32 : lo);
    hi = hi < 0 ? 0 : (hi > 32 ? 32 : hi);
    const unsigned long long m = ((1ull << hi) - 1ull) ^ ((1ull << lo) - 1ull);
    return (unsigned)m;
}
__device__ __forceinline__ void rr_incidence(unsigned bm, int lh, rr_u4 (&M)[2]) {
    const unsigned b = bm >> (4 * lh);
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
        unsigned v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k0 = 16 * cc + 8 * (q >> 1) + 2 * (q & 1);
            const unsigned t0 = (k0 <= 14 ? b << (14 - k0) : b >> (k0 - 14)) & 0x4000u;
            v[q] = ((b << (29 - k0)) & 0x40000000u) | t0;
        }
        M[cc] = rr_u4{v[0], v[1], v[2], v[3]};
    }
}

}  // namespace gsn
