// Internal declarations shared by the translation units of libgsn_hip.so (not part of the ABI).
#pragma once

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <atomic>

#include "../../include/gsn_abi.h"

namespace gsn {

// packed counting-plan table (uint32 words) -- written by gsn_count_plan_build, read by the kernel
//   header : [0] magic  [1] mode  [2] induced  [3] n_plans  [4] n_cols  [5] kmax  [6] directed_orbits | directed<<1  [7] plans_off
//   col_ptr: [8 .. 8+n_cols]  plans col_ptr[c]..col_ptr[c+1] all write output column c (plans are sorted by column)
//   col_order: [9+n_cols .. 9+2 n_cols)  the columns by falling estimated search cost (the order the kernel hands cells out in)
//   plan p : at plans_off + p*plan_stride(header[6]): [0] k | n_fixed<<8 | out_col<<16     [1] pattern | root_a<<16 | min_degree<<20 | root_b<<24 | tail_mode<<28 (closed form of the last two levels: patterns.cpp)
//            [2+l] level l: adj_mask | nonadj_mask<<8 | gt_mask<<16 | lt_mask<<24   (bit j = earlier level j)
//            [2+KMAX + l/4] byte l%4: distance constraint of level l:  j | r<<3  (r = 0 none, 2 or 3): the image of level
//                         l must lie within r hops of the image of level j (r = their distance in the pattern)
constexpr uint32_t PLAN_MAGIC = 0x47534e31u;  // 'GSN1'
constexpr int PLAN_HEADER_WORDS = 8;
//            directed patterns only: [PLAN_STRIDE_WORDS + l] level l: in_adj_mask | in_nonadj_mask<<8 -- the image must (not) be
//                         an IN-neighbour of f_j (pattern arc level l -> level j); [2+l] then speaks of OUT-neighbours of f_j
constexpr int PLAN_BALL_WORDS = (GSN_KMAX + 3) / 4;
constexpr int PLAN_STRIDE_WORDS = 2 + GSN_KMAX + PLAN_BALL_WORDS;
// (the partial map of a search holds levels 0 .. 7 -- 8 bits each, or 16 in two registers: count_core.h FVec -- and the masks of a level name
//  EARLIER levels by bit: a ninth level (GSN_KMAX = 9) needs neither a ninth slot nor a ninth bit, because the last level of a plan is never
//  enumerated: it is counted by popcount or by a closed form)
static_assert(GSN_KMAX <= 9, "level masks are 8 bits wide: the last level may be the ninth, no more");
constexpr int PLAN_STRIDE_DIRECTED = PLAN_STRIDE_WORDS + GSN_KMAX;
inline int plan_stride(uint32_t flags) { return (flags & 2u) ? PLAN_STRIDE_DIRECTED : PLAN_STRIDE_WORDS; }

int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// "done once per DEVICE" flag for per-device kernel attributes (hipFuncSetAttribute applies to the current device only; a
// process may drive several GPUs).  Lock-free: a racing second setter merely repeats the idempotent call.
struct DeviceOnce {
    std::atomic<unsigned long long> mask{0};
    bool done(int dev) const { return dev >= 0 && dev < 64 && ((mask.load(std::memory_order_acquire) >> dev) & 1ull); }
    void mark(int dev) { if (dev >= 0 && dev < 64) mask.fetch_or(1ull << dev, std::memory_order_release); }
};
int current_device();   // hipGetDevice, -1 on error (abi.cpp)
int64_t gsn_wgrad_slab_rows(int64_t m_rows, int64_t tiles, int64_t wg_target);   // backward.hip: rows per slab of a weight-gradient call

}  // namespace gsn
