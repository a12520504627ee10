// In-kernel stamps (s_memtime / s_memrealtime) for the diagnostic builds under tools/diag/: a kernel source compiled with -DBR_STAMPS
// records where a wave spends its cycles.  No stamp exists in the product build (the macros expand to nothing).
#pragma once
#include "common.h"

namespace br {

#ifdef BR_STAMPS
constexpr int kStampSlots = 12;
__device__ unsigned long long* g_stamp_buf = nullptr;
#define BR_STAMP_DECL unsigned long long stamps_[kStampSlots] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define BR_STAMP(i)                                                                             \
  do {                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    unsigned long long t_;                                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
    if (stamps_[i] == 0) stamps_[i] = t_;                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  } while (0)
#define BR_STAMP_RT(i)                                                                          \
  do {                                                                                          \
    unsigned long long t_;                                                                      \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
    stamps_[i] = t_;                                                                            \
  } while (0)
#define BR_STAMP_FLUSH(widx)                                                                    \
  do {                                                                                          \
    if ((threadIdx.x & 63) == 0 && g_stamp_buf)                                                 \
      for (int i_ = 0; i_ < kStampSlots; ++i_) g_stamp_buf[(size_t)(widx) * kStampSlots + i_] = stamps_[i_]; \
  } while (0)
#else
#define BR_STAMP_DECL
#define BR_STAMP(i)
#define BR_STAMP_RT(i)
#define BR_STAMP_FLUSH(widx)
#endif

}  // namespace br
