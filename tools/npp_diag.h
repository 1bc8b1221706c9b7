// tools/npp_diag.h -- in-kernel time stamps for DIAGNOSTIC builds only (tools/build_variant.sh <name> "-DNPP_DIAG -I tools").
// The product build never defines NPP_DIAG: every macro below expands to nothing and no kernel carries a stamp.
// A stamp build exports npp_diag_set_stamps(buf, n_words): launches made afterwards write, per wave, 8 words at
//   ((blockIdx.y * gridDim.x + blockIdx.x) * waves + wave) * 8 :  [0] s_memrealtime at entry (100 MHz, chip-wide), [1..6] s_memtime at
//   the kernel's phase marks (shader clock), [7] (XCC_ID << 32) | HW_ID.
// Stamp values go only to that buffer; no output is computed from them (MI355X_MICROARCH.md, DVFS give-back (6)).
#pragma once
#ifdef NPP_DIAG
namespace npp {
extern unsigned long long* g_diag_stamps;
extern long long g_diag_n;
}
#define NPP_DIAG_FIELD unsigned long long* stamps; long long stamps_n;
#define NPP_DIAG_FILL(a) do { (a).stamps = npp::g_diag_stamps; (a).stamps_n = npp::g_diag_n; } while (0)
#define NPP_STAMP(a, k)                                                                                                         \
  do {                                                                                                                          \
    if ((a).stamps && (threadIdx.x & 63) == 0) {                                                                                \
      const long long i_ = ((long long)(blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8;      \
      if (i_ + 8 <= (a).stamps_n) {                                                                                             \
        if ((k) == 0) {                                                                                                         \
          (a).stamps[i_] = __builtin_amdgcn_s_memrealtime();                                                                    \
          (a).stamps[i_ + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) |        \
                               (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));                  \
        } else {                                                                                                                \
          (a).stamps[i_ + (k)] = __builtin_readcyclecounter();                                                                  \
        }                                                                                                                       \
      }                                                                                                                         \
    }                                                                                                                           \
  } while (0)
#define NPP_STAMP_DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#else
#define NPP_DIAG_FIELD
#define NPP_DIAG_FILL(a) do { } while (0)
#define NPP_STAMP(a, k) do { } while (0)
#define NPP_STAMP_DRAIN() do { } while (0)
#endif
