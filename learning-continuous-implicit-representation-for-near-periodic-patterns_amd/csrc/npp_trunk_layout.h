// npp_trunk_layout.h -- the "flat padded" activation layout of the trunk kernels (npp_conv.hip has the description), shared with
// the kernels that write a trunk tensor directly (npp_cx.hip: the contextual core's gradient lands in it without an fp32 detour).
#pragma once
#include "npp_layout.h"

namespace npp {

constexpr int kConvGuard = 1024;      // zero units before / after the position axis (>= W + 3): images up to 1021 wide
                                      // (the loop's patches are <= 160; the proposal ranking scores crops of whole images)
constexpr int kPosRound = 512;        // position count is rounded up to a multiple of this

NPP_HD int64_t conv_npos_round(int N, int H, int W) {
  const int64_t s = (int64_t)N * (H + 2) * (W + 2);
  return (s + kPosRound - 1) / kPosRound * kPosRound;
}
NPP_HD int64_t conv_nposp(int N, int H, int W) { return conv_npos_round(N, H, W) + 2 * kConvGuard; }

// The trunk input x * scale + shift as fp16: the fma rounded to fp32 FIRST, then to fp16 (what `(x * scale + shift).half()` does).  The
// register barrier keeps the compiler from fusing the two into v_fma_mixlo_f16 -- one rounding instead of two -- which it did for
// SOME channels of SOME kernels: the same input then differed in the last fp16 bit between two launches that compose it (round 5).
#if defined(__HIPCC__)
__device__ __forceinline__ _Float16 trunk_in_f16(float x, float s, float b) {
  float t = fmaf(x, s, b);
  asm volatile("" : "+v"(t));
  return (_Float16)t;
}
#endif

// true channel of element j of chunk c8 in the stored (accumulator) order
NPP_HD int conv_chan(int c8, int j) { return 32 * (c8 >> 2) + 16 * ((c8 >> 1) & 1) + perm16(c8 & 1, j); }

}  // namespace npp
