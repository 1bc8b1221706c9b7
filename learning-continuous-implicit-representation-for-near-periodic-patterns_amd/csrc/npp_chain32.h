// npp_chain32.h -- the building blocks of the fused exact-fp32 chains (v_mfma_f32_32x32x2_f32): weight fragments streamed from a
// packed fp32 buffer (16 bytes = four k-steps of one lane), activations as the B operand from an LDS region [feature][64 rows].
// Shared by npp_mlp_fwd32.hip (the coordinate MLP's full-image render, BASELINE c4) and npp_light.hip (NPP_Net_light's training
// chains).  Layout of a packed layer: unit u (16 bytes) = [k-step group g][neuron tile nt][lane]; a group is four
// k-steps, k-step 4 g + e contracts input features (8 g + e, 8 g + e + 4) -- lane half h holds feature 8 g + e + 4 h.
#pragma once
#include "npp_common.h"

namespace npp {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ f32x16 mfma32(float a, float b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}


// A fragments of one k-step GROUP (4 k-steps) for this wave's NTW neuron tiles: [group][nt][lane][4 floats]
template <int NTW>
struct WG32 { f32x4_t w[NTW]; };
template <int NTW, int NT>
__device__ __forceinline__ void wg32_load(WG32<NTW>& r, const wrsrc_t& rsrc, uint32_t base16, int g, int nt0, int lane) {
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const u32x4_t raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, (int)((base16 + (uint32_t)((g * NT + nt0 + nt) * 64)) * 16u), 0);
    r.w[nt] = __builtin_bit_cast(f32x4_t, raw);
  }
}

// B operands of a k-step group from the activation region [feature][NB * 32 rows]: k-step 4g + e contracts features
// (8g + e, 8g + e + 4).  NB (batch tiles of 32 rows per workgroup) is deduced from the operand array.
struct ActSrc32 {
  const char* lane_base;     // region + ((4 h) * (NB * 32) + (lane & 31)) * 4
  template <int NB>
  __device__ __forceinline__ void frag(int g, int e, float (&b)[NB]) const {
#pragma unroll
    for (int bt = 0; bt < NB; ++bt) b[bt] = *(const float*)(lane_base + (g * 8 + e) * (NB * 128) + bt * 128);
  }
};
// acc += W[part] X over NG k-step groups (runtime loop, two groups per trip: the weight slots are named statically and
// refilled right after the MFMAs that read them have been issued -- one group = 16 MFMAs = 1024 cycles of cover)
template <int NTW, int NT, typename Src, int NB>
__device__ __forceinline__ void part32(f32x16 (&acc)[NTW][NB], const wrsrc_t& rsrc, uint32_t base16, int ngroups, int nt0, int lane,
                                       const Src& src) {
  WG32<NTW> w0, w1;
  wg32_load<NTW, NT>(w0, rsrc, base16, 0, nt0, lane);
  wg32_load<NTW, NT>(w1, rsrc, base16, 1, nt0, lane);
  auto group = [&](WG32<NTW>& w, int g) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float b[NB];
      src.frag(g, e, b);
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int bt = 0; bt < NB; ++bt) acc[nt][bt] = mfma32(w.w[nt][e], b[bt], acc[nt][bt]);
    }
  };
#pragma unroll 1
  for (int g = 0; g < ngroups; g += 2) {           // ngroups is even for every part (128 / 4, 232 / 4)
    group(w0, g);
    if (g + 2 < ngroups) wg32_load<NTW, NT>(w0, rsrc, base16, g + 2, nt0, lane);
    group(w1, g + 1);
    if (g + 3 < ngroups) wg32_load<NTW, NT>(w1, rsrc, base16, g + 3, nt0, lane);
  }
}

template <int NTW, int NB>
__device__ __forceinline__ void bias32(f32x16 (&acc)[NTW][NB], const float* __restrict__ bias, int nt0, int h) {
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float bv = bias[(nt0 + nt) * 32 + acc_row(r, h)];
#pragma unroll
      for (int bt = 0; bt < NB; ++bt) acc[nt][bt][r] = bv;
    }
}

// snake (or nothing) + store as the next layer's input: region[feature][row]
template <bool SNAKE, int NTW>
__device__ __forceinline__ void epi32(f32x16 (&acc)[NTW][kNB], char* region, int nt0, int b, int h) {
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int bt = 0; bt < kNB; ++bt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float z = acc[nt][bt][r];
        if (SNAKE) {
          const float s = __builtin_amdgcn_sinf(z * kInv2Pi);
          z = fmaf(s, s, z);
        }
        acc[nt][bt][r] = z;
        if (region) *(float*)(region + (((nt0 + nt) * 32 + acc_row(r, h)) * kRowTile + bt * 32 + b) * 4) = z;
      }
}

}  // namespace npp
