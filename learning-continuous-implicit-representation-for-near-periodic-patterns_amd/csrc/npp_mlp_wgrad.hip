// npp_mlp_wgrad.hip -- K3b: weight / bias gradients of every layer in ONE grouped,
// split-K bf16 MFMA GEMM launch.
//
// dW_l[n][k] = sum_b dz_l[n][b] * a_{l-1}[k][b]   (autograd of F.linear, networks.py:56-95)
// db_l[n]    = sum_b dz_l[n][b]
// The contraction runs over the batch.  Both operands arrive as the 16-bit fragments the
// fused forward / backward kernels hold in registers (one 16-byte unit = 8 features of one
// row; "W-format" arrays, npp_layout.h): the 32 batch rows x 32 features of one k-step pair are
// a contiguous 2-KiB run that LDS-DMA (buffer_load_dwordx4 ... lds) copies straight into LDS, and
// the MFMA operand fragments (one feature, 8 consecutive rows per lane) are produced by the hardware
// transposing read ds_read_b64_tr_b16 -- conflict-free by construction of the line layout.
// A job = one (dz array, input array) pair; jobs are cut into 256x256 output tiles (8 waves of
// 64x128; with 128x128 tiles every array was read twice, 1.0 GB per pass at c2) and the
// batch is split over the grid; every (tile, split) writes its partial sums with plain
// stores into slab `split` of the gradient buffer, in the reference's parameter layout
// ([out][in] row-major).  npp_adam_step adds the slabs: no atomics, bit-reproducible.
// Main loop (round 4, wgrad_loop_hybrid): dz by LDS-DMA into an 8-slot ring, the layer-input operand through a 4-deep register ring
// with snake(z) formed in the registers; bit-identical to the round-3 loop, 107.9 -> 104.0 us in sequence.  Measured with every
// load hitting L2 (NPP_DIAG_WGRAD_SAMETILE) and the phases switched off one at a time (profiles/r04_wgrad_ablation.txt): the loop
// is a chain of phases that ADD per step -- loads + LDS stores + barrier 0.31 us, fragment reads + conversion 0.31, MFMAs 0.36
// (0.51 alone) per 32-row half -- and HBM costs 14 us on top: the launch is bound by its per-step dependent chain inside the CU,
// not by the memory side.
// Round 3: a five-slot LDS ring of half tiles filled by LDS-DMA with three halves in flight, counted vmcnt + raw
// barrier per half (wgrad_loop below); snake(z) of the inputs that are stored as fp16 pre-activations is formed in place in
// LDS under the MFMAs.  Measured at c2 (26 624 rows, in sequence): 115 -> 106-108 us; the loop now runs at the LDS-DMA
// streaming rate of a CU (~30 GB/s, the guide's "ldsdma-fill"), i.e. the launch is memory-side-bound.
//
// Algorithmic work: 2 * sum_l n_out*n_in FLOP per batch row (embedding pad slots and the
// padding of the 3-row rgb job are not counted).
#include "npp_common.h"
#include "npp_light_layout.h"
#include <stdlib.h>
#include <type_traits>

namespace npp {

constexpr int kWT = 256;            // output tile (both dims)
constexpr int kWBK = 64;            // batch rows per main-loop step
constexpr int kWThreads = 512;      // 8 waves: 4 (m) x 2 (n), 64 x 128 outputs each
constexpr int kWPairs = kWT / 32;   // k-step pairs (32 features) per operand tile
constexpr int kWTileBytes = kWT * kWBK * 2;          // 32 KiB per operand tile
#ifndef NPP_WGRAD_VALU_PER_MFMA
#define NPP_WGRAD_VALU_PER_MFMA 5
#endif
#ifndef NPP_WGRAD_SLOTS
#define NPP_WGRAD_SLOTS 5
#endif
constexpr int kSmemW = NPP_WGRAD_SLOTS * kWTileBytes;   // ring of half tiles (A | B of 32 batch rows each): 5 x 32 KiB = the whole LDS of a CU
constexpr int kMaxJobs = 24;
#ifndef NPP_WGRAD_NT_SLABS
#define NPP_WGRAD_NT_SLABS 0
#endif
#ifndef NPP_WGRAD_XCD
#define NPP_WGRAD_XCD 1
#endif
// (The timing-only diagnostic builds behind profiles/r04_wgrad_ablation.txt -- every load hitting L2, prologue + n tiles + epilogue,
// no slab stores, phases switched off one at a time -- left the source in round 5; they are in the git history of round 4.)

constexpr bool kSameTile = false;

struct WJob {
  int32_t a_ks0, a_nks, m;     // dz array: k-step offset inside dzF, k-steps, valid outputs
  int32_t b_ks0, b_nks, n;     // input array inside actF, k-steps, valid inputs / emb slots
  int32_t colmode, col0;       // 0: col = col0 + idx ; 1: col = col0 + emb_col(slot idx), pad slots skipped
  int32_t ld, bias_on;         // reference row stride (n_in) ; 1 = this job also produces db
  int32_t b_is_z, pad_;        // input array holds fp16 pre-activations: layer input = snake(z)
  int64_t w_off, b_off;        // float offsets in the parameter blob
  int32_t tile0, tiles_n;      // first global tile index, tiles along n
};

struct WArgs {
  const char* dzF;
  const char* actF;
  int64_t dz_bytes, act_bytes;   // sizes of the two stash buffers (range of the buffer descriptors)
  int64_t n_wg;                // 64-row workgroup tiles in the batch (Bp / 64)
  float* gslabs;
  int64_t slab_stride;
  int32_t njobs, wg_chunk;     // workgroup tiles per split
  int32_t ntiles, ksplit;
  // stacked launch (npp_mlp_wgrad_stack): image m reads its stash at + m * stride and writes its own ksplit slabs
  Stack S;
  int64_t dz_img_stride, act_img_stride, slab_img_stride;      // bytes, bytes, floats
  WJob jobs[kMaxJobs];
};

using rsrc_t = __amdgpu_buffer_rsrc_t;
typedef __attribute__((address_space(3))) void lds_void;

// ---- main loop: an LDS ring of HALF tiles filled by LDS-DMA --------------------------------------------------------------
// Unit of the ring = the 32 batch rows (one batch tile bt) of a 64-row workgroup tile, both operands: for each of the 8
// k-step pairs tt of an operand the W-format holds those rows as ONE contiguous 2-KiB run (two 16-row k-steps of 1 KiB),
// so wave w copies pair w of the dz tile and pair w of the input tile with two `buffer_load_dwordx4 ... lds` each
// (1 KiB per instruction, lane-linear on both sides: the LDS image is [operand][pair][2 KiB]).  No operand byte passes
// through a register and nothing is converted (the forward kernel stores the bf16 layer inputs a = snake(z) itself,
// npp_layout.h kActKsA0).  4 slots x 32 KiB: while half s is multiplied (16 MFMAs per wave), halves s + 1 and s + 2 are in
// flight and the DMA of s + 3 is issued into the slot that held s - 1, right behind the barrier that ends its reads.
// Round 2's loop moved every tile through registers (32 VGPRs, ds_write_b128 at 79 B/clk, fp16 -> snake -> bf16 conversion of
// the input tile) with ONE tile in flight per CU: 2.77 us per 64-row tile against 0.85-1.0 us of MFMA time, the same with the
// stash resident in the Infinity Cache (profiles/r03_rejected_experiments.txt) -- bound by its own LDS stores and lockstep.
// Ordering (cdna_hip_programming.md "Pipelining across barriers"): a wave waits for ITS OWN DMAs of half s with a counted
// vmcnt (the 8 younger ones = halves s + 1, s + 2 stay in flight), then the raw barrier publishes everybody's; reads of half
// s happen after that barrier, the refill of a slot after the barrier that follows its last read (lgkmcnt(0) before it).
// Halves past the split's end are requested through a ZERO-length descriptor: in-order dummy loads that touch no memory and
// keep the vmcnt arithmetic uniform (no peeled tail).
constexpr int kHalfOp = 16 * 1024;                  // one operand's half tile: 8 pairs x 2 KiB
constexpr int kSlot = 2 * kHalfOp;                  // A | B
constexpr int kSlots = NPP_WGRAD_SLOTS;
static_assert(kSlots == 5 && kSlots * kSlot == kSmemW && kSlots >= 3 && kSmemW >= 128 * 1024 && kSmemW <= 160 * 1024, "ring (the epilogue stages 8 x 16 KiB in it)");
// offset of the first ds_read_b64_tr_b16 of fragment (32-feature tile tt, 16-row k-step q of the half) inside an operand's
// half image; the second read is 256 B on (wfrag_offset with the pair stride 2 KiB instead of 4)
__device__ __forceinline__ int hfrag_offset(int tt, int lane) {
  const int g = lane >> 4, i = lane & 15, q4 = i >> 2, p = i & 3, h = g >> 1;
  return tt * 2048 + 2 * h * 256 + (g & 1) * 128 + (p & 1) * 64 + q4 * 16 + (p >> 1) * 8;
}

template <bool ZB, bool BIAS>
__device__ __forceinline__ void wgrad_loop(f32x16 (&acc)[2][4], float (&bsum)[2], char* smem, const rsrc_t ra, const rsrc_t rb,
                                           const rsrc_t rzero, uint32_t a_stride, uint32_t b_stride, int g0, int g1, int wave,
                                           int lane, const int (&offA)[2], const int (&offB)[4]) {
  const int nh = 2 * (g1 - g0);
  if (nh <= 0) return;
  const int voff = wave * 4096 + lane * 16;          // pair `wave` of the tile, this lane's 16 bytes
  // LDS-DMA through inline asm: hipcc must not count these loads -- it would wait vmcnt(0) before every ds_read that follows a
  // pending LDS-DMA (it cannot tell the slots apart), draining the ring at each step; completion is counted by hand below
  // (cdna_hip_programming.md 5.7: no VGPR destination = register-safe; M0 written in the statement that uses it)
  auto dma_half = [&](int hidx) {
    const bool ok = hidx < nh;                       // wave-uniform
    const int g = g0 + (kSameTile ? 0 : (hidx >> 1)), bt = hidx & 1;
    const rsrc_t xa = ok ? ra : rzero, xb = ok ? rb : rzero;
    const int soa = ok ? (int)((uint32_t)g * a_stride) + bt * 2048 : 0;
    const int sob = ok ? (int)((uint32_t)g * b_stride) + bt * 2048 : 0;
    const uint32_t d0 = (uint32_t)(uintptr_t)(lds_void*)(smem + (hidx % kSlots) * kSlot + wave * 2048);
    unsigned keep;
    asm volatile(
        "s_mov_b32 %[keep], m0\n\t"
        "s_mov_b32 m0, %[d0]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v], %[xa], %[sa0] offen lds\n\t"
        "s_mov_b32 m0, %[d1]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v], %[xa], %[sa1] offen lds\n\t"
        "s_mov_b32 m0, %[d2]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v], %[xb], %[sb0] offen lds\n\t"
        "s_mov_b32 m0, %[d3]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v], %[xb], %[sb1] offen lds\n\t"
        "s_mov_b32 m0, %[keep]"
        : [keep] "=&s"(keep)
        : [v] "v"(voff), [xa] "s"(xa), [xb] "s"(xb), [sa0] "s"(soa), [sa1] "s"(soa + 1024), [sb0] "s"(sob), [sb1] "s"(sob + 1024),
          [d0] "s"(d0), [d1] "s"(d0 + 1024u), [d2] "s"(d0 + (uint32_t)kHalfOp), [d3] "s"(d0 + (uint32_t)kHalfOp + 1024u)
        : "memory");
  };
  // ZB (the input array holds the fp16 pre-activations z of a snake layer): the layer input snake(z) is formed IN PLACE in the
  // LDS image, bf16 over fp16, one half ahead of its use -- every lane converts exactly the 2 x 16 bytes its own two DMA
  // instructions delivered, so its wave's counted vmcnt is all the ordering the conversion needs; the barrier of the next step
  // publishes it.  The loop is bound by the memory side (~30 GB/s of LDS-DMA per CU, MI355X_MICROARCH.md "ldsdma-fill"): the
  // conversion's vector work and its 2 + 2 LDS accesses per lane hide under the wait for the next half.
  auto convert_half = [&](int hidx) {
    char* base = smem + (hidx % kSlots) * kSlot + kHalfOp + wave * 2048 + lane * 16;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const f16x8 z = *(const f16x8*)(base + q * 1024);
      bf16x8 a;
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = (__bf16)snake_fast((float)z[j]);
      *(bf16x8*)(base + q * 1024) = a;
    }
  };
#pragma unroll
  for (int i = 0; i < kSlots - 1; ++i) dma_half(i);
  if (ZB) {
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");          // (kSlots - 2) x 4 younger DMAs: half 0 has landed
    convert_half(0);
  }
  int slot_off = 0;                                           // (s % kSlots) * kSlot, kept as a rotating scalar
  for (int s = 0; s < nh; ++s) {
    if (ZB) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // this wave's DMAs of halves s and s + 1 have landed
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");    // ... of half s (halves s + 1 .. s + 3 stay in flight)
    wg_barrier();                                             // everybody's half s is complete; every read of half s - 1 is over
    dma_half(s + kSlots - 1);                                 // -> the slot half s - 1 occupied
    const char* sA = smem + slot_off;
    const char* sB = sA + kHalfOp;
    // Source order = dependence order the compiler must keep between LDS accesses it cannot tell apart: all fragment reads of
    // half s FIRST, then (ZB) the in-place conversion of half s + 1 -- its stores may then sink below the MFMAs and its vector
    // work interleave with them (sched_group_barrier below); the barrier of step s + 1 publishes it.  Unconditional (no branch =
    // one scheduling region): past the split's end it rewrites a slot nobody reads.
    bf16x8 a[2][2], b[2][4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int i = 0; i < 2; ++i) a[q][i] = wfrag_read(sA, offA[0] + i * 2048 + q * 1024);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[q][j] = wfrag_read(sB, offB[0] + j * 2048 + q * 1024);
    }
    if (ZB) convert_half(s + 1);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (BIAS) {
        const bf16x2 ones = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 8; j += 2) {
            const bf16x2 pr = {a[q][i][j], a[q][i][j + 1]};
            bsum[i] = __builtin_amdgcn_fdot2_f32_bf16(pr, ones, bsum[i], false);
          }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma_bf16(a[q][i], b[q][j], acc[i][j]);
    }
    if (ZB) {
      // 16 MFMAs, ~70 vector instructions of the conversion: one MFMA, then its share of the vector work
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NPP_WGRAD_VALU_PER_MFMA, 0);
      }
    }
    slot_off = slot_off + kSlot == kSlots * kSlot ? 0 : slot_off + kSlot;
  }
  // the dummy / tail DMAs issued by the last steps must not land in LDS after the epilogue starts staging there
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wg_barrier();
}


// ---- hybrid ingest (round 4, NPP_WGRAD_HYBRID): dz by LDS-DMA, the layer-input operand through registers ---------------------
// The five-slot ring above moves BOTH operands of a half by LDS-DMA: 32 KiB per half step land at the ~29 GB/s a CU's DMA path
// delivers, 1.1 us against 0.45 us of MFMA time.  Here the dz operand keeps that path (ring of kNA 16-KiB slots, DA halves in
// flight) and the input operand goes memory -> VGPR (2 x buffer_load_dwordx4 per lane and half, RB halves in flight in RB register
// sets) -> [snake(z) formed IN the registers for the fp16 z arrays: no LDS round trip] -> ds_write_b128 into one of two 16-KiB B
// slots, one half ahead of its use.  Every vector-memory instruction of the loop is inline asm and counted by hand (an LDS-DMA
// hipcc knows about makes it wait vmcnt(0) before the next ds_read; a register load it knows about is waited for with a count
// that ignores the DMAs in between and drains the ring): per step the issue order is [DMA dz(s + DA) x 2, load in(s + RB) x 2],
// so `s_waitcnt vmcnt(4 (RB - 2))` at the top of step s covers this wave's in(s + 1) registers and, with DA = RB - 1, its dz(s)
// DMAs; the raw barrier publishes them.  The loop body is unrolled RB times so that the register set of a half is a compile-time
// name (a rotating copy of an in-flight register would read stale data: nothing interlocks a VGPR with a pending load the
// compiler does not know about).
#ifndef NPP_WGRAD_HYBRID
#define NPP_WGRAD_HYBRID 1     // 0: the round-3 all-LDS-DMA ring (bit-identical results; 107.9 vs 104.0 us in sequence at c2)
#endif
#ifndef NPP_WGRAD_LATE_ISSUE
#define NPP_WGRAD_LATE_ISSUE 0
#endif
#ifndef NPP_WGRAD_RB
#define NPP_WGRAD_RB 4
#endif
constexpr int kRB = NPP_WGRAD_RB, kDA = kRB - 1;
constexpr int kNA = 8;                                  // dz ring: 8 x 16 KiB, then the two input slots: 160 KiB in all
constexpr int kBBase = kNA * kHalfOp;
static_assert(kBBase + 2 * kHalfOp <= kSmemW && kNA >= kDA + 1 && kRB >= 3 && kRB <= 6, "hybrid ring");

template <bool ZB, bool BIAS>
__device__ __forceinline__ void wgrad_loop_hybrid(f32x16 (&acc)[2][4], float (&bsum)[2], char* smem, const rsrc_t ra, const rsrc_t rb,
                                                  const rsrc_t rzero, uint32_t a_stride, uint32_t b_stride, int g0, int g1, int wave,
                                                  int lane, const int (&offA)[2], const int (&offB)[4]) {
  const int nh = 2 * (g1 - g0);
  if (nh <= 0) return;
  const int voff = wave * 4096 + lane * 16;          // pair `wave` of the tile, this lane's 16 bytes
  u32x4 R[kRB][2];                                   // register sets of the input operand: half h lives in R[h % kRB]
  auto dma_a = [&](int hidx) {                       // dz half hidx -> A slot hidx % kNA (zero-length descriptor past the end)
    const bool ok = hidx >= 0 && hidx < nh;          // wave-uniform
    const int g = g0 + (kSameTile ? 0 : (hidx >> 1)), bt = hidx & 1;
    const rsrc_t xa = ok ? ra : rzero;
    const int soa = ok ? (int)((uint32_t)g * a_stride) + bt * 2048 : 0;
    const uint32_t d0 = (uint32_t)(uintptr_t)(lds_void*)(smem + ((hidx + kNA) % kNA) * kHalfOp + wave * 2048);
    unsigned keep;
    asm volatile(
        "s_mov_b32 %[keep], m0\n\t"
        "s_mov_b32 m0, %[d0]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v], %[xa], %[sa0] offen lds\n\t"
        "s_mov_b32 m0, %[d1]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v], %[xa], %[sa1] offen lds\n\t"
        "s_mov_b32 m0, %[keep]"
        : [keep] "=&s"(keep)
        : [v] "v"(voff), [xa] "s"(xa), [sa0] "s"(soa), [sa1] "s"(soa + 1024), [d0] "s"(d0), [d1] "s"(d0 + 1024u)
        : "memory");
  };
  auto load_b = [&](int hidx, u32x4 (&r)[2]) {       // input half hidx -> registers (zeros past the end)
    const bool ok = hidx < nh;
    const int g = g0 + (kSameTile ? 0 : (hidx >> 1)), bt = hidx & 1;
    const rsrc_t xb = ok ? rb : rzero;
    const int sob = ok ? (int)((uint32_t)g * b_stride) + bt * 2048 : 0;
    asm volatile(
        "buffer_load_dwordx4 %[r0], %[v], %[xb], %[sb] offen\n\t"
        "buffer_load_dwordx4 %[r1], %[v], %[xb], %[sb] offen offset:1024"
        : [r0] "=&v"(r[0]), [r1] "=&v"(r[1])
        : [v] "v"(voff), [xb] "s"(xb), [sb] "s"(sob)
        : "memory");
  };
  // registers of a landed half -> (snake) -> the B slot of that half
  auto store_b = [&](int hidx, u32x4 (&r)[2]) {
    char* base = smem + kBBase + (hidx & 1) * kHalfOp + wave * 2048 + lane * 16;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (ZB) {
        const f16x8 z = __builtin_bit_cast(f16x8, r[q]);
        bf16x8 a;
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = (__bf16)snake_fast((float)z[j]);
        *(bf16x8*)(base + q * 1024) = a;
      } else {
        *(u32x4*)(base + q * 1024) = r[q];
      }
    }
  };
  // prologue = steps -kRB .. -1 of the steady-state issue pattern (dummy DMAs for the dz halves < 0 keep the count uniform)
#pragma unroll
  for (int t = -kRB; t < 0; ++t) {
    dma_a(t + kDA);
    load_b(t + kRB, R[(t + kRB) % kRB]);
  }
  asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(R[0][0]), "+v"(R[0][1]) : [n] "n"(4 * (kRB - 1)) : "memory");     // in(0) has landed
  store_b(0, R[0]);
  int a_off = 0;                                              // (s % kNA) * kHalfOp
  // one step; I = s % kRB at compile time
  auto step = [&](int s, auto I_) {
    constexpr int I = decltype(I_)::value, I1 = (I + 1) % kRB;
    asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(R[I1][0]), "+v"(R[I1][1]) : [n] "n"(4 * (kRB - 2)) : "memory");   // in(s + 1), own dz(s)
    wg_barrier();                                             // everybody's dz(s) and in(s); every read of step s - 1 is over
#if !NPP_WGRAD_LATE_ISSUE
    dma_a(s + kDA);
    load_b(s + kRB, R[I]);
#endif
    const char* sA = smem + a_off;
    const char* sB = smem + kBBase + (s & 1) * kHalfOp;
    bf16x8 a[2][2], b[2][4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int i = 0; i < 2; ++i) a[q][i] = wfrag_read(sA, offA[0] + i * 2048 + q * 1024);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[q][j] = wfrag_read(sB, offB[0] + j * 2048 + q * 1024);
    }
    store_b(s + 1, R[I1]);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (BIAS) {
        const bf16x2 ones = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 8; j += 2) {
            const bf16x2 pr = {a[q][i][j], a[q][i][j + 1]};
            bsum[i] = __builtin_amdgcn_fdot2_f32_bf16(pr, ones, bsum[i], false);
          }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j] = mfma_bf16(a[q][i], b[q][j], acc[i][j]);
        }
#if NPP_WGRAD_LATE_ISSUE
      // the step's memory requests go out in the shadow of the first k-step's MFMAs instead of in front of the fragment reads:
      // an LDS-DMA piece costs its wave ~60-185 cycles of issue (MI355X_MICROARCH.md), which was exposed before every step
      if (ZB) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, NPP_WGRAD_VALU_PER_MFMA, 0);
        }
      }
      if (q == 0) {
        __builtin_amdgcn_sched_barrier(0);
        dma_a(s + kDA);
        load_b(s + kRB, R[I]);
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
    }
#if !NPP_WGRAD_LATE_ISSUE
    if (ZB) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NPP_WGRAD_VALU_PER_MFMA, 0);
      }
    }
#endif
    a_off = a_off + kHalfOp == kNA * kHalfOp ? 0 : a_off + kHalfOp;
  };
  int s = 0;
  for (; s + kRB <= nh; s += kRB) {
    step(s, std::integral_constant<int, 0>{});
    step(s + 1, std::integral_constant<int, 1>{});
    step(s + 2, std::integral_constant<int, 2>{});
    if constexpr (kRB > 3) step(s + 3, std::integral_constant<int, 3>{});
    if constexpr (kRB > 4) step(s + 4, std::integral_constant<int, 4>{});
    if constexpr (kRB > 5) step(s + 5, std::integral_constant<int, 5>{});
  }
  const int rem = nh - s;                                     // < kRB, wave-uniform
  if (rem > 0) step(s, std::integral_constant<int, 0>{});
  if (rem > 1) step(s + 1, std::integral_constant<int, 1>{});
  if (rem > 2) step(s + 2, std::integral_constant<int, 2>{});
  if constexpr (kRB > 4) if (rem > 3) step(s + 3, std::integral_constant<int, 3>{});
  if constexpr (kRB > 5) if (rem > 4) step(s + 4, std::integral_constant<int, 4>{});
  // dummy / tail DMAs and loads must have landed before the epilogue stages in LDS -- and before the register sets die: the
  // sets are operands of this wait, so every one of them stays allocated up to here (a set the compiler considers dead is
  // re-used at once, and the load still in flight then lands on top of the new value: seen in the tail steps, whose loads
  // nothing consumes)
#pragma unroll
  for (int i = 0; i < kRB; ++i) asm volatile("" : "+v"(R[i][0]), "+v"(R[i][1]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < kRB; ++i) asm volatile("" : "+v"(R[i][0]), "+v"(R[i][1]));
  wg_barrier();
}

// Work item (tile, split) of this workgroup.  Workgroups are dealt round-robin over the 8 XCDs (observed, speed only:
// MI355X_MICROARCH.md "Workgroup dispatch"), so linear id i runs on XCD group i % 8; the items are numbered so that
// every group owns a CONTIGUOUS range of them = all tiles of one batch split (+ part of the next): the tiles of a split
// that read the same dz / input array (3 tiles share dz_5, 5 share dz_S, emb_0 feeds L0 and L5, f1 feeds S and P) then
// stream it through ONE L2 at about the same time instead of each fetching it from HBM.  false: surplus workgroup / idle image.
__device__ __forceinline__ bool wgrad_item(const WArgs& A, int& item, const char*& dzF_, const char*& actF_, float*& gslabs_) {
  if (A.S.M) {
    // stacked: image m owns S.g XCDs (npp_common.h); inside the image the same rule -- each of its XCDs a contiguous item range
    const int n_items = A.S.n_items, xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    int img, xl, gx;
    if (A.S.g) { img = xcd / A.S.g; xl = xcd % A.S.g; gx = A.S.g; }
    else { img = (int)blockIdx.x / n_items; xl = 0; gx = 1; }
    const int q_ = n_items / gx, r_ = n_items % gx;
    const int lslot = A.S.g ? slot : (int)blockIdx.x - img * n_items;
    if (lslot >= (xl < r_ ? q_ + 1 : q_)) return false;
    item = (xl < r_ ? xl * (q_ + 1) : r_ * (q_ + 1) + (xl - r_) * q_) + lslot;
    if (A.S.iter && !A.S.iter[img].active) return false;
    dzF_ += (int64_t)img * A.dz_img_stride;
    actF_ += (int64_t)img * A.act_img_stride;
    gslabs_ += (int64_t)img * A.slab_img_stride;
  } else {
#if NPP_WGRAD_XCD
    const int n_items = (int)gridDim.x, xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    const int q_ = n_items >> 3, r_ = n_items & 7;
    item = (xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) + slot;
#else
    item = (int)blockIdx.x;
#endif
  }
  return true;
}

// ---- epilogue shared by the 16-bit and the 8-bit launch: stores into this split's slab, reference layout.  P8: the operands came
// through ds_read_b64_tr_b8, so accumulator row / column index i of a 32-feature tile is feature w8_feat(i) (npp_layout.h); BI:
// the accumulator row tile whose bias sum this wave carries in bsum[.] (16-bit launch: both, wave column 0; 8-bit: tile `wn`)
template <bool P8>
__device__ __forceinline__ void wgrad_epilogue(const WJob& J, f32x16 (&acc)[2][4], float (&bsum)[2], char* smem, float* gslabs_,
                                               int64_t slab_stride, int split_id, int tm, int tn, int wm, int wn, int wave, int lane,
                                               bool do_bias) {
  const int m_l = lane & 31, h = lane >> 5;
  const int m_f = P8 ? w8_feat(m_l) : m_l;                   // feature (within its 32-tile) of this lane's accumulator column
  // ---- epilogue: stores into this split's slab, reference layout
  float* slab = gslabs_ + (int64_t)split_id * slab_stride;
  if (J.colmode == 0 && ((J.ld | J.col0) & 1) == 0) {        // (every layer of this network has an even input width)
    // Plain-column jobs (17 of 21 tiles at K = 3): the accumulator holds one column per lane, i.e. 4-byte stores, 128 per wave
    // -- the tail was bound by store INSTRUCTIONS, not bytes (cdna_hip_programming.md T21).  Each wave transposes its 32 x 128
    // strips through its 16 KiB of the (now idle) operand buffers and stores float4s (two float2s where the reference rows
    // are only 8-byte aligned: ld or col0 not a multiple of 4): 16-32 store instructions per strip instead of 64.
    float* stg = (float*)(smem + wave * 16384);
    const bool a16 = ((J.ld | J.col0) & 3) == 0;            // (w_off and the slab stride are multiples of 4 floats)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) stg[(P8 ? w8_feat(acc_row(r, h)) : acc_row(r, h)) * 128 + 32 * j + m_f] = acc[i][j][r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // same wave: the writes have landed before other lanes' elements are read
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        const int row = 2 * p + h, c4 = 4 * m_l;
        const float4 v = *(const float4*)(stg + row * 128 + c4);
        const int mrow = tm * kWT + wm * 64 + i * 32 + row;
        const int n_idx = tn * kWT + wn * 128 + c4;
        if (mrow < J.m && n_idx < J.n) {                    // J.n is a multiple of 4 for these jobs
          float* dst = slab + J.w_off + (int64_t)mrow * J.ld + J.col0 + n_idx;
          typedef float f4v __attribute__((ext_vector_type(4)));
          typedef float f2v __attribute__((ext_vector_type(2)));
#if NPP_WGRAD_NT_SLABS
          if (a16) {
            __builtin_nontemporal_store((f4v){v.x, v.y, v.z, v.w}, (f4v*)dst);
          } else {
            __builtin_nontemporal_store((f2v){v.x, v.y}, (f2v*)dst);
            __builtin_nontemporal_store((f2v){v.z, v.w}, (f2v*)(dst + 2));
          }
#else
          if (a16) {
            *(f4v*)dst = (f4v){v.x, v.y, v.z, v.w};
          } else {
            *(f2v*)dst = (f2v){v.x, v.y};
            *(f2v*)(dst + 2) = (f2v){v.z, v.w};
          }
#endif
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // reads done before the next strip overwrites the staging area
    }
  } else
  {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n_idx = tn * kWT + wn * 128 + j * 32 + m_f;     // accumulator column = lane & 31 (P8: its feature)
    int col = -1;
    if (n_idx < J.n) {
      if (J.colmode == 0) col = J.col0 + n_idx;
      else {
        // column c of k-step ks is element unperm_j(c) of lane-half unperm_hh(c) (perm16 order)
        const int c16 = n_idx & 15;
        const int c = emb_col(n_idx >> 4, unperm_hh(c16), unperm_j(c16));
        col = c < 0 ? -1 : J.col0 + c;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mrow = tm * kWT + wm * 64 + i * 32 + (P8 ? w8_feat(acc_row(r, h)) : acc_row(r, h));
        if (col >= 0 && mrow < J.m) slab[J.w_off + (int64_t)mrow * J.ld + col] = acc[i][j][r];
      }
    }
  }
  }
  if (do_bias) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (P8 && i != wn) continue;                           // 8-bit launch: wave column wn sums tile wn (in bsum[0])
      const float b = P8 ? bsum[0] : bsum[i];
      const float v = b + __shfl_xor(b, 32, 64);
      const int mrow = tm * kWT + wm * 64 + i * 32 + m_f;
      if (h == 0 && mrow < J.m) slab[J.b_off + mrow] = v;
    }
  }
}

__global__ __launch_bounds__(kWThreads, 2) void wgrad_kernel(WArgs A) {
  const char* dzF_ = A.dzF;
  const char* actF_ = A.actF;
  float* gslabs_ = A.gslabs;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int m_l = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;     // wave tile: rows [64 wm, +64), cols [128 wn, +128)

  int item;
  if (!wgrad_item(A, item, dzF_, actF_, gslabs_)) return;
  const int tile_id = item % A.ntiles, split_id = item / A.ntiles;
  // locate the job of this tile: compile-time indices into the kernel-argument table so
  // it is read with scalar loads (a run-time index would force a scratch copy of it)
  WJob J = A.jobs[0];
#pragma unroll
  for (int j = 1; j < kMaxJobs; ++j)
    if (j < A.njobs && tile_id >= A.jobs[j].tile0) J = A.jobs[j];
  const int t_local = tile_id - J.tile0;
  const int tm = t_local / J.tiles_n, tn = t_local - tm * J.tiles_n;
  // operand tiles: 8 k-step pairs (256 features); fewer are valid at the array's end
  const int64_t n_wg = A.n_wg;
  const int64_t wg_begin = (int64_t)split_id * A.wg_chunk;
  const int64_t wg_end = min(n_wg, wg_begin + (int64_t)A.wg_chunk);
  // byte address of (workgroup tile g, pair p) inside an array: ((g * nks/2 + p) * 2) * 2048
  const int64_t a_col = wfmt_array_base(J.a_ks0, n_wg) + (int64_t)tm * kWPairs * 4096;
  const int64_t b_col = wfmt_array_base(J.b_ks0, n_wg) + (int64_t)tn * kWPairs * 4096;
  const int64_t a_left = A.dz_bytes - a_col, b_left = A.act_bytes - b_col;
  const rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dzF_ + a_col), 0, (int)(a_left > 0x7fffffffLL ? 0x7fffffffLL : a_left), 0x00020000);
  const rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(actF_ + b_col), 0, (int)(b_left > 0x7fffffffLL ? 0x7fffffffLL : b_left), 0x00020000);
  const rsrc_t rzero = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dzF_), 0, 0, 0x00020000);      // every access out of range
  const uint32_t a_stride = (uint32_t)J.a_nks * 2048u, b_stride = (uint32_t)J.b_nks * 2048u;

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  float bsum[2] = {0.0f, 0.0f};
  const bool do_bias = J.bias_on && tn == 0 && wn == 0;          // wave-uniform
  // per-lane fragment offsets inside an operand's half image: feature tile (wm|wn)*2 + i
  int offA[2], offB[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) offA[i] = hfrag_offset(wm * 2 + i, lane);
#pragma unroll
  for (int j = 0; j < 4; ++j) offB[j] = hfrag_offset(wn * 4 + j, lane);

  // four straight-line forms of the loop (input array holds z or ready operands; this wave sums db or not), selected once by uniform branches
  const int g0 = (int)wg_begin, g1 = (int)wg_end;
  // (round 5: a static s_setprio 1 for waves 4-7 here -- MI355X_MICROARCH.md 'Two waves per SIMD' item 4 -- changed nothing:
  //  100.5-101.9 us without, 100.9-102.8 with, same box; profiles/r05_rejected.txt)
#if NPP_WGRAD_HYBRID
#define wgrad_loop wgrad_loop_hybrid
#endif
  if (J.b_is_z) {
    if (do_bias) wgrad_loop<true, true>(acc, bsum, smem, ra, rb, rzero, a_stride, b_stride, g0, g1, wave, lane, offA, offB);
    else wgrad_loop<true, false>(acc, bsum, smem, ra, rb, rzero, a_stride, b_stride, g0, g1, wave, lane, offA, offB);
  } else {
    if (do_bias) wgrad_loop<false, true>(acc, bsum, smem, ra, rb, rzero, a_stride, b_stride, g0, g1, wave, lane, offA, offB);
    else wgrad_loop<false, false>(acc, bsum, smem, ra, rb, rzero, a_stride, b_stride, g0, g1, wave, lane, offA, offB);
  }
#if NPP_WGRAD_HYBRID
#undef wgrad_loop
#endif

  wgrad_epilogue<false>(J, acc, bsum, smem, gslabs_, A.slab_stride, split_id, tm, tn, wm, wn, wave, lane, do_bias);
}

// ======== the 8-bit launch (round 6, npp_tune "stash8"): bf8 gradients x fp8 layer inputs on v_mfma_scale_f32_32x32x64_f8f6f4 ========
// Both operands arrive in the W8-format (npp_layout.h): the (k-step pair, 64-row workgroup tile) chunk of an array is one contiguous
// 2-KiB run, so a main-loop step = ONE workgroup tile (64 rows = the K of one matrix instruction): wave w copies pair w of the dz tile
// and pair w of the input tile with two 1-KiB LDS-DMA pieces each, nothing passes through a register, nothing is converted (the forward
// stashes snake(z) itself).  Per step and wave: 6 fragments x 4 ds_read_b64_tr_b8 + 8 MFMAs of 64 cycles -- half the LDS bytes, half
// the DMA bytes and half the matrix-pipe time of the bf16 loop for the same rows.  The power-of-two scale of the tile's gradients
// (npp_mlp_bwd: one int32 per workgroup tile behind the arrays) is the instruction's block scale for operand A.
// LDS: a ring of four 32-KiB slots (A | B: three steps in flight behind the one being multiplied; the epilogue stages in it) + the
// split's scale words.
constexpr int kOp8 = 16 * 1024;                    // one operand's tile image: 8 pairs x 2 KiB
constexpr int kSlot8 = 2 * kOp8;
constexpr int kSlots8 = 4;
constexpr int kScale8Max = 8192;                   // workgroup tiles per split whose scale words fit the 32 KiB behind the ring
constexpr int kSmemW8 = kSlots8 * kSlot8 + kScale8Max * 4;
static_assert(kSmemW8 <= 160 * 1024 && kSlots8 * kSlot8 >= 8 * 16384, "LDS (the epilogue stages 8 x 16 KiB)");
typedef int v2i_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2i_t lds_v2i_t;
// this lane's offset of the first of the four ds_read_b64_tr_b8 that build the operand fragment "feature w8_feat(lane & 31) of
// pair tt, rows 32 (lane >> 5) + 0..31" inside an operand's tile image; reads t = 1..3 are 256 B (one row group) further each
__device__ __forceinline__ int frag8_offset(int tt, int lane) {
  return tt * 2048 + (lane >> 5) * 1024 + ((lane >> 4) & 1) * 128 + (lane & 1) * 64 + ((lane & 15) >> 1) * 8;
}
__device__ __forceinline__ i32x8 frag8_read(const char* tile, int off) {
  i32x8 r;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const v2i_t v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i_t*)(tile + off + t * 256));
    r[2 * t] = v[0];
    r[2 * t + 1] = v[1];
  }
  return r;
}

template <bool BIAS>
__device__ __forceinline__ void wgrad8_loop(f32x16 (&acc)[2][4], float& bsum, char* smem, const int* sScale, const rsrc_t ra, const rsrc_t rb,
                                            const rsrc_t rzero, uint32_t a_stride, uint32_t b_stride, int g0, int g1, int wave, int lane,
                                            const int (&offA)[2], const int (&offB)[4], int bias_i) {
  const int ns = g1 - g0;
  if (ns <= 0) return;
  const int voff = wave * 2048 + lane * 16;          // pair `wave` of the tile, this lane's 16 bytes
  // (LDS-DMA through inline asm and counted by hand, as in wgrad_loop above)
  auto dma_step = [&](int sidx) {
    const bool ok = sidx < ns;                       // wave-uniform
    const rsrc_t xa = ok ? ra : rzero, xb = ok ? rb : rzero;
    const int soa = ok ? (int)((uint32_t)(g0 + sidx) * a_stride) : 0;
    const int sob = ok ? (int)((uint32_t)(g0 + sidx) * b_stride) : 0;
    const uint32_t d0 = (uint32_t)(uintptr_t)(lds_void*)(smem + (sidx % kSlots8) * kSlot8 + wave * 2048);
    unsigned keep;
    asm volatile(
        "s_mov_b32 %[keep], m0\n\t"
        "s_mov_b32 m0, %[d0]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v], %[xa], %[sa0] offen lds\n\t"
        "s_mov_b32 m0, %[d1]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v], %[xa], %[sa1] offen lds\n\t"
        "s_mov_b32 m0, %[d2]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v], %[xb], %[sb0] offen lds\n\t"
        "s_mov_b32 m0, %[d3]\n\ts_nop 0\n\tbuffer_load_dwordx4 %[v], %[xb], %[sb1] offen lds\n\t"
        "s_mov_b32 m0, %[keep]"
        : [keep] "=&s"(keep)
        : [v] "v"(voff), [xa] "s"(xa), [xb] "s"(xb), [sa0] "s"(soa), [sa1] "s"(soa + 1024), [sb0] "s"(sob), [sb1] "s"(sob + 1024),
          [d0] "s"(d0), [d1] "s"(d0 + 1024u), [d2] "s"(d0 + (uint32_t)kOp8), [d3] "s"(d0 + (uint32_t)kOp8 + 1024u)
        : "memory");
  };
#pragma unroll
  for (int i = 0; i < kSlots8 - 1; ++i) dma_step(i);
  int slot_off = 0;
  for (int s = 0; s < ns; ++s) {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * (kSlots8 - 2)) : "memory");    // this wave's pieces of step s have landed
    wg_barrier();                                     // everybody's have; every read of step s - 1 is over
    dma_step(s + kSlots8 - 1);                        // -> the slot step s - 1 occupied
    const char* sA = smem + slot_off;
    const char* sB = sA + kOp8;
    const int sc = sScale[s];                         // E8M0 byte of the tile's gradient scale (uniform)
    i32x8 a[2], b[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = frag8_read(sA, offA[i]);
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = frag8_read(sB, offB[j]);
    if (BIAS) {
      // db: this wave's share (tile bias_i of its two) -- the 32 rows this lane holds of its feature, decoded and summed
      const i32x8 av = bias_i ? a[1] : a[0];
      float t0 = 0.0f, t1 = 0.0f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_bf8(av[q], false), hi = __builtin_amdgcn_cvt_pk_f32_bf8(av[q], true);
        t0 += lo[0] + hi[0];
        t1 += lo[1] + hi[1];
      }
      bsum = fmaf(t0 + t1, __uint_as_float((uint32_t)sc << 23), bsum);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i], b[j], acc[i][j], 1 /* A: bf8 */, 0 /* B: fp8 */, 0, sc, 0, 0x7f7f7f7f);
    slot_off = slot_off + kSlot8 == kSlots8 * kSlot8 ? 0 : slot_off + kSlot8;
  }
  // the dummy / tail DMAs issued by the last steps must not land in LDS after the epilogue starts staging there
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wg_barrier();
}

__global__ __launch_bounds__(kWThreads, 2) void wgrad8_kernel(WArgs A) {
  const char* dzF_ = A.dzF;
  const char* actF_ = A.actF;
  float* gslabs_ = A.gslabs;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;     // wave tile: rows [64 wm, +64), cols [128 wn, +128)
  int item;
  if (!wgrad_item(A, item, dzF_, actF_, gslabs_)) return;
  const int tile_id = item % A.ntiles, split_id = item / A.ntiles;
  WJob J = A.jobs[0];
#pragma unroll
  for (int j = 1; j < kMaxJobs; ++j)
    if (j < A.njobs && tile_id >= A.jobs[j].tile0) J = A.jobs[j];
  const int t_local = tile_id - J.tile0;
  const int tm = t_local / J.tiles_n, tn = t_local - tm * J.tiles_n;
  const int64_t n_wg = A.n_wg;
  const int64_t wg_begin = (int64_t)split_id * A.wg_chunk;
  const int64_t wg_end = min(n_wg, wg_begin + (int64_t)A.wg_chunk);
  const int g0 = (int)wg_begin, g1 = (int)wg_end;
  // the split's scale words -> LDS behind the ring, before any LDS-DMA is in flight (plain loads, waited for here)
  int* sScale = (int*)(smem + kSlots8 * kSlot8);
  {
    const int32_t* sc = (const int32_t*)(dzF_ + dz8_scale_base(n_wg));
    for (int i = tid; i < g1 - g0; i += kWThreads) sScale[i] = sc[g0 + i];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();
  }
  // byte address of (workgroup tile g, pair p) inside an array: (g * nks/2 + p) * 2048; actF_ = the base of the 8-bit region
  const int64_t a_col = wfmt8_array_base(J.a_ks0, n_wg) + (int64_t)tm * kWPairs * 2048;
  const int64_t b_col = wfmt8_array_base(J.b_ks0, n_wg) + (int64_t)tn * kWPairs * 2048;
  const int64_t a_left = A.dz_bytes - a_col, b_left = A.act_bytes - b_col;
  const rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dzF_ + a_col), 0, (int)(a_left > 0x7fffffffLL ? 0x7fffffffLL : a_left), 0x00020000);
  const rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(actF_ + b_col), 0, (int)(b_left > 0x7fffffffLL ? 0x7fffffffLL : b_left), 0x00020000);
  const rsrc_t rzero = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dzF_), 0, 0, 0x00020000);      // every access out of range
  const uint32_t a_stride = (uint32_t)J.a_nks * 1024u, b_stride = (uint32_t)J.b_nks * 1024u;

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  float bsum[2] = {0.0f, 0.0f};
  const bool do_bias = J.bias_on && tn == 0;                     // wave column wn sums accumulator row tile wn
  int offA[2], offB[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) offA[i] = frag8_offset(wm * 2 + i, lane);
#pragma unroll
  for (int j = 0; j < 4; ++j) offB[j] = frag8_offset(wn * 4 + j, lane);
  if (do_bias) wgrad8_loop<true>(acc, bsum[0], smem, sScale, ra, rb, rzero, a_stride, b_stride, g0, g1, wave, lane, offA, offB, wn);
  else wgrad8_loop<false>(acc, bsum[0], smem, sScale, ra, rb, rzero, a_stride, b_stride, g0, g1, wave, lane, offA, offB, wn);
  wgrad_epilogue<true>(J, acc, bsum, smem, gslabs_, A.slab_stride, split_id, tm, tn, wm, wn, wave, lane, do_bias);
}

// Build the job table for NPP_Net (K>1) / NPP_Net_top1 (K==1).
static int build_jobs(int K, WArgs& A, int tile_n = kWT) {
  const NetDesc d = make_desc(K);
  int nj = 0, tile = 0;
  auto add = [&](int layer, int a_ks0, int a_nks, int b_ks0, int b_nks, int n, int colmode, int col0, int bias_on,
                 int b_is_z) {
    WJob& j = A.jobs[nj++];
    j.b_is_z = b_is_z;
    j.a_ks0 = a_ks0; j.a_nks = a_nks; j.m = d.n_out[layer];
    j.b_ks0 = b_ks0; j.b_nks = b_nks; j.n = n;
    j.colmode = colmode; j.col0 = col0;
    j.ld = d.n_in[layer]; j.bias_on = bias_on;
    j.w_off = d.w_off[layer]; j.b_off = d.b_off[layer];
    j.tile0 = tile;
    j.tiles_n = (n + tile_n - 1) / tile_n;
    tile += ((j.m + kWT - 1) / kWT) * j.tiles_n;
  };
  const int A16 = kKSAct;
  // source arrays 0..7 (L0..L7 outputs) and a_s, a_p hold z (fp16); f1 / f2 are linear (bf16)
  auto act = [&](int layer, int dz_idx, int src_idx, int col0, int bias_on) {
    add(layer, dz_idx * A16, A16, src_idx * A16, A16, kW, 0, col0, bias_on, src_idx != kActF1 && src_idx != kActF2);
  };
  auto emb = [&](int layer, int dz_idx, int p, int col0, int bias_on) {
    add(layer, dz_idx * A16, A16, kActKsEmb0 + p * kKSEmb, kKSEmb, kEmbSlots, 1, col0, bias_on, 0);
  };
  emb(L0, 0, 0, 0, 1);
  for (int l = L1; l <= L4; ++l) act(l, l, l - 1, 0, 1);
  emb(L5, 5, 0, 0, 1);
  act(L5, 5, 4, kE, 0);
  act(L6, 6, 5, 0, 1);
  act(L7, 7, 6, 0, 1);
  act(LF1, kDzF1, 7, 0, 1);
  if (K > 1) {
    act(LS, kDzS, kActF1, 0, 1);
    for (int p = 1; p < K; ++p) emb(LS, kDzS, p, kW + (p - 1) * kE, 0);
    act(LF2, kDzF2, kActAS, 0, 1);
    add(LP, kDzKsP, A16 / 2, kActF1 * A16, A16, kW, 0, 0, 1, 0);
    add(LP, kDzKsP, A16 / 2, kActF2 * A16, A16, kW, 0, kW, 0, 0);
  } else {
    add(LP, kDzKsP, A16 / 2, kActF1 * A16, A16, kW, 0, 0, 1, 0);
  }
  add(LRGB, kDzKsRgb, 2, kActKsAP, A16 / 2, kW / 2, 0, 0, 1, 1);
  A.njobs = nj;
  return tile;
}

}  // namespace npp

using namespace npp;

extern "C" int npp_mlp_wgrad_tiles(int K) {
  if (K < 1 || K > NPP_MAX_K) { set_error("npp_mlp_wgrad_tiles: K=%d", K); return NPP_ERR_ARG; }
  WArgs A{};
  return build_jobs(K, A);
}

static int wgrad_launch(const void* d_dzT, const void* d_actT, int64_t Bp, int K, int width, int ksplit, float* d_gslabs, int M,
                        int64_t dz_stride, int64_t act_stride, int64_t slab_stride, const void* d_iter, void* stream);

extern "C" int npp_mlp_wgrad(const void* d_dzT, const void* d_actT, int64_t Bp, int K, int width, int ksplit,
                             float* d_gslabs, void* stream) {
  return wgrad_launch(d_dzT, d_actT, Bp, K, width, ksplit, d_gslabs, 0, 0, 0, 0, nullptr, stream);
}

// stacked form: M images per launch, image m's ksplit slabs at d_gslabs + m * slab_img_stride floats
extern "C" int npp_mlp_wgrad_stack(const void* d_dzT, int64_t dz_stride_bytes, const void* d_actT, int64_t act_stride_bytes,
                                   int64_t Bp, int M, int K, int width, int ksplit, float* d_gslabs, int64_t slab_img_stride,
                                   const void* d_iter, void* stream) {
  if (M < 1 || M > NPP_MAX_STACK || dz_stride_bytes % 16 || act_stride_bytes % 16 || slab_img_stride % 4) {
    set_error("npp_mlp_wgrad_stack: bad M=%d / strides", M);
    return NPP_ERR_ARG;
  }
  return wgrad_launch(d_dzT, d_actT, Bp, K, width, ksplit, d_gslabs, M, dz_stride_bytes, act_stride_bytes, slab_img_stride, d_iter, stream);
}

static int wgrad_launch(const void* d_dzT, const void* d_actT, int64_t Bp, int K, int width, int ksplit, float* d_gslabs, int M,
                        int64_t dz_stride, int64_t act_stride, int64_t slab_stride, const void* d_iter, void* stream) {
  if (K < 1 || K > NPP_MAX_K) { set_error("npp_mlp_wgrad: K=%d", K); return NPP_ERR_ARG; }
  if (width != NPP_WIDTH) { set_error("npp_mlp_wgrad: width %d unsupported (build is %d)", width, NPP_WIDTH); return NPP_ERR_UNSUPPORTED; }
  if (Bp <= 0 || Bp % kRowTile || ksplit < 1 || ksplit > 64) { set_error("npp_mlp_wgrad: bad Bp=%lld / ksplit=%d", (long long)Bp, ksplit); return NPP_ERR_ARG; }
  if (!d_dzT || !d_actT || !d_gslabs) { set_error("npp_mlp_wgrad: null pointer"); return NPP_ERR_ARG; }
  WArgs A{};
  A.dzF = (const char*)d_dzT;
  A.actF = (const char*)d_actT;
  A.n_wg = Bp / kRowTile;
  if (A.n_wg > 65536) { set_error("npp_mlp_wgrad: Bp=%lld too large (32-bit tile offsets: <= %d rows per call)", (long long)Bp, 65536 * kRowTile); return NPP_ERR_ARG; }
  const bool s8 = __atomic_load_n(&g_tune.stash8, __ATOMIC_RELAXED) != 0;     // npp_tune "stash8": both stashes are W8-format
  A.dz_bytes = s8 ? dz8_scale_base(A.n_wg) + 4 * A.n_wg : wfmt_array_base(kDzTotalKs, A.n_wg);
  A.act_bytes = wfmt_array_base(act_total_ks(K), A.n_wg);
  const int64_t act_need = A.act_bytes + (s8 ? wfmt8_array_base(act_total_ks(K), A.n_wg) : 0);
  if (s8) {                                          // the 8-bit arrays follow the 16-bit region, same k-step table
    A.actF += act8_region_base(K, A.n_wg);
    A.act_bytes = wfmt8_array_base(act_total_ks(K), A.n_wg);
  }
  A.gslabs = d_gslabs;
  A.slab_stride = slab_stride_of(make_desc(K).total_params);
  const int ntiles = build_jobs(K, A);
  A.wg_chunk = (int)((A.n_wg + ksplit - 1) / ksplit);
  if (s8 && A.wg_chunk > kScale8Max) {
    set_error("npp_mlp_wgrad: %d workgroup tiles per split exceed the %d whose scales fit in LDS: raise ksplit", A.wg_chunk, kScale8Max);
    return NPP_ERR_ARG;
  }
  static SmemOnce once, once8;
  if (!smem_attr(once, (const void*)wgrad_kernel, kSmemW)) { set_error("npp_mlp_wgrad: smem attribute"); return NPP_ERR_LAUNCH; }
  if (!smem_attr(once8, (const void*)wgrad8_kernel, kSmemW8)) { set_error("npp_mlp_wgrad: smem attribute"); return NPP_ERR_LAUNCH; }
  A.ntiles = ntiles; A.ksplit = ksplit;
  unsigned grid = (unsigned)(ntiles * ksplit);
  if (M) {
    A.S = make_stack(M, ntiles * ksplit, d_iter);
    A.dz_img_stride = dz_stride; A.act_img_stride = act_stride; A.slab_img_stride = slab_stride;
    if (dz_stride < A.dz_bytes || act_stride < act_need || slab_stride < (int64_t)ksplit * A.slab_stride) {
      set_error("npp_mlp_wgrad_stack: image strides smaller than one image's arrays");
      return NPP_ERR_ARG;
    }
    grid = stack_grid(A.S);
  }
  if (s8) hipLaunchKernelGGL(wgrad8_kernel, dim3(grid), dim3(kWThreads), kSmemW8, (hipStream_t)stream, A);
  else hipLaunchKernelGGL(wgrad_kernel, dim3(grid), dim3(kWThreads), kSmemW, (hipStream_t)stream, A);
  return check_launch("npp_mlp_wgrad");
}

// ---- the same launch over NPP_Net_light's 16-bit stashes (csrc/npp_light16.hip): seven jobs per candidate, candidate =
// image of a stacked launch.  d_gslabs: (C, ksplit, slab_stride) floats in the light blob's own layout (npp_light_desc offsets, stored
// leading dimensions); every weight and bias of the blob is written by plain stores (pad columns of the stored matrices meet zero
// inputs), so the buffer needs no clearing.
extern "C" int npp_light16_wgrad(const npp_light_desc* L, const void* d_actF, int64_t act_stride_bytes, const void* d_dzF,
                                 int64_t dz_stride_bytes, int C, int64_t B, int ksplit, float* d_gslabs, int64_t slab_stride,
                                 int64_t slab_cand_stride, void* stream) {
#if NPP_WIDTH != 256
  set_error("npp_light16_wgrad: built for W = 256 only");
  return NPP_ERR_UNSUPPORTED;
#else
  if (!L || !d_actF || !d_dzF || !d_gslabs || C < 1 || C > NPP_MAX_STACK || B < kRowTile || B % kRowTile || B / kRowTile > 65536 || ksplit < 1 ||
      ksplit > 64 || act_stride_bytes % 16 || dz_stride_bytes % 16 || slab_stride % 4 || slab_cand_stride % 4 ||
      slab_cand_stride < (int64_t)ksplit * slab_stride) {
    set_error("npp_light16_wgrad: bad arguments (C=%d <= %d, B=%lld a multiple of %d, ksplit=%d)", C, NPP_MAX_STACK, (long long)B, kRowTile, ksplit);
    return NPP_ERR_ARG;
  }
  for (int i = 0; i < 7; ++i)
    if (L->ld[i] % 4 || L->w_off[i] % 4 || L->ld[i] < L->n_in[i] || L->w_off[i] + (int64_t)L->n_out[i] * L->ld[i] > slab_stride ||
        L->b_off[i] + L->n_out[i] > slab_stride) {
      set_error("npp_light16_wgrad: layer %d: leading dimension %d / offset %lld must be multiples of 4 inside the slab", i, L->ld[i], (long long)L->w_off[i]);
      return NPP_ERR_ARG;
    }
  WArgs A{};
  A.dzF = (const char*)d_dzF;
  A.actF = (const char*)d_actF;
  A.n_wg = B / kRowTile;
  A.dz_bytes = wfmt_array_base(L16D_TOTAL, A.n_wg);
  A.act_bytes = wfmt_array_base(L16A_TOTAL, A.n_wg);
  if (dz_stride_bytes < A.dz_bytes || act_stride_bytes < A.act_bytes) { set_error("npp_light16_wgrad: candidate strides smaller than one candidate's arrays"); return NPP_ERR_ARG; }
  A.gslabs = d_gslabs;
  A.slab_stride = slab_stride;
  int nj = 0, tile = 0;
  auto add = [&](int li, int a_ks0, int a_nks, int b_ks0, int b_nks, int n, int b_is_z) {
    WJob& j = A.jobs[nj++];
    j.a_ks0 = a_ks0; j.a_nks = a_nks; j.m = L->n_out[li];
    j.b_ks0 = b_ks0; j.b_nks = b_nks; j.n = n;
    j.colmode = 0; j.col0 = 0; j.ld = L->ld[li]; j.bias_on = 1; j.b_is_z = b_is_z;
    j.w_off = L->w_off[li]; j.b_off = L->b_off[li];
    j.tile0 = tile;
    j.tiles_n = (n + kWT - 1) / kWT;
    tile += ((j.m + kWT - 1) / kWT) * j.tiles_n;
  };
  const int A16 = kKSAct;
  add(0, L16D_Z0, A16, L16A_XP, 2, kLPer, 0);                                            // periodic_linears.0: x_per
  for (int l = 1; l <= 3; ++l) add(l, L16D_Z0 + A16 * l, A16, L16A_Z0 + A16 * (l - 1), A16, kLW, 1);     // snake(z_{l-1})
  add(5, L16D_F1, A16, L16A_Z0 + A16 * 3, A16, kLW, 1);                                   // feature_linear1: snake(z_3)
  add(4, L16D_ZP, kLPosOut / 16, L16A_HP, kL16KsHp, (kLW + kLPos + 3) / 4 * 4, 0);         // pos_linears.0: [f1 | x_pos | 0 0]
  add(6, L16D_RAW, 2, L16A_ZP, kLPosOut / 16, kLPosOut, 1);                               // rgb_linear: snake(z_p)
  if (L->ld[4] < (kLW + kLPos + 3) / 4 * 4) { set_error("npp_light16_wgrad: pos_linears.0 must be stored %d wide", (kLW + kLPos + 3) / 4 * 4); return NPP_ERR_ARG; }
  A.njobs = nj;
  A.wg_chunk = (int)((A.n_wg + ksplit - 1) / ksplit);
  static SmemOnce once;
  if (!smem_attr(once, (const void*)wgrad_kernel, kSmemW)) { set_error("npp_light16_wgrad: smem attribute"); return NPP_ERR_LAUNCH; }
  A.ntiles = tile; A.ksplit = ksplit;
  A.S = make_stack(C, tile * ksplit, nullptr);
  A.dz_img_stride = dz_stride_bytes; A.act_img_stride = act_stride_bytes; A.slab_img_stride = slab_cand_stride;
  hipLaunchKernelGGL(wgrad_kernel, dim3(stack_grid(A.S)), dim3(kWThreads), kSmemW, (hipStream_t)stream, A);
  return check_launch("npp_light16_wgrad");
#endif
}
