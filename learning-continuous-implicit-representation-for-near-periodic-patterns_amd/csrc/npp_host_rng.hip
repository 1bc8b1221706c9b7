// npp_host_rng.hip -- host-only: NumPy's legacy RandomState stream, bit for bit, for the loop's sampler.
//
// The reference draws its per-iteration indices from the GLOBAL np.random state: np.random.uniform (patch source,
// models/sampler.py:324) and np.random.choice(n, size, replace=False) (patch centres :260, the N_rand pixel rows
// NPP_completion/train.py:172).  choice(replace=False) is permutation(n)[:size], i.e. a FULL Fisher-Yates shuffle of the
// population per call (245 k elements for the pixel rows of a 512^2 image): 1.9 ms in NumPy, under the GIL -- more than
// twice the device time of the whole iteration.  This file restates the three primitives (MT19937 init_genrand seeding,
// the 53-bit double, the masked-rejection random_interval shuffle) so that the same stream comes out of a GIL-free native
// call; tests/test_host_rng.py compares state words and outputs against numpy.random.RandomState.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#if defined(__x86_64__)
#include <immintrin.h>
#define NPP_HOST_X86 1
#else
#define NPP_HOST_X86 0          // other hosts: the scalar forms (bit-identical), std::this_thread::yield() in the spin loops
#endif

#include <atomic>
#include <thread>
#include <vector>

#include "npp_hip.h"

namespace npp { void set_error(const char* fmt, ...); }   // npp_api.hip

namespace {

struct MT {
  uint32_t key[624];       // the raw state words (numpy's get_state()[1])
  uint32_t out[624];       // the same words tempered, produced a block at a time (the loop vectorises)
  int pos;
};

void mt_temper(MT* s) {
  for (int i = 0; i < 624; ++i) {
    uint32_t y = s->key[i];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    s->out[i] = y;
  }
}

void mt_seed(MT* s, uint32_t seed) {          // init_genrand (numpy _legacy_seeding with an integer seed)
  s->key[0] = seed;
  for (int i = 1; i < 624; ++i) s->key[i] = 1812433253u * (s->key[i - 1] ^ (s->key[i - 1] >> 30)) + (uint32_t)i;
  s->pos = 624;
}

void mt_gen(MT* s) {
  const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX = 0x9908b0dfu;
  uint32_t* k = s->key;
  int i;
  for (i = 0; i < 624 - 397; ++i) {
    const uint32_t y = (k[i] & UPPER) | (k[i + 1] & LOWER);
    k[i] = k[i + 397] ^ (y >> 1) ^ ((y & 1u) ? MATRIX : 0u);
  }
  for (; i < 623; ++i) {
    const uint32_t y = (k[i] & UPPER) | (k[i + 1] & LOWER);
    k[i] = k[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MATRIX : 0u);
  }
  const uint32_t y = (k[623] & UPPER) | (k[0] & LOWER);
  k[623] = k[396] ^ (y >> 1) ^ ((y & 1u) ? MATRIX : 0u);
  mt_temper(s);
  s->pos = 0;
}

inline uint32_t mt_next32(MT* s) {
  if (s->pos == 624) mt_gen(s);
  return s->out[s->pos++];
}

inline uint64_t mt_next64(MT* s) {
  const uint64_t hi = mt_next32(s);
  return (hi << 32) | mt_next32(s);
}

inline double mt_double(MT* s) {              // mt19937_next_double
  const int32_t a = (int32_t)(mt_next32(s) >> 5), b = (int32_t)(mt_next32(s) >> 6);
  return (a * 67108864.0 + b) / 9007199254740992.0;
}

inline uint64_t interval(MT* s, uint64_t max) {   // legacy random_interval: masked rejection
  if (max == 0) return 0;
  uint64_t mask = max, value;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
  if (max <= 0xffffffffULL) {
    while ((value = (mt_next32(s) & mask)) > max) {}
  } else {
    while ((value = (mt_next64(s) & mask)) > max) {}
  }
  return value;
}

#if NPP_HOST_X86
// ---- AVX2 forms (x86 hosts that have it; chosen once at run time, NPP_RNG_AVX2=0 turns them off) -----------------------------
__attribute__((target("avx2"))) void mt_gen_avx2(MT* s) {      // mt_gen + mt_temper, the same loops compiled 8 wide
  const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX = 0x9908b0dfu;
  uint32_t* k = s->key;
  int i;
  for (i = 0; i < 624 - 397; ++i) {
    const uint32_t y = (k[i] & UPPER) | (k[i + 1] & LOWER);
    k[i] = k[i + 397] ^ (y >> 1) ^ ((y & 1u) ? MATRIX : 0u);
  }
  for (; i < 623; ++i) {
    const uint32_t y = (k[i] & UPPER) | (k[i + 1] & LOWER);
    k[i] = k[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MATRIX : 0u);
  }
  const uint32_t y = (k[623] & UPPER) | (k[0] & LOWER);
  k[623] = k[396] ^ (y >> 1) ^ ((y & 1u) ? MATRIX : 0u);
  for (i = 0; i < 624; ++i) {
    uint32_t t = k[i];
    t ^= (t >> 11);
    t ^= (t << 7) & 0x9d2c5680u;
    t ^= (t << 15) & 0xefc60000u;
    t ^= (t >> 18);
    s->out[i] = t;
  }
  s->pos = 0;
}

// compaction table: for an 8-bit accept mask, the lane order that moves the accepted lanes to the front
struct CompactLut {
  alignas(32) uint32_t idx[256][8];
  CompactLut() {
    for (int m = 0; m < 256; ++m) {
      int c = 0;
      for (int k = 0; k < 8; ++k) if (m >> k & 1) idx[m][c++] = (uint32_t)k;
      for (int k = 0; k < 8; ++k) if (!(m >> k & 1)) idx[m][c++] = (uint32_t)k;
    }
  }
};
const CompactLut& compact_lut() { static const CompactLut t; return t; }

// The masked-rejection walk of the shuffle (gen_block) on runs of 64 words: inside a run the bound moves down by < 64, so a word
// v <= bound - 64 is accepted and v > bound rejected whatever came before it; a word in the band between (probability 64 / bound)
// ends the fast path and leaves the run to the exact loop.  With the thresholds fixed over the run, the accept masks of its eight
// groups do not depend on each other; the only carried value is the store index (one popcount-add per group).  Returns the number
// of words consumed (a multiple of 64).  jbuf needs 8 words of slack behind *cnt.
__attribute__((target("avx2"))) int compact_runs_avx2(const uint32_t* w, int avail, uint32_t mask, uint32_t lo, uint32_t* jbuf, int* cnt_io,
                                                       uint32_t* bound_io, int cnt_max) {
  const CompactLut& lut = compact_lut();
  int u = 0, cnt = *cnt_io;
  uint32_t bound = *bound_io;
  const __m256i vmask = _mm256_set1_epi32((int)mask);
  while (u + 64 <= avail && cnt + 64 <= cnt_max && bound > lo + 64) {
    const __m256i vs = _mm256_set1_epi32((int)(bound - 64)), vb = _mm256_set1_epi32((int)bound);
    __m256i v[8];
    int m[8], band = 0;
    for (int g = 0; g < 8; ++g) {
      v[g] = _mm256_and_si256(_mm256_loadu_si256((const __m256i*)(w + u + 8 * g)), vmask);
      m[g] = _mm256_movemask_ps(_mm256_castsi256_ps(_mm256_cmpeq_epi32(_mm256_max_epu32(v[g], vs), vs)));        // v <= bound - 64 (unsigned)
      band |= m[g] ^ _mm256_movemask_ps(_mm256_castsi256_ps(_mm256_cmpeq_epi32(_mm256_max_epu32(v[g], vb), vb)));   // ... xor v <= bound
    }
    if (band) break;
    int c = cnt;
    for (int g = 0; g < 8; ++g) {
      _mm256_storeu_si256((__m256i*)(jbuf + c), _mm256_permutevar8x32_epi32(v[g], _mm256_load_si256((const __m256i*)lut.idx[m[g]])));
      c += __builtin_popcount((unsigned)m[g]);
    }
    bound -= (uint32_t)(c - cnt);
    cnt = c;
    u += 64;
  }
  *cnt_io = cnt;
  *bound_io = bound;
  return u;
}
#else
static void mt_gen_avx2(MT*) {}
static int compact_runs_avx2(const uint32_t*, int, uint32_t, uint32_t, uint32_t*, int*, uint32_t*, int) { return 0; }
#endif
static inline void cpu_relax() {              // spin politely: pause, and every 1024th spin give the core away (the other side
  static thread_local unsigned spins = 0;   // of the hand-off may be waiting for it on a host with few cores per rank)
  if ((++spins & 1023u) == 0) { std::this_thread::yield(); return; }
#if NPP_HOST_X86
  __builtin_ia32_pause();
#else
  std::this_thread::yield();
#endif
}

}  // namespace

extern "C" void* npp_rng_create(uint32_t seed) {
  MT* s = (MT*)malloc(sizeof(MT));
  if (s) mt_seed(s, seed);
  return s;
}

extern "C" void npp_rng_destroy(void* h) { free(h); }

extern "C" int npp_rng_seed(void* h, uint32_t seed) {
  if (!h) return NPP_ERR_ARG;
  mt_seed((MT*)h, seed);
  return NPP_OK;
}

/* key[624] + pos, the layout of numpy's RandomState.get_state()[1:3] */
extern "C" int npp_rng_get_state(void* h, uint32_t* key624, int32_t* pos) {
  if (!h || !key624 || !pos) return NPP_ERR_ARG;
  memcpy(key624, ((MT*)h)->key, sizeof(uint32_t) * 624);
  *pos = ((MT*)h)->pos;
  return NPP_OK;
}

extern "C" int npp_rng_set_state(void* h, const uint32_t* key624, int32_t pos) {
  if (!h || !key624 || pos < 0 || pos > 624) return NPP_ERR_ARG;
  memcpy(((MT*)h)->key, key624, sizeof(uint32_t) * 624);
  mt_temper((MT*)h);
  ((MT*)h)->pos = pos;
  return NPP_OK;
}

/* np.random.uniform(lo, hi): lo + (hi - lo) * next_double */
extern "C" double npp_rng_uniform(void* h, double lo, double hi) { return lo + (hi - lo) * mt_double((MT*)h); }

/* np.random.choice(n, size, replace=False) == permutation(n)[:size]: arange(n), Fisher-Yates from the top with
 * random_interval, first `size` entries.  scratch: caller-owned int64[n] (kept between calls to avoid reallocating).
 *
 * The straightforward loop (draw j, swap, repeat) spends its time on one unpredictable branch per draw (the masked
 * rejection accepts 50 - 100 % of the words) and one dependent random access per swap.  Here the walk is blocked: a
 * branch-free pass turns raw words into the next <= 2048 accepted j's (the compaction index advances by the accept
 * bit; the mask follows the shrinking bound), then the swaps of those steps run with the targets prefetched ahead.  The
 * permutation is kept as int32 inside the caller's scratch (1 MB instead of 2 MB for the pixel rows of a 512^2 image).
 * Same words consumed in the same order: bit-identical to the legacy generator (tests/test_host_rng.py).
 * Round 3: measured on the 1 048 576-element shuffles of a 1024^2 image (two per iteration, more host time than the iteration's
 * device time), generation -- not the swaps -- was the longer half: the accept/compact walk carried compare -> subtract from word
 * to word (3 cycles per word) on top of the generator itself (1.5).  Now the walk runs in groups whose decisions cannot depend on
 * each other (a word far enough below the bound is accepted whatever came before; a word in the narrow band under the bound sends
 * the group to the exact loop), 64 words at a time with AVX2 compares + a table-driven lane compaction where the host has AVX2, and
 * the generator's loops are compiled 8 wide there: 3.5 -> 2.1 ms per 1 M-element shuffle on the build container's Xeon (the swaps,
 * prefetched 32 ahead, are now the longer half), 0.85 -> 0.52 ms for the 245 k pixel rows of a 512^2 image. */
extern "C" int npp_rng_choice_noreplace(void* h, int64_t n, int64_t size, int64_t* scratch, int64_t* out) {
  if (!h || !scratch || !out || n < 1 || size < 0 || size > n) return NPP_ERR_ARG;
  MT* s = (MT*)h;
  if (n > 0x7fffffffLL) {                              // 64-bit bounds: the plain loop
    for (int64_t i = 0; i < n; ++i) scratch[i] = i;
    for (int64_t i = n - 1; i > 0; --i) {
      const int64_t j = (int64_t)interval(s, (uint64_t)i);
      const int64_t t = scratch[i];
      scratch[i] = scratch[j];
      scratch[j] = t;
    }
    memcpy(out, scratch, sizeof(int64_t) * (size_t)size);
    return NPP_OK;
  }
  uint32_t* perm = (uint32_t*)scratch;
  constexpr int kBlock = 2048, kAhead = 32;
  // generation of one block: the next <= kBlock accepted targets for the bounds i, i - 1, ... (consumes generator words)
#if NPP_HOST_X86
  static const bool have_avx2 = [] { const char* e = getenv("NPP_RNG_AVX2"); return !(e && e[0] == '0') && __builtin_cpu_supports("avx2"); }();
#else
  static const bool have_avx2 = false;
#endif
  auto gen_block = [s](uint32_t i, uint32_t* jbuf) -> int {
    int cnt = 0;
    uint32_t bound = i;
    while (cnt < kBlock && bound > 0) {
      // the mask is constant while the bound stays inside (mask / 2, mask]; words are taken straight from the tempered
      // block: the only loop-carried chain is compare -> subtract
      const uint32_t mask = 0xffffffffu >> __builtin_clz(bound), lo = mask >> 1;
      if (s->pos == 624) { if (have_avx2) mt_gen_avx2(s); else mt_gen(s); }
      const uint32_t* w = s->out + s->pos;
      const int avail = 624 - s->pos;
      int u = 0;
      // (AVX2 hosts: runs of 64 first.)  Eight words at a time while nothing in the group can depend on the group's own earlier decisions: inside a group the bound
      // only moves down by < 8, so v <= bound - 8 is accepted and v > bound is rejected whatever came before; a word in the 8-wide
      // band between (probability 8 / bound) sends the group to the exact loop below.  The compares are then independent of each
      // other and the only carried chain is the store index (one add per word) -- the exact loop carries compare -> subtract.
      if (have_avx2) u = compact_runs_avx2(w, avail, mask, lo, jbuf, &cnt, &bound, kBlock);
      while (u + 8 <= avail && cnt + 8 <= kBlock && bound > lo + 8) {
        const uint32_t sure = bound - 8;
        uint32_t v[8], band = 0;
        for (int k = 0; k < 8; ++k) {
          v[k] = w[u + k] & mask;
          band |= (uint32_t)(v[k] > sure) & (uint32_t)(v[k] <= bound);
        }
        if (band) break;
        int c = cnt;
        for (int k = 0; k < 8; ++k) {
          jbuf[c] = v[k];
          c += (int)(v[k] <= sure);
        }
        bound -= (uint32_t)(c - cnt);
        cnt = c;
        u += 8;
      }
      for (int k = 0; k < 8 && u < avail && cnt < kBlock && bound > lo; ++k) {      // the exact form: one group's worth, then retry the fast one
        const uint32_t v = w[u++] & mask;
        jbuf[cnt] = v;
        const uint32_t acc = v <= bound;
        cnt += (int)acc;
        bound -= acc;
      }
      s->pos += u;
    }
    for (int k = cnt; k < cnt + kAhead; ++k) jbuf[k] = 0;
    return cnt;
  };
  auto apply_block = [perm](uint32_t i, const uint32_t* jbuf, int cnt) {
    for (int k = 0; k < cnt; ++k) {
      __builtin_prefetch(&perm[jbuf[k + kAhead]], 1, 3);
      const uint32_t j = jbuf[k], t = perm[i - (uint32_t)k];
      perm[i - (uint32_t)k] = perm[j];
      perm[j] = t;
    }
  };
  // Large populations (the 1 048 576 pixel rows / patch-centre pool of a 1024^2 image: two such shuffles per iteration, 3.5 ms,
  // more than the device time of the iteration): the two halves of a block do not depend on each other beyond the targets --
  // generating them needs only the generator, applying them only the array -- so a helper thread generates blocks into a small
  // ring while this thread applies them.  Same words in the same order; the generator is touched by the helper alone until
  // it is joined.  (A thread per call: ~30 us against >= 1 ms of work; below kThreadedMin the plain loop.)
  constexpr int64_t kThreadedMin = 200000;
  // ... measured on the GPU box's EPYC 9575F once generation had its AVX2 form: 1.19 ms for 1 M elements on ONE thread against 1.36
  // with the helper (thread start + a hand-off per 2048 targets cost more than the overlap still buys); without AVX2 the helper wins
  // (2.0 vs 1.75 ... on the build container's Xeon 3.0 vs 4.1).  Default: helper thread only where generation is the scalar form;
  // NPP_RNG_THREADS=0 / 1 force it off / on.
  static const bool threaded_ok = [] { const char* e = getenv("NPP_RNG_THREADS"); return e ? e[0] != '0' : !have_avx2; }();
  constexpr int kRing = 8;
  struct Slot { uint32_t j[kBlock + kAhead]; uint32_t i; int cnt; };
  std::vector<Slot> ring;
  std::atomic<int64_t> produced{0}, consumed{0};
  std::atomic<bool> done{false};
  const uint32_t top = (uint32_t)(n - 1);
  bool threaded = n >= kThreadedMin && threaded_ok;
  std::thread gen;
  if (threaded) {
    // (an extern "C" entry point must not let an exception escape: if the helper thread cannot be started -- EAGAIN under a
    //  container's pid limit -- or its ring not be allocated, the single-thread loop below does the same work)
    try {
      ring.resize(kRing);
      gen = std::thread([&] {
      uint32_t i = top;
      int64_t b = 0;
      while (i > 0) {
        while (b - consumed.load(std::memory_order_acquire) >= kRing) cpu_relax();
        Slot& sl = ring[b % kRing];
        sl.i = i;
        sl.cnt = gen_block(i, sl.j);
        i -= (uint32_t)sl.cnt;
        ++b;
        produced.store(b, std::memory_order_release);
      }
      done.store(true, std::memory_order_release);
      });
    } catch (...) {
      threaded = false;
    }
  }
  if (threaded) {
    for (uint32_t q = 0; q < (uint32_t)n; ++q) perm[q] = q;          // (overlaps the first blocks' generation)
    int64_t b = 0;
    for (;;) {
      while (produced.load(std::memory_order_acquire) <= b) {
        if (done.load(std::memory_order_acquire) && produced.load(std::memory_order_acquire) <= b) goto finished;
        cpu_relax();
      }
      const Slot& sl = ring[b % kRing];
      apply_block(sl.i, sl.j, sl.cnt);
      ++b;
      consumed.store(b, std::memory_order_release);
    }
  finished:
    gen.join();
  } else {
    for (uint32_t q = 0; q < (uint32_t)n; ++q) perm[q] = q;
    uint32_t jbuf[kBlock + kAhead];
    uint32_t i = (uint32_t)(n - 1);
    while (i > 0) {
      const int cnt = gen_block(i, jbuf);
      apply_block(i, jbuf, cnt);
      i -= (uint32_t)cnt;
    }
  }
  for (int64_t k = size - 1; k >= 0; --k) out[k] = (int64_t)perm[k];   // out may alias scratch: widen from the top
  return NPP_OK;
}

// ---- the sampler's per-iteration draw, host side (models/sampler.py:242-354, NPP_completion/train.py:152-172) ----------
// What GridPatchSampler.sample_patches decides on the host -- patch source (:324), fake-patch centres (:260), and per fake
// patch the k nearest lattice candidates c + a s1 + b s2, a, b in [-10, 10) (:148-214), kept when they lie inside the image
// and their P x P window has at most invalid_ratio * P^2 unknown pixels -- plus the N_rand pixel rows of train.py:172, in
// the reference's RNG order, in ONE GIL-free call.  The unknown-pixel counts come from a summed-area table of the known
// mask (4 look-ups per candidate instead of cropping it; zero padding counts as unknown, sampler.py:181).  Mirrors
// sampler.py's Python restatement (npp_amd/sampler.py: GridPatchSampler.draw) operation by operation in float64, so both
// produce identical draws (tests/test_host_rng.py).
namespace {

struct Sampler {
  int H, W, half, n_samples;
  int64_t* sat;                       // (H + 1) x (W + 1) summed-area table of known pixels
  int32_t* pool[2];                   // [0] train, [1] val: (row, col) pairs that satisfy the margin rule of reset_pool
  int64_t pool_n[2], pool_cap[2];
  int32_t* raw[2];                    // unfiltered pools
  int64_t raw_n[2];
  double shift[2][2];                 // (dy, dx) of the two lattice shifts (sampler.py:35 flips the (dx, dy) of config.odgt)
  int64_t* scratch;                   // permutation scratch for choice(replace=False)
  int64_t scratch_n;
};

inline int64_t clampi(int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

int64_t unknown_count(const Sampler* S, double cy, double cx, int P) {
  // np.rint: round half to even
  const int64_t y = (int64_t)__builtin_rint(cy), x = (int64_t)__builtin_rint(cx);
  const int64_t y0 = clampi(y - P / 2, 0, S->H), y1 = clampi(y + P / 2, 0, S->H);
  const int64_t x0 = clampi(x - P / 2, 0, S->W), x1 = clampi(x + P / 2, 0, S->W);
  const int64_t w = S->W + 1;
  const int64_t known = S->sat[y1 * w + x1] - S->sat[y0 * w + x1] - S->sat[y1 * w + x0] + S->sat[y0 * w + x0];
  return (int64_t)P * P - known;
}

void filter_pool(Sampler* S, int which) {        // sampler.py:102-124 reset_pool
  const int h = S->half;
  int64_t n = 0;
  for (int64_t i = 0; i < S->raw_n[which]; ++i) {
    const int32_t r = S->raw[which][2 * i], c = S->raw[which][2 * i + 1];
    if (r > h && r < S->H - (h + 1) && c > h && c < S->W - (h + 1)) {
      S->pool[which][2 * n] = r;
      S->pool[which][2 * n + 1] = c;
      ++n;
    }
  }
  S->pool_n[which] = n;
}

}  // namespace

extern "C" void* npp_sampler_create(const int64_t* sat, int H, int W, const int32_t* pool_train, int64_t n_train,
                                    const int32_t* pool_val, int64_t n_val, const double* shifts_dydx) {
  if (!sat || H < 1 || W < 1 || !pool_train || !pool_val || n_train < 0 || n_val < 0 || !shifts_dydx) return nullptr;
  Sampler* S = (Sampler*)calloc(1, sizeof(Sampler));
  if (!S) return nullptr;
  S->H = H; S->W = W;
  const size_t sat_n = (size_t)(H + 1) * (W + 1);
  S->sat = (int64_t*)malloc(sat_n * sizeof(int64_t));
  const int64_t ns[2] = {n_train, n_val};
  const int32_t* src[2] = {pool_train, pool_val};
  bool ok = S->sat != nullptr;
  for (int w = 0; w < 2 && ok; ++w) {
    S->raw_n[w] = ns[w];
    S->raw[w] = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)(ns[w] + 1));
    S->pool[w] = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)(ns[w] + 1));
    ok = S->raw[w] && S->pool[w];
    if (ok) memcpy(S->raw[w], src[w], sizeof(int32_t) * 2 * (size_t)ns[w]);
  }
  const int64_t nmax = n_train > n_val ? n_train : n_val;
  S->scratch = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nmax + 1));
  S->scratch_n = nmax;
  if (!ok || !S->scratch) { npp_sampler_destroy(S); return nullptr; }
  memcpy(S->sat, sat, sat_n * sizeof(int64_t));
  for (int i = 0; i < 4; ++i) S->shift[i / 2][i % 2] = shifts_dydx[i];
  return S;
}

extern "C" void npp_sampler_destroy(void* h) {
  Sampler* S = (Sampler*)h;
  if (!S) return;
  free(S->sat); free(S->scratch);
  for (int w = 0; w < 2; ++w) { free(S->raw[w]); free(S->pool[w]); }
  free(S);
}

/* reset_patchsize + reset_pool (sampler.py:49-124): patch size, number of fake patches, margin-filtered pools */
extern "C" int npp_sampler_set_patch(void* h, int patch_size, int n_samples, int64_t* pool_train_n, int64_t* pool_val_n) {
  Sampler* S = (Sampler*)h;
  if (!S || patch_size < 2 || n_samples < 1) return NPP_ERR_ARG;
  S->half = patch_size / 2;
  S->n_samples = n_samples;
  filter_pool(S, 0);
  filter_pool(S, 1);
  if (pool_train_n) *pool_train_n = S->pool_n[0];
  if (pool_val_n) *pool_val_n = S->pool_n[1];
  return NPP_OK;
}

/* One sample_patches() worth of host decisions.  Outputs: source (0 val, 1 train, 2 same), k (0: no valid real patch ->
 * the iteration is skipped), cen int32[n][2], real_cen double[n * topk][2] (first n * k rows valid, candidate order),
 * weights float[n * topk] (1/d normalised per fake patch, first n * k valid).  Consumes the generator exactly like
 * np.random.uniform(0, 1) followed by np.random.choice(pool_n, [n], replace=False). */
extern "C" int npp_sampler_draw(void* h, void* rng, int topk, double invalid_ratio, int32_t* source, int32_t* k_out,
                                int32_t* cen, double* real_cen, float* weights) {
  Sampler* S = (Sampler*)h;
  if (!S || !rng || topk < 1 || topk > 16 || !source || !k_out || !cen || !real_cen || !weights || S->half < 1) {
    npp::set_error("npp_sampler_draw: bad arguments (topk=%d in [1, 16], patch size set?)", topk);
    return NPP_ERR_ARG;
  }
  const int n = S->n_samples, P = 2 * S->half;
  const double prob = npp_rng_uniform(rng, 0.0, 1.0);
  const int src = prob < 0.5 ? 0 : ((0.5 < prob && prob < 0.8) ? 1 : 2);          // sampler.py:326-331
  *source = src;
  const int which = src == 0 ? 1 : 0;                                                // 'val' -> pool_val, else pool_train
  if (S->pool_n[which] < n) {            // np.random.choice(pool, [n], replace=False) raises ValueError in the reference
    npp::set_error("npp_sampler_draw: %lld %s centres are at least half a patch (%d) from the border, %d patches asked for "
                   "(patch larger than the image region it samples?)", (long long)S->pool_n[which], which ? "unknown-region" : "known-region", S->half, n);
    return NPP_ERR_ARG;
  }
  int64_t sel[64];
  if (n > 64) { npp::set_error("npp_sampler_draw: %d patches per iteration (<= 64)", n); return NPP_ERR_UNSUPPORTED; }
  int rc = npp_rng_choice_noreplace(rng, S->pool_n[which], n, S->scratch, sel);
  if (rc) return rc;
  for (int i = 0; i < n; ++i) {
    cen[2 * i] = S->pool[which][2 * sel[i]];
    cen[2 * i + 1] = S->pool[which][2 * sel[i] + 1];
  }
  if (src == 2) {                                                                    // 'same': real := fake, k = 1
    *k_out = 1;
    for (int i = 0; i < n; ++i) weights[i] = 1.0f;
    return NPP_OK;
  }
  const double thresh = (double)(P * P) * invalid_ratio;                             // sampler.py:181
  int topk_min = topk;
  struct Cand { double y, x, d; };
  Cand cand[400];
  int kept_k[64];
  for (int i = 0; i < n; ++i) {
    int m = 0;
    for (int ai = -10; ai < 10; ++ai)
      for (int bi = -10; bi < 10; ++bi) {                                            // meshgrid(indexing='ij'): a outer, b inner
        const double a = (double)ai, b = (double)bi;
        double y = (double)cen[2 * i] + a * S->shift[0][0];
        y = y + b * S->shift[1][0];
        double x = (double)cen[2 * i + 1] + a * S->shift[0][1];
        x = x + b * S->shift[1][1];
        if (!(y > 0 && y < S->H - 1 && x > 0 && x < S->W - 1)) continue;
        if ((double)unknown_count(S, y, x, P) > thresh) continue;
        double d = (double)((ai < 0 ? -ai : ai) + (bi < 0 ? -bi : bi));
        if (d == 0) d = 10000;                                                       // :197 exclude itself
        cand[m++] = Cand{y, x, d};
      }
    const int avail = (m - 1 < topk) ? m - 1 : topk;
    if (avail < topk_min) {
      topk_min = avail;
      if (topk_min <= 0) { *k_out = 0; return NPP_OK; }
    }
    // stable selection of the topk_min smallest distances (np.argsort(kind='stable'))
    double wsum = 0.0, inv[16];
    bool used[400];
    for (int c = 0; c < m; ++c) used[c] = false;
    for (int t = 0; t < topk_min; ++t) {
      int best = -1;
      for (int c = 0; c < m; ++c)
        if (!used[c] && (best < 0 || cand[c].d < cand[best].d)) best = c;
      used[best] = true;
      real_cen[2 * (i * topk + t)] = cand[best].y;
      real_cen[2 * (i * topk + t) + 1] = cand[best].x;
      inv[t] = 1.0 / cand[best].d;
      wsum += inv[t];
    }
    for (int t = 0; t < topk_min; ++t) weights[i * topk + t] = (float)(inv[t] / wsum);
    kept_k[i] = topk_min;
  }
  *k_out = topk_min;
  // fake patches that were processed while topk_min was still larger keep their first topk_min candidates (:205-209);
  // their weights stay normalised over the number they had (as in the reference, which slices without renormalising)
  (void)kept_k;
  return NPP_OK;
}
