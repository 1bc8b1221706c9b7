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

#include "npp_hip.h"

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
 * Same words consumed in the same order: bit-identical to the legacy generator (tests/test_host_rng.py). */
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
  for (uint32_t i = 0; i < (uint32_t)n; ++i) perm[i] = i;
  constexpr int kBlock = 2048, kAhead = 12;
  uint32_t jbuf[kBlock + kAhead];
  uint32_t i = (uint32_t)(n - 1);
  while (i > 0) {
    int cnt = 0;
    uint32_t bound = i;
    while (cnt < kBlock && bound > 0) {
      // the mask is constant while the bound stays inside (mask / 2, mask]; words are taken straight from the tempered
      // block: the only loop-carried chain is compare -> subtract
      const uint32_t mask = 0xffffffffu >> __builtin_clz(bound), lo = mask >> 1;
      if (s->pos == 624) mt_gen(s);
      const uint32_t* w = s->out + s->pos;
      const int avail = 624 - s->pos;
      int u = 0;
      while (u < avail && cnt < kBlock && bound > lo) {
        const uint32_t v = w[u++] & mask;
        jbuf[cnt] = v;
        const uint32_t acc = v <= bound;
        cnt += (int)acc;
        bound -= acc;
      }
      s->pos += u;
    }
    for (int k = cnt; k < cnt + kAhead; ++k) jbuf[k] = 0;
    for (int k = 0; k < cnt; ++k) {
      __builtin_prefetch(&perm[jbuf[k + kAhead]], 1, 1);
      const uint32_t j = jbuf[k], t = perm[i - (uint32_t)k];
      perm[i - (uint32_t)k] = perm[j];
      perm[j] = t;
    }
    i -= (uint32_t)cnt;
  }
  for (int64_t k = size - 1; k >= 0; --k) out[k] = (int64_t)perm[k];   // out may alias scratch: widen from the top
  return NPP_OK;
}
