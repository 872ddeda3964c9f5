/* crux_rng.h -- counter-based randomness spec of libcruxhip (plain C, usable from HIP device code).
 *
 * The reference (sisl/Crux.jl) draws from Julia's task-local Xoshiro256++ stream
 * (src/sampler.jl:73 -> src/policies.jl:137-142 `rand(Categorical(v))`, src/experience_buffer.jl:119
 * `shuffle(1:length(b))`, :318 `rand(1:length(source), B)`, :337 `rand(B)`); that stream cannot be
 * reproduced outside Julia, so this library DEFINES every random draw as a pure function of
 * (seed, stream, counter, purpose) using Philox4x32-10 (Salmon et al., SC'11, public algorithm).
 * Both the HIP kernels and the CPU oracle evaluate the same functions, which is what makes
 * seeded parity tests possible without injecting arrays of random numbers.
 */
#ifndef CRUX_RNG_H
#define CRUX_RNG_H
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define CRUX_HD __host__ __device__ static inline
#else
#define CRUX_HD static inline
#endif

/* purpose tags (4th counter word) */
enum {
  CRUX_RNG_ACTION = 1,     /* uniform for categorical sampling / eps-greedy test             */
  CRUX_RNG_NOISE = 2,      /* standard normal for Gaussian heads / exploration noise         */
  CRUX_RNG_RESET = 3,      /* initial-state draws                                            */
  CRUX_RNG_ENVDYN = 4,     /* stochastic env transitions (GridWorld slip, synthetic envs)    */
  CRUX_RNG_SAMPLE = 5,     /* replay-buffer sampling                                         */
  CRUX_RNG_SHUFFLE = 6,    /* epoch permutations                                             */
  CRUX_RNG_INIT = 7,       /* glorot-uniform parameter init                                  */
  CRUX_RNG_RANDACT = 8,    /* the random action of an eps-greedy draw                        */
  CRUX_RNG_RESERVOIR = 9   /* push_reservoir!: v[0..1] -> rand() of the weight test, v[2..3] -> rand(1:total_count) */
};

typedef struct { uint32_t v[4]; } crux_u32x4;

CRUX_HD uint32_t crux_mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }

CRUX_HD crux_u32x4 crux_philox(uint64_t seed, uint64_t counter, uint32_t stream, uint32_t purpose) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  uint32_t c0 = (uint32_t)counter, c1 = (uint32_t)(counter >> 32), c2 = stream, c3 = purpose;
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = crux_mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = crux_mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  crux_u32x4 o; o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3; return o;
}

/* Float32 uniform in [0,1): 24 high bits (what `rand(Float32)` means in Julia: multiples of 2^-24). */
CRUX_HD float crux_u32_to_f32(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }
/* Float64 uniform in [0,1): 53 bits from two words. */
CRUX_HD double crux_u32x2_to_f64(uint32_t hi, uint32_t lo) {
  uint64_t x = (((uint64_t)hi << 32) | (uint64_t)lo) >> 11;
  return (double)x * 1.1102230246251565e-16;
}

/* Pseudo-random permutation of [0,n): 4-round Feistel network on ceil(log2 n) bits keyed by Philox
 * words, cycle-walked into range. perm(j) is O(1), needs no memory and is a bijection for every key.
 * Stands in for `shuffle(1:n)` (src/experience_buffer.jl:119). */
typedef struct { uint32_t key[4]; uint32_t lbits, rbits, n; } crux_perm;

CRUX_HD crux_perm crux_perm_make(uint64_t seed, uint64_t epoch_counter, uint32_t stream, uint32_t n) {
  crux_perm p; crux_u32x4 k = crux_philox(seed, epoch_counter, stream, CRUX_RNG_SHUFFLE);
  uint32_t bits = 1; while (((uint64_t)1 << bits) < (uint64_t)n) ++bits;
  if (bits < 2) bits = 2;
  p.lbits = bits / 2; p.rbits = bits - p.lbits; p.n = n;
  for (int i = 0; i < 4; ++i) p.key[i] = k.v[i];
  return p;
}

CRUX_HD uint32_t crux_perm_round(uint32_t x, uint32_t key) {
  x ^= key; x *= 0x9E3779B1u; x ^= x >> 15; x *= 0x85EBCA77u; x ^= x >> 13; x *= 0xC2B2AE3Du; x ^= x >> 16;
  return x;
}

CRUX_HD uint32_t crux_perm_at(const crux_perm* p, uint32_t j) {
  const uint32_t lmask = (1u << p->lbits) - 1u, rmask = (1u << p->rbits) - 1u;
  uint32_t x = j;
  do {
    uint32_t l = x >> p->rbits, r = x & rmask;            /* l: lbits wide, r: rbits wide */
    for (int i = 0; i < 4; ++i) {
      /* alternate unbalanced Feistel halves so widths stay (lbits, rbits) after two rounds */
      if ((i & 1) == 0) { l = (l ^ crux_perm_round(r, p->key[i])) & lmask; }
      else              { r = (r ^ crux_perm_round(l, p->key[i])) & rmask; }
    }
    x = (l << p->rbits) | r;
  } while (x >= p->n);
  return x;
}

#endif /* CRUX_RNG_H */
