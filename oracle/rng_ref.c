/*
 * oracle/rng_ref.c -- TEST INFRASTRUCTURE ONLY (see ora.h).
 *
 * Replay-index sampling of the reference: DeepQNAg::sample_transition
 * (agent/deepqn/include/DeepQNAg.hpp:275-288) draws
 *     std::uniform_int_distribution<int>(0, from.size()-1)(*bib::Seed::random_engine())
 * on a std::mt19937 (common/include/bib/Seed.hpp:47-57).  mt19937 is specified bit-for-bit by
 * the C++ standard (10000th output of the default-seeded engine = 4123659995).  The distribution
 * is a libstdc++ implementation detail: this restates GCC >= 11's algorithm (Lemire's nearly
 * divisionless method, bits/uniform_int_dist.h `_S_nd`), which is what g++ 11.4 in this image
 * gives; oracle/pin_stdlib.cpp checks this file against the real std:: classes.
 */
#include "ora.h"

void ora_mt_seed(ora_mt19937 *g, uint32_t seed) {
  g->mt[0] = seed;
  for (int i = 1; i < 624; ++i)
    g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
  g->idx = 624;
}

static void mt_twist(ora_mt19937 *g) {
  uint32_t *mt = g->mt;
  for (int i = 0; i < 624; ++i) {
    uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
    uint32_t x = mt[(i + 397) % 624] ^ (y >> 1);
    if (y & 1u) x ^= 0x9908b0dfu;
    mt[i] = x;
  }
  g->idx = 0;
}

uint32_t ora_mt_next(ora_mt19937 *g) {
  if (g->idx >= 624) mt_twist(g);
  uint32_t y = g->mt[g->idx++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

/* uniform_int_distribution<int>(0, n-1): urng range 2^32-1 > urange = n-1  => _S_nd<uint64_t> */
int ora_uniform_index(ora_mt19937 *g, uint32_t n) {
  uint32_t range = n; /* urange + 1 */
  uint64_t product = (uint64_t)ora_mt_next(g) * (uint64_t)range;
  uint32_t low = (uint32_t)product;
  if (low < range) {
    uint32_t threshold = (uint32_t)(-range) % range;
    while (low < threshold) {
      product = (uint64_t)ora_mt_next(g) * (uint64_t)range;
      low = (uint32_t)product;
    }
  }
  return (int)(product >> 32);
}
