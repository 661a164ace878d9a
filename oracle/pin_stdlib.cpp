// oracle/pin_stdlib.cpp -- TEST INFRASTRUCTURE ONLY.
// Pins oracle/rng_ref.c against the real std::mt19937 + std::uniform_int_distribution<int> of the
// libstdc++ in this image, i.e. the exact expression of DeepQNAg::sample_transition
// (agent/deepqn/include/DeepQNAg.hpp:277).  Prints "OK <n_checked>" or the first mismatch.
#include <cstdio>
#include <cstdlib>
#include <random>
#include "ora.h"

int main(int argc, char **argv) {
  long checked = 0;
  // the standard's own check value
  {
    std::mt19937 e;
    ora_mt19937 g; ora_mt_seed(&g, 5489u);
    uint32_t a = 0, b = 0;
    for (int i = 0; i < 10000; ++i) { a = e(); b = ora_mt_next(&g); }
    if (a != 4123659995u || b != a) { std::printf("MISMATCH mt19937 10000th %u %u\n", a, b); return 1; }
  }
  const uint32_t seeds[] = {0u, 1u, 1000u, 3000u, 4294967295u};
  const uint32_t sizes[] = {1u, 2u, 250u, 251u, 1000u, 10000u, 65536u, 1000000u, 2147483647u, 3000000000u};
  for (uint32_t seed : seeds)
    for (uint32_t n : sizes) {
      if (n > 2147483647u) continue;  // int distribution
      std::mt19937 e(seed);
      ora_mt19937 g; ora_mt_seed(&g, seed);
      for (int i = 0; i < 20000; ++i) {
        int r = std::uniform_int_distribution<int>(0, (int)(n - 1))(e);
        int q = ora_uniform_index(&g, n);
        if (r != q) { std::printf("MISMATCH seed %u n %u draw %d: std %d ora %d\n", seed, n, i, r, q); return 1; }
        ++checked;
      }
    }
  // a range that rejects often (n just above 2^31 is not an int range; use n = 2^31-1: p(reject) ~ 0.5)
  std::printf("OK %ld\n", checked);
  return 0;
}
