// tests/cpp_adaptor_main.cpp -- compiled by tests/test_cpp_adaptor.py against include/ddrl_agent.hpp + libddrl_pop.so.
// argv[1] = ini path, argv[2] = "run" (needs a GPU) or "nogpu" (expects the library's "no HIP device" error).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

#include "ddrl_agent.hpp"

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  ddrl::DeepQNAg agent(1, 4);                                    // (nb_motors, nb_sensors): cartpole dims
  try {
    agent.runf(0., std::vector<double>(4, 0.), true, false, false);
    return 3;                                                    // must refuse before unique_invoke
  } catch (const std::runtime_error &) {}
  try {
    agent.unique_invoke(argv[1], 5);
  } catch (const std::runtime_error &e) {
    if (!strcmp(argv[2], "nogpu") && strstr(e.what(), "no HIP device")) { printf("nogpu ok\n"); return 0; }
    fprintf(stderr, "unique_invoke: %s\n", e.what());
    return 4;
  }
  if (!strcmp(argv[2], "nogpu")) return 5;
  std::mt19937 eng(3);
  std::uniform_real_distribution<double> u(-1., 1.);
  const int na = agent.number_of_parameters(DDRL_NET_ACTOR);
  const std::vector<double> w0 = agent.copyWeightsTo(DDRL_NET_ACTOR);
  for (int ep = 0; ep < 3; ++ep) {
    std::vector<double> s(4);
    for (auto &x : s) x = u(eng);
    agent.start_instance(true);
    agent.start_episode(s, true);
    for (int step = 0; step < 150; ++step) {
      for (auto &x : s) x = u(eng);
      const std::vector<double> &a = agent.runf(u(eng), s, true, false, step == 149);
      if (a.size() != 1 || !std::isfinite(a[0]) || std::fabs(a[0]) > 1.0 + 1e-6) return 6;
    }
    agent.end_episode(true);
    agent.end_instance(true);
  }
  if (agent.episode() != 3) return 7;
  const std::vector<double> w1 = agent.copyWeightsTo(DDRL_NET_ACTOR);
  bool moved = false;
  for (int i = 0; i < na; ++i) moved = moved || w1[i] != w0[i];
  if (!moved) return 8;                                          // 450 stored transitions > 250: updates ran
  agent.save_run("/tmp/ddrl_cpp_adaptor.run");
  ddrl::DeepQNAg twin(1, 4);
  twin.unique_invoke(argv[1], 99);
  twin.load_previous_run("/tmp/ddrl_cpp_adaptor.run");
  if (twin.copyWeightsTo(DDRL_NET_ACTOR) != w1) return 9;
  printf("run ok: %d actor parameters, return %.4f\n", na, agent.sum_weighted_reward());
  return 0;
}
