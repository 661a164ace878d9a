/*
 * ddrl_agent.hpp -- the G = 1 C++ flavour of the drop-in boundary: one agent behind the reference's own
 * agent interface, header-only over the C ABI of ddrl_pop.h (link libddrl_pop.so).
 *
 * Signatures follow /root/reference/common/include/arch/AAgent.hpp:25-27 (the (nb_motors, nb_sensors) constructor),
 * :39-64 (run / runf), :71-82 (start_episode / start_instance / end_instance / end_episode), :93-123 (save / load /
 * save_run / load_previous_run) and ARLAgent.hpp:37-74,109-137 (runf's decision_each / reward pooling and
 * start_episode, both of which run INSIDE the library: ddrl_agents_run / ddrl_agents_start_episode).  What differs:
 *   - unique_invoke takes the path of the .ini (what the reference finds under --config in its variables_map,
 *     Simulator.hpp:266,283-285) instead of a boost::property_tree / program_options pair: this header has no boost
 *     dependency; a maintainer's overload for the boost types is the three lines shown in INTEGRATION.md 1.1;
 *   - errors throw std::runtime_error(ddrl_last_error()) where the reference logs and exit(1)s (MLP.hpp:99-100).
 * Returned action vectors are owned by the agent and valid until the next call, like ARLAgent's returned_ac (:64-73,220).
 */
#ifndef DDRL_AGENT_HPP
#define DDRL_AGENT_HPP
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "ddrl_pop.h"

namespace ddrl {

enum class Algo { DDPG, NFAC, TD3 };

template <Algo ALGO>
class AgentG1 {
 public:
  AgentG1(unsigned int _nb_motors, unsigned int _nb_sensors) : nb_motors(_nb_motors), nb_sensors(_nb_sensors) {}   // AAgent.hpp:25-27
  AgentG1(const AgentG1 &) = delete;
  AgentG1 &operator=(const AgentG1 &) = delete;
  virtual ~AgentG1() { if (h) ddrl_agents_destroy(h); }

  // ARLAgent::unique_invoke (ARLAgent.hpp:139-179) + the agent's _unique_invoke (DeepQNAg.hpp:167-235,
  // OfflineCaclaAg.hpp:122-209, TD3Ag.hpp:165-231): reads [agent] (+ [devnn]) of the ini, builds the nets on `device`
  void unique_invoke(const std::string &config_ini, uint64_t seed = 1, int device = -1, int max_trajectory = 0) {
    if (h) { ddrl_agents_destroy(h); h = nullptr; }
    int rc;
    if (ALGO == Algo::DDPG) rc = ddrl_agents_create_ddpg(config_ini.c_str(), 1, (int)nb_sensors, (int)nb_motors, device, seed, &h);
    else if (ALGO == Algo::TD3) rc = ddrl_agents_create_td3(config_ini.c_str(), 1, (int)nb_sensors, (int)nb_motors, device, seed, &h);
    else rc = ddrl_agents_create_nfac(config_ini.c_str(), 1, (int)nb_sensors, (int)nb_motors, max_trajectory, device, seed, &h);
    check(rc);
    returned_ac.assign(nb_motors, 0.);
  }

  virtual void start_instance(bool) {}                                          // AAgent.hpp:79
  virtual void end_instance(bool) {}                                            // :81 (counted inside ddrl_agents_end_episode)

  // ARLAgent::start_episode (ARLAgent.hpp:109-137)
  virtual void start_episode(const std::vector<double> &perceptions, bool learning) {
    need();
    to_float(perceptions, st);
    const uint8_t l = learning;
    check(ddrl_agents_start_episode(h, st.data(), &l, nullptr));
  }

  // AAgent::run (:39-48) and ARLAgent::runf (ARLAgent.hpp:37-74)
  virtual const std::vector<double> &run(double reward, const std::vector<double> &perceptions, bool learning, bool absorbing_state) {
    return runf(reward, perceptions, learning, absorbing_state, false);
  }
  virtual const std::vector<double> &runf(double reward, const std::vector<double> &perceptions, bool learning, bool absorbing_state,
                                          bool finished) {
    need();
    to_float(perceptions, st);
    const float r = (float)reward;
    const uint8_t l = learning, g = absorbing_state, f = finished;
    ac.assign(nb_motors, 0.f);
    check(ddrl_agents_run(h, &r, st.data(), &l, &g, &f, nullptr, ac.data()));
    for (unsigned i = 0; i < nb_motors; ++i) returned_ac[i] = ac[i];
    return returned_ac;
  }

  // end_episode(learning) followed by end_instance(learning), as the reference's C flavour calls them (ddpg.cpp:52-55)
  virtual void end_episode(bool learning) {
    need();
    const uint8_t l = learning;
    check(ddrl_agents_end_episode(h, &l, nullptr));
  }

  // AAgent::save_run / load_previous_run (:109-117; DeepQNAg.hpp:504-537): one file, complete enough to resume bit for bit
  virtual void save_run(const std::string &path = "continue.agent.data") { need(); check(ddrl_agents_save_run(h, path.c_str())); }
  virtual void load_previous_run(const std::string &path = "continue.agent.data") { need(); check(ddrl_agents_load_previous_run(h, path.c_str())); }

  // ARLAgent's bookkeeping (ARLAgent.hpp:181-218)
  double sum_weighted_reward() { double w = 0, s = 0; int32_t e = 0; need(); check(ddrl_agents_stats(h, &w, &s, &e)); return w; }
  double sum_reward() { double w = 0, s = 0; int32_t e = 0; need(); check(ddrl_agents_stats(h, &w, &s, &e)); return s; }
  int episode() { double w = 0, s = 0; int32_t e = 0; need(); check(ddrl_agents_stats(h, &w, &s, &e)); return e; }

  // MLP::copyWeightsTo / copyWeightsFrom / number_of_parameters of one of the agent's nets (MLP.hpp:610-676)
  int number_of_parameters(int net) { need(); return ddrl_pop_num_params(ddrl_agents_population(h), net); }
  std::vector<double> copyWeightsTo(int net) {
    std::vector<double> w(number_of_parameters(net));
    check(ddrl_pop_get_weights(ddrl_agents_population(h), net, w.data()));
    return w;
  }
  void copyWeightsFrom(int net, const std::vector<double> &w) {
    if ((int)w.size() != number_of_parameters(net)) throw std::runtime_error("copyWeightsFrom: wrong parameter count");
    check(ddrl_pop_set_weights(ddrl_agents_population(h), net, w.data()));
  }

  ddrl_agents_t *handle() { return h; }
  const unsigned int nb_motors, nb_sensors;

 protected:
  void need() const { if (!h) throw std::runtime_error("unique_invoke() has not been called"); }
  static void check(int rc) { if (rc != DDRL_OK) throw std::runtime_error(ddrl_last_error()); }
  void to_float(const std::vector<double> &v, std::vector<float> &out) const {
    if (v.size() != nb_sensors) throw std::runtime_error("perceptions: wrong length");
    out.assign(v.begin(), v.end());
  }
  ddrl_agents_t *h = nullptr;
  std::vector<float> st, ac;
  std::vector<double> returned_ac;
};

typedef AgentG1<Algo::DDPG> DeepQNAg;          // agent/deepqn/include/DeepQNAg.hpp:92-98
typedef AgentG1<Algo::TD3> TD3Ag;              // agent/deepqn/include/TD3Ag.hpp:90-98
typedef AgentG1<Algo::NFAC> OfflineCaclaAg;    // agent/cacla/include/OfflineCaclaAg.hpp:56-62

}  // namespace ddrl
#endif
