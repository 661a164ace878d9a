// agents.hip -- the agent-level front end over a DDPG / NFAC / CMA-ES population: what ARLAgent::runf /
// start_episode (common/include/arch/ARLAgent.hpp:37-137) and DeepQNAg::_run / _start_episode /
// end_episode / end_instance (agent/deepqn/include/DeepQNAg.hpp:122-165, 237-317) do around the nets,
// for G agents at once, plus the G = 1 C flavour with the reference's own names and argument types
// (DDPGAg_*, agent/deepqn/src/pybinding/ddpg.cpp:9-88; caller gym/agent.py:112-157).
//
// Host code only: decision_each / reward max-pooling / discounted-return bookkeeping and the
// exploration noise (SURVEY.md 8a row a32 keeps it on the host).  Every net evaluation, replay
// insertion and update goes through the ddrl_pop_* entry points, i.e. runs on the GPU.
#include "pop_host.h"
#include "pool.h"

#include <memory>
#include <sstream>

struct AgentState {
  std::vector<double> inter_rewards;
  int time_for_ac = 1;
  double global_pow_gamma = 1., sum_weighted_reward = 0., sum_reward = 0., last_sum_weighted_reward = 0.;
  double last_receive_reward = 0.;
  bool last_episode_learning = false, have_last = false;
  int episode = 0, step = 0;
  int td3_step = 0;                          // TD3Ag::current_step_counter (TD3Ag.hpp:157, 244)
  std::vector<double> ou_x;                  // OrnsteinUhlenbeckNoise::x_t
  std::mt19937 engine;
};

// DODevMLP's developmental schedule (common/include/nn/DODevMLP.hpp:76-214 exploit, :403-574
// going_to_develop / develop / developIM / inform, :764-775 neutral_action): which controlled state /
// action dimensions are open.  Host logic; the gates themselves act on the device (ddrl_pop_set_gates).
struct DevCfg {
  bool on = false, disable_st = false;
  std::vector<long> st_control, ac_control, devpoints;
  bool im = false, reset_learning_algo = false, informed = false;
  int heuristic = 1;                         // 0: gate weights are parameters, 1: switches at heuristic_devpoints, 2: linear ramps of Bernoulli gates
  bool st_bern = false, ac_bern = false;     // per-sample Bernoulli(w) gates (the layer's probabilist = 0)
  bool st_scale = false, ac_scale = false;
  std::vector<double> linearcoef;            // heuristic 2 (DODevMLP.hpp:123-130, 444-467)
  int im_smooth = 0, im_window = 0;
  std::vector<double> pd;                    // pd_controller_factor [nb_motors]
};
struct DevState {
  std::vector<double> st_w, ac_w;            // the gate blobs, zero-initialised (DODevMLP.hpp:358-374)
  std::vector<double> all_scores;
  int last_episode_changed = 0;
  unsigned im_index = 0;
};

struct ddrl_agents {
  ddrl_pop *pop = nullptr;
  DevCfg dev;
  std::vector<DevState> devst;
  int G = 0, S = 0, A = 0;
  double gamma = 0.99, noise = 0., noise2 = 0., noise3 = 0., reward_scale = 1.;
  int gaussian_policy = 0, decision_each = 1, force_more_update = 0;
  int algo = 0;                              // 0 DDPG (DeepQNAg), 1 NFAC (OfflineCaclaAg), 2 CMA-ES (CMAESAg), 3 TD3 (TD3Ag)
  int policy_freq = 1;                       // TD3Ag.hpp:186
  int update_each_episode = 1;
  // CMAESAg: one agent slot per individual of the generation
  ddrl_cmaes_t *es = nullptr;
  float *d_best = nullptr;                   // xbestever as fp32 (loadPolicyParameters(getBestSolution()))
  std::vector<double> funvals;               // arFunvals
  std::vector<uint8_t> evaluated;
  int n_evaluated = 0, generations = 0, error_count = 0;
  bool done_once = false, check_feasible = true, racing = false;
  std::vector<AgentState> ag;
  std::vector<float> last_state, last_action, last_pure, returned_ac, pure, rew;   // [G x S], [G x A] x 4, [G]
  std::vector<uint8_t> insert_mask, goal;
  std::vector<double> ret_double;            // DDPGAg_run's return buffer (valid until the next call)
  std::unique_ptr<DdrlPool> pool;            // per-agent host work of a step (exploration noise) for large populations
};

// N(mu, sigma) truncated to [a, b].  The reference calls Chopin's table-driven rtnorm
// (common/src/RTNorm.cpp) on its clock-seeded engine; its stream cannot be pinned, so this is the
// same distribution by rejection sampling (Robert 1995 for the tails), not the same stream.
static double trunc_normal(std::mt19937 &eng, double a, double b, double mu, double sigma) {
  std::uniform_real_distribution<double> U(0., 1.);
  if (!(sigma > 0.)) return std::min(std::max(mu, a), b);
  const double al = (a - mu) / sigma, be = (b - mu) / sigma;
  auto tail = [&](double lo, double hi) {      // lo > 0: exponential proposal
    const double lam = 0.5 * (lo + std::sqrt(lo * lo + 4.));
    for (;;) {
      const double z = lo - std::log(1. - U(eng)) / lam;
      if (z <= hi && U(eng) <= std::exp(-0.5 * (z - lam) * (z - lam))) return z;
    }
  };
  double z;
  if (al > 2.) z = tail(al, be);
  else if (be < -2.) z = -tail(-be, -al);
  else if (be - al > 1.5) {
    std::normal_distribution<double> Nn(0., 1.);
    do z = Nn(eng); while (z < al || z > be);
  } else {                                     // narrow central interval: uniform proposal
    const double m = (al <= 0. && be >= 0.) ? 0. : std::min(std::fabs(al), std::fabs(be));
    for (;;) {
      z = al + (be - al) * U(eng);
      if (U(eng) <= std::exp(-0.5 * (z * z - m * m))) break;
    }
  }
  return mu + sigma * z;
}

// the learning branch of DeepQNAg::_run's action perturbation (DeepQNAg.hpp:134-145)
static void explore(ddrl_agents *h, int g, float *act) {
  AgentState &s = h->ag[g];
  const int A = h->A;
  std::uniform_real_distribution<double> U01(0., 1.);
  // OfflineCaclaAg adds two schedules of the truncated Gaussian (OfflineCaclaAg.hpp:95-102)
  int policy = h->gaussian_policy;
  double sigma = h->noise;
  if (h->algo == 1 && policy == 3) policy = U01(s.engine) < h->noise2 ? 1 : -1;
  if (h->algo == 1 && policy == 4) { sigma = h->noise * std::pow(h->noise2, h->noise3 - (double)s.step); policy = 1; }
  if (policy == -1) return;
  if (policy == 1) {                           // multidimentionnalTruncatedGaussian (MetropolisHasting.hpp:31-46)
    for (int i = 0; i < A; ++i) {
      double mu = act[i];
      if (mu > 1.0f + 3.f * sigma) mu = 1.0f + 3.f * sigma;
      else if (mu < -1.0f - 3.f * sigma) mu = -1.0f - 3.f * sigma;
      act[i] = (float)trunc_normal(s.engine, -1.0, 1.0, mu, sigma);
    }
  } else if (policy == 2) {                    // OrnsteinUhlenbeckNoise::step (OrnsteinUhlenbeckNoise.hpp:22-37)
    std::normal_distribution<double> dist(0., 1.);
    for (int i = 0; i < A; ++i) {
      const double normal = dist(s.engine);
      const double x_tplus = s.ou_x[i] - h->noise2 * h->noise3 * s.ou_x[i] + h->noise * std::sqrt(h->noise3) * normal;
      double m = act[i] + x_tplus;
      if (m < -1.) m = -1.; else if (m > 1.) m = 1.;
      act[i] = (float)m;
      s.ou_x[i] = x_tplus;
    }
  } else {                                     // e-greedy (:141-144)
    std::uniform_real_distribution<double> U(0., 1.);
    if (U(s.engine) < h->noise)
      for (int i = 0; i < A; ++i) act[i] = (float)(-1.f + U(s.engine) * 2.);
  }
}

extern "C" int ddrl_agents_destroy(ddrl_agents_t *h) {
  if (!h) return DDRL_OK;
  if (h->d_best) (void)hipFree(h->d_best);
  if (h->es) ddrl_cmaes_destroy(h->es);
  if (h->pop) ddrl_pop_destroy(h->pop);
  delete h;
  return DDRL_OK;
}

// bib::to_array<double>: ':' separated
static std::vector<double> colon_doubles(const std::string &raw) {
  std::vector<double> out;
  for (size_t pos = 0; pos <= raw.size();) {
    size_t e = raw.find(':', pos);
    if (e == std::string::npos) e = raw.size();
    if (e > pos) out.push_back(std::stod(raw.substr(pos, e - pos)));
    pos = e + 1;
  }
  return out;
}

// DODevMLP::exploit (DODevMLP.hpp:76-214): the [devnn] keys, with the reference's consistency checks
// as errors instead of exit(1).  Active when the ini has devnn.st_control.
static int read_devnn(ddrl_agents *h, DdrlIni &ini, const char *ini_path, uint64_t seed) {
  DevCfg &d = h->dev;
  if (!ini.has("devnn.st_control")) return DDRL_OK;
  try {
    const bool st_scale = ini.get_bool("devnn.st_scale"), ac_scale = ini.get_bool("devnn.ac_scale");
    const long st_prob = ini.get_long("devnn.st_probabilistic"), ac_prob = ini.get_long("devnn.ac_probabilistic");
    d.disable_st = ini.get("devnn.st_control") == "None";                       // :132-136
    if (!d.disable_st) d.st_control = ini.get_list("devnn.st_control");
    if (ini.get("devnn.ac_control") != "None") d.ac_control = ini.get_list("devnn.ac_control");   // "None" only removes the STATE layer (:132); here it lists nothing (CaffeUtest.cpp:702-708: every action is gated)
    const long heuristic = ini.has("devnn.heuristic") ? ini.get_long("devnn.heuristic") : 0;
    // heuristic 0 (the reference's default, :84): the gate weights are ordinary parameters of the nets.  Served: the layer's
    // forward, the parameter counts and the flat order (ddrl_pop_dev_*); the fork's backward for the gate weights themselves is not
    // in the tree, so they stay where the caller puts them (they start at 1: the fork's filler is unknown) -- parity unpinned
    if (heuristic < 0 || heuristic > 2) return ddrl_fail(DDRL_EINVAL, "devnn.heuristic = %ld unknown", heuristic);
    if (heuristic == 0 && (st_prob < 0 || st_prob > 1 || ac_prob < 0 || ac_prob > 1))
      return ddrl_fail(DDRL_EINVAL, "devnn.*_probabilistic = 2 / 3 are not documented by the reference (smoke-tested only, CaffeUtest.cpp:1164, 1213)");
    if (heuristic == 0 && ini.has("devnn.reset_learning_algo") && ini.get_bool("devnn.reset_learning_algo"))
      return ddrl_fail(DDRL_EINVAL, "when sould I reset, I'm learning alone all my parameters");                        // :202-205
    if (heuristic == 1 && (st_prob != 1 || ac_prob != 1)) return ddrl_fail(DDRL_EINVAL, "heuristic is 1 so probabilistic should be 1");   // :90-93
    if (heuristic == 2 && (st_prob != 0 || ac_prob != 0)) return ddrl_fail(DDRL_EINVAL, "heuristic is 2 so probabilistic should be 0");   // :94-97
    if ((ac_scale && ac_prob != 0) || (st_scale && st_prob != 0))
      return ddrl_fail(DDRL_EINVAL, "if not probabilistic then why scaling? of how much?");                              // :192-195
    d.heuristic = (int)heuristic; d.st_scale = st_scale; d.ac_scale = ac_scale;
    d.st_bern = heuristic == 2 || (heuristic == 0 && st_prob == 0);
    d.ac_bern = heuristic == 2 || (heuristic == 0 && ac_prob == 0);
    if (heuristic == 0) {                       // an empty list gates every dimension (CaffeUtest.cpp:629-636)
      if (!d.disable_st && d.st_control.empty()) for (int i = 0; i < h->S; ++i) d.st_control.push_back(i);
      if (d.ac_control.empty()) for (int i = 0; i < h->A; ++i) d.ac_control.push_back(i);
    }
    if (ini.has("devnn.intrasec_motivation")) d.im = ini.get_bool("devnn.intrasec_motivation");
    if (d.im && heuristic != 1)
      return ddrl_fail(DDRL_EINVAL, "intrasec motiv works only with determinist activation and heuristic = 1");          // :104-107
    if (d.im) { d.im_smooth = (int)ini.get_long("devnn.im_smooth"); d.im_window = (int)ini.get_long("devnn.im_window"); }
    if (heuristic == 1) {
      d.devpoints = ini.get_list("devnn.heuristic_devpoints");
      if (d.devpoints.size() != d.st_control.size() + d.ac_control.size())                 // :119-120
        return ddrl_fail(DDRL_EINVAL, "heuristic_devpoints has %zu entries for %zu + %zu controlled dimensions", d.devpoints.size(),
                         d.st_control.size(), d.ac_control.size());
    } else if (heuristic == 2) {
      d.linearcoef = colon_doubles(ini.get("devnn.heuristic_linearcoef"));
      if (d.linearcoef.size() < d.st_control.size() + d.ac_control.size())                 // :126-127
        return ddrl_fail(DDRL_EINVAL, "heuristic_linearcoef has %zu entries for %zu + %zu controlled dimensions", d.linearcoef.size(),
                         d.st_control.size(), d.ac_control.size());
    }
    if (ini.has("devnn.compute_diff_backward") && ini.get_bool("devnn.compute_diff_backward"))
      return ddrl_fail(DDRL_EINVAL, "why do you compute diff if you don't use local solver?");                         // :207-210
    if (ini.has("devnn.reset_learning_algo")) d.reset_learning_algo = ini.get_bool("devnn.reset_learning_algo");
    if (ini.has("devnn.informed_sensorimotor_space")) d.informed = ini.get_bool("devnn.informed_sensorimotor_space");
    if (d.informed) {
      d.pd = colon_doubles(ini.get("devnn.pd_controller_factor"));
      if ((int)d.pd.size() != h->A) return ddrl_fail(DDRL_EINVAL, "wrong pd_controller_factor %zu", d.pd.size());       // :155-158
    }
  } catch (const std::exception &e) {
    return ddrl_fail(DDRL_EIO, "%s: %s", ini_path, e.what());
  }
  if (d.im && (d.im_smooth < 1 || d.im_window < 0)) return ddrl_fail(DDRL_EINVAL, "devnn.im_smooth < 1 or im_window < 0");
  if (d.st_control.empty() && !d.disable_st) return ddrl_fail(DDRL_EINVAL, "devnn.st_control: list the gated dimensions or None");   // (heuristic 0 filled an empty list above)
  for (long c : d.st_control) if (c < 0 || c >= h->S) return ddrl_fail(DDRL_EINVAL, "devnn.st_control %ld outside 0..%d", c, h->S - 1);
  for (long c : d.ac_control) if (c < 0 || c >= h->A) return ddrl_fail(DDRL_EINVAL, "devnn.ac_control %ld outside 0..%d", c, h->A - 1);
  if (d.heuristic == 2 && d.informed) return ddrl_fail(DDRL_EINVAL, "informed_sensorimotor_space is not implemented with proba action");   // :161-169
  if (d.heuristic == 2 && d.reset_learning_algo) return ddrl_fail(DDRL_EINVAL, "when sould I reset ? I'm progressive");                    // :197-200
  if (h->algo == 1 && d.reset_learning_algo) return ddrl_fail(DDRL_EINVAL, "NFAC cannot reset anything with DODevMLP");   // OfflineCaclaAg.hpp:199-203
  if (h->algo == 2) return ddrl_fail(DDRL_EINVAL, "the developmental schedule is not supported for CMA-ES agents");
  if (d.st_bern || d.ac_bern) {
    int rc = ddrl_pop_set_gate_mode(h->pop, d.st_bern ? 1 : 0, d.st_scale ? 1 : 0, d.ac_bern ? 1 : 0, d.ac_scale ? 1 : 0,
                                    (uint32_t)(seed * 2246822519ull + 374761393u));
    if (rc) return rc;
  }
  {
    std::vector<int> st(d.st_control.begin(), d.st_control.end()), ac(d.ac_control.begin(), d.ac_control.end());
    int rc = ddrl_pop_set_gate_layout(h->pop, st.data(), d.disable_st ? -1 : (int)st.size(), ac.data(), (int)ac.size(), d.heuristic == 0);
    if (rc) return rc;
  }
  d.on = true;
  h->devst.resize(h->G);
  const double w0 = d.heuristic == 0 ? 1. : 0.;           // :358-374 zeroes the blobs under heuristic != 0 only
  for (auto &st : h->devst) { st.st_w.assign(d.st_control.size(), w0); st.ac_w.assign(d.ac_control.size(), w0); }
  return DDRL_OK;
}

// the gate weights into the population.  Deterministic Developmental layer (probabilist = 1, heuristic 1): a controlled
// dimension whose weight is below 0.5 is forced to 0, every other one passes.  Probabilistic layer
// (heuristic 2): the weight itself is the pass probability (capped at 1: episode x coefficient grows without bound in
// the reference, DODevMLP.hpp:453, and what its layer does with w > 1 is not pinned by any test in the tree).
static int push_gates(ddrl_agents *h) {
  const DevCfg &d = h->dev;
  std::vector<double> sg((size_t)h->G * h->S, 1.), ag((size_t)h->G * h->A, 1.);
  auto gate = [](bool prob, double w) { return prob ? std::min(1., std::max(0., w)) : (w < 0.5 ? 0. : 1.); };
  for (int g = 0; g < h->G; ++g) {
    for (size_t i = 0; i < d.st_control.size(); ++i) sg[(size_t)g * h->S + d.st_control[i]] = gate(d.st_bern, h->devst[g].st_w[i]);
    for (size_t i = 0; i < d.ac_control.size(); ++i) ag[(size_t)g * h->A + d.ac_control[i]] = gate(d.ac_bern, h->devst[g].ac_w[i]);
  }
  return ddrl_pop_set_gates(h->pop, sg.data(), ag.data());
}

// DODevMLP::update_DL_IM (:539-562) / still_something_to_develop (:517-537)
static void update_dl_im(const DevCfg &d, DevState &st) {
  for (size_t i = 0; i < d.devpoints.size(); ++i)
    if ((unsigned long)d.devpoints[i] <= st.im_index) {
      double &w = i < d.st_control.size() ? st.st_w[i] : st.ac_w[i - d.st_control.size()];
      if (w < 0.5) w = 1.;
    }
  st.im_index++;
}
static bool still_something(const DevCfg &d, const DevState &st) {
  for (size_t i = 0; i < d.devpoints.size(); ++i)
    if ((unsigned long)d.devpoints[i] <= st.im_index + 1) {
      const double w = i < d.st_control.size() ? st.st_w[i] : st.ac_w[i - d.st_control.size()];
      if (w == 0.) return true;
    }
  return false;
}

// DODevMLP::inform (:566-574) for one agent: returns changed; *setup = ewc_setup() was called
static bool dev_inform(const DevCfg &d, DevState &st, int episode, double score, bool *setup) {
  *setup = false;
  if (d.heuristic == 0) return false;                                  // nothing is scheduled: the gate weights are parameters
  if (d.heuristic == 2) {                                              // develop (:444-467): every episode, ewc_setup() first
    for (size_t i = 0; i < d.st_control.size(); ++i) st.st_w[i] = d.linearcoef[i] >= 1. ? 1. : (double)episode * d.linearcoef[i];
    for (size_t i = 0; i < d.ac_control.size(); ++i) {
      const double c = d.linearcoef[i + d.st_control.size()];
      st.ac_w[i] = c >= 1. ? 1. : (double)episode * c;
    }
    *setup = true;
    return true;
  }
  if (!d.im) {                                                         // develop (:420-470), heuristic 1
    bool changed = false;
    for (size_t i = 0; i < d.devpoints.size(); ++i)
      if ((long)episode == d.devpoints[i]) {
        (i < d.st_control.size() ? st.st_w[i] : st.ac_w[i - d.st_control.size()]) = 1.;
        changed = true;
      }
    *setup = changed;                                                  // going_to_develop => ewc_setup() first (:422-424)
    return changed;
  }
  if (episode == 0) { update_dl_im(d, st); return true; }               // developIM (:472-515)
  const int ep = episode - st.last_episode_changed - 1;
  st.all_scores.push_back(score);
  double new_e = 0., new_ew = 0.;
  for (int i = 0; i < d.im_smooth; ++i) {
    new_e += ep - i < 0 ? -1e8 : st.all_scores[ep - i];                 // LOWER_REWARD (:7)
    new_ew += ep - i - d.im_window < 0 ? -1e8 : st.all_scores[ep - i - d.im_window];
  }
  new_e *= 1. / (double)d.im_smooth;
  new_ew *= 1. / (double)d.im_smooth;
  const bool changed = new_e - new_ew <= 0. && ep >= d.im_smooth + d.im_window && still_something(d, st);
  if (changed) {
    *setup = true;
    st.last_episode_changed += (int)st.all_scores.size();
    update_dl_im(d, st);
    st.all_scores.clear();
  }
  return changed;
}

// the learning branch of _start_episode (DeepQNAg.hpp:250-272, OfflineCaclaAg.hpp:225-234)
static int dev_start_episode(ddrl_agents *h, const uint8_t *learning, const uint8_t *mask) {
  const int G = h->G;
  std::vector<uint8_t> changed(G, 0), setup(G, 0);
  bool any_c = false, any_s = false;
  for (int g = 0; g < G; ++g) {
    if ((mask && !mask[g]) || !learning[g]) continue;
    bool s = false;
    if (dev_inform(h->dev, h->devst[g], h->ag[g].episode, h->ag[g].last_sum_weighted_reward, &s)) { changed[g] = 1; any_c = true; }
    if (s) { setup[g] = 1; any_s = true; }
  }
  int rc;
  if (any_s && h->pop->cfg.ewc >= 0. && (rc = ddrl_pop_ewc_setup_net(h->pop, DDRL_NET_ACTOR, setup.data()))) return rc;
  if (!any_c) return DDRL_OK;
  if ((rc = push_gates(h))) return rc;
  if (h->dev.reset_learning_algo && h->algo == 0 && (rc = ddrl_pop_replay_clear(h->pop, changed.data()))) return rc;   // "reset learning catched"
  if (h->pop->cfg.ewc >= 0. && h->pop->cfg.ewc_for_critic == 2 &&                                                      // ewc_force_constraint
      (rc = ddrl_pop_ewc_setup_net(h->pop, DDRL_NET_CRITIC, changed.data()))) return rc;
  return DDRL_OK;
}

// the keys both agents read around their nets: ARLAgent.hpp:142-159 + noise / gaussian_policy
static int read_common(ddrl_agents *h, const char *ini_path, DdrlIni &ini) {
  std::string err;
  if (!ini.load(ini_path, &err)) return ddrl_fail(DDRL_EIO, "%s", err.c_str());
  try {
    h->decision_each = (int)ini.get_long("agent.decision_each");          // ARLAgent.hpp:143
    if (h->algo != 2) {                                                   // CMAESAg explores in parameter space (CMAESAg.hpp:43)
      h->noise = ini.get_double("agent.noise");                           // DeepQNAg.hpp:170-171, OfflineCaclaAg.hpp:126-127
      h->gaussian_policy = (int)ini.get_long("agent.gaussian_policy");
    }
    if (ini.has("agent.reward_scale")) h->reward_scale = ini.get_double("agent.reward_scale");
    if (ini.has("agent.history_size") && ini.get_long("agent.history_size") != 1)
      return ddrl_fail(DDRL_EINVAL, "agent.history_size != 1 is not supported");
    if (h->gaussian_policy == 2 || (h->algo == 1 && h->gaussian_policy == 4)) {   // DeepQNAg.hpp:203-207, OfflineCaclaAg.hpp:147-156
      h->noise2 = ini.get_double("agent.noise2");
      h->noise3 = ini.get_double("agent.noise3");
    } else if (h->algo == 1 && h->gaussian_policy == 3) {
      h->noise2 = ini.get_double("agent.noise2");
    }
    if (h->algo == 1 && ini.has("agent.update_each_episode")) h->update_each_episode = (int)ini.get_long("agent.update_each_episode");
  } catch (const std::exception &e) {
    return ddrl_fail(DDRL_EIO, "%s: %s", ini_path, e.what());
  }
  if (h->decision_each < 1) return ddrl_fail(DDRL_EINVAL, "agent.decision_each < 1");
  if (h->update_each_episode < 1) return ddrl_fail(DDRL_EINVAL, "agent.update_each_episode < 1");
  return DDRL_OK;
}

static void init_state(ddrl_agents *h, int G, int S, int A, uint64_t seed) {
  h->G = G; h->S = S; h->A = A;
  h->ag.resize(G);
  for (int g = 0; g < G; ++g) {
    h->ag[g].engine.seed((uint32_t)(seed * 2654435761ull + 2u * g + 1u));
    h->ag[g].ou_x.assign(A, 0.);
  }
  h->last_state.assign((size_t)G * S, 0.f);
  h->last_action.assign((size_t)G * A, 0.f);
  h->last_pure.assign((size_t)G * A, 0.f);
  h->returned_ac.assign((size_t)G * A, 0.f);
  h->pure.assign((size_t)G * A, 0.f);
  h->rew.assign(G, 0.f);
  h->insert_mask.assign(G, 0);
  h->goal.assign(G, 0);
  h->ret_double.assign(A, 0.);
}

extern "C" int ddrl_agents_create_ddpg(const char *ini_path, int n_agents, int nb_sensors, int nb_motors, int device,
                                       uint64_t seed, ddrl_agents_t **out) {
  if (!ini_path || !out) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  *out = nullptr;
  ddrl_ddpg_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.n_agents = n_agents; cfg.nb_sensors = nb_sensors; cfg.nb_motors = nb_motors; cfg.device = device;
  int rc = ddrl_ddpg_config_from_ini(ini_path, &cfg);
  if (rc) return rc;
  std::unique_ptr<ddrl_agents> h(new ddrl_agents());
  DdrlIni ini;
  if ((rc = read_common(h.get(), ini_path, ini))) return rc;
  h->gamma = cfg.gamma;
  if ((rc = ddrl_ddpg_create(&cfg, seed, &h->pop))) return rc;
  h->force_more_update = cfg.force_more_update;
  init_state(h.get(), cfg.n_agents, cfg.nb_sensors, cfg.nb_motors, seed);
  if ((rc = read_devnn(h.get(), ini, ini_path, seed)) || (h->dev.on && (rc = push_gates(h.get())))) { ddrl_agents_destroy(h.release()); return rc; }
  std::vector<uint32_t> rs(cfg.n_agents);
  for (int g = 0; g < cfg.n_agents; ++g) rs[g] = (uint32_t)(seed * 40503ull + 7919u * g + 17u);
  if ((rc = ddrl_pop_seed_replay(h->pop, rs.data()))) { ddrl_agents_destroy(h.release()); return rc; }
  *out = h.release();
  return DDRL_OK;
}

// TD3Ag (agent/deepqn/include/TD3Ag.hpp:95-98, 165-231): DeepQNAg's front end with one update per
// stored transition, the actor and the targets moving every policy_freq-th step of the episode
extern "C" int ddrl_agents_create_td3(const char *ini_path, int n_agents, int nb_sensors, int nb_motors, int device,
                                      uint64_t seed, ddrl_agents_t **out) {
  if (!ini_path || !out) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  *out = nullptr;
  ddrl_ddpg_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.n_agents = n_agents; cfg.nb_sensors = nb_sensors; cfg.nb_motors = nb_motors; cfg.device = device;
  int rc = ddrl_td3_config_from_ini(ini_path, &cfg);
  if (rc) return rc;
  std::unique_ptr<ddrl_agents> h(new ddrl_agents());
  h->algo = 3;
  DdrlIni ini;
  if ((rc = read_common(h.get(), ini_path, ini))) return rc;
  h->gamma = cfg.gamma;
  h->policy_freq = cfg.policy_freq;
  if ((rc = ddrl_td3_create(&cfg, seed, &h->pop))) return rc;
  init_state(h.get(), cfg.n_agents, cfg.nb_sensors, cfg.nb_motors, seed);
  if ((rc = read_devnn(h.get(), ini, ini_path, seed)) || (h->dev.on && (rc = push_gates(h.get())))) { ddrl_agents_destroy(h.release()); return rc; }
  std::vector<uint32_t> rs(cfg.n_agents);
  for (int g = 0; g < cfg.n_agents; ++g) rs[g] = (uint32_t)(seed * 40503ull + 7919u * g + 17u);
  if ((rc = ddrl_pop_seed_replay(h->pop, rs.data()))) { ddrl_agents_destroy(h.release()); return rc; }
  *out = h.release();
  return DDRL_OK;
}

// OfflineCaclaAg (agent/cacla/include/OfflineCaclaAg.hpp:59-62, 122-209).  max_trajectory <= 0: the
// buffer is sized from environment.max_step_per_instance / decision_each x update_each_episode when the
// ini has it, 1024 x update_each_episode otherwise.
extern "C" int ddrl_agents_create_nfac(const char *ini_path, int n_agents, int nb_sensors, int nb_motors, int max_trajectory,
                                       int device, uint64_t seed, ddrl_agents_t **out) {
  if (!ini_path || !out) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  *out = nullptr;
  std::unique_ptr<ddrl_agents> h(new ddrl_agents());
  h->algo = 1;
  DdrlIni ini;
  int rc = read_common(h.get(), ini_path, ini);
  if (rc) return rc;
  if (max_trajectory <= 0) {
    long steps = 1024;
    try {
      if (ini.has("environment.max_step_per_instance")) steps = (ini.get_long("environment.max_step_per_instance") + h->decision_each - 1) / h->decision_each;
    } catch (const std::exception &e) { return ddrl_fail(DDRL_EIO, "%s: %s", ini_path, e.what()); }
    max_trajectory = (int)std::min<long>(steps * h->update_each_episode, 32768);
  }
  ddrl_nfac_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.n_agents = n_agents; cfg.nb_sensors = nb_sensors; cfg.nb_motors = nb_motors; cfg.device = device;
  cfg.max_trajectory = max_trajectory;
  if ((rc = ddrl_nfac_config_from_ini(ini_path, &cfg))) return rc;
  h->gamma = cfg.gamma;
  if ((rc = ddrl_nfac_create(&cfg, seed, &h->pop))) return rc;
  init_state(h.get(), cfg.n_agents, cfg.nb_sensors, cfg.nb_motors, seed);
  if ((rc = read_devnn(h.get(), ini, ini_path, seed)) || (h->dev.on && (rc = push_gates(h.get())))) { ddrl_agents_destroy(h.release()); return rc; }
  *out = h.release();
  return DDRL_OK;
}

// CMAESAg::new_population (CMAESAg.hpp:144-172)
static int new_population(ddrl_agents *h) {
  char msg[256];
  int n = ddrl_cmaes_test_for_termination(h->es, msg, (int)sizeof(msg));
  if (n < 0) return n;
  if (n > 0) {
    fprintf(stderr, "mismatch %s\n", msg);           // LOG_INFO("mismatch " << terminate)
    h->error_count++;
    if (h->error_count > 20 && h->racing)
      return ddrl_fail(DDRL_ESTATE, "CMA-ES met a stopping criterion more than 20 times with agent.racing set (the reference exits here)");
  }
  int rc = ddrl_cmaes_sample(h->es);
  if (rc) return rc;
  if (h->check_feasible) {
    int resampled = 0;
    if ((rc = ddrl_cmaes_make_feasible(h->es, 500., &resampled))) return rc;   // is_feasible: |x_i| < 500 (:135-142)
    if (resampled) fprintf(stderr, "non feasible solution produced\n");
  }
  std::fill(h->evaluated.begin(), h->evaluated.end(), 0);
  h->n_evaluated = 0;
  return DDRL_OK;
}

// CMAESAg (agent/cmaes/include/CMAESAg.hpp:50-133): the policy net from the [agent] keys, c-cmaes
// started at its initial weights with deviation agent.initial_deviation.  One agent slot per
// individual: G = lambda policies are evaluated side by side (the reference evaluates them one
// episode after the other; a mask with one bit set reproduces that order).
extern "C" int ddrl_agents_create_cmaes(const char *ini_path, int nb_sensors, int nb_motors, int device, uint64_t seed,
                                        ddrl_agents_t **out) {
  if (!ini_path || !out) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  *out = nullptr;
  std::unique_ptr<ddrl_agents> h(new ddrl_agents());
  h->algo = 2;
  DdrlIni ini;
  int rc = read_common(h.get(), ini_path, ini);
  if (rc) return rc;
  std::vector<long> hid;
  long population = 0, hidden_type = 1, out_type = 0;
  double deviation = 0.;
  try {
    h->gamma = ini.get_double("agent.gamma");                                    // ARLAgent.hpp:141
    hid = ini.get_list("agent.hidden_unit_a");                                   // CMAESAg.hpp:51-56
    hidden_type = ini.get_long("agent.actor_hidden_layer_type");
    out_type = ini.get_long("agent.actor_output_layer_type");
    if (ini.get_long("agent.batch_norm") != 0) return ddrl_fail(DDRL_EINVAL, "agent.batch_norm != 0 is not supported");
    population = ini.get_long("agent.population");
    deviation = ini.get_double("agent.initial_deviation");
    if (ini.has("agent.check_feasible")) h->check_feasible = ini.get_bool("agent.check_feasible");   // :64-77
    if (ini.has("agent.racing")) h->racing = ini.get_bool("agent.racing");
    if (ini.has("devnn.ewc") && ini.get_double("devnn.ewc") >= 0.)
      return ddrl_fail(DDRL_EINVAL, "CMA-ES doesn't rely on gradient (devnn.ewc must be < 0, CMAESAg.hpp:124-131)");
  } catch (const std::exception &e) {
    return ddrl_fail(DDRL_EIO, "%s: %s", ini_path, e.what());
  }
  if (hid.empty() || hid.size() > DDRL_MAX_HIDDEN) return ddrl_fail(DDRL_EINVAL, "hidden_unit_a: 1..%d layers supported", DDRL_MAX_HIDDEN);
  if (!(deviation > 0.)) return ddrl_fail(DDRL_EINVAL, "agent.initial_deviation <= 0");
  int hidden[DDRL_MAX_HIDDEN];
  long N = 0, prev = nb_sensors;
  for (size_t i = 0; i < hid.size(); ++i) { hidden[i] = (int)hid[i]; N += (prev + 1) * hid[i]; prev = hid[i]; }
  N += (prev + 1) * nb_motors;
  // cmaes_init: lambda < 2 => 4 + (int)(3 log N) (cmaes.c readpara_SupplementDefaults; CMAESAg.hpp:108-109)
  const int lambda = population >= 2 ? (int)population : 4 + (int)(3. * std::log((double)N));
  if ((rc = ddrl_policy_create(lambda, nb_sensors, nb_motors, hidden, (int)hid.size(), (int)hidden_type, (int)out_type, device,
                               seed, &h->pop))) return rc;
  if (ddrl_pop_num_params(h->pop, DDRL_NET_ACTOR) != N) { ddrl_agents_destroy(h.release()); return ddrl_fail(DDRL_ESTATE, "parameter count mismatch"); }
  {
    std::vector<double> w((size_t)lambda * N), stds((size_t)N, deviation);
    if ((rc = ddrl_pop_get_weights(h->pop, DDRL_NET_ACTOR, w.data()))) { ddrl_agents_destroy(h.release()); return rc; }
    // startx = the net's initial weights (:95); the reference passes seed 0 = clock to cmaes_init (:98)
    if ((rc = ddrl_cmaes_create((int)N, w.data(), stds.data(), (long)(seed % 2147483646ull) + 1, lambda, h->pop->device, &h->es))) {
      ddrl_agents_destroy(h.release());
      return rc;
    }
  }
  if (hipMalloc((void **)&h->d_best, (size_t)N * sizeof(float)) != hipSuccess) { ddrl_agents_destroy(h.release()); return ddrl_fail(DDRL_ENOMEM, "hipMalloc failed"); }
  init_state(h.get(), lambda, nb_sensors, nb_motors, seed);
  h->funvals.assign(lambda, 0.);
  h->evaluated.assign(lambda, 0);
  if ((rc = new_population(h.get()))) { ddrl_agents_destroy(h.release()); return rc; }
  *out = h.release();
  return DDRL_OK;
}

extern "C" ddrl_pop_t *ddrl_agents_population(ddrl_agents_t *h) { return h ? h->pop : nullptr; }
extern "C" int ddrl_agents_shape(ddrl_agents_t *h, int *kind, int *n_agents, int *nb_sensors, int *nb_motors) {
  if (!h) return ddrl_fail(DDRL_EINVAL, "NULL handle");
  if (kind) *kind = h->algo;
  if (n_agents) *n_agents = h->G;
  if (nb_sensors) *nb_sensors = h->S;
  if (nb_motors) *nb_motors = h->A;
  return DDRL_OK;
}
extern "C" ddrl_cmaes_t *ddrl_agents_cmaes(ddrl_agents_t *h) { return h ? h->es : nullptr; }

extern "C" int ddrl_agents_cmaes_state(ddrl_agents_t *h, int *lambda, int *n_evaluated, int *generations, int *error_count) {
  if (!h || h->algo != 2) return ddrl_fail(DDRL_EINVAL, "not a CMA-ES front end");
  if (lambda) *lambda = h->G;
  if (n_evaluated) *n_evaluated = h->n_evaluated;
  if (generations) *generations = h->generations;
  if (error_count) *error_count = h->error_count;
  return DDRL_OK;
}

// CMAESAg::start_instance (CMAESAg.hpp:212-225): individual g of the current generation goes into
// policy g while learning or before the first distribution update, the best-ever solution afterwards
static int cmaes_start_instance(ddrl_agents *h, const uint8_t *learning, const uint8_t *mask) {
  const int G = h->G;
  std::vector<uint8_t> mp(G, 0), mb(G, 0);
  bool any_p = false, any_b = false;
  for (int g = 0; g < G; ++g) {
    if (mask && !mask[g]) continue;
    if (learning[g] || !h->done_once) { mp[g] = 1; any_p = true; } else { mb[g] = 1; any_b = true; }
    if (learning[g]) h->ag[g].episode++;
  }
  int rc;
  if (any_p) {
    const float *dX; int ld;
    if ((rc = ddrl_cmaes_population_dev(h->es, &dX, &ld))) return rc;
    if ((rc = ddrl_pop_load_flat_dev(h->pop, DDRL_NET_ACTOR, dX, ld, mp.data()))) return rc;
  }
  if (any_b) {
    const int N = ddrl_pop_num_params(h->pop, DDRL_NET_ACTOR);
    std::vector<double> xb(N);
    if ((rc = ddrl_cmaes_get_vec(h->es, "xbestever", xb.data()))) return rc;     // getBestSolution (:328-335)
    std::vector<float> xf(xb.begin(), xb.end());
    HIPCHK(hipSetDevice(h->pop->device));
    HIPCHK(hipMemcpy(h->d_best, xf.data(), (size_t)N * sizeof(float), hipMemcpyHostToDevice));
    if ((rc = ddrl_pop_load_flat_dev(h->pop, DDRL_NET_ACTOR, h->d_best, 0, mb.data()))) return rc;
  }
  return DDRL_OK;
}

// ARLAgent::start_episode (ARLAgent.hpp:109-137) + DeepQNAg::_start_episode (DeepQNAg.hpp:237-247)
extern "C" int ddrl_agents_start_episode(ddrl_agents_t *h, const float *states, const uint8_t *learning, const uint8_t *mask) {
  if (!h || !states || !learning) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  for (int g = 0; g < h->G; ++g) {
    if (mask && !mask[g]) continue;
    AgentState &s = h->ag[g];
    s.time_for_ac = 1;
    s.inter_rewards.clear();
    if (s.last_episode_learning) s.last_sum_weighted_reward = s.sum_weighted_reward;
    s.sum_weighted_reward = 0.;
    s.sum_reward = 0.;
    s.last_episode_learning = learning[g] != 0;
    s.global_pow_gamma = 1.;
    memcpy(&h->last_state[(size_t)g * h->S], states + (size_t)g * h->S, h->S * sizeof(float));
    std::fill(s.ou_x.begin(), s.ou_x.end(), 0.);
    s.have_last = false;
    s.step = 0;
    s.td3_step = 0;                                       // TD3Ag.hpp:244
  }
  if (h->algo == 2) return cmaes_start_instance(h, learning, mask);
  if (h->dev.on) return dev_start_episode(h, learning, mask);
  return DDRL_OK;
}

// one update round = 1 + force_more_update passes of the fupd loop (DeepQNAg.hpp:341); the
// replay-size guard (:338) is applied per agent on the device
static int update_round(ddrl_agents *h, const uint8_t *mask) {
  if (h->algo == 3) {
    // TD3Ag::end_episode (TD3Ag.hpp:284-380): one update; the actor and the targets move when
    // current_step_counter % policy_freq == 0 (:333)
    std::vector<uint8_t> astep(h->G, 0);
    for (int g = 0; g < h->G; ++g) astep[g] = h->ag[g].td3_step % h->policy_freq == 0;
    return ddrl_pop_td3_update(h->pop, nullptr, nullptr, astep.data(), mask);
  }
  return ddrl_pop_ddpg_update_n(h->pop, 1 + h->force_more_update, nullptr, mask);   // one launch for the whole fupd loop
}

// ARLAgent::runf (ARLAgent.hpp:37-74) around DeepQNAg::_run (DeepQNAg.hpp:122-157) for every agent
// with mask[g] != 0; actions_out [G x A] (agents between two decisions get their previous action)
extern "C" int ddrl_agents_run(ddrl_agents_t *h, const float *reward, const float *states, const uint8_t *learning,
                               const uint8_t *goal, const uint8_t *last, const uint8_t *mask, float *actions_out) {
  if (!h || !reward || !states || !learning || !goal || !last || !actions_out) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  const int G = h->G, S = h->S, A = h->A;
  std::vector<uint8_t> decide(G, 0);
  bool any = false, any_insert = false;
  for (int g = 0; g < G; ++g) {
    h->insert_mask[g] = 0;
    if (mask && !mask[g]) continue;
    AgentState &s = h->ag[g];
    s.inter_rewards.push_back(reward[g]);
    s.time_for_ac--;
    if (s.time_for_ac == 0 || goal[g] || last[g]) {
      double r = *std::max_element(s.inter_rewards.begin(), s.inter_rewards.end());
      if (goal[g]) r = reward[g];                   // the last reward when entering an absorbing state (:56-58)
      s.last_receive_reward = r;
      h->rew[g] = (float)(r * h->reward_scale);
      h->goal[g] = goal[g] ? 1 : 0;
      decide[g] = 1;
      any = true;
      if (s.have_last && learning[g] && h->algo != 2) { h->insert_mask[g] = 1; any_insert = true; }   // DeepQNAg.hpp:128-131
    }
  }
  if (any) {
    // next_action = ann_testing->computeOut(sensors) with the weights BEFORE this step's update (:126)
    int rc = ddrl_pop_infer(h->pop, states, h->pure.data());
    if (rc) return rc;
    if (any_insert && (h->algo == 0 || h->algo == 3)) {
      // insertSample: push {last_state, last_action, sensors, reward, goal}, then end_episode(true) (:159-165;
      // TD3Ag.hpp:156-163 counts the step first)
      for (int g = 0; g < G; ++g) if (h->insert_mask[g]) h->ag[g].td3_step++;
      if ((rc = ddrl_pop_replay_append(h->pop, h->last_state.data(), h->last_action.data(), h->rew.data(), states,
                                       h->goal.data(), h->insert_mask.data()))) return rc;
      if ((rc = update_round(h, h->insert_mask.data()))) return rc;
    } else if (any_insert) {
      // trajectory.push_back({last_state, last_pure_action, last_action, sensors, reward, goal}) (OfflineCaclaAg.hpp:84-85)
      if ((rc = ddrl_nfac_traj_append(h->pop, h->last_state.data(), h->last_pure.data(), h->last_action.data(), states,
                                      h->rew.data(), h->goal.data(), h->insert_mask.data()))) return rc;
    }
    // per-agent host work: every agent has its own noise engine, so blocks of agents go to the pool
    // (same results whatever the thread count)
    if (G >= 256 && !h->pool) h->pool.reset(new DdrlPool((int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()))));
    constexpr int BLK = 32;
    auto block = [&](int b) {
    for (int g = b * BLK; g < std::min(G, (b + 1) * BLK); ++g) {
      if (!decide[g]) continue;
      AgentState &s = h->ag[g];
      float *act = &h->returned_ac[(size_t)g * A];
      memcpy(act, &h->pure[(size_t)g * A], A * sizeof(float));
      memcpy(&h->last_pure[(size_t)g * A], act, A * sizeof(float));
      if (learning[g] && h->algo != 2) explore(h, g, act);
      if (h->dev.informed)                       // neutral_action (DODevMLP.hpp:764-775): closed action dimensions get the PD set point
        for (size_t i = 0; i < h->dev.ac_control.size(); ++i)
          if (h->devst[g].ac_w[i] < 0.5) act[h->dev.ac_control[i]] = (float)h->dev.pd[h->dev.ac_control[i]];
      memcpy(&h->last_action[(size_t)g * A], act, A * sizeof(float));
      memcpy(&h->last_state[(size_t)g * S], states + (size_t)g * S, S * sizeof(float));
      s.have_last = true;
      s.step++;
      s.time_for_ac = h->decision_each;
      s.inter_rewards.clear();
      s.sum_weighted_reward += s.last_receive_reward * s.global_pow_gamma;        // ARLAgent.hpp:68-70
      s.sum_reward += s.last_receive_reward;
      s.global_pow_gamma *= h->gamma;
    }
    };
    const int nblk = (G + BLK - 1) / BLK;
    if (h->pool) h->pool->run(nblk, block);
    else for (int b = 0; b < nblk; ++b) block(b);
  }
  memcpy(actions_out, h->returned_ac.data(), (size_t)G * A * sizeof(float));
  return DDRL_OK;
}

// DDPGAg_end_episode = end_episode(learning) + end_instance(learning) (ddpg.cpp:52-55;
// DeepQNAg.hpp:336-339, 290-317)
extern "C" int ddrl_agents_end_episode(ddrl_agents_t *h, const uint8_t *learning, const uint8_t *mask) {
  if (!h || !learning) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  const int G = h->G;
  std::vector<uint8_t> lm(G, 0), tm(G, 0);
  std::vector<double> score(G, 0.);
  bool any_l = false, any_t = false;
  for (int g = 0; g < G; ++g) {
    if (mask && !mask[g]) continue;
    score[g] = h->ag[g].sum_weighted_reward;
    if (learning[g]) { lm[g] = 1; any_l = true; } else { tm[g] = 1; any_t = true; }
  }
  int rc;
  if (h->algo == 2) {
    // CMAESAg::end_episode + end_instance (CMAESAg.hpp:233-253), one episode per instance:
    // arFunvals[g] = accumulate(scores, 0.f) / size -- a float accumulator
    for (int g = 0; g < G; ++g) {
      if (!lm[g]) continue;
      h->funvals[g] = (double)(0.f + (float)(-score[g]));
      if (!h->evaluated[g]) { h->evaluated[g] = 1; h->n_evaluated++; }
    }
    if (h->n_evaluated >= G) {
      if ((rc = ddrl_cmaes_update(h->es, h->funvals.data()))) return rc;           // cmaes_UpdateDistribution (:246)
      h->done_once = true;
      h->generations++;
      if ((rc = new_population(h))) return rc;
    }
    return DDRL_OK;
  }
  const int method = h->pop->cfg.ewc_best_param_method;
  if (h->algo == 1) {
    // OfflineCaclaAg::end_episode (OfflineCaclaAg.hpp:403-595) + end_instance (:597-600)
    if (h->pop->cfg.ewc >= 0.) {
      if (any_t && method >= 4 && (rc = ddrl_pop_update_best_param(h->pop, score.data(), tm.data()))) return rc;
      if (any_l && method <= 3 && (rc = ddrl_pop_update_best_param(h->pop, score.data(), lm.data()))) return rc;
    }
    if (any_l) {
      if ((rc = ddrl_nfac_episode_end(h->pop, lm.data()))) return rc;               // trajectory_end_points.push_back (:419)
      std::vector<uint8_t> um(G, 0);
      bool any_u = false;
      for (int g = 0; g < G; ++g)
        if (lm[g] && h->ag[g].episode % h->update_each_episode == 0) { um[g] = 1; any_u = true; }   // :420-421
      if (any_u && (rc = ddrl_nfac_update(h->pop, um.data()))) return rc;
    }
    for (int g = 0; g < G; ++g) if (lm[g]) h->ag[g].episode++;
    return ddrl_pop_sync(h->pop);
  }
  if (any_l && (rc = update_round(h, lm.data()))) return rc;
  if (h->pop->cfg.ewc >= 0. && h->algo != 3) {            // TD3Ag::end_instance only counts the episode (TD3Ag.hpp:278-282)
    if (any_l && method <= 3 && (rc = ddrl_pop_update_best_param(h->pop, score.data(), lm.data()))) return rc;
    if (any_l && (rc = ddrl_pop_ewc_decay_update(h->pop, lm.data()))) return rc;
    if (any_t && method >= 4 && (rc = ddrl_pop_update_best_param(h->pop, score.data(), tm.data()))) return rc;
  }
  for (int g = 0; g < G; ++g) if (lm[g]) h->ag[g].episode++;
  return ddrl_pop_sync(h->pop);
}

extern "C" int ddrl_agents_stats(ddrl_agents_t *h, double *sum_weighted_reward, double *sum_reward, int32_t *episode) {
  if (!h) return ddrl_fail(DDRL_EINVAL, "NULL handle");
  for (int g = 0; g < h->G; ++g) {
    if (sum_weighted_reward) sum_weighted_reward[g] = h->ag[g].sum_weighted_reward;
    if (sum_reward) sum_reward[g] = h->ag[g].sum_reward;
    if (episode) episode[g] = h->ag[g].episode;
  }
  return DDRL_OK;
}

// ---- run state of the front end: ARLAgent / DeepQNAg / CMAESAg save_run + load_previous_run
// (DeepQNAg.hpp:504-537: nets, trajectory, episode; CMAESAg.hpp:279-308: scores, counters, the c-cmaes resume
// file) for all agents in one file, followed by the population's records and the optimiser's ----
static int agents_checkpoint(ddrl_agents *h, Ckpt &c) {
  int rc;
  const int meta[4] = {h->algo, h->G, h->S, h->A};
  if ((rc = c.same("agents.meta", meta, sizeof(meta)))) return rc;
  for (int g = 0; g < h->G; ++g) {
    AgentState &s = h->ag[g];
    double d[5] = {s.global_pow_gamma, s.sum_weighted_reward, s.sum_reward, s.last_sum_weighted_reward, s.last_receive_reward};
    int i[6] = {s.time_for_ac, s.last_episode_learning, s.have_last, s.episode, s.step, s.td3_step};
    std::ostringstream os;
    os << s.engine;
    std::string es = os.str();
    std::vector<char> eng(es.begin(), es.end());
    if ((rc = c.host("agent.d", d, sizeof(d))) || (rc = c.host("agent.i", i, sizeof(i))) || (rc = c.vec("agent.inter_rewards", s.inter_rewards)) ||
        (rc = c.vec("agent.ou_x", s.ou_x)) || (rc = c.vec("agent.engine", eng))) return rc;
    if (!c.writing) {
      s.global_pow_gamma = d[0]; s.sum_weighted_reward = d[1]; s.sum_reward = d[2]; s.last_sum_weighted_reward = d[3]; s.last_receive_reward = d[4];
      s.time_for_ac = i[0]; s.last_episode_learning = i[1] != 0; s.have_last = i[2] != 0; s.episode = i[3]; s.step = i[4]; s.td3_step = i[5];
      if ((int)s.ou_x.size() != h->A) return ddrl_fail(DDRL_EIO, "%s: noise state of the wrong length", c.path.c_str());
      std::istringstream is(std::string(eng.begin(), eng.end()));
      is >> s.engine;
      if (is.fail()) return ddrl_fail(DDRL_EIO, "%s: broken random engine state", c.path.c_str());
    }
  }
  int dev_on = h->dev.on ? 1 : 0;
  if ((rc = c.same("dev.on", &dev_on, sizeof(dev_on)))) return rc;
  if (dev_on) {
    int informed = h->dev.informed ? 1 : 0;
    if ((rc = c.val("dev.informed", informed))) return rc;
    if (!c.writing) h->dev.informed = informed != 0;
    for (int g = 0; g < h->G; ++g) {
      DevState &d = h->devst[g];
      int i[2] = {d.last_episode_changed, (int)d.im_index};
      if ((rc = c.vec("dev.st_w", d.st_w)) || (rc = c.vec("dev.ac_w", d.ac_w)) || (rc = c.vec("dev.all_scores", d.all_scores)) ||
          (rc = c.host("dev.i", i, sizeof(i)))) return rc;
      if (!c.writing) {
        d.last_episode_changed = i[0]; d.im_index = (unsigned)i[1];
        if (d.st_w.size() != h->dev.st_control.size() || d.ac_w.size() != h->dev.ac_control.size())
          return ddrl_fail(DDRL_EIO, "%s: developmental gates of the wrong length", c.path.c_str());
      }
    }
  }
  const size_t G = h->G, S = h->S, A = h->A;
  if ((rc = c.vec("last_state", h->last_state)) || (rc = c.vec("last_action", h->last_action)) || (rc = c.vec("last_pure", h->last_pure)) ||
      (rc = c.vec("returned_ac", h->returned_ac))) return rc;
  if (h->last_state.size() != G * S || h->last_action.size() != G * A || h->last_pure.size() != G * A || h->returned_ac.size() != G * A)
    return ddrl_fail(DDRL_EIO, "%s: agent buffers of the wrong length", c.path.c_str());
  if (h->algo == 2) {
    int i[5] = {h->n_evaluated, h->generations, h->error_count, h->done_once, h->racing};
    if ((rc = c.host("cmaesag.i", i, sizeof(i))) || (rc = c.vec("cmaesag.funvals", h->funvals)) || (rc = c.vec("cmaesag.evaluated", h->evaluated))) return rc;
    if (!c.writing) { h->n_evaluated = i[0]; h->generations = i[1]; h->error_count = i[2]; h->done_once = i[3] != 0; h->racing = i[4] != 0; }
    if (h->funvals.size() != G || h->evaluated.size() != G) return ddrl_fail(DDRL_EIO, "%s: fitness buffers of the wrong length", c.path.c_str());
    if ((rc = ddrl_cmaes_checkpoint(h->es, c))) return rc;
  }
  return ddrl_pop_checkpoint(h->pop, c);
}

static int agents_run_file(ddrl_agents *h, const char *path, bool save) {
  if (!h || !path) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  int rc = ddrl_pop_sync(h->pop);
  if (rc) return rc;
  Ckpt c;
  if ((rc = c.open(path, save)) || (rc = agents_checkpoint(h, c))) return rc;
  if ((rc = c.finish())) return rc;
  if (!save && h->dev.on && h->dev.heuristic != 0) return push_gates(h);          // the device gates follow the restored schedule state (heuristic 0: the population's file holds them)
  return DDRL_OK;
}
extern "C" int ddrl_agents_save_run(ddrl_agents_t *h, const char *path) { return agents_run_file(h, path, true); }
extern "C" int ddrl_agents_load_previous_run(ddrl_agents_t *h, const char *path) { return agents_run_file(h, path, false); }

// ---------------------------------------------------------------------------------------------
// G = 1 C flavour: the reference's names and argument types (ddpg.cpp:9-88; gym/agent.py:113-118)
// ---------------------------------------------------------------------------------------------
struct DDPGAgShim {
  unsigned nb_motors, nb_sensors;
  int algo = 0;
  ddrl_agents *ag = nullptr;
};

extern "C" int64_t DDPGAg_new(unsigned nb_motors, unsigned nb_sensors) {
  DDPGAgShim *s = new DDPGAgShim();
  s->nb_motors = nb_motors;
  s->nb_sensors = nb_sensors;
  return (int64_t)(intptr_t)s;
}

// argv: "--config <file>" (default config.ini, ddpg.cpp:22,35-37); unknown options are ignored like
// allow_unregistered().  The library never exits: failures are reported on stderr and leave the
// agent unconfigured (every later call is then a no-op returning NULL / nothing).
extern "C" void DDPGAg_unique_invoke(int64_t hnd, int argc, char **argv) {
  DDPGAgShim *s = (DDPGAgShim *)(intptr_t)hnd;
  if (!s) return;
  std::string config = "config.ini";
  uint64_t seed = std::random_device{}();
  for (int i = 0; i < argc; ++i) {
    if (!argv[i]) continue;
    const std::string a(argv[i]);
    if (a == "--config" && i + 1 < argc && argv[i + 1]) config = argv[++i];
    else if (a.rfind("--config=", 0) == 0) config = a.substr(9);
    else if (a == "--seed" && i + 1 < argc && argv[i + 1]) seed = strtoull(argv[++i], nullptr, 10);
  }
  const int rc = s->algo == 0 ? ddrl_agents_create_ddpg(config.c_str(), 1, (int)s->nb_sensors, (int)s->nb_motors, -1, seed, &s->ag)
               : s->algo == 3 ? ddrl_agents_create_td3(config.c_str(), 1, (int)s->nb_sensors, (int)s->nb_motors, -1, seed, &s->ag)
                              : ddrl_agents_create_nfac(config.c_str(), 1, (int)s->nb_sensors, (int)s->nb_motors, 0, -1, seed, &s->ag);
  if (rc != DDRL_OK) fprintf(stderr, "%s_unique_invoke: %s\n", s->algo == 0 ? "DDPGAg" : s->algo == 3 ? "TD3Ag" : "OfflineCaclaAg", ddrl_last_error());
}

extern "C" void DDPGAg_start_episode(int64_t hnd, const double *state, bool learning) {
  DDPGAgShim *s = (DDPGAgShim *)(intptr_t)hnd;
  if (!s || !s->ag || !state) return;
  std::vector<float> st(state, state + s->nb_sensors);
  const uint8_t l = learning;
  if (ddrl_agents_start_episode(s->ag, st.data(), &l, nullptr) != DDRL_OK) fprintf(stderr, "DDPGAg_start_episode: %s\n", ddrl_last_error());
}

extern "C" const double *DDPGAg_run(int64_t hnd, double reward, const double *sensors, bool learning, bool goal_reached, bool last) {
  DDPGAgShim *s = (DDPGAgShim *)(intptr_t)hnd;
  if (!s || !s->ag || !sensors) return nullptr;
  std::vector<float> st(sensors, sensors + s->nb_sensors), act(s->nb_motors);
  const float r = (float)reward;
  const uint8_t l = learning, g = goal_reached, e = last;
  if (ddrl_agents_run(s->ag, &r, st.data(), &l, &g, &e, nullptr, act.data()) != DDRL_OK) {
    fprintf(stderr, "DDPGAg_run: %s\n", ddrl_last_error());
    return nullptr;
  }
  for (unsigned i = 0; i < s->nb_motors; ++i) s->ag->ret_double[i] = act[i];
  return s->ag->ret_double.data();
}

extern "C" void DDPGAg_end_episode(int64_t hnd, bool learning) {
  DDPGAgShim *s = (DDPGAgShim *)(intptr_t)hnd;
  if (!s || !s->ag) return;
  const uint8_t l = learning;
  if (ddrl_agents_end_episode(s->ag, &l, nullptr) != DDRL_OK) fprintf(stderr, "DDPGAg_end_episode: %s\n", ddrl_last_error());
}

// the reference appends "<episode> <step> <treward> <agent dump>" to 0.learning.data / 0.testing.data
// (ddpg.cpp:64-68); the agent dump fields are specific to its Caffe wrapper, only the prefix is kept
extern "C" void DDPGAg_dump(int64_t hnd, bool learning, int episode, int step, double treward) {
  DDPGAgShim *s = (DDPGAgShim *)(intptr_t)hnd;
  if (!s || !s->ag) return;
  FILE *f = fopen(learning ? "0.learning.data" : "0.testing.data", "a");
  if (!f) return;
  fprintf(f, "%d %d %f %f\n", episode, step, treward, s->ag->ag[0].sum_weighted_reward);
  fclose(f);
}

extern "C" void DDPGAg_display(int64_t hnd, bool learning, int episode, int step, double treward) {
  DDPGAgShim *s = (DDPGAgShim *)(intptr_t)hnd;
  if (!s || !s->ag) return;
  printf("%s %-6d%-7d%-7f %f\n", learning ? "L" : "T", episode, step, treward, s->ag->ag[0].sum_weighted_reward);
}

// weights of the four nets, flat reference order, as raw doubles: agent<episode>.ddrl_pop
static int save_load(DDPGAgShim *s, int episode, bool save) {
  const std::string path = "agent" + std::to_string(episode) + ".ddrl_pop";
  FILE *f = fopen(path.c_str(), save ? "wb" : "rb");
  if (!f) return ddrl_fail(DDRL_EIO, "cannot open %s", path.c_str());
  int rc = DDRL_OK;
  for (int net = 0; net < (s->algo == 0 ? 4 : s->algo == 3 ? 6 : 2) && rc == DDRL_OK; ++net) {
    const int n = ddrl_pop_num_params(s->ag->pop, net);
    std::vector<double> w(n);
    if (save) {
      rc = ddrl_pop_get_weights(s->ag->pop, net, w.data());
      if (rc == DDRL_OK && fwrite(w.data(), sizeof(double), n, f) != (size_t)n) rc = ddrl_fail(DDRL_EIO, "short write to %s", path.c_str());
    } else {
      if (fread(w.data(), sizeof(double), n, f) != (size_t)n) rc = ddrl_fail(DDRL_EIO, "%s does not match this agent's nets", path.c_str());
      else rc = ddrl_pop_set_weights(s->ag->pop, net, w.data());
    }
  }
  fclose(f);
  return rc;
}

extern "C" void DDPGAg_save(int64_t hnd, int episode) {
  DDPGAgShim *s = (DDPGAgShim *)(intptr_t)hnd;
  if (s && s->ag && save_load(s, episode, true) != DDRL_OK) fprintf(stderr, "DDPGAg_save: %s\n", ddrl_last_error());
}

extern "C" void DDPGAg_load(int64_t hnd, int episode) {
  DDPGAgShim *s = (DDPGAgShim *)(intptr_t)hnd;
  if (s && s->ag && save_load(s, episode, false) != DDRL_OK) fprintf(stderr, "DDPGAg_load: %s\n", ddrl_last_error());
}

// not in the reference's binding (it leaks the agent); provided so that callers can release the GPU
extern "C" void DDPGAg_delete(int64_t hnd) {
  DDPGAgShim *s = (DDPGAgShim *)(intptr_t)hnd;
  if (!s) return;
  ddrl_agents_destroy(s->ag);
  delete s;
}

// NFAC(lambda)-V: the same C flavour under the reference's names (agent/cacla/src/pybinding/nfac.cpp:17-97;
// gym/agent.py:17-62)
extern "C" int64_t OfflineCaclaAg_new(unsigned nb_motors, unsigned nb_sensors) {
  DDPGAgShim *s = (DDPGAgShim *)(intptr_t)DDPGAg_new(nb_motors, nb_sensors);
  s->algo = 1;
  return (int64_t)(intptr_t)s;
}
extern "C" void OfflineCaclaAg_unique_invoke(int64_t ag, int argc, char **argv) { DDPGAg_unique_invoke(ag, argc, argv); }
extern "C" void OfflineCaclaAg_start_episode(int64_t ag, const double *state, bool learning) { DDPGAg_start_episode(ag, state, learning); }
extern "C" const double *OfflineCaclaAg_run(int64_t ag, double reward, const double *sensors, bool learning, bool goal_reached, bool last) {
  return DDPGAg_run(ag, reward, sensors, learning, goal_reached, last);
}
extern "C" void OfflineCaclaAg_end_episode(int64_t ag, bool learning) { DDPGAg_end_episode(ag, learning); }
extern "C" void OfflineCaclaAg_dump(int64_t ag, bool learning, int episode, int step, double treward) { DDPGAg_dump(ag, learning, episode, step, treward); }
extern "C" void OfflineCaclaAg_display(int64_t ag, bool learning, int episode, int step, double treward) { DDPGAg_display(ag, learning, episode, step, treward); }
extern "C" void OfflineCaclaAg_save(int64_t ag, int episode) { DDPGAg_save(ag, episode); }
extern "C" void OfflineCaclaAg_load(int64_t ag, int episode) { DDPGAg_load(ag, episode); }
extern "C" void OfflineCaclaAg_delete(int64_t ag) { DDPGAg_delete(ag); }

// TD3: the reference builds the same binding file with WANT_TD3 into libddrl-td3, which therefore exports
// the DDPGAg_* names (agent/deepqn/src/pybinding/ddpg.cpp:3-5, CMakeLists.txt:84-88; gym/agent.py:112,157).
// One library cannot export them twice: here the TD3 flavour carries its own prefix, same signatures.
extern "C" int64_t TD3Ag_new(unsigned nb_motors, unsigned nb_sensors) {
  DDPGAgShim *s = (DDPGAgShim *)(intptr_t)DDPGAg_new(nb_motors, nb_sensors);
  s->algo = 3;
  return (int64_t)(intptr_t)s;
}
extern "C" void TD3Ag_unique_invoke(int64_t ag, int argc, char **argv) { DDPGAg_unique_invoke(ag, argc, argv); }
extern "C" void TD3Ag_start_episode(int64_t ag, const double *state, bool learning) { DDPGAg_start_episode(ag, state, learning); }
extern "C" const double *TD3Ag_run(int64_t ag, double reward, const double *sensors, bool learning, bool goal_reached, bool last) {
  return DDPGAg_run(ag, reward, sensors, learning, goal_reached, last);
}
extern "C" void TD3Ag_end_episode(int64_t ag, bool learning) { DDPGAg_end_episode(ag, learning); }
extern "C" void TD3Ag_dump(int64_t ag, bool learning, int episode, int step, double treward) { DDPGAg_dump(ag, learning, episode, step, treward); }
extern "C" void TD3Ag_display(int64_t ag, bool learning, int episode, int step, double treward) { DDPGAg_display(ag, learning, episode, step, treward); }
extern "C" void TD3Ag_save(int64_t ag, int episode) { DDPGAg_save(ag, episode); }
extern "C" void TD3Ag_load(int64_t ag, int episode) { DDPGAg_load(ag, episode); }
extern "C" void TD3Ag_delete(int64_t ag) { DDPGAg_delete(ag); }
