// sim.cpp -- the population analogue of arch::Simulator (SURVEY.md 8f-4;
// /root/reference/common/include/arch/Simulator.hpp:76-248): G environments, one per agent, stepped by a
// thread pool in lock step and fed to the agent front end (ddrl_agents_*) as [G x ...] batches, with the
// reference's episode schedule ([simulation] keys, :300-328), its per-episode log lines (:210-248) in one
// directory per agent so that the existing plotting scripts keep working, and its --continue protocol
// (:76-84, 196-205) on top of ddrl_agents_save_run.
//
// Host code only.  The environments themselves (ODE worlds in the reference) are the caller's: a table of
// C callbacks, one environment instance per agent, each driven by one thread at a time.
#include <unistd.h>
#include "../../include/ddrl_pop.h"
#include "host_util.h"
#include "pool.h"

#include <errno.h>
#include <sys/stat.h>
#include <algorithm>
#include <atomic>
#include <memory>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>


struct ddrl_sim {
  ddrl_agents_t *agents = nullptr;      // borrowed
  ddrl_env_vtable env;
  int G = 0, S = 0, A = 0, kind = 0;
  // [simulation] (Simulator.hpp:300-328)
  long max_episode = 0, test_episode_per_episode = 0, test_episode_at_end = 0;
  long dump_log_each = 1, display_log_each = 1, save_agent_each = 0, continue_save_each = 50;
  bool display_learning = true, can_continue = false;
  std::string out_dir;
  long starting_episode = 0;
  std::unique_ptr<DdrlPool> pool;
  // per-tick batches
  std::vector<float> states, rewards, actions;
  std::vector<double> percept, act_d;
  std::vector<uint8_t> learning, goal, last, mask, running;
  std::vector<int> steps;
  double minutes = 0.;
};

static std::string agent_dir(const ddrl_sim *s, int g) { return s->out_dir + "/" + std::to_string(g); }

extern "C" int ddrl_sim_create(ddrl_agents_t *agents, const ddrl_env_vtable *env, const char *ini_path, const char *out_dir,
                               int n_threads, int can_continue, ddrl_sim_t **out) {
  if (!agents || !env || !ini_path || !out) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  *out = nullptr;
  if (!env->reset_episode || !env->perceptions || !env->performance || !env->apply || !env->running || !env->final_state)
    return ddrl_fail(DDRL_EINVAL, "environment table: reset_episode, perceptions, performance, apply, running and final_state are mandatory");
  std::unique_ptr<ddrl_sim> s(new ddrl_sim());
  s->agents = agents;
  s->env = *env;
  int rc = ddrl_agents_shape(agents, &s->kind, &s->G, &s->S, &s->A);
  if (rc) return rc;
  DdrlIni ini;
  std::string err;
  if (!ini.load(ini_path, &err)) return ddrl_fail(DDRL_EIO, "%s", err.c_str());
  try {
    s->max_episode = ini.get_long("simulation.max_episode");
    s->test_episode_per_episode = ini.get_long("simulation.test_episode_per_episode");
    s->test_episode_at_end = ini.get_long("simulation.test_episode_at_end");
    s->dump_log_each = ini.get_long("simulation.dump_log_each");
    s->display_log_each = ini.get_long("simulation.display_log_each");
    s->save_agent_each = ini.get_long("simulation.save_agent_each");
    if (ini.has("simulation.display_learning")) s->display_learning = ini.get_bool("simulation.display_learning") != 0;
    if (ini.has("simulation.continue_save_each")) s->continue_save_each = ini.get_long("simulation.continue_save_each");
  } catch (const std::exception &e) {
    return ddrl_fail(DDRL_EIO, "%s: %s", ini_path, e.what());
  }
  if (s->max_episode < 0 || s->test_episode_at_end < 0 || s->dump_log_each < 1 || s->display_log_each < 1 || s->continue_save_each < 1)
    return ddrl_fail(DDRL_EINVAL, "[simulation]: negative episode counts or log periods < 1");
  s->can_continue = can_continue != 0;
  s->out_dir = out_dir && *out_dir ? out_dir : ".";
  mkdir(s->out_dir.c_str(), 0777);
  for (int g = 0; g < s->G; ++g)
    if (mkdir(agent_dir(s.get(), g).c_str(), 0777) != 0 && errno != EEXIST)
      return ddrl_fail(DDRL_EIO, "cannot create %s", agent_dir(s.get(), g).c_str());
  if (n_threads < 1) n_threads = 1;
  s->pool.reset(new DdrlPool(std::min(n_threads, s->G)));
  const size_t G = s->G;
  s->states.assign(G * s->S, 0.f); s->rewards.assign(G, 0.f); s->actions.assign(G * s->A, 0.f);
  s->percept.assign(G * s->S, 0.); s->act_d.assign(G * s->A, 0.);
  s->learning.assign(G, 0); s->goal.assign(G, 0); s->last.assign(G, 0); s->mask.assign(G, 0); s->running.assign(G, 0);
  s->steps.assign(G, 0);
  // --continue: a previous run's state is picked up (Simulator.hpp:76-84)
  if (s->can_continue) {
    const std::string simu = s->out_dir + "/continue.simu.data", run = s->out_dir + "/continue.ddrl_pop";
    FILE *f = fopen(simu.c_str(), "r");
    if (f) {
      long ep = 0;
      const int got = fscanf(f, "%ld", &ep);
      fclose(f);
      if (got != 1 || ep < 0) return ddrl_fail(DDRL_EIO, "%s is broken", simu.c_str());
      if ((rc = ddrl_agents_load_previous_run(agents, run.c_str()))) return rc;
      s->starting_episode = ep;
    }
  }
  *out = s.release();
  return DDRL_OK;
}

extern "C" int ddrl_sim_destroy(ddrl_sim_t *s) {
  delete s;
  return DDRL_OK;
}

// the per-episode lines of Simulator::dump_and_display (:210-248), one directory per agent
static int dump_and_display(ddrl_sim *s, bool learning, long episode, long tepisode) {
  bool display = episode % s->display_log_each == 0, dump = episode % s->dump_log_each == 0;
  if (!learning) { display = (episode + tepisode) % s->display_log_each == 0; dump = true; }
  if (!dump && !display) return DDRL_OK;
  const int G = s->G;
  std::vector<double> swr(G);
  std::vector<int32_t> stored(G, 0);
  int rc = ddrl_agents_stats(s->agents, swr.data(), nullptr, nullptr);
  if (rc) return rc;
  ddrl_pop_t *pop = ddrl_agents_population(s->agents);
  if (s->kind == 0 || s->kind == 3) rc = ddrl_pop_replay_len(pop, stored.data());       // trajectory.size() (DeepQNAg.hpp:552-556)
  else if (s->kind == 1) rc = ddrl_nfac_traj_len(pop, stored.data());
  if (rc) return rc;
  char envtxt[256];
  for (int g = 0; g < G; ++g) {
    envtxt[0] = 0;
    if (s->env.dump) s->env.dump(s->env.user, g, envtxt, (int)sizeof(envtxt));
    envtxt[sizeof(envtxt) - 1] = 0;
    if (display && (learning ? s->display_learning : true))
      printf("%s %-4d %-6ld%-7d %12.10f %s\n", learning ? "L" : "T", g, episode, s->steps[g], swr[g], envtxt);
    if (dump) {
      const std::string path = agent_dir(s, g) + (learning ? "/0.learning.data" : "/0." + std::to_string(tepisode) + ".testing.data");
      FILE *f = fopen(path.c_str(), "a");
      if (!f) return ddrl_fail(DDRL_EIO, "cannot append to %s", path.c_str());
      if (s->kind == 2) fprintf(f, "%ld %d %8.5f %s\n", episode, s->steps[g], swr[g], envtxt);          // CMAESAg::_dump
      else fprintf(f, "%ld %d %25.22f %8d %s\n", episode, s->steps[g], swr[g], stored[g], envtxt);       // DeepQNAg::_dump
      fclose(f);
    }
  }
  return DDRL_OK;
}

// Simulator::run_episode (:139-208) for all G (agent, environment) pairs at once.  Environments finish at
// different steps: an agent whose environment stopped gets its closing runf(..., final_state, last = true)
// in the tick it stops and is masked out from then on; end_episode follows for everybody.
extern "C" int ddrl_sim_run_episode(ddrl_sim_t *s, int learning, long lepisode, long tepisode) {
  if (!s) return ddrl_fail(DDRL_EINVAL, "NULL handle");
  const int G = s->G, S = s->S, A = s->A;
  const ddrl_env_vtable &E = s->env;
  std::fill(s->learning.begin(), s->learning.end(), (uint8_t)(learning != 0));
  s->pool->run(G, [&](int g) {
    E.reset_episode(E.user, g, learning);
    E.perceptions(E.user, g, &s->percept[(size_t)g * S]);
    for (int i = 0; i < S; ++i) s->states[(size_t)g * S + i] = (float)s->percept[(size_t)g * S + i];
    s->steps[g] = 0;
    s->running[g] = 1;
  });
  int rc = ddrl_agents_start_episode(s->agents, s->states.data(), s->learning.data(), nullptr);
  if (rc) return rc;
  int active = G;
  while (active > 0) {
    s->pool->run(G, [&](int g) {
      s->mask[g] = s->running[g];
      if (!s->running[g]) return;
      const bool run = E.running(E.user, g) != 0;
      E.perceptions(E.user, g, &s->percept[(size_t)g * S]);
      for (int i = 0; i < S; ++i) s->states[(size_t)g * S + i] = (float)s->percept[(size_t)g * S + i];
      s->rewards[g] = (float)E.performance(E.user, g);
      s->goal[g] = run ? 0 : (uint8_t)(E.final_state(E.user, g) != 0);
      s->last[g] = run ? 0 : 1;
    });
    if ((rc = ddrl_agents_run(s->agents, s->rewards.data(), s->states.data(), s->learning.data(), s->goal.data(), s->last.data(),
                              s->mask.data(), s->actions.data()))) return rc;
    s->pool->run(G, [&](int g) {
      if (!s->running[g]) return;
      if (s->last[g]) { s->running[g] = 0; return; }
      for (int i = 0; i < A; ++i) s->act_d[(size_t)g * A + i] = s->actions[(size_t)g * A + i];
      E.apply(E.user, g, &s->act_d[(size_t)g * A]);
      s->steps[g]++;
    });
    active = 0;
    for (int g = 0; g < G; ++g) active += s->running[g];
  }
  if (E.next_instance) s->pool->run(G, [&](int g) { E.next_instance(E.user, g, learning); });
  if ((rc = ddrl_agents_end_episode(s->agents, s->learning.data(), nullptr))) return rc;
  if ((rc = dump_and_display(s, learning != 0, lepisode, tepisode))) return rc;
  if (s->can_continue && learning && lepisode % s->continue_save_each == 0 && lepisode > 0) {          // :196-200
    const std::string run = s->out_dir + "/continue.ddrl_pop", simu = s->out_dir + "/continue.simu.data";
    if ((rc = ddrl_agents_save_run(s->agents, run.c_str()))) return rc;
    // the run file went to disk under a temporary name and was renamed over the old one (ckpt.h); the episode marker
    // follows the same way, so at any instant both files are complete (a stop between the two renames resumes the
    // new state under the old episode number: episodes are repeated in the logs, nothing is lost)
    const std::string tmp = simu + ".tmp";
    FILE *f = fopen(tmp.c_str(), "w");
    if (!f) return ddrl_fail(DDRL_EIO, "cannot write %s", tmp.c_str());
    const bool okw = fprintf(f, "%ld\n", lepisode + 1) > 0 && fflush(f) == 0 && fsync(fileno(f)) == 0;
    if (fclose(f) != 0 || !okw || rename(tmp.c_str(), simu.c_str()) != 0) { remove(tmp.c_str()); return ddrl_fail(DDRL_EIO, "cannot write %s", simu.c_str()); }
  }
  return DDRL_OK;
}

// Simulator::run_loop + run (:86-128)
extern "C" int ddrl_sim_run(ddrl_sim_t *s) {
  if (!s) return ddrl_fail(DDRL_EINVAL, "NULL handle");
  const auto t0 = std::chrono::steady_clock::now();
  const long first = s->starting_episode, fepisode = s->starting_episode > 0 ? s->max_episode : s->max_episode + first;
  int rc;
  for (long episode = first; episode < fepisode; ++episode) {
    if ((rc = ddrl_sim_run_episode(s, 1, episode, 0))) return rc;
    for (long t = 0; t < s->test_episode_per_episode; ++t)
      if ((rc = ddrl_sim_run_episode(s, 0, episode, t))) return rc;
    if (s->test_episode_per_episode < -1 && episode % (-s->test_episode_per_episode) == 0)
      if ((rc = ddrl_sim_run_episode(s, 0, episode, 0))) return rc;
  }
  for (long t = 0; t < s->test_episode_at_end; ++t)
    if ((rc = ddrl_sim_run_episode(s, 0, s->max_episode + first, t))) return rc;       // max_episode + starting_ep (:107)
  s->minutes = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / 60.;
  FILE *f = fopen((s->out_dir + "/time_elapsed").c_str(), "a");                                          // DEFAULT_END_FILE (:127)
  if (f) { fprintf(f, "%f\n", s->minutes); fclose(f); }
  return DDRL_OK;
}
