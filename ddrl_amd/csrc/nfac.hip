// nfac.hip -- host side of the NFAC(lambda)-V population (ddrl_nfac_* of include/ddrl_pop.h):
// configuration, trajectory buffers and the launch sequence of one OfflineCaclaAg::end_episode
// (/root/reference/agent/cacla/include/OfflineCaclaAg.hpp:403-595).  All arithmetic is in
// kernels_nfac.hip; there is no CPU path.
#include "pop_host.h"
#include <random>
#include <sstream>

extern "C" {
size_t nfac_lds_need(const NfacParams *P);
hipError_t nfac_values_launch(const NfacParams *P, int chunks, hipStream_t st);
hipError_t nfac_scan_launch(const NfacParams *P, int mode, int maxlen, hipStream_t st);
hipError_t nfac_grad_launch(const NfacParams *P, int mode, int chunks, int iter_add, int last, hipStream_t st);
hipError_t nfac_apply_launch(const NfacParams *P, int mode, int iter_add, int last, hipStream_t st);
hipError_t nfac_finish_launch(const NfacParams *P, int critic_steps, int actor_steps, hipStream_t st);
hipError_t nfac_reinit_launch(const NfacParams *P, const float *fresh, hipStream_t st);
}

struct NfacHost {
  ddrl_nfac_config cfg;
  NfacParams P;
  int rowlen = 0, maxchunks = 0;
  float *d_traj = nullptr;
  int *d_len = nullptr;
  uint8_t *d_endflag = nullptr;
  std::vector<int> h_len;
  std::vector<std::vector<int>> h_ends;       // trajectory_end_points per agent
  float *d_V = nullptr, *d_nextV = nullptr, *d_tgt = nullptr, *d_lw = nullptr, *d_deltas = nullptr, *d_gpart_a = nullptr, *d_gpart_v = nullptr,
        *d_loss_part = nullptr, *d_loss = nullptr, *d_pa = nullptr;
  uint8_t *d_rowmode = nullptr;
  int *d_npos = nullptr;
  int *d_chunk_done = nullptr;                // [G] nfac_grad_kernel's per-agent count of published chunk partials (self-resetting)
  float *d_fresh = nullptr;                   // vnn_from_scratch: the freshly drawn critics of one fitted iteration [G x stride]
  std::mt19937_64 scratch_eng;                // ... and the engine they come from
};

void ddrl_nfac_host_free(NfacHost *n) { delete n; }

extern "C" int ddrl_nfac_config_default(ddrl_nfac_config *c) {
  if (!c) return ddrl_fail(DDRL_EINVAL, "cfg is NULL");
  memset(c, 0, sizeof(*c));
  c->n_agents = 1; c->nb_sensors = 45; c->nb_motors = 17;
  c->n_hidden_a = 2; c->hidden_unit_a[0] = 64; c->hidden_unit_a[1] = 64;
  c->n_hidden_v = 2; c->hidden_unit_v[0] = 64; c->hidden_unit_v[1] = 64;
  c->update_delta_neg = 0; c->update_critic_first = 1;
  c->number_fitted_iteration = 10; c->stoch_iter_actor = 10; c->stoch_iter_critic = 10;
  c->actor_output_layer_type = 2; c->hidden_layer_type = 1; c->momentum = 0;
  c->gae = 1; c->inverting_gradient = 0; c->max_trajectory = 1024;
  c->gamma = 0.99; c->lambda = 0.9; c->alpha_a = 1e-4; c->alpha_v = 1e-3;
  c->ewc = -1.; c->ewc_decay = -1.; c->ewc_fisher_beta = 0.999; c->ewc_best_param_method = 0; c->ewc_for_critic = 0;
  c->device = -1;
  c->vnn_from_scratch = 0; c->corrected_update_ac = 0;
  c->corrected_update_ac_factor = 1.;      // (the reference leaves it uninitialised when the key is missing, OfflineCaclaAg.hpp:171-176,676)
  return DDRL_OK;
}

extern "C" int ddrl_nfac_config_from_ini(const char *path, ddrl_nfac_config *c) {
  if (!path || !c) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  DdrlIni ini;
  std::string err;
  if (!ini.load(path, &err)) return ddrl_fail(DDRL_EIO, "%s", err.c_str());
  const int G = c->n_agents, S = c->nb_sensors, A = c->nb_motors, dev = c->device, maxT = c->max_trajectory;
  ddrl_nfac_config_default(c);
  c->n_agents = G > 0 ? G : 1;
  if (S > 0) c->nb_sensors = S;
  if (A > 0) c->nb_motors = A;
  if (maxT > 0) c->max_trajectory = maxT;
  c->device = dev;
  try {
    std::vector<long> hv = ini.get_list("agent.hidden_unit_v"), ha = ini.get_list("agent.hidden_unit_a");
    if (hv.empty() || ha.empty() || hv.size() > DDRL_MAX_HIDDEN || ha.size() > DDRL_MAX_HIDDEN)
      return ddrl_fail(DDRL_EINVAL, "hidden_unit_v/a: 1..%d layers supported", DDRL_MAX_HIDDEN);
    c->n_hidden_v = (int)hv.size(); for (size_t i = 0; i < hv.size(); ++i) c->hidden_unit_v[i] = (int)hv[i];
    c->n_hidden_a = (int)ha.size(); for (size_t i = 0; i < ha.size(); ++i) c->hidden_unit_a[i] = (int)ha[i];
    // mandatory keys, OfflineCaclaAg.hpp:124-141 (+ agent.gamma, ARLAgent.hpp:141)
    c->gamma = ini.get_double("agent.gamma");
    c->update_delta_neg = ini.get_bool("agent.update_delta_neg");
    c->update_critic_first = ini.get_bool("agent.update_critic_first");
    c->number_fitted_iteration = (int)ini.get_long("agent.number_fitted_iteration");
    c->stoch_iter_actor = (int)ini.get_long("agent.stoch_iter_actor");
    c->stoch_iter_critic = (int)ini.get_long("agent.stoch_iter_critic");
    c->batch_norm_actor = (int)ini.get_long("agent.batch_norm_actor");
    c->batch_norm_critic = (int)ini.get_long("agent.batch_norm_critic");
    c->actor_output_layer_type = (int)ini.get_long("agent.actor_output_layer_type");
    c->hidden_layer_type = (int)ini.get_long("agent.hidden_layer_type");
    c->alpha_a = ini.get_double("agent.alpha_a");
    c->alpha_v = ini.get_double("agent.alpha_v");
    c->lambda = ini.get_double("agent.lambda");
    c->momentum = (int)ini.get_long("agent.momentum");
    c->vnn_from_scratch = ini.get_bool("agent.vnn_from_scratch");        // mandatory like the others (:129)
    // optional keys with the reference's defaults (:142-179)
    c->gae = 0; c->inverting_gradient = 0;
    c->corrected_update_ac = 0;
    if (ini.has("agent.corrected_update_ac")) c->corrected_update_ac = ini.get_bool("agent.corrected_update_ac");
    if (c->corrected_update_ac && ini.has("agent.corrected_update_ac_factor"))
      c->corrected_update_ac_factor = ini.get_double("agent.corrected_update_ac_factor");
    if (ini.has("agent.inverting_gradient")) c->inverting_gradient = ini.get_bool("agent.inverting_gradient");
    if (c->lambda >= 0.) c->gae = ini.get_bool("agent.gae");
    if (ini.has("devnn.ewc")) c->ewc = ini.get_double("devnn.ewc");
    if (c->ewc >= 0.) {
      if (ini.has("devnn.ewc_decay")) c->ewc_decay = ini.get_double("devnn.ewc_decay");
      c->ewc_fisher_beta = ini.get_double("devnn.ewc_fisher_beta");
      c->ewc_best_param_method = (int)ini.get_long("devnn.ewc_best_param_method");
      c->ewc_for_critic = (int)ini.get_long("devnn.ewc_for_critic");
    }
  } catch (const std::exception &e) {
    return ddrl_fail(DDRL_EIO, "%s: %s", path, e.what());
  }
  return DDRL_OK;
}

static int nfac_validate(const ddrl_nfac_config *c) {
  if (c->n_agents < 1) return ddrl_fail(DDRL_EINVAL, "n_agents < 1");
  if (c->nb_sensors < 1 || c->nb_motors < 1) return ddrl_fail(DDRL_EINVAL, "nb_sensors/nb_motors < 1");
  if (c->nb_sensors > DDRL_MAXW || c->nb_motors > DDRL_MAXW) return ddrl_fail(DDRL_EINVAL, "nb_sensors/nb_motors > %d", DDRL_MAXW);
  if (c->n_hidden_a < 1 || c->n_hidden_a > DDRL_MAX_HIDDEN || c->n_hidden_v < 1 || c->n_hidden_v > DDRL_MAX_HIDDEN)
    return ddrl_fail(DDRL_EINVAL, "1..%d hidden layers supported", DDRL_MAX_HIDDEN);
  for (int i = 0; i < c->n_hidden_a; ++i)
    if (c->hidden_unit_a[i] < 1 || c->hidden_unit_a[i] > DDRL_MAXW) return ddrl_fail(DDRL_EINVAL, "hidden_unit_a[%d]=%d outside 1..%d", i, c->hidden_unit_a[i], DDRL_MAXW);
  for (int i = 0; i < c->n_hidden_v; ++i)
    if (c->hidden_unit_v[i] < 1 || c->hidden_unit_v[i] > DDRL_MAXW) return ddrl_fail(DDRL_EINVAL, "hidden_unit_v[%d]=%d outside 1..%d", i, c->hidden_unit_v[i], DDRL_MAXW);
  if (c->hidden_layer_type < 1 || c->hidden_layer_type > 3) return ddrl_fail(DDRL_EINVAL, "hidden_layer_type not in (1,2,3)");
  if (c->actor_output_layer_type < 0 || c->actor_output_layer_type > 3) return ddrl_fail(DDRL_EINVAL, "actor_output_layer_type not in 0..3");
  if (c->momentum < 0 || c->momentum > 3) return ddrl_fail(DDRL_EINVAL, "momentum not in 0..3");
  if (c->batch_norm_actor || c->batch_norm_critic) return ddrl_fail(DDRL_EINVAL, "batch_norm != 0 is not supported (BatchNorm is outside this round's scope)");
  if (!(c->alpha_a > 0) || !(c->alpha_v > 0)) return ddrl_fail(DDRL_EINVAL, "alpha <= 0");
  if (c->number_fitted_iteration < 0 || c->stoch_iter_actor < 0 || c->stoch_iter_critic < 0) return ddrl_fail(DDRL_EINVAL, "negative iteration count");
  if (c->vnn_from_scratch && c->ewc >= 0. && c->ewc_for_critic >= 1)
    return ddrl_fail(DDRL_EINVAL, "vnn_from_scratch with EWC on the critic: the reference's re-created critic has no EWC state (OfflineCaclaAg.hpp:281-284)");
  if (c->max_trajectory < 1 || c->max_trajectory > 32768) return ddrl_fail(DDRL_EINVAL, "max_trajectory outside 1..32768 (the lambda-return scan is LDS resident)");
  return DDRL_OK;
}

extern "C" int ddrl_nfac_create(const ddrl_nfac_config *cfg, uint64_t init_seed, ddrl_pop_t **out) {
  if (!cfg || !out) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  *out = nullptr;
  int rc = nfac_validate(cfg);
  if (rc) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return ddrl_fail(DDRL_EHIP, "no HIP device: libddrl_pop has no CPU path");
  ddrl_pop *h = new ddrl_pop();
  NfacHost *n = new NfacHost();
  h->algo = 1;
  h->nfac = n;
  n->cfg = *cfg;
  // the fields shared entry points read
  ddrl_ddpg_config_default(&h->cfg);
  h->cfg.n_agents = cfg->n_agents; h->cfg.nb_sensors = cfg->nb_sensors; h->cfg.nb_motors = cfg->nb_motors;
  h->cfg.gamma = cfg->gamma; h->cfg.alpha_a = cfg->alpha_a; h->cfg.alpha_v = cfg->alpha_v;
  h->cfg.ewc = cfg->ewc; h->cfg.ewc_decay = cfg->ewc_decay; h->cfg.ewc_fisher_beta = cfg->ewc_fisher_beta;
  h->cfg.ewc_best_param_method = cfg->ewc_best_param_method; h->cfg.ewc_for_critic = cfg->ewc_for_critic;
  h->cfg.momentum = cfg->momentum; h->cfg.mini_batch_size = 0; h->cfg.replay_memory = 0;
  h->G = cfg->n_agents;
  if (cfg->device >= 0) h->device = cfg->device;
  else if (hipGetDevice(&h->device) != hipSuccess) h->device = 0;
  auto bail = [&](int code) { ddrl_pop_destroy(h); return code; };
  if ((rc = set_device(h))) return bail(rc);
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return bail(ddrl_fail(DDRL_EHIP, "hipStreamCreate failed"));
  const int G = h->G, S = cfg->nb_sensors, A = cfg->nb_motors, maxT = cfg->max_trajectory;
  // OfflineCaclaAg.hpp:186 (actor) and :190 (V critic: input = sensors, one linear output)
  build_desc(h->net[0].nd, S, cfg->hidden_unit_a, cfg->n_hidden_a, A, cfg->hidden_layer_type, cfg->actor_output_layer_type);
  build_desc(h->net[1].nd, S, cfg->hidden_unit_v, cfg->n_hidden_v, 1, cfg->hidden_layer_type, 0);
  const bool ewc = cfg->ewc >= 0.;
  if ((rc = alloc_net(h, h->net[0], ewc, false))) return bail(rc);
  if ((rc = alloc_net(h, h->net[1], ewc && cfg->ewc_for_critic >= 1, false))) return bail(rc);
  n->rowlen = ddrl_roundup(2 * S + 2 * A + 2, 4);
  n->maxchunks = (maxT + DDRL_MB - 1) / DDRL_MB;
  if ((rc = dalloc(h, &n->d_traj, (size_t)G * maxT * n->rowlen))) return bail(rc);
  if ((rc = dalloc(h, &n->d_len, G))) return bail(rc);
  if ((rc = dalloc(h, &n->d_endflag, (size_t)G * maxT))) return bail(rc);
  if ((rc = dalloc(h, &n->d_V, (size_t)G * maxT))) return bail(rc);
  if ((rc = dalloc(h, &n->d_nextV, (size_t)G * maxT))) return bail(rc);
  if ((rc = dalloc(h, &n->d_tgt, (size_t)G * maxT))) return bail(rc);
  if ((rc = dalloc(h, &n->d_deltas, (size_t)G * maxT))) return bail(rc);
  if ((rc = dalloc(h, &n->d_rowmode, (size_t)G * maxT))) return bail(rc);
  if ((rc = dalloc(h, &n->d_npos, G))) return bail(rc);
  if ((rc = dalloc(h, &n->d_chunk_done, G))) return bail(rc);
  // zeroed once: the kernels never write the pad elements, so they stay 0
  if ((rc = dalloc(h, &n->d_gpart_a, (size_t)G * n->maxchunks * h->net[0].nd.stride()))) return bail(rc);
  if ((rc = dalloc(h, &n->d_gpart_v, (size_t)G * n->maxchunks * h->net[1].nd.stride()))) return bail(rc);
  if ((rc = dalloc(h, &n->d_loss_part, (size_t)G * n->maxchunks))) return bail(rc);
  if ((rc = dalloc(h, &n->d_loss, G))) return bail(rc);
  if ((rc = dalloc(h, &n->d_pa, (size_t)G * A))) return bail(rc);
  if ((rc = dalloc(h, &h->d_gamma, G))) return bail(rc);
  if ((rc = dalloc(h, &h->d_s, (size_t)G * S))) return bail(rc);
  if ((rc = dalloc(h, &h->d_s2, (size_t)G * S))) return bail(rc);
  if ((rc = dalloc(h, &h->d_a, (size_t)G * A))) return bail(rc);
  if ((rc = dalloc(h, &h->d_act, (size_t)G * A))) return bail(rc);
  if ((rc = dalloc(h, &h->d_r, G))) return bail(rc);
  if ((rc = dalloc(h, &h->d_cost, G))) return bail(rc);
  if ((rc = dalloc(h, &h->d_goal, G))) return bail(rc);
  if ((rc = dalloc(h, &h->d_mask, G))) return bail(rc);
  if ((rc = dalloc(h, &h->d_flag, G))) return bail(rc);
  n->h_len.assign(G, 0);
  n->h_ends.assign(G, {});

  std::vector<double> v(G);
  std::fill(v.begin(), v.end(), cfg->alpha_a); if ((rc = ddrl_pop_set_hyper(h, "alpha_a", v.data()))) return bail(rc);
  std::fill(v.begin(), v.end(), cfg->alpha_v); if ((rc = ddrl_pop_set_hyper(h, "alpha_v", v.data()))) return bail(rc);
  std::fill(v.begin(), v.end(), cfg->gamma); if ((rc = ddrl_pop_set_hyper(h, "gamma", v.data()))) return bail(rc);
  {
    std::mt19937_64 eng(init_seed);                       // gaussian(0, 0.01) fill (MLP.hpp:936-944)
    std::normal_distribution<double> nd01(0.0, 0.01);
    for (int k = 0; k < 2; ++k) {
      std::vector<double> flat((size_t)G * h->net[k].nd.np);
      for (auto &x : flat) x = nd01(eng);
      if ((rc = upload_flat(h, h->net[k].nd, h->net[k].st.w, flat.data()))) return bail(rc);
    }
  }
  double b1, b2;   // MLP.hpp:160-173
  switch (cfg->momentum) { case 0: b1 = 0.; b2 = 0.999; break; case 1: b1 = 0.9; b2 = 0.999; break;
                           case 2: b1 = 0.; b2 = 0.; break; default: b1 = 0.9; b2 = 0.; }
  h->beta1 = b1; h->beta2 = b2;
  NfacParams &P = n->P;
  memset(&P, 0, sizeof(P));
  P.a = h->net[0].nd; P.v = h->net[1].nd;
  P.sa = h->net[0].st; P.sv = h->net[1].st;
  P.G = G; P.S = S; P.A = A; P.maxT = maxT; P.rowlen = n->rowlen; P.maxchunks = n->maxchunks;
  P.ldh = 4;
  for (int k = 0; k < 2; ++k)
    for (int l = 1; l < h->net[k].nd.L(); ++l) P.ldh = std::max(P.ldh, h->net[k].nd.ld(l));
  P.inverting = cfg->inverting_gradient ? 1 : 0;
  P.corrected = cfg->corrected_update_ac ? 1 : 0;
  P.corr_factor = (float)cfg->corrected_update_ac_factor;
  P.value_passes = 3;
  n->scratch_eng.seed(init_seed ^ 0x5c7a7c45eedull);
  if (cfg->vnn_from_scratch && (rc = dalloc(h, &n->d_fresh, (size_t)G * h->net[1].nd.stride()))) return bail(rc);
  P.update_delta_neg = cfg->update_delta_neg ? 1 : 0;
  P.gae = cfg->gae ? 1 : 0;
  P.lambda = (float)cfg->lambda;
  P.beta1 = (float)b1; P.beta2 = (float)b2; P.omb1 = (float)(1. - b1); P.omb2 = (float)(1. - b2);
  P.eps = 1e-8f;
  P.fisher_beta = (float)cfg->ewc_fisher_beta;
  P.dbeta1 = b1; P.dbeta2 = b2;
  P.traj = n->d_traj; P.len = n->d_len; P.endflag = n->d_endflag; P.gamma = h->d_gamma;
  P.V = n->d_V; P.nextV = n->d_nextV; P.tgt = n->d_tgt; P.deltas = n->d_deltas; P.rowmode = n->d_rowmode;
  P.chunk_done = n->d_chunk_done;
  P.n_pos = n->d_npos; P.gpart_a = n->d_gpart_a; P.gpart_v = n->d_gpart_v; P.loss_part = n->d_loss_part; P.loss_out = n->d_loss;
  if (nfac_lds_need(&P) > 160 * 1024) return bail(ddrl_fail(DDRL_EINVAL, "network too wide for the LDS-resident kernels (%zu bytes of LDS)", nfac_lds_need(&P)));
  if (hipStreamSynchronize(h->stream) != hipSuccess) return bail(ddrl_fail(DDRL_EHIP, "sync failed"));
  *out = h;
  return DDRL_OK;
}

#define NFAC_ONLY(h) do { if (!(h) || (h)->algo != 1 || !(h)->nfac) return ddrl_fail(DDRL_EINVAL, "%s: not an NFAC population", __func__); } while (0)

extern "C" int ddrl_nfac_traj_append(ddrl_pop_t *h, const float *s, const float *pure_a, const float *a,
                                     const float *s2, const float *r, const uint8_t *goal, const uint8_t *mask) {
  NFAC_ONLY(h);
  if (!s || !pure_a || !a || !s2 || !r || !goal) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  int rc = set_device(h);
  if (rc) return rc;
  NfacHost *n = h->nfac;
  const int G = h->G, S = n->cfg.nb_sensors, A = n->cfg.nb_motors, rl = n->rowlen;
  for (int g = 0; g < G; ++g)
    if ((!mask || mask[g]) && n->h_len[g] >= n->cfg.max_trajectory)
      return ddrl_fail(DDRL_ESTATE, "agent %d: trajectory buffer full (max_trajectory = %d)", g, n->cfg.max_trajectory);
  // one strided copy per agent row: rows are short, the call is per environment step
  std::vector<float> row(rl, 0.f);
  for (int g = 0; g < G; ++g) {
    if (mask && !mask[g]) continue;
    memcpy(&row[0], s + (size_t)g * S, S * 4);
    memcpy(&row[S], pure_a + (size_t)g * A, A * 4);
    memcpy(&row[S + A], a + (size_t)g * A, A * 4);
    memcpy(&row[S + 2 * A], s2 + (size_t)g * S, S * 4);
    row[2 * S + 2 * A] = r[g];
    row[2 * S + 2 * A + 1] = goal[g] ? 1.f : 0.f;
    HIPCHK(hipMemcpy(n->d_traj + ((size_t)g * n->cfg.max_trajectory + n->h_len[g]) * rl, row.data(), rl * 4, hipMemcpyHostToDevice));
    n->h_len[g]++;
  }
  return DDRL_OK;
}

extern "C" int ddrl_nfac_traj_load(ddrl_pop_t *h, int agent, int cnt, const float *rows, int n_ends, const int32_t *ends) {
  NFAC_ONLY(h);
  NfacHost *n = h->nfac;
  if (agent < 0 || agent >= h->G || cnt < 0 || cnt > n->cfg.max_trajectory || (cnt > 0 && !rows) || n_ends < 0 || (n_ends > 0 && !ends))
    return ddrl_fail(DDRL_EINVAL, "bad argument");
  int rc = set_device(h);
  if (rc) return rc;
  const int S = n->cfg.nb_sensors, A = n->cfg.nb_motors, rl = 2 * S + 2 * A + 2;
  std::vector<float> img((size_t)cnt * n->rowlen, 0.f);
  for (int i = 0; i < cnt; ++i) memcpy(&img[(size_t)i * n->rowlen], rows + (size_t)i * rl, rl * sizeof(float));
  if (cnt)
    HIPCHK(hipMemcpy(n->d_traj + (size_t)agent * n->cfg.max_trajectory * n->rowlen, img.data(), img.size() * 4, hipMemcpyHostToDevice));
  n->h_len[agent] = cnt;
  n->h_ends[agent].assign(ends, ends + n_ends);
  return DDRL_OK;
}

extern "C" int ddrl_nfac_episode_end(ddrl_pop_t *h, const uint8_t *mask) {
  NFAC_ONLY(h);
  NfacHost *n = h->nfac;
  for (int g = 0; g < h->G; ++g)
    if (!mask || mask[g]) n->h_ends[g].push_back(n->h_len[g]);          // OfflineCaclaAg.hpp:419
  return DDRL_OK;
}

extern "C" int ddrl_nfac_traj_len(ddrl_pop_t *h, int32_t *len) {
  NFAC_ONLY(h);
  if (!len) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  for (int g = 0; g < h->G; ++g) len[g] = h->nfac->h_len[g];
  return DDRL_OK;
}

extern "C" int ddrl_nfac_max_trajectory(ddrl_pop_t *h) {
  if (!h || h->algo != 1 || !h->nfac) return ddrl_fail(DDRL_EINVAL, "not an NFAC population");
  return h->nfac->cfg.max_trajectory;
}

// update_critic (OfflineCaclaAg.hpp:242-401): number_fitted_iteration x {targets, stoch_iter_critic steps}
static int launch_critic(ddrl_pop *h, NfacHost *n, int chunks, int maxlen, int &steps) {
  const ddrl_nfac_config &c = n->cfg;
  for (int fit = 0; fit < c.number_fitted_iteration; ++fit) {
    if (c.vnn_from_scratch) {
      // :266-284: the targets come from the OLD critic's V(s'); then a new net (weights ~N(0, 0.01^2), MLP.hpp:936-944,
      // new solver: zero moments, iter 0) whose V(s) enters the lambda return (:315)
      n->P.value_passes = 2;
      n->P.gm = next_draw(h, 1);
      HIPCHK(nfac_values_launch(&n->P, chunks, h->stream));
      std::normal_distribution<double> nd01(0.0, 0.01);
      std::vector<double> flat((size_t)h->G * h->net[1].nd.np);
      for (auto &x : flat) x = nd01(n->scratch_eng);
      int rc = upload_flat(h, h->net[1].nd, n->d_fresh, flat.data());
      if (rc) return rc;
      HIPCHK(nfac_reinit_launch(&n->P, n->d_fresh, h->stream));
      n->P.value_passes = 1;
      n->P.gm = next_draw(h, 1);
      HIPCHK(nfac_values_launch(&n->P, chunks, h->stream));
      n->P.value_passes = 3;
      steps = 0;
    } else {
      n->P.gm = next_draw(h, 1);
      HIPCHK(nfac_values_launch(&n->P, chunks, h->stream));
    }
    HIPCHK(nfac_scan_launch(&n->P, 0, maxlen, h->stream));
    for (int it = 0; it < c.stoch_iter_critic; ++it) {
      n->P.gm = next_draw(h, 1);
      hipEvent_t pe;                                   // profiling on: the critic's gradient launches are the ones timed
      int prc = prof_begin(h, &pe);
      if (prc) return prc;
      const int last = fit == c.number_fitted_iteration - 1 && it == c.stoch_iter_critic - 1;
      HIPCHK(nfac_grad_launch(&n->P, 0, chunks, steps, last, h->stream));
      if ((prc = prof_end(h, pe))) return prc;
      if (!n->P.fuse) HIPCHK(nfac_apply_launch(&n->P, 0, steps, last, h->stream));
      ++steps;
    }
  }
  return DDRL_OK;
}

extern "C" int ddrl_nfac_update(ddrl_pop_t *h, const uint8_t *mask) {
  NFAC_ONLY(h);
  int rc = set_device(h);
  if (rc) return rc;
  NfacHost *n = h->nfac;
  const ddrl_nfac_config &c = n->cfg;
  const int G = h->G, maxT = c.max_trajectory;
  const uint8_t *dmask;
  if ((rc = upload_mask(h, mask, &dmask))) return rc;
  // episode-end flags exactly as the reference's scan consumes trajectory_end_points (:329-336):
  // only the current last end point is tested, so duplicates shadow earlier ones
  std::vector<uint8_t> flags((size_t)G * maxT, 0);
  int maxlen = 0;
  for (int g = 0; g < G; ++g) {
    if (mask && !mask[g]) continue;
    const int T = n->h_len[g];
    maxlen = std::max(maxlen, T);
    const std::vector<int> &ends = n->h_ends[g];
    int index_ep = (int)ends.size() - 1;
    for (int li = T - 1; li >= 0; --li)
      if (index_ep >= 0 && ends[index_ep] - 1 == li) { flags[(size_t)g * maxT + li] = 1; index_ep--; }
  }
  HIPCHK(hipMemcpyAsync(n->d_endflag, flags.data(), flags.size(), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(n->d_len, n->h_len.data(), G * sizeof(int), hipMemcpyHostToDevice, h->stream));
  n->P.mask = dmask;
  n->P.lw = nullptr;
  n->P.generic = h->force_generic ? 1 : 0;
  n->P.fuse = h->nfac_fuse ? 1 : 0;
  n->P.sgate = h->gates_on ? h->d_sgate : nullptr;
  n->P.agate = h->gates_on ? h->d_agate : nullptr;
  n->P.sa = h->net[0].st;
  n->P.sv = h->net[1].st;
  const int chunks = (maxlen + DDRL_MB - 1) / DDRL_MB;
  int critic_steps = 0;
  if (chunks > 0) {
    if (c.update_critic_first && (rc = launch_critic(h, n, chunks, maxlen, critic_steps))) return rc;   // :430-431
    n->P.gm = next_draw(h, 1);
    HIPCHK(nfac_values_launch(&n->P, chunks, h->stream));                                               // :460-461
    HIPCHK(nfac_scan_launch(&n->P, 1, maxlen, h->stream));                                              // :464-522
    for (int sia = 0; sia < c.stoch_iter_actor; ++sia) {                                                // :527-572
      n->P.gm = next_draw(h, 1);
      HIPCHK(nfac_grad_launch(&n->P, 1, chunks, sia, 0, h->stream));
      if (!n->P.fuse) HIPCHK(nfac_apply_launch(&n->P, 1, sia, 0, h->stream));
    }
    if (!c.update_critic_first && (rc = launch_critic(h, n, chunks, maxlen, critic_steps))) return rc;  // :585-587
    HIPCHK(nfac_finish_launch(&n->P, critic_steps, c.stoch_iter_actor, h->stream));
  }
  HIPCHK(hipStreamSynchronize(h->stream));       // flags / mask were read asynchronously
  for (int g = 0; g < G; ++g) {
    if (mask && !mask[g]) continue;
    n->h_len[g] = 0;                             // trajectory.clear(); trajectory_end_points.clear() (:590-591)
    n->h_ends[g].clear();
  }
  return ddrl_pop_ewc_decay_update(h, mask);     // :593-594
}

// MLP::learn_batch_lw (MLP.hpp:487-506) on the critic of every agent with mask[g] != 0: `iters` full-batch Adam steps on the n
// rows (x, y) with label weights lw (null: learn_batch, MLP.hpp:468-485), EWC terms included when live.  The rows pass
// through the agents' trajectory buffers, so nothing may be waiting there.
extern "C" int ddrl_nfac_learn_batch_lw(ddrl_pop_t *h, const float *x, const float *y, const float *lw, int cnt, int iters, const uint8_t *mask) {
  NFAC_ONLY(h);
  NfacHost *n = h->nfac;
  const ddrl_nfac_config &c = n->cfg;
  const int G = h->G, maxT = c.max_trajectory, S = c.nb_sensors;
  if (!x || !y || cnt <= 0 || cnt > maxT || iters < 0) return ddrl_fail(DDRL_EINVAL, "bad argument (1 <= n <= max_trajectory = %d)", maxT);
  if (c.vnn_from_scratch) return ddrl_fail(DDRL_EINVAL, "learn_batch_lw with vnn_from_scratch is not supported");
  for (int g = 0; g < G; ++g)
    if ((!mask || mask[g]) && n->h_len[g] != 0)
      return ddrl_fail(DDRL_ESTATE, "agent %d holds %d gathered transitions: learn_batch_lw would overwrite them", g, n->h_len[g]);
  int rc = set_device(h);
  if (rc) return rc;
  const uint8_t *dmask;
  if ((rc = upload_mask(h, mask, &dmask))) return rc;
  if (!n->d_lw && (rc = dalloc(h, &n->d_lw, (size_t)G * maxT))) return rc;
  std::vector<float> img((size_t)cnt * n->rowlen), col(cnt);
  std::vector<int> len(G, 0);
  for (int g = 0; g < G; ++g) {
    if (mask && !mask[g]) continue;
    len[g] = cnt;
    std::fill(img.begin(), img.end(), 0.f);
    for (int i = 0; i < cnt; ++i) memcpy(&img[(size_t)i * n->rowlen], x + ((size_t)g * cnt + i) * S, S * sizeof(float));
    HIPCHK(hipMemcpy(n->d_traj + (size_t)g * maxT * n->rowlen, img.data(), img.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(n->d_tgt + (size_t)g * maxT, y + (size_t)g * cnt, (size_t)cnt * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < cnt; ++i) col[i] = lw ? lw[(size_t)g * cnt + i] : 1.f;
    HIPCHK(hipMemcpy(n->d_lw + (size_t)g * maxT, col.data(), (size_t)cnt * 4, hipMemcpyHostToDevice));
  }
  HIPCHK(hipMemcpyAsync(n->d_len, len.data(), G * sizeof(int), hipMemcpyHostToDevice, h->stream));
  n->P.mask = dmask;
  n->P.lw = lw ? n->d_lw : nullptr;
  n->P.generic = h->force_generic ? 1 : 0;
  n->P.fuse = 0;
  n->P.sgate = h->gates_on ? h->d_sgate : nullptr;
  n->P.agate = h->gates_on ? h->d_agate : nullptr;
  n->P.sa = h->net[0].st;
  n->P.sv = h->net[1].st;
  const int chunks = (cnt + DDRL_MB - 1) / DDRL_MB;
  for (int it = 0; it < iters; ++it) {
    n->P.gm = next_draw(h, 1);
    HIPCHK(nfac_grad_launch(&n->P, 0, chunks, it, it == iters - 1, h->stream));
    HIPCHK(nfac_apply_launch(&n->P, 0, it, it == iters - 1, h->stream));
  }
  if (iters > 0) HIPCHK(nfac_finish_launch(&n->P, iters, 0, h->stream));      // Solver::iter += iters (the critic's solver only)
  n->P.lw = nullptr;
  HIPCHK(hipStreamSynchronize(h->stream));       // len / mask were read asynchronously
  return DDRL_OK;
}

extern "C" int ddrl_nfac_last(ddrl_pop_t *h, float *deltas, int32_t *n_pos, float *critic_loss) {
  NFAC_ONLY(h);
  int rc = set_device(h);
  if (rc) return rc;
  NfacHost *n = h->nfac;
  const size_t G = h->G;
  if (deltas) HIPCHK(hipMemcpyAsync(deltas, n->d_deltas, G * n->cfg.max_trajectory * 4, hipMemcpyDeviceToHost, h->stream));
  if (n_pos) HIPCHK(hipMemcpyAsync(n_pos, n->d_npos, G * 4, hipMemcpyDeviceToHost, h->stream));
  if (critic_loss) HIPCHK(hipMemcpyAsync(critic_loss, n->d_loss, G * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return DDRL_OK;
}

// the trajectories gathered since the last update (OfflineCaclaAg keeps them in `trajectory` /
// `trajectory_end_points`; its save_run writes none of it, a population resumed mid-batch needs them)
int ddrl_nfac_checkpoint(ddrl_pop *h, Ckpt &c) {
  NfacHost *n = h->nfac;
  const size_t G = h->G, maxT = n->cfg.max_trajectory;
  int rc;
  const int meta[2] = {n->cfg.max_trajectory, n->rowlen};
  if ((rc = c.same("nfac.meta", meta, sizeof(meta)))) return rc;
  if ((rc = c.dev("nfac.traj", n->d_traj, G * maxT * n->rowlen * sizeof(float))) || (rc = c.vec("nfac.len", n->h_len))) return rc;
  if (n->h_len.size() != G) return ddrl_fail(DDRL_EIO, "%s: trajectory bookkeeping of the wrong length", c.path.c_str());
  std::vector<int> flat;                                 // [count, ends...] per agent
  if (c.writing)
    for (size_t g = 0; g < G; ++g) { flat.push_back((int)n->h_ends[g].size()); flat.insert(flat.end(), n->h_ends[g].begin(), n->h_ends[g].end()); }
  if ((rc = c.vec("nfac.ends", flat))) return rc;
  if (n->cfg.vnn_from_scratch) {                          // the stream the next re-created critics are drawn from
    std::vector<char> es;
    if (c.writing) { std::ostringstream os; os << n->scratch_eng; const std::string t = os.str(); es.assign(t.begin(), t.end()); }
    if ((rc = c.vec("nfac.scratch_eng", es))) return rc;
    if (!c.writing) { std::istringstream is(std::string(es.begin(), es.end())); is >> n->scratch_eng; }
  }
  if (!c.writing) {
    size_t p = 0;
    for (size_t g = 0; g < G; ++g) {
      if (p >= flat.size() || flat[p] < 0 || p + 1 + (size_t)flat[p] > flat.size()) return ddrl_fail(DDRL_EIO, "%s: broken trajectory end points", c.path.c_str());
      n->h_ends[g].assign(flat.begin() + p + 1, flat.begin() + p + 1 + flat[p]);
      p += 1 + (size_t)flat[p];
    }
  }
  return DDRL_OK;
}
