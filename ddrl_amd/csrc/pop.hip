// pop.hip -- host side of libddrl_pop.so: handle, device memory, the C ABI of include/ddrl_pop.h.
// All arithmetic of the hot path runs in the gfx950 kernels (kernels_*.hip); there is no CPU
// fallback: without a HIP device every create call fails with DDRL_EHIP.
#include "pop_host.h"

extern "C" {
hipError_t ddpg_update_launch(const DdpgParams *P, int force_generic, hipStream_t st);
size_t ddpg_update_lds_bytes(const DdpgParams *P);
hipError_t ddpg_update_pair_launch(const DdpgParams *P, hipStream_t st);
int ddpg_update_pair_ok(const DdpgParams *P);
hipError_t ddpg_update_resident_launch(const DdpgParams *P, hipStream_t st);
int ddpg_update_resident_ok(const DdpgParams *P);
hipError_t ddpg_update_big_launch(const DdpgParams *P, hipStream_t st);
hipError_t ddpg_update_wide_launch(const DdpgParams *P, hipStream_t st);
size_t ddpg_wide_block_floats(const DdpgParams *P);
hipError_t replay_draw_launch(uint32_t *mt, int *pos, const int *rep_len, int G, int B, int nupd, const uint8_t *mask,
                              int *idx_out, int draw, double beta1, double beta2, const float *lr_a,
                              const float *lr_q, const int *iter_a, const int *iter_q, float *lrt_a, float *lrt_q,
                              int *upd_done, hipStream_t st);
hipError_t td3_update_launch(const DdpgParams *P, int force_generic, hipStream_t st);
hipError_t td3_update_big_launch(const DdpgParams *P, int any_actor, hipStream_t st);
hipError_t td3_update_wide_launch(const DdpgParams *P, int any_actor, hipStream_t st);
hipError_t td3_update_bn_launch(const DdpgParams *P, int any_actor, hipStream_t st);
hipError_t td3_prep_launch(const uint64_t *seed, uint32_t *count, const int *rep_len, int G, int B, int A,
                           const uint8_t *mask, float *noise, int draw, float sigma, float clip, double beta1,
                           double beta2, const float *lr_q2, const int *iter_q2, float *lrt_q2, hipStream_t st);
hipError_t mt_seed_launch(uint32_t *mt, int *pos, const uint32_t *seeds, int G, hipStream_t st);
hipError_t replay_append_launch(float *rep, int *len, int *head, int G, int cap, int rowlen, int S, int A,
                                const float *s, const float *a, const float *r, const float *s2,
                                const uint8_t *goal, const uint8_t *mask, hipStream_t st);
int ddpg_update_pair_pays(const DdpgParams *P);
int ddpg_update_loop_ok(const DdpgParams *P);
int ddpg_update_loop_pays(const DdpgParams *P);
hipError_t ddpg_update_loop_launch(const DdpgParams *P, hipStream_t st);
int ddpg_update_loop8_ok(const DdpgParams *P);
hipError_t ddpg_update_loop8_launch(const DdpgParams *P, hipStream_t st);
hipError_t ddpg_update_bn_launch(const DdpgParams *P, hipStream_t st);
size_t ddpg_update_bn_lds_bytes(const DdpgParams *P);
hipError_t net_forward_bn_launch(const NetDesc *nd, const float *w, const float *X, float *out, int n, int G,
                                 const float *sgate, int S, const float *agate, int A, int gate_out, const GateMode *gm, hipStream_t st);
hipError_t infer_launch(const NetDesc *nd, const float *w, const float *states, float *actions,
                        const float *in_gate, const float *out_gate, const GateMode *gm, int G, hipStream_t st);
hipError_t net_forward_launch(const NetDesc *nd, const float *w, const float *X, float *out, int n, int G,
                              const float *sgate, int S, const float *agate, int A, int gate_out, const GateMode *gm, hipStream_t st);
size_t net_forward_wide_scratch_floats(const NetDesc *nd, int n, int G);
hipError_t net_forward_wide_launch(const NetDesc *nd, const float *w, const float *X, float *out, float *scratch, int n, int G,
                                   const float *sgate, int S, const float *agate, int A, int gate_out, const GateMode *gm, hipStream_t st);
hipError_t snapshot_launch(float *dst, const float *src, int stride, const uint8_t *flag, int G, hipStream_t st);
hipError_t load_flat_launch(const NetDesc *nd, float *w, const float *flat, int ld, const uint8_t *flag, int G, hipStream_t st);
hipError_t ewc_setup_launch(const NetDesc *nd, float *w, const float *best, float *wstar, float *fisher, float *fhat, float *nbr,
                            int stride, const uint8_t *flag, int G, hipStream_t st);
hipError_t ewc_cost_launch(const float *w, const float *wstar, const float *fhat, int stride, float *out, int G,
                           hipStream_t st);
}

// ---------------------------------------------------------------------------------------------
#ifdef DDRL_F64_DEBUG
extern "C" const char *ddrl_version(void) { return "ddrl_amd 0.2 (gfx950) +f64 debug build: the one-tile DDPG path in fp64, float arguments of the ABI are double"; }
#else
extern "C" const char *ddrl_version(void) { return "ddrl_amd 0.2 (gfx950)"; }
#endif

extern "C" int ddrl_ddpg_config_default(ddrl_ddpg_config *c) {
  if (!c) return ddrl_fail(DDRL_EINVAL, "cfg is NULL");
  memset(c, 0, sizeof(*c));
  c->n_agents = 1; c->nb_sensors = 18; c->nb_motors = 6;
  c->n_hidden_a = 2; c->hidden_unit_a[0] = 100; c->hidden_unit_a[1] = 100;
  c->n_hidden_q = 2; c->hidden_unit_q[0] = 100; c->hidden_unit_q[1] = 100;
  c->mini_batch_size = 64; c->replay_memory = 10000; c->inverting_grad = 1; c->force_more_update = 0;
  c->hidden_layer_type = 1; c->actor_output_layer_type = 2; c->momentum = 0; c->nstep_return = 0;
  c->gamma = 0.99; c->tau_soft_update = 0.001; c->alpha_a = 1e-4; c->alpha_v = 1e-3; c->decay_v = -1.;
  c->ewc = -1.; c->ewc_decay = -1.; c->ewc_fisher_beta = 0.999; c->ewc_best_param_method = 0; c->ewc_for_critic = 0;
  c->device = -1;
  c->td3 = 0; c->policy_freq = 2; c->policy_noise = 0.2; c->noise_clip = 0.5;   // (TD3 paper defaults; the reference ships no td3 ini)
  return DDRL_OK;
}

static int config_from_ini(const char *path, ddrl_ddpg_config *c, bool td3);
extern "C" int ddrl_ddpg_config_from_ini(const char *path, ddrl_ddpg_config *c) { return config_from_ini(path, c, false); }
extern "C" int ddrl_td3_config_from_ini(const char *path, ddrl_ddpg_config *c) { return config_from_ini(path, c, true); }

static int config_from_ini(const char *path, ddrl_ddpg_config *c, bool td3) {
  if (!path || !c) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  DdrlIni ini;
  std::string err;
  if (!ini.load(path, &err)) return ddrl_fail(DDRL_EIO, "%s", err.c_str());
  const int G = c->n_agents, S = c->nb_sensors, A = c->nb_motors, dev = c->device;
  ddrl_ddpg_config_default(c);
  c->n_agents = G > 0 ? G : 1;
  if (S > 0) c->nb_sensors = S;
  if (A > 0) c->nb_motors = A;
  c->device = dev;
  try {
    std::vector<long> hq = ini.get_list("agent.hidden_unit_q"), ha = ini.get_list("agent.hidden_unit_a");
    if (hq.empty() || ha.empty() || hq.size() > DDRL_MAX_HIDDEN || ha.size() > DDRL_MAX_HIDDEN)
      return ddrl_fail(DDRL_EINVAL, "hidden_unit_q/a: 1..%d layers supported", DDRL_MAX_HIDDEN);
    c->n_hidden_q = (int)hq.size(); for (size_t i = 0; i < hq.size(); ++i) c->hidden_unit_q[i] = (int)hq[i];
    c->n_hidden_a = (int)ha.size(); for (size_t i = 0; i < ha.size(); ++i) c->hidden_unit_a[i] = (int)ha[i];
    c->gamma = ini.get_double("agent.gamma");
    c->mini_batch_size = (int)ini.get_long("agent.mini_batch_size");
    c->replay_memory = (int)ini.get_long("agent.replay_memory");
    if (td3) {                                     // TD3Ag.hpp:165-186 reads neither of the three DDPG-only keys
      c->td3 = 1; c->inverting_grad = 0; c->force_more_update = 0;
      c->policy_noise = ini.get_double("agent.policy_noise");
      c->noise_clip = ini.get_double("agent.noise_clip");
      c->policy_freq = (int)ini.get_long("agent.policy_freq");
    } else {
      c->inverting_grad = ini.get_bool("agent.inverting_grad");
      c->force_more_update = (int)ini.get_long("agent.force_more_update");
    }
    c->tau_soft_update = ini.get_double("agent.tau_soft_update");
    c->alpha_a = ini.get_double("agent.alpha_a");
    c->alpha_v = ini.get_double("agent.alpha_v");
    c->decay_v = ini.get_double("agent.decay_v");
    c->batch_norm_critic = (int)ini.get_long("agent.batch_norm_critic");
    c->batch_norm_actor = (int)ini.get_long("agent.batch_norm_actor");
    c->actor_output_layer_type = (int)ini.get_long("agent.actor_output_layer_type");
    c->hidden_layer_type = (int)ini.get_long("agent.hidden_layer_type");
    c->momentum = (int)ini.get_long("agent.momentum");
    if (!td3) c->nstep_return = (int)ini.get_long("agent.nstep_return");
    // DeepQNAg.hpp:184-185 reads both unconditionally; they only matter with towers, and the tuned ini the tree ships
    // (agent/deepqn/data/best-ddpg-halfcheetah.ini) predates bn_adapt: absent = false
    if (ini.has("agent.test_net")) c->test_net = ini.get_bool("agent.test_net");
    if (ini.has("agent.bn_adapt")) c->bn_adapt = ini.get_bool("agent.bn_adapt");
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

static int validate(const ddrl_ddpg_config *c) {
  if (c->n_agents < 1) return ddrl_fail(DDRL_EINVAL, "n_agents < 1");
  if (c->nb_sensors < 1 || c->nb_motors < 1) return ddrl_fail(DDRL_EINVAL, "nb_sensors/nb_motors < 1");
  if (c->nb_sensors + c->nb_motors > DDRL_WIDE_MAXW) return ddrl_fail(DDRL_EINVAL, "nb_sensors + nb_motors > %d", DDRL_WIDE_MAXW);
  if (c->n_hidden_a < 1 || c->n_hidden_a > DDRL_MAX_HIDDEN || c->n_hidden_q < 1 || c->n_hidden_q > DDRL_MAX_HIDDEN)
    return ddrl_fail(DDRL_EINVAL, "1..%d hidden layers supported", DDRL_MAX_HIDDEN);
  for (int i = 0; i < c->n_hidden_a; ++i)
    if (c->hidden_unit_a[i] < 1 || c->hidden_unit_a[i] > DDRL_WIDE_MAXW) return ddrl_fail(DDRL_EINVAL, "hidden_unit_a[%d]=%d outside 1..%d", i, c->hidden_unit_a[i], DDRL_WIDE_MAXW);
  for (int i = 0; i < c->n_hidden_q; ++i)
    if (c->hidden_unit_q[i] < 1 || c->hidden_unit_q[i] > DDRL_WIDE_MAXW) return ddrl_fail(DDRL_EINVAL, "hidden_unit_q[%d]=%d outside 1..%d", i, c->hidden_unit_q[i], DDRL_WIDE_MAXW);
  if (c->mini_batch_size < 1 || c->mini_batch_size > DDRL_MAXB) return ddrl_fail(DDRL_EINVAL, "mini_batch_size outside 1..%d", DDRL_MAXB);
  if (c->replay_memory < c->mini_batch_size) return ddrl_fail(DDRL_EINVAL, "replay_memory < mini_batch_size");
  if (c->hidden_layer_type < 1 || c->hidden_layer_type > 3) return ddrl_fail(DDRL_EINVAL, "hidden_layer_type not in (1,2,3)");
  if (c->actor_output_layer_type < 0 || c->actor_output_layer_type > 3) return ddrl_fail(DDRL_EINVAL, "actor_output_layer_type not in 0..3");
  if (c->momentum < 0 || c->momentum > 3) return ddrl_fail(DDRL_EINVAL, "momentum not in 0..3");
  if (c->force_more_update < 0 || c->force_more_update >= DDRL_MAXUPD) return ddrl_fail(DDRL_EINVAL, "force_more_update outside 0..%d", DDRL_MAXUPD - 1);
  if (c->batch_norm_actor || c->batch_norm_critic) {
    for (int code : {c->batch_norm_actor, c->batch_norm_critic}) {
      if (code < 0 || code > 31) return ddrl_fail(DDRL_EINVAL, "batch_norm %d outside the 5-bit field of MLP.hpp:53-79", code);
      if ((code & 1) && !(code & 2)) return ddrl_fail(DDRL_EINVAL, "batch_norm %d: try to scale with bias without scaling", code);   // :71-72
      if (!(code & 28) && (code & 3)) return ddrl_fail(DDRL_EINVAL, "batch_norm %d: try to scale without bn", code);                 // :73-76
    }
    if ((c->batch_norm_actor & 16) && !(c->batch_norm_actor & 4))
      return ddrl_fail(DDRL_EINVAL, "batch_norm_actor %d: first_except_action is a critic layout", c->batch_norm_actor);
    if (c->mini_batch_size > DDRL_MB) return ddrl_fail(DDRL_EINVAL, "batch_norm != 0 needs mini_batch_size <= %d (the towers couple the rows of a minibatch)", DDRL_MB);
    for (int i = 0; i < c->n_hidden_a; ++i) if (c->hidden_unit_a[i] > DDRL_MAXW) return ddrl_fail(DDRL_EINVAL, "batch_norm != 0 needs layers of at most %d units", DDRL_MAXW);
    for (int i = 0; i < c->n_hidden_q; ++i) if (c->hidden_unit_q[i] > DDRL_MAXW) return ddrl_fail(DDRL_EINVAL, "batch_norm != 0 needs layers of at most %d units", DDRL_MAXW);
    if (c->nb_sensors + c->nb_motors > DDRL_MAXW) return ddrl_fail(DDRL_EINVAL, "batch_norm != 0 needs nb_sensors + nb_motors <= %d", DDRL_MAXW);
  }
  if (!(c->alpha_a > 0) || !(c->alpha_v > 0)) return ddrl_fail(DDRL_EINVAL, "alpha <= 0");
  if (c->td3) {
    if (c->policy_freq < 1) return ddrl_fail(DDRL_EINVAL, "policy_freq < 1");
    if (!(c->policy_noise >= 0.) || !(c->noise_clip >= 0.)) return ddrl_fail(DDRL_EINVAL, "policy_noise / noise_clip < 0");
    if (c->nstep_return > 0 || c->inverting_grad || c->force_more_update)
      return ddrl_fail(DDRL_EINVAL, "TD3Ag has no nstep_return / inverting_grad / force_more_update");
  }
  return DDRL_OK;
}

// per-chunk partial gradients (minibatches of more than one 64-row tile, and every wide update) and
// the wide path's activation scratch; allocated once, when first needed
static int chunk_buffers(ddrl_pop *h) {
  DdpgParams &P = h->P;
  const size_t G = h->G, C = P.maxchunks;
  int rc;
  if ((C > 1 || h->wide || h->force_chunked || h->bn) && !P.gpart_a) {
    if ((rc = dalloc(h, &P.gpart_a, G * C * h->net[0].nd.stride()))) return rc;
    if ((rc = dalloc(h, &P.gpart_q, G * C * h->net[1].nd.stride()))) return rc;
    if ((rc = dalloc(h, &P.loss_part, G * C))) return rc;
    if (h->nnets == 3) {
      if ((rc = dalloc(h, &P.gpart_q2, G * C * h->net[2].nd.stride()))) return rc;
      if ((rc = dalloc(h, &P.loss_part2, G * C))) return rc;
    }
  }
  if (h->wide && !P.wide) {
    P.wide_stride = ddpg_wide_block_floats(&P);
    if ((rc = dalloc(h, &P.wide, G * C * P.wide_stride))) return rc;
  }
  return DDRL_OK;
}

extern "C" int ddrl_ddpg_create(const ddrl_ddpg_config *cfg, uint64_t init_seed, ddrl_pop_t **out) {
  if (!cfg || !out) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  *out = nullptr;
  int rc = validate(cfg);
  if (rc) return rc;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return ddrl_fail(DDRL_EHIP, "no HIP device: libddrl_pop has no CPU path");
  ddrl_pop *h = new ddrl_pop();
  h->cfg = *cfg;
  h->G = cfg->n_agents;
  if (cfg->device >= 0) h->device = cfg->device;
  else if (hipGetDevice(&h->device) != hipSuccess) h->device = 0;
  auto bail = [&](int code) { ddrl_pop_destroy(h); return code; };
  if ((rc = set_device(h))) return bail(rc);
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return bail(ddrl_fail(DDRL_EHIP, "hipStreamCreate failed"));
  const int G = h->G, S = cfg->nb_sensors, A = cfg->nb_motors, B = cfg->mini_batch_size;
  // DeepQNAg.hpp:209-210 (actor) and :224-225 (critic: input = sensors + motors, 1 output, linear)
  build_desc(h->net[0].nd, S, cfg->hidden_unit_a, cfg->n_hidden_a, A, cfg->hidden_layer_type, cfg->actor_output_layer_type,
             cfg->batch_norm_actor, 0);
  build_desc(h->net[1].nd, S + A, cfg->hidden_unit_q, cfg->n_hidden_q, 1, cfg->hidden_layer_type, 0, cfg->batch_norm_critic, S);
  h->bn = cfg->batch_norm_actor != 0 || cfg->batch_norm_critic != 0;
  const bool ewc = cfg->ewc >= 0.;
  if ((rc = alloc_net(h, h->net[0], ewc))) return bail(rc);
  if ((rc = alloc_net(h, h->net[1], ewc && cfg->ewc_for_critic >= 1))) return bail(rc);   // DODevMLP.hpp:714-720
  if (cfg->td3) {                                          // TD3Ag.hpp:221-229: qnn2 like qnn
    h->nnets = 3;
    h->net[2].nd = h->net[1].nd;
    if ((rc = alloc_net(h, h->net[2], ewc && cfg->ewc_for_critic >= 1))) return bail(rc);
    if ((rc = dalloc(h, &h->d_nseed, G))) return bail(rc);
    if ((rc = dalloc(h, &h->d_ncount, G))) return bail(rc);
    if ((rc = dalloc(h, &h->d_noise, (size_t)G * B * A))) return bail(rc);
    if ((rc = dalloc(h, &h->d_loss2, G))) return bail(rc);
    if ((rc = dalloc(h, &h->d_astep, G))) return bail(rc);
    std::vector<uint64_t> ns(G);
    for (int g = 0; g < G; ++g) ns[g] = init_seed * 0x9e3779b97f4a7c15ull + 0x632be59bd9b4e019ull * (uint64_t)(g + 1);
    if ((rc = upload_vec(h, h->d_nseed, ns))) return bail(rc);
  }
  h->rowlen = ddrl_roundup(2 * S + A + 2, 4);
  if ((rc = dalloc(h, &h->d_rep, (size_t)G * cfg->replay_memory * h->rowlen, false))) return bail(rc);
  if ((rc = dalloc(h, &h->d_len, G))) return bail(rc);
  if ((rc = dalloc(h, &h->d_head, G))) return bail(rc);
  if ((rc = dalloc(h, &h->d_mt, (size_t)G * 624))) return bail(rc);
  if ((rc = dalloc(h, &h->d_mtpos, G))) return bail(rc);
  if ((rc = dalloc(h, &h->d_idx, (size_t)G * B))) return bail(rc);
  h->idx_cap = (size_t)G * B;
  if ((rc = dalloc(h, &h->d_tau, G))) return bail(rc);
  if ((rc = dalloc(h, &h->d_omtau, G))) return bail(rc);
  if ((rc = dalloc(h, &h->d_gamma, G))) return bail(rc);
  if ((rc = dalloc(h, &h->d_loss, G))) return bail(rc);
  if ((rc = dalloc(h, &h->d_qt, (size_t)G * B))) return bail(rc);
  if ((rc = dalloc(h, &h->d_dact, (size_t)G * B * A))) return bail(rc);
  if ((rc = dalloc(h, &h->d_stamps, (size_t)G * 16))) return bail(rc);
  if ((rc = dalloc(h, &h->d_s, (size_t)G * S))) return bail(rc);
  if ((rc = dalloc(h, &h->d_s2, (size_t)G * S))) return bail(rc);
  if ((rc = dalloc(h, &h->d_a, (size_t)G * A))) return bail(rc);
  if ((rc = dalloc(h, &h->d_act, (size_t)G * A))) return bail(rc);
  if ((rc = dalloc(h, &h->d_r, G))) return bail(rc);
  if ((rc = dalloc(h, &h->d_cost, G))) return bail(rc);
  if ((rc = dalloc(h, &h->d_goal, G))) return bail(rc);
  if ((rc = dalloc(h, &h->d_mask, G))) return bail(rc);
  if ((rc = dalloc(h, &h->d_flag, G))) return bail(rc);
  h->h_len.assign(G, 0);
  h->h_head.assign(G, 0);

  // hyper-parameter defaults
  std::vector<double> v(G);
  std::fill(v.begin(), v.end(), cfg->alpha_a); if ((rc = ddrl_pop_set_hyper(h, "alpha_a", v.data()))) return bail(rc);
  std::fill(v.begin(), v.end(), cfg->alpha_v); if ((rc = ddrl_pop_set_hyper(h, "alpha_v", v.data()))) return bail(rc);
  std::fill(v.begin(), v.end(), cfg->tau_soft_update); if ((rc = ddrl_pop_set_hyper(h, "tau_soft_update", v.data()))) return bail(rc);
  std::fill(v.begin(), v.end(), cfg->gamma); if ((rc = ddrl_pop_set_hyper(h, "gamma", v.data()))) return bail(rc);
  std::fill(v.begin(), v.end(), cfg->decay_v); if ((rc = ddrl_pop_set_hyper(h, "decay_v", v.data()))) return bail(rc);
  {
    std::vector<float> nod(G, -1.f);   // the actor never has weight decay (MLP.hpp:252-270)
    if ((rc = upload_vec(h, h->net[0].d_decay, nod))) return bail(rc);
  }
  // replay engines: default seeds 0..G-1 (bib::Seed seeds from clock+thread id, Seed.hpp:52-54)
  {
    std::vector<uint32_t> seeds(G);
    for (int g = 0; g < G; ++g) seeds[g] = (uint32_t)g;
    if ((rc = ddrl_pop_seed_replay(h, seeds.data()))) return bail(rc);
  }
  // gaussian(0, 0.01) fill of W and b (MLP.hpp:936-944); targets are clones (DeepQNAg.hpp:216-234)
  {
    std::mt19937_64 eng(init_seed);
    std::normal_distribution<double> nd01(0.0, 0.01);
    for (int k = 0; k < h->nnets; ++k) {
      std::vector<double> flat((size_t)G * h->net[k].nd.np);
      for (auto &x : flat) x = nd01(eng);
      for (int g = 0; g < G; ++g) fill_bn_defaults(h->net[k].nd, flat.data() + (size_t)g * h->net[k].nd.np);   // statistics 0, gamma 1, beta 0
      if ((rc = upload_flat(h, h->net[k].nd, h->net[k].st.w, flat.data()))) return bail(rc);
      if ((rc = upload_flat(h, h->net[k].nd, h->net[k].st.wt, flat.data()))) return bail(rc);
    }
  }

  // kernel parameter block
  DdpgParams &P = h->P;
  memset(&P, 0, sizeof(P));
  P.a = h->net[0].nd; P.q = h->net[1].nd;
  P.sa = h->net[0].st; P.sq = h->net[1].st; P.sq2 = h->net[2].st;
  P.noise = h->d_noise; P.loss2_out = h->d_loss2;
  P.G = G; P.S = S; P.A = A; P.B = B; P.cap = cfg->replay_memory; P.rowlen = h->rowlen;
  P.inverting = cfg->inverting_grad ? 1 : 0;
  P.nstep = cfg->nstep_return;
  P.ldxc = ddrl_ldpad(S + A); P.ldxa = ddrl_ldpad(S);
  P.ldh = 4;
  for (int k = 0; k < 2; ++k)
    for (int l = 1; l < h->net[k].nd.L(); ++l) P.ldh = std::max(P.ldh, h->net[k].nd.ld(l));
  double b1, b2;   // MLP.hpp:160-173
  switch (cfg->momentum) { case 0: b1 = 0.; b2 = 0.999; break; case 1: b1 = 0.9; b2 = 0.999; break;
                           case 2: b1 = 0.; b2 = 0.; break; default: b1 = 0.9; b2 = 0.; }
  h->beta1 = b1; h->beta2 = b2;
  P.beta1 = (float)b1; P.beta2 = (float)b2; P.omb1 = (float)(1. - b1); P.omb2 = (float)(1. - b2);
  P.eps = (float)1e-8;   // caffe SolverParameter.delta default
  P.fisher_beta = (float)cfg->ewc_fisher_beta;
  P.rep = h->d_rep; P.rep_len = h->d_len; P.rep_head = h->d_head; P.idx = h->d_idx;
  P.tau = h->d_tau; P.omtau = h->d_omtau; P.gamma = h->d_gamma;
  P.loss_out = h->d_loss; P.qt_out = h->d_qt; P.dact_out = h->d_dact;
  if ((rc = dalloc(h, &P.upd_done, G))) return bail(rc);
  if ((rc = dalloc(h, &P.upd_error, 1))) return bail(rc);
  P.nupd = 1;
  P.maxchunks = (B + DDRL_MB - 1) / DDRL_MB;
  // nets wider than the LDS-resident kernels take go layer by layer (kernels_ddpg_wide.hip)
  h->wide = S + A > DDRL_MAXW;
  for (int k = 0; k < 2; ++k)
    for (int l = 1; l < h->net[k].nd.L(); ++l) h->wide = h->wide || h->net[k].nd.d(l) > DDRL_MAXW;
  if ((rc = chunk_buffers(h))) return bail(rc);
  if (h->bn) {
    P.test_net = cfg->test_net ? 1 : 0; P.bn_adapt = cfg->bn_adapt ? 1 : 0;
    if ((rc = dalloc(h, &h->d_atest, (size_t)G * h->net[0].nd.stride()))) return bail(rc);
    P.a_test = h->d_atest;
    if ((rc = refresh_testing(h))) return bail(rc);
    if (ddpg_update_bn_lds_bytes(&P) > 160 * 1024)
      return bail(ddrl_fail(DDRL_EINVAL, "BatchNorm nets too wide for the LDS-resident update (%zu bytes of LDS)", ddpg_update_bn_lds_bytes(&P)));
  }
  if (!h->wide && ddpg_update_lds_bytes(&P) > 160 * 1024) return bail(ddrl_fail(DDRL_EINVAL, "network too wide for the LDS-resident kernel (%zu bytes of LDS)", ddpg_update_lds_bytes(&P)));
  if (hipStreamSynchronize(h->stream) != hipSuccess) return bail(ddrl_fail(DDRL_EHIP, "sync failed"));
  *out = h;
  return DDRL_OK;
}

// Policies only (algo 2): G actor nets and nothing else -- no optimiser state, targets or replay.
// CMAESAg's `ann` (CMAESAg.hpp:81-82: MLP(nb_sensors, hidden_unit_a, nb_motors, 0.1, 1, hidden type,
// output type, batch_norm)) for all lambda individuals of a generation at once.
extern "C" int ddrl_policy_create(int n_agents, int nb_sensors, int nb_motors, const int *hidden, int n_hidden,
                                  int hidden_layer_type, int output_layer_type, int device, uint64_t init_seed,
                                  ddrl_pop_t **out) {
  if (!hidden || !out) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  *out = nullptr;
  if (n_agents < 1) return ddrl_fail(DDRL_EINVAL, "n_agents < 1");
  if (nb_sensors < 1 || nb_motors < 1 || nb_sensors > DDRL_MAXW || nb_motors > DDRL_MAXW)
    return ddrl_fail(DDRL_EINVAL, "nb_sensors/nb_motors outside 1..%d", DDRL_MAXW);
  if (n_hidden < 1 || n_hidden > DDRL_MAX_HIDDEN) return ddrl_fail(DDRL_EINVAL, "1..%d hidden layers supported", DDRL_MAX_HIDDEN);
  for (int i = 0; i < n_hidden; ++i)
    if (hidden[i] < 1 || hidden[i] > DDRL_MAXW) return ddrl_fail(DDRL_EINVAL, "hidden[%d]=%d outside 1..%d", i, hidden[i], DDRL_MAXW);
  if (hidden_layer_type < 1 || hidden_layer_type > 3) return ddrl_fail(DDRL_EINVAL, "hidden_layer_type not in (1,2,3)");
  if (output_layer_type < 0 || output_layer_type > 3) return ddrl_fail(DDRL_EINVAL, "output_layer_type not in 0..3");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return ddrl_fail(DDRL_EHIP, "no HIP device: libddrl_pop has no CPU path");
  ddrl_pop *h = new ddrl_pop();
  h->algo = 2;
  memset(&h->cfg, 0, sizeof(h->cfg));
  h->cfg.n_agents = n_agents; h->cfg.nb_sensors = nb_sensors; h->cfg.nb_motors = nb_motors;
  h->cfg.n_hidden_a = n_hidden;
  for (int i = 0; i < n_hidden; ++i) h->cfg.hidden_unit_a[i] = hidden[i];
  h->cfg.hidden_layer_type = hidden_layer_type; h->cfg.actor_output_layer_type = output_layer_type;
  h->cfg.ewc = -1.; h->cfg.device = device;
  h->G = n_agents;
  if (device >= 0) h->device = device;
  else if (hipGetDevice(&h->device) != hipSuccess) h->device = 0;
  auto bail = [&](int code) { ddrl_pop_destroy(h); return code; };
  int rc;
  if ((rc = set_device(h))) return bail(rc);
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return bail(ddrl_fail(DDRL_EHIP, "hipStreamCreate failed"));
  const size_t G = n_agents;
  build_desc(h->net[0].nd, nb_sensors, hidden, n_hidden, nb_motors, hidden_layer_type, output_layer_type);
  if ((rc = dalloc(h, &h->net[0].st.w, G * h->net[0].nd.stride()))) return bail(rc);
  if ((rc = dalloc(h, &h->d_s, G * nb_sensors))) return bail(rc);
  if ((rc = dalloc(h, &h->d_act, G * nb_motors))) return bail(rc);
  if ((rc = dalloc(h, &h->d_mask, G))) return bail(rc);
  if ((rc = dalloc(h, &h->d_flag, G))) return bail(rc);
  {
    std::mt19937_64 eng(init_seed);                    // gaussian(0, 0.01) fill like the learners (MLP.hpp:936-944)
    std::normal_distribution<double> nd01(0.0, 0.01);
    std::vector<double> flat(G * h->net[0].nd.np);
    for (auto &x : flat) x = nd01(eng);
    if ((rc = upload_flat(h, h->net[0].nd, h->net[0].st.w, flat.data()))) return bail(rc);
  }
  *out = h;
  return DDRL_OK;
}

// MLP::copyWeightsFrom from device memory (fp32, learnable_params() order): row g of d_flat [G x ld]
// -> agent g's net; ld == 0 broadcasts one row to every (masked) agent.  This is how a CMA-ES
// generation (ddrl_cmaes_population_dev) becomes lambda policies without leaving the GPU.  The caller
// orders the producer of d_flat before this call (the ddrl_cmaes_* calls return synchronised).
extern "C" int ddrl_pop_load_flat_dev(ddrl_pop_t *h, int net, const float *d_flat, int ld, const uint8_t *mask) {
  if (!h || !d_flat || bad_net(h, net)) return ddrl_fail(DDRL_EINVAL, "bad argument");
  const TrainNet *t = train_net(h, net);
  if (ld != 0 && ld < t->nd.np) return ddrl_fail(DDRL_EINVAL, "ld=%d < %d parameters", ld, t->nd.np);
  int rc = set_device(h);
  if (rc) return rc;
  const uint8_t *dmask;
  if ((rc = upload_mask(h, mask, &dmask))) return rc;
  // load_flat_kernel walks the InnerProduct layers only: a net with BatchNorm / Scale towers keeps their statistics and
  // parameters ahead of the InnerProducts in the flat order (pack_flat), which this device-side unpacking does not do
  if (t->nd.bn_any()) return ddrl_fail(DDRL_EINVAL, "ddrl_pop_load_flat_dev does not serve nets with BatchNorm towers: use ddrl_pop_set_weights");
  HIPCHK(load_flat_launch(&t->nd, net_weights(h, net), d_flat, ld, dmask, h->G, h->stream));
  if (net == DDRL_NET_ACTOR && h->bn && (rc = refresh_testing(h))) return rc;       // ann_testing follows the actor (DeepQNAg.hpp:469-472)
  DDRL_WAIT(h);
  return DDRL_OK;
}

extern "C" int ddrl_pop_destroy(ddrl_pop_t *h) {
  if (!h) return DDRL_OK;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (void *p : h->allocs) (void)hipFree(p);
  for (auto &e : h->ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  if (h->stream) (void)hipStreamDestroy(h->stream);
  if (h->nfac) ddrl_nfac_host_free(h->nfac);
  delete h;
  return DDRL_OK;
}

extern "C" int ddrl_pop_shape(ddrl_pop_t *h, int *kind, int *n_agents, int *nb_sensors, int *nb_motors, int *mini_batch_size) {
  if (!h) return ddrl_fail(DDRL_EINVAL, "NULL handle");
  if (kind) *kind = h->algo == 0 && h->nnets == 3 ? 3 : h->algo;
  if (n_agents) *n_agents = h->G;
  if (nb_sensors) *nb_sensors = h->cfg.nb_sensors;
  if (nb_motors) *nb_motors = h->cfg.nb_motors;
  if (mini_batch_size) *mini_batch_size = h->algo == 0 ? h->cfg.mini_batch_size : 0;
  return DDRL_OK;
}

extern "C" int ddrl_pop_sync(ddrl_pop_t *h) {
  if (!h) return ddrl_fail(DDRL_EINVAL, "NULL handle");
  return ddrl_pop_wait(h);
}

extern "C" int ddrl_pop_set_hyper(ddrl_pop_t *h, const char *key, const double *v) {
  if (!h || !key || !v) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  int rc = set_device(h);
  if (rc) return rc;
  const int G = h->G;
  std::vector<float> f(G);
  std::string k(key);
  if (h->algo == 2) return ddrl_fail(DDRL_ESTATE, "a policies-only population has no hyper-parameters");
  if (k == "alpha_a" || k == "alpha_v") {
    for (int g = 0; g < G; ++g) { if (!(v[g] > 0)) return ddrl_fail(DDRL_EINVAL, "%s[%d] <= 0", key, g); f[g] = (float)v[g]; }
    if (k == "alpha_v" && h->nnets == 3 && (rc = upload_vec(h, h->net[2].d_lr, f))) return rc;   // qnn2 is built with alpha_v too
    return upload_vec(h, k == "alpha_a" ? h->net[0].d_lr : h->net[1].d_lr, f);
  }
  if (k == "tau_soft_update") {
    DDPG_ONLY(h);
    std::vector<float> om(G);
    // MLP.hpp:449-450: axpby(tau, from, (1.f - tau), to) -- (1.f - tau) evaluated in double
    for (int g = 0; g < G; ++g) { f[g] = (float)v[g]; om[g] = (float)(1.f - v[g]); }
    if ((rc = upload_vec(h, h->d_tau, f))) return rc;
    return upload_vec(h, h->d_omtau, om);
  }
  if (k == "gamma") { for (int g = 0; g < G; ++g) f[g] = (float)v[g]; return upload_vec(h, h->d_gamma, f); }
  if (k == "decay_v") {
    DDPG_ONLY(h);
    for (int g = 0; g < G; ++g) f[g] = (float)v[g];
    if (h->nnets == 3 && (rc = upload_vec(h, h->net[2].d_decay, f))) return rc;
    return upload_vec(h, h->net[1].d_decay, f);
  }
  return ddrl_fail(DDRL_EINVAL, "unknown hyper-parameter '%s'", key);
}

extern "C" int ddrl_pop_num_params(ddrl_pop_t *h, int net) {
  if (!h || bad_net(h, net)) return ddrl_fail(DDRL_EINVAL, "bad handle/net");
  return train_net(h, net)->nd.np;
}

extern "C" int ddrl_pop_set_weights(ddrl_pop_t *h, int net, const double *flat) {
  if (!h || !flat || bad_net(h, net)) return ddrl_fail(DDRL_EINVAL, "bad argument");
  int rc = set_device(h);
  if (rc) return rc;
  if ((rc = upload_flat(h, train_net(h, net)->nd, net_weights(h, net), flat))) return rc;
  return h->bn && net == DDRL_NET_ACTOR ? refresh_testing(h) : DDRL_OK;     // the acting copy follows (DeepQNAg.hpp:469-472)
}

extern "C" int ddrl_pop_get_weights(ddrl_pop_t *h, int net, double *flat) {
  if (!h || !flat || bad_net(h, net)) return ddrl_fail(DDRL_EINVAL, "bad argument");
  int rc = set_device(h);
  if (rc) return rc;
  return download_flat(h, train_net(h, net)->nd, net_weights(h, net), flat);
}

extern "C" int ddrl_pop_get_adam(ddrl_pop_t *h, int net, double *m, double *v, int32_t *iter) {
  if (!h || !trainable_net(h, net) || h->algo == 2) return ddrl_fail(DDRL_EINVAL, "Adam state exists for the actor and critic(s) of a learning population only");
  int rc = set_device(h);
  if (rc) return rc;
  TrainNet *t = train_net(h, net);
  if (m && (rc = download_flat(h, t->nd, t->st.m, m))) return rc;
  if (v && (rc = download_flat(h, t->nd, t->st.v, v))) return rc;
  if (iter) {
    HIPCHK(hipMemcpyAsync(iter, t->st.iter, h->G * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    DDRL_WAIT(h);
  }
  return DDRL_OK;
}

extern "C" int ddrl_pop_set_adam(ddrl_pop_t *h, int net, const double *m, const double *v, const int32_t *iter) {
  if (!h || !trainable_net(h, net) || h->algo == 2) return ddrl_fail(DDRL_EINVAL, "Adam state exists for the actor and critic(s) of a learning population only");
  int rc = set_device(h);
  if (rc) return rc;
  TrainNet *t = train_net(h, net);
  if (m && (rc = upload_flat(h, t->nd, t->st.m, m))) return rc;
  if (v && (rc = upload_flat(h, t->nd, t->st.v, v))) return rc;
  if (iter) {
    HIPCHK(hipMemcpyAsync(t->st.iter, iter, h->G * sizeof(int), hipMemcpyHostToDevice, h->stream));
    DDRL_WAIT(h);
  }
  return DDRL_OK;
}

extern "C" int ddrl_pop_seed_replay(ddrl_pop_t *h, const uint32_t *seeds) {
  if (!h || !seeds) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  DDPG_ONLY(h);
  int rc = set_device(h);
  if (rc) return rc;
  uint32_t *d_seeds = (uint32_t *)h->d_idx;   // G words of scratch (d_idx holds G*B >= G ints)
  HIPCHK(hipMemcpyAsync(d_seeds, seeds, h->G * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
  HIPCHK(mt_seed_launch(h->d_mt, h->d_mtpos, d_seeds, h->G, h->stream));
  DDRL_WAIT(h);
  return DDRL_OK;
}

extern "C" int ddrl_pop_replay_append(ddrl_pop_t *h, const float *s, const float *a, const float *r,
                                      const float *s2, const uint8_t *goal, const uint8_t *mask) {
  if (!h || !s || !a || !r || !s2 || !goal) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  DDPG_ONLY(h);
  int rc = set_device(h);
  if (rc) return rc;
  const int G = h->G, S = h->cfg.nb_sensors, A = h->cfg.nb_motors;
  const uint8_t *dmask;
  if ((rc = upload_mask(h, mask, &dmask))) return rc;
  HIPCHK(hipMemcpyAsync(h->d_s, s, (size_t)G * S * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->d_a, a, (size_t)G * A * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->d_r, r, (size_t)G * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->d_s2, s2, (size_t)G * S * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->d_goal, goal, (size_t)G, hipMemcpyHostToDevice, h->stream));
  HIPCHK(replay_append_launch(h->d_rep, h->d_len, h->d_head, G, h->cfg.replay_memory, h->rowlen, S, A,
                              h->d_s, h->d_a, h->d_r, h->d_s2, h->d_goal, dmask, h->stream));
  DDRL_WAIT(h);   // host buffers may be reused by the caller
  for (int g = 0; g < G; ++g) {
    if (mask && !mask[g]) continue;
    if (h->h_len[g] >= h->cfg.replay_memory) h->h_head[g] = (h->h_head[g] + 1) % h->cfg.replay_memory;
    else h->h_len[g]++;
  }
  return DDRL_OK;
}

extern "C" int ddrl_pop_replay_load(ddrl_pop_t *h, int agent, int n, const float *rows) {
  if (!h || !rows || agent < 0 || agent >= h->G || n < 0 || n > h->cfg.replay_memory)
    return ddrl_fail(DDRL_EINVAL, "bad argument");
  DDPG_ONLY(h);
  int rc = set_device(h);
  if (rc) return rc;
  const int S = h->cfg.nb_sensors, A = h->cfg.nb_motors, rl = 2 * S + A + 2;
  std::vector<float> img((size_t)n * h->rowlen, 0.f);
  for (int i = 0; i < n; ++i) memcpy(&img[(size_t)i * h->rowlen], rows + (size_t)i * rl, rl * sizeof(float));
  HIPCHK(hipMemcpyAsync(h->d_rep + (size_t)agent * h->cfg.replay_memory * h->rowlen, img.data(),
                        img.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
  h->h_len[agent] = n;
  h->h_head[agent] = 0;
  HIPCHK(hipMemcpyAsync(h->d_len + agent, &h->h_len[agent], sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->d_head + agent, &h->h_head[agent], sizeof(int), hipMemcpyHostToDevice, h->stream));
  DDRL_WAIT(h);
  return DDRL_OK;
}

// trajectory.clear() (DeepQNAg.hpp:254-257) for every (masked) agent
extern "C" int ddrl_pop_replay_clear(ddrl_pop_t *h, const uint8_t *mask) {
  if (!h) return ddrl_fail(DDRL_EINVAL, "NULL handle");
  DDPG_ONLY(h);
  int rc = set_device(h);
  if (rc) return rc;
  for (int g = 0; g < h->G; ++g)
    if (!mask || mask[g]) { h->h_len[g] = 0; h->h_head[g] = 0; }
  HIPCHK(hipMemcpyAsync(h->d_len, h->h_len.data(), h->G * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->d_head, h->h_head.data(), h->G * sizeof(int), hipMemcpyHostToDevice, h->stream));
  DDRL_WAIT(h);
  return DDRL_OK;
}

extern "C" int ddrl_pop_replay_len(ddrl_pop_t *h, int32_t *len) {
  if (!h || !len) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  DDPG_ONLY(h);
  int rc = set_device(h);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(len, h->d_len, h->G * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  DDRL_WAIT(h);
  return DDRL_OK;
}

extern "C" int ddrl_pop_infer_dev(ddrl_pop_t *h, const float *d_states, float *d_actions) {
  if (!h || !d_states || !d_actions) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  int rc = set_device(h);
  if (rc) return rc;
  const GateMode gm = next_draw(h, 1);
  HIPCHK(infer_launch(&h->net[0].nd, h->net[0].st.w, d_states, d_actions, h->gates_on ? h->d_sgate : nullptr,
                      h->gates_on ? h->d_agate : nullptr, &gm, h->G, h->stream));
  return DDRL_OK;
}

extern "C" int ddrl_pop_infer(ddrl_pop_t *h, const float *states, float *actions) {
  if (!h || !states || !actions) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  int rc = set_device(h);
  if (rc) return rc;
  const int G = h->G, S = h->cfg.nb_sensors, A = h->cfg.nb_motors;
  HIPCHK(hipMemcpyAsync(h->d_s, states, (size_t)G * S * sizeof(float), hipMemcpyHostToDevice, h->stream));
  if ((rc = ddrl_pop_infer_dev(h, h->d_s, h->d_act))) return rc;
  HIPCHK(hipMemcpyAsync(actions, h->d_act, (size_t)G * A * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  DDRL_WAIT(h);
  return DDRL_OK;
}

// MLP::computeOutBatch / computeOutVFBatch (MLP.hpp:531-579): n rows per agent through one net
extern "C" int ddrl_pop_forward_batch(ddrl_pop_t *h, int net, int n, const float *x, float *out) {
  if (!h || !x || !out || n < 1 || bad_net(h, net)) return ddrl_fail(DDRL_EINVAL, "bad argument");
  int rc = set_device(h);
  if (rc) return rc;
  const TrainNet *t = train_net(h, net);
  bool wide = false;                                       // some layer beyond the LDS-resident passes: the global-scratch kernel
  for (int l = 0; l <= t->nd.L(); ++l) wide = wide || t->nd.d(l) > DDRL_MAXW;
  const size_t G = h->G, d0 = t->nd.d(0), dl = t->nd.d(t->nd.L());
  float *dx = nullptr, *dout = nullptr, *dscr = nullptr;
  HIPCHK(hipMalloc(&dx, G * n * d0 * sizeof(float)));
  if (hipMalloc(&dout, G * n * dl * sizeof(float)) != hipSuccess) { (void)hipFree(dx); return ddrl_fail(DDRL_ENOMEM, "hipMalloc failed"); }
  if (wide && hipMalloc(&dscr, net_forward_wide_scratch_floats(&t->nd, n, h->G) * sizeof(float)) != hipSuccess) {
    (void)hipFree(dx); (void)hipFree(dout);
    return ddrl_fail(DDRL_ENOMEM, "hipMalloc failed");
  }
  const bool actor = net == DDRL_NET_ACTOR || net == DDRL_NET_ACTOR_TARGET;
  const bool towers = t->nd.bn_any();                      // evaluated as a TEST-phase copy would (kernels_ddpg_bn.hip)
  const float *sg = h->gates_on ? h->d_sgate : nullptr, *ag = h->gates_on ? h->d_agate : nullptr;
  const GateMode gm = next_draw(h, 1);
  hipError_t e = hipMemcpyAsync(dx, x, G * n * d0 * sizeof(float), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess)
    e = towers ? net_forward_bn_launch(&t->nd, net_weights(h, net), dx, dout, n, h->G, sg, h->cfg.nb_sensors, ag, h->cfg.nb_motors,
                                       actor ? 1 : 0, &gm, h->stream) :
        wide ? net_forward_wide_launch(&t->nd, net_weights(h, net), dx, dout, dscr, n, h->G, sg, h->cfg.nb_sensors, ag,
                                       h->cfg.nb_motors, actor ? 1 : 0, &gm, h->stream)
             : net_forward_launch(&t->nd, net_weights(h, net), dx, dout, n, h->G, sg, h->cfg.nb_sensors, ag, h->cfg.nb_motors,
                                  actor ? 1 : 0, &gm, h->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(out, dout, G * n * dl * sizeof(float), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void)hipFree(dx);
  (void)hipFree(dout);
  if (dscr) (void)hipFree(dscr);
  if (e != hipSuccess) return ddrl_fail(DDRL_EHIP, "forward_batch: %s", hipGetErrorString(e));
  return DDRL_OK;
}

// n consecutive passes of the fupd loop for every (masked) agent.  idx: host indices [G x n x B] or NULL
// (device draws).  The one-tile kernel runs them in ONE launch; the chunked / layer-by-layer paths
// launch n times.
static int ddpg_update_n(ddrl_pop *h, int n, const int32_t *idx, const uint8_t *mask) {
  if (h->nnets == 3) return ddrl_fail(DDRL_ESTATE, "a TD3 population is updated with ddrl_pop_td3_update");
  int rc = set_device(h);
  if (rc) return rc;
  const int G = h->G, B = h->cfg.mini_batch_size;
  const bool fused = !h->wide && h->P.maxchunks == 1 && !h->force_chunked && !h->bn;   // towers: one update per launch (kernels_ddpg_bn.hip)
  const uint8_t *dmask;
  if ((rc = upload_mask(h, mask, &dmask))) return rc;
  if (idx) {
    for (int g = 0; g < G; ++g) {
      if (mask && !mask[g]) continue;
      if (h->h_len[g] < 250 || h->h_len[g] < B) continue;
      for (size_t i = 0; i < (size_t)n * B; ++i)
        if (idx[(size_t)g * n * B + i] < 0 || idx[(size_t)g * n * B + i] >= h->h_len[g])
          return ddrl_fail(DDRL_EINVAL, "idx[%d][%zu]=%d outside the replay of %d samples", g, i, idx[(size_t)g * n * B + i], h->h_len[g]);
    }
  }
  const int per_launch = fused ? n : 1;
  if ((size_t)G * B * per_launch > h->idx_cap) {       // grows once; the old buffer stays on the handle's free list
    DDRL_WAIT(h);
    if ((rc = dalloc(h, &h->d_idx, (size_t)G * B * per_launch))) return rc;
    h->idx_cap = (size_t)G * B * per_launch;
    h->P.idx = h->d_idx;
  }
  h->P.mask = dmask;
  h->P.sgate = h->gates_on ? h->d_sgate : nullptr;
  h->P.agate = h->gates_on ? h->d_agate : nullptr;
  h->P.sa = h->net[0].st;
  h->P.sq = h->net[1].st;
  h->P.nupd = per_launch;
  // launches of several updates: two agents per workgroup, one quarter of an update apart (kernels_ddpg_pair.hip)
  const bool pair = fused && !h->force_generic && h->pair_mode > 0 &&
                    (h->pair_mode == 2 || (per_launch >= DDRL_PAIR_MIN_UPD && ddpg_update_pair_pays(&h->P))) && ddpg_update_pair_ok(&h->P);
  // small nets: the agent's whole state in LDS for the launch (worth the copy in / out from two updates on)
  // (opt-in, "resident" option: measured SLOWER than the one-agent kernel -- one workgroup per CU instead of three,
  // and the small nets' update is bound by its chain of tiny dependent passes, not by HBM: DESIGN.md section 4.1e)
  const bool resident = fused && !h->force_generic && h->resident_mode > 0 && per_launch >= 2 &&
                        !h->P.stamps && ddpg_update_resident_ok(&h->P);
  // the default for multi-update launches of the compile-time shapes: one workgroup walks its agent through the launch
  // (ddpg_update_loop_kernel); option "loop" 0 falls back to one workgroup per (update, agent) with the sc1 hand-off
  const bool loop = fused && !h->force_generic && h->loop_mode > 0 && per_launch >= 2 && !pair && !resident && ddpg_update_loop_ok(&h->P) &&
                    (h->loop_mode == 2 || ddpg_update_loop_pays(&h->P));
  // opt-in (option "w8": 0 never (default), 1 up to DDRL_W8_MAX_AGENTS agents, 2 always): the same loop with eight waves per agent
  // (kernels_ddpg_w8.hip), meant for populations that leave CUs without a second workgroup.  Measured SLOWER than four waves
  // (128 agents: 1.19 M against 1.26 M updates/s): the update is bound by what a SIMD can issue (fp32 MFMAs run on the vector
  // ALU), not by the waits a second wave could fill, and the shared row tiles cost a barrier per layer
  const bool w8 = fused && !h->force_generic && h->w8_mode > 0 && !pair && !resident && ddpg_update_loop8_ok(&h->P) &&
                  (h->w8_mode == 2 || (h->loop_mode > 0 && G <= DDRL_W8_MAX_AGENTS));
  h->P.pair = pair || resident || loop || w8 ? 1 : 0;
  std::vector<int32_t> one;
  for (int l = 0; l < n; l += per_launch) {
    if (idx) {
      const int32_t *src = idx;
      if (!fused && n > 1) {                            // launch l takes update l of every agent
        one.resize((size_t)G * B);
        for (int g = 0; g < G; ++g) memcpy(&one[(size_t)g * B], idx + ((size_t)g * n + l) * B, B * sizeof(int32_t));
        src = one.data();
      }
      HIPCHK(hipMemcpyAsync(h->d_idx, src, (size_t)G * B * per_launch * sizeof(int), hipMemcpyHostToDevice, h->stream));
      if (!fused && n > 1) HIPCHK(hipStreamSynchronize(h->stream));
    }
    HIPCHK(replay_draw_launch(h->d_mt, h->d_mtpos, h->d_len, G, B, per_launch, dmask, h->d_idx, idx ? 0 : 1, h->beta1, h->beta2,
                              h->net[0].d_lr, h->net[1].d_lr, h->net[0].st.iter, h->net[1].st.iter,
                              h->net[0].st.lr_t, h->net[1].st.lr_t, h->P.upd_done, h->stream));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->profiling) {
      if (h->ev_used == h->ev_pool.size()) {
        HIPCHK(hipEventCreate(&e0));
        HIPCHK(hipEventCreate(&e1));
        h->ev_pool.emplace_back(e0, e1);
      }
      e0 = h->ev_pool[h->ev_used].first; e1 = h->ev_pool[h->ev_used].second;
      h->ev_used++;
      HIPCHK(hipEventRecord(e0, h->stream));
    }
    h->P.gm = next_draw(h, per_launch);
    if (h->bn) HIPCHK(ddpg_update_bn_launch(&h->P, h->stream));
    else if (h->wide) HIPCHK(ddpg_update_wide_launch(&h->P, h->stream));
    else if (!fused) HIPCHK(ddpg_update_big_launch(&h->P, h->stream));
    else if (resident) HIPCHK(ddpg_update_resident_launch(&h->P, h->stream));
    else if (pair) HIPCHK(ddpg_update_pair_launch(&h->P, h->stream));
    else if (w8) HIPCHK(ddpg_update_loop8_launch(&h->P, h->stream));
    else if (loop) HIPCHK(ddpg_update_loop_launch(&h->P, h->stream));
    else HIPCHK(ddpg_update_launch(&h->P, h->force_generic, h->stream));
    if (h->profiling) HIPCHK(hipEventRecord(e1, h->stream));
  }
  h->P.pair = 0;
  if (h->bn && (rc = refresh_testing(h))) return rc;        // DeepQNAg.hpp:469-472, after the fupd loop
  if (per_launch > 1 || pair || resident || loop || w8) h->multi_pending = true;     // every kernel that can raise the error word (team barriers too)
  if (idx || mask) HIPCHK(hipStreamSynchronize(h->stream));   // host arrays were read asynchronously
  return DDRL_OK;
}

extern "C" int ddrl_pop_ddpg_update(ddrl_pop_t *h, const int32_t *idx, const uint8_t *mask) {
  if (!h) return ddrl_fail(DDRL_EINVAL, "NULL handle");
  DDPG_ONLY(h);
  return ddpg_update_n(h, 1, idx, mask);
}

extern "C" int ddrl_pop_ddpg_update_n(ddrl_pop_t *h, int n, const int32_t *idx, const uint8_t *mask) {
  if (!h) return ddrl_fail(DDRL_EINVAL, "NULL handle");
  DDPG_ONLY(h);
  if (n < 1 || n > DDRL_MAXUPD) return ddrl_fail(DDRL_EINVAL, "n outside 1..%d", DDRL_MAXUPD);
  return ddpg_update_n(h, n, idx, mask);
}

// TD3Ag's end_episode (TD3Ag.hpp:284-380) for every (masked) agent: one launch of the fused kernel
extern "C" int ddrl_pop_td3_update(ddrl_pop_t *h, const int32_t *idx, const float *noise, const uint8_t *actor_step,
                                   const uint8_t *mask) {
  if (!h) return ddrl_fail(DDRL_EINVAL, "NULL handle");
  DDPG_ONLY(h);
  if (h->nnets != 3) return ddrl_fail(DDRL_ESTATE, "not a TD3 population (ddrl_td3_create)");
  int rc = set_device(h);
  if (rc) return rc;
  const int G = h->G, B = h->cfg.mini_batch_size, A = h->cfg.nb_motors;
  const uint8_t *dmask;
  if ((rc = upload_mask(h, mask, &dmask))) return rc;
  if (idx) {
    for (int g = 0; g < G; ++g) {
      if (mask && !mask[g]) continue;
      if (h->h_len[g] < 250 || h->h_len[g] < B) continue;
      for (int i = 0; i < B; ++i)
        if (idx[(size_t)g * B + i] < 0 || idx[(size_t)g * B + i] >= h->h_len[g])
          return ddrl_fail(DDRL_EINVAL, "idx[%d][%d]=%d outside the replay of %d samples", g, i, idx[(size_t)g * B + i], h->h_len[g]);
    }
    HIPCHK(hipMemcpyAsync(h->d_idx, idx, (size_t)G * B * sizeof(int), hipMemcpyHostToDevice, h->stream));
  }
  if (noise) HIPCHK(hipMemcpyAsync(h->d_noise, noise, (size_t)G * B * A * sizeof(float), hipMemcpyHostToDevice, h->stream));
  if (actor_step) HIPCHK(hipMemcpyAsync(h->d_astep, actor_step, (size_t)G, hipMemcpyHostToDevice, h->stream));
  DdpgParams &P = h->P;
  P.mask = dmask;
  P.sgate = h->gates_on ? h->d_sgate : nullptr;
  P.agate = h->gates_on ? h->d_agate : nullptr;
  P.gm = next_draw(h, 1);
  P.sa = h->net[0].st; P.sq = h->net[1].st; P.sq2 = h->net[2].st;
  P.actor_step = actor_step ? h->d_astep : nullptr;
  P.nupd = 1;
  P.inverting = 0;
  HIPCHK(replay_draw_launch(h->d_mt, h->d_mtpos, h->d_len, G, B, 1, dmask, h->d_idx, idx ? 0 : 1, h->beta1, h->beta2,
                            h->net[0].d_lr, h->net[1].d_lr, h->net[0].st.iter, h->net[1].st.iter,
                            h->net[0].st.lr_t, h->net[1].st.lr_t, P.upd_done, h->stream));
  HIPCHK(td3_prep_launch(h->d_nseed, h->d_ncount, h->d_len, G, B, A, dmask, h->d_noise, noise ? 0 : 1,
                         (float)h->cfg.policy_noise, (float)h->cfg.noise_clip, h->beta1, h->beta2, h->net[2].d_lr,
                         h->net[2].st.iter, h->net[2].st.lr_t, h->stream));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (h->profiling) {
    if (h->ev_used == h->ev_pool.size()) {
      HIPCHK(hipEventCreate(&e0));
      HIPCHK(hipEventCreate(&e1));
      h->ev_pool.emplace_back(e0, e1);
    }
    e0 = h->ev_pool[h->ev_used].first; e1 = h->ev_pool[h->ev_used].second;
    h->ev_used++;
    HIPCHK(hipEventRecord(e0, h->stream));
  }
  int any_actor = actor_step ? 0 : 1;
  for (int g = 0; g < G && !any_actor; ++g) any_actor = actor_step[g] && (!mask || mask[g]);
  if (h->bn) {
    HIPCHK(td3_update_bn_launch(&P, any_actor, h->stream));
  } else if (h->wide) {
    HIPCHK(td3_update_wide_launch(&P, any_actor, h->stream));
  } else if (P.maxchunks > 1 || h->force_chunked) {
    HIPCHK(td3_update_big_launch(&P, any_actor, h->stream));
  } else {
    HIPCHK(td3_update_launch(&P, h->force_generic, h->stream));
  }
  if (h->profiling) HIPCHK(hipEventRecord(e1, h->stream));
  // TD3Ag.hpp:372-375: the acting copy follows the actor inside the policy step (agents off their policy step have not moved:
  // copying theirs again changes nothing)
  if (h->bn && any_actor && (rc = refresh_testing(h))) return rc;
  if (idx || mask || noise || actor_step) HIPCHK(hipStreamSynchronize(h->stream));   // host arrays were read asynchronously
  return DDRL_OK;
}

extern "C" int ddrl_pop_td3_seed_noise(ddrl_pop_t *h, const uint64_t *seeds) {
  if (!h || !seeds) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  if (h->nnets != 3) return ddrl_fail(DDRL_ESTATE, "not a TD3 population");
  int rc = set_device(h);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(h->d_nseed, seeds, h->G * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemsetAsync(h->d_ncount, 0, h->G * sizeof(uint32_t), h->stream));
  DDRL_WAIT(h);
  return DDRL_OK;
}

extern "C" int ddrl_pop_td3_last(ddrl_pop_t *h, float *loss2, float *noise) {
  if (!h) return ddrl_fail(DDRL_EINVAL, "NULL handle");
  if (h->nnets != 3) return ddrl_fail(DDRL_ESTATE, "not a TD3 population");
  int rc = set_device(h);
  if (rc) return rc;
  const size_t G = h->G, B = h->cfg.mini_batch_size, A = h->cfg.nb_motors;
  if (loss2) HIPCHK(hipMemcpyAsync(loss2, h->d_loss2, G * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  if (noise) HIPCHK(hipMemcpyAsync(noise, h->d_noise, G * B * A * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  DDRL_WAIT(h);
  return DDRL_OK;
}

extern "C" int ddrl_td3_create(const ddrl_ddpg_config *cfg, uint64_t init_seed, ddrl_pop_t **out) {
  if (!cfg || !out) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  ddrl_ddpg_config c = *cfg;
  c.td3 = 1;
  return ddrl_ddpg_create(&c, init_seed, out);
}

extern "C" int ddrl_pop_ddpg_last(ddrl_pop_t *h, int32_t *idx, float *loss, float *qt, float *dact) {
  if (!h) return ddrl_fail(DDRL_EINVAL, "NULL handle");
  DDPG_ONLY(h);
  int rc = set_device(h);
  if (rc) return rc;
  const size_t G = h->G, B = h->cfg.mini_batch_size, A = h->cfg.nb_motors;
  if (idx) {   // the indices of the last update of the last launch
    const size_t nu = h->P.nupd > 0 ? h->P.nupd : 1;
    HIPCHK(hipMemcpy2DAsync(idx, B * 4, h->d_idx + (nu - 1) * B, nu * B * 4, B * 4, G, hipMemcpyDeviceToHost, h->stream));
  }
  if (loss) HIPCHK(hipMemcpyAsync(loss, h->d_loss, G * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  if (qt) HIPCHK(hipMemcpyAsync(qt, h->d_qt, G * B * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  if (dact) HIPCHK(hipMemcpyAsync(dact, h->d_dact, G * B * A * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  DDRL_WAIT(h);
  return DDRL_OK;
}

// Developmental gates (DODevMLP.hpp:376-400; semantics SURVEY.md 8h): multiplicative per-dimension
// gates on the state inputs of every net, on the actor's outputs and on the critic's action inputs,
// shared by all nets of an agent; not learned (heuristic != 0: lr_mult = 0).  NULL, NULL switches
// them off.  The schedule (DODevMLP::inform) is the caller's.
extern "C" int ddrl_pop_set_gates(ddrl_pop_t *h, const double *state_gate, const double *action_gate) {
  if (!h) return ddrl_fail(DDRL_EINVAL, "NULL handle");
  int rc = set_device(h);
  if (rc) return rc;
  if (!state_gate && !action_gate) { h->gates_on = false; h->gate_w_s.clear(); h->gate_w_a.clear(); return DDRL_OK; }
  const size_t G = h->G, S = h->cfg.nb_sensors, A = h->cfg.nb_motors;
  if (!h->d_sgate) {
    if ((rc = dalloc(h, &h->d_sgate, G * S))) return rc;
    if ((rc = dalloc(h, &h->d_agate, G * A))) return rc;
  }
  std::vector<float> sg(G * S, 1.f), ag(G * A, 1.f);
  if (state_gate) for (size_t i = 0; i < G * S; ++i) sg[i] = (float)state_gate[i];
  if (action_gate) for (size_t i = 0; i < G * A; ++i) ag[i] = (float)action_gate[i];
  if ((rc = upload_vec(h, h->d_sgate, sg))) return rc;
  if ((rc = upload_vec(h, h->d_agate, ag))) return rc;
  h->gates_on = true;
  h->gate_w_s.assign(sg.begin(), sg.end()); h->gate_w_a.assign(ag.begin(), ag.end());   // the view's weights, until dev_set_weights says otherwise
  return DDRL_OK;
}

extern "C" int ddrl_pop_get_gates(ddrl_pop_t *h, double *state_gate, double *action_gate) {
  if (!h) return ddrl_fail(DDRL_EINVAL, "NULL handle");
  int rc = set_device(h);
  if (rc) return rc;
  const size_t G = h->G, S = h->cfg.nb_sensors, A = h->cfg.nb_motors;
  std::vector<float> sg(G * S, 1.f), ag(G * A, 1.f);
  if (h->gates_on) {
    HIPCHK(hipMemcpyAsync(sg.data(), h->d_sgate, sg.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(ag.data(), h->d_agate, ag.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    DDRL_WAIT(h);
  }
  if (state_gate) for (size_t i = 0; i < G * S; ++i) state_gate[i] = sg[i];
  if (action_gate) for (size_t i = 0; i < G * A; ++i) action_gate[i] = ag[i];
  return DDRL_OK;
}

// DODevMLP's `probabilist` / `scale` keys (DODevMLP.hpp:276-304; per gate kind: st_probabilistic / ac_probabilistic,
// st_scale / ac_scale): *_bernoulli != 0 selects the per-sample Bernoulli(w) gate (the reference's probabilist = 0),
// *_scale divides a passed value by w.  seed starts the counter-based draw stream again.
extern "C" int ddrl_pop_set_gate_mode(ddrl_pop_t *h, int st_bernoulli, int st_scale, int ac_bernoulli, int ac_scale, uint32_t seed) {
  if (!h) return ddrl_fail(DDRL_EINVAL, "NULL handle");
  h->gm.st_bern = st_bernoulli ? 1 : 0; h->gm.st_scale = st_scale ? 1 : 0;
  h->gm.ac_bern = ac_bernoulli ? 1 : 0; h->gm.ac_scale = ac_scale ? 1 : 0;
  h->gm.seed = seed; h->gm.draw = 0;
  h->P.gm = h->gm;
  return DDRL_OK;
}

// ---- DODevMLP's flat parameter view (gate blobs included) ----
extern "C" int ddrl_pop_set_gate_layout(ddrl_pop_t *h, const int *st_control, int n_st, const int *ac_control, int n_ac, int learnable) {
  if (!h) return ddrl_fail(DDRL_EINVAL, "NULL handle");
  const int S = h->cfg.nb_sensors, A = h->cfg.nb_motors;
  if ((n_st > 0 && !st_control) || (n_ac > 0 && !ac_control) || n_st > S || n_ac > A) return ddrl_fail(DDRL_EINVAL, "bad control list");
  std::vector<int> st, ac;
  if (n_st == 0) for (int i = 0; i < S; ++i) st.push_back(i);                    // an empty list gates every dimension (CaffeUtest.cpp:629-636)
  for (int i = 0; i < n_st; ++i) { if (st_control[i] < 0 || st_control[i] >= S) return ddrl_fail(DDRL_EINVAL, "st_control %d outside 0..%d", st_control[i], S - 1); st.push_back(st_control[i]); }
  if (n_ac == 0) for (int i = 0; i < A; ++i) ac.push_back(i);
  for (int i = 0; i < n_ac; ++i) { if (ac_control[i] < 0 || ac_control[i] >= A) return ddrl_fail(DDRL_EINVAL, "ac_control %d outside 0..%d", ac_control[i], A - 1); ac.push_back(ac_control[i]); }
  h->st_control = st; h->ac_control = ac;
  h->st_layer = n_st >= 0; h->ac_layer = n_ac >= 0;
  h->gates_learnable = learnable != 0;
  if (!h->gates_on && (h->st_layer || h->ac_layer)) {
    std::vector<double> one((size_t)h->G * (size_t)std::max(S, A), 1.);
    return ddrl_pop_set_gates(h, one.data(), one.data());
  }
  return DDRL_OK;
}

// gate weights in the view: which, and where.  Actor: [state gate][InnerProducts][action gate]; critic: [action gate][state gate][InnerProducts]
static void dev_view(const ddrl_pop *h, int net, int ignore_null_lr, bool &actor, size_t &ns, size_t &na) {
  actor = net == DDRL_NET_ACTOR || net == DDRL_NET_ACTOR_TARGET;
  const bool counted = h->gates_learnable || !ignore_null_lr;
  ns = counted && h->st_layer ? h->st_control.size() : 0;
  na = counted && h->ac_layer ? h->ac_control.size() : 0;
}

// the flat positions of a net's lr_mult-0 blobs -- a tower's [mean_sum, var_sum, factor] -- which number_of_parameters(true) /
// copyWeightsTo(w, true) / copyWeightsFrom(w, true) leave out (MLP.hpp:610-676; the reference's Developmental tests copy through
// that view on every batch_norm code, CaffeUtest.cpp:646-656)
static std::vector<char> null_lr_mask(const NetDesc &nd, bool ignore_null_lr) {
  std::vector<char> m(nd.np, 0);
  if (!ignore_null_lr) return m;
  size_t k = 0;
  for (int l = 0; l <= nd.L(); ++l) {
    if (nd.bn(l)) {
      const int C = nd.bn_c(l);
      for (int c = 0; c < 2 * C + 1; ++c) m[k++] = 1;
      if (nd.bn_scale()) k += C;
      if (nd.bn_bias()) k += C;
    }
    if (l == nd.L()) break;
    k += (size_t)nd.d(l + 1) * nd.d(l) + nd.d(l + 1);
  }
  return m;
}

extern "C" int ddrl_pop_dev_num_params(ddrl_pop_t *h, int net, int ignore_null_lr, int *n) {
  if (!h || !n || bad_net(h, net)) return ddrl_fail(DDRL_EINVAL, "bad argument");
  bool actor; size_t ns, na;
  dev_view(h, net, ignore_null_lr, actor, ns, na);
  const std::vector<char> skip = null_lr_mask(train_net(h, net)->nd, ignore_null_lr != 0);
  *n = train_net(h, net)->nd.np - (int)std::count(skip.begin(), skip.end(), (char)1) + (int)(ns + na);
  return DDRL_OK;
}

extern "C" int ddrl_pop_dev_get_weights(ddrl_pop_t *h, int net, int ignore_null_lr, double *flat) {
  if (!h || !flat || bad_net(h, net)) return ddrl_fail(DDRL_EINVAL, "bad argument");
  bool actor; size_t ns, na;
  dev_view(h, net, ignore_null_lr, actor, ns, na);
  const std::vector<char> skip = null_lr_mask(train_net(h, net)->nd, ignore_null_lr != 0);
  const size_t G = h->G, S = h->cfg.nb_sensors, A = h->cfg.nb_motors, np = train_net(h, net)->nd.np;
  const size_t n = np - (size_t)std::count(skip.begin(), skip.end(), (char)1) + ns + na;
  std::vector<double> w(G * np), sg(h->gate_w_s), ag(h->gate_w_a);
  sg.resize(G * S, 1.); ag.resize(G * A, 1.);
  int rc = ddrl_pop_get_weights(h, net, w.data());
  if (rc) return rc;
  for (size_t g = 0; g < G; ++g) {
    double *o = flat + g * n;
    size_t k = 0;
    if (!actor) for (size_t i = 0; i < na; ++i) o[k++] = ag[g * A + h->ac_control[i]];
    for (size_t i = 0; i < ns; ++i) o[k++] = sg[g * S + h->st_control[i]];
    for (size_t i = 0; i < np; ++i) if (!skip[i]) o[k++] = w[g * np + i];
    if (actor) for (size_t i = 0; i < na; ++i) o[k++] = ag[g * A + h->ac_control[i]];
  }
  return DDRL_OK;
}

extern "C" int ddrl_pop_dev_set_weights(ddrl_pop_t *h, int net, int ignore_null_lr, const double *flat) {
  if (!h || !flat || bad_net(h, net)) return ddrl_fail(DDRL_EINVAL, "bad argument");
  bool actor; size_t ns, na;
  dev_view(h, net, ignore_null_lr, actor, ns, na);
  const std::vector<char> skip = null_lr_mask(train_net(h, net)->nd, ignore_null_lr != 0);
  const size_t nskip = (size_t)std::count(skip.begin(), skip.end(), (char)1);
  const size_t G = h->G, S = h->cfg.nb_sensors, A = h->cfg.nb_motors, np = train_net(h, net)->nd.np, n = np - nskip + ns + na;
  std::vector<double> w(G * np), sg(h->gate_w_s), ag(h->gate_w_a);
  sg.resize(G * S, 1.); ag.resize(G * A, 1.);
  int rc;
  if (nskip && (rc = ddrl_pop_get_weights(h, net, w.data()))) return rc;     // the blobs the view leaves out stay as they are
  for (size_t g = 0; g < G; ++g) {
    const double *o = flat + g * n;
    size_t k = 0;
    if (!actor) for (size_t i = 0; i < na; ++i) ag[g * A + h->ac_control[i]] = o[k++];
    for (size_t i = 0; i < ns; ++i) sg[g * S + h->st_control[i]] = o[k++];
    for (size_t i = 0; i < np; ++i) if (!skip[i]) w[g * np + i] = o[k++];
    if (actor) for (size_t i = 0; i < na; ++i) ag[g * A + h->ac_control[i]] = o[k++];
  }
  if ((rc = ddrl_pop_set_weights(h, net, w.data()))) return rc;
  if (!(ns + na)) return DDRL_OK;
  // the layer's forward from its weight (DevelopmentalLayer): probabilist = 1 passes a dimension whose weight is
  // at least 0.5 and zeroes the others; probabilist = 0 draws Bernoulli(w) per sample, so the device holds w itself (within [0, 1])
  std::vector<double> fs(sg), fa(ag);
  for (size_t g = 0; g < G; ++g) {
    for (int c : h->st_control) { double &v = fs[g * S + c]; v = h->gm.st_bern ? std::min(1., std::max(0., v)) : (v < 0.5 ? 0. : 1.); }
    for (int c : h->ac_control) { double &v = fa[g * A + c]; v = h->gm.ac_bern ? std::min(1., std::max(0., v)) : (v < 0.5 ? 0. : 1.); }
  }
  if ((rc = ddrl_pop_set_gates(h, fs.data(), fa.data()))) return rc;
  h->gate_w_s = sg; h->gate_w_a = ag;
  return DDRL_OK;
}

// ---- EWC life-cycle ----
extern "C" int ddrl_pop_update_best_param(ddrl_pop_t *h, const double *score, const uint8_t *mask) {
  if (!h || !score) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  int rc = set_device(h);
  if (rc) return rc;
  for (int k = 0; k < h->nnets; ++k) {
    TrainNet &t = h->net[k];
    if (!t.ewc.enabled) continue;
    std::vector<uint8_t> flag(h->G, 0);
    bool any = false;
    for (int g = 0; g < h->G; ++g) {
      if (mask && !mask[g]) continue;
      bool go = false, too_old;
      EwcHost &e = t.ewc;
      switch (h->cfg.ewc_best_param_method) {   // DODevMLP.hpp:726-747
        case 3: e.last_update_best[g]++; go = e.last_update_best[g] == 1; break;
        case 2: too_old = e.last_update_best[g] > 50; e.last_update_best[g]++;
                go = (score[g] > e.best_score[g]) || too_old; if (go) e.last_update_best[g] = 0; break;
        case 4: case 0: go = score[g] > e.best_score[g]; break;
        case 1: go = true; break;
      }
      if (go) { e.best_score[g] = score[g]; e.best_valid[g] = 1; flag[g] = 1; any = true; }
    }
    if (!any) continue;
    if ((rc = upload_vec(h, h->d_flag, flag))) return rc;
    HIPCHK(snapshot_launch(t.st.best, t.st.w, t.nd.stride(), h->d_flag, h->G, h->stream));
    DDRL_WAIT(h);
  }
  return DDRL_OK;
}

// DODevMLP::ewc_setup (DODevMLP.hpp:604-633) of one net for every (masked) agent with a best_param
static int ewc_setup_one(ddrl_pop *h, TrainNet &t, const uint8_t *mask) {
  if (!t.ewc.enabled) return DDRL_OK;
  std::vector<uint8_t> flag(h->G, 0);
  std::vector<float> nbr(h->G);
  HIPCHK(hipMemcpyAsync(nbr.data(), t.st.fisher_nbr, h->G * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  DDRL_WAIT(h);
  bool any = false;
  for (int g = 0; g < h->G; ++g) {
    if (mask && !mask[g]) continue;
    if (!t.ewc.best_valid[g]) continue;             // DODevMLP.hpp:605: best_param == nullptr
    // no gradient step since the last setup: the reference dereferences its dropped Fisher vector
    // here (:620-621, fisher == nullptr after :629-630); the previous anchors are kept instead
    if (!(nbr[g] > 0.f)) continue;
    flag[g] = 1; any = true;
    t.ewc.decay_mult[g] = 1.0;
    t.ewc.best_score[g] = std::numeric_limits<double>::lowest();
    t.ewc.live[g] = 1;
  }
  if (!any) return DDRL_OK;
  int rc;
  if ((rc = upload_vec(h, h->d_flag, flag))) return rc;
  HIPCHK(ewc_setup_launch(&t.nd, t.st.w, t.st.best, t.st.wstar, t.st.fisher, t.st.fhat, t.st.fisher_nbr, t.nd.stride(),
                          h->d_flag, h->G, h->stream));
  DDRL_WAIT(h);
  return refresh_ewc_coef(h, t);
}

extern "C" int ddrl_pop_ewc_setup(ddrl_pop_t *h, const uint8_t *mask) {
  if (!h) return ddrl_fail(DDRL_EINVAL, "NULL handle");
  int rc = set_device(h);
  if (rc) return rc;
  for (int k = 0; k < h->nnets; ++k)
    if ((rc = ewc_setup_one(h, h->net[k], mask))) return rc;
  return DDRL_OK;
}

extern "C" int ddrl_pop_ewc_setup_net(ddrl_pop_t *h, int net, const uint8_t *mask) {
  if (!h || !trainable_net(h, net)) return ddrl_fail(DDRL_EINVAL, "bad argument");
  int rc = set_device(h);
  if (rc) return rc;
  return ewc_setup_one(h, *train_net(h, net), mask);
}

extern "C" int ddrl_pop_ewc_decay_update(ddrl_pop_t *h, const uint8_t *mask) {
  if (!h) return ddrl_fail(DDRL_EINVAL, "NULL handle");
  int rc = set_device(h);
  if (rc) return rc;
  for (int k = 0; k < h->nnets; ++k) {
    TrainNet &t = h->net[k];
    if (!t.ewc.enabled || !(h->cfg.ewc_decay >= 0.)) continue;   // DODevMLP.hpp:709-712
    for (int g = 0; g < h->G; ++g)
      if (!mask || mask[g]) t.ewc.decay_mult[g] *= h->cfg.ewc_decay;
    if ((rc = refresh_ewc_coef(h, t))) return rc;
  }
  return DDRL_OK;
}

extern "C" int ddrl_pop_ewc_cost(ddrl_pop_t *h, int net, double *cost) {
  if (!h || !cost || !trainable_net(h, net)) return ddrl_fail(DDRL_EINVAL, "bad argument");
  int rc = set_device(h);
  if (rc) return rc;
  TrainNet &t = *train_net(h, net);
  std::fill(cost, cost + h->G, 0.0);
  if (!t.ewc.enabled) return DDRL_OK;
  HIPCHK(ewc_cost_launch(t.st.w, t.st.wstar, t.st.fhat, t.nd.stride(), h->d_cost, h->G, h->stream));
  std::vector<float> c(h->G);
  HIPCHK(hipMemcpyAsync(c.data(), h->d_cost, h->G * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  DDRL_WAIT(h);
  int learnable = t.nd.np;                                  // the k of DODevMLP.hpp:679: parameters with lr_mult != 0 (not a tower's statistics)
  for (int r = 0; r <= t.nd.L(); ++r) if (t.nd.bn(r)) learnable -= 2 * t.nd.bn_c(r) + 1;
  for (int g = 0; g < h->G; ++g) {
    if (!t.ewc.live[g]) continue;
    double r = (double)c[g] * h->cfg.ewc / (double)learnable;             // DODevMLP.hpp:679-683
    cost[g] = h->cfg.ewc_decay < 0. ? r : r * t.ewc.decay_mult[g];
  }
  return DDRL_OK;
}

extern "C" int ddrl_pop_get_ewc(ddrl_pop_t *h, int net, double *fisher, double *fisher_hat, double *wstar,
                                double *fisher_nbr, double *decay_mult) {
  if (!h || !trainable_net(h, net)) return ddrl_fail(DDRL_EINVAL, "bad argument");
  int rc = set_device(h);
  if (rc) return rc;
  TrainNet &t = *train_net(h, net);
  if (!t.ewc.enabled) return ddrl_fail(DDRL_ESTATE, "EWC is not enabled for this net");
  if (fisher && (rc = download_flat(h, t.nd, t.st.fisher, fisher))) return rc;
  if (fisher_hat && (rc = download_flat(h, t.nd, t.st.fhat, fisher_hat))) return rc;
  if (wstar && (rc = download_flat(h, t.nd, t.st.wstar, wstar))) return rc;
  if (fisher_nbr) {
    std::vector<float> f(h->G);
    HIPCHK(hipMemcpyAsync(f.data(), t.st.fisher_nbr, h->G * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    DDRL_WAIT(h);
    for (int g = 0; g < h->G; ++g) fisher_nbr[g] = f[g];
  }
  if (decay_mult) for (int g = 0; g < h->G; ++g) decay_mult[g] = t.ewc.decay_mult[g];
  return DDRL_OK;
}

// ---- run state: save_run / load_previous_run (DeepQNAg.hpp:504-537) for the whole population ----
int ddrl_pop_checkpoint(ddrl_pop *h, Ckpt &c) {
  c.st = h->stream;
  int rc;
  const int B = h->algo == 0 ? h->cfg.mini_batch_size : 0, cap = h->algo == 0 ? h->cfg.replay_memory : 0;
  const int meta[8] = {h->algo, h->G, h->cfg.nb_sensors, h->cfg.nb_motors, h->nnets, B, cap, h->rowlen};
  if ((rc = c.same("pop.meta", meta, sizeof(meta)))) return rc;
  const int nn = h->algo == 2 ? 1 : h->nnets;
  for (int k = 0; k < nn; ++k) {
    char tag[48];
    snprintf(tag, sizeof(tag), "net%d.shape", k);
    if ((rc = c.same(tag, &h->net[k].nd, sizeof(NetDesc)))) return rc;
  }
  const size_t G = h->G;
  for (int k = 0; k < nn; ++k) {
    TrainNet &t = h->net[k];
    const size_t n = G * t.nd.stride() * sizeof(float);
    char tag[48];
    auto T = [&](const char *what) { snprintf(tag, sizeof(tag), "net%d.%s", k, what); return tag; };
    if ((rc = c.dev(T("w"), t.st.w, n))) return rc;
    if (t.st.wt && (rc = c.dev(T("wt"), t.st.wt, n))) return rc;
    if (t.st.m) {
      if ((rc = c.dev(T("m"), t.st.m, n)) || (rc = c.dev(T("v"), t.st.v, n)) || (rc = c.dev(T("iter"), t.st.iter, G * sizeof(int))) ||
          (rc = c.dev(T("lr"), t.d_lr, G * sizeof(float))) || (rc = c.dev(T("decay"), t.d_decay, G * sizeof(float)))) return rc;
    }
    int ewc = t.ewc.enabled ? 1 : 0;
    if ((rc = c.same(T("ewc"), &ewc, sizeof(ewc)))) return rc;
    if (ewc) {
      if ((rc = c.dev(T("fisher"), t.st.fisher, n)) || (rc = c.dev(T("fhat"), t.st.fhat, n)) || (rc = c.dev(T("wstar"), t.st.wstar, n)) ||
          (rc = c.dev(T("best"), t.st.best, n)) || (rc = c.dev(T("fisher_nbr"), t.st.fisher_nbr, G * sizeof(float))) ||
          (rc = c.dev(T("ewc_coef"), t.st.ewc_coef, G * sizeof(float))) ||
          (rc = c.vec(T("best_score"), t.ewc.best_score)) || (rc = c.vec(T("decay_mult"), t.ewc.decay_mult)) ||
          (rc = c.vec(T("last_update_best"), t.ewc.last_update_best)) || (rc = c.vec(T("best_valid"), t.ewc.best_valid)) ||
          (rc = c.vec(T("live"), t.ewc.live))) return rc;
      if (t.ewc.best_score.size() != G || t.ewc.decay_mult.size() != G || t.ewc.last_update_best.size() != G ||
          t.ewc.best_valid.size() != G || t.ewc.live.size() != G) return ddrl_fail(DDRL_EIO, "%s: EWC bookkeeping of the wrong length", c.path.c_str());
    }
  }
  if (h->d_tau && ((rc = c.dev("tau", h->d_tau, G * sizeof(float))) || (rc = c.dev("omtau", h->d_omtau, G * sizeof(float))))) return rc;
  if (h->d_gamma && (rc = c.dev("gamma", h->d_gamma, G * sizeof(float)))) return rc;
  if (h->algo == 0) {
    // the replay rings (the reference's continue.trajectory.data) and the sampling engines
    if ((rc = c.dev("replay", h->d_rep, G * cap * h->rowlen * sizeof(float))) || (rc = c.dev("replay.len", h->d_len, G * sizeof(int))) ||
        (rc = c.dev("replay.head", h->d_head, G * sizeof(int))) || (rc = c.dev("mt19937", h->d_mt, G * 624 * sizeof(uint32_t))) ||
        (rc = c.dev("mt19937.pos", h->d_mtpos, G * sizeof(int))) || (rc = c.vec("replay.h_len", h->h_len)) ||
        (rc = c.vec("replay.h_head", h->h_head))) return rc;
    if (h->h_len.size() != G || h->h_head.size() != G) return ddrl_fail(DDRL_EIO, "%s: replay bookkeeping of the wrong length", c.path.c_str());
  }
  int gates = h->gates_on ? 1 : 0;
  if ((rc = c.val("gates.on", gates))) return rc;
  if (gates) {
    if (!h->d_sgate) {                                   // reading into a handle that never had gates: allocate them
      std::vector<double> one(G * (size_t)std::max(h->cfg.nb_sensors, h->cfg.nb_motors), 1.0);
      if ((rc = ddrl_pop_set_gates(h, one.data(), one.data()))) return rc;
    }
    if ((rc = c.dev("gates.state", h->d_sgate, G * h->cfg.nb_sensors * sizeof(float))) ||
        (rc = c.dev("gates.action", h->d_agate, G * h->cfg.nb_motors * sizeof(float)))) return rc;
  }
  if (!c.writing) {
    h->gates_on = gates != 0;
    h->gate_w_s.clear(); h->gate_w_a.clear();
    if (gates) {          // the flat view's gate weights are not in the file: they restart from the factors the device holds
      h->gate_w_s.resize(G * h->cfg.nb_sensors); h->gate_w_a.resize(G * h->cfg.nb_motors);
      if ((rc = ddrl_pop_get_gates(h, h->gate_w_s.data(), h->gate_w_a.data()))) return rc;
    }
  }
  {
    int g6[6] = {h->gm.st_bern, h->gm.st_scale, h->gm.ac_bern, h->gm.ac_scale, (int)h->gm.seed, (int)h->gm.draw};
    for (int i = 0; i < 6; ++i) {
      char key[24];
      snprintf(key, sizeof key, "gates.mode%d", i);
      if ((rc = c.val(key, g6[i]))) return rc;
    }
    if (!c.writing) {
      h->gm.st_bern = g6[0]; h->gm.st_scale = g6[1]; h->gm.ac_bern = g6[2]; h->gm.ac_scale = g6[3];
      h->gm.seed = (unsigned)g6[4]; h->gm.draw = (unsigned)g6[5];
      h->P.gm = h->gm;
    }
  }
  if (h->nnets == 3 && ((rc = c.dev("td3.noise_seed", h->d_nseed, G * sizeof(uint64_t))) ||
                        (rc = c.dev("td3.noise_count", h->d_ncount, G * sizeof(uint32_t))))) return rc;
  if (h->algo == 1 && (rc = ddrl_nfac_checkpoint(h, c))) return rc;
  return DDRL_OK;
}

static int pop_run_file(ddrl_pop *h, const char *path, bool save) {
  if (!h || !path) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  int rc = set_device(h);
  if (rc) return rc;
  DDRL_WAIT(h);
  Ckpt c;
  if ((rc = c.open(path, save)) || (rc = ddrl_pop_checkpoint(h, c))) return rc;
  if ((rc = c.finish())) return rc;
  if (!save && h->bn) {                      // the acting copy equals the actor between update calls (DeepQNAg.hpp:469-472)
    if ((rc = refresh_testing(h))) return rc;
    DDRL_WAIT(h);
  }
  return DDRL_OK;
}
extern "C" int ddrl_pop_save_run(ddrl_pop_t *h, const char *path) { return pop_run_file(h, path, true); }
extern "C" int ddrl_pop_load_previous_run(ddrl_pop_t *h, const char *path) { return pop_run_file(h, path, false); }

// ---- measurement hooks ----
extern "C" int ddrl_pop_set_profiling(ddrl_pop_t *h, int on) {
  if (!h) return ddrl_fail(DDRL_EINVAL, "NULL handle");
  h->profiling = on != 0;
  return DDRL_OK;
}

extern "C" int ddrl_pop_profile_read(ddrl_pop_t *h, double *kernel_ms, int *launches) {
  if (!h || !kernel_ms || !launches) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  int rc = set_device(h);
  if (rc) return rc;
  DDRL_WAIT(h);
  double tot = 0.;
  for (size_t i = 0; i < h->ev_used; ++i) {
    ddrl_f32 ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, h->ev_pool[i].first, h->ev_pool[i].second));
    tot += ms;
  }
  *kernel_ms = tot;
  *launches = (int)h->ev_used;
  h->ev_used = 0;
  return DDRL_OK;
}

// diagnostic: s_memtime stamps of the next/last update's phases, [G x 16] (enable, run an update, read)
extern "C" int ddrl_pop_debug_stamps(ddrl_pop_t *h, int enable, long long *out) {
  if (!h) return ddrl_fail(DDRL_EINVAL, "NULL handle");
  int rc = set_device(h);
  if (rc) return rc;
  h->P.stamps = enable ? h->d_stamps : nullptr;
  if (out) {
    HIPCHK(hipMemcpyAsync(out, h->d_stamps, (size_t)h->G * 16 * sizeof(long long), hipMemcpyDeviceToHost, h->stream));
    DDRL_WAIT(h);
  }
  return DDRL_OK;
}

// options: "force_generic" (0/1): run the run-time-shape kernel even when a compile-time
// instantiation matches (parity tests exercise both on the same inputs)
extern "C" int ddrl_pop_set_option(ddrl_pop_t *h, const char *key, int value) {
  if (!h || !key) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  if (std::string(key) == "force_generic") { h->force_generic = value != 0; return DDRL_OK; }
  if (std::string(key) == "nfac_fuse") { h->nfac_fuse = value != 0; return DDRL_OK; }
  if (std::string(key) == "resident") { h->resident_mode = value != 0; return DDRL_OK; }
  if (std::string(key) == "debug_raise_update_error") {      // test hook: what an expired hand-off / team-barrier wait leaves behind
    const int one = value != 0;
    HIPCHK(hipMemcpyAsync(h->P.upd_error, &one, sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->multi_pending = true;
    return DDRL_OK;
  }
  if (std::string(key) == "loop") { h->loop_mode = value < 0 ? 0 : value > 2 ? 2 : value; return DDRL_OK; }
  if (std::string(key) == "w8") { h->w8_mode = value < 0 ? 0 : value > 2 ? 2 : value; return DDRL_OK; }
  if (std::string(key) == "pair") { h->pair_mode = value < 0 ? 0 : value > 2 ? 2 : value; return DDRL_OK; }
  if (std::string(key) == "force_chunked") {     // the gradient + apply kernels of the large-minibatch path on one-tile minibatches
    DDPG_ONLY(h);
    h->force_chunked = value != 0;
    int rc = set_device(h);
    if (rc) return rc;
    if ((rc = chunk_buffers(h))) return rc;
    DDRL_WAIT(h);
    return DDRL_OK;
  }
  if (std::string(key) == "force_wide") {        // the layer-by-layer kernels on nets the fused kernel would take
    DDPG_ONLY(h);
    bool need = h->cfg.nb_sensors + h->cfg.nb_motors > DDRL_MAXW;
    for (int k = 0; k < 2; ++k)
      for (int l = 1; l < h->net[k].nd.L(); ++l) need = need || h->net[k].nd.d(l) > DDRL_MAXW;
    h->wide = need || value != 0;
    int rc = set_device(h);
    if (rc) return rc;
    if ((rc = chunk_buffers(h))) return rc;
    DDRL_WAIT(h);
    return DDRL_OK;
  }
  return ddrl_fail(DDRL_EINVAL, "unknown option '%s'", key);
}
