// pop_host.h -- host-side handle of libddrl_pop.so and the helpers shared by pop.hip (DDPG) and
// nfac.hip (NFAC(lambda)-V).  Not part of the C ABI.
#pragma once
#include "pop_internal.h"            // first: the fp64 debug build retypes `float` for everything after it
#include "../../include/ddrl_pop.h"
#include "host_util.h"
#include "ckpt.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <random>
#include <string>
#include <vector>

struct NfacHost;   // nfac.hip
void ddrl_nfac_host_free(NfacHost *n);
struct ddrl_pop;
// one traversal per object writes or reads its run state (ckpt.h); they nest: an agent front end's
// file holds its population's records and, for CMA-ES, the optimiser's
int ddrl_pop_checkpoint(ddrl_pop *h, Ckpt &c);          // pop.hip
int ddrl_nfac_checkpoint(ddrl_pop *h, Ckpt &c);         // nfac.hip
int ddrl_cmaes_checkpoint(ddrl_cmaes_t *h, Ckpt &c);    // cmaes.hip

// ---------------------------------------------------------------------------------------------
// per-net host bookkeeping of the EWC life-cycle (DODevMLP.hpp:803-822)
struct EwcHost {
  bool enabled = false;
  std::vector<double> best_score, decay_mult;
  std::vector<int> last_update_best;
  std::vector<uint8_t> best_valid, live;
};

struct TrainNet {
  NetDesc nd;
  NetState st;           // device pointers
  EwcHost ewc;
  float *d_lr = nullptr, *d_decay = nullptr;
};

struct ddrl_pop {
  int algo = 0;                    // 0 DDPG, 1 NFAC, 2 policies only (CMA-ES individuals)
  NfacHost *nfac = nullptr;        // algo == 1
  ddrl_ddpg_config cfg;            // NFAC handles fill the fields the shared entry points read
  int G = 0, device = 0;
  hipStream_t stream = nullptr;
  TrainNet net[3];                 // 0 actor, 1 critic, 2 second critic (TD3)  (targets hang off st.wt)
  int nnets = 2;
  // TD3: smoothing-noise streams and staging
  uint64_t *d_nseed = nullptr;
  uint32_t *d_ncount = nullptr;
  float *d_noise = nullptr, *d_loss2 = nullptr;
  uint8_t *d_astep = nullptr;
  // replay
  int rowlen = 0;
  float *d_rep = nullptr;
  int *d_len = nullptr, *d_head = nullptr;
  std::vector<int> h_len, h_head;
  uint32_t *d_mt = nullptr;
  int *d_mtpos = nullptr;
  int *d_idx = nullptr;
  size_t idx_cap = 0;              // ints d_idx holds (grows for multi-update launches)
  // developmental gates (null until ddrl_pop_set_gates)
  float *d_sgate = nullptr, *d_agate = nullptr;
  GateMode gm = {0, 0, 0, 0, 0u, 0u};              // ddrl_pop_set_gate_mode; gm.draw counts the passes drawn so far
  bool gates_on = false;
  // hyper
  float *d_tau = nullptr, *d_omtau = nullptr, *d_gamma = nullptr;
  // diagnostics
  float *d_loss = nullptr, *d_qt = nullptr, *d_dact = nullptr;
  long long *d_stamps = nullptr;
  // staging
  float *d_s = nullptr, *d_a = nullptr, *d_r = nullptr, *d_s2 = nullptr, *d_act = nullptr, *d_cost = nullptr;
  uint8_t *d_goal = nullptr, *d_mask = nullptr, *d_flag = nullptr;
  std::vector<void *> allocs;
  DdpgParams P;
  // measurement
  bool profiling = false;
  int force_generic = 0;
  bool multi_pending = false;   // a multi-update launch was issued since the error word was last read (ddrl_pop_wait)
  int resident_mode = 0;      // 1: small nets keep an agent's whole state in LDS for a multi-update launch (ddpg_update_resident_kernel)
  int nfac_fuse = 0;          // NFAC, option "nfac_fuse": 1 = the optimiser step inside the gradient launch (bit-identical; measured SLOWER, kernels_nfac.hip), 0 = separate nfac_apply_kernel launches
  int pair_mode = 0;          // 0: never the two-agents-per-workgroup kernel (the default since round 3: the loop kernel is faster), 1: for launches of >= DDRL_PAIR_MIN_UPD updates where pairs paid in round 2, 2: whenever it fits
  bool force_chunked = false;
  bool wide = false;               // some layer is wider than DDRL_MAXW: layer-by-layer update kernels
  bool bn = false;                 // some net has BatchNorm towers: kernels_ddpg_bn.hip
  float *d_atest = nullptr;        // ann_testing [G x stride]: the actor's parameters as of the last sync (DeepQNAg.hpp:469-472)
  double beta1 = 0., beta2 = 0.999;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  size_t ev_used = 0;
  // DODevMLP's gate blobs as part of the flat parameter view (ddrl_pop_set_gate_layout): controlled dimensions per gate
  // (-1 entries never; n < 0: no such layer), and whether the weights are learnable_params with lr_mult 1 (heuristic 0)
  std::vector<double> gate_w_s, gate_w_a;   // the gate WEIGHTS [G x S], [G x A] the flat view shows (the device holds the factors made of them)
  std::vector<int> st_control, ac_control;
  bool st_layer = false, ac_layer = false, gates_learnable = false;
  int w8_mode = 0;            // eight waves per agent (kernels_ddpg_w8.hip): 0 never (default: measured slower than four waves, 1.19 M against 1.26 M updates/s at 128 agents), 1 up to DDRL_W8_MAX_AGENTS agents, 2 always
  int loop_mode = 1;          // multi-update launches keep an agent's updates in one workgroup (ddpg_update_loop_kernel): 0 never, 1 where it pays (mid nets), 2 whenever the shapes are compile-time ones
};

#define HIPCHK(expr)                                                                        \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) return ddrl_fail(DDRL_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

template <class T>
static int dalloc(ddrl_pop *h, T **p, size_t n, bool zero = true) {
  void *q = nullptr;
  hipError_t e = hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T));
  if (e != hipSuccess) return ddrl_fail(DDRL_ENOMEM, "hipMalloc(%zu bytes): %s", n * sizeof(T), hipGetErrorString(e));
  h->allocs.push_back(q);
  if (zero) {
    e = hipMemsetAsync(q, 0, std::max<size_t>(n, 1) * sizeof(T), h->stream);
    if (e != hipSuccess) return ddrl_fail(DDRL_EHIP, "hipMemset: %s", hipGetErrorString(e));
  }
  *p = (T *)q;
  return DDRL_OK;
}

// bn_code: the batch_norm bit-field (MLP.hpp:53-79); critic_sensors: size_sensors for a critic (its first_except_action
// variant normalises the states only, :130-134, and it has no output tower), 0 for an actor
static void build_desc(NetDesc &nd, int d0, const int *hidden, int nh, int dout, int hidden_act, int out_act,
                       int bn_code = 0, int critic_sensors = 0) {
  memset(&nd, 0, sizeof(nd));
  nd.L_ = nh + 1;
  nd.d_[0] = d0;
  for (int i = 0; i < nh; ++i) nd.d_[i + 1] = hidden[i];
  nd.d_[nh + 1] = dout;
  int off = 0, np = 0;
  for (int l = 0; l < nd.L_; ++l) {
    nd.kp_[l] = ddrl_roundup(nd.d_[l], 4);
    nd.offw_[l] = off; off += ddrl_wsize(nd.d_[l + 1], nd.kp_[l]);
    nd.offb_[l] = off; off += ddrl_roundup(nd.d_[l + 1], 4);
    np += nd.d_[l + 1] * nd.d_[l] + nd.d_[l + 1];
  }
  for (int i = 0; i <= nd.L_; ++i) nd.ld_[i] = ddrl_ldpad(nd.d_[i]);
  // towers: convertBN + where the constructors place them (Tower(): rank i - 1 before hidden InnerProduct i, :756-758; rank H
  // on the tower's top, :141, :220; rank H + 1 on the actor's output for very_all, :238-240)
  const int H = nh;
  const bool b2 = bn_code & 4, b3 = bn_code & 8, b4 = bn_code & 16;
  nd.bn_flags_ = bn_code & 3;
  for (int r = 0; r <= nd.L_; ++r) nd.bn_c_[r] = nd.d_[r];
  if (b4 && !b2) { nd.bn_mask_ = 1; nd.bn_c_[0] = critic_sensors; }                          // first_except_action
  else if (b4) nd.bn_mask_ = ((1 << (H + 1)) - 1) | (critic_sensors == 0 ? 1 << nd.L_ : 0);  // very_all
  else if (b2 && b3) nd.bn_mask_ = (1 << (H + 1)) - 1;                                       // all
  else if (b2) nd.bn_mask_ = 1;                                                              // first
  else if (b3) nd.bn_mask_ = (1 << H) - 1;                                                   // all_except_last
  for (int r = 0; r <= nd.L_; ++r)
    if (nd.bn(r)) {
      nd.bn_off_[r] = off; off += nd.bn_floats(r);
      np += 2 * nd.bn_c(r) + 1 + (nd.bn_scale() ? nd.bn_c(r) : 0) + (nd.bn_bias() ? nd.bn_c(r) : 0);
    }
  nd.stride_ = ddrl_roundup(off, 64);
  nd.np = np;
  nd.hidden_act_ = hidden_act;
  nd.out_act_ = out_act;
}

// flat reference order <-> padded device image of ONE agent.  Reference order = learnable_params() (MLP.hpp:610-676):
// the net's layers in order, a tower's [mean_sum, var_sum, factor][gamma][beta] ahead of the InnerProduct it feeds.
static void pack_flat(const NetDesc &nd, const double *flat, float *img) {
  size_t k = 0;
  for (int l = 0; l <= nd.L(); ++l) {
    if (nd.bn(l)) {
      const int C = nd.bn_c(l), C4 = nd.bn_c4(l);
      float *t = img + nd.bn_off(l);
      for (int c = 0; c < C; ++c) t[c] = (float)flat[k++];
      for (int c = 0; c < C; ++c) t[C4 + c] = (float)flat[k++];
      t[2 * C4] = (float)flat[k++];
      if (nd.bn_scale()) for (int c = 0; c < C; ++c) t[2 * C4 + 4 + c] = (float)flat[k++];
      if (nd.bn_bias()) for (int c = 0; c < C; ++c) t[3 * C4 + 4 + c] = (float)flat[k++];
    }
    if (l == nd.L()) break;
    for (int n = 0; n < nd.d(l + 1); ++n)
      for (int c = 0; c < nd.d(l); ++c) img[nd.offw(l) + ddrl_widx(n, c, nd.kp(l))] = (float)flat[k++];
    for (int n = 0; n < nd.d(l + 1); ++n) img[nd.offb(l) + n] = (float)flat[k++];
  }
}
static void unpack_flat(const NetDesc &nd, const float *img, double *flat) {
  size_t k = 0;
  for (int l = 0; l <= nd.L(); ++l) {
    if (nd.bn(l)) {
      const int C = nd.bn_c(l), C4 = nd.bn_c4(l);
      const float *t = img + nd.bn_off(l);
      for (int c = 0; c < C; ++c) flat[k++] = (double)t[c];
      for (int c = 0; c < C; ++c) flat[k++] = (double)t[C4 + c];
      flat[k++] = (double)t[2 * C4];
      if (nd.bn_scale()) for (int c = 0; c < C; ++c) flat[k++] = (double)t[2 * C4 + 4 + c];
      if (nd.bn_bias()) for (int c = 0; c < C; ++c) flat[k++] = (double)t[3 * C4 + 4 + c];
    }
    if (l == nd.L()) break;
    for (int n = 0; n < nd.d(l + 1); ++n)
      for (int c = 0; c < nd.d(l); ++c) flat[k++] = (double)img[nd.offw(l) + ddrl_widx(n, c, nd.kp(l))];
    for (int n = 0; n < nd.d(l + 1); ++n) flat[k++] = (double)img[nd.offb(l) + n];
  }
}
// the flat positions a fresh net leaves at the fillers' constants instead of N(0, 0.01): tower statistics 0, gamma 1, beta 0
static void fill_bn_defaults(const NetDesc &nd, double *flat) {
  size_t k = 0;
  for (int l = 0; l <= nd.L(); ++l) {
    if (nd.bn(l)) {
      const int C = nd.bn_c(l);
      for (int c = 0; c < 2 * C + 1; ++c) flat[k++] = 0.;
      if (nd.bn_scale()) for (int c = 0; c < C; ++c) flat[k++] = 1.;
      if (nd.bn_bias()) for (int c = 0; c < C; ++c) flat[k++] = 0.;
    }
    if (l == nd.L()) break;
    k += (size_t)nd.d(l + 1) * nd.d(l) + nd.d(l + 1);
  }
}

static inline int ddrl_pop_wait(ddrl_pop *h);      // (below) the stream, and the multi-update launches' error word

// every transfer of state through these helpers waits with ddrl_pop_wait: a launch that skipped updates (an expired
// hand-off / team-barrier wait) must not be read back as healthy weights, moments or EWC arrays
static int upload_flat(ddrl_pop *h, const NetDesc &nd, float *dst, const double *flat) {
  std::vector<float> img((size_t)h->G * nd.stride(), 0.f);
  for (int g = 0; g < h->G; ++g) pack_flat(nd, flat + (size_t)g * nd.np, img.data() + (size_t)g * nd.stride());
  HIPCHK(hipMemcpyAsync(dst, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
  return ddrl_pop_wait(h);
}
static int download_flat(ddrl_pop *h, const NetDesc &nd, const float *src, double *flat) {
  std::vector<float> img((size_t)h->G * nd.stride());
  HIPCHK(hipMemcpyAsync(img.data(), src, img.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  { const int rc = ddrl_pop_wait(h); if (rc) return rc; }
  for (int g = 0; g < h->G; ++g) unpack_flat(nd, img.data() + (size_t)g * nd.stride(), flat + (size_t)g * nd.np);
  return DDRL_OK;
}

template <class T>
static int upload_vec(ddrl_pop *h, T *dst, const std::vector<T> &v) {
  HIPCHK(hipMemcpyAsync(dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, h->stream));
  return ddrl_pop_wait(h);
}

static int upload_mask(ddrl_pop *h, const uint8_t *mask, const uint8_t **dmask) {
  *dmask = nullptr;
  if (!mask) return DDRL_OK;
  HIPCHK(hipMemcpyAsync(h->d_mask, mask, h->G, hipMemcpyHostToDevice, h->stream));
  *dmask = h->d_mask;
  return DDRL_OK;
}

static int set_device(ddrl_pop *h) {
  HIPCHK(hipSetDevice(h->device));
  return DDRL_OK;
}

static TrainNet *train_net(ddrl_pop *h, int net) {
  if (net == DDRL_NET_ACTOR || net == DDRL_NET_ACTOR_TARGET) return &h->net[0];
  return (net == DDRL_NET_CRITIC2 || net == DDRL_NET_CRITIC2_TARGET) ? &h->net[2] : &h->net[1];
}
static float *net_weights(ddrl_pop *h, int net) {
  TrainNet *t = train_net(h, net);
  return (net == DDRL_NET_ACTOR || net == DDRL_NET_CRITIC || net == DDRL_NET_CRITIC2) ? t->st.w : t->st.wt;
}
// nets with an optimiser (Adam state, EWC arrays)
static bool trainable_net(ddrl_pop *h, int net) {
  return net == DDRL_NET_ACTOR || net == DDRL_NET_CRITIC || (net == DDRL_NET_CRITIC2 && h->nnets == 3);
}

static int refresh_ewc_coef(ddrl_pop *h, TrainNet &t) {
  if (!t.ewc.enabled) return DDRL_OK;
  std::vector<float> c(h->G);
  for (int g = 0; g < h->G; ++g) {
    double f = h->cfg.ewc_decay >= 0. ? t.ewc.decay_mult[g] : 1.;   // DODevMLP.hpp:691-693
    c[g] = t.ewc.live[g] ? (float)(h->cfg.ewc * f) : 0.f;
  }
  return upload_vec(h, t.st.ewc_coef, c);
}


static int alloc_net(ddrl_pop *h, TrainNet &t, bool ewc, bool target = true) {
  const size_t n = (size_t)h->G * t.nd.stride();
  int rc;
  if ((rc = dalloc(h, &t.st.w, n))) return rc;
  if ((rc = dalloc(h, &t.st.m, n))) return rc;
  if ((rc = dalloc(h, &t.st.v, n))) return rc;
  if (target && (rc = dalloc(h, &t.st.wt, n))) return rc;
  if ((rc = dalloc(h, &t.st.iter, h->G))) return rc;
  if ((rc = dalloc(h, &t.d_lr, h->G))) return rc;
  if ((rc = dalloc(h, &t.st.lr_t, (size_t)h->G * DDRL_MAXUPD))) return rc;
  if ((rc = dalloc(h, &t.d_decay, h->G))) return rc;
  t.st.lr = t.d_lr;
  t.st.decay = t.d_decay;
  t.ewc.enabled = ewc;
  if (ewc) {
    if ((rc = dalloc(h, &t.st.fisher, n))) return rc;
    if ((rc = dalloc(h, &t.st.fhat, n))) return rc;
    if ((rc = dalloc(h, &t.st.wstar, n))) return rc;
    if ((rc = dalloc(h, &t.st.best, n))) return rc;
    if ((rc = dalloc(h, &t.st.fisher_nbr, h->G))) return rc;
    if ((rc = dalloc(h, &t.st.ewc_coef, h->G))) return rc;
    t.ewc.best_score.assign(h->G, std::numeric_limits<double>::lowest());
    t.ewc.decay_mult.assign(h->G, 1.0);
    t.ewc.last_update_best.assign(h->G, 0);
    t.ewc.best_valid.assign(h->G, 0);
    t.ewc.live.assign(h->G, 0);
  }
  return DDRL_OK;
}


// NFAC populations have no target nets (OfflineCaclaAg.hpp:186-196)
static bool bad_net(ddrl_pop *h, int net) {
  return net < 0 || net > DDRL_NET_CRITIC2_TARGET || (net > DDRL_NET_CRITIC_TARGET && h->nnets < 3) ||
         (h->algo == 1 && net > DDRL_NET_CRITIC) || (h->algo == 2 && net != DDRL_NET_ACTOR);
}
// Waits for the handle's stream.  After a multi-update launch (ddrl_pop_ddpg_update_n) it also looks at the launch's error
// word: a workgroup that gave up waiting for its agent's previous update (kernels_ddpg.hip) or a team barrier that
// expired (kernels_ddpg_pair.hip) left updates undone, and NO reader of the state (weights, Adam state, diagnostics,
// run files) may then return it as if it were healthy.
static inline int ddrl_pop_wait(ddrl_pop *h) {
  hipError_t e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) return ddrl_fail(DDRL_EHIP, "hipStreamSynchronize: %s", hipGetErrorString(e));
  if (h->multi_pending && h->P.upd_error) {
    h->multi_pending = false;
    int err = 0;
    e = hipMemcpy(&err, h->P.upd_error, sizeof(int), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return ddrl_fail(DDRL_EHIP, "hipMemcpy: %s", hipGetErrorString(e));
    if (err) {
      (void)hipMemset(h->P.upd_error, 0, sizeof(int));
      return ddrl_fail(DDRL_ESTATE, "a multi-update launch gave up waiting for an agent's previous update: some updates were skipped");
    }
  }
  return DDRL_OK;
}
#define DDRL_WAIT(h) do { int rc_w_ = ddrl_pop_wait(h); if (rc_w_) return rc_w_; } while (0)

// the gate mode for the next launch, and `passes` draws consumed
static inline GateMode next_draw(ddrl_pop *h, int passes) {
  const GateMode g = h->gm;
  h->gm.draw += (unsigned)passes;
  return g;
}
static inline bool bernoulli_gates(const ddrl_pop *h) { return h->gm.st_bern || h->gm.ac_bern; }

// ann_testing->copyWeightsFrom(ann->copyWeightsTo) (DeepQNAg.hpp:469-472): every parameter, statistics included
static inline int refresh_testing(ddrl_pop *h) {
  if (!h->d_atest) return DDRL_OK;
  HIPCHK(hipMemcpyAsync(h->d_atest, h->net[0].st.w, (size_t)h->G * h->net[0].nd.stride() * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
  return DDRL_OK;
}

// measurement hook (ddrl_pop_set_profiling): a HIP event pair around one launch on the handle's stream
static inline int prof_begin(ddrl_pop *h, hipEvent_t *e1) {
  *e1 = nullptr;
  if (!h->profiling) return DDRL_OK;
  hipEvent_t e0 = nullptr;
  if (h->ev_used == h->ev_pool.size()) {
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(e1));
    h->ev_pool.emplace_back(e0, *e1);
  }
  e0 = h->ev_pool[h->ev_used].first; *e1 = h->ev_pool[h->ev_used].second;
  h->ev_used++;
  HIPCHK(hipEventRecord(e0, h->stream));
  return DDRL_OK;
}
static inline int prof_end(ddrl_pop *h, hipEvent_t e1) {
  if (e1) HIPCHK(hipEventRecord(e1, h->stream));
  return DDRL_OK;
}

#define DDPG_ONLY(h) do { if ((h)->algo != 0) return ddrl_fail(DDRL_ESTATE, "%s: DDPG populations only", __func__); } while (0)

