// kernels_ddpg_bn.hip -- the DDPG update for nets with BatchNorm / Scale towers (agent.batch_norm_actor / _critic != 0).
//
// One 256-thread workgroup per agent runs one pass of the fupd loop of DeepQNAg::end_episode
// (/root/reference/agent/deepqn/include/DeepQNAg.hpp:343-466) on a minibatch of up to 64 rows, like kernels_ddpg.hip,
// with the towers of bn_device.h between the InnerProduct passes:
//   * every TRAIN-phase forward moves its net's running statistics -- the targets' too unless agent.test_net (:216-234), the
//     critic's twice per update (learn_batch, then Q(s, pi(s)));
//   * parameter gradients are stored to a per-agent image and applied by a sweep over the agent's block (Adam + L2 on
//     InnerProducts and Scales, nothing on the statistics: lr_mult 0), the soft updates cover every blob, statistics
//     included (MLP.hpp:437-452), and run where the reference runs them -- after both nets have stepped;
//   * agent.bn_adapt (:415-420): the actions fed to the critic come from ann_testing, a TEST-phase copy holding the actor's
//     parameters as of the end of the previous update call (:469-472).
// Towers couple the rows of a minibatch, so this path has no chunked variant: mini_batch_size <= 64.
#include "ddpg_device.h"
#include "bn_device.h"

struct BnLds {
  LdsMap M;
  BnBufs a, q;          // x_norm copies / denominators of the actor's and the critic's towers (hidden tiles shared)
  float *pa;            // bn_adapt: the TEST-phase copy's actions [64 x lda]
};

static __host__ __device__ size_t bn_lds_floats(const DdpgParams &P) {
  const int nh = (P.a.L_ > P.q.L_ ? P.a.L_ : P.q.L_) - 1;
  const int ldq = P.q.ld_top(), lda = P.a.ld_top();
  size_t f = 2 * (size_t)nh * DDRL_MB * P.ldh;                        // hidden tiles + their x_norm copies
  f += 2 * (size_t)DDRL_MB * (P.ldxc + P.ldxa) + DDRL_NWAVES * 64;   // input tiles + copies, dump
  f += (size_t)DDRL_MB * (ldq + 3 * lda + 5);                         // topq, topa, its x_norm copy, pa, per-row scalars
  f += 2 * (size_t)(DDRL_MAXL + 1) * DDRL_MAXW;                       // denominators
  return f;
}

__device__ __forceinline__ BnLds bn_carve(float *p, const DdpgParams &P, int ldq, int lda) {
  BnLds Z;
  const int nh = (P.a.L_ > P.q.L_ ? P.a.L_ : P.q.L_) - 1;
  Z.M = ddpg_carve(p, P.a.L_, P.q.L_, P.ldh, P.ldxc, P.ldxa, ldq, lda);
  p = (float *)(Z.M.nvalid + DDRL_MB);
  float *xh = p; p += (size_t)nh * Z.M.hstride;
  Z.a.h = xh; Z.q.h = xh;
  Z.q.x0 = p; p += DDRL_MB * P.ldxc;
  Z.a.x0 = p; p += DDRL_MB * P.ldxa;
  Z.a.top = p; p += DDRL_MB * lda;
  Z.q.top = nullptr;                                                  // a critic has no output tower
  Z.pa = p; p += DDRL_MB * lda;
  Z.a.den = p; p += (DDRL_MAXL + 1) * DDRL_MAXW;
  Z.q.den = p;
  return Z;
}

// Adam + L2 (+ the soft target update) over the learnable part of an agent's block, gradients from gout
template <bool EWC>
__device__ __forceinline__ void bn_apply(const NetDesc &nd, const OptCtx &o, const float *__restrict__ gout, int tid) {
  for (int i = tid * 4; i < nd.stride(); i += DDRL_NTHREADS * 4) {
    bool stats = false;
    for (int r = 0; r <= nd.L(); ++r)
      stats = stats || (nd.bn(r) && i >= nd.bn_off(r) && i < nd.bn_off(r) + nd.bn_stats_floats(r));
    if (stats) continue;
    OptRegs rg;
    opt_load4<EWC, true, false>(o, i, rg);
    opt_step4<EWC, true, false>(o, i, *(const f32x4 *)(gout + i), rg, true);
  }
}
// MLP::soft_update over the statistics blobs (they are in params(), MLP.hpp:441-451)
__device__ __forceinline__ void bn_soft_stats(const NetDesc &nd, const float *__restrict__ w, float *__restrict__ wt, float tau, float omtau, int tid) {
  for (int r = 0; r <= nd.L(); ++r)
    if (nd.bn(r))
      for (int i = nd.bn_off(r) + tid; i < nd.bn_off(r) + nd.bn_stats_floats(r); i += DDRL_NTHREADS) wt[i] = tau * w[i] + omtau * wt[i];
}

template <bool EWC_A, bool EWC_Q>
__global__ void __launch_bounds__(DDRL_NTHREADS)
ddpg_update_bn_kernel(const DdpgParams P) {
  extern __shared__ __align__(16) float lds[];
  const int agent = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint8_t live = P.mask ? P.mask[agent] : (uint8_t)1;
  const int len = P.rep_len[agent], head = P.rep_head[agent];
  const float gamma = P.gamma[agent], tau = P.tau[agent], omtau = P.omtau[agent];
  if (!live || len < 250 || len < P.B) return;             // DeepQNAg.hpp:338
  const NetDesc &na = P.a, &nq = P.q;
  const int S = P.S, A = P.A, B = P.B, ldxc = P.ldxc, ldxa = P.ldxa, ldq = nq.ld_top(), lda = na.ld_top();
  const BnLds Z = bn_carve(lds, P, ldq, lda);
  const LdsMap &M = Z.M;
  const int *idx = P.idx + (size_t)agent * B;
  float *w_a = P.sa.w + (size_t)agent * na.stride(), *w_at = P.sa.wt + (size_t)agent * na.stride();
  float *w_q = P.sq.w + (size_t)agent * nq.stride(), *w_qt = P.sq.wt + (size_t)agent * nq.stride();
  float *g_a = P.gpart_a + (size_t)agent * na.stride(), *g_q = P.gpart_q + (size_t)agent * nq.stride();
  const int tmode = P.test_net ? BN_TEST : BN_TRAIN;
  auto none = [](int, int, float) {};
  // Developmental gates (DODevMLP.hpp:246-316): the state gates are gather_tile's; the action layer sits behind the actor's last
  // layer (behind its output tower) and behind the critic's input.  Per-dimension factors or per-sample Bernoulli masks: every
  // layer instance draws under its own site id and a backward pass regenerates its forward pass's mask (ddrl_gate_factor).
  const float *agate = P.agate ? P.agate + (size_t)agent * A : nullptr;
  const int ac_bern = P.gm.ac_bern, ac_scale = P.gm.ac_scale;
  const unsigned gseed = P.gm.seed, gdraw = P.gm.draw;
  long long *stamps = P.stamps ? P.stamps + (size_t)agent * 16 : nullptr;       // (diagnostic: tools/stamps_bn.py)
#define BN_STAMP(i) do { if (stamps && tid == 0) stamps[i] = clock64(); } while (0)
  BN_STAMP(0);
  // actions into the critic's input tile: the actor's output gate, then the critic's action-input gate.  src keeps what the
  // actor's last layer (or output tower) wrote: the Developmental layer has a top blob of its own (DODevMLP.hpp:262-268), the
  // last activation's backward sees the ungated values
  auto actions_in = [&](const float *src, bool target) {
    for (int e = tid; e < DDRL_MB * A; e += DDRL_NTHREADS) {
      const int row = e / A, col = e - row * A;
      float a = src[row * lda + col];
      if (agate) {
        a *= ddrl_gate_factor(agate[col], ac_bern, ac_scale, gseed, gdraw, agent, target ? GS_ACTOR_T_OUT : GS_ACTOR_OUT, row, col);
        a *= ddrl_gate_factor(agate[col], ac_bern, ac_scale, gseed, gdraw, agent, target ? GS_CRITIC_T_A : GS_CRITIC_A, row, col);
      }
      M.xc[row * ldxc + S + col] = a;
    }
  };

  // ---- a' = ann_target(s'), q' = qnn_target(s', a')  (:379-383) ----
  if (P.nstep > 0) nstep_horizon(P, S, A, agent, idx, B, len, head, gamma, M, tid);
  gather_tile(P, S, A, ldxc, ldxa, agent, idx, B, head, true, M, tid);
  __syncthreads();
  fwd_net_bn(na, na.L(), w_at, tmode, B, 0, M.xa, ldxa, M.hbase, M.hstride, M.topa, lda, Z.a, wave, lane, tid);
  BN_STAMP(1);
  actions_in(M.topa, true);
  __syncthreads();
  fwd_net_bn(nq, nq.L(), w_qt, tmode, B, 0, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, Z.q, wave, lane, tid);
  BN_STAMP(2);
  // ---- TD targets (:386-405) ----
  gather_tile(P, S, A, ldxc, ldxa, agent, idx, B, head, false, M, tid);
  int trow;
  if (gather_owns_row(tid, trow)) {
    const float qn = M.topq[trow * ldq];
    M.yv[trow] = M.dn[trow] != 0.f ? M.rv[trow] : (P.nstep > 0 ? M.nrew[trow] : M.rv[trow]) + gamma * qn;
  }
  __syncthreads();
  if (P.qt_out && tid < B) P.qt_out[(size_t)agent * B + tid] = M.yv[tid];
  BN_STAMP(3);
  // ---- critic update: qnn->learn_batch(s, a, y, 1)  (:408; MLP.hpp:468-485) ----
  fwd_net_bn(nq, nq.L(), w_q, BN_TRAIN, B, 0, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, Z.q, wave, lane, tid);
  BN_STAMP(4);
  if (tid < DDRL_MB) {
    const float df = tid < B ? M.topq[tid * ldq] - M.yv[tid] : 0.f;
    float sq = df * df;
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if (tid == 0) P.loss_out[agent] = sq / (float)B * 0.5f;
    M.topq[tid * ldq] = df / (float)B;
  }
  __syncthreads();
  bwd_net_bn<DW_STORE>(nq, w_q, g_q, false, B, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, Z.q, 0, 0, none, wave, lane, tid);
  BN_STAMP(5);
  __syncthreads();                                          // (workgroup-scope fence: the sweep reads other threads' gradient stores)
  // EWC (DODevMLP.hpp:636-707: Fisher average of the squared gradients, penalty towards the previous task's parameters) rides
  // in the optimiser sweep as on the tower-free path; a tower's statistics take no part (lr_mult 0: not in the EWC vectors)
  {                                                         // (EWC_Q: agent.ewc >= 0 with ewc_for_critic)
    OptCtx oq = make_opt<EWC_Q>(P, P.sq, nq.stride(), agent, tau, omtau, (float)B, 0);
    oq.handoff = 0;
    bn_apply<EWC_Q>(nq, oq, g_q, tid);
  }
  __syncthreads();
  BN_STAMP(6);
  // ---- actor update (:411-459) ----
  gather_tile(P, S, A, ldxc, ldxa, agent, idx, B, head, false, M, tid);
  __syncthreads();
  const bool adapt = P.bn_adapt && na.bn_any();
  if (adapt) {   // :415-420, the TEST-phase copy's actions (it keeps no activations: the TRAIN pass below rebuilds the tiles)
    BnBufs nokeep = {nullptr, nullptr, nullptr, nullptr};
    fwd_net_bn(na, na.L(), const_cast<float *>(P.a_test) + (size_t)agent * na.stride(), BN_TEST, B, 0, M.xa, ldxa, M.hbase, M.hstride,
               M.topa, lda, nokeep, wave, lane, tid);
    for (int e = tid; e < DDRL_MB * lda; e += DDRL_NTHREADS) Z.pa[e] = M.topa[e];
    __syncthreads();
    gather_tile(P, S, A, ldxc, ldxa, agent, idx, B, head, false, M, tid);
    __syncthreads();
  }
  fwd_net_bn(na, na.L(), w_a, BN_TRAIN, B, 0, M.xa, ldxa, M.hbase, M.hstride, M.topa, lda, Z.a, wave, lane, tid);   // pi = ann(s)
  BN_STAMP(7);
  float *acts = adapt ? Z.pa : M.topa;
  actions_in(acts, false);
  __syncthreads();
  fwd_net_bn(nq, nq.L(), w_q, BN_TRAIN, B, 0, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, Z.q, wave, lane, tid);   // Q(s, pi(s)), new critic
  BN_STAMP(8);
  if (tid < DDRL_MB) M.topq[tid * ldq] = tid < B ? -1.0f : 0.f;            // :424-428, NOT /B
  __syncthreads();
  {
    float *topa = M.topa;
    const int inverting = P.inverting, out_act = na.out_act();
    const bool top_tower = na.bn(na.L());
    float *dact_out = P.dact_out ? P.dact_out + (size_t)agent * B * A : nullptr;
    auto store = [=](int row, int h, float diff) {
      const float gin = agate ? ddrl_gate_factor(agate[h], ac_bern, ac_scale, gseed, gdraw, agent, GS_CRITIC_A, row, h) : 1.f;
      const float gt = agate ? ddrl_gate_factor(agate[h], ac_bern, ac_scale, gseed, gdraw, agent, GS_ACTOR_OUT, row, h) : 1.f;
      diff *= gin;                                                       // the critic's action-input gate
      if (inverting) {                                                   // :432-450, on the actions the actor's net returned (gated)
        const float out = acts[row * lda + h] * gt;
        if (diff < 0.f) diff *= (1.0f - out) / 2.0f;
        else if (diff > 0.f) diff *= (out + 1.0f) / 2.0f;
      }
      if (row >= B) diff = 0.f;
      if (dact_out && row < B) dact_out[row * A + h] = diff;
      diff *= gt;                                                        // the actor's output gate
      // the actor's output layer: its own blob values (ungated) feed the activation's backward; under an output tower the
      // tower's backward comes first (bwd_net_bn)
      const float out = topa[row * lda + h];
      topa[row * lda + h] = top_tower ? diff : act_bwd(out_act, out, diff);
    };
    bwd_net_bn<DW_NONE>(nq, w_q, nullptr, false, B, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, Z.q, S, A, store, wave, lane, tid);
  }
  __syncthreads();
  BN_STAMP(9);
  // the actor's hidden tiles shared their LDS with the critic pass: rebuild them (same batch statistics, running sums untouched;
  // the input tile kept its tower's output)
  fwd_net_bn(na, na.L() - 1, w_a, BN_RECOMPUTE, B, 1, M.xa, ldxa, M.hbase, M.hstride, nullptr, 0, Z.a, wave, lane, tid);
  BN_STAMP(10);
  bwd_net_bn<DW_STORE>(na, w_a, g_a, false, B, M.xa, ldxa, M.hbase, M.hstride, M.topa, lda, Z.a, 0, 0, none, wave, lane, tid);
  __syncthreads();
  BN_STAMP(11);
  {
    OptCtx oa = make_opt<EWC_A>(P, P.sa, na.stride(), agent, tau, omtau, (float)B, 0);
    oa.handoff = 0;
    bn_apply<EWC_A>(na, oa, g_a, tid);
  }
  BN_STAMP(12);
  // ---- soft updates of the statistics (:462-463: after both nets have stepped and the critic's second forward) ----
  bn_soft_stats(nq, w_q, w_qt, tau, omtau, tid);
  bn_soft_stats(na, w_a, w_at, tau, omtau, tid);
  BN_STAMP(13);
  if (tid == 0) {
    P.sq.iter[agent] += 1; P.sa.iter[agent] += 1;
    if (EWC_Q) P.sq.fisher_nbr[agent] = P.sq.fisher_nbr[agent] * P.fisher_beta + (float)B;     // DODevMLP.hpp:656
    if (EWC_A) P.sa.fisher_nbr[agent] = P.sa.fisher_nbr[agent] * P.fisher_beta + (float)B;
  }
}

extern "C" size_t ddpg_update_bn_lds_bytes(const DdpgParams *P) { return bn_lds_floats(*P) * sizeof(float); }

extern "C" hipError_t ddpg_update_bn_launch(const DdpgParams *P, hipStream_t st) {
  static LdsConfig configured[3];
  const size_t lds = ddpg_update_bn_lds_bytes(P);
  // the actor carries EWC whenever it is on, the critic with ewc_for_critic (DODevMLP.hpp:714-720)
  const int which = P->sa.fisher ? (P->sq.fisher ? 2 : 1) : 0;
  void (*k)(const DdpgParams) = which == 2 ? ddpg_update_bn_kernel<true, true> : which == 1 ? ddpg_update_bn_kernel<true, false>
                                                                                            : ddpg_update_bn_kernel<false, false>;
  hipError_t e = ddrl_set_lds(k, lds, configured[which]);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3(P->G), dim3(DDRL_NTHREADS), lds, st, *P);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// TD3Ag's update on nets with towers (agent/deepqn/include/TD3Ag.hpp:176-234: the same batch_norm_actor / _critic, test_net and
// bn_adapt keys as DeepQNAg; :284-380): the smoothed target action through TWO target critics, both critics' learn_batch, and on
// the agent's policy steps the actor update through the first critic, the three soft updates and the acting copy's refresh
// (the host's, after the launch).  The same passes and sweeps as ddpg_update_bn_kernel.  A tower works in place, so a critic's
// input tile is rebuilt from the replay rows before every critic pass.  The soft updates ride in the optimiser sweeps with
// tau = 0 off the policy steps (nothing reads a target between a critic's step and the policy step's soft update, as in
// kernels_td3.hip); the statistics' soft updates run at the end, on policy steps only.
// ---------------------------------------------------------------------------------------------------------------
// Two launches: PHASE 0 the targets and both critics, PHASE 1 the actor's step for the agents on a policy step (one kernel
// holding both referred to the argument block often enough for the compiler to give up splitting it into registers: it was
// copied to 1000 bytes of stack at entry and read from there).
template <bool EWC_A, bool EWC_Q, int PHASE>
__global__ void __launch_bounds__(DDRL_NTHREADS)
td3_update_bn_kernel(const DdpgParams P) {
  extern __shared__ __align__(16) float lds[];
  const int agent = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint8_t live = P.mask ? P.mask[agent] : (uint8_t)1;
  const uint8_t pstep = P.actor_step ? P.actor_step[agent] : (uint8_t)1;   // current_step_counter % policy_freq == 0 (:333)
  const int len = P.rep_len[agent], head = P.rep_head[agent];
  const float gamma = P.gamma[agent], tau = P.tau[agent], omtau = P.omtau[agent];
  if (!live || len < 250 || len < P.B) return;             // TD3Ag.hpp:286
  const bool actor_step = pstep != 0;
  if (PHASE == 1 && !actor_step) return;                   // uniform over the workgroup
  const float tau_e = actor_step ? tau : 0.f, omtau_e = actor_step ? omtau : 1.f;
  const NetDesc &na = P.a, &nq = P.q;
  const int S = P.S, A = P.A, B = P.B, ldxc = P.ldxc, ldxa = P.ldxa, ldq = nq.ld_top(), lda = na.ld_top();
  const BnLds Z = bn_carve(lds, P, ldq, lda);
  const LdsMap &M = Z.M;
  const int *idx = P.idx + (size_t)agent * B;
  float *w_a = P.sa.w + (size_t)agent * na.stride(), *w_at = P.sa.wt + (size_t)agent * na.stride();
  float *w_q = P.sq.w + (size_t)agent * nq.stride(), *w_qt = P.sq.wt + (size_t)agent * nq.stride();
  float *w_q2 = P.sq2.w + (size_t)agent * nq.stride(), *w_q2t = P.sq2.wt + (size_t)agent * nq.stride();
  float *g_a = P.gpart_a + (size_t)agent * na.stride(), *g_q = P.gpart_q + (size_t)agent * nq.stride();
  const float *noise = P.noise + (size_t)agent * B * A;
  const int tmode = P.test_net ? BN_TEST : BN_TRAIN;
  (void)tau_e; (void)omtau_e; (void)w_at; (void)w_qt; (void)w_q2t; (void)g_a; (void)tmode; (void)noise; (void)gamma;
  auto none = [](int, int, float) {};
  const float *agate = P.agate ? P.agate + (size_t)agent * A : nullptr;     // per-dimension factors or per-sample Bernoulli masks
  const int ac_bern = P.gm.ac_bern, ac_scale = P.gm.ac_scale;
  const unsigned gseed = P.gm.seed, gdraw = P.gm.draw;
  // Bernoulli mode: every Developmental layer instance has a mask of its own, the second critic and its target included
  // (td3_regate_xc builds the critic's input tile with that net's sites)
  const bool bern = (P.sgate || P.agate) && (P.gm.st_bern || P.gm.ac_bern);
  (void)bern;
  float *q1n = M.nrew;                                     // Q1'(s', a') per row (the n-step scratch: TD3Ag has no n-step targets)
  // actions into the critic's input tile: the actor's output gate, (target pass) the smoothing noise and the clip to [-1, 1]
  // (:307-310), then the critic's action-input gate.  src keeps the ungated values (see ddpg_update_bn_kernel)
  auto actions_in = [&](const float *src, bool smooth, int site_out, int site_in) __attribute__((always_inline)) {
    for (int e = tid; e < DDRL_MB * A; e += DDRL_NTHREADS) {
      const int row = e / A, col = e - row * A;
      float a = src[row * lda + col];
      if (agate) a *= ddrl_gate_factor(agate[col], ac_bern, ac_scale, gseed, gdraw, agent, site_out, row, col);
      if (smooth && row < B) a = fminf(fmaxf(a + noise[row * A + col], -1.f), 1.f);
      if (agate) a *= ddrl_gate_factor(agate[col], ac_bern, ac_scale, gseed, gdraw, agent, site_in, row, col);
      M.xc[row * ldxc + S + col] = a;
    }
  };

  if constexpr (PHASE == 0) {
  // ---- a' = clip(ann_target(s') + eps, -1, 1) (:304-310), Q1', Q2' (:313-314) ----
  gather_tile(P, S, A, ldxc, ldxa, agent, idx, B, head, true, M, tid);
  __syncthreads();
  fwd_net_bn(na, na.L(), w_at, tmode, B, 0, M.xa, ldxa, M.hbase, M.hstride, M.topa, lda, Z.a, wave, lane, tid);
  actions_in(M.topa, true, GS_ACTOR_T_OUT, GS_CRITIC_T_A);
  __syncthreads();
  fwd_net_bn(nq, nq.L(), w_qt, tmode, B, 0, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, Z.q, wave, lane, tid);
  if (tid < DDRL_MB) q1n[tid] = M.topq[tid * ldq];
  __syncthreads();
  if (bern) td3_regate_xc(P, S, A, ldxc, agent, idx, B, head, true, M, tid, GS_CRITIC2_T_S, 0);
  else gather_tile(P, S, A, ldxc, ldxa, agent, idx, B, head, true, M, tid);      // (the first target critic's input tower worked in place)
  actions_in(M.topa, true, GS_ACTOR_T_OUT, GS_CRITIC2_T_A);
  __syncthreads();
  fwd_net_bn(nq, nq.L(), w_q2t, tmode, B, 0, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, Z.q, wave, lane, tid);
  // ---- targets: r, or r + gamma min(Q1', Q2') (:317-326) ----
  gather_tile(P, S, A, ldxc, ldxa, agent, idx, B, head, false, M, tid);
  int trow;
  if (gather_owns_row(tid, trow)) {
    const float qn = fminf(q1n[trow], M.topq[trow * ldq]);
    M.yv[trow] = M.dn[trow] != 0.f ? M.rv[trow] : M.rv[trow] + gamma * qn;
  }
  __syncthreads();
  if (P.qt_out && tid < B) P.qt_out[(size_t)agent * B + tid] = M.yv[tid];
  // ---- both critics learn the same targets (:329-330; MLP.hpp:468-485) ----
  // (a macro, not a lambda or a loop over the two: anything that picks between P.sq and P.sq2 at run time, or takes P's address,
  // makes the compiler copy the kernel argument block to the stack)
#define TD3_BN_LEARN(WK, ST, LOSS, REGATHER)                                                                                          \
  {                                                                                                                                   \
    if (REGATHER) {   /* the first critic's input tower worked in place */                                                            \
      if (bern) td3_regate_xc(P, S, A, ldxc, agent, idx, B, head, false, M, tid, GS_CRITIC2_S, GS_CRITIC2_A_REPLAY);                  \
      else gather_tile(P, S, A, ldxc, ldxa, agent, idx, B, head, false, M, tid);                                                      \
      __syncthreads();                                                                                                                \
    }                                                                                                                                 \
    fwd_net_bn(nq, nq.L(), WK, BN_TRAIN, B, 0, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, Z.q, wave, lane, tid);                    \
    if (tid < DDRL_MB) {                                                                                                              \
      const float df = tid < B ? M.topq[tid * ldq] - M.yv[tid] : 0.f;                                                                 \
      float sq = df * df;                                                                                                             \
      for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);                                                                       \
      if (tid == 0) LOSS[agent] = sq / (float)B * 0.5f;                                                                               \
      M.topq[tid * ldq] = df / (float)B;                                                                                              \
    }                                                                                                                                 \
    __syncthreads();                                                                                                                  \
    bwd_net_bn<DW_STORE>(nq, WK, g_q, false, B, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, Z.q, 0, 0, none, wave, lane, tid);       \
    __syncthreads();   /* workgroup-scope fence: the sweep reads other threads' gradient stores */                                    \
    OptCtx oq = make_opt<EWC_Q>(P, ST, nq.stride(), agent, tau_e, omtau_e, (float)B, 0);                                              \
    oq.handoff = 0;                                                                                                                   \
    bn_apply<EWC_Q>(nq, oq, g_q, tid);                                                                                                \
    __syncthreads();                                                                                                                  \
  }
  TD3_BN_LEARN(w_q, P.sq, P.loss_out, false)
  TD3_BN_LEARN(w_q2, P.sq2, P.loss2_out, true)
#undef TD3_BN_LEARN
  if (tid == 0) {
    P.sq.iter[agent] += 1;                                 // Solver::Step ++iter, both critics
    P.sq2.iter[agent] += 1;
    if (EWC_Q) {
      P.sq.fisher_nbr[agent] = P.sq.fisher_nbr[agent] * P.fisher_beta + (float)B;
      P.sq2.fisher_nbr[agent] = P.sq2.fisher_nbr[agent] * P.fisher_beta + (float)B;
    }
  }
  } else {
  // ---- delayed actor update through the FIRST critic (:333-363) ----
  {
    gather_tile(P, S, A, ldxc, ldxa, agent, idx, B, head, false, M, tid);
    __syncthreads();
    const bool adapt = P.bn_adapt && na.bn_any();
    if (adapt) {   // :339-344, the TEST-phase copy's actions
      BnBufs nokeep = {nullptr, nullptr, nullptr, nullptr};
      fwd_net_bn(na, na.L(), const_cast<float *>(P.a_test) + (size_t)agent * na.stride(), BN_TEST, B, 0, M.xa, ldxa, M.hbase, M.hstride,
                 M.topa, lda, nokeep, wave, lane, tid);
      for (int e = tid; e < DDRL_MB * lda; e += DDRL_NTHREADS) Z.pa[e] = M.topa[e];
      __syncthreads();
      gather_tile(P, S, A, ldxc, ldxa, agent, idx, B, head, false, M, tid);
      __syncthreads();
    }
    fwd_net_bn(na, na.L(), w_a, BN_TRAIN, B, 0, M.xa, ldxa, M.hbase, M.hstride, M.topa, lda, Z.a, wave, lane, tid);   // pi = ann(s)
    float *acts = adapt ? Z.pa : M.topa;
    actions_in(acts, false, GS_ACTOR_OUT, GS_CRITIC_A);
    __syncthreads();
    fwd_net_bn(nq, nq.L(), w_q, BN_TRAIN, B, 0, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, Z.q, wave, lane, tid);   // Q1(s, pi(s)), new critic
    if (tid < DDRL_MB) M.topq[tid * ldq] = tid < B ? -1.0f : 0.f;            // :351-352, NOT /B
    __syncthreads();
    {
      float *topa = M.topa;
      const int out_act = na.out_act();
      const bool top_tower = na.bn(na.L());
      float *dact_out = P.dact_out ? P.dact_out + (size_t)agent * B * A : nullptr;
      auto store = [=](int row, int h, float diff) {
        const float gin = agate ? ddrl_gate_factor(agate[h], ac_bern, ac_scale, gseed, gdraw, agent, GS_CRITIC_A, row, h) : 1.f;
        const float gt = agate ? ddrl_gate_factor(agate[h], ac_bern, ac_scale, gseed, gdraw, agent, GS_ACTOR_OUT, row, h) : 1.f;
        diff *= gin;                                                       // the critic's action-input gate
        if (row >= B) diff = 0.f;
        if (dact_out && row < B) dact_out[row * A + h] = diff;
        diff *= gt;                                                        // the actor's output gate
        const float out = topa[row * lda + h];
        topa[row * lda + h] = top_tower ? diff : act_bwd(out_act, out, diff);
      };
      bwd_net_bn<DW_NONE>(nq, w_q, nullptr, false, B, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, Z.q, S, A, store, wave, lane, tid);
    }
    __syncthreads();
    fwd_net_bn(na, na.L() - 1, w_a, BN_RECOMPUTE, B, 1, M.xa, ldxa, M.hbase, M.hstride, nullptr, 0, Z.a, wave, lane, tid);
    bwd_net_bn<DW_STORE>(na, w_a, g_a, false, B, M.xa, ldxa, M.hbase, M.hstride, M.topa, lda, Z.a, 0, 0, none, wave, lane, tid);
    __syncthreads();
    {
      OptCtx oa = make_opt<EWC_A>(P, P.sa, na.stride(), agent, tau, omtau, (float)B, 0);
      oa.handoff = 0;
      bn_apply<EWC_A>(na, oa, g_a, tid);
    }
    // ---- the statistics' soft updates (:368-370), after every net has stepped and the critic's last forward ----
    bn_soft_stats(nq, w_q, w_qt, tau, omtau, tid);
    bn_soft_stats(nq, w_q2, w_q2t, tau, omtau, tid);
    bn_soft_stats(na, w_a, w_at, tau, omtau, tid);
  }
  if (tid == 0) {
    P.sa.iter[agent] += 1;                                 // set_iter(iter() + 1) :363
    if (EWC_A) P.sa.fisher_nbr[agent] = P.sa.fisher_nbr[agent] * P.fisher_beta + (float)B;
  }
  }
}

extern "C" hipError_t td3_update_bn_launch(const DdpgParams *P, int any_actor, hipStream_t st) {
  static LdsConfig configured[3][2];
  const size_t lds = ddpg_update_bn_lds_bytes(P);
  const int which = P->sa.fisher ? (P->sq.fisher ? 2 : 1) : 0;
  for (int phase = 0; phase < (any_actor ? 2 : 1); ++phase) {
    void (*k)(const DdpgParams);
    if (phase == 0) k = which == 2 ? td3_update_bn_kernel<true, true, 0> : which == 1 ? td3_update_bn_kernel<true, false, 0> : td3_update_bn_kernel<false, false, 0>;
    else k = which == 2 ? td3_update_bn_kernel<true, true, 1> : which == 1 ? td3_update_bn_kernel<true, false, 1> : td3_update_bn_kernel<false, false, 1>;
    hipError_t e = ddrl_set_lds(k, lds, configured[which][phase]);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(P->G), dim3(DDRL_NTHREADS), lds, st, *P);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  return hipSuccess;
}

// ---------------------------------------------------------------------------------------------------------------
// TEST-phase evaluation of n rows per agent through one net with towers (ddrl_pop_forward_batch on such a net: the value
// a TEST-phase copy returns, MLP.hpp:284-338, 531-579; no state changes)
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(DDRL_NTHREADS)
net_forward_bn_kernel(const NetDesc nd, const float *__restrict__ w, const float *__restrict__ X, float *__restrict__ out, int n, int ldh,
                      const float *__restrict__ sgate, int S, const float *__restrict__ agate, int A, int gate_out, const GateMode gm) {
  extern __shared__ __align__(16) float lds[];
  const int chunk = blockIdx.x, agent = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = chunk * DDRL_MB, rows = n - row0 < DDRL_MB ? n - row0 : DDRL_MB;
  const int d0 = nd.d(0), dl = nd.d(nd.L()), ldx = nd.ld(0), ldt = nd.ld(nd.L());
  float *hbase = lds, *x = hbase + (size_t)(nd.L() - 1) * DDRL_MB * ldh, *top = x + DDRL_MB * ldx;
  const float *xin = X + (size_t)agent * n * d0;
  for (int e = tid; e < DDRL_MB * ldx; e += DDRL_NTHREADS) {
    const int row = e / ldx, col = e - row * ldx;
    float v = (row < rows && col < d0) ? xin[(size_t)(row0 + row) * d0 + col] : 0.f;
    // the Developmental layers: per-dimension factors or per-sample Bernoulli gates (the reference runs its Developmental tests
    // over every batch_norm code, CaffeUtest.cpp:620, 693, 805, 917); the state layer sits ahead of an input tower
    if (sgate && col < S)
      v *= ddrl_gate_factor(sgate[(size_t)agent * S + col], gm.st_bern, gm.st_scale, gm.seed, gm.draw, agent, GS_BATCH_S, row0 + row, col);
    else if (agate && !gate_out && col >= S && col < S + A)
      v *= ddrl_gate_factor(agate[(size_t)agent * A + col - S], gm.ac_bern, gm.ac_scale, gm.seed, gm.draw, agent, GS_BATCH_A, row0 + row, col - S);
    x[e] = v;
  }
  __syncthreads();
  const BnBufs nokeep = {nullptr, nullptr, nullptr, nullptr};
  fwd_net_bn(nd, nd.L(), const_cast<float *>(w) + (size_t)agent * nd.stride(), BN_TEST, rows, 0, x, ldx, hbase, DDRL_MB * ldh, top, ldt,
             nokeep, wave, lane, tid);
  float *o = out + (size_t)agent * n * dl;
  for (int e = tid; e < DDRL_MB * dl; e += DDRL_NTHREADS) {
    const int row = e / dl, col = e - row * dl;
    if (row < rows) {
      float v = top[row * ldt + col];                       // the action layer sits behind an output tower
      if (agate && gate_out)
        v *= ddrl_gate_factor(agate[(size_t)agent * A + col], gm.ac_bern, gm.ac_scale, gm.seed, gm.draw, agent, GS_BATCH_A, row0 + row, col);
      o[(size_t)(row0 + row) * dl + col] = v;
    }
  }
}

extern "C" hipError_t net_forward_bn_launch(const NetDesc *nd, const float *w, const float *X, float *out, int n, int G,
                                            const float *sgate, int S, const float *agate, int A, int gate_out, const GateMode *gm, hipStream_t st) {
  static LdsConfig conf;
  int ldh = 4;
  for (int l = 1; l < nd->L_; ++l) ldh = std::max(ldh, nd->ld_[l]);
  const size_t lds = ((size_t)(nd->L_ - 1) * DDRL_MB * ldh + (size_t)DDRL_MB * (nd->ld_[0] + nd->ld_[nd->L_])) * sizeof(float);
  hipError_t e = ddrl_set_lds(net_forward_bn_kernel, lds, conf);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(net_forward_bn_kernel, dim3((n + DDRL_MB - 1) / DDRL_MB, G), dim3(DDRL_NTHREADS), lds, st, *nd, w, X, out, n, ldh, sgate, S, agate, A, gate_out, *gm);
  return hipGetLastError();
}
