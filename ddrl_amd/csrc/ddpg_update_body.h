// ddpg_update_body.h -- one pass of the fupd loop (DeepQNAg.hpp:343-466) for one agent, cut into the four
// QUARTERS the two update kernels schedule:
//   q0  gather s', actor-target and critic-target passes, TD targets, critic forward, loss seed      (compute)
//   q1  critic backward with the fused optimiser epilogue                                            (streams state)
//   q2  actor forward, critic forward on (s, pi(s)), input-gradient pass, actor hidden recompute     (compute)
//   q3  actor backward with the fused optimiser epilogue, iteration counters                         (streams state)
// kernels_ddpg.hip runs them back to back in one 256-thread workgroup per (update, agent);
// kernels_ddpg_pair.hip runs two agents per 512-thread workgroup, one quarter apart, so that one team's streaming
// quarter always overlaps the other's compute quarter.  Nothing but LDS contents (the LdsMap) and HBM state is
// carried from one quarter to the next.  `tid` is the thread's index in its 256-thread team.
#pragma once
#include "ddpg_device.h"

template <class NA, class NQ>
struct UpdView {
  int S, A, B, ldxc, ldxa, ldq, lda, tid, lane, wave;
  LdsMap M;
  const float *agate;
  const float *w_a, *w_at, *w_q, *w_qt;
  long long *stamps;
  bool diag;
  const int *idx;
  float *park;
};

template <class NA, class NQ>
__device__ __forceinline__ UpdView<NA, NQ> upd_view(const DdpgParams &P, float *lds, int agent, int u, int tid) {
  const auto &na = Pick<NA>::get(P.a);
  const auto &nq = Pick<NQ>::get(P.q);
  UpdView<NA, NQ> V;
  V.S = na.d(0); V.A = na.d(na.L()); V.B = P.B;
  V.ldxc = nq.ld(0); V.ldxa = na.ld(0);
  V.ldq = nq.ld(nq.L()); V.lda = na.ld(na.L());
  V.tid = tid; V.lane = tid & 63;
  V.wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: branches on it are uniform, waits are counted exactly
  V.M = ddpg_carve(lds, na.L(), nq.L(), P.ldh, V.ldxc, V.ldxa, V.ldq, V.lda);
  V.stamps = P.stamps ? P.stamps + (size_t)agent * 16 : nullptr;
  V.agate = P.agate ? P.agate + (size_t)agent * V.A : nullptr;
  const size_t sa = P.resident ? 0 : (size_t)agent;      // resident: the state arrays are one agent's LDS copy
  V.w_a = P.sa.w + sa * na.stride(); V.w_at = P.sa.wt + sa * na.stride();
  V.w_q = P.sq.w + sa * nq.stride(); V.w_qt = P.sq.wt + sa * nq.stride();
#ifdef DDRL_EXP_HOTW      // experiment: every agent's passes read agent 0's weights (always cache-hot)
  V.w_a = P.sa.w; V.w_at = P.sa.wt; V.w_q = P.sq.w; V.w_qt = P.sq.wt;
#endif
  // the diagnostics (ddrl_pop_ddpg_last) are those of the agent's LAST update of the launch: written by
  // that workgroup only (two updates of one agent may run on different XCDs, whose L2s would write the
  // same addresses back in no particular order)
  V.diag = u == P.nupd - 1;
  V.idx = P.idx + ((size_t)agent * P.nupd + u) * V.B;
  // compile-time shapes: the tiles from topq to the end of the allocation (topq after the top layer,
  // topa, the per-row scalars and the tail the host adds, ddpg_update_lds_bytes) are dead during the
  // optimiser loops below the top layer and park input-gradient accumulators there
  V.park = ParkOf<NA, NQ>::value ? V.M.topq : nullptr;
  return V;
}

#define STAMP(i) do { if (V.stamps && V.tid == 0) V.stamps[i] = clock64(); } while (0)

// the actor's actions into the critic's input tile (the actor's output gate, then the critic's action-input gate)
// (the gate mode is read from the kernel arguments inside the gated branch only: nothing of it is live in ungated runs)
template <class NA, class NQ>
__device__ __forceinline__ void actions_to_critic(const DdpgParams &P, const UpdView<NA, NQ> &V, int agent, int u, bool target) {
  const LdsMap &M = V.M;
  for (int e = V.tid; e < DDRL_MB * V.A; e += DDRL_NTHREADS) {
    const int row = e / V.A, col = e - row * V.A;
    float a = M.topa[row * V.lda + col];
    if (V.agate) {
      const GateMode &gm = P.gm;
      const unsigned draw = gm.draw + (unsigned)u;
      a *= ddrl_gate_factor(V.agate[col], gm.ac_bern, gm.ac_scale, gm.seed, draw, agent, target ? GS_ACTOR_T_OUT : GS_ACTOR_OUT, row, col);
      M.topa[row * V.lda + col] = a;
      a *= ddrl_gate_factor(V.agate[col], gm.ac_bern, gm.ac_scale, gm.seed, draw, agent, target ? GS_CRITIC_T_A : GS_CRITIC_A, row, col);
    }
    M.xc[row * V.ldxc + V.S + col] = a;
  }
  ddrl_sync();
}

// ---- q0: a' = ann_target(s'), q' = qnn_target(s', a') (DeepQNAg.hpp:379-383), TD targets (:386-405), critic forward + loss seed (:408) ----
// columns of the critic's input tile per gathering thread (compile-time shapes)
template <class NQ> struct GatherCols { static constexpr int value = (NQ::ld(0) + DDRL_GATHER_TPR - 1) / DDRL_GATHER_TPR; };
template <> struct GatherCols<NetDesc> { static constexpr int value = 1; };
// can this instantiation split its gathers into request / commit halves (ddpg_device.h)?
template <class NA, class NQ> struct GatherSplit {
  static constexpr bool value = !__is_same(NA, NetDesc) && !__is_same(NQ, NetDesc) && GatherCols<NQ>::value <= DDRL_GATHER_MAXC;
};

// the s' rows of update u of this launch, requested ahead of the update (before the hand-off wait of kernels_ddpg.hip,
// during the previous update's last quarter in kernels_ddpg_pair.hip); false: not this instantiation / configuration
template <class NA, class NQ>
__device__ __forceinline__ bool ddpg_prefetch_rows(const DdpgParams &P, int agent, int u, int head, int tid, GatherRegs &g) {
  if (!GatherSplit<NA, NQ>::value || P.nstep > 0 || u >= P.nupd) return false;
  const auto &na = Pick<NA>::get(P.a);
  const auto &nq = Pick<NQ>::get(P.q);
  gather_load<GatherCols<NQ>::value>(P, na.d(0), na.d(na.L()), nq.ld(0), agent, P.idx + ((size_t)agent * P.nupd + u) * P.B, P.B, head, true, tid, g);
  return true;
}

// pre / have_pre: the s' rows already requested by ddpg_prefetch_rows
template <class NA, class NQ, bool EWC_A, bool EWC_Q>
__device__ __forceinline__ void ddpg_q0(const DdpgParams &P, float *lds, int agent, int u, int len, int head, float gamma, int tid,
                                        const GatherRegs &pre, bool have_pre) {
  const auto &na = Pick<NA>::get(P.a);
  const auto &nq = Pick<NQ>::get(P.q);
  const UpdView<NA, NQ> V = upd_view<NA, NQ>(P, lds, agent, u, tid);
  const LdsMap &M = V.M;
  const int S = V.S, A = V.A, B = V.B, ldxc = V.ldxc, ldxa = V.ldxa, ldq = V.ldq, lda = V.lda, lane = V.lane, wave = V.wave;
  STAMP(0);
  // the first weight fragments of the actor-target pass travel while the minibatch is gathered
  FwdCarry cy;
  const FwdNext no_next = {nullptr, nullptr, 4, 1};
  if (!__is_same(NA, NetDesc)) fwd_request(fwd_first(na, V.w_at), lane, cy, wave);
#ifdef DDRL_EXP_TOUCHW   // measured: the passes 2 k cycles shorter each, this quarter 21 k longer (the touches queue ahead of the gather)
  l2_touch_block(V.w_at, na.stride(), M.dump, tid);
  l2_touch_block(V.w_qt, nq.stride(), M.dump, tid);
  l2_touch_block(V.w_q, nq.stride(), M.dump, tid);
#endif
  // n-step horizon (DeepQNAg.hpp:360-373, 394-402): stops before a transition that reached the goal and at the end
  // of the replay; rewards of steps [0, valid) are discounted
  // compile-time shapes, one-step targets: both gathers of the update are REQUESTED here (the second one's rows wait in
  // registers through the two target passes); else the one-piece gathers
  constexpr int NCOL = GatherCols<NQ>::value;
  const bool split = GatherSplit<NA, NQ>::value && P.nstep <= 0;
  GatherRegs g1, g2;
  if (split) {
    if (have_pre) g1 = pre; else gather_load<NCOL>(P, S, A, ldxc, agent, V.idx, B, head, true, tid, g1);
    gather_load<NCOL>(P, S, A, ldxc, agent, V.idx, B, head, false, tid, g2);
    gather_store<NCOL>(P, S, A, ldxc, ldxa, agent, B, true, M, tid, u, g1);
  } else {
    if (P.nstep > 0) nstep_horizon(P, S, A, agent, V.idx, B, len, head, gamma, M, tid);
    gather_tile(P, S, A, ldxc, ldxa, agent, V.idx, B, head, true, M, tid, u);
  }
  ddrl_sync();
  fwd_net(na, na.L(), V.w_at, M.xa, ldxa, M.hbase, M.hstride, M.topa, lda, wave, lane, cy, true, fwd_first(nq, V.w_qt));
  actions_to_critic(P, V, agent, u, true);
  STAMP(1);
  fwd_net(nq, nq.L(), V.w_qt, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, wave, lane, cy, true, fwd_first(nq, V.w_q));
  STAMP(2);
  // ---- TD targets (DeepQNAg.hpp:386-405) ----
  if (split) gather_store<NCOL>(P, S, A, ldxc, ldxa, agent, B, false, M, tid, u, g2);
  else gather_tile(P, S, A, ldxc, ldxa, agent, V.idx, B, head, false, M, tid, u);
  int trow;
  if (gather_owns_row(tid, trow)) {     // the thread that gathered the row's reward / goal flag
    const float qn = M.topq[trow * ldq];
    // :388-403.  n-step: sum_{j<valid} gamma^j r_{t+j} + gamma * Q' (the reference's exponent, not gamma^valid)
    M.yv[trow] = M.dn[trow] != 0.f ? M.rv[trow] : (P.nstep > 0 ? M.nrew[trow] : M.rv[trow]) + gamma * qn;
  }
  ddrl_sync();
  if (P.qt_out && V.diag && tid < B) P.qt_out[(size_t)agent * B + tid] = M.yv[tid];
  STAMP(3);
  // ---- critic update: qnn->learn_batch(s, a, y, 1)  (DeepQNAg.hpp:408; MLP.hpp:468-485) ----
  fwd_net<NQ, false>(nq, nq.L(), V.w_q, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, wave, lane, cy, true, no_next);
  if (tid < DDRL_MB) {
    // EuclideanLoss: loss = sum (q - y)^2 / (2B), dq = (q - y) / B
    const float df = tid < B ? M.topq[tid * ldq] - M.yv[tid] : 0.f;
    float sq = df * df;
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if (tid == 0 && V.diag) P.loss_out[agent] = sq / (float)B * 0.5f;
    M.topq[tid * ldq] = df / (float)B;
  }
  ddrl_sync();
  STAMP(4);
}

// ---- q1: the critic's backward pass, dW fused with Adam + L2 + EWC + the soft target update ----
template <class NA, class NQ, bool EWC_A, bool EWC_Q>
__device__ __forceinline__ void ddpg_q1(const DdpgParams &P, float *lds, int agent, int u, float tau, float omtau, int tid) {
  const auto &nq = Pick<NQ>::get(P.q);
  const UpdView<NA, NQ> V = upd_view<NA, NQ>(P, lds, agent, u, tid);
  const LdsMap &M = V.M;
  auto none = [](int, int, float) {};
#ifdef DDRL_EXP_TOUCHW
  l2_touch_block(V.w_a, Pick<NA>::get(P.a).stride(), M.dump, tid);      // the actor's weights for the next quarter
#endif
  OptCtx oq = make_opt<EWC_Q>(P, P.sq, nq.stride(), agent, tau, omtau, (float)V.B, u);
  oq.dump = M.dump;
#ifdef DDRL_EXP_PRIO
  __builtin_amdgcn_s_setprio(2);
#endif
  bwd_net<DW_OPT, EWC_Q>(nq, V.w_q, oq, nullptr, M.xc, V.ldxc, M.hbase, M.hstride, M.topq, V.ldq, 0, 0, none, V.wave, V.lane, tid,
                         V.stamps ? V.stamps + 11 : nullptr, V.park);
#ifdef DDRL_EXP_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif
  STAMP(5);
}

// The actor's hidden activations share their LDS tiles with the critic passes that follow pi = ann(s).  Row-parallel
// passes: a wave wrote rows 16w .. 16w+15 of every hidden tile itself, and those 16 rows are one contiguous run of
// 16 ld floats.  Compile-time shapes keep them in REGISTERS across the two critic passes (mid nets: 13 float4 per lane)
// and put them back before the actor's backward pass, instead of running the actor's hidden layers a second time
// (252 of an update's 2 912 MFMAs per wave, one more read of the actor's weights, 25-30 k of its ~500 k cycles).
// Exact: the second forward pass recomputed the same values from the same inputs.
template <class NA> struct KeepHidden {
  static constexpr int per(int l) { return ddrl_slots((4 * NA::ld(l + 1) + 63) / 64); }   // float4 per lane of hidden tile l (two waves per row tile: every other one)
  static constexpr int before(int l) { int t = 0; for (int i = 0; i < l; ++i) t += per(i); return t; }
  static constexpr int total = before(NA::L() - 1);
  static constexpr bool value = total >= 1 && total <= 16;
};
template <> struct KeepHidden<NetDesc> { static constexpr int total = 1; static constexpr bool value = false; };

template <class NA, bool SAVE>
__device__ __forceinline__ void keep_hidden(float *hbase, int hstride, int wave, int lane, f32x4 (&keep)[KeepHidden<NA>::total]) {
  if constexpr (KeepHidden<NA>::value) {
#pragma unroll
    for (int l = 0; l < NA::L() - 1; ++l) {
      const int ld = NA::ld(l + 1), nf4 = 4 * ld;                  // float4s in this wave's 16 rows
      float *base = hbase + l * hstride + 16 * ddrl_rt(wave) * ld;
#pragma unroll
      for (int j = 0; j < KeepHidden<NA>::per(l); ++j) {
        const int f = lane + 64 * (ddrl_part(wave) + DDRL_PARTS * j);
        if (f < nf4) {
          if (SAVE) keep[KeepHidden<NA>::before(l) + j] = *(const f32x4 *)(base + 4 * f);
          else *(f32x4 *)(base + 4 * f) = keep[KeepHidden<NA>::before(l) + j];
        }
      }
    }
  }
}

// ---- q2: pi = ann(s), Q(s, pi(s)) with the NEW critic, dQ/da (DeepQNAg.hpp:411-455), the actor's hidden activations again ----
template <class NA, class NQ, bool EWC_A, bool EWC_Q>
__device__ __forceinline__ void ddpg_q2(const DdpgParams &P, float *lds, int agent, int u, int tid) {
  const auto &na = Pick<NA>::get(P.a);
  const auto &nq = Pick<NQ>::get(P.q);
  const UpdView<NA, NQ> V = upd_view<NA, NQ>(P, lds, agent, u, tid);
  const LdsMap &M = V.M;
  const int S = V.S, A = V.A, B = V.B, ldxc = V.ldxc, ldxa = V.ldxa, ldq = V.ldq, lda = V.lda, lane = V.lane, wave = V.wave;
  fwd_net(na, na.L(), V.w_a, M.xa, ldxa, M.hbase, M.hstride, M.topa, lda, wave, lane);       // pi = ann(s)
  f32x4 keep[KeepHidden<NA>::total];
  keep_hidden<NA, true>(M.hbase, M.hstride, wave, lane, keep);            // own rows: written by this wave, no barrier needed
  actions_to_critic(P, V, agent, u, false);
  STAMP(6);
  fwd_net(nq, nq.L(), V.w_q, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, wave, lane);       // Q(s, pi(s)), new critic
  STAMP(7);
  if (tid < DDRL_MB) M.topq[tid * ldq] = tid < B ? -1.0f : 0.f;            // :424-428, NOT /B
  ddrl_sync();
  {
    // critic_backward (:429): only d/d(actions) is needed; critic parameter diffs are discarded
    float *topa = M.topa;
    const float *agate = V.agate;
    const DdpgParams *Pg = &P;                                             // the gate mode: read only under a gate
    const int inverting = P.inverting, out_act = na.out_act();
    float *dact_out = P.dact_out && V.diag ? P.dact_out + (size_t)agent * B * A : nullptr;
    auto store = [=](int row, int h, float diff) {
      float out = topa[row * lda + h];
      // the masks of the forward passes these gradients belong to, regenerated from their keys
      float gin = 1.f, gt = 1.f;
      if (agate) {
        const GateMode &gm = Pg->gm;
        const unsigned draw = gm.draw + (unsigned)u;
        gin = ddrl_gate_factor(agate[h], gm.ac_bern, gm.ac_scale, gm.seed, draw, agent, GS_CRITIC_A, row, h);
        gt = ddrl_gate_factor(agate[h], gm.ac_bern, gm.ac_scale, gm.seed, draw, agent, GS_ACTOR_OUT, row, h);
      }
      diff *= gin;                                                       // the critic's action-input gate
      if (inverting) {                                                   // :432-450
        if (diff < 0.f) diff *= (1.0f - out) / 2.0f;
        else if (diff > 0.f) diff *= (out + 1.0f) / 2.0f;
      }
      if (dact_out && row < B) dact_out[row * A + h] = diff;
      diff *= gt;                                                        // the actor's output gate
      if (gt != 0.f) out /= gt;                                          // its activation saw the ungated value
      topa[row * lda + h] = act_bwd(out_act, out, diff);                 // actor's output layer, in place
    };
    OptCtx unused = {};
    bwd_net<DW_NONE, false>(nq, V.w_q, unused, nullptr, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, S, A, store, wave, lane, tid);
  }
  STAMP(8);
  // the actor's hidden activations shared their LDS with the critic pass: recompute them (one extra
  // actor forward is cheaper than doubling LDS and halving occupancy); topa now holds dZ_L
  if constexpr (KeepHidden<NA>::value) {
    // the critic's hidden tiles have the critic's row stride: another wave's rows of its passes may lie where this
    // wave's actor rows go, so every wave must be through the input-gradient pass first
    ddrl_sync();
    keep_hidden<NA, false>(M.hbase, M.hstride, wave, lane, keep);
    ddrl_sync();
  } else {
    fwd_net(na, na.L() - 1, V.w_a, M.xa, ldxa, M.hbase, M.hstride, nullptr, 0, wave, lane);
  }
  STAMP(9);
}

// ---- q3: the actor's backward pass with its optimiser epilogue (:455-463), then the solvers' iteration counters ----
// handoff: the agent's next update runs in ANOTHER workgroup of this launch (kernels_ddpg.hip with nupd > 1)
template <class NA, class NQ, bool EWC_A, bool EWC_Q>
__device__ __forceinline__ void ddpg_q3(const DdpgParams &P, float *lds, int agent, int u, float tau, float omtau, int tid, bool handoff) {
  const auto &na = Pick<NA>::get(P.a);
  const UpdView<NA, NQ> V = upd_view<NA, NQ>(P, lds, agent, u, tid);
  const LdsMap &M = V.M;
  auto none = [](int, int, float) {};
  {
    OptCtx oa = make_opt<EWC_A>(P, P.sa, na.stride(), agent, tau, omtau, (float)V.B, u);  // updateFisher(traj.size()) :456
    oa.dump = M.dump;
#ifdef DDRL_EXP_PRIO
    __builtin_amdgcn_s_setprio(2);
#endif
    bwd_net<DW_OPT, EWC_A>(na, V.w_a, oa, nullptr, M.xa, V.ldxa, M.hbase, M.hstride, M.topa, V.lda, 0, 0, none, V.wave, V.lane, tid, nullptr, V.park);
  }
#ifdef DDRL_EXP_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif
  STAMP(10);
  if (tid == 0) {
    if (handoff) {     // read by the agent's next update in this launch: sc1 stores (see store_sc1)
      __hip_atomic_store(P.sq.iter + agent, P.sq.iter[agent] + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(P.sa.iter + agent, P.sa.iter[agent] + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (EWC_Q) store_sc1(P.sq.fisher_nbr + agent, P.sq.fisher_nbr[agent] * P.fisher_beta + (float)V.B);
      if (EWC_A) store_sc1(P.sa.fisher_nbr + agent, P.sa.fisher_nbr[agent] * P.fisher_beta + (float)V.B);
    } else {
      P.sq.iter[agent] += 1;                                             // Solver::Step ++iter
      P.sa.iter[agent] += 1;                                             // set_iter(iter()+1) :459
      if (EWC_Q) P.sq.fisher_nbr[agent] = P.sq.fisher_nbr[agent] * P.fisher_beta + (float)V.B;
      if (EWC_A) P.sa.fisher_nbr[agent] = P.sa.fisher_nbr[agent] * P.fisher_beta + (float)V.B;
    }
  }
}
#undef STAMP
