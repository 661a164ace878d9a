// kernels_ddpg_big.hip -- the DDPG update for minibatches of more than 64 rows (the reference's own
// ddpg.ini / ddpg-cp.ini use mini_batch_size = 512).  Same arithmetic as the fused kernel
// (kernels_ddpg.hip, DeepQNAg.hpp:343-466), cut where a full-batch gradient has to be complete:
//   critic chunks:  targets, critic forward / backward on 64 rows each -> partial gradients
//   critic apply:   sum of the partials in chunk order, Fisher / EWC / L2 / Adam / soft target update
//   actor chunks:   pi(s), Q(s, pi(s)) with the NEW critic, dQ/da, inverting gradients, actor backward
//   actor apply
// One workgroup per (chunk, agent): a single agent with B = 512 still spreads over 8 CUs.
// TD3 = true: TD3Ag's update (TD3Ag.hpp:284-380, see kernels_td3.hip) cut the same way: the critic chunk
// smooths the target action, takes min(Q1', Q2') and produces BOTH critics' partial gradients; the actor
// chunk and the actor's apply run for the agents on a policy step only; three applies.
#include "ddpg_device.h"

// PHASE 0: critic chunk.  PHASE 1: actor chunk.
template <int PHASE, bool TD3>
__global__ void __launch_bounds__(DDRL_NTHREADS, DDRL_NTHREADS / 128)   // two workgroups per CU (<= 256 registers)
ddpg_chunk_kernel(const DdpgParams P) {
  extern __shared__ __align__(16) float lds[];
  int chunk, one, agent;                // an agent's chunks on one XCD: its weights come from that L2 after the first
  if (!xcd_decode((P.B + DDRL_MB - 1) / DDRL_MB, 1, P.G, chunk, one, agent)) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (P.mask && !P.mask[agent]) return;
  const int len = P.rep_len[agent], head = P.rep_head[agent];
  if (len < 250 || len < P.B) return;                      // DeepQNAg.hpp:338
  if (TD3 && PHASE == 1 && P.actor_step && !P.actor_step[agent]) return;   // delayed policy update (TD3Ag.hpp:333)
  const NetDesc &na = P.a, &nq = P.q;
  const int S = na.d(0), A = na.d(na.L()), B = P.B;
  const int row0 = chunk * DDRL_MB;
  const int nrows = B - row0 < DDRL_MB ? B - row0 : DDRL_MB;   // > 0 by the grid
  const int ldxc = nq.ld(0), ldxa = na.ld(0);
  const int ldq = nq.ld(nq.L()), lda = na.ld(na.L());
  const LdsMap M = ddpg_carve(lds, na.L(), nq.L(), P.ldh, ldxc, ldxa, ldq, lda);
  const int *idx = P.idx + (size_t)agent * B + row0;
  const float gamma = P.gamma[agent];
  const float *agate = P.agate ? P.agate + (size_t)agent * A : nullptr;
  const float *w_a = P.sa.w + (size_t)agent * na.stride(), *w_at = P.sa.wt + (size_t)agent * na.stride();
  const float *w_q = P.sq.w + (size_t)agent * nq.stride(), *w_qt = P.sq.wt + (size_t)agent * nq.stride();
  // actor output (gated) into the critic's action columns (gated again).  Bernoulli gates (DDPG only here): each layer instance's
  // mask is drawn per MINIBATCH row (row0 + row), the same masks the one-tile kernel would draw
  auto actions_to_critic = [&]() {
    for (int e = tid; e < DDRL_MB * A; e += DDRL_NTHREADS) {
      const int row = e / A, col = e - row * A;
      float a = M.topa[row * lda + col];
      if (agate) {
        const GateMode &gm = P.gm;
        a *= ddrl_gate_factor(agate[col], gm.ac_bern, gm.ac_scale, gm.seed, gm.draw, agent, PHASE == 0 ? GS_ACTOR_T_OUT : GS_ACTOR_OUT, row0 + row, col);
        M.topa[row * lda + col] = a;
        a *= ddrl_gate_factor(agate[col], gm.ac_bern, gm.ac_scale, gm.seed, gm.draw, agent, PHASE == 0 ? GS_CRITIC_T_A : GS_CRITIC_A, row0 + row, col);
      }
      M.xc[row * ldxc + S + col] = a;
    }
    __syncthreads();
  };
  auto none = [](int, int, float) {};
  OptCtx unused = {};
  const bool bern = (P.sgate || P.agate) && (P.gm.st_bern || P.gm.ac_bern);     // uniform

  if (PHASE == 0) {
    if (P.nstep > 0) nstep_horizon(P, S, A, agent, idx, nrows, len, head, gamma, M, tid);
    gather_tile(P, S, A, ldxc, ldxa, agent, idx, nrows, head, true, M, tid, 0, row0);
    __syncthreads();
    fwd_net(na, na.L(), w_at, M.xa, ldxa, M.hbase, M.hstride, M.topa, lda, wave, lane);    // a' = ann_target(s')
    if (TD3) {
      // a' = clip(a' + eps, -1, 1) (TD3Ag.hpp:307-309) between the actor's output gate and the critics' input gate
      const float *noise = P.noise + ((size_t)agent * B + row0) * A;
      for (int e = tid; e < DDRL_MB * A; e += DDRL_NTHREADS) {
        const int row = e / A, col = e - row * A;
        float a = M.topa[row * lda + col];
        const GateMode &gm = P.gm;
        if (agate) a *= ddrl_gate_factor(agate[col], gm.ac_bern, gm.ac_scale, gm.seed, gm.draw, agent, GS_ACTOR_T_OUT, row0 + row, col);
        if (row < nrows) a = fminf(fmaxf(a + noise[row * A + col], -1.f), 1.f);
        if (bern) M.topa[row * lda + col] = a;              // a' itself: the second target critic gates it with its own mask
        if (agate) a *= ddrl_gate_factor(agate[col], gm.ac_bern, gm.ac_scale, gm.seed, gm.draw, agent, GS_CRITIC_T_A, row0 + row, col);
        M.xc[row * ldxc + S + col] = a;
      }
      __syncthreads();
    } else actions_to_critic();
    fwd_net(nq, nq.L(), w_qt, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, wave, lane);    // q' = qnn_target(s', a')
    if (TD3) {                                                                              // Q2'(s', a') (:314)
      if (tid < DDRL_MB) M.nrew[tid] = M.topq[tid * ldq];
      if (bern) {                                           // qnn_target2's own layer instances (kernels_td3.hip)
        td3_regate_xc(P, S, A, ldxc, agent, idx, nrows, head, true, M, tid, GS_CRITIC2_T_S, 0, row0);
        for (int e = tid; e < DDRL_MB * A; e += DDRL_NTHREADS) {
          const int row = e / A, col = e - row * A;
          float a = M.topa[row * lda + col];
          if (agate) a *= ddrl_gate_factor(agate[col], P.gm.ac_bern, P.gm.ac_scale, P.gm.seed, P.gm.draw, agent, GS_CRITIC2_T_A, row0 + row, col);
          M.xc[row * ldxc + S + col] = a;
        }
      }
      __syncthreads();
      fwd_net(nq, nq.L(), P.sq2.wt + (size_t)agent * nq.stride(), M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, wave, lane);
    }
    gather_tile(P, S, A, ldxc, ldxa, agent, idx, nrows, head, false, M, tid, 0, row0);
    int trow;
    if (gather_owns_row(tid, trow)) {
      const float qn = TD3 ? fminf(M.nrew[trow], M.topq[trow * ldq]) : M.topq[trow * ldq];
      M.yv[trow] = M.dn[trow] != 0.f ? M.rv[trow] : (!TD3 && P.nstep > 0 ? M.nrew[trow] : M.rv[trow]) + gamma * qn;
    }
    __syncthreads();
    const float ykeep = tid < DDRL_MB ? M.yv[tid] : 0.f;
    if (P.qt_out && tid < nrows) P.qt_out[(size_t)agent * B + row0 + tid] = ykeep;
#pragma unroll
    for (int k = 0; k < (TD3 ? 2 : 1); ++k) {              // qnn (and qnn2, TD3Ag.hpp:329-330) learn the same targets
      const float *wk = k == 0 ? w_q : P.sq2.w + (size_t)agent * nq.stride();
      if (TD3 && k == 1 && bern) {                           // qnn2's own masks over (s, a)
        td3_regate_xc(P, S, A, ldxc, agent, idx, nrows, head, false, M, tid, GS_CRITIC2_S, GS_CRITIC2_A_REPLAY, row0);
        __syncthreads();
      }
      fwd_net(nq, nq.L(), wk, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, wave, lane);
      if (tid < DDRL_MB) {
        const float df = tid < nrows ? M.topq[tid * ldq] - ykeep : 0.f;
        float sq = df * df;
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
        if (tid == 0) (k == 0 ? P.loss_part : P.loss_part2)[(size_t)agent * P.maxchunks + chunk] = sq;
        M.topq[tid * ldq] = df / (float)B;                   // EuclideanLoss over the WHOLE minibatch
      }
      __syncthreads();
      float *gout = (k == 0 ? P.gpart_q : P.gpart_q2) + ((size_t)agent * P.maxchunks + chunk) * nq.stride();
      bwd_net<DW_STORE, false>(nq, wk, unused, gout, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, 0, 0, none, wave, lane, tid);
    }
  } else {
    gather_tile(P, S, A, ldxc, ldxa, agent, idx, nrows, head, false, M, tid, 0, row0);
    __syncthreads();
    fwd_net(na, na.L(), w_a, M.xa, ldxa, M.hbase, M.hstride, M.topa, lda, wave, lane);     // pi = ann(s)
    actions_to_critic();
    fwd_net(nq, nq.L(), w_q, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, wave, lane);     // Q(s, pi(s)), new critic
    if (tid < DDRL_MB) M.topq[tid * ldq] = tid < nrows ? -1.0f : 0.f;                       // :424-428, NOT /B
    __syncthreads();
    {
      float *topa = M.topa;
      const int inverting = P.inverting, out_act = na.out_act();
      float *dact_out = P.dact_out ? P.dact_out + ((size_t)agent * B + row0) * A : nullptr;
      const DdpgParams *Pg = &P;
      auto store = [=](int row, int h, float diff) {
        float out = topa[row * lda + h];
        float gin = 1.f, gt = 1.f;                                          // the masks of the forward passes these gradients belong to
        if (agate) {
          const GateMode &gm = Pg->gm;
          gin = ddrl_gate_factor(agate[h], gm.ac_bern, gm.ac_scale, gm.seed, gm.draw, agent, GS_CRITIC_A, row0 + row, h);
          gt = ddrl_gate_factor(agate[h], gm.ac_bern, gm.ac_scale, gm.seed, gm.draw, agent, GS_ACTOR_OUT, row0 + row, h);
        }
        diff *= gin;
        if (inverting) {                                                   // :432-450
          if (diff < 0.f) diff *= (1.0f - out) / 2.0f;
          else if (diff > 0.f) diff *= (out + 1.0f) / 2.0f;
        }
        if (dact_out && row < nrows) dact_out[row * A + h] = diff;
        diff *= gt;
        if (gt != 0.f) out /= gt;
        topa[row * lda + h] = act_bwd(out_act, out, diff);
      };
      bwd_net<DW_NONE, false>(nq, w_q, unused, nullptr, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, S, A, store, wave, lane, tid);
    }
    fwd_net(na, na.L() - 1, w_a, M.xa, ldxa, M.hbase, M.hstride, nullptr, 0, wave, lane);   // the actor's hidden activations again
    float *gout = P.gpart_a + ((size_t)agent * P.maxchunks + chunk) * na.stride();
    bwd_net<DW_STORE, false>(na, w_a, unused, gout, M.xa, ldxa, M.hbase, M.hstride, M.topa, lda, 0, 0, none, wave, lane, tid);
  }
}

// Sum of the chunk partials in chunk order, then the same fused optimiser epilogue as the one-tile
// kernel: updateFisher(B), regularize(), Adam, Solver::iter, soft target update.
template <int NET, bool EWC>
__global__ void __launch_bounds__(256)
ddpg_apply_kernel(const DdpgParams P) {
  const int agent = blockIdx.y, tid = threadIdx.x;
  if (P.mask && !P.mask[agent]) return;
  const int len = P.rep_len[agent];
  if (len < 250 || len < P.B) return;
  const NetDesc &nd = NET == 0 ? P.a : P.q;
  const NetState &st = NET == 0 ? P.sa : NET == 1 ? P.sq : P.sq2;     // NET 2: TD3's second critic
  const int nch = (P.B + DDRL_MB - 1) / DDRL_MB;
  // TD3 (the only caller that sets actor_step): the actor waits for its policy step, and the critics' targets
  // move on that step only (TD3Ag.hpp:333, 368-370; tau = 0 leaves w' bit-identical)
  const bool pstep = P.actor_step ? P.actor_step[agent] != 0 : true;
  if (NET == 0 && !pstep) return;
  const OptCtx o = make_opt<EWC>(P, st, nd.stride(), agent, pstep ? P.tau[agent] : 0.f, pstep ? P.omtau[agent] : 1.f, (float)P.B);
  const int i = (blockIdx.x * 256 + tid) * 4;
  if (i < nd.stride()) {
    OptRegs r;
    opt_load4<EWC, true>(o, i, r);
    const float *gp = (NET == 0 ? P.gpart_a : NET == 1 ? P.gpart_q : P.gpart_q2) + (size_t)agent * P.maxchunks * nd.stride() + i;
    f32x4 g = *(const f32x4 *)gp;
    for (int c = 1; c < nch; ++c) g += *(const f32x4 *)(gp + (size_t)c * nd.stride());
    opt_step4<EWC, true>(o, i, g, r, true);
  }
  if (blockIdx.x == 0 && tid == 0) {
    st.iter[agent] += 1;
    if (EWC) st.fisher_nbr[agent] = st.fisher_nbr[agent] * P.fisher_beta + (float)P.B;
    if (NET >= 1) {
      float s = 0.f;
      for (int c = 0; c < nch; ++c) s += (NET == 1 ? P.loss_part : P.loss_part2)[(size_t)agent * P.maxchunks + c];
      (NET == 1 ? P.loss_out : P.loss2_out)[agent] = s / (float)P.B * 0.5f;
    }
  }
}

extern "C" size_t ddpg_update_lds_bytes(const DdpgParams *P);

template <int PHASE, bool TD3 = false>
static hipError_t launch_chunks(const DdpgParams *P, hipStream_t st) {
  static LdsConfig configured;
  const size_t lds = ddpg_update_lds_bytes(P);
  hipError_t e = ddrl_set_lds(ddpg_chunk_kernel<PHASE, TD3>, lds, configured);
  if (e != hipSuccess) return e;
  const int nch = (P->B + DDRL_MB - 1) / DDRL_MB;
  hipLaunchKernelGGL((ddpg_chunk_kernel<PHASE, TD3>), dim3(xcd_grid(nch, 1, P->G)), dim3(DDRL_NTHREADS), lds, st, *P);
  return hipGetLastError();
}

template <int NET>
static hipError_t launch_apply(const DdpgParams *P, hipStream_t st) {
  const NetDesc &nd = NET == 0 ? P->a : P->q;
  const bool ewc = (NET == 0 ? P->sa : NET == 1 ? P->sq : P->sq2).fisher != nullptr;
  const dim3 grid((nd.stride_ + 1023) / 1024, P->G);
  if (ewc) hipLaunchKernelGGL((ddpg_apply_kernel<NET, true>), grid, dim3(256), 0, st, *P);
  else hipLaunchKernelGGL((ddpg_apply_kernel<NET, false>), grid, dim3(256), 0, st, *P);
  return hipGetLastError();
}

extern "C" hipError_t ddpg_apply_launch(const DdpgParams *P, int net, hipStream_t st) {
  return net == 0 ? launch_apply<0>(P, st) : net == 1 ? launch_apply<1>(P, st) : launch_apply<2>(P, st);
}

extern "C" hipError_t ddpg_update_big_launch(const DdpgParams *P, hipStream_t st) {
  hipError_t e;
  if ((e = launch_chunks<0>(P, st)) != hipSuccess) return e;
  if ((e = launch_apply<1>(P, st)) != hipSuccess) return e;
  if ((e = launch_chunks<1>(P, st)) != hipSuccess) return e;
  return launch_apply<0>(P, st);
}

// TD3Ag's update for minibatches of more than 64 rows (or forced): critic chunks for both critics, their
// applies, then -- `any_actor`: some agent is on a policy step -- the actor chunks and the actor's apply
extern "C" hipError_t td3_update_big_launch(const DdpgParams *P, int any_actor, hipStream_t st) {
  hipError_t e;
  if ((e = launch_chunks<0, true>(P, st)) != hipSuccess) return e;
  if ((e = launch_apply<1>(P, st)) != hipSuccess) return e;
  if ((e = launch_apply<2>(P, st)) != hipSuccess) return e;
  if (!any_actor) return hipSuccess;
  if ((e = launch_chunks<1, true>(P, st)) != hipSuccess) return e;
  return launch_apply<0>(P, st);
}
