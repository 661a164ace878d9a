// kernels_td3.hip -- the fused TD3 population update for gfx950.
//
// One launch = one call of TD3Ag's DeepQNAg<NN>::end_episode
// (/root/reference/agent/deepqn/include/TD3Ag.hpp:284-380) for every agent of the population: one
// 256-thread workgroup per agent runs the actor-target pass, the target-policy smoothing, both
// critic-target passes, the clipped double-Q targets, both critics' learn_batch (forward, backward,
// Adam, EWC terms) and -- on the agent's delayed policy steps -- the actor update through the FIRST
// critic and the three soft target updates.  Same building blocks as the DDPG kernel
// (ddpg_device.h, net_device.h, mlp_device.h), same LDS map, same state layout; the second critic is
// one more NetState.  One-tile minibatches (B <= 64) and layers up to DDRL_MAXW wide.
//
// The reference couples the soft updates to the policy step (:368-370) and nothing reads a critic's
// target between that critic's Adam step and its soft update, so the soft update rides in the critic's
// optimiser epilogue as in DDPG, with tau = 0 on the steps in between (w' = 0 w + 1 w' = w' exactly).
// (A second instantiation of the backward pass without the target's read and write was tried: the two
// copies push the mid-size kernel over its 256 registers -- 109 spilled -- and the policy step from 735 to
// 774 us for 10 us gained on the steps in between.)
#include "ddpg_device.h"

// ------------------------------------------------------------------------------------------
// Target-policy smoothing noise (TD3Ag.hpp:307): N(0, policy_noise) truncated to +-noise_clip, one
// draw per (row, action) of every agent's minibatch.  The reference draws from its clock-seeded host
// engine (Chopin's rtnorm); here a counter-based generator (splitmix64 over seed / agent / update
// count / element / attempt) feeds a Box-Muller rejection sampler: the same distribution, an own
// stream, reproducible from the seed.  After 32 rejected draws (|clip| << sigma, where the truncated
// normal is flat to within e^{-clip^2/2sigma^2}) the draw is uniform on the interval.
// The same launch prepares the second critic's Adam step size (see replay_draw_kernel).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}

extern "C" __global__ void __launch_bounds__(256)
td3_prep_kernel(const uint64_t *__restrict__ seed, uint32_t *__restrict__ count, const int *__restrict__ rep_len,
                int B, int A, int min_len, const uint8_t *__restrict__ mask, float *__restrict__ noise, int draw,
                float sigma, float clip, double beta1, double beta2, const float *__restrict__ lr_q2,
                const int *__restrict__ iter_q2, float *__restrict__ lrt_q2) {
  const int agent = blockIdx.x, tid = threadIdx.x;
  if (mask && !mask[agent]) return;
  const int n = rep_len[agent];
  if (n < min_len || n < B) return;                 // TD3Ag.hpp:286: no update, nothing consumed
  if (tid == 0) {
    const int t = iter_q2[agent] + 1;
    lrt_q2[agent] = (float)((double)lr_q2[agent] * (sqrt(1.0 - pow(beta2, (double)t)) / (1.0 - pow(beta1, (double)t))));
  }
  if (!draw) return;
  const uint32_t cnt = count[agent];
  const uint64_t key = splitmix64(splitmix64(seed[agent]) ^ (uint64_t)cnt);
  float *out = noise + (size_t)agent * B * A;
  for (int e = tid; e < B * A; e += blockDim.x) {
    float z = 0.f;
    bool done = !(sigma > 0.f);
    for (int attempt = 0; attempt < 32 && !done; ++attempt) {
      const uint64_t r = splitmix64(key ^ (((uint64_t)e << 8) | (uint64_t)attempt));
      const float u1 = ((float)(uint32_t)(r >> 40) + 1.0f) * (1.0f / 16777216.0f);     // (0, 1]
      const float u2 = (float)(uint32_t)((r >> 8) & 0xffffffu) * (1.0f / 16777216.0f); // [0, 1)
      z = sigma * sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530718f * u2);
      done = fabsf(z) <= clip;
    }
    if (!done) {
      const uint64_t r = splitmix64(key ^ (((uint64_t)e << 8) | 0xffull));
      z = clip * (2.0f * ((float)(uint32_t)(r >> 40) * (1.0f / 16777216.0f)) - 1.0f);
    }
    out[e] = z;
  }
  __syncthreads();
  if (tid == 0) count[agent] = cnt + 1;
}

extern "C" hipError_t td3_prep_launch(const uint64_t *seed, uint32_t *count, const int *rep_len, int G, int B, int A,
                                      const uint8_t *mask, float *noise, int draw, float sigma, float clip, double beta1,
                                      double beta2, const float *lr_q2, const int *iter_q2, float *lrt_q2, hipStream_t st) {
  hipLaunchKernelGGL(td3_prep_kernel, dim3(G), dim3(256), 0, st, seed, count, rep_len, B, A, 250, mask, noise, draw, sigma,
                     clip, beta1, beta2, lr_q2, iter_q2, lrt_q2);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
template <class NA, class NQ, bool EWC_A, bool EWC_Q>
__global__ void __launch_bounds__(DDRL_NTHREADS, DDRL_NTHREADS / 128)   // two workgroups per CU
td3_update_kernel(const DdpgParams P) {
  extern __shared__ __align__(16) float lds[];
  const int agent = blockIdx.x;
  const uint8_t live = P.mask ? P.mask[agent] : (uint8_t)1;
  const uint8_t pstep = P.actor_step ? P.actor_step[agent] : (uint8_t)1;   // current_step_counter % policy_freq == 0 (:333)
  const int len = P.rep_len[agent], head = P.rep_head[agent];
  const float gamma = P.gamma[agent], tau = P.tau[agent], omtau = P.omtau[agent];
  if (!live || len < 250 || len < P.B) return;             // TD3Ag.hpp:286
  const bool actor_step = pstep != 0;
  const float tau_e = actor_step ? tau : 0.f, omtau_e = actor_step ? omtau : 1.f;

  const auto &na = Pick<NA>::get(P.a);
  const auto &nq = Pick<NQ>::get(P.q);
  const int S = na.d(0), A = na.d(na.L()), B = P.B;
  const int ldxc = nq.ld(0), ldxa = na.ld(0);
  const int ldq = nq.ld(nq.L()), lda = na.ld(na.L());
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const LdsMap M = ddpg_carve(lds, na.L(), nq.L(), P.ldh, ldxc, ldxa, ldq, lda);
  const float *agate = P.agate ? P.agate + (size_t)agent * A : nullptr;
  const float *w_a = P.sa.w + (size_t)agent * na.stride(), *w_at = P.sa.wt + (size_t)agent * na.stride();
  const float *w_q = P.sq.w + (size_t)agent * nq.stride(), *w_qt = P.sq.wt + (size_t)agent * nq.stride();
  const float *w_q2 = P.sq2.w + (size_t)agent * nq.stride(), *w_q2t = P.sq2.wt + (size_t)agent * nq.stride();
  const int *idx = P.idx + (size_t)agent * B;
  const float *noise = P.noise + (size_t)agent * B * A;
  float *q1n = M.nrew;                                     // Q1'(s', a') per row (the n-step scratch: TD3Ag has no n-step targets)

  // ---- a' = clip(ann_target(s') + eps, -1, 1) (:304-310), Q1', Q2' (:313-314) ----
  gather_tile(P, S, A, ldxc, ldxa, agent, idx, B, head, true, M, tid);
  __syncthreads();
  fwd_net(na, na.L(), w_at, M.xa, ldxa, M.hbase, M.hstride, M.topa, lda, wave, lane);
  const bool bern = (P.sgate || P.agate) && (P.gm.st_bern || P.gm.ac_bern);     // uniform
  for (int e = tid; e < DDRL_MB * A; e += DDRL_NTHREADS) {
    const int row = e / A, col = e - row * A;
    float a = M.topa[row * lda + col];
    const GateMode &gm = P.gm;
    if (agate) a *= ddrl_gate_factor(agate[col], gm.ac_bern, gm.ac_scale, gm.seed, gm.draw, agent, GS_ACTOR_T_OUT, row, col);   // the actor's output gate
    if (row < B) a = fminf(fmaxf(a + noise[row * A + col], -1.f), 1.f);
    if (bern) M.topa[row * lda + col] = a;                 // a' itself: the second target critic gates it with its own mask
    if (agate) a *= ddrl_gate_factor(agate[col], gm.ac_bern, gm.ac_scale, gm.seed, gm.draw, agent, GS_CRITIC_T_A, row, col);    // the critic's action-input gate
    M.xc[row * ldxc + S + col] = a;
  }
  __syncthreads();
  fwd_net(nq, nq.L(), w_qt, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, wave, lane);
  if (tid < DDRL_MB) q1n[tid] = M.topq[tid * ldq];
  if (bern) {
    td3_regate_xc(P, S, A, ldxc, agent, idx, B, head, true, M, tid, GS_CRITIC2_T_S, 0);
    for (int e = tid; e < DDRL_MB * A; e += DDRL_NTHREADS) {
      const int row = e / A, col = e - row * A;
      float a = M.topa[row * lda + col];
      if (agate) a *= ddrl_gate_factor(agate[col], P.gm.ac_bern, P.gm.ac_scale, P.gm.seed, P.gm.draw, agent, GS_CRITIC2_T_A, row, col);
      M.xc[row * ldxc + S + col] = a;
    }
  }
  __syncthreads();
  fwd_net(nq, nq.L(), w_q2t, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, wave, lane);

  // ---- targets: r, or r + gamma min(Q1', Q2') (:317-326) ----
  gather_tile(P, S, A, ldxc, ldxa, agent, idx, B, head, false, M, tid);
  int trow;
  if (gather_owns_row(tid, trow)) {
    const float qn = fminf(q1n[trow], M.topq[trow * ldq]);
    M.yv[trow] = M.dn[trow] != 0.f ? M.rv[trow] : M.rv[trow] + gamma * qn;
  }
  __syncthreads();
  // the targets stay in a register of the row's thread: the per-row LDS scalars lie in the park area
  // of the first critic's backward pass (bwd_net) and the second critic needs them again
  const float ykeep = tid < DDRL_MB ? M.yv[tid] : 0.f;
  if (P.qt_out && tid < B) P.qt_out[(size_t)agent * B + tid] = ykeep;

  auto none = [](int, int, float) {};
  float *park = ParkOf<NA, NQ>::value ? M.topq : nullptr;
  // ---- qnn->learn_batch(s, a, y, 1); qnn2->learn_batch(s, a, y, 1) (:329-330; MLP.hpp:468-485) ----
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    int ag1 = agent, z1 = 0, tid1 = tid;                   // (laundered per pass: see the actor phase below)
    asm volatile("" : "+s"(ag1), "+s"(z1));
    asm volatile("" : "+v"(tid1));
    const int agent = ag1, tid = tid1, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const LdsMap M = ddpg_carve(lds + z1, na.L(), nq.L(), P.ldh, ldxc, ldxa, ldq, lda);
    float *park = ParkOf<NA, NQ>::value ? M.topq : nullptr;
    const float *wk = (k == 0 ? P.sq.w : P.sq2.w) + (size_t)agent * nq.stride();
    if (k == 1 && bern) {                                  // qnn2's own masks over (s, a)
      td3_regate_xc(P, S, A, ldxc, agent, P.idx + (size_t)agent * B, B, head, false, M, tid, GS_CRITIC2_S, GS_CRITIC2_A_REPLAY);
      __syncthreads();
    }
    fwd_net(nq, nq.L(), wk, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, wave, lane);
    if (tid < DDRL_MB) {
      // EuclideanLoss: loss = sum (q - y)^2 / (2B), dq = (q - y) / B
      const float df = tid < B ? M.topq[tid * ldq] - ykeep : 0.f;
      float sq = df * df;
      for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
      if (tid == 0) (k == 0 ? P.loss_out : P.loss2_out)[agent] = sq / (float)B * 0.5f;
      M.topq[tid * ldq] = df / (float)B;
    }
    __syncthreads();
    const OptCtx oq = make_opt<EWC_Q>(P, k == 0 ? P.sq : P.sq2, nq.stride(), agent, tau_e, omtau_e, (float)B);
    bwd_net<DW_OPT, EWC_Q>(nq, wk, oq, nullptr, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, 0, 0, none, wave, lane, tid, nullptr, park);
  }

  // ---- delayed actor update through the first critic (:333-363) ----
  // Every value the actor phase shares with the critic phases (agent index, LDS base, thread id) is re-derived from a
  // laundered copy: with the common subexpressions visible the compiler kept the actor phase's addresses and fragments
  // alive across both critic passes and spilled ~110 vector registers (kernels_ddpg_pair.hip has the same cure).
  int ag2 = agent, z2 = 0, tid2 = tid;
  asm volatile("" : "+s"(ag2), "+s"(z2));
  asm volatile("" : "+v"(tid2));
  if (actor_step) {                                        // uniform over the workgroup
    const int agent = ag2, tid = tid2, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const LdsMap M = ddpg_carve(lds + z2, na.L(), nq.L(), P.ldh, ldxc, ldxa, ldq, lda);
    const float *agate = P.agate ? P.agate + (size_t)agent * A : nullptr;
    const float *w_a = P.sa.w + (size_t)agent * na.stride(), *w_q = P.sq.w + (size_t)agent * nq.stride();
    float *park = ParkOf<NA, NQ>::value ? M.topq : nullptr;
    fwd_net(na, na.L(), w_a, M.xa, ldxa, M.hbase, M.hstride, M.topa, lda, wave, lane);       // pi = ann(s)
    if (bern) td3_regate_xc(P, S, A, ldxc, agent, P.idx + (size_t)agent * B, B, head, false, M, tid, GS_CRITIC_S, GS_CRITIC_A_REPLAY);   // the first critic's state masks again
    if (bern) __syncthreads();                                                               // (its action columns are rewritten below)
    for (int e = tid; e < DDRL_MB * A; e += DDRL_NTHREADS) {
      const int row = e / A, col = e - row * A;
      float a = M.topa[row * lda + col];
      if (agate) {
        const GateMode &gm = P.gm;
        a *= ddrl_gate_factor(agate[col], gm.ac_bern, gm.ac_scale, gm.seed, gm.draw, agent, GS_ACTOR_OUT, row, col);
        M.topa[row * lda + col] = a;
        a *= ddrl_gate_factor(agate[col], gm.ac_bern, gm.ac_scale, gm.seed, gm.draw, agent, GS_CRITIC_A, row, col);
      }
      M.xc[row * ldxc + S + col] = a;
    }
    __syncthreads();
    fwd_net(nq, nq.L(), w_q, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, wave, lane);       // Q1(s, pi(s)), new critic
    if (tid < DDRL_MB) M.topq[tid * ldq] = tid < B ? -1.0f : 0.f;                            // :351-352, not / B
    __syncthreads();
    {
      float *topa = M.topa;
      const int out_act = na.out_act();
      float *dact_out = P.dact_out ? P.dact_out + (size_t)agent * B * A : nullptr;
      const DdpgParams *Pg = &P;
      auto store = [=](int row, int h, float diff) {       // ShareDiff (:358): no inverting gradients in this agent
        float out = topa[row * lda + h];
        float gin = 1.f, gt = 1.f;                         // the masks of the forward passes these gradients belong to
        if (agate) {
          const GateMode &gm = Pg->gm;
          gin = ddrl_gate_factor(agate[h], gm.ac_bern, gm.ac_scale, gm.seed, gm.draw, agent, GS_CRITIC_A, row, h);
          gt = ddrl_gate_factor(agate[h], gm.ac_bern, gm.ac_scale, gm.seed, gm.draw, agent, GS_ACTOR_OUT, row, h);
        }
        diff *= gin;
        if (dact_out && row < B) dact_out[row * A + h] = diff;
        diff *= gt;
        if (gt != 0.f) out /= gt;
        topa[row * lda + h] = act_bwd(out_act, out, diff);
      };
      OptCtx unused = {};
      bwd_net<DW_NONE, false>(nq, w_q, unused, nullptr, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, S, A, store, wave, lane, tid);
    }
    fwd_net(na, na.L() - 1, w_a, M.xa, ldxa, M.hbase, M.hstride, nullptr, 0, wave, lane);    // hidden activations again (LDS was shared)
    const OptCtx oa = make_opt<EWC_A>(P, P.sa, na.stride(), agent, tau, omtau, (float)B);    // updateFisher(traj.size()) :360
    bwd_net<DW_OPT, EWC_A>(na, w_a, oa, nullptr, M.xa, ldxa, M.hbase, M.hstride, M.topa, lda, 0, 0, none, wave, lane, tid, nullptr, park);
  }
  if (tid == 0) {
    P.sq.iter[agent] += 1;                                 // Solver::Step ++iter, both critics
    P.sq2.iter[agent] += 1;
    if (EWC_Q) {
      P.sq.fisher_nbr[agent] = P.sq.fisher_nbr[agent] * P.fisher_beta + (float)B;
      P.sq2.fisher_nbr[agent] = P.sq2.fisher_nbr[agent] * P.fisher_beta + (float)B;
    }
    if (actor_step) {
      P.sa.iter[agent] += 1;                               // set_iter(iter() + 1) :363
      if (EWC_A) P.sa.fisher_nbr[agent] = P.sa.fisher_nbr[agent] * P.fisher_beta + (float)B;
    }
  }
}

extern "C" size_t ddpg_update_lds_bytes(const DdpgParams *P);

template <class NA, class NQ, bool EA, bool EQ>
static hipError_t td3_launch_t(const DdpgParams *P, size_t lds, hipStream_t st) {
  static LdsConfig configured;
  hipError_t e = ddrl_set_lds(td3_update_kernel<NA, NQ, EA, EQ>, lds, configured);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((td3_update_kernel<NA, NQ, EA, EQ>), dim3(P->G), dim3(DDRL_NTHREADS), lds, st, *P);
  return hipGetLastError();
}

template <class NA, class NQ>
static hipError_t td3_launch_ewc(const DdpgParams *P, size_t lds, hipStream_t st) {
  const bool ea = P->sa.fisher != nullptr, eq = P->sq.fisher != nullptr;
  if (ea && eq) return td3_launch_t<NA, NQ, true, true>(P, lds, st);
  if (ea) return td3_launch_t<NA, NQ, true, false>(P, lds, st);
  return td3_launch_t<NA, NQ, false, false>(P, lds, st);
}

extern "C" hipError_t td3_update_launch(const DdpgParams *P, int force_generic, hipStream_t st) {
  const size_t lds = ddpg_update_lds_bytes(P);
  if (!force_generic && MidActor::matches(P->a) && MidCritic::matches(P->q)) return td3_launch_ewc<MidActor, MidCritic>(P, lds, st);
  return td3_launch_ewc<NetDesc, NetDesc>(P, lds, st);
}
