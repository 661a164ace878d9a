// kernels_ddpg.hip -- the fused DDPG population update for gfx950.
//
// One launch = one pass (or, P.nupd > 1, several consecutive passes: ddpg_update_kernel below) of the
// `fupd` loop body of DeepQNAg::end_episode
// (/root/reference/agent/deepqn/include/DeepQNAg.hpp:343-466) for every agent of the population:
// one 256-thread workgroup per (pass, agent) runs all 5 forward passes, both critic backward passes, the
// actor backward pass, both Adam steps (+EWC terms) and both soft target updates without leaving
// the CU.  The minibatch and all activations live in LDS; weights / optimiser state are touched in
// HBM exactly once per update by the fused epilogue (opt_step4), plus L2-served re-reads of the
// weights by the products.
//
// The kernel is a template over the two net shapes.  Instantiated on SNet<...> (compile-time
// shapes: the BASELINE "mid" and "small" HalfCheetah nets) every loop bound and guard is a
// constant; instantiated on NetDesc it is the generic kernel for any shape up to DDRL_MAXW.
// Minibatches above 64 rows: kernels_ddpg_big.hip; layers wider than DDRL_MAXW: kernels_ddpg_wide.hip.
// DDRL_EXP_* switches are measurement builds, never defined in the product.
#include "ddpg_device.h"

// one pass of the fupd loop (DeepQNAg.hpp:343-466) for one agent
template <class NA, class NQ, bool EWC_A, bool EWC_Q>
__device__ __forceinline__ void ddpg_update_once(const DdpgParams &P, float *lds, int agent, int u, int len, int head,
                                                           float gamma, float tau, float omtau) {
  const int tid0 = threadIdx.x;
  const auto &na = Pick<NA>::get(P.a);
  const auto &nq = Pick<NQ>::get(P.q);
  const int S = na.d(0), A = na.d(na.L()), B = P.B;
  const int ldxc = nq.ld(0), ldxa = na.ld(0);
  const int ldq = nq.ld(nq.L()), lda = na.ld(na.L());
  const int tid = tid0, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: branches on it are uniform, waits are counted exactly
  const LdsMap M = ddpg_carve(lds, na.L(), nq.L(), P.ldh, ldxc, ldxa, ldq, lda);
#ifdef DDRL_EXP_STAGGER   // experiment: spread the first round's start times over DDRL_EXP_STAGGER cycles
  if (agent < 512 && u == 0) {
    const long long t0 = clock64();
    const long long delay = (long long)(((unsigned)agent * 2654435761u) >> 8) % DDRL_EXP_STAGGER;
    while (clock64() - t0 < delay) __builtin_amdgcn_s_sleep(32);
  }
#endif

  long long *stamps = P.stamps ? P.stamps + (size_t)agent * 16 : nullptr;
#define STAMP(i) do { if (stamps && tid == 0) stamps[i] = clock64(); } while (0)
  const float *agate = P.agate ? P.agate + (size_t)agent * A : nullptr;
  const float *w_a = P.sa.w + (size_t)agent * na.stride(), *w_at = P.sa.wt + (size_t)agent * na.stride();
  const float *w_q = P.sq.w + (size_t)agent * nq.stride(), *w_qt = P.sq.wt + (size_t)agent * nq.stride();
#ifdef DDRL_EXP_HOTW      // experiment: every agent's passes read agent 0's weights (always cache-hot)
  w_a = P.sa.w; w_at = P.sa.wt; w_q = P.sq.w; w_qt = P.sq.wt;
#endif

  STAMP(0);
  // the diagnostics (ddrl_pop_ddpg_last) are those of the agent's LAST update of the launch: written by
  // that workgroup only (two updates of one agent may run on different XCDs, whose L2s would write the
  // same addresses back in no particular order)
  const bool diag = u == P.nupd - 1;
  const int *idx = P.idx + ((size_t)agent * P.nupd + u) * B;
  // ---- n-step horizon (DeepQNAg.hpp:360-373, 394-402): the horizon stops before a transition that
  // reached the goal and at the end of the replay; rewards of steps [0, valid) are discounted ----
  // the first weight fragments of the actor-target pass travel while the minibatch is gathered
  FwdCarry cy;
  const FwdNext no_next = {nullptr, nullptr, 4, 1};
  if (!__is_same(NA, NetDesc)) fwd_request(fwd_first(na, w_at), lane, cy);
  if (P.nstep > 0) nstep_horizon(P, S, A, agent, idx, B, len, head, gamma, M, tid);

  // ---- a' = ann_target(s'), q' = qnn_target(s', a')  (DeepQNAg.hpp:379-383) ----
  gather_tile(P, S, A, ldxc, ldxa, agent, idx, B, head, true, M, tid);
  __syncthreads();
  fwd_net(na, na.L(), w_at, M.xa, ldxa, M.hbase, M.hstride, M.topa, lda, wave, lane, cy, true, fwd_first(nq, w_qt));
  for (int e = tid; e < DDRL_MB * A; e += DDRL_NTHREADS) {
    const int row = e / A, col = e - row * A;
    float a = M.topa[row * lda + col];
    if (agate) {                       // the actor's output gate, then the critic's action-input gate
      a *= agate[col];
      M.topa[row * lda + col] = a;
      a *= agate[col];
    }
    M.xc[row * ldxc + S + col] = a;
  }
  __syncthreads();
  STAMP(1);
  fwd_net(nq, nq.L(), w_qt, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, wave, lane, cy, true, fwd_first(nq, w_q));
  STAMP(2);

  // ---- TD targets (DeepQNAg.hpp:386-405, nstep_return == 0) ----
  gather_tile(P, S, A, ldxc, ldxa, agent, idx, B, head, false, M, tid);
  int trow;
  if (gather_owns_row(tid, trow)) {     // the thread that gathered the row's reward / goal flag
    const float qn = M.topq[trow * ldq];
    // :388-403.  n-step: sum_{j<valid} gamma^j r_{t+j} + gamma * Q' (the reference's exponent, not gamma^valid)
    M.yv[trow] = M.dn[trow] != 0.f ? M.rv[trow] : (P.nstep > 0 ? M.nrew[trow] : M.rv[trow]) + gamma * qn;
  }
  __syncthreads();
  if (P.qt_out && diag && tid < B) P.qt_out[(size_t)agent * B + tid] = M.yv[tid];
  STAMP(3);

  // ---- critic update: qnn->learn_batch(s, a, y, 1)  (DeepQNAg.hpp:408; MLP.hpp:468-485) ----
  fwd_net(nq, nq.L(), w_q, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, wave, lane, cy, true, no_next);
  if (tid < DDRL_MB) {
    // EuclideanLoss: loss = sum (q - y)^2 / (2B), dq = (q - y) / B
    const float df = tid < B ? M.topq[tid * ldq] - M.yv[tid] : 0.f;
    float sq = df * df;
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if (tid == 0 && diag) P.loss_out[agent] = sq / (float)B * 0.5f;
    M.topq[tid * ldq] = df / (float)B;
  }
  __syncthreads();
  STAMP(4);
  auto none = [](int, int, float) {};
  // compile-time shapes: the tiles from topq to the end of the allocation (topq after the top layer,
  // topa, the per-row scalars and the tail the host adds, ddpg_update_lds_bytes) are dead during the
  // optimiser loops below the top layer and park input-gradient accumulators there
  float *park = ParkOf<NA, NQ>::value ? M.topq : nullptr;
  {
    const OptCtx oq = make_opt<EWC_Q>(P, P.sq, nq.stride(), agent, tau, omtau, (float)B, u);
    bwd_net<DW_OPT, EWC_Q>(nq, w_q, oq, nullptr, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, 0, 0, none, wave, lane, tid,
                           stamps ? stamps + 11 : nullptr, park);
  }
  STAMP(5);

  // ---- actor update (DeepQNAg.hpp:411-459) ----
  fwd_net(na, na.L(), w_a, M.xa, ldxa, M.hbase, M.hstride, M.topa, lda, wave, lane);       // pi = ann(s)
  for (int e = tid; e < DDRL_MB * A; e += DDRL_NTHREADS) {
    const int row = e / A, col = e - row * A;
    float a = M.topa[row * lda + col];
    if (agate) {                       // the actor's output gate, then the critic's action-input gate
      a *= agate[col];
      M.topa[row * lda + col] = a;
      a *= agate[col];
    }
    M.xc[row * ldxc + S + col] = a;
  }
  __syncthreads();
  STAMP(6);
  fwd_net(nq, nq.L(), w_q, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, wave, lane);       // Q(s, pi(s)), new critic
  STAMP(7);
  if (tid < DDRL_MB) M.topq[tid * ldq] = tid < B ? -1.0f : 0.f;            // :424-428, NOT /B
  __syncthreads();
  {
    // critic_backward (:429): only d/d(actions) is needed; critic parameter diffs are discarded
    float *topa = M.topa;
    const int inverting = P.inverting, out_act = na.out_act();
    float *dact_out = P.dact_out && diag ? P.dact_out + (size_t)agent * B * A : nullptr;
    auto store = [=](int row, int h, float diff) {
      float out = topa[row * lda + h];
      const float gt = agate ? agate[h] : 1.f;
      diff *= gt;                                                        // the critic's action-input gate
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
    bwd_net<DW_NONE, false>(nq, w_q, unused, nullptr, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, S, A, store, wave, lane, tid);
  }
  STAMP(8);
  // the actor's hidden activations shared their LDS with the critic pass: recompute them (one extra
  // actor forward is cheaper than doubling LDS and halving occupancy); topa now holds dZ_L
  fwd_net(na, na.L() - 1, w_a, M.xa, ldxa, M.hbase, M.hstride, nullptr, 0, wave, lane);
  STAMP(9);
  {
    const OptCtx oa = make_opt<EWC_A>(P, P.sa, na.stride(), agent, tau, omtau, (float)B, u);  // updateFisher(traj.size()) :456
    bwd_net<DW_OPT, EWC_A>(na, w_a, oa, nullptr, M.xa, ldxa, M.hbase, M.hstride, M.topa, lda, 0, 0, none, wave, lane, tid, nullptr, park);
  }
  STAMP(10);
  if (tid == 0) {
    if (P.nupd > 1) {     // read by the agent's next update in this launch: sc1 stores (see store_sc1)
      __hip_atomic_store(P.sq.iter + agent, P.sq.iter[agent] + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(P.sa.iter + agent, P.sa.iter[agent] + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (EWC_Q) store_sc1(P.sq.fisher_nbr + agent, P.sq.fisher_nbr[agent] * P.fisher_beta + (float)B);
      if (EWC_A) store_sc1(P.sa.fisher_nbr + agent, P.sa.fisher_nbr[agent] * P.fisher_beta + (float)B);
    } else {
      P.sq.iter[agent] += 1;                                             // Solver::Step ++iter
      P.sa.iter[agent] += 1;                                             // set_iter(iter()+1) :459
      if (EWC_Q) P.sq.fisher_nbr[agent] = P.sq.fisher_nbr[agent] * P.fisher_beta + (float)B;
      if (EWC_A) P.sa.fisher_nbr[agent] = P.sa.fisher_nbr[agent] * P.fisher_beta + (float)B;
    }
  }
}

// One workgroup per (update, agent): blockIdx = u * G + agent.  P.nupd > 1 runs that many consecutive
// passes of the fupd loop (DeepQNAg.hpp:341: 1 + force_more_update per stored transition) in ONE
// launch.  An agent's updates depend only on each other, so nothing forces the population through
// the grid-wide barrier of a kernel boundary between them: workgroup (u, agent) waits for P.upd_done[agent]
// to reach u, which (u - 1, agent) publishes when its stores are out.  Workgroups are dispatched in
// block order, (u - 1, agent) long before (u, agent) -- G - 1 workgroups lie between them -- so the wait
// is rarely more than a glance, and it cannot deadlock: the lowest-numbered unfinished workgroup never
// waits on an undispatched one.  It is bounded all the same: on expiry the workgroup gives up without
// touching the agent and raises P.upd_error.  After the first pass the workgroups are out of phase, which
// spreads the optimiser's HBM streaming over time (section 4.1 of DESIGN.md), and the tail of one pass
// overlaps the head of the next.
template <class NA, class NQ, bool EWC_A, bool EWC_Q>
#ifdef DDRL_EXP_SPILL    // measurement / robustness build: a register budget the kernel does not fit (forces scratch spills)
__global__ void __launch_bounds__(DDRL_NTHREADS, 3)
#else
__global__ void __launch_bounds__(DDRL_NTHREADS, DDRL_NTHREADS / 128)   // two workgroups per CU
#endif
ddpg_update_kernel(const DdpgParams P) {
  extern __shared__ __align__(16) float lds[];
  const int u = P.nupd > 1 ? (int)blockIdx.x / P.G : 0;
  const int agent = (int)blockIdx.x - u * P.G;
  // every per-agent scalar is requested before the first one is tested: one memory round trip, not four
  const uint8_t live = P.mask ? P.mask[agent] : (uint8_t)1;
  const int len = P.rep_len[agent], head = P.rep_head[agent];
  const float gamma = P.gamma[agent], tau = P.tau[agent], omtau = P.omtau[agent];
  if (!live || len < 250 || len < P.B) return;             // DeepQNAg.hpp:338
  if (u > 0) {
    // consumer side of the hand-off (MI355X_MICROARCH.md, inter-workgroup visibility): relaxed polls by
    // one lane, ONE agent-scope acquire (it invalidates this CU's L1), the wait for that invalidate,
    // then the workgroup barrier, then plain loads
    __shared__ int go;
    if (threadIdx.x == 0) {
      int ok = 0;
      for (long long spin = 0; spin < (1ll << 22); ++spin) {              // ~ seconds: never reached in a healthy run
        if (__hip_atomic_load(P.upd_done + agent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= u) { ok = 1; break; }
        __builtin_amdgcn_s_sleep(8);
      }
      if (!ok) __hip_atomic_store(P.upd_error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      go = ok;
    }
    __syncthreads();
    if (!go) return;
  }
  ddpg_update_once<NA, NQ, EWC_A, EWC_Q>(P, lds, agent, u, len, head, gamma, tau, omtau);
  if (P.nupd > 1) {
    // producer side: everything the agent's next update reads was stored sc1 (store_sc1: no L2 write-back
    // needed); every storing wave waits for its stores, the barrier, and only then the counter (an sc1 store)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(P.upd_done + agent, u + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

static bool ddpg_parks(const DdpgParams *P);

// host-visible LDS sizing, same carve order as the kernel
extern "C" size_t ddpg_update_lds_bytes(const DdpgParams *P) {
  size_t f = 0;
  { const int nh = (P->a.L_ > P->q.L_ ? P->a.L_ : P->q.L_) - 1; f += (size_t)nh * DDRL_MB * P->ldh; }
  f += (size_t)DDRL_MB * (P->ldxc + P->ldxa);
  // from topq on: the output tiles and per-row scalars, at least the park area of bwd_net
  // (DDRL_PARK_TILES KiB per wave; used by the compile-time instantiations)
  const size_t tail = (size_t)DDRL_MB * (P->q.ld_[P->q.L_] + P->a.ld_[P->a.L_] + 5);
  const size_t parkf = ddpg_parks(P) ? (size_t)DDRL_NWAVES * DDRL_PARK_TILES * 256 : 0;
  f += tail > parkf ? tail : parkf;
  return f * sizeof(float);
}

// does the instantiation the launcher will pick park input-gradient tiles (ParkOf)?
static bool ddpg_parks(const DdpgParams *P) {
  return MidActor::matches(P->a) && MidCritic::matches(P->q) && ParkOf<MidActor, MidCritic>::value;
}

template <class NA, class NQ, bool EA, bool EQ>
static hipError_t launch_t(const DdpgParams *P, size_t lds, hipStream_t st) {
  static LdsConfig configured;
  hipError_t e = ddrl_set_lds(ddpg_update_kernel<NA, NQ, EA, EQ>, lds, configured);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((ddpg_update_kernel<NA, NQ, EA, EQ>), dim3(P->G * (P->nupd > 1 ? P->nupd : 1)), dim3(DDRL_NTHREADS), lds, st, *P);
  return hipGetLastError();
}

template <class NA, class NQ>
static hipError_t launch_ewc(const DdpgParams *P, size_t lds, hipStream_t st) {
  const bool ea = P->sa.fisher != nullptr, eq = P->sq.fisher != nullptr;
  if (ea && eq) return launch_t<NA, NQ, true, true>(P, lds, st);
  if (ea) return launch_t<NA, NQ, true, false>(P, lds, st);
  return launch_t<NA, NQ, false, false>(P, lds, st);
}

// force_generic == 0: pick a compile-time instantiation when the shapes match; else the generic kernel
extern "C" hipError_t ddpg_update_launch(const DdpgParams *P, int force_generic, hipStream_t st) {
  const size_t lds = ddpg_update_lds_bytes(P);
  if (!force_generic) {
    if (MidActor::matches(P->a) && MidCritic::matches(P->q)) return launch_ewc<MidActor, MidCritic>(P, lds, st);
    if (SmallActor::matches(P->a) && SmallCritic::matches(P->q)) return launch_ewc<SmallActor, SmallCritic>(P, lds, st);
  }
  return launch_ewc<NetDesc, NetDesc>(P, lds, st);
}
