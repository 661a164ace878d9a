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
#include "ddpg_update_body.h"

// one pass of the fupd loop (DeepQNAg.hpp:343-466) for one agent: the four quarters of ddpg_update_body.h back to back
template <class NA, class NQ, bool EWC_A, bool EWC_Q>
__device__ __forceinline__ void ddpg_update_once(const DdpgParams &P, float *lds, int agent, int u, int len, int head,
                                                 float gamma, float tau, float omtau, const GatherRegs &pre, bool have_pre) {
  const int tid = threadIdx.x;
#ifdef DDRL_EXP_STAGGER   // experiment: spread the first round's start times over DDRL_EXP_STAGGER cycles
  if (agent < 512 && u == 0) {
    const long long t0 = clock64();
    const long long delay = (long long)(((unsigned)agent * 2654435761u) >> 8) % DDRL_EXP_STAGGER;
    while (clock64() - t0 < delay) __builtin_amdgcn_s_sleep(32);
  }
#endif
  ddpg_q0<NA, NQ, EWC_A, EWC_Q>(P, lds, agent, u, len, head, gamma, tid, pre, have_pre);
  ddpg_q1<NA, NQ, EWC_A, EWC_Q>(P, lds, agent, u, tau, omtau, tid);
  ddpg_q2<NA, NQ, EWC_A, EWC_Q>(P, lds, agent, u, tid);
  ddpg_q3<NA, NQ, EWC_A, EWC_Q>(P, lds, agent, u, tau, omtau, tid, P.nupd > 1);
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
  // the minibatch's s' rows are requested BEFORE the hand-off wait: the replay ring and the index block are not
  // written during a launch, so the two dependent round trips of the gather overlap the predecessor's tail
  GatherRegs pre;
  const bool have_pre = ddpg_prefetch_rows<NA, NQ>(P, agent, u, head, (int)threadIdx.x, pre);
  if (u > 0) {
    // consumer side of the hand-off (MI355X_MICROARCH.md, inter-workgroup visibility): relaxed polls by
    // one lane, ONE agent-scope acquire (it invalidates this CU's L1), the wait for that invalidate,
    // then the workgroup barrier, then plain loads
    __shared__ int go;
    if (threadIdx.x == 0) {
      int ok = 0;
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();     // 100 MHz: bounded by TIME (4 s), never reached in a healthy run
      while (true) {
        if (__hip_atomic_load(P.upd_done + agent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= u) { ok = 1; break; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > 400000000ull) break;
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
  ddpg_update_once<NA, NQ, EWC_A, EWC_Q>(P, lds, agent, u, len, head, gamma, tau, omtau, pre, have_pre);
  if (P.nupd > 1) {
    // producer side: everything the agent's next update reads was stored sc1 (store_sc1: no L2 write-back
    // needed); every storing wave waits for its stores, the barrier, and only then the counter (an sc1 store)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(P.upd_done + agent, u + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The same multi-update launch with an agent's consecutive updates kept in ONE workgroup: grid = G, the workgroup walks
// its agent through the launch's nupd updates.  Nothing is handed to another workgroup, so the state is stored with plain
// (L2-resident) stores instead of sc1 write-through ones, there is no ordering wait and no reliance on dispatch order, and
// the next update's minibatch rows are requested while the current one finishes (ddpg_prefetch_rows).  Round 1 tried this
// loop first and lost 550 registers to values hoisted out of it; the quarters of ddpg_update_body.h with their laundered
// arguments (as in kernels_ddpg_pair.hip, whose two-team workgroups pay an LDS-polling barrier this kernel does not need)
// keep every address computation inside its quarter.  Compile-time shapes only.  Bit-identical to nupd launches.
template <class NA, class NQ, bool EWC_A, bool EWC_Q>
__global__ void __launch_bounds__(DDRL_NTHREADS, DDRL_NTHREADS / 128)
ddpg_update_loop_kernel(const DdpgParams P) {
  extern __shared__ __align__(16) float lds[];
  const int tid = (int)threadIdx.x;
  const int agent = (int)blockIdx.x;
  const uint8_t live = P.mask ? P.mask[agent] : (uint8_t)1;
  const int len = P.rep_len[agent], head = P.rep_head[agent];
  const float gamma = P.gamma[agent], tau = P.tau[agent], omtau = P.omtau[agent];
  if (!live || len < 250 || len < P.B) return;             // DeepQNAg.hpp:338 (uniform)
  GatherRegs pre;
  bool have_pre = ddpg_prefetch_rows<NA, NQ>(P, agent, 0, head, tid, pre);
#ifdef DDRL_EXP_STAGGER   // experiment: spread the workgroups' phases over DDRL_EXP_STAGGER cycles (a population of one workgroup per CU runs in lockstep)
  {
    const long long t0 = clock64();
    const long long delay = (long long)((((unsigned)agent * 2654435761u) >> 8) % (unsigned)DDRL_EXP_STAGGER);
    while (clock64() - t0 < delay) __builtin_amdgcn_s_sleep(32);
  }
#endif
#pragma unroll 1
  for (int u = 0; u < P.nupd; ++u) {
    int a_it = agent, tid_it = tid, z = 0;                    // nothing is hoisted out of this loop (see the pair kernel)
    asm volatile("" : "+s"(a_it), "+s"(z));
    asm volatile("" : "+v"(tid_it));
    float *tl = lds + z;
    ddpg_q0<NA, NQ, EWC_A, EWC_Q>(P, tl, a_it, u, len, head, gamma, tid_it, pre, have_pre);
    ddpg_q1<NA, NQ, EWC_A, EWC_Q>(P, tl, a_it, u, tau, omtau, tid_it);
    ddpg_q2<NA, NQ, EWC_A, EWC_Q>(P, tl, a_it, u, tid_it);
    ddpg_q3<NA, NQ, EWC_A, EWC_Q>(P, tl, a_it, u, tau, omtau, tid_it, false);
    have_pre = ddpg_prefetch_rows<NA, NQ>(P, a_it, u + 1, head, tid_it, pre);
    // (the actor's backward pass ends behind a workgroup barrier: every wave is through with the input tiles before the
    // next update's gather overwrites them; the iteration counters are read back by the thread that wrote them)
  }
}

static bool ddpg_parks(const DdpgParams *P);

// host-visible LDS sizing, same carve order as the kernel
extern "C" size_t ddpg_update_lds_bytes(const DdpgParams *P) {
  size_t f = 0;
  { const int nh = (P->a.L_ > P->q.L_ ? P->a.L_ : P->q.L_) - 1; f += (size_t)nh * DDRL_MB * P->ldh; }
  f += (size_t)DDRL_MB * (P->ldxc + P->ldxa) + DDRL_NWAVES * 64;
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
  if (eq) return hipErrorInvalidValue;      // EWC on the critic alone: no ini produces it (DODevMLP.hpp:714-720 needs ewc >= 0), and no instantiation exists
  return launch_t<NA, NQ, false, false>(P, lds, st);
}

template <class NA, class NQ, bool EA, bool EQ>
static hipError_t launch_loop_t(const DdpgParams *P, size_t lds, hipStream_t st) {
  static LdsConfig configured;
  hipError_t e = ddrl_set_lds(ddpg_update_loop_kernel<NA, NQ, EA, EQ>, lds, configured);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((ddpg_update_loop_kernel<NA, NQ, EA, EQ>), dim3(P->G), dim3(DDRL_NTHREADS), lds, st, *P);
  return hipGetLastError();
}
template <class NA, class NQ>
static hipError_t launch_loop_ewc(const DdpgParams *P, size_t lds, hipStream_t st) {
  const bool ea = P->sa.fisher != nullptr, eq = P->sq.fisher != nullptr;
  if (ea && eq) return launch_loop_t<NA, NQ, true, true>(P, lds, st);
  if (ea) return launch_loop_t<NA, NQ, true, false>(P, lds, st);
  if (eq) return hipErrorInvalidValue;
  return launch_loop_t<NA, NQ, false, false>(P, lds, st);
}
// the in-workgroup loop over a launch's updates: compile-time shapes only (caller sets P->pair: no hand-off stores)
extern "C" int ddpg_update_loop_ok(const DdpgParams *P) {
  return (MidActor::matches(P->a) && MidCritic::matches(P->q)) || (SmallActor::matches(P->a) && SmallCritic::matches(P->q));
}
// ... and does it pay?  Measured at 1024 agents, 8 updates per launch: mid nets 0.381 ms per population update against
// 0.402 with one workgroup per (update, agent); small nets 0.073 against 0.065 (three workgroups fit a CU, 1024 agents
// are 1.33 rounds of 768 slots, and a workgroup that holds its agent for the whole launch cannot fill the last round)
extern "C" int ddpg_update_loop_pays(const DdpgParams *P) { return MidActor::matches(P->a) && MidCritic::matches(P->q); }
extern "C" hipError_t ddpg_update_loop_launch(const DdpgParams *P, hipStream_t st) {
  const size_t lds = ddpg_update_lds_bytes(P);
  if (MidActor::matches(P->a) && MidCritic::matches(P->q)) return launch_loop_ewc<MidActor, MidCritic>(P, lds, st);
  if (SmallActor::matches(P->a) && SmallCritic::matches(P->q)) return launch_loop_ewc<SmallActor, SmallCritic>(P, lds, st);
  return hipErrorInvalidValue;
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
