// kernels_ddpg_pair.hip -- the fused DDPG population update, two agents per workgroup in opposite phases.
//
// Why: one pass of the fupd loop (DeepQNAg.hpp:343-466) alternates between quarters that only compute (forward
// passes, input gradients: matrix pipe busy, HBM idle) and quarters that stream the optimiser state (the backward
// passes with the fused Adam / EWC / soft-update epilogue: HBM busy, matrix pipe mostly idle) -- ddpg_update_body.h.
// With one workgroup per (update, agent) the two kinds of quarter of the two workgroups sharing a CU overlap only
// by chance (measured: the times of the compute-only and the memory-only builds ADD).
// Here a 512-thread workgroup holds TWO agents, one per 256-thread team, and walks them through all nupd updates
// of the launch with team 1 one quarter behind team 0; a workgroup barrier after every quarter keeps them there.
// So while one team streams its state the other one computes, on every CU, for the whole launch: HBM and the
// matrix pipe are both busy by construction.  Inside a quarter the teams synchronise on their own (ddrl_sync with
// DDRL_TEAMS: a counter in LDS), never on the hardware barrier.
// An agent's consecutive updates stay on one CU, so there is no hand-off between workgroups (no sc1 stores, no
// ordering waits, no reliance on dispatch order): the state an update reads was written by the same CU.
// The price is one idle quarter per team and launch (1 / (4 nupd + 1)), which is why launches of fewer than
// DDRL_PAIR_MIN_UPD updates keep the one-agent kernel (kernels_ddpg.hip).
#define DDRL_TEAMS 1
#define DDRL_STATE_IN_LDS      // the resident kernel below keeps an agent's state in LDS: see opaque_base (mlp_device.h)
#include "ddpg_update_body.h"

template <class NA, class NQ, bool EWC_A, bool EWC_Q>
__global__ void __launch_bounds__(2 * DDRL_NTHREADS, 1)
ddpg_update_pair_kernel(const DdpgParams P, int team_floats) {
  extern __shared__ __align__(16) float lds[];
  const int team = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
  const int tid = (int)threadIdx.x & (DDRL_NTHREADS - 1);
  const int agent = 2 * (int)blockIdx.x + team;
  if (threadIdx.x == 0) { ddrl_team_cnt[0] = 0; ddrl_team_cnt[1] = 0; ddrl_team_err = 0; }
  bool live = agent < P.G;
  const int ag = live ? agent : P.G - 1;
  const uint8_t on = P.mask ? P.mask[ag] : (uint8_t)1;
  const int len = P.rep_len[ag], head = P.rep_head[ag];
  const float gamma = P.gamma[ag], tau = P.tau[ag], omtau = P.omtau[ag];
  live = live && on && len >= 250 && len >= P.B;              // DeepQNAg.hpp:338
  float *tl = lds + (size_t)team * team_floats;
  const int nq = 4 * P.nupd;
  // the s' rows of the NEXT update are requested at the end of an update's last quarter (and those of the first update
  // here): they are in registers when its first quarter starts (ddpg_prefetch_rows)
  GatherRegs pre;
  bool have_pre = live && ddpg_prefetch_rows<NA, NQ>(P, ag, 0, head, tid, pre);
  __syncthreads();
#pragma unroll 1
  for (int t = 0; t <= nq; ++t) {
#if defined(DDRL_EXP_INPHASE)      // measurement: both teams of a workgroup in the same quarter; odd workgroups one quarter behind even ones
    const int tt = t - ((int)blockIdx.x & 1);
#elif defined(DDRL_EXP_INPHASE0)   // measurement: every team of the chip in the same quarter
    const int tt = t;
#else
    const int tt = t - team;                                  // team 1 runs one quarter behind team 0
#endif
    if (live && tt >= 0 && tt < nq) {
      const int u = tt >> 2;
      // every quarter derives its addresses afresh from values the optimiser cannot see through: hoisted out of
      // this loop they would all stay live across it (the four bodies together: hundreds of registers, spilled)
      int a_it = agent, tid_it = tid, z = 0;
      asm volatile("" : "+s"(a_it), "+s"(z));
      asm volatile("" : "+v"(tid_it));
      float *tl_it = tl + z;
      switch (tt & 3) {
        case 0: ddpg_q0<NA, NQ, EWC_A, EWC_Q>(P, tl_it, a_it, u, len, head, gamma, tid_it, pre, have_pre); break;
        case 1: ddpg_q1<NA, NQ, EWC_A, EWC_Q>(P, tl_it, a_it, u, tau, omtau, tid_it); break;
        case 2: ddpg_q2<NA, NQ, EWC_A, EWC_Q>(P, tl_it, a_it, u, tid_it); break;
        default:
          ddpg_q3<NA, NQ, EWC_A, EWC_Q>(P, tl_it, a_it, u, tau, omtau, tid_it, false);
          have_pre = ddpg_prefetch_rows<NA, NQ>(P, a_it, u + 1, head, tid_it, pre);
          break;
      }
    }
    __syncthreads();                                          // both teams: the quarter boundary
  }
  if (threadIdx.x == 0 && ddrl_team_err) __hip_atomic_store(P.upd_error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

extern "C" size_t ddpg_update_lds_bytes(const DdpgParams *P);
// ---------------------------------------------------------------------------------------------------------------
// Small nets: the agent's WHOLE state in LDS for the launch.  north_star: "weights and optimizer state are staged in
// LDS where they fit".  The BASELINE "small" nets (a = 20, q = 50 x 7) hold 3 584 padded floats per array, 7 arrays
// with EWC live (w, m, v, target, Fisher, F-hat, w*) = 98 KiB beside 45 KiB of activations: one workgroup per CU
// copies its agent's arrays in once, runs the launch's nupd updates on the LDS copy -- the passes read their
// weights from LDS, the fused optimiser epilogue reads and writes LDS -- and writes the five arrays that change
// back once.  HBM traffic per agent drops from 52 B/param per UPDATE to 48 B/param per LAUNCH (+ the minibatch
// rows), so the bound is no longer HBM but the matrix pipe / LDS.  Same device code as the other two kernels: the
// NetState pointers of a local copy of the parameters are redirected to the LDS copy (generic pointers into the LDS
// aperture; the optimiser addresses them as flat memory).  Results are bit-identical to nupd separate launches.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void block_copy(float *dst, const float *src, int floats, int tid) {   // floats % 4 == 0, 16-B aligned
  for (int i = 4 * tid; i < floats; i += 4 * DDRL_NTHREADS) *(f32x4 *)(dst + i) = *(const f32x4 *)(src + i);
}

template <class NA, class NQ, bool EWC_A, bool EWC_Q>
__global__ void __launch_bounds__(DDRL_NTHREADS, 1)
ddpg_update_resident_kernel(const DdpgParams P, int act_floats) {
  extern __shared__ __align__(16) float lds[];
  const int tid = (int)threadIdx.x;
  const int agent = (int)blockIdx.x;
  if (tid == 0) { ddrl_team_cnt[0] = 0; ddrl_team_cnt[1] = 0; ddrl_team_err = 0; }
  const uint8_t on = P.mask ? P.mask[agent] : (uint8_t)1;
  const int len = P.rep_len[agent], head = P.rep_head[agent];
  const float gamma = P.gamma[agent], tau = P.tau[agent], omtau = P.omtau[agent];
  if (!on || len < 250 || len < P.B) return;                  // DeepQNAg.hpp:338 (uniform: the whole workgroup leaves)
  constexpr int SA = NA::stride(), SQ = NQ::stride();
  float *st = lds + act_floats;
  DdpgParams Q = P;
  Q.resident = 1;
  const size_t oa = (size_t)agent * SA, oq = (size_t)agent * SQ;
  float *p = st;
  auto stage = [&](float *&field, float *global, size_t off, int floats) {   // one array: copy in, redirect
    block_copy(p, global + off, floats, tid);
    field = p;
    p += floats;
  };
  stage(Q.sa.w, P.sa.w, oa, SA); stage(Q.sa.m, P.sa.m, oa, SA); stage(Q.sa.v, P.sa.v, oa, SA); stage(Q.sa.wt, P.sa.wt, oa, SA);
  stage(Q.sq.w, P.sq.w, oq, SQ); stage(Q.sq.m, P.sq.m, oq, SQ); stage(Q.sq.v, P.sq.v, oq, SQ); stage(Q.sq.wt, P.sq.wt, oq, SQ);
  if (EWC_A) { stage(Q.sa.fisher, P.sa.fisher, oa, SA); stage(Q.sa.fhat, P.sa.fhat, oa, SA); stage(Q.sa.wstar, P.sa.wstar, oa, SA); }
  if (EWC_Q) { stage(Q.sq.fisher, P.sq.fisher, oq, SQ); stage(Q.sq.fhat, P.sq.fhat, oq, SQ); stage(Q.sq.wstar, P.sq.wstar, oq, SQ); }
  __syncthreads();
#pragma unroll 1
  for (int u = 0; u < P.nupd; ++u) {
    int a_it = agent, tid_it = tid, z = 0;                    // see ddpg_update_pair_kernel: nothing is hoisted out of this loop
    asm volatile("" : "+s"(a_it), "+s"(z));
    asm volatile("" : "+v"(tid_it));
    float *tl = lds + z;
    GatherRegs pre;
    ddpg_q0<NA, NQ, EWC_A, EWC_Q>(Q, tl, a_it, u, len, head, gamma, tid_it, pre, false);
    ddpg_q1<NA, NQ, EWC_A, EWC_Q>(Q, tl, a_it, u, tau, omtau, tid_it);
    ddpg_q2<NA, NQ, EWC_A, EWC_Q>(Q, tl, a_it, u, tid_it);
    ddpg_q3<NA, NQ, EWC_A, EWC_Q>(Q, tl, a_it, u, tau, omtau, tid_it, false);
    ddrl_sync();
  }
  // what an update changes goes back (F-hat and w* are read-only here)
  block_copy(P.sa.w + oa, Q.sa.w, SA, tid); block_copy(P.sa.m + oa, Q.sa.m, SA, tid); block_copy(P.sa.v + oa, Q.sa.v, SA, tid);
  block_copy(P.sa.wt + oa, Q.sa.wt, SA, tid);
  block_copy(P.sq.w + oq, Q.sq.w, SQ, tid); block_copy(P.sq.m + oq, Q.sq.m, SQ, tid); block_copy(P.sq.v + oq, Q.sq.v, SQ, tid);
  block_copy(P.sq.wt + oq, Q.sq.wt, SQ, tid);
  if (EWC_A) block_copy(P.sa.fisher + oa, Q.sa.fisher, SA, tid);
  if (EWC_Q) block_copy(P.sq.fisher + oq, Q.sq.fisher, SQ, tid);
  if (tid == 0 && ddrl_team_err) __hip_atomic_store(P.upd_error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


template <class NA, class NQ, bool EA, bool EQ>
static hipError_t launch_t(const DdpgParams *P, size_t team_bytes, hipStream_t st) {
  static LdsConfig configured;
  const size_t lds = 2 * team_bytes;
  hipError_t e = ddrl_set_lds(ddpg_update_pair_kernel<NA, NQ, EA, EQ>, lds, configured);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((ddpg_update_pair_kernel<NA, NQ, EA, EQ>), dim3((P->G + 1) / 2), dim3(2 * DDRL_NTHREADS), lds, st, *P,
                     (int)(team_bytes / sizeof(float)));
  return hipGetLastError();
}

template <class NA, class NQ>
static hipError_t launch_ewc(const DdpgParams *P, size_t team_bytes, hipStream_t st) {
  const bool ea = P->sa.fisher != nullptr, eq = P->sq.fisher != nullptr;
  if (ea && eq) return launch_t<NA, NQ, true, true>(P, team_bytes, st);
  if (ea) return launch_t<NA, NQ, true, false>(P, team_bytes, st);
  if (eq) return hipErrorInvalidValue;          // EWC on the critic alone does not exist (DODevMLP: ewc_for_critic needs ewc)
  return launch_t<NA, NQ, false, false>(P, team_bytes, st);
}

// can this population's update run as pairs?  (compile-time shapes only: two teams of the run-time-shape kernel
// would not fit the register file at two waves per SIMD any better than the one-agent kernel does)
extern "C" int ddpg_update_pair_ok(const DdpgParams *P) {
  const size_t team_bytes = (ddpg_update_lds_bytes(P) + 255) / 256 * 256;
  if (2 * team_bytes + 64 > 160 * 1024) return 0;
  // mid nets only: as pairs the small nets measured 0.86 ms against 0.58 ms per population update (below), and their instantiation
  // sat at the edge of its register budget (one extra live value turned scalar-register spills into scratch memory)
  return MidActor::matches(P->a) && MidCritic::matches(P->q);
}

// ... and does it pay?  Measured at 1024 agents, 8 updates per launch: mid nets 3.91 ms as pairs against 3.97 ms with the
// hand-off kernel; small nets 0.86 ms against 0.58 ms (their update is a chain of short dependent passes, and the team
// barrier's LDS polling costs more than the hand-off it saves).
// Below ~512 agents pairs leave compute units idle (128 agents: 0.88 M updates/s one agent per workgroup against 0.58 M as
// pairs; 256: 1.56 M against 1.12 M; 512: equal).
extern "C" int ddpg_update_pair_pays(const DdpgParams *P) { return P->G > 512 && MidActor::matches(P->a) && MidCritic::matches(P->q); }

extern "C" hipError_t ddpg_update_pair_launch(const DdpgParams *P, hipStream_t st) {
  const size_t team_bytes = (ddpg_update_lds_bytes(P) + 255) / 256 * 256;
  if (MidActor::matches(P->a) && MidCritic::matches(P->q)) return launch_ewc<MidActor, MidCritic>(P, team_bytes, st);
  return hipErrorInvalidValue;
}

// ---- the resident kernel's host side ----
static size_t resident_state_floats(const DdpgParams *P) {
  const size_t per = (size_t)P->a.stride_ + P->q.stride_;
  return 4 * per + (P->sa.fisher ? 3 * (size_t)P->a.stride_ : 0) + (P->sq.fisher ? 3 * (size_t)P->q.stride_ : 0);
}
extern "C" int ddpg_update_resident_ok(const DdpgParams *P) {
  if (!(SmallActor::matches(P->a) && SmallCritic::matches(P->q))) return 0;
  if (P->sq.fisher && !P->sa.fisher) return 0;
  const size_t act = (ddpg_update_lds_bytes(P) + 255) / 256 * 256;
  return act + resident_state_floats(P) * sizeof(float) + 64 <= 160 * 1024;
}
template <bool EA, bool EQ>
static hipError_t launch_resident_t(const DdpgParams *P, hipStream_t st) {
  static LdsConfig configured;
  const size_t act = (ddpg_update_lds_bytes(P) + 255) / 256 * 256;
  const size_t lds = act + resident_state_floats(P) * sizeof(float);
  hipError_t e = ddrl_set_lds(ddpg_update_resident_kernel<SmallActor, SmallCritic, EA, EQ>, lds, configured);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((ddpg_update_resident_kernel<SmallActor, SmallCritic, EA, EQ>), dim3(P->G), dim3(DDRL_NTHREADS), lds, st, *P,
                     (int)(act / sizeof(float)));
  return hipGetLastError();
}
extern "C" hipError_t ddpg_update_resident_launch(const DdpgParams *P, hipStream_t st) {
  const bool ea = P->sa.fisher != nullptr, eq = P->sq.fisher != nullptr;
  if (ea && eq) return launch_resident_t<true, true>(P, st);
  if (ea) return launch_resident_t<true, false>(P, st);
  return launch_resident_t<false, false>(P, st);
}

