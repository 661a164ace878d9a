// kernels_ddpg_w8.hip -- the in-workgroup multi-update DDPG kernel (kernels_ddpg.hip, ddpg_update_loop_kernel) with EIGHT waves
// per agent, for populations that do not fill the GPU with 256-thread workgroups.
//
// A 256-thread workgroup runs one wave per SIMD; with fewer agents than CUs x 2 -- the metric's per-GPU share of a node is 128 --
// nothing else is resident on the CU to issue while that wave waits for an LDS or memory round trip, and such waits are a
// third of a lone workgroup's update (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = 0.34 at 128 agents).  Here the
// agent's workgroup has 512 threads: every 16-row tile of the minibatch belongs to two waves that share its column tiles
// (mlp_device.h, DDRL_PARTS), the optimiser's tile slots are dealt to eight waves, the gathers to eight threads per row.  The
// arithmetic is that of the four-wave kernels element for element (each output element's k-sum is one wave's, in the same
// order), so the results are bit-identical to theirs.
// This translation unit is compiled with its own DDRL_NWAVES; nothing with external linkage in the shared headers depends on it.
#undef DDRL_NWAVES
#define DDRL_NWAVES 8
#include "ddpg_update_body.h"

template <class NA, class NQ, bool EWC_A, bool EWC_Q>
__global__ void __launch_bounds__(DDRL_NTHREADS, 1)
ddpg_update_loop8_kernel(const DdpgParams P) {
  extern __shared__ __align__(16) float lds[];
  const int tid = (int)threadIdx.x;
  const int agent = (int)blockIdx.x;
  const uint8_t live = P.mask ? P.mask[agent] : (uint8_t)1;
  const int len = P.rep_len[agent], head = P.rep_head[agent];
  const float gamma = P.gamma[agent], tau = P.tau[agent], omtau = P.omtau[agent];
  if (!live || len < 250 || len < P.B) return;             // DeepQNAg.hpp:338 (uniform)
  GatherRegs pre;
  bool have_pre = ddpg_prefetch_rows<NA, NQ>(P, agent, 0, head, tid, pre);
#pragma unroll 1
  for (int u = 0; u < P.nupd; ++u) {
    int a_it = agent, tid_it = tid, z = 0;                    // nothing is hoisted out of this loop (see kernels_ddpg.hip)
    asm volatile("" : "+s"(a_it), "+s"(z));
    asm volatile("" : "+v"(tid_it));
    float *tl = lds + z;
    ddpg_q0<NA, NQ, EWC_A, EWC_Q>(P, tl, a_it, u, len, head, gamma, tid_it, pre, have_pre);
    ddpg_q1<NA, NQ, EWC_A, EWC_Q>(P, tl, a_it, u, tau, omtau, tid_it);
    ddpg_q2<NA, NQ, EWC_A, EWC_Q>(P, tl, a_it, u, tid_it);
    ddpg_q3<NA, NQ, EWC_A, EWC_Q>(P, tl, a_it, u, tau, omtau, tid_it, false);
    have_pre = ddpg_prefetch_rows<NA, NQ>(P, a_it, u + 1, head, tid_it, pre);
  }
}

// LDS sizing, same carve order as the kernel (ddpg_carve with this unit's DDRL_NWAVES / DDRL_PARK_TILES)
static size_t w8_lds_bytes(const DdpgParams *P) {
  size_t f = 0;
  { const int nh = (P->a.L_ > P->q.L_ ? P->a.L_ : P->q.L_) - 1; f += (size_t)nh * DDRL_MB * P->ldh; }
  f += (size_t)DDRL_MB * (P->ldxc + P->ldxa) + DDRL_NWAVES * 64;
  const size_t tail = (size_t)DDRL_MB * (P->q.ld_[P->q.L_] + P->a.ld_[P->a.L_] + 5);
  const size_t parkf = ParkOf<MidActor, MidCritic>::value ? (size_t)DDRL_NWAVES * DDRL_PARK_TILES * 256 : 0;
  f += tail > parkf ? tail : parkf;
  return f * sizeof(float);
}

template <class NA, class NQ, bool EA, bool EQ>
static hipError_t launch_loop8_t(const DdpgParams *P, size_t lds, hipStream_t st) {
  static LdsConfig configured;
  hipError_t e = ddrl_set_lds(ddpg_update_loop8_kernel<NA, NQ, EA, EQ>, lds, configured);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((ddpg_update_loop8_kernel<NA, NQ, EA, EQ>), dim3(P->G), dim3(DDRL_NTHREADS), lds, st, *P);
  return hipGetLastError();
}

// compile-time shapes only: the mid nets (caller sets P->pair: plain stores, no hand-off)
extern "C" int ddpg_update_loop8_ok(const DdpgParams *P) { return MidActor::matches(P->a) && MidCritic::matches(P->q); }
extern "C" hipError_t ddpg_update_loop8_launch(const DdpgParams *P, hipStream_t st) {
  if (!ddpg_update_loop8_ok(P)) return hipErrorInvalidValue;
  const size_t lds = w8_lds_bytes(P);
  const bool ea = P->sa.fisher != nullptr, eq = P->sq.fisher != nullptr;
  if (ea && eq) return launch_loop8_t<MidActor, MidCritic, true, true>(P, lds, st);
  if (ea) return launch_loop8_t<MidActor, MidCritic, true, false>(P, lds, st);
  if (eq) return hipErrorInvalidValue;
  return launch_loop8_t<MidActor, MidCritic, false, false>(P, lds, st);
}
