// kernels_ddpg_big.hip -- the DDPG update for minibatches of more than 64 rows (the reference's own
// ddpg.ini / ddpg-cp.ini use mini_batch_size = 512).  Same arithmetic as the fused kernel
// (kernels_ddpg.hip, DeepQNAg.hpp:343-466), cut where a full-batch gradient has to be complete:
//   critic chunks:  targets, critic forward / backward on 64 rows each -> partial gradients
//   critic apply:   sum of the partials in chunk order, Fisher / EWC / L2 / Adam / soft target update
//   actor chunks:   pi(s), Q(s, pi(s)) with the NEW critic, dQ/da, inverting gradients, actor backward
//   actor apply
// One workgroup per (chunk, agent): a single agent with B = 512 still spreads over 8 CUs.
#include "ddpg_device.h"

// PHASE 0: critic chunk.  PHASE 1: actor chunk.
template <int PHASE>
__global__ void __launch_bounds__(DDRL_NTHREADS, DDRL_NTHREADS / 128)   // two workgroups per CU (<= 256 registers)
ddpg_chunk_kernel(const DdpgParams P) {
  extern __shared__ __align__(16) float lds[];
  const int chunk = blockIdx.x, agent = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (P.mask && !P.mask[agent]) return;
  const int len = P.rep_len[agent], head = P.rep_head[agent];
  if (len < 250 || len < P.B) return;                      // DeepQNAg.hpp:338
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
  auto actions_to_critic = [&]() {     // actor output (gated) into the critic's action columns (gated again)
    for (int e = tid; e < DDRL_MB * A; e += DDRL_NTHREADS) {
      const int row = e / A, col = e - row * A;
      float a = M.topa[row * lda + col];
      if (agate) {
        a *= agate[col];
        M.topa[row * lda + col] = a;
        a *= agate[col];
      }
      M.xc[row * ldxc + S + col] = a;
    }
    __syncthreads();
  };
  auto none = [](int, int, float) {};
  OptCtx unused = {};

  if (PHASE == 0) {
    if (P.nstep > 0) nstep_horizon(P, S, A, agent, idx, nrows, len, head, gamma, M, tid);
    gather_tile(P, S, A, ldxc, ldxa, agent, idx, nrows, head, true, M, tid);
    __syncthreads();
    fwd_net(na, na.L(), w_at, M.xa, ldxa, M.hbase, M.hstride, M.topa, lda, wave, lane);    // a' = ann_target(s')
    actions_to_critic();
    fwd_net(nq, nq.L(), w_qt, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, wave, lane);    // q' = qnn_target(s', a')
    gather_tile(P, S, A, ldxc, ldxa, agent, idx, nrows, head, false, M, tid);
    int trow;
    if (gather_owns_row(tid, trow)) {
      const float qn = M.topq[trow * ldq];
      M.yv[trow] = M.dn[trow] != 0.f ? M.rv[trow] : (P.nstep > 0 ? M.nrew[trow] : M.rv[trow]) + gamma * qn;
    }
    __syncthreads();
    if (P.qt_out && tid < nrows) P.qt_out[(size_t)agent * B + row0 + tid] = M.yv[tid];
    fwd_net(nq, nq.L(), w_q, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, wave, lane);
    if (tid < DDRL_MB) {
      const float df = tid < nrows ? M.topq[tid * ldq] - M.yv[tid] : 0.f;
      float sq = df * df;
      for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
      if (tid == 0) P.loss_part[(size_t)agent * P.maxchunks + chunk] = sq;
      M.topq[tid * ldq] = df / (float)B;                     // EuclideanLoss over the WHOLE minibatch
    }
    __syncthreads();
    float *gout = P.gpart_q + ((size_t)agent * P.maxchunks + chunk) * nq.stride();
    bwd_net<DW_STORE, false>(nq, w_q, unused, gout, M.xc, ldxc, M.hbase, M.hstride, M.topq, ldq, 0, 0, none, wave, lane, tid);
  } else {
    gather_tile(P, S, A, ldxc, ldxa, agent, idx, nrows, head, false, M, tid);
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
      auto store = [=](int row, int h, float diff) {
        float out = topa[row * lda + h];
        const float gt = agate ? agate[h] : 1.f;
        diff *= gt;
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
  const NetState &st = NET == 0 ? P.sa : P.sq;
  const int nch = (P.B + DDRL_MB - 1) / DDRL_MB;
  const OptCtx o = make_opt<EWC>(P, st, nd.stride(), agent, P.tau[agent], P.omtau[agent], (float)P.B);
  const int i = (blockIdx.x * 256 + tid) * 4;
  if (i < nd.stride()) {
    OptRegs r;
    opt_load4<EWC, true>(o, i, r);
    const float *gp = (NET == 0 ? P.gpart_a : P.gpart_q) + (size_t)agent * P.maxchunks * nd.stride() + i;
    f32x4 g = *(const f32x4 *)gp;
    for (int c = 1; c < nch; ++c) g += *(const f32x4 *)(gp + (size_t)c * nd.stride());
    opt_step4<EWC, true>(o, i, g, r, true);
  }
  if (blockIdx.x == 0 && tid == 0) {
    st.iter[agent] += 1;
    if (EWC) st.fisher_nbr[agent] = st.fisher_nbr[agent] * P.fisher_beta + (float)P.B;
    if (NET == 1) {
      float s = 0.f;
      for (int c = 0; c < nch; ++c) s += P.loss_part[(size_t)agent * P.maxchunks + c];
      P.loss_out[agent] = s / (float)P.B * 0.5f;
    }
  }
}

extern "C" size_t ddpg_update_lds_bytes(const DdpgParams *P);

template <int PHASE>
static hipError_t launch_chunks(const DdpgParams *P, hipStream_t st) {
  static LdsConfig configured;
  const size_t lds = ddpg_update_lds_bytes(P);
  hipError_t e = ddrl_set_lds(ddpg_chunk_kernel<PHASE>, lds, configured);
  if (e != hipSuccess) return e;
  const int nch = (P->B + DDRL_MB - 1) / DDRL_MB;
  hipLaunchKernelGGL(ddpg_chunk_kernel<PHASE>, dim3(nch, P->G), dim3(DDRL_NTHREADS), lds, st, *P);
  return hipGetLastError();
}

template <int NET>
static hipError_t launch_apply(const DdpgParams *P, hipStream_t st) {
  const NetDesc &nd = NET == 0 ? P->a : P->q;
  const bool ewc = (NET == 0 ? P->sa : P->sq).fisher != nullptr;
  const dim3 grid((nd.stride_ + 1023) / 1024, P->G);
  if (ewc) hipLaunchKernelGGL((ddpg_apply_kernel<NET, true>), grid, dim3(256), 0, st, *P);
  else hipLaunchKernelGGL((ddpg_apply_kernel<NET, false>), grid, dim3(256), 0, st, *P);
  return hipGetLastError();
}

extern "C" hipError_t ddpg_apply_launch(const DdpgParams *P, int net, hipStream_t st) {
  return net == 0 ? launch_apply<0>(P, st) : launch_apply<1>(P, st);
}

extern "C" hipError_t ddpg_update_big_launch(const DdpgParams *P, hipStream_t st) {
  hipError_t e;
  if ((e = launch_chunks<0>(P, st)) != hipSuccess) return e;
  if ((e = launch_apply<1>(P, st)) != hipSuccess) return e;
  if ((e = launch_chunks<1>(P, st)) != hipSuccess) return e;
  return launch_apply<0>(P, st);
}
