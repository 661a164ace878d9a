// kernels_ddpg.hip -- the fused DDPG population update for gfx950.
//
// One launch = one pass of the `fupd` loop body of DeepQNAg::end_episode
// (/root/reference/agent/deepqn/include/DeepQNAg.hpp:343-466) for every agent of the population:
// one 256-thread workgroup per agent runs all 5 forward passes, both critic backward passes, the
// actor backward pass, both Adam steps (+EWC terms) and both soft target updates without leaving
// the CU.  The minibatch and all activations live in LDS; weights / optimiser state are touched in
// HBM exactly once per update by the fused epilogue (opt_step), plus L2-served re-reads of the
// weights by the products.
#include "mlp_device.h"

struct LdsMap {
  float *h[2][DDRL_MAXL];   // h[net][l]: hidden activation l (1..L-1) of the pass in flight
  float *xc, *xa;           // critic input [s|a] tile, actor input s tile
  float *topq, *topa;       // output tiles: q [64 x 4], actions [64 x ld(A)]
  float *yv, *rv, *dn;      // TD target, reward, goal flag per row
};

// forward through a whole net; hidden activation l goes to hb[l]
__device__ __forceinline__ void fwd_net(const NetDesc &nd, const float *__restrict__ w,
                                        const float *X, int ldx, float *const *hb, float *top, int ldtop,
                                        int wave, int lane) {
  const float *in = X;
  int ldi = ldx;
  for (int l = 0; l < nd.L; ++l) {
    const bool last = l == nd.L - 1;
    float *out = last ? top : hb[l + 1];
    const int ldo = last ? ldtop : nd.ld[l + 1];
    fwd_layer(in, ldi, nd.kp[l], w + nd.offw[l], w + nd.offb[l], nd.d[l + 1],
              last ? nd.out_act : nd.hidden_act, out, ldo, wave, lane);
    __syncthreads();
    in = out;
    ldi = ldo;
  }
}

// Backward through a whole net from dZ_L held in `top` (diff w.r.t. the last InnerProduct's
// output).  train: parameter gradients go through the fused optimiser epilogue.  dx0: when
// non-null, the input gradient columns [c0, c0 + nc) are written to dx0[row * lddx0 + (col - c0)].
template <class DX0>
__device__ __forceinline__ void bwd_net(const NetDesc &nd, const float *__restrict__ w, const OptCtx *opt,
                                        const float *X, int ldx, float *const *hb, float *top, int ldtop,
                                        int c0, int nc, DX0 dx0_store, int wave, int lane, int tid) {
  const int g = lane >> 4, c = lane & 15;
  float *dZ = top;
  int ldz = ldtop;
  for (int l = nd.L - 1; l >= 0; --l) {
    const int K = nd.d[l], N = nd.d[l + 1], kp = nd.kp[l];
    const float *W = w + nd.offw[l];
    float *in = l == 0 ? const_cast<float *>(X) : hb[l];
    const int ldi = l == 0 ? ldx : nd.ld[l];
    const bool need_dx = l > 0 || nc > 0;
    f32x4 acc[2][4];
    int kt0 = 0, kt1 = (K + 15) >> 4;
    if (l == 0) { kt0 = c0 >> 4; kt1 = (c0 + nc + 15) >> 4; }
    if (need_dx) bwd_dx<2>(dZ, ldz, N, W, kp, kt0, kt1, acc, wave, lane);
    __syncthreads();   // every wave has read W_l before any wave's epilogue rewrites it
    if (opt) bwd_dw_opt(dZ, ldz, N, in, ldi, K, kp, nd.offw[l], nd.offb[l], *opt, wave, lane, tid);
    __syncthreads();   // every wave has read H_l before it is overwritten by dZ_l
    if (need_dx) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int kt = kt0 + wave + DDRL_NWAVES * i;
        if (kt < kt1) {
          const int col = kt * 16 + c;
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = mt * 16 + 4 * g + r;
              if (l > 0) {
                if (col < K) in[row * ldi + col] = act_bwd(nd.hidden_act, in[row * ldi + col], acc[i][mt][r]);
              } else if (col >= c0 && col < c0 + nc) {
                dx0_store(row, col - c0, acc[i][mt][r]);
              }
            }
        }
      }
    }
    __syncthreads();
    dZ = in;
    ldz = ldi;
  }
}

// gather one 64-row tile of the minibatch from the agent's replay ring (DeepQNAg.hpp:353-377,
// nstep_return == 0).  next=true: s' into xa and xc[:, :S];  next=false: s into xa, [s|a] into xc,
// reward / goal into rv / dn.
__device__ __forceinline__ void gather_tile(const DdpgParams &P, int agent, const int *__restrict__ idx,
                                            int len, int head, bool next, const LdsMap &M, int tid) {
  const int S = P.S, A = P.A, B = P.B;
  const float *rep = P.rep + (size_t)agent * P.cap * P.rowlen;
  (void)len;
  for (int e = tid; e < DDRL_MB * P.ldxc; e += DDRL_NTHREADS) {
    const int row = e / P.ldxc, col = e - row * P.ldxc;
    float v = 0.f;
    if (row < B) {
      int phys = head + idx[row];
      if (phys >= P.cap) phys -= P.cap;
      const float *rr = rep + (size_t)phys * P.rowlen;
      if (next) { if (col < S) v = rr[S + A + col]; }
      else if (col < S + A) v = rr[col];
    }
    M.xc[e] = v;
    if (col < P.ldxa) M.xa[row * P.ldxa + col] = col < S ? v : 0.f;
  }
  if (!next)
    for (int row = tid; row < DDRL_MB; row += DDRL_NTHREADS) {
      float r = 0.f, d = 0.f;
      if (row < B) {
        int phys = head + idx[row];
        if (phys >= P.cap) phys -= P.cap;
        const float *rr = rep + (size_t)phys * P.rowlen;
        r = rr[2 * S + A];
        d = rr[2 * S + A + 1];
      }
      M.rv[row] = r;
      M.dn[row] = d;
    }
}

__device__ __forceinline__ OptCtx make_opt(const DdpgParams &P, const NetState &s, const NetDesc &nd,
                                           int agent, float tau, float omtau, float nsample) {
  OptCtx o;
  const size_t off = (size_t)agent * nd.stride;
  o.w = s.w + off; o.m = s.m + off; o.v = s.v + off;
  o.wt = s.wt ? s.wt + off : nullptr;
  o.fisher = s.fisher ? s.fisher + off : nullptr;
  o.fhat = s.fhat ? s.fhat + off : nullptr;
  o.wstar = s.wstar ? s.wstar + off : nullptr;
  // caffe AdamSolver: t = iter + 1, correction = sqrt(1 - beta2^t) / (1 - beta1^t); in fp64 because
  // 1 - 0.999^t loses 4 digits in fp32 at small t
  const int t = s.iter[agent] + 1;
  const double corr = sqrt(1.0 - pow((double)P.beta2, (double)t)) / (1.0 - pow((double)P.beta1, (double)t));
  o.lr_t = (float)((double)s.lr[agent] * corr);
  o.tau = tau; o.omtau = omtau;
  o.decay = s.decay ? s.decay[agent] : -1.f;
  o.ewc_coef = (s.ewc_coef && s.fhat) ? s.ewc_coef[agent] : 0.f;
  o.fisher_beta = P.fisher_beta;
  o.nsample = nsample;
  o.beta1 = P.beta1; o.beta2 = P.beta2; o.omb1 = P.omb1; o.omb2 = P.omb2; o.eps = P.eps;
  return o;
}

extern "C" __global__ void __launch_bounds__(DDRL_NTHREADS)
ddpg_update_kernel(const DdpgParams P) {
  extern __shared__ __align__(16) float lds[];
  const int agent = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (P.mask && !P.mask[agent]) return;
  const int len = P.rep_len[agent], head = P.rep_head[agent];
  if (len < 250 || len < P.B) return;                      // DeepQNAg.hpp:338
  const int S = P.S, A = P.A, B = P.B;
  const NetDesc &na = P.a, &nq = P.q;

  // ---- carve LDS ----
  LdsMap M;
  float *p = lds;
  for (int l = 1; l < DDRL_MAXL; ++l) {
    const int wa = l < na.L ? na.ld[l] : 0, wq = l < nq.L ? nq.ld[l] : 0;
    const int wmax = wa > wq ? wa : wq;
    M.h[0][l] = M.h[1][l] = p;        // the actor and critic passes never overlap in time
    p += DDRL_MB * wmax;
  }
  M.h[0][0] = M.h[1][0] = nullptr;
  M.xc = p; p += DDRL_MB * P.ldxc;
  M.xa = p; p += DDRL_MB * P.ldxa;
  const int ldq = nq.ld[nq.L], lda = na.ld[na.L];
  M.topq = p; p += DDRL_MB * ldq;
  M.topa = p; p += DDRL_MB * lda;
  M.yv = p; p += DDRL_MB;
  M.rv = p; p += DDRL_MB;
  M.dn = p; p += DDRL_MB;

  const int *idx = P.idx + (size_t)agent * B;
  const float gamma = P.gamma[agent], tau = P.tau[agent], omtau = P.omtau[agent];
  const float *w_a = P.sa.w + (size_t)agent * na.stride, *w_at = P.sa.wt + (size_t)agent * na.stride;
  const float *w_q = P.sq.w + (size_t)agent * nq.stride, *w_qt = P.sq.wt + (size_t)agent * nq.stride;

  // ---- a' = ann_target(s'), q' = qnn_target(s', a')  (DeepQNAg.hpp:379-383) ----
  gather_tile(P, agent, idx, len, head, true, M, tid);
  __syncthreads();
  fwd_net(na, w_at, M.xa, P.ldxa, M.h[0], M.topa, lda, wave, lane);
  for (int e = tid; e < DDRL_MB * A; e += DDRL_NTHREADS) {
    const int row = e / A, col = e - row * A;
    M.xc[row * P.ldxc + S + col] = M.topa[row * lda + col];
  }
  __syncthreads();
  fwd_net(nq, w_qt, M.xc, P.ldxc, M.h[1], M.topq, ldq, wave, lane);

  // ---- TD targets (DeepQNAg.hpp:386-405, nstep_return == 0) ----
  gather_tile(P, agent, idx, len, head, false, M, tid);
  if (tid < DDRL_MB) {
    const float qn = M.topq[tid * ldq];
    M.yv[tid] = M.dn[tid] != 0.f ? M.rv[tid] : M.rv[tid] + gamma * qn;
  }
  __syncthreads();
  if (P.qt_out && tid < B) P.qt_out[(size_t)agent * B + tid] = M.yv[tid];

  // ---- critic update: qnn->learn_batch(s, a, y, 1)  (DeepQNAg.hpp:408; MLP.hpp:468-485) ----
  fwd_net(nq, w_q, M.xc, P.ldxc, M.h[1], M.topq, ldq, wave, lane);
  if (tid < DDRL_MB) {
    // EuclideanLoss: loss = sum (q - y)^2 / (2B), dq = (q - y) / B
    const float df = tid < B ? M.topq[tid * ldq] - M.yv[tid] : 0.f;
    float sq = df * df;
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if (tid == 0) P.loss_out[agent] = sq / (float)B * 0.5f;
    M.topq[tid * ldq] = df / (float)B;
  }
  __syncthreads();
  {
    OptCtx oq = make_opt(P, P.sq, nq, agent, tau, omtau, (float)B);
    auto none = [](int, int, float) {};
    bwd_net(nq, w_q, &oq, M.xc, P.ldxc, M.h[1], M.topq, ldq, 0, 0, none, wave, lane, tid);
  }

  // ---- actor update (DeepQNAg.hpp:411-459) ----
  fwd_net(na, w_a, M.xa, P.ldxa, M.h[0], M.topa, lda, wave, lane);       // pi = ann(s)
  for (int e = tid; e < DDRL_MB * A; e += DDRL_NTHREADS) {
    const int row = e / A, col = e - row * A;
    M.xc[row * P.ldxc + S + col] = M.topa[row * lda + col];
  }
  __syncthreads();
  // the actor's hidden activations must survive the critic pass: the critic uses its own copies
  // of the buffers only if they are distinct; they are shared, so the actor forward is re-run
  // after the critic backward (cheaper than doubling LDS: one 2 P_a product set).
  fwd_net(nq, w_q, M.xc, P.ldxc, M.h[1], M.topq, ldq, wave, lane);       // Q(s, pi(s)) with the new critic
  if (tid < DDRL_MB) M.topq[tid * ldq] = tid < B ? -1.0f : 0.f;            // :424-428, NOT /B
  __syncthreads();
  {
    // critic_backward (:429): only d/d(actions) is needed; critic parameter diffs are discarded
    float *topa = M.topa;
    const int inverting = P.inverting, out_act = na.out_act;
    float *dact_out = P.dact_out ? P.dact_out + (size_t)agent * B * A : nullptr;
    auto store = [=](int row, int h, float diff) {
      const float out = topa[row * lda + h];
      if (inverting) {                                                   // :432-450
        if (diff < 0.f) diff *= (1.0f - out) / 2.0f;
        else if (diff > 0.f) diff *= (out + 1.0f) / 2.0f;
      }
      if (dact_out && row < B) dact_out[row * A + h] = diff;
      topa[row * lda + h] = act_bwd(out_act, out, diff);                 // actor's output layer, in place
    };
    bwd_net(nq, w_q, nullptr, M.xc, P.ldxc, M.h[1], M.topq, ldq, S, A, store, wave, lane, tid);
  }
  // hidden activations of the actor were overwritten by the critic pass: recompute them without
  // touching topa (which now holds dZ_L of the actor)
  {
    const float *in = M.xa;
    int ldi = P.ldxa;
    for (int l = 0; l < na.L - 1; ++l) {
      fwd_layer(in, ldi, na.kp[l], w_a + na.offw[l], w_a + na.offb[l], na.d[l + 1], na.hidden_act,
                M.h[0][l + 1], na.ld[l + 1], wave, lane);
      __syncthreads();
      in = M.h[0][l + 1];
      ldi = na.ld[l + 1];
    }
  }
  {
    OptCtx oa = make_opt(P, P.sa, na, agent, tau, omtau, (float)B);      // updateFisher(traj.size()) :456
    auto none = [](int, int, float) {};
    bwd_net(na, w_a, &oa, M.xa, P.ldxa, M.h[0], M.topa, lda, 0, 0, none, wave, lane, tid);
  }
  if (tid == 0) {
    P.sq.iter[agent] += 1;                                               // Solver::Step ++iter
    P.sa.iter[agent] += 1;                                               // set_iter(iter()+1) :459
    if (P.sq.fisher) P.sq.fisher_nbr[agent] = P.sq.fisher_nbr[agent] * P.fisher_beta + (float)B;
    if (P.sa.fisher) P.sa.fisher_nbr[agent] = P.sa.fisher_nbr[agent] * P.fisher_beta + (float)B;
  }
}

// host-visible LDS sizing, same carve order as the kernel
extern "C" size_t ddpg_update_lds_bytes(const DdpgParams *P) {
  size_t f = 0;
  for (int l = 1; l < DDRL_MAXL; ++l) {
    const int wa = l < P->a.L ? P->a.ld[l] : 0, wq = l < P->q.L ? P->q.ld[l] : 0;
    f += (size_t)DDRL_MB * (wa > wq ? wa : wq);
  }
  f += (size_t)DDRL_MB * (P->ldxc + P->ldxa + P->q.ld[P->q.L] + P->a.ld[P->a.L] + 3);
  return f * sizeof(float);
}

extern "C" hipError_t ddpg_update_launch(const DdpgParams *P, hipStream_t st) {
  const size_t lds = ddpg_update_lds_bytes(P);
  static size_t configured = 0;
  if (lds > 48 * 1024 && lds > configured) {
    hipError_t e = hipFuncSetAttribute((const void *)ddpg_update_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    configured = lds;
  }
  hipLaunchKernelGGL(ddpg_update_kernel, dim3(P->G), dim3(DDRL_NTHREADS), lds, st, *P);
  return hipGetLastError();
}
