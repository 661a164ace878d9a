// kernels_nfac.hip -- NFAC(lambda)-V on gfx950: the arithmetic of OfflineCaclaAg::update_critic and
// ::end_episode (/root/reference/agent/cacla/include/OfflineCaclaAg.hpp:242-401, 423-594) for a
// population of agents.  The batch of every Adam step is the agent's WHOLE trajectory (T rows,
// MLP::increase_batchsize, MLP.hpp:403-435), so one step is split over 64-row chunks:
//
//   nfac_values_kernel   V(s), V(s')  per row                       grid (chunks, G)
//   nfac_scan_kernel     TD targets / lambda-return backward scan   grid (G)
//                        (:268-348) or advantages + CACLA selection (:464-522)
//   nfac_grad_kernel     forward + backward of one chunk, partial dW to gpart   grid (chunks, G)
//   nfac_apply_kernel    fixed-order reduction of the partials, Fisher EMA, EWC penalty, Adam
//                        (MLP::learn_batch's loop body, MLP.hpp:476-483)        grid (stride/1024, G)
//
// Weights are read from L2 by the MFMA products (same blocks as the DDPG kernel, net_device.h);
// the reduction order over chunks is fixed, so results are deterministic.
#include "net_device.h"

struct NfacLds { float *hbase; int hstride; float *x, *top; };

// compile-time shapes: SURVEY.md 8(d) config 3 (S = 45, A = 17, 64 x 64 hidden, leaky-ReLU, tanh actor output)
typedef SNet<1, 0, 45, 64, 64, 1> Cfg3Critic;
typedef SNet<1, 2, 45, 64, 64, 17> Cfg3Actor;

template <class ND>
__device__ __forceinline__ NfacLds nfac_carve(float *lds, const ND &nd, int ldh) {
  NfacLds M;
  float *p = lds;
  M.hbase = p; M.hstride = DDRL_MB * ldh; p += (nd.L() - 1) * M.hstride;
  M.x = p; p += DDRL_MB * nd.ld(0);
  M.top = p;
  return M;
}
static size_t nfac_lds_bytes(const NetDesc &nd, int ldh) {
  return ((size_t)(nd.L_ - 1) * DDRL_MB * ldh + (size_t)DDRL_MB * (nd.ld_[0] + nd.ld_[nd.L_])) * sizeof(float);
}

// 64 rows [row0, row0 + 64) of the state columns starting at `col0` of the trajectory into Xs
__device__ __forceinline__ void load_states(const float *__restrict__ traj, int rowlen, int T, int row0, int col0,
                                            int S, int ldx, const float *__restrict__ sgate, float *Xs, int tid,
                                            const GateMode &gm, int agent, int site) {
  for (int e = tid; e < DDRL_MB * ldx; e += DDRL_NTHREADS) {
    const int row = e / ldx, col = e - row * ldx;
    float v = (row0 + row < T && col < S) ? traj[(size_t)(row0 + row) * rowlen + col0 + col] : 0.f;
    // developmental state gate (weights shared by actor and critic; a Bernoulli draw per layer instance, pass and row)
    if (sgate && col < S) v *= ddrl_gate_factor(sgate[col], gm.st_bern, gm.st_scale, gm.seed, gm.draw, agent, site, row0 + row, col);
    Xs[e] = v;
  }
}

// all_nextV = vnn(s'), all_V = vnn(s)   (OfflineCaclaAg.hpp:266,315 / :460-461)
template <class ND>
__global__ void __launch_bounds__(DDRL_NTHREADS, DDRL_NTHREADS / 128)   // two workgroups per CU (<= 256 registers)
nfac_values_kernel(const NfacParams P) {
  extern __shared__ __align__(16) float lds[];
  const int chunk = blockIdx.x, agent = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: branches on it are uniform, waits are counted exactly
  if (P.mask && !P.mask[agent]) return;
  const int T = P.len[agent], row0 = chunk * DDRL_MB;
  if (row0 >= T) return;
  const auto &nv = Pick<ND>::get(P.v);
  const NfacLds M = nfac_carve(lds, nv, P.ldh);
  const int ldx = nv.ld(0), ldt = nv.ld(nv.L());
  const float *traj = P.traj + (size_t)agent * P.maxT * P.rowlen;
  const float *w = P.sv.w + (size_t)agent * nv.stride();
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    if (!((P.value_passes >> pass) & 1)) continue;     // uniform
    load_states(traj, P.rowlen, T, row0, pass == 0 ? 0 : P.S + 2 * P.A, P.S, ldx,
                P.sgate ? P.sgate + (size_t)agent * P.S : nullptr, M.x, tid, P.gm, agent, pass == 0 ? GS_NFAC_V_S : GS_NFAC_V_S2);
    __syncthreads();
    fwd_net(nv, nv.L(), w, M.x, ldx, M.hbase, M.hstride, M.top, ldt, wave, lane);
    float *out = (pass == 0 ? P.V : P.nextV) + (size_t)agent * P.maxT;
    if (tid < DDRL_MB && row0 + tid < T) out[row0 + tid] = M.top[tid * ldt];
    __syncthreads();
  }
}

// MODE 0, update_critic (OfflineCaclaAg.hpp:268-348): v_target = r + gamma V(s') (1 - goal);
//   lambda < 0: tgt = v_target; else deltas = v_target - V, backward scan
//   diff_t = delta_t + gamma lambda diff_{t+1} restarted at every episode's last row, tgt = diff + V.
// MODE 1, end_episode (:464-522): deltas (through the same scan when gae), n = #(delta > 0), and the
//   CACLA row selection.
// The scan runs sequentially in the reference's order from LDS.
template <int MODE>
__global__ void __launch_bounds__(256)
nfac_scan_kernel(const NfacParams P, int flag_off) {
  extern __shared__ __align__(16) float sd[];      // [T] deltas; from sd + flag_off (floats; 0: no room) the episode-end flags
  const int agent = blockIdx.x, tid = threadIdx.x;
  if (P.mask && !P.mask[agent]) return;
  const int T = P.len[agent];
  if (T <= 0) { if (MODE == 1 && tid == 0) P.n_pos[agent] = 0; return; }
  const size_t ro = (size_t)agent * P.maxT;
  const float *traj = P.traj + ro * P.rowlen;
  const float gamma = P.gamma[agent];
  const int cr = 2 * P.S + 2 * P.A;
  const bool scan = MODE == 0 ? P.lambda >= 0.f : P.gae != 0;
  for (int li = tid; li < T; li += 256) {
    const float *row = traj + (size_t)li * P.rowlen;
    float vt = row[cr];
    if (row[cr + 1] == 0.f) vt += gamma * P.nextV[ro + li];
    sd[li] = (MODE == 0 && P.lambda < 0.f) ? vt : vt - P.V[ro + li];
  }
  uint8_t *lf = (uint8_t *)(sd + flag_off);
  if (scan && flag_off)      // the flags beside the deltas: the walk below tests one per step (from memory: a round trip per step)
    for (int li = tid; li < T; li += 256) lf[li] = P.endflag[ro + li];
  __syncthreads();
  if (scan) {
    // the recurrence restarts at every episode's last row, so the episodes scan independently: one thread per episode end
    // walks its episode backwards (the reference's order within an episode, bit for bit)
    const float gl = gamma * P.lambda;
    auto walk = [&](const auto *ef) {
      for (int end = tid; end < T; end += 256) {
        if (!(ef[end] || end == T - 1)) continue;
        float prev = 0.f;
        for (int li = end; li >= 0; --li) {
          if (li != end && ef[li]) break;
          const float d = sd[li] + prev;
          sd[li] = d;
          prev = gl * d;
        }
      }
    };
    if (flag_off) walk(lf);
    else walk(P.endflag + ro);
  }
  __syncthreads();
  int n = 0;
  for (int li = tid; li < T; li += 256) {
    const float d = sd[li];
    if (MODE == 0) {
      P.tgt[ro + li] = P.lambda < 0.f ? d : d + P.V[ro + li];
    } else {
      P.deltas[ro + li] = d;
      uint8_t mode;
      if (d > 0.f) { mode = 1; ++n; }
      else mode = P.update_delta_neg ? 2 : 0;
      P.rowmode[ro + li] = mode;
    }
  }
  if (MODE == 1) {
    __shared__ int cnt[256];
    cnt[tid] = n;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) cnt[tid] += cnt[tid + s]; __syncthreads(); }
    if (tid == 0) P.n_pos[agent] = cnt[0];
  }
}

// Sum of the chunk partials in chunk order, then updateFisher(nsample) / regularize() /
// ApplyUpdate() / set_iter(iter + 1) (MLP.hpp:479-482; OfflineCaclaAg.hpp:567-570) on 4 parameters
// per thread and step.  iter_add: Adam steps already applied during this update (iter itself is bumped once
// at the end by nfac_finish_kernel).  One agent: elements [i0, i0 + 4 * 256 * nsweep) of its block, 256 threads.
template <int MODE, bool EWC>
__device__ __forceinline__ void nfac_apply_agent(const NfacParams &P, int agent, int iter_add, int last, int i0, int nsweep, bool bookkeeping,
                                                 int tid, float *s_lrt) {
  const int T = P.len[agent];
  const int npos = MODE == 1 ? P.n_pos[agent] : 1;
  const NetDesc &nd = MODE == 0 ? P.v : P.a;
  const NetState &st = MODE == 0 ? P.sv : P.sa;
  const int nch = (T + DDRL_MB - 1) / DDRL_MB;
  if (tid == 0) {
    const int t = st.iter[agent] + iter_add + 1;
    const double corr = sqrt(1.0 - pow(P.dbeta2, (double)t)) / (1.0 - pow(P.dbeta1, (double)t));
    *s_lrt = (float)((double)st.lr[agent] * corr);
  }
  __syncthreads();
  const float nsample = MODE == 0 ? (float)T : (float)npos;
  OptCtx o;
  o.dump = nullptr;
  const size_t off = (size_t)agent * nd.stride();
  o.w = st.w + off; o.m = st.m + off; o.v = st.v + off; o.wt = nullptr;
  o.fisher = EWC ? st.fisher + off : nullptr;
  o.fhat = EWC ? st.fhat + off : nullptr;
  o.wstar = EWC ? st.wstar + off : nullptr;
  o.lr_t = *s_lrt; o.tau = 0.f; o.omtau = 0.f;
  o.decay = -1.f;                                  // OfflineCaclaAg.hpp:186,190: no weight decay
  o.ewc_coef = EWC ? st.ewc_coef[agent] : 0.f;
  o.fisher_beta = P.fisher_beta; o.nsample = nsample;
  o.beta1 = P.beta1; o.beta2 = P.beta2; o.omb1 = P.omb1; o.omb2 = P.omb2; o.eps = P.eps;
  o.mld = P.beta1 == 0.f ? o.w : o.m;
  o.handoff = 0;
  for (int sw = 0; sw < nsweep; ++sw) {
    const int i = i0 + (sw * 256 + tid) * 4;
    if (i < nd.stride()) {
      OptRegs r;
      opt_load4<EWC, false, false>(o, i, r);
      const float *gp = (MODE == 0 ? P.gpart_v : P.gpart_a) + (size_t)agent * P.maxchunks * nd.stride() + i;
      f32x4 g = *(const f32x4 *)gp;
      for (int c = 1; c < nch; ++c) g += *(const f32x4 *)(gp + (size_t)c * nd.stride());
      opt_step4<EWC, false, false>(o, i, g, r, true);
    }
  }
  if (bookkeeping && tid == 0) {
    if (EWC) st.fisher_nbr[agent] = st.fisher_nbr[agent] * P.fisher_beta + nsample;   // DODevMLP.hpp:655
    if (MODE == 0 && last) {
      float s = 0.f;
      for (int c = 0; c < nch; ++c) s += P.loss_part[(size_t)agent * P.maxchunks + c];
      P.loss_out[agent] = s / (float)T * 0.5f;
    }
  }
}

template <int MODE, bool EWC>
__global__ void __launch_bounds__(256)
nfac_apply_kernel(const NfacParams P, int iter_add, int last) {
  const int agent = blockIdx.y, tid = threadIdx.x;
  if (P.mask && !P.mask[agent]) return;
  if (P.len[agent] <= 0) return;
  if (MODE == 1 && P.n_pos[agent] <= 0) return;
  __shared__ float s_lrt;
  nfac_apply_agent<MODE, EWC>(P, agent, iter_add, last, blockIdx.x * 1024, 1, blockIdx.x == 0, tid, &s_lrt);
}

// One chunk of one full-batch gradient.  MODE 0: critic, EuclideanLoss against tgt (MLP.hpp:950-957:
// d yhat = (yhat - y) / T).  MODE 1: actor, top diff = -(target - pi(s)) on the selected rows with
// optional inverting gradient (OfflineCaclaAg.hpp:531-566), no 1/T.
// P.fuse: the step itself rides in this launch -- every chunk workgroup publishes its partial (sc1 stores, the wave's vmcnt(0),
// barrier, one relaxed agent-scope counter increment: the hand-off forms of kernels_ddpg.hip), and the workgroup that
// finds the agent's other chunks done takes an agent-scope acquire and runs nfac_apply_agent over the agent's block while
// other agents' chunks are still computing.  Same reduction order and arithmetic as nfac_apply_kernel: bit-identical
// (tests/test_nfac_gpu.py).  Measured on config 3 (256 agents): 20.9 ms per update against 15.2 ms with the separate apply
// launches -- the write-through stores of 59 MB of partials per launch hold every chunk workgroup at its vmcnt(0), and the
// one workgroup that steps an agent walks 14 streams with nothing else in flight -- so it is opt-in (option "nfac_fuse").
template <int MODE, class ND, bool EWC>
__global__ void __launch_bounds__(DDRL_NTHREADS, DDRL_NTHREADS / 128)   // two workgroups per CU (<= 256 registers)
nfac_grad_kernel(const NfacParams P, int iter_add, int last) {
  extern __shared__ __align__(16) float lds[];
  const int chunk = blockIdx.x, agent = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: branches on it are uniform, waits are counted exactly
  if (P.mask && !P.mask[agent]) return;
  const int T = P.len[agent], row0 = chunk * DDRL_MB;
  if (row0 >= T) return;
  if (MODE == 1 && P.n_pos[agent] <= 0) return;
  const auto &nd = Pick<ND>::get(MODE == 0 ? P.v : P.a);
  const NetState &st = MODE == 0 ? P.sv : P.sa;
  const NfacLds M = nfac_carve(lds, nd, P.ldh);
  const int ldx = nd.ld(0), ldt = nd.ld(nd.L());
  const size_t ro = (size_t)agent * P.maxT;
  const float *traj = P.traj + ro * P.rowlen;
  const float *w = st.w + (size_t)agent * nd.stride();
  load_states(traj, P.rowlen, T, row0, 0, P.S, ldx, P.sgate ? P.sgate + (size_t)agent * P.S : nullptr, M.x, tid, P.gm, agent,
              MODE == 0 ? GS_NFAC_V_S : GS_NFAC_A_S);
  __syncthreads();
  fwd_net(nd, nd.L(), w, M.x, ldx, M.hbase, M.hstride, M.top, ldt, wave, lane);
  if (MODE == 0) {
    if (tid < DDRL_MB) {
      const int li = row0 + tid;
      float df = li < T ? M.top[tid * ldt] - P.tgt[ro + li] : 0.f;
      float sq = df * df;
      if (P.lw) {          // (uniform) the weighted loss of learn_batch_lw: sum w (yhat - y)^2 / (2T), d yhat = w (yhat - y) / T
        const float wl = li < T ? P.lw[ro + li] : 0.f;
        sq *= wl;
        df *= wl;
      }
      for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
      if (tid == 0) store_sc1(P.loss_part + (size_t)agent * P.maxchunks + chunk, sq);
      M.top[tid * ldt] = df / (float)T;
    }
  } else {
    const int A = P.A;
    for (int e = tid; e < DDRL_MB * A; e += DDRL_NTHREADS) {
      const int row = e / A, h = e - row * A, li = row0 + row;
      const float gt = P.agate ? ddrl_gate_factor(P.agate[(size_t)agent * A + h], P.gm.ac_bern, P.gm.ac_scale, P.gm.seed, P.gm.draw, agent,
                                                  GS_NFAC_A_OUT, li, h) : 1.f;
      const float raw = M.top[row * ldt + h];
      const float out = raw * gt;                                      // the actor's output gate
      float diff = 0.f;
      const int mode = li < T ? P.rowmode[ro + li] : 0;
      if (mode) {
        const float target = traj[(size_t)li * P.rowlen + P.S + (mode == 1 ? A : 0) + h];
        diff = -(target - out);
        if (P.corrected) diff *= P.corr_factor;                          // OfflineCaclaAg.hpp:561-563
        else if (P.inverting) {
          if (diff < 0.f) diff *= (1.0f - out) / 2.0f;
          else if (diff > 0.f) diff *= (out + 1.0f) / 2.0f;
        }
      }
      M.top[row * ldt + h] = act_bwd(nd.out_act(), raw, diff * gt);   // the actor's output layer, in place
    }
  }
  __syncthreads();
  auto none = [](int, int, float) {};
  OptCtx unused = {};
  unused.handoff = P.fuse;                            // partials read by another workgroup of this launch: sc1 stores
  float *gout = (MODE == 0 ? P.gpart_v : P.gpart_a) + ((size_t)agent * P.maxchunks + chunk) * nd.stride();
  bwd_net<DW_STORE, false>(nd, w, unused, gout, M.x, ldx, M.hbase, M.hstride, M.top, ldt, 0, 0, none, wave, lane, tid);
  if (!P.fuse) return;                                // uniform
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __shared__ int s_last;
  __shared__ float s_lrt;
  if (tid == 0) {
    const int nch = (T + DDRL_MB - 1) / DDRL_MB;
    const int before = __hip_atomic_fetch_add(P.chunk_done + agent, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int mine = before == nch - 1;
    if (mine) {
      __hip_atomic_store(P.chunk_done + agent, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    s_last = mine;
  }
  __syncthreads();
  if (!s_last) return;
  nfac_apply_agent<MODE, EWC>(P, agent, iter_add, last, 0, (nd.stride() + 1023) / 1024, true, tid, &s_lrt);
}

// vnn_from_scratch (OfflineCaclaAg.hpp:281-284): the critic of every agent with a trajectory is replaced by the freshly
// drawn net in `fresh` (same layout, pads zero), its Adam moments zeroed; iter_base[agent] receives the solver's old
// iteration count so that the steps that follow count from a new solver (nfac_apply_kernel adds iter_add to iter).
__global__ void __launch_bounds__(256)
nfac_reinit_kernel(const NfacParams P, const float *__restrict__ fresh) {
  const int agent = blockIdx.y;
  if (P.mask && !P.mask[agent]) return;
  if (P.len[agent] <= 0) return;
  const int stride = P.v.stride();
  const size_t off = (size_t)agent * stride;
  for (int i = 4 * (blockIdx.x * 256 + threadIdx.x); i < stride; i += 4 * 256 * gridDim.x) {
    *(f32x4 *)(P.sv.w + off + i) = *(const f32x4 *)(fresh + off + i);
    *(f32x4 *)(P.sv.m + off + i) = zero4();
    *(f32x4 *)(P.sv.v + off + i) = zero4();
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) P.sv.iter[agent] = 0;       // a new solver
}

// Solver::iter after the update: + number_fitted_iteration * stoch_iter_critic for the critic,
// + stoch_iter_actor for the actor when at least one transition had a positive advantage
__global__ void nfac_finish_kernel(const NfacParams P, int critic_steps, int actor_steps) {
  const int agent = blockIdx.x * blockDim.x + threadIdx.x;
  if (agent >= P.G) return;
  if (P.mask && !P.mask[agent]) return;
  if (P.len[agent] <= 0) return;
  P.sv.iter[agent] += critic_steps;
  if (P.n_pos[agent] > 0) P.sa.iter[agent] += actor_steps;
}

// ---------------------------------------------------------------------------------------------
extern "C" size_t nfac_lds_need(const NfacParams *P) {
  return std::max(nfac_lds_bytes(P->a, P->ldh), nfac_lds_bytes(P->v, P->ldh));
}

template <class ND>
static hipError_t values_launch_t(const NfacParams *P, int chunks, size_t lds, hipStream_t st) {
  static LdsConfig conf;
  hipError_t e = ddrl_set_lds(nfac_values_kernel<ND>, lds, conf);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(nfac_values_kernel<ND>, dim3(chunks, P->G), dim3(DDRL_NTHREADS), lds, st, *P);
  return hipGetLastError();
}
extern "C" hipError_t nfac_values_launch(const NfacParams *P, int chunks, hipStream_t st) {
  const size_t lds = nfac_lds_bytes(P->v, P->ldh);
  if (!P->generic && Cfg3Critic::matches(P->v)) return values_launch_t<Cfg3Critic>(P, chunks, lds, st);
  return values_launch_t<NetDesc>(P, chunks, lds, st);
}

extern "C" hipError_t nfac_scan_launch(const NfacParams *P, int mode, int maxlen, hipStream_t st) {
  static LdsConfig conf0, conf1;
  const int nf = std::max(maxlen, 1);
  size_t lds = (size_t)nf * sizeof(float);
  int flag_off = 0;                                  // the end flags in LDS too where they fit (max_trajectory up to 32768 rows: 128 KB of deltas)
  if (lds + (size_t)((nf + 15) & ~15) + 2048 <= 160 * 1024) { flag_off = nf; lds += (size_t)((nf + 15) & ~15); }
  hipError_t e = mode == 0 ? ddrl_set_lds(nfac_scan_kernel<0>, lds, conf0) : ddrl_set_lds(nfac_scan_kernel<1>, lds, conf1);
  if (e != hipSuccess) return e;
  if (mode == 0) hipLaunchKernelGGL(nfac_scan_kernel<0>, dim3(P->G), dim3(256), lds, st, *P, flag_off);
  else hipLaunchKernelGGL(nfac_scan_kernel<1>, dim3(P->G), dim3(256), lds, st, *P, flag_off);
  return hipGetLastError();
}

template <int MODE, class ND, bool EWC>
static hipError_t grad_launch_t(const NfacParams *P, int chunks, size_t lds, int iter_add, int last, hipStream_t st) {
  static LdsConfig conf;
  hipError_t e = ddrl_set_lds(nfac_grad_kernel<MODE, ND, EWC>, lds, conf);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((nfac_grad_kernel<MODE, ND, EWC>), dim3(chunks, P->G), dim3(DDRL_NTHREADS), lds, st, *P, iter_add, last);
  return hipGetLastError();
}
template <int MODE, class ND>
static hipError_t grad_launch_e(const NfacParams *P, int chunks, size_t lds, int iter_add, int last, hipStream_t st) {
  const bool ewc = P->fuse && (MODE == 0 ? P->sv : P->sa).fisher != nullptr;     // (unfused: the template flag is unused)
  return ewc ? grad_launch_t<MODE, ND, true>(P, chunks, lds, iter_add, last, st) : grad_launch_t<MODE, ND, false>(P, chunks, lds, iter_add, last, st);
}
extern "C" hipError_t nfac_grad_launch(const NfacParams *P, int mode, int chunks, int iter_add, int last, hipStream_t st) {
  const size_t lds = nfac_lds_bytes(mode == 0 ? P->v : P->a, P->ldh);
  if (mode == 0) {
    if (!P->generic && Cfg3Critic::matches(P->v)) return grad_launch_e<0, Cfg3Critic>(P, chunks, lds, iter_add, last, st);
    return grad_launch_e<0, NetDesc>(P, chunks, lds, iter_add, last, st);
  }
  if (!P->generic && Cfg3Actor::matches(P->a)) return grad_launch_e<1, Cfg3Actor>(P, chunks, lds, iter_add, last, st);
  return grad_launch_e<1, NetDesc>(P, chunks, lds, iter_add, last, st);
}

extern "C" hipError_t nfac_apply_launch(const NfacParams *P, int mode, int iter_add, int last, hipStream_t st) {
  const NetDesc &nd = mode == 0 ? P->v : P->a;
  const bool ewc = (mode == 0 ? P->sv : P->sa).fisher != nullptr;
  const dim3 grid((nd.stride_ + 1023) / 1024, P->G);
  if (mode == 0) {
    if (ewc) hipLaunchKernelGGL((nfac_apply_kernel<0, true>), grid, dim3(256), 0, st, *P, iter_add, last);
    else hipLaunchKernelGGL((nfac_apply_kernel<0, false>), grid, dim3(256), 0, st, *P, iter_add, last);
  } else {
    if (ewc) hipLaunchKernelGGL((nfac_apply_kernel<1, true>), grid, dim3(256), 0, st, *P, iter_add, last);
    else hipLaunchKernelGGL((nfac_apply_kernel<1, false>), grid, dim3(256), 0, st, *P, iter_add, last);
  }
  return hipGetLastError();
}

extern "C" hipError_t nfac_reinit_launch(const NfacParams *P, const float *fresh, hipStream_t st) {
  hipLaunchKernelGGL(nfac_reinit_kernel, dim3(4, P->G), dim3(256), 0, st, *P, fresh);
  return hipGetLastError();
}

extern "C" hipError_t nfac_finish_launch(const NfacParams *P, int critic_steps, int actor_steps, hipStream_t st) {
  hipLaunchKernelGGL(nfac_finish_kernel, dim3((P->G + 255) / 256), dim3(256), 0, st, *P, critic_steps, actor_steps);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// MLP::computeOutBatch / computeOutVFBatch (MLP.hpp:531-579) for every agent: n rows per agent through
// one net.  X [G x n x d0], out [G x n x dL]; in_gate [G x d0] / out_gate [G x dL] or null.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(DDRL_NTHREADS, DDRL_NTHREADS / 128)   // two workgroups per CU (<= 256 registers)
net_forward_kernel(const NetDesc nd, const float *__restrict__ w, const float *__restrict__ X, float *__restrict__ out,
                   int n, int ldh, const float *__restrict__ sgate, int S, const float *__restrict__ agate, int A,
                   int gate_out, const GateMode gm) {
  extern __shared__ __align__(16) float lds[];
  const int chunk = blockIdx.x, agent = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = chunk * DDRL_MB;
  const NfacLds M = nfac_carve(lds, nd, ldh);
  const int d0 = nd.d(0), dl = nd.d(nd.L()), ldx = nd.ld(0), ldt = nd.ld(nd.L());
  const float *xin = X + (size_t)agent * n * d0;
  for (int e = tid; e < DDRL_MB * ldx; e += DDRL_NTHREADS) {
    const int row = e / ldx, col = e - row * ldx;
    float v = (row0 + row < n && col < d0) ? xin[(size_t)(row0 + row) * d0 + col] : 0.f;
    if (sgate && col < S)
      v *= ddrl_gate_factor(sgate[(size_t)agent * S + col], gm.st_bern, gm.st_scale, gm.seed, gm.draw, agent, GS_BATCH_S, row0 + row, col);
    if (agate && !gate_out && col >= S && col < S + A)                                              // critic's action input
      v *= ddrl_gate_factor(agate[(size_t)agent * A + col - S], gm.ac_bern, gm.ac_scale, gm.seed, gm.draw, agent, GS_BATCH_A, row0 + row, col - S);
    M.x[e] = v;
  }
  __syncthreads();
  fwd_net(nd, nd.L(), w + (size_t)agent * nd.stride(), M.x, ldx, M.hbase, M.hstride, M.top, ldt, wave, lane);
  float *o = out + (size_t)agent * n * dl;
  for (int e = tid; e < DDRL_MB * dl; e += DDRL_NTHREADS) {
    const int row = e / dl, col = e - row * dl;
    if (row0 + row < n) {
      float v = M.top[row * ldt + col];
      if (agate && gate_out)                                                                        // actor's output
        v *= ddrl_gate_factor(agate[(size_t)agent * A + col], gm.ac_bern, gm.ac_scale, gm.seed, gm.draw, agent, GS_BATCH_A, row0 + row, col);
      o[(size_t)(row0 + row) * dl + col] = v;
    }
  }
}

// The same for nets wider than the LDS-resident passes take (DDRL_MAXW < width <= DDRL_WIDE_MAXW; BASELINE.md
// "large"): one workgroup per (64-row chunk, agent), activations ping-pong through a global scratch block
// [G x chunks x 2 x 64 x ldw] that stays in L2, one thread per output element, plain fp32 FMA chains over k
// (this entry point serves inspection and evaluation calls, not the update path: kernels_ddpg_wide.hip).
__global__ void __launch_bounds__(256)
net_forward_wide_kernel(const NetDesc nd, const float *__restrict__ w, const float *__restrict__ X, float *__restrict__ out,
                        float *__restrict__ scratch, int ldw, int n, const float *__restrict__ sgate, int S,
                        const float *__restrict__ agate, int A, int gate_out, const GateMode gm) {
  const int chunk = blockIdx.x, agent = blockIdx.y, tid = threadIdx.x;
  const int row0 = chunk * DDRL_MB, d0 = nd.d(0), dl = nd.d(nd.L());
  float *buf = scratch + ((size_t)agent * gridDim.x + chunk) * 2 * DDRL_MB * ldw;
  const float *xin = X + (size_t)agent * n * d0;
  const float *wn = w + (size_t)agent * nd.stride();
  for (int e = tid; e < DDRL_MB * d0; e += 256) {
    const int row = e / d0, col = e - row * d0;
    float v = row0 + row < n ? xin[(size_t)(row0 + row) * d0 + col] : 0.f;
    if (sgate && col < S)
      v *= ddrl_gate_factor(sgate[(size_t)agent * S + col], gm.st_bern, gm.st_scale, gm.seed, gm.draw, agent, GS_BATCH_S, row0 + row, col);
    if (agate && !gate_out && col >= S && col < S + A)
      v *= ddrl_gate_factor(agate[(size_t)agent * A + col - S], gm.ac_bern, gm.ac_scale, gm.seed, gm.draw, agent, GS_BATCH_A, row0 + row, col - S);
    buf[row * ldw + col] = v;
  }
  __syncthreads();
  for (int l = 0; l < nd.L(); ++l) {
    const float *xi = buf + (l & 1) * DDRL_MB * ldw;
    float *yo = buf + ((l + 1) & 1) * DDRL_MB * ldw;
    const int K = nd.d(l), N = nd.d(l + 1), kp = nd.kp(l);
    const float *Wl = wn + nd.offw(l), *bias = wn + nd.offb(l);
    const int act = l == nd.L() - 1 ? nd.out_act() : nd.hidden_act();
    for (int e = tid; e < DDRL_MB * N; e += 256) {
      const int col = e % N, row = e / N;                  // consecutive threads: consecutive weight rows of one batch row
      float acc = bias[col];
      for (int k = 0; k < K; ++k) acc = fmaf(xi[row * ldw + k], Wl[ddrl_widx(col, k, kp)], acc);
      yo[row * ldw + col] = act_fwd(act, acc);
    }
    __syncthreads();                                       // (workgroup-scope fence included: the next layer reads other threads' outputs)
  }
  const float *top = buf + (nd.L() & 1) * DDRL_MB * ldw;
  float *o = out + (size_t)agent * n * dl;
  for (int e = tid; e < DDRL_MB * dl; e += 256) {
    const int row = e / dl, col = e - row * dl;
    if (row0 + row < n) {
      float v = top[row * ldw + col];
      if (agate && gate_out)
        v *= ddrl_gate_factor(agate[(size_t)agent * A + col], gm.ac_bern, gm.ac_scale, gm.seed, gm.draw, agent, GS_BATCH_A, row0 + row, col);
      o[(size_t)(row0 + row) * dl + col] = v;
    }
  }
}

extern "C" size_t net_forward_wide_scratch_floats(const NetDesc *nd, int n, int G) {
  int ldw = 4;
  for (int l = 0; l <= nd->L_; ++l) ldw = std::max(ldw, nd->d_[l]);
  return (size_t)G * ((n + DDRL_MB - 1) / DDRL_MB) * 2 * DDRL_MB * ldw;
}

extern "C" hipError_t net_forward_wide_launch(const NetDesc *nd, const float *w, const float *X, float *out, float *scratch, int n,
                                              int G, const float *sgate, int S, const float *agate, int A, int gate_out, const GateMode *gm, hipStream_t st) {
  int ldw = 4;
  for (int l = 0; l <= nd->L_; ++l) ldw = std::max(ldw, nd->d_[l]);
  hipLaunchKernelGGL(net_forward_wide_kernel, dim3((n + DDRL_MB - 1) / DDRL_MB, G), dim3(256), 0, st, *nd, w, X, out, scratch, ldw,
                     n, sgate, S, agate, A, gate_out, *gm);
  return hipGetLastError();
}

extern "C" hipError_t net_forward_launch(const NetDesc *nd, const float *w, const float *X, float *out, int n, int G,
                                         const float *sgate, int S, const float *agate, int A, int gate_out, const GateMode *gm, hipStream_t st) {
  static LdsConfig conf;
  int ldh = 4;
  for (int l = 1; l < nd->L_; ++l) ldh = std::max(ldh, nd->ld_[l]);
  const size_t lds = nfac_lds_bytes(*nd, ldh);
  hipError_t e = ddrl_set_lds(net_forward_kernel, lds, conf);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(net_forward_kernel, dim3((n + DDRL_MB - 1) / DDRL_MB, G), dim3(DDRL_NTHREADS), lds, st, *nd, w, X, out,
                     n, ldh, sgate, S, agate, A, gate_out, *gm);
  return hipGetLastError();
}
