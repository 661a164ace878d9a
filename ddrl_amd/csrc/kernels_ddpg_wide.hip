// kernels_ddpg_wide.hip -- the DDPG update for nets wider than the LDS-resident kernels take
// (DDRL_MAXW = 128 < width <= DDRL_WIDE_MAXW; BASELINE.md "large": 400 x 300).  Same arithmetic and
// the same order of operations as the fused kernel (kernels_ddpg.hip, DeepQNAg.hpp:343-466), but one
// launch per layer operation: the 64-row activation tiles of an (agent, chunk) live in an HBM scratch
// block (L2 resident between launches) instead of LDS, every launch spreads an agent's layer over
// several workgroups (64-column blocks), and a layer's dW kernel sums the minibatch's chunks in its
// MFMA accumulators and runs the optimiser epilogue itself (no per-chunk partials, no apply launch).
// Weights keep the tile-major layout, so weight fragments are still contiguous 1-KiB wave loads.
#include "ddpg_device.h"
#include <mutex>

struct WideMap {
  LdsMap M;              // the fused kernel's tiles, here pointers into the scratch block
  float *ha, *hq, *da;   // hidden activations of the actor / critic pass, raw dQ/da [64 x ldpad(A)]
  float *dz;             // dZ_l of the hidden layers of the backward pass in flight: the activations stay intact for dW
  int hstride;
};

__host__ __device__ inline size_t wide_block_floats(const DdpgParams &P) {
  const int A = P.a.d_[P.a.L_];
  size_t f = (size_t)DDRL_MB * (P.ldxc + P.ldxa + P.q.ld_[P.q.L_] + P.a.ld_[P.a.L_] + 5 + ddrl_ldpad(A));
  f += (size_t)(P.a.L_ - 1 + P.q.L_ - 1 + (P.a.L_ > P.q.L_ ? P.a.L_ : P.q.L_) - 1) * DDRL_MB * P.ldh;
  return (f + 63) / 64 * 64;
}

__device__ __forceinline__ WideMap wide_map(const DdpgParams &P, int agent, int chunk) {
  WideMap W;
  float *p = P.wide + ((size_t)agent * P.maxchunks + chunk) * P.wide_stride;
  const int A = P.a.d(P.a.L());
  W.M.hbase = nullptr; W.M.hstride = 0;
  W.M.xc = p; p += DDRL_MB * P.ldxc;
  W.M.xa = p; p += DDRL_MB * P.ldxa;
  W.M.topq = p; p += DDRL_MB * P.q.ld(P.q.L());
  W.M.topa = p; p += DDRL_MB * P.a.ld(P.a.L());
  W.M.yv = p; p += DDRL_MB;
  W.M.rv = p; p += DDRL_MB;
  W.M.dn = p; p += DDRL_MB;
  W.M.nrew = p; p += DDRL_MB;
  W.M.nvalid = (int *)p; p += DDRL_MB;
  W.da = p; p += DDRL_MB * ddrl_ldpad(A);
  W.hstride = DDRL_MB * P.ldh;
  W.ha = p; p += (P.a.L() - 1) * W.hstride;
  W.hq = p; p += (P.q.L() - 1) * W.hstride;
  W.dz = p;
  return W;
}

// the update guard of DeepQNAg.hpp:338 + mask + chunks beyond the minibatch; policy: a launch of TD3's
// delayed actor update, only for the agents on a policy step (TD3Ag.hpp:333)
__device__ __forceinline__ bool wide_skip(const DdpgParams &P, int agent, int chunk, int policy = 0) {
  if (P.mask && !P.mask[agent]) return true;
  if (policy && P.actor_step && !P.actor_step[agent]) return true;
  const int len = P.rep_len[agent];
  return len < 250 || len < P.B || chunk * DDRL_MB >= P.B;
}

struct WideOp { int net, target, l, mode, policy, agent0, ng; };      // net: 0 actor, 1 critic, 2 second critic (TD3); the launch covers agents [agent0, agent0 + ng)

// where layer l of a net reads its input / writes its output inside the scratch block
__device__ __forceinline__ void wide_io(const DdpgParams &P, const WideMap &W, int net, int l, const NetDesc &nd,
                                        float *&in, int &ldi, float *&out, int &ldo) {
  float *h = net ? W.hq : W.ha;
  in = l == 0 ? (net ? W.M.xc : W.M.xa) : h + (l - 1) * W.hstride;
  ldi = nd.ld(l);
  const bool last = l == nd.L() - 1;
  out = last ? (net ? W.M.topq : W.M.topa) : h + l * W.hstride;
  ldo = nd.ld(l + 1);
}
// where the backward pass keeps dZ_l (gradient at layer l's pre-activation output): the top tile for the last
// layer (the loss steps write it there), a buffer of its own below
__device__ __forceinline__ float *wide_dz(const WideMap &W, int net, int l, const NetDesc &nd) {
  return l == nd.L() - 1 ? (net ? W.M.topq : W.M.topa) : W.dz + l * W.hstride;
}

// Y[64 x N] = act(X W^T + b): one 64-column block of outputs per workgroup, wave w owns rows 16w..16w+15
__global__ void __launch_bounds__(256)
wide_fwd_kernel(const DdpgParams P, const WideOp op) {
  const NetDesc &nd = op.net ? P.q : P.a;
  const int l = op.l, kp = nd.kp(l), N = nd.d(l + 1);
  int nb, chunk, agent;
  if (!xcd_decode((nd.ld(l + 1) + 63) / 64, (P.B + DDRL_MB - 1) / DDRL_MB, op.ng, nb, chunk, agent)) return;
  agent += op.agent0;
  if (wide_skip(P, agent, chunk, op.policy)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  const NetState &st = op.net == 2 ? P.sq2 : op.net ? P.sq : P.sa;
  const float *wn = (op.target ? st.wt : st.w) + (size_t)agent * nd.stride();
  const float *Wl = wn + nd.offw(l), *bias = wn + nd.offb(l);
  const WideMap W = wide_map(P, agent, chunk);
  float *X, *Y; int ldx, ldy;
  wide_io(P, W, op.net, l, nd, X, ldx, Y, ldy);
  const int act = l == nd.L() - 1 ? nd.out_act() : nd.hidden_act();
  const float *xrow = X + (16 * wave + c) * ldx;
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
  const int KC = (kp + 15) >> 4;
  // The four waves need the same weight fragments (they differ in their batch rows only): wave w fetches tile w
  // of the k-chunk once and hands it to the others through LDS, double-buffered, one barrier per k-chunk;
  // chunk kc + 1 travels (global -> registers) while chunk kc is multiplied.
  __shared__ __align__(16) float sB[2][2][4][256];     // two k-chunks per barrier
  auto load_a = [&](int kc) {              // chunks beyond the layer multiply as zeros
    const int kcc = kc < KC ? kc : KC - 1;
    const bool in = kc < KC && kcc * 16 + 4 * g < kp;
    f32x4 a = *(const f32x4 *)(xrow + (kcc * 16 + 4 * g < kp ? kcc * 16 + 4 * g : kp - 4));
    return in ? a : zero4();
  };
  auto load_b = [&](int kc) {              // this wave's tile (t = wave) of chunk kc
    const int kcc = kc < KC ? kc : KC - 1;
    const int kw = ddrl_tile_kw(kcc, kp);
    const int nt = nb * 4 + wave < (N + 15) / 16 ? nb * 4 + wave : (N + 15) / 16 - 1;     // clamped: tiles beyond N are discarded below
    const float *blk = Wl + ddrl_tile_base(nt, kcc, kp) + c * kw;
    return *(const f32x4 *)(blk + (4 * g < kw ? 4 * g : kw - 4));
  };
  f32x4 a_cur[2] = {load_a(0), load_a(1)}, b_nxt[2] = {load_b(0), load_b(1)}, a_nxt[2];
  *(f32x4 *)&sB[0][0][wave][lane * 4] = b_nxt[0];
  *(f32x4 *)&sB[0][1][wave][lane * 4] = b_nxt[1];
  for (int kc = 0, it = 0; kc < KC; kc += 2, ++it) {
    const int cur = it & 1;
    a_nxt[0] = load_a(kc + 2); a_nxt[1] = load_a(kc + 3);
    b_nxt[0] = load_b(kc + 2); b_nxt[1] = load_b(kc + 3);
    __syncthreads();                       // sB[cur] is complete; nobody reads sB[cur ^ 1] any more
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 b = *(const f32x4 *)&sB[cur][h][t][lane * 4];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t] = mfma16(a_cur[h][e], b[e], acc[t]);
      }
    *(f32x4 *)&sB[cur ^ 1][0][wave][lane * 4] = b_nxt[0];
    *(f32x4 *)&sB[cur ^ 1][1][wave][lane * 4] = b_nxt[1];
    a_cur[0] = a_nxt[0]; a_cur[1] = a_nxt[1];
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int n = (nb * 4 + t) * 16 + c;
    if (n < ldy) {
      const float bv = n < N ? bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) Y[(16 * wave + 4 * g + r) * ldy + n] = n < N ? act_fwd(act, acc[t][r] + bv) : 0.f;
    }
  }
}

// dX[64 x 64-column block] = dZ W for layer l.  mode 0: dZ_{l-1} = dX x the derivative of layer l-1's
// activation, taken from its output H_{l-1} (MLP.hpp:799-803), into the dZ buffer -- H_{l-1} stays intact: the
// dW kernel of layer l reads it AFTER this kernel, because with a one-chunk minibatch it updates W_l in place.
// mode 1 (layer 0 of the critic): the action columns [S, S + A) go to W.da untouched.
__global__ void __launch_bounds__(256)
wide_dx_kernel(const DdpgParams P, const WideOp op) {
  const NetDesc &nd = op.net ? P.q : P.a;
  const int l = op.l, kp = nd.kp(l), K = nd.d(l), N = nd.d(l + 1);
  int kb, chunk, agent;
  if (!xcd_decode((kp + 63) / 64, (P.B + DDRL_MB - 1) / DDRL_MB, op.ng, kb, chunk, agent)) return;
  agent += op.agent0;
  if (wide_skip(P, agent, chunk, op.policy)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
  const NetState &st = op.net == 2 ? P.sq2 : op.net ? P.sq : P.sa;
  const float *Wl = st.w + (size_t)agent * nd.stride() + nd.offw(l);
  const WideMap W = wide_map(P, agent, chunk);
  float *in, *dZ; int ldi, ldz;
  wide_io(P, W, op.net, l, nd, in, ldi, dZ, ldz);
  dZ = wide_dz(W, op.net, l, nd);
  const int n4 = ddrl_roundup(N, 4);
  const float *zrow = dZ + (16 * wave + c) * ldz;
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
  const int NC = (N + 15) >> 4, KT = (kp + 15) >> 4;
  // as in wide_fwd_kernel: wave w fetches the weight fragments of column tile w of the chunk, the others take
  // them from LDS (double-buffered, one barrier per chunk)
  __shared__ __align__(16) float sB[2][4][256];
  auto load_a = [&](int nc) {
    const int ncc = nc < NC ? nc : NC - 1;
    const bool inn = ncc * 16 + 4 * g < n4;    // columns [N, n4) of dZ hold zeros
    f32x4 a = *(const f32x4 *)(zrow + (inn ? ncc * 16 + 4 * g : n4 - 4));
    return inn ? a : zero4();
  };
  auto load_b = [&](int nc) {
    const int ncc = nc < NC ? nc : NC - 1;
    const int kt = kb * 4 + wave < KT ? kb * 4 + wave : KT - 1;                          // clamped: discarded below
    const int kw = ddrl_tile_kw(kt, kp);
    const float *blk = Wl + ddrl_tile_base(ncc, kt, kp) + 4 * g * kw + (c < kw ? c : kw - 1);   // pad rows (n >= N) hold zeros
    f32x4 b;
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = blk[j * kw];
    return b;
  };
  f32x4 a_cur = load_a(0), b_nxt = load_b(0), a_nxt = load_a(1), a_far, b_far;
  *(f32x4 *)&sB[0][wave][lane * 4] = b_nxt;
  b_nxt = load_b(1);
  for (int nc = 0; nc < NC; ++nc) {
    const int cur = nc & 1;
    a_far = load_a(nc + 2);
    b_far = load_b(nc + 2);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 b = *(const f32x4 *)&sB[cur][t][lane * 4];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[t] = mfma16(a_cur[j], b[j], acc[t]);
    }
    *(f32x4 *)&sB[cur ^ 1][wave][lane * 4] = b_nxt;
    a_cur = a_nxt; a_nxt = a_far; b_nxt = b_far;
  }
  const int S = P.S, A = P.A, ldda = ddrl_ldpad(A);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int col = (kb * 4 + t) * 16 + c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * wave + 4 * g + r;
      if (op.mode == 0) {
        if (col < kp) wide_dz(W, op.net, l - 1, nd)[row * ldi + col] = col < K ? act_bwd(nd.hidden_act(), in[row * ldi + col], acc[t][r]) : 0.f;
      } else if (col >= S && col < S + A) {
        W.da[row * ldda + col - S] = acc[t][r];
      }
    }
  }
}

// dW[64 n-block x all k] = sum over the minibatch's 64-row chunks of dZ^T X (+ db), every tile handed from the
// MFMA accumulators straight to the optimiser epilogue of the fused kernel (opt_load4 / opt_step4: Fisher EMA,
// EWC penalty, L2, Adam, soft target update): no partial-gradient image, no apply launch.  One workgroup per
// 64-column block of dZ; for every 64-column block of X the chunks' tiles of dZ and X are staged in LDS in turn
// (one chunk: the block of dZ is staged once and the blocks of X stream past it).  The layer's input-gradient
// kernel has run before (old weights), and the activations are still intact (wide_dz).
template <bool EWC>
__global__ void __launch_bounds__(256)
wide_dw_kernel(const DdpgParams P, const WideOp op) {
  __shared__ __align__(16) float sY[DDRL_MB * 68], sX[DDRL_MB * 68];
  const int nb = blockIdx.x, agent = blockIdx.z + op.agent0;
  if (wide_skip(P, agent, 0, op.policy)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
  const NetDesc &nd = op.net ? P.q : P.a;
  const int l = op.l, kp = nd.kp(l), N = nd.d(l + 1);
  const int KB = (kp + 63) >> 6, NCH = (P.B + DDRL_MB - 1) / DDRL_MB;
  // 16-byte pieces: row strides and block offsets are multiples of 4 floats
  auto stage = [&](float *dst, const float *src, int ld, int col0, int width) {
    for (int e = tid; e < DDRL_MB * 16; e += 256) {
      const int row = e >> 4, c4 = (e & 15) * 4;
      f32x4 v = zero4();
      if (col0 + c4 < width) v = *(const f32x4 *)(src + row * ld + col0 + c4);
#pragma unroll
      for (int j = 0; j < 4; ++j) if (col0 + c4 + j >= width) v[j] = 0.f;
      *(f32x4 *)(dst + row * 68 + c4) = v;
    }
  };
  const int Nloc = N - nb * 64 < 64 ? N - nb * 64 : 64;
  // TD3 (the only caller that sets actor_step): the critics' targets move on a policy step only (tau = 0 in between)
  const bool pstep = P.actor_step ? P.actor_step[agent] != 0 : true;
  const NetState &st = op.net == 2 ? P.sq2 : op.net ? P.sq : P.sa;
  const OptCtx o = make_opt<EWC>(P, st, nd.stride(), agent, pstep ? P.tau[agent] : 0.f, pstep ? P.omtau[agent] : 1.f, (float)P.B);
  float db = 0.f;
  for (int kb = 0; kb < KB; ++kb) {
    const int kploc = kp - kb * 64 < 64 ? kp - kb * 64 : 64;
    auto elem = [&](int t, bool &ok) {     // lane's four parameters of tile t of this (n, k) block
      const int p = wave * 4 + t;
      const int n = nb * 64 + (p >> 2) * 16 + c, k = kb * 64 + (p & 3) * 16 + 4 * g;
      ok = n < N && k < kp;
      return ok ? nd.offw(l) + ddrl_widx(n, k, kp) : nd.offw(l);      // (clamped: the state loaded there is ignored)
    };
    f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
    for (int ch = 0; ch < NCH; ++ch) {
      const WideMap W = wide_map(P, agent, ch);
      float *X, *dZ; int ldx, ldz;
      wide_io(P, W, op.net, l, nd, X, ldx, dZ, ldz);
      if (kb || ch) __syncthreads();         // every wave is done with the previous tiles
      if (NCH > 1 || kb == 0) stage(sY, wide_dz(W, op.net, l, nd), ldz, nb * 64, N);   // one chunk: staged once
      stage(sX, X, ldx, kb * 64, kp);
      __syncthreads();
      if (kb == 0 && tid < 64) db += bias_grad(sY, 68, tid);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] += dw_tile(sY, 68, Nloc, sX, 68, kploc, wave * 4 + t, 16, 4, c, g);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {          // (requesting the state before the products measured no faster)
      bool ok;
      const int i = elem(t, ok);
      OptRegs r;
      opt_load4<EWC, true>(o, i, r);
      opt_step4<EWC, true>(o, i, acc[t], r, ok);
    }
  }
  if (tid < 64 && nb * 64 + tid < N) opt_step<EWC, true>(o, nd.offb(l) + nb * 64 + tid, db);
}

// the element-wise steps between the layer launches, one workgroup per (chunk, agent)
__global__ void __launch_bounds__(DDRL_NTHREADS)
wide_step_kernel(const DdpgParams P, int step, int policy, int agent0) {
  const int chunk = blockIdx.x, agent = blockIdx.y + agent0, tid = threadIdx.x;
  if (wide_skip(P, agent, chunk, policy)) return;
  const WideMap W = wide_map(P, agent, chunk);
  const LdsMap &M = W.M;
  const int S = P.S, A = P.A, B = P.B;
  const int row0 = chunk * DDRL_MB;
  const int nrows = B - row0 < DDRL_MB ? B - row0 : DDRL_MB;
  const int ldxc = P.ldxc, ldxa = P.ldxa, ldq = P.q.ld(P.q.L()), lda = P.a.ld(P.a.L());
  const int *idx = P.idx + (size_t)agent * B + row0;
  const int len = P.rep_len[agent], head = P.rep_head[agent];
  const float gamma = P.gamma[agent];
  const float *agate = P.agate ? P.agate + (size_t)agent * A : nullptr;
  const bool bern = (P.sgate || P.agate) && (P.gm.st_bern || P.gm.ac_bern);     // uniform
  if (step == 0) {                       // s' tiles for the target pass
    if (P.nstep > 0) nstep_horizon(P, S, A, agent, idx, nrows, len, head, gamma, M, tid);
    gather_tile(P, S, A, ldxc, ldxa, agent, idx, nrows, head, true, M, tid, 0, row0);
  } else if (step == 1 || step == 13) {  // actor output (gated) into the critic's action columns (gated again); 1: target nets, 13: trained nets
    for (int e = tid; e < DDRL_MB * A; e += DDRL_NTHREADS) {
      const int row = e / A, col = e - row * A;
      float a = M.topa[row * lda + col];
      if (agate) {
        const GateMode &gm = P.gm;        // Bernoulli gates: each layer instance's mask per minibatch row, as in the one-tile kernel
        a *= ddrl_gate_factor(agate[col], gm.ac_bern, gm.ac_scale, gm.seed, gm.draw, agent, step == 1 ? GS_ACTOR_T_OUT : GS_ACTOR_OUT, row0 + row, col);
        M.topa[row * lda + col] = a;
        a *= ddrl_gate_factor(agate[col], gm.ac_bern, gm.ac_scale, gm.seed, gm.draw, agent, step == 1 ? GS_CRITIC_T_A : GS_CRITIC_A, row0 + row, col);
      }
      M.xc[row * ldxc + S + col] = a;
    }
  } else if (step == 2) {                // [s|a] tiles + TD targets (DeepQNAg.hpp:386-405)
    gather_tile(P, S, A, ldxc, ldxa, agent, idx, nrows, head, false, M, tid, 0, row0);
    int trow;
    if (gather_owns_row(tid, trow)) {
      const float qn = M.topq[trow * ldq];
      const float y = M.dn[trow] != 0.f ? M.rv[trow] : (P.nstep > 0 ? M.nrew[trow] : M.rv[trow]) + gamma * qn;
      M.yv[trow] = y;
      if (P.qt_out && trow < nrows) P.qt_out[(size_t)agent * B + row0 + trow] = y;
    }
  } else if (step == 3) {                // EuclideanLoss over the whole minibatch
    if (tid < DDRL_MB) {
      const float df = tid < nrows ? M.topq[tid * ldq] - M.yv[tid] : 0.f;
      float sq = df * df;
      for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
      if (tid == 0) P.loss_part[(size_t)agent * P.maxchunks + chunk] = sq;
      M.topq[tid * ldq] = df / (float)B;
    }
  } else if (step == 4) {                // DPG seed (:424-428), NOT / B
    if (tid < DDRL_MB) M.topq[tid * ldq] = tid < nrows ? -1.0f : 0.f;
  } else if (step >= 10 && step <= 12) { // the bookkeeping of a net's update (the optimiser itself runs inside the dW kernels)
    const int net = step - 10;
    const NetState &st = net == 2 ? P.sq2 : net ? P.sq : P.sa;
    if (chunk == 0 && tid == 0) {
      st.iter[agent] += 1;
      if (st.fisher) st.fisher_nbr[agent] = st.fisher_nbr[agent] * P.fisher_beta + (float)B;
      if (net >= 1) {
        float sum = 0.f;
        for (int ch = 0; ch * DDRL_MB < B; ++ch) sum += (net == 1 ? P.loss_part : P.loss_part2)[(size_t)agent * P.maxchunks + ch];
        (net == 1 ? P.loss_out : P.loss2_out)[agent] = sum / (float)B * 0.5f;
      }
    }
  } else if (step == 6) {                // TD3: a' = clip(a' + eps, -1, 1) between the two gates (TD3Ag.hpp:307-309)
    const float *noise = P.noise + ((size_t)agent * B + row0) * A;
    for (int e = tid; e < DDRL_MB * A; e += DDRL_NTHREADS) {
      const int row = e / A, col = e - row * A;
      float a = M.topa[row * lda + col];
      const GateMode &gm = P.gm;
      if (agate) a *= ddrl_gate_factor(agate[col], gm.ac_bern, gm.ac_scale, gm.seed, gm.draw, agent, GS_ACTOR_T_OUT, row0 + row, col);
      if (row < nrows) a = fminf(fmaxf(a + noise[row * A + col], -1.f), 1.f);
      if (bern) M.topa[row * lda + col] = a;            // a' itself: the second target critic gates it with its own mask
      if (agate) a *= ddrl_gate_factor(agate[col], gm.ac_bern, gm.ac_scale, gm.seed, gm.draw, agent, GS_CRITIC_T_A, row0 + row, col);
      M.xc[row * ldxc + S + col] = a;
    }
  } else if (step == 7) {                // TD3: Q1'(s', a') aside while the second target critic runs (:313-314)
    if (tid < DDRL_MB) M.nrew[tid] = M.topq[tid * ldq];
    if (bern) {                          // ... which has layer instances of its own (kernels_td3.hip)
      td3_regate_xc(P, S, A, ldxc, agent, idx, nrows, head, true, M, tid, GS_CRITIC2_T_S, 0, row0);
      for (int e = tid; e < DDRL_MB * A; e += DDRL_NTHREADS) {
        const int row = e / A, col = e - row * A;
        float a = M.topa[row * lda + col];
        if (agate) a *= ddrl_gate_factor(agate[col], P.gm.ac_bern, P.gm.ac_scale, P.gm.seed, P.gm.draw, agent, GS_CRITIC2_T_A, row0 + row, col);
        M.xc[row * ldxc + S + col] = a;
      }
    }
  } else if (step == 14 || step == 15) { // TD3, Bernoulli gates: the critic's input tile again with qnn2's (14) / qnn's (15) masks
    if (bern) td3_regate_xc(P, S, A, ldxc, agent, idx, nrows, head, false, M, tid, step == 14 ? GS_CRITIC2_S : GS_CRITIC_S,
                            step == 14 ? GS_CRITIC2_A_REPLAY : GS_CRITIC_A_REPLAY, row0);
  } else if (step == 8) {                // TD3: [s|a] tiles + y = r + gamma min(Q1', Q2') (:317-326)
    gather_tile(P, S, A, ldxc, ldxa, agent, idx, nrows, head, false, M, tid, 0, row0);
    int trow;
    if (gather_owns_row(tid, trow)) {
      const float qn = fminf(M.nrew[trow], M.topq[trow * ldq]);
      const float y = M.dn[trow] != 0.f ? M.rv[trow] : M.rv[trow] + gamma * qn;
      M.yv[trow] = y;
      if (P.qt_out && trow < nrows) P.qt_out[(size_t)agent * B + row0 + trow] = y;
    }
  } else if (step == 9) {                // TD3: the second critic's EuclideanLoss
    if (tid < DDRL_MB) {
      const float df = tid < nrows ? M.topq[tid * ldq] - M.yv[tid] : 0.f;
      float sq = df * df;
      for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
      if (tid == 0) P.loss_part2[(size_t)agent * P.maxchunks + chunk] = sq;
      M.topq[tid * ldq] = df / (float)B;
    }
  } else {                               // inverting gradients (:432-450) and the actor's output layer
    const int ldda = ddrl_ldpad(A), out_act = P.a.out_act();
    float *dact_out = P.dact_out ? P.dact_out + ((size_t)agent * B + row0) * A : nullptr;
    for (int e = tid; e < DDRL_MB * A; e += DDRL_NTHREADS) {
      const int row = e / A, h = e - row * A;
      float out = M.topa[row * lda + h];
      float gin = 1.f, gt = 1.f;          // the masks of the forward passes these gradients belong to
      if (agate) {
        const GateMode &gm = P.gm;
        gin = ddrl_gate_factor(agate[h], gm.ac_bern, gm.ac_scale, gm.seed, gm.draw, agent, GS_CRITIC_A, row0 + row, h);
        gt = ddrl_gate_factor(agate[h], gm.ac_bern, gm.ac_scale, gm.seed, gm.draw, agent, GS_ACTOR_OUT, row0 + row, h);
      }
      float diff = W.da[row * ldda + h] * gin;
      if (P.inverting) {
        if (diff < 0.f) diff *= (1.0f - out) / 2.0f;
        else if (diff > 0.f) diff *= (out + 1.0f) / 2.0f;
      }
      if (dact_out && row < nrows) dact_out[row * A + h] = diff;
      diff *= gt;
      if (gt != 0.f) out /= gt;
      M.topa[row * lda + h] = act_bwd(out_act, out, diff);
    }
  }
}

extern "C" size_t ddpg_wide_block_floats(const DdpgParams *P) { return wide_block_floats(*P); }

#define WIDE_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)

// td3 == 0: DeepQNAg's update.  td3 != 0: TD3Ag's (TD3Ag.hpp:284-380) -- smoothed target action, two target
// critics, both critics' learn_batch, and (any_actor: some agent is on a policy step) the actor phase
// launched for those agents only.  The launches cover agents [a0, a0 + G) and go to stream st.
static hipError_t wide_launch_range(const DdpgParams &P, int td3, int any_actor, int a0, int G, hipStream_t st) {
  const int C = (P.B + DDRL_MB - 1) / DDRL_MB;
  int policy = 0;                                           // set for the launches of TD3's actor phase
  auto step = [&](int s) { hipLaunchKernelGGL(wide_step_kernel, dim3(C, G), dim3(DDRL_NTHREADS), 0, st, P, s, policy, a0); };
  auto fwd = [&](int net, int target) {
    const NetDesc &nd = net ? P.q : P.a;
    for (int l = 0; l < nd.L_; ++l) {
      const WideOp op = {net, target, l, 0, policy, a0, G};
      hipLaunchKernelGGL(wide_fwd_kernel, dim3(xcd_grid((nd.ld_[l + 1] + 63) / 64, C, G)), dim3(256), 0, st, P, op);
    }
  };
  auto dx = [&](int net, int l, int mode) {
    const NetDesc &nd = net ? P.q : P.a;
    const WideOp op = {net, 0, l, mode, policy, a0, G};
    hipLaunchKernelGGL(wide_dx_kernel, dim3(xcd_grid((nd.kp_[l] + 63) / 64, C, G)), dim3(256), 0, st, P, op);
  };
  auto dw = [&](int net, int l) {          // all chunks of the minibatch in one workgroup: grid.y = 1
    const NetDesc &nd = net ? P.q : P.a;
    const WideOp op = {net, 0, l, 0, policy, a0, G};
    const dim3 grid((nd.d_[l + 1] + 63) / 64, 1, G);
    const bool ewc = (net == 2 ? P.sq2 : net ? P.sq : P.sa).fisher != nullptr;
    if (ewc) hipLaunchKernelGGL((wide_dw_kernel<true>), grid, dim3(256), 0, st, P, op);
    else hipLaunchKernelGGL((wide_dw_kernel<false>), grid, dim3(256), 0, st, P, op);
  };
  // after a net's backward pass: Solver::iter, the Fisher count, the loss (the optimiser ran inside the dW kernels)
  auto finish = [&](int net) -> hipError_t {
    step(10 + net);
    return hipGetLastError();
  };
  hipError_t e;
  if (td3) {
    // y = r + gamma min(Q1', Q2')(s', clip(pi'(s') + eps)), then both critics learn it (:304-330)
    step(0); fwd(0, 1); step(6); fwd(1, 1); step(7); fwd(2, 1); step(8);
    const bool bern = (P.sgate || P.agate) && (P.gm.st_bern || P.gm.ac_bern);
    for (int k = 1; k <= 2; ++k) {
      if (k == 2 && bern) step(14);
      fwd(k, 0); step(k == 1 ? 3 : 9);
      for (int l = P.q.L_ - 1; l >= 0; --l) { if (l > 0) dx(k, l, 0); dw(k, l); }
      WIDE_CHECK();
      if ((e = finish(k)) != hipSuccess) return e;
    }
    if (!any_actor) return hipSuccess;
    policy = 1;                                             // the delayed actor update through the first critic (:333-363)
    if (bern) step(15);
    fwd(0, 0); step(13); fwd(1, 0); step(4);
    for (int l = P.q.L_ - 1; l >= 1; --l) dx(1, l, 0);
    dx(1, 0, 1);
    step(5);
    for (int l = P.a.L_ - 1; l >= 0; --l) { if (l > 0) dx(0, l, 0); dw(0, l); }
    WIDE_CHECK();
    return finish(0);
  }
  // critic: y = r + gamma Q'(s', pi'(s')), then learn_batch (DeepQNAg.hpp:379-408)
  step(0); fwd(0, 1); step(1); fwd(1, 1); step(2); fwd(1, 0); step(3);
  for (int l = P.q.L_ - 1; l >= 0; --l) { if (l > 0) dx(1, l, 0); dw(1, l); }
  WIDE_CHECK();
  if ((e = finish(1)) != hipSuccess) return e;
  // actor: dQ/da through the NEW critic, inverting gradients, actor backward (:411-459)
  fwd(0, 0); step(13); fwd(1, 0); step(4);
  for (int l = P.q.L_ - 1; l >= 1; --l) dx(1, l, 0);
  dx(1, 0, 1);
  step(5);
  for (int l = P.a.L_ - 1; l >= 0; --l) { if (l > 0) dx(0, l, 0); dw(0, l); }
  WIDE_CHECK();
  return finish(0);
}

// The update is a chain of ~45 launches that alternate between matrix-core-bound kernels (forward, input gradients) and
// HBM-bound ones (dW + optimiser: 52 B of state per parameter).  Agents are independent, so a large population is cut in two
// parts (WIDE_PARTS_DEFAULT) whose chains run on streams of their own: while one half streams its optimiser state the other half's products run, instead
// of the whole chip waiting on one resource at a time.  The second stream and the two events of the fork / join are made
// once per device and kept; results are the same launches over the same agents (bit-identical).
#define WIDE_SPLIT_MIN_AGENTS 128
#define WIDE_MAX_PARTS 4
#define WIDE_PARTS_DEFAULT 2
static hipError_t wide_launch(const DdpgParams *Pp, int td3, int any_actor, hipStream_t st) {
  const DdpgParams &P = *Pp;
  static const int parts_env = getenv("DDRL_WIDE_PARTS") ? atoi(getenv("DDRL_WIDE_PARTS")) : 0;      // (measurement switch; 1: no split)
  const int parts = parts_env >= 1 && parts_env <= WIDE_MAX_PARTS ? parts_env : WIDE_PARTS_DEFAULT;
  if (P.G < WIDE_SPLIT_MIN_AGENTS || parts == 1) return wide_launch_range(P, td3, any_actor, 0, P.G, st);
  struct Aux { hipStream_t s[WIDE_MAX_PARTS - 1] = {}; hipEvent_t fork = nullptr, join[WIDE_MAX_PARTS - 1] = {}; bool made = false; };
  static Aux aux[64];
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return wide_launch_range(P, td3, any_actor, 0, P.G, st);
  // (the streams and events are per device, not per handle: handles driven from several host threads take turns here)
  static std::mutex turn;
  std::lock_guard<std::mutex> hold(turn);
  Aux &a = aux[dev];
  if (!a.made) {
    if ((e = hipEventCreateWithFlags(&a.fork, hipEventDisableTiming)) != hipSuccess) return e;
    for (int k = 0; k < WIDE_MAX_PARTS - 1; ++k) {
      if ((e = hipStreamCreateWithFlags(&a.s[k], hipStreamNonBlocking)) != hipSuccess) return e;
      if ((e = hipEventCreateWithFlags(&a.join[k], hipEventDisableTiming)) != hipSuccess) return e;
    }
    a.made = true;
  }
  if ((e = hipEventRecord(a.fork, st)) != hipSuccess) return e;
  for (int k = 0; k < parts - 1; ++k)
    if ((e = hipStreamWaitEvent(a.s[k], a.fork, 0)) != hipSuccess) return e;
  hipError_t first = hipSuccess;
  for (int k = 0; k < parts; ++k) {
    const int a0 = (int)((long long)P.G * k / parts), a1 = (int)((long long)P.G * (k + 1) / parts);
    const hipError_t ek = wide_launch_range(P, td3, any_actor, a0, a1 - a0, k == 0 ? st : a.s[k - 1]);
    if (first == hipSuccess) first = ek;
  }
  for (int k = 0; k < parts - 1; ++k) {               // joined whatever the launches said: st never runs ahead of a part
    if ((e = hipEventRecord(a.join[k], a.s[k])) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(st, a.join[k], 0)) != hipSuccess) return e;
  }
  return first;
}

extern "C" hipError_t ddpg_update_wide_launch(const DdpgParams *P, hipStream_t st) { return wide_launch(P, 0, 0, st); }
extern "C" hipError_t td3_update_wide_launch(const DdpgParams *P, int any_actor, hipStream_t st) { return wide_launch(P, 1, any_actor, st); }
