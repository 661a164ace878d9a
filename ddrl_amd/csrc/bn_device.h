// bn_device.h -- BatchNorm (+ Scale) towers on the 64-row LDS tiles of net_device.h.
//
// Reference: MLP::BatchNormTower / BatchNormLayer / ScaleLayer (common/include/nn/MLP.hpp:978-1021), placed by Tower()
// (:756-758) and the constructors (:130-141, 220, 238-240); arithmetic = BVLC Caffe 1.0 BatchNormLayer (eps 1e-5,
// moving_average_fraction 0.999) and ScaleLayer.  Every tower works IN PLACE on the blob the preceding activation wrote,
// so that activation's backward sees the tower's output (Caffe's in-place semantics, restated as is).
//
// A tower couples the rows of the minibatch: its passes are column-parallel (2 to 16 threads per channel) and
// sit between workgroup barriers, unlike the row-parallel InnerProduct passes.  Only rows < B take part; rows >= B are
// kept at zero so that they stay out of the dW products.
#pragma once
#include "net_device.h"

#define DDRL_BN_EPS 1e-5f
#define DDRL_BN_MAF 0.999f
enum { BN_TRAIN = 0,       // batch statistics, running sums updated (a TRAIN-phase net's Forward)
       BN_RECOMPUTE = 1,   // batch statistics, running sums untouched (activations rebuilt for a backward pass)
       BN_TEST = 2 };      // use_global_stats (a TEST-phase copy, MLP.hpp:284-338)

// x_norm copies (the towers' outputs before Scale) of one net: input tile, hidden tiles (same stride as the hidden
// activations), output tile; den: sqrt(var + eps) per tower and channel [(DDRL_MAXL + 1) x DDRL_MAXW].  Null members: not kept
// (TEST-phase evaluation without a backward pass).
struct BnBufs { float *x0, *h, *top, *den; };

// threads per channel of a column pass: the largest power of two (2..16) that keeps every channel of the tower in one sweep of
// the workgroup; a channel's threads are consecutive lanes of one wave, each walks every tpc-th row
__device__ __forceinline__ int bn_threads_per_channel(int C) {
  int tpc = 16;
  while (tpc > 2 && C * tpc > DDRL_NTHREADS) tpc >>= 1;
  return tpc;
}
__device__ __forceinline__ float bn_group_sum(float v, int tpc) {
  for (int o = 1; o < tpc; o <<= 1) v += __shfl_xor(v, o);
  return v;
}

// tower r over buf [64 x ld] (columns [0, C)), between barriers supplied by the caller
__device__ __forceinline__ void bn_forward_cols(const NetDesc &nd, float *__restrict__ wblk, int r, int mode, int B,
                                                float *buf, int ld, float *xn, float *den, int tid) {
  const int C = nd.bn_c(r), C4 = nd.bn_c4(r);
  float *Pm = wblk + nd.bn_off(r);
  const float *gam = nd.bn_scale() ? Pm + 2 * C4 + 4 : nullptr;
  const float *bet = nd.bn_bias() ? Pm + 3 * C4 + 4 : nullptr;
  const float factor = Pm[2 * C4];                       // read by every thread before thread 0 replaces it (barrier below)
  const float sf = factor == 0.f ? 0.f : 1.0f / factor;
  const int tpc = bn_threads_per_channel(C), sub = tid & (tpc - 1);
  const float invB = 1.0f / (float)B;
  for (int c = tid / tpc; c < C; c += DDRL_NTHREADS / tpc) {
    float mean, var;
    if (mode == BN_TEST) {
      mean = sf * Pm[c]; var = sf * Pm[C4 + c];
    } else {
      float s = 0.f;
      for (int i = sub; i < DDRL_MB; i += tpc) s += i < B ? buf[i * ld + c] : 0.f;
      s = bn_group_sum(s, tpc);
      mean = s * invB;
      float q = 0.f;
      for (int i = sub; i < DDRL_MB; i += tpc) { const float xc = buf[i * ld + c] - mean; q += i < B ? xc * xc : 0.f; }
      q = bn_group_sum(q, tpc);
      var = q * invB;
      if (mode == BN_TRAIN && sub == 0) {
        const float bc = B > 1 ? (float)B / (float)(B - 1) : 1.f;
        Pm[c] = mean + DDRL_BN_MAF * Pm[c];
        Pm[C4 + c] = bc * var + DDRL_BN_MAF * Pm[C4 + c];
      }
    }
    const float dn = sqrtf(var + DDRL_BN_EPS);
    if (den && sub == 0) den[r * DDRL_MAXW + c] = dn;
    const float g = gam ? gam[c] : 1.f, b = bet ? bet[c] : 0.f;
    for (int i = sub; i < DDRL_MB; i += tpc) {
      const float x = i < B ? (buf[i * ld + c] - mean) / dn : 0.f;
      if (xn) xn[i * ld + c] = x;
      buf[i * ld + c] = i < B ? (gam ? fmaf(x, g, b) : x) : 0.f;
    }
  }
  if (mode == BN_TRAIN) {
    __syncthreads();                                     // every thread holds the old factor
    if (tid == 0) Pm[2 * C4] = factor * DDRL_BN_MAF + 1.0f;
  }
}

// Scale's and the tower's backward over d [64 x ld] in place (columns [clo, chi) of [0, C)): d gamma = sum d x_norm,
// d beta = sum d into gout (null: discarded), d <- gamma d, then TRAIN d <- (d - mean(d) - mean(d x_norm) x_norm) / den,
// TEST d <- d / den; act >= 0: followed by the backward of the in-place activation that precedes the tower, fed the
// tower's OUTPUT (gamma x_norm + beta) as Caffe's in-place layering does.
__device__ __forceinline__ void bn_backward_cols(const NetDesc &nd, const float *__restrict__ wblk, float *gout, int r, bool test,
                                                 int B, float *d, int ld, const float *xn, const float *den, int act,
                                                 int clo, int chi, int tid) {
  const int C4 = nd.bn_c4(r);
  const float *Pm = wblk + nd.bn_off(r);
  const float *gam = nd.bn_scale() ? Pm + 2 * C4 + 4 : nullptr;
  const float *bet = nd.bn_bias() ? Pm + 3 * C4 + 4 : nullptr;
  const int tpc = bn_threads_per_channel(chi - clo), sub = tid & (tpc - 1);
  const float invB = 1.0f / (float)B;
  for (int c = clo + tid / tpc; c < chi; c += DDRL_NTHREADS / tpc) {
    float s1 = 0.f, s2 = 0.f;
    for (int i = sub; i < DDRL_MB; i += tpc) {
      const float di = i < B ? d[i * ld + c] : 0.f;
      s1 += di; s2 += di * xn[i * ld + c];
    }
    s1 = bn_group_sum(s1, tpc);
    s2 = bn_group_sum(s2, tpc);
    const float g = gam ? gam[c] : 1.f, b = bet ? bet[c] : 0.f;
    if (gout && sub == 0) {
      if (gam) gout[nd.bn_off(r) + 2 * C4 + 4 + c] = s2;
      if (bet) gout[nd.bn_off(r) + 3 * C4 + 4 + c] = s1;
    }
    const float m1 = g * s1 * invB, m2 = g * s2 * invB, dn = den[r * DDRL_MAXW + c];
    for (int i = sub; i < DDRL_MB; i += tpc) {
      const float x = xn[i * ld + c];
      float v = g * d[i * ld + c];
      v = test ? v / dn : (v - m1 - m2 * x) / dn;
      if (act >= 0) v = act_bwd(act, gam ? fmaf(x, g, b) : x, v);
      d[i * ld + c] = i < B ? v : 0.f;
    }
  }
}

// forward through layers [0, nl) with the towers of ranks [r_first, nl] in `mode`.
// wblk: the agent's parameter block (running sums are updated in it in BN_TRAIN).  Ends with a workgroup barrier.
__device__ __forceinline__ void fwd_net_bn(const NetDesc &nd, int nl, float *__restrict__ wblk, int mode, int B, int r_first,
                                           float *X, int ldx, float *hbase, int hstride, float *top, int ldtop,
                                           const BnBufs xb, int wave, int lane, int tid) {
  float *in = X;
  int ldi = ldx;
  const FwdNext none = {nullptr, nullptr, 4, 1};
  for (int l = 0; l < nl; ++l) {
    if (nd.bn(l) && l >= r_first) {
      __syncthreads();
      bn_forward_cols(nd, wblk, l, mode, B, in, ldi, l == 0 ? xb.x0 : (xb.h ? xb.h + (l - 1) * hstride : nullptr), xb.den, tid);
      __syncthreads();
    }
    const bool last = l == nd.L() - 1;
    float *out = last ? top : hbase + l * hstride;
    const int ldo = last ? ldtop : nd.ld(l + 1);
    BFrag b0, b1;
    fwd_layer(in, ldi, nd.kp(l), wblk + nd.offw(l), wblk + nd.offb(l), nd.d(l + 1), last ? nd.out_act() : nd.hidden_act(),
              out, ldo, wave, lane, b0, b1, false, none);
    wave_lds_fence();
    in = out;
    ldi = ldo;
  }
  // the tower on what the last of these layers wrote: the net's output (rank L), or the input of the first layer left out
  if (nd.bn(nl) && nl >= r_first) {
    __syncthreads();
    bn_forward_cols(nd, wblk, nl, mode, B, in, ldi, nl == nd.L() ? xb.top : (xb.h ? xb.h + (nl - 1) * hstride : nullptr), xb.den, tid);
  }
  __syncthreads();
}

// Backward through the whole net.  `top` holds, for every row < B (zeros below): the diff w.r.t. the net's final output
// when the net has an output tower (rank L), else dZ_L as bwd_net expects it (the caller applied the output activation's
// backward).  TRAIN == DW_STORE: parameter gradients (InnerProducts and Scales) are written to gout, an agent-net image.
// The input gradient columns [c0, c0 + nc) go to dx0_store(row, col - c0, value) after the pass's last barrier.
template <int TRAIN, class DX0>
__device__ __forceinline__ void bwd_net_bn(const NetDesc &nd, const float *__restrict__ wblk, float *gout, bool test, int B,
                                           float *X, int ldx, float *hbase, int hstride, float *top, int ldtop,
                                           const BnBufs xb, int c0, int nc, DX0 dx0_store, int wave, int lane, int tid) {
  const int g = lane >> 4, c = lane & 15;
  const int L = nd.L();
  if (nd.bn(L)) {
    bn_backward_cols(nd, wblk, TRAIN == DW_STORE ? gout : nullptr, L, test, B, top, ldtop, xb.top, xb.den, nd.out_act(), 0, nd.bn_c(L), tid);
    __syncthreads();
  }
  float *dZ = top;
  int ldz = ldtop;
  for (int ll = 0; ll < L; ++ll) {
    const int l = L - 1 - ll;
    const int K = nd.d(l), N = nd.d(l + 1), kp = nd.kp(l);
    const float *W = wblk + nd.offw(l);
    float *in = l == 0 ? X : hbase + (l - 1) * hstride;
    const int ldi = l == 0 ? ldx : nd.ld(l);
    const bool tower = nd.bn(l);
    const bool scale_grad = tower && nd.bn_scale() && TRAIN == DW_STORE;
    const bool need_dx = l > 0 || nc > 0 || scale_grad;
    int clo = 0, chi = K;                                  // the input-gradient columns wanted
    if (l == 0 && !scale_grad) { clo = c0; chi = c0 + nc; }
    f32x4 acc[DDRL_MAXKC];
    const int kt0 = clo >> 4, kt1 = (chi + 15) >> 4;
    if (need_dx) bwd_dx(dZ, ldz, N, W, kp, kt0, kt1, acc, wave, lane);
    if (TRAIN == DW_STORE) {
      __syncthreads();   // every wave's rows of dZ_l are final and every wave has read W_l
      bwd_dw_store(dZ, ldz, N, in, ldi, kp, nd.offw(l), nd.offb(l), gout, wave, lane, tid);
      __syncthreads();   // every wave has read H_{l-1} before its rows are overwritten
    }
    if (need_dx) {
#pragma unroll
      for (int i = 0; i < DDRL_MAXKC; ++i) {
        const int kt = kt0 + i;
        if (kt < kt1) {
          const int col = kt * 16 + c;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * wave + 4 * g + r;
            if (col >= clo && col < chi) {
              // under a tower: the raw gradient, finished by the tower's column pass below
              if (tower || l == 0) in[row * ldi + col] = acc[i][r];
              else in[row * ldi + col] = act_bwd(nd.hidden_act(), in[row * ldi + col], acc[i][r]);
            }
          }
        }
      }
      wave_lds_fence();
    }
    if (tower || (l == 0 && nc > 0)) __syncthreads();
    if (tower && need_dx) {
      const int lo = clo, hi = chi < nd.bn_c(l) ? chi : nd.bn_c(l);
      bn_backward_cols(nd, wblk, TRAIN == DW_STORE ? gout : nullptr, l, test, B, in, ldi,
                       l == 0 ? xb.x0 : xb.h + (l - 1) * hstride, xb.den, l > 0 ? nd.hidden_act() : -1, lo, hi, tid);
      __syncthreads();
    }
    if (l == 0 && nc > 0) {
      for (int e = tid; e < DDRL_MB * nc; e += DDRL_NTHREADS) {
        const int row = e / nc, col = e - row * nc;
        dx0_store(row, col, in[row * ldi + c0 + col]);
      }
    }
    dZ = in;
    ldz = ldi;
  }
}

// TEST-phase tower on one value (batch-1 inference): (x - mean) / sqrt(var + eps) gamma + beta with the running sums
__device__ __forceinline__ float bn_test_value(const NetDesc &nd, const float *__restrict__ wblk, int r, int c, float x) {
  const int C4 = nd.bn_c4(r);
  const float *Pm = wblk + nd.bn_off(r);
  const float f = Pm[2 * C4], sf = f == 0.f ? 0.f : 1.0f / f;
  float y = (x - sf * Pm[c]) / sqrtf(sf * Pm[C4 + c] + DDRL_BN_EPS);
  if (nd.bn_scale()) y = fmaf(y, Pm[2 * C4 + 4 + c], nd.bn_bias() ? Pm[3 * C4 + 4 + c] : 0.f);
  return y;
}
