// mlp_device.h -- workgroup-level fp32 MFMA building blocks shared by the fused DDPG / NFAC /
// batched-inference kernels (gfx950 only).
//
// One 256-thread workgroup (4 waves, one per SIMD) owns one agent.  A batch tile of MB = 64 rows
// lives in LDS; weights are read straight from global memory (L2/MALL resident between passes)
// with one 16-byte load per lane per 4 MFMAs; products run on v_mfma_f32_16x16x4_f32, which is
// bit-for-bit an fp32 fma chain (cdna_hip_programming.md section 3), so results do not depend on
// tile shapes beyond the order of the k-sum.
//
// Lane roles for v_mfma_f32_16x16x4_f32 (lane l, c = l & 15, g = l >> 4):
//   A operand: A[i = c][kk = g]     B operand: B[kk = g][j = c]     D reg r: D[i = 4g + r][j = c]
// A 16-deep k-chunk is issued as 4 MFMAs; in MFMA jj the kk slot g carries k = k0 + 4g + jj, so
// each lane feeds the 4 MFMAs from ONE float4 of its operand row (k0 + 4g .. k0 + 4g + 3).
#pragma once
#include "pop_internal.h"

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Caffe ReLULayer (negative_slope 0.01 for hidden_layer_type 1, MLP.hpp:888-897), TanHLayer
__device__ __forceinline__ float act_fwd(int t, float x) {
  switch (t) {
    case 1: return x > 0.f ? x : 0.01f * x;
    case 2: return tanhf(x);
    case 3: return x > 0.f ? x : 0.f;
    default: return x;
  }
}
// in-place layers (top == bottom, MLP.hpp:799-803): the derivative is taken from the OUTPUT y
__device__ __forceinline__ float act_bwd(int t, float y, float dy) {
  switch (t) {
    case 1: return y > 0.f ? dy : 0.01f * dy;
    case 2: return dy * (1.f - y * y);
    case 3: return y > 0.f ? dy : 0.f;
    default: return dy;
  }
}

// Y[64 x N] = act(X[64 x K] W^T + b).  Xs: LDS, row stride ldx, columns [K, kp) hold zeros.
// W: global, [N x kp] row-major.  Ys: LDS, row stride ldy; columns [N, roundup(N,4)) are zeroed.
// Wave w owns output column tiles nt = w, w+4, ... and all four row tiles, so W is read once.
__device__ __forceinline__ void fwd_layer(const float *__restrict__ Xs, int ldx, int kp,
                                          const float *__restrict__ W, const float *__restrict__ bias,
                                          int N, int act, float *__restrict__ Ys, int ldy,
                                          int wave, int lane) {
  const int g = lane >> 4, c = lane & 15;
  const int NT = (N + 15) >> 4;
  for (int nt = wave; nt < NT; nt += DDRL_NWAVES) {
    f32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int n = nt * 16 + c;
    const float *wrow = W + (size_t)(n < N ? n : N - 1) * kp;
    for (int k0 = 0; k0 < kp; k0 += 16) {
      const int k = k0 + 4 * g;
      const bool in = k < kp;
      f32x4 b = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (in) b = *(const f32x4 *)(wrow + k);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (in) a = *(const f32x4 *)(Xs + (mt * 16 + c) * ldx + k);
        acc[mt] = mfma16(a[0], b[0], acc[mt]);
        acc[mt] = mfma16(a[1], b[1], acc[mt]);
        acc[mt] = mfma16(a[2], b[2], acc[mt]);
        acc[mt] = mfma16(a[3], b[3], acc[mt]);
      }
    }
    const float bv = n < N ? bias[n] : 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = mt * 16 + 4 * g + r;
        if (n < N) Ys[row * ldy + n] = act_fwd(act, acc[mt][r] + bv);
        else if (n < ldy) Ys[row * ldy + n] = 0.f;
      }
  }
}

// dX[64 x cols] = dY[64 x N] W[N x kp], for column tiles kt in [kt0, kt1); wave w owns tiles
// kt0 + w + 4 i, i < MAXKT, results stay in registers (acc[i][mt]).  dYs: LDS, columns [N, roundup
// (N,4)) must hold zeros.
template <int MAXKT>
__device__ __forceinline__ void bwd_dx(const float *__restrict__ dYs, int ldy, int N,
                                       const float *__restrict__ W, int kp, int kt0, int kt1,
                                       f32x4 (&acc)[MAXKT][4], int wave, int lane) {
  const int g = lane >> 4, c = lane & 15;
  const int n4 = ddrl_roundup(N, 4);
#pragma unroll
  for (int i = 0; i < MAXKT; ++i) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[i][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int kt = kt0 + wave + DDRL_NWAVES * i;
    if (kt < kt1) {
      const int col = kt * 16 + c;
      const bool cin = col < kp;
      for (int n0 = 0; n0 < N; n0 += 16) {
        const int nn = n0 + 4 * g;
        float b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = (cin && nn + j < N) ? W[(size_t)(nn + j) * kp + col] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (nn < n4) a = *(const f32x4 *)(dYs + (mt * 16 + c) * ldy + nn);
          acc[i][mt] = mfma16(a[0], b[0], acc[i][mt]);
          acc[i][mt] = mfma16(a[1], b[1], acc[i][mt]);
          acc[i][mt] = mfma16(a[2], b[2], acc[i][mt]);
          acc[i][mt] = mfma16(a[3], b[3], acc[i][mt]);
        }
      }
    }
  }
}

// Everything the fused optimiser epilogue needs for one net of one agent (pointers already
// offset to the agent's block).
struct OptCtx {
  float *w, *m, *v, *wt, *fisher;
  const float *fhat, *wstar;
  float lr_t, tau, omtau, decay, ewc_coef, fisher_beta, nsample;
  float beta1, beta2, omb1, omb2, eps;
};

// One parameter: Fisher EMA (DODevMLP.hpp:651), EWC penalty (DODevMLP.hpp:702), L2 decay
// (caffe SGDSolver::Regularize), Adam (caffe AdamSolver::ComputeUpdateValue + Net::Update) and the
// target net's soft update (MLP.hpp:449-450) in a single read-modify-write of each array.
__device__ __forceinline__ void opt_step(const OptCtx &o, int i, float grad) {
  float w = o.w[i];
  if (o.fisher) o.fisher[i] = o.fisher[i] * o.fisher_beta + (grad * grad) * o.nsample;
  if (o.ewc_coef != 0.f) grad += o.ewc_coef * o.fhat[i] * (w - o.wstar[i]);
  if (o.decay > 0.f) grad += o.decay * w;
  const float m = o.omb1 * grad + o.beta1 * o.m[i];
  const float v = o.omb2 * (grad * grad) + o.beta2 * o.v[i];
  o.m[i] = m;
  o.v[i] = v;
  w -= o.lr_t * (m / (sqrtf(v) + o.eps));
  o.w[i] = w;
  if (o.wt) o.wt[i] = o.tau * w + o.omtau * o.wt[i];
}

// dW[N x K] = dY^T X  (+ db = dY^T 1), each element handed to opt_step() straight from the MFMA
// accumulator.  dYs [64 x N] and Xs [64 x K] in LDS; rows >= batch must be zero in dYs.
// Output tiles (nt, kt) are dealt to the waves in contiguous runs (balanced to within one tile).
__device__ __forceinline__ void bwd_dw_opt(const float *__restrict__ dYs, int ldy, int N,
                                           const float *__restrict__ Xs, int ldx, int K, int kp,
                                           int offw, int offb, const OptCtx &o,
                                           int wave, int lane, int tid) {
  const int g = lane >> 4, c = lane & 15;
  const int NT = (N + 15) >> 4, KT = (K + 15) >> 4;
  const int total = NT * KT;
  const int per = (total + DDRL_NWAVES - 1) / DDRL_NWAVES;
  const int p1 = min(total, (wave + 1) * per);
  for (int p = wave * per; p < p1; ++p) {
    const int nt = p / KT, kt = p - nt * KT;
    const int an = nt * 16 + c, bk = kt * 16 + c;
    const bool ain = an < N, bin = bk < K;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mc = 0; mc < 4; ++mc)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = mc * 16 + 4 * g + j;
        const float a = ain ? dYs[row * ldy + an] : 0.f;
        const float b = bin ? Xs[row * ldx + bk] : 0.f;
        acc = mfma16(a, b, acc);
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = nt * 16 + 4 * g + r;
      if (n < N && bk < K) opt_step(o, offw + n * kp + bk, acc[r]);
    }
  }
  // bias gradient: column sums of dY, ascending row order
  for (int n = tid; n < N; n += DDRL_NTHREADS) {
    float s = 0.f;
    for (int row = 0; row < DDRL_MB; ++row) s += dYs[row * ldy + n];
    opt_step(o, offb + n, s);
  }
}
