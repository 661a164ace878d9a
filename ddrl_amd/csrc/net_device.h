// net_device.h -- whole-net forward / backward passes over one 64-row LDS tile, built from the
// layer blocks of mlp_device.h.  Shared by the fused DDPG kernel and the NFAC kernels.
#pragma once
#include "mlp_device.h"

// dW / db of one layer for one 64-row chunk, written (not accumulated) to gout: the chunk's partial
// gradient, reduced over chunks in a fixed order by the apply kernel (deterministic, no atomics)
__device__ __forceinline__ void bwd_dw_store(const float *__restrict__ dYs, int ldy, int N,
                                             const float *__restrict__ Xs, int ldx, int kp,
                                             int offw, int offb, float *__restrict__ gout,
                                             int wave, int lane, int tid) {
  const int g = lane >> 4, c = lane & 15;
  const int NT = (N + 15) >> 4, KT = (kp + 15) >> 4;
  const int total = NT * KT;
  for (int p = wave; p < total; p += DDRL_NWAVES) {
    bool ok;
    const int i = dw_elem(p, total, KT, N, kp, offw, c, g, ok);
    const f32x4 acc = dw_tile(dYs, ldy, N, Xs, ldx, kp, p, total, KT, c, g);
    if (ok) *(f32x4 *)(gout + i) = acc;
  }
  for (int n = tid; n < N; n += DDRL_NTHREADS) {
    float s = 0.f;
    for (int row = 0; row < DDRL_MB; ++row) s += dYs[row * ldy + n];
    gout[offb + n] = s;
  }
}

// forward through layers [0, nl) of a net; hidden activation l goes to hbase + (l-1)*hstride, the
// output of the last InnerProduct (when nl == L) to top
template <class ND>
__device__ __forceinline__ void fwd_net(const ND nd, int nl, const float *__restrict__ w,
                                        const float *X, int ldx, float *hbase, int hstride, float *top, int ldtop,
                                        int wave, int lane) {
  const float *in = X;
  int ldi = ldx;
  BFrag b0;
  bool have = false;
#pragma unroll
  for (int l = 0; l < nd.L(); ++l) {
    if (l < nl) {
      const bool last = l == nd.L() - 1;
      const bool more = l + 1 < nl;
      const int ln = more ? l + 1 : l;       // keeps the accessors in range when there is no next layer
      float *out = last ? top : hbase + l * hstride;
      const int ldo = last ? ldtop : nd.ld(l + 1);
      fwd_layer(in, ldi, nd.kp(l), w + nd.offw(l), w + nd.offb(l), nd.d(l + 1),
                last ? nd.out_act() : nd.hidden_act(), out, ldo, wave, lane, b0, have,
                more ? w + nd.offw(ln) : nullptr, more ? nd.kp(ln) : 0, more ? nd.d(ln + 1) : 0);
      __syncthreads();
      in = out;
      ldi = ldo;
    }
  }
}

// Backward through a whole net from dZ_L held in `top` (diff w.r.t. the last InnerProduct's
// output).  DW_OPT: parameter gradients go through the fused optimiser epilogue (with a target twin).
// DW_STORE: they are written to gout (one agent-net image, same layout as the weights) for a later
// reduction over batch chunks.  nc > 0: the input gradient columns [c0, c0 + nc) are handed to
// dx0_store(row, col - c0, value).
enum { DW_NONE = 0, DW_OPT = 1, DW_STORE = 2 };
template <int TRAIN, bool EWC, class ND, class DX0>
__device__ __forceinline__ void bwd_net(const ND nd, const float *__restrict__ w, const OptCtx opt, float *gout,
                                        const float *X, int ldx, float *hbase, int hstride, float *top, int ldtop,
                                        int c0, int nc, DX0 dx0_store, int wave, int lane, int tid) {
  const int g = lane >> 4, c = lane & 15;
  float *dZ = top;
  int ldz = ldtop;
#pragma unroll
  for (int ll = 0; ll < nd.L(); ++ll) {
    const int l = nd.L() - 1 - ll;
    const int K = nd.d(l), N = nd.d(l + 1), kp = nd.kp(l);
    const float *W = w + nd.offw(l);
    float *in = l == 0 ? const_cast<float *>(X) : hbase + (l - 1) * hstride;
    const int ldi = l == 0 ? ldx : nd.ld(l);
    const bool need_dx = l > 0 || nc > 0;
    f32x4 acc[DDRL_MAXKT][4];
    int kt0 = 0, kt1 = (K + 15) >> 4;
    if (l == 0) { kt0 = c0 >> 4; kt1 = (c0 + nc + 15) >> 4; }
    if (need_dx) bwd_dx<DDRL_MAXKT>(dZ, ldz, N, W, kp, kt0, kt1, acc, wave, lane);
    if (TRAIN == DW_OPT) {
      __syncthreads();   // every wave has read W_l before any wave's epilogue rewrites it
      bwd_dw_opt<EWC, true>(dZ, ldz, N, in, ldi, kp, nd.offw(l), nd.offb(l), opt, wave, lane, tid);
    }
    if (TRAIN == DW_STORE) bwd_dw_store(dZ, ldz, N, in, ldi, kp, nd.offw(l), nd.offb(l), gout, wave, lane, tid);
    __syncthreads();     // every wave has read H_l before it is overwritten by dZ_l
    if (need_dx) {
#pragma unroll
      for (int i = 0; i < DDRL_MAXKT; ++i) {
        const int kt = kt0 + wave + DDRL_NWAVES * i;
        if (kt < kt1) {
          const int col = kt * 16 + c;
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = mt * 16 + 4 * g + r;
              if (l > 0) {
                if (col < K) in[row * ldi + col] = act_bwd(nd.hidden_act(), in[row * ldi + col], acc[i][mt][r]);
              } else if (col >= c0 && col < c0 + nc) {
                dx0_store(row, col - c0, acc[i][mt][r]);
              }
            }
        }
      }
      __syncthreads();
    }
    dZ = in;
    ldz = ldi;
  }
}

// run-time descriptors live in the kernel arguments; compile-time ones are empty tag objects
template <class ND> struct Pick { static __device__ __forceinline__ ND get(const NetDesc &) { return ND(); } };
template <> struct Pick<NetDesc> { static __device__ __forceinline__ const NetDesc &get(const NetDesc &n) { return n; } };

