// pop_internal.h -- shared between the host side (pop.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DDRL_MAXL 4          // InnerProduct layers per net = hidden (<=3) + output
#define DDRL_MB 64           // rows per workgroup batch tile (4 MFMA m-tiles of 16)
#define DDRL_NWAVES 4        // 256-thread workgroups: one wave per SIMD of the CU
#define DDRL_NTHREADS (DDRL_NWAVES * 64)
#define DDRL_MAXW 128        // max layer width handled by the fused (LDS-resident) kernels

typedef float f32x4 __attribute__((ext_vector_type(4)));

// One MLP's shape and its padded device layout.  Per agent the net is one block of `stride`
// floats: layer l weights at offw[l], row-major [d[l+1] x kp[l]] with kp = roundup(d[l], 4) so that
// every row starts 16-byte aligned (pad columns hold 0 and are never updated), bias at offb[l].
struct NetDesc {
  int L;
  int d[DDRL_MAXL + 1];
  int kp[DDRL_MAXL];
  int ld[DDRL_MAXL + 1];    // LDS row stride of the activation of width d[i] (see ldpad())
  int offw[DDRL_MAXL];
  int offb[DDRL_MAXL];
  int stride;               // floats per agent (multiple of 64 => 256-B aligned blocks)
  int np;                   // logical (flat, reference-order) parameter count
  int hidden_act, out_act;
};

// Per-net optimiser / EWC state pointers (device), each [G x stride] unless noted
struct NetState {
  float *w, *m, *v;         // weights, Adam first/second moment
  float *wt;                // target net weights (soft update destination), may be null
  float *fisher;            // running Fisher (DODevMLP.hpp:635-658), null if EWC off for this net
  float *fhat, *wstar;      // normalised Fisher + anchor (previous_fisher, best_param_previous_task)
  float *best;              // best_param snapshot
  int   *iter;              // [G] Solver::iter
  float *fisher_nbr;        // [G]
  float *ewc_coef;          // [G] ewc * decay_multiplier if penalty live else 0
  const float *lr;          // [G]
  const float *decay;       // [G] L2 weight decay (<0 or 0: off), null if never used
};

struct DdpgParams {
  NetDesc a, q;
  NetState sa, sq;
  int G, S, A, B, cap, rowlen;
  int inverting;
  int ldxc, ldxa;           // LDS strides of the critic / actor input tiles
  int ldh;                  // LDS row stride reserved for the two hidden-activation buffers
  float beta1, beta2, omb1, omb2, eps;
  float fisher_beta;
  const float *rep;         // [G x cap x rowlen]  rows: s, a, s2, r, goal
  const int *rep_len, *rep_head;
  const int *idx;           // [G x B]
  const float *tau, *omtau, *gamma;   // [G]
  const uint8_t *mask;      // [G] or null
  float *loss_out;          // [G]
  float *qt_out;            // [G x B] or null
  float *dact_out;          // [G x B x A] or null
};

__host__ __device__ inline int ddrl_roundup(int x, int m) { return (x + m - 1) / m * m; }
// LDS row stride for a width-n activation: multiple of 4 floats (16-B aligned float4 rows) and
// == 4 (mod 8), so that rows 4 apart land 16 banks apart for the column-walking ds_read_b32 of the
// dW products.
__host__ __device__ inline int ddrl_ldpad(int n) {
  int r = ddrl_roundup(n, 4);
  while ((r & 7) != 4) r += 4;
  return r;
}
