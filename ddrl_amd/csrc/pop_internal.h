// pop_internal.h -- shared between the host side (pop.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>

// ---- DDRL_F64_DEBUG: the fp64 debug instantiation of the fused DDPG update (`make f64`) ----
// The SAME sources with every `float` of the device state, the LDS tiles and the arithmetic turned into `double`
// (the matrix products run as fp64 fma chains in the MFMA's own order, mlp_device.h).  It exists to separate
// rounding from algorithm: device-fp64 against the fp64 CPU restatement agrees to ~1e-14 over 100 free-running updates
// (tests/test_f64_debug_gpu.py), so what the fp32 product differs by is rounding.  Speed is irrelevant; only the
// one-tile DDPG path is built (libddrl_pop_f64.so: pop.hip, kernels_ddpg.hip, kernels_misc.hip).  In this build the
// C ABI's `float` arrays are `double` arrays too (ddrl_version() says "+f64"; ddrl_amd/lib.py follows).
// Every system header goes in BEFORE the macro: the standard library specialises on both types.
typedef float ddrl_f32;        // a real fp32 where an outside API wants one (hipEventElapsedTime)
#ifdef DDRL_F64_DEBUG
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>
#define float double
#define tanhf tanh
#define fmaxf fmax
#endif

#define DDRL_MAXL 4          // InnerProduct layers per net = hidden (<=3) + output
#define DDRL_MB 64           // rows per workgroup batch tile (4 MFMA m-tiles of 16)
#ifndef DDRL_NWAVES
#define DDRL_NWAVES 4        // 256-thread workgroups, one wave per SIMD: each wave carries 16 of the 64 batch rows (mlp_device.h);
#endif                       // two workgroups per CU (LDS, and 256 VGPRs per wave) => two waves per SIMD
#define DDRL_NTHREADS (DDRL_NWAVES * 64)
#define DDRL_W8_MAX_AGENTS 256  // populations up to this size run eight waves per agent (kernels_ddpg_w8.hip): one workgroup per CU at most
#define DDRL_PAIR_MIN_UPD 4     // shortest launch (updates per agent) that runs as agent pairs (kernels_ddpg_pair.hip: one idle quarter per launch)
#define DDRL_MAXUPD 64       // consecutive updates one launch may run (ddrl_pop_ddpg_update_n)
#define DDRL_MAXB 4096       // largest minibatch (chunks of DDRL_MB rows, kernels_ddpg_big.hip)
#define DDRL_WIDE_MAXW 1024  // widest layer of the layer-by-layer kernels (kernels_ddpg_wide.hip)
#define DDRL_MAXW 128        // max layer width handled by the fused (LDS-resident) kernels
#define DDRL_MAXKC (DDRL_MAXW / 16)   // 16-deep k-chunks in the widest supported layer
#define DDRL_MAXKT ((DDRL_MAXKC + DDRL_NWAVES - 1) / DDRL_NWAVES)   // dX column tiles per wave

typedef float f32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int ddrl_roundup(int x, int m) { return (x + m - 1) / m * m; }
// LDS row stride for a width-n activation: multiple of 4 floats (16-B aligned float4 rows) and
// == 4 (mod 8), so that rows 4 apart land 16 banks apart for the column-walking ds_read_b32 of the
// dW products.
__host__ __device__ constexpr int ddrl_ldpad(int n) {
  int r = ddrl_roundup(n, 4);
  while ((r & 7) != 4) r += 4;
  return r;
}

// Weight storage of one layer, [N x kp] with kp = roundup(K, 4): TILE-MAJOR.  Rows are grouped in
// strips of 16 (the last strip padded with zero rows); inside a strip the 16-column blocks follow
// each other, each block stored row-major [16 x kw] with kw = min(16, kp - 16 kt).  A full block
// is therefore 1 KiB of contiguous memory in exactly the lane order of a 16 x 16 MFMA accumulator
// tile (lane (c, g) <-> row c, floats 4g..4g+3), so the fused optimiser's seven loads and five
// stores per tile are each one contiguous 1-KiB wave access (measured with tools/membench.hip:
// 4.9 TB/s against 3.4 TB/s and 1.33x over-fetch for row-major 64-byte row segments), and the
// forward pass's weight fragments are contiguous too.  Pad rows/columns hold 0 and stay 0.
__host__ __device__ constexpr int ddrl_tile_kw(int kt, int kp) { return kp - kt * 16 < 16 ? kp - kt * 16 : 16; }
__host__ __device__ constexpr int ddrl_tile_base(int nt, int kt, int kp) { return nt * 16 * kp + kt * 256; }
__host__ __device__ constexpr int ddrl_widx(int n, int k, int kp) {
  return ddrl_tile_base(n >> 4, k >> 4, kp) + (n & 15) * ddrl_tile_kw(k >> 4, kp) + (k & 15);
}
__host__ __device__ constexpr int ddrl_wsize(int N, int kp) { return ddrl_roundup(N, 16) * kp; }

// Developmental gates (fork layer "Developmental", DODevMLP.hpp:276-304, 378-401; semantics as far as the tree pins them:
// CaffeUtest.cpp:619-690, 1048-1158).  A gate weight w in [0, 1] per controlled dimension.
//   probabilist = 1 (bern == 0 here): deterministic -- the stored gate value multiplies the input (the host stores 0 / 1);
//   probabilist = 0 (bern == 1): per SAMPLE and dimension the input passes with probability w, else 0; with `scale` a
//   passed value is divided by w (CaffeUtest.cpp:1126-1131).  The draw is a counter-based hash of (seed, draw counter,
//   agent, site, row, dimension): a backward pass regenerates the mask of its forward pass from the same key, every
//   layer instance ("site": the actor's and the critic's copies share the WEIGHT blob, DODevMLP.hpp:343-355, not the
//   masks) and every update draws afresh.  The fork's own generator cannot be matched: parity unpinned beyond the
//   pass rates the reference tests state.
struct GateMode { int st_bern, st_scale, ac_bern, ac_scale; unsigned seed, draw; };
enum { GS_ACTOR_T_S = 0, GS_CRITIC_T_S, GS_ACTOR_T_OUT, GS_CRITIC_T_A, GS_CRITIC_S, GS_CRITIC_A_REPLAY, GS_ACTOR_S, GS_ACTOR_OUT,
       GS_CRITIC_A, GS_INFER_S, GS_INFER_OUT, GS_BATCH_S, GS_BATCH_A, GS_NFAC_V_S, GS_NFAC_V_S2, GS_NFAC_A_S, GS_NFAC_A_OUT,
       GS_CRITIC2_T_S, GS_CRITIC2_T_A, GS_CRITIC2_S, GS_CRITIC2_A_REPLAY };      // TD3's second critic and its target: layer instances of their own
__host__ __device__ inline unsigned ddrl_mix32(unsigned x) {
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}
// the factor a gated value is multiplied by
__host__ __device__ inline float ddrl_gate_factor(float w, int bern, int scale, unsigned seed, unsigned draw, unsigned agent,
                                                  unsigned site, unsigned row, unsigned dim) {
  if (!bern) return w;
  unsigned h = ddrl_mix32(seed ^ 0x9e3779b9u);
  h = ddrl_mix32(h ^ draw); h = ddrl_mix32(h ^ (agent * 0x632be5abu + site));
  h = ddrl_mix32(h ^ (row * 0x27d4eb2fu + dim * 0x165667b1u + 1u));
  const float u = (float)(h >> 8) * (1.0f / 16777216.0f);              // [0, 1)
  if (!(u < w)) return 0.f;
  return scale ? 1.0f / w : 1.0f;
}

// One MLP's shape and its padded device layout.  Per agent the net is one block of `stride`
// floats: layer l weights at offw(l) (tile-major, above), bias at offb(l).
// Two providers with the same accessors: NetDesc (run-time shapes, the generic kernels) and
// SNet<...> (compile-time shapes: every loop bound, guard and offset folds to a constant).
struct NetDesc {
  int L_;
  int d_[DDRL_MAXL + 1];
  int kp_[DDRL_MAXL];
  int ld_[DDRL_MAXL + 1];    // LDS row stride of the activation of width d(i) (ddrl_ldpad)
  int offw_[DDRL_MAXL];
  int offb_[DDRL_MAXL];
  int stride_;               // floats per agent (multiple of 64 => 256-B aligned blocks)
  int np;                    // logical (flat, reference-order) parameter count
  int hidden_act_, out_act_;
  // BatchNorm (+ Scale) towers (MLP.hpp:53-79, 978-1021): bit r of bn_mask_ = a tower, in place, on the input of
  // InnerProduct r (r < L) or on the net's output (r == L: the actor's "very_all").  Tower r of an agent sits in the
  // agent's block at bn_off_[r]: [mean_sum C4][var_sum C4][factor, 3 pads][gamma C4][beta C4], C4 = roundup(C, 4);
  // gamma / beta are in use per bn_flags_ (the bit-field's low bits: bit 1 Scale, bit 0 its bias).
  int bn_mask_, bn_flags_;
  int bn_c_[DDRL_MAXL + 1], bn_off_[DDRL_MAXL + 1];
  __host__ __device__ bool bn(int r) const { return (bn_mask_ >> r) & 1; }
  __host__ __device__ bool bn_any() const { return bn_mask_ != 0; }
  __host__ __device__ bool bn_scale() const { return (bn_flags_ & 2) != 0; }
  __host__ __device__ bool bn_bias() const { return (bn_flags_ & 1) != 0; }
  // (select chains over constant indices: a run-time index into a by-value kernel argument makes the compiler copy the
  // whole argument block to scratch)
  __host__ __device__ int bn_c(int r) const {
    int v = bn_c_[0];
#pragma unroll
    for (int i = 1; i <= DDRL_MAXL; ++i) v = r == i ? bn_c_[i] : v;
    return v;
  }
  __host__ __device__ int ld_top() const {               // ld(L()) without a run-time index
    int v = ld_[1];
#pragma unroll
    for (int i = 2; i <= DDRL_MAXL; ++i) v = L_ == i ? ld_[i] : v;
    return v;
  }
  __host__ __device__ int bn_c4(int r) const { return ddrl_roundup(bn_c(r), 4); }
  __host__ __device__ int bn_off(int r) const {
    int v = bn_off_[0];
#pragma unroll
    for (int i = 1; i <= DDRL_MAXL; ++i) v = r == i ? bn_off_[i] : v;
    return v;
  }
  __host__ __device__ int bn_stats_floats(int r) const { return 2 * bn_c4(r) + 4; }
  __host__ __device__ int bn_floats(int r) const { return 4 * bn_c4(r) + 4; }
  __host__ __device__ int L() const { return L_; }
  __host__ __device__ int d(int i) const { return d_[i]; }
  __host__ __device__ int kp(int l) const { return kp_[l]; }
  __host__ __device__ int ld(int i) const { return ld_[i]; }
  __host__ __device__ int offw(int l) const { return offw_[l]; }
  __host__ __device__ int offb(int l) const { return offb_[l]; }
  __host__ __device__ int stride() const { return stride_; }
  __host__ __device__ int hidden_act() const { return hidden_act_; }
  __host__ __device__ int out_act() const { return out_act_; }
};

template <int HACT, int OACT, int... D>
struct SNet {
  static constexpr int NL = (int)sizeof...(D) - 1;
  __host__ __device__ static constexpr int L() { return NL; }
  __host__ __device__ static constexpr int d(int i) { constexpr int dd[] = {D...}; return dd[i]; }
  __host__ __device__ static constexpr int kp(int l) { return ddrl_roundup(d(l), 4); }
  __host__ __device__ static constexpr int ld(int i) { return ddrl_ldpad(d(i)); }
  __host__ __device__ static constexpr int offw(int l) {
    int off = 0;
    for (int i = 0; i < l; ++i) off += ddrl_wsize(d(i + 1), kp(i)) + ddrl_roundup(d(i + 1), 4);
    return off;
  }
  __host__ __device__ static constexpr int offb(int l) { return offw(l) + ddrl_wsize(d(l + 1), kp(l)); }
  __host__ __device__ static constexpr int stride() { return ddrl_roundup(offw(NL), 64); }
  __host__ __device__ static constexpr int hidden_act() { return HACT; }
  __host__ __device__ static constexpr int out_act() { return OACT; }
  // deep layers with enough tiles per wave to be worth the optimiser's third state set (bwd_net parks
  // input-gradient tiles in LDS for it, which costs LDS: small nets rather keep a third workgroup per CU)
  static constexpr bool kPark = NL >= 2 && d(1) >= 64;
  // does a run-time descriptor describe this shape?
  static bool matches(const NetDesc &n) {
    if (n.L_ != NL || n.hidden_act_ != HACT || n.out_act_ != OACT) return false;
    for (int i = 0; i <= NL; ++i) if (n.d_[i] != d(i)) return false;
    return n.stride_ == stride();
  }
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
  float *lr_t;              // [G x nupd] lr * sqrt(1-beta2^t)/(1-beta1^t) for the coming update(s) (prep kernel)
  const float *decay;       // [G] L2 weight decay (<0 or 0: off), null if never used
};

struct DdpgParams {
  NetDesc a, q;
  NetState sa, sq;
  int G, S, A, B, cap, rowlen;
  int inverting;
  int nstep;                // agent.nstep_return (<= 0: one-step targets)
  int ldxc, ldxa;           // LDS strides of the critic / actor input tiles
  int ldh;                  // LDS row stride reserved for each hidden-activation buffer
  float beta1, beta2, omb1, omb2, eps;
  float fisher_beta;
  const float *rep;         // [G x cap x rowlen]  rows: s, a, s2, r, goal
  const int *rep_len, *rep_head;
  const int *idx;           // [G x B]
  const float *tau, *omtau, *gamma;   // [G]
  const uint8_t *mask;      // [G] or null
  const float *sgate, *agate;   // developmental gates [G x S], [G x A] (DODevMLP.hpp:376-400), or null
  GateMode gm;                  // deterministic or Bernoulli gates (above)
  float *loss_out;          // [G]
  float *qt_out;            // [G x B] or null
  float *dact_out;          // [G x B x A] or null
  long long *stamps;        // [G x 16] s_memtime at phase boundaries (diagnostic), or null
  // minibatches of more than 64 rows (kernels_ddpg_big.hip): per-chunk partial gradients and losses
  int maxchunks;
  float *gpart_a, *gpart_q; // [G x maxchunks x stride]
  float *loss_part;         // [G x maxchunks]
  // nets wider than DDRL_MAXW (kernels_ddpg_wide.hip): activation tiles in HBM, [G x maxchunks x wide_stride]
  float *wide;
  size_t wide_stride;
  // TD3 (kernels_td3.hip): second critic, target-policy smoothing draws, delayed actor update
  NetState sq2;
  const float *noise;       // [G x B x A]
  const uint8_t *actor_step;   // [G] or null (= every agent's actor is updated)
  float *loss2_out;         // [G]
  float *gpart_q2;          // [G x maxchunks x stride]: the second critic's per-chunk partial gradients (minibatches > 64 rows)
  float *loss_part2;        // [G x maxchunks]
  int nupd;                 // consecutive updates per launch (>= 1): idx is [G x nupd x B], lr_t [G x nupd]
  int *upd_done;            // [G] updates of this launch an agent has completed (zeroed by the prep kernel)
  int *upd_error;           // [1] raised when a workgroup gave up waiting for its predecessor
  // BatchNorm nets (kernels_ddpg_bn.hip)
  int test_net, bn_adapt;   // agent.test_net: the targets run in TEST phase (DeepQNAg.hpp:216-234); agent.bn_adapt (:415-420)
  const float *a_test;      // [G x stride] ann_testing: the actor's parameters as of the end of the previous update call (:469-472)
  int resident;             // the NetState arrays w, m, v, wt, fisher, fhat, wstar point at ONE agent's block (LDS copy, ddpg_update_resident_kernel): no per-agent offset
  int pair;                 // launched as ddpg_update_pair_kernel (kernels_ddpg_pair.hip): an agent's nupd updates stay in one workgroup
};

// NFAC(lambda)-V kernels (kernels_nfac.hip)
struct NfacParams {
  NetDesc a, v;
  NetState sa, sv;
  int G, S, A, maxT, rowlen, maxchunks;
  int ldh;                        // LDS row stride reserved for hidden activations
  int inverting, update_delta_neg, gae;
  int corrected;                  // agent.corrected_update_ac: actor diff = -(target - pi) * corr_factor, no inverting gradient
  float corr_factor;
  int value_passes;               // nfac_values_kernel: bit 0 = V(s), bit 1 = V(s')
  int generic;                    // 1: never the compile-time-shape instantiations (option "force_generic")
  int fuse;                       // 1: the last chunk workgroup of an agent reduces the partials and steps (nfac_grad_kernel); 0: nfac_apply_kernel
  int *chunk_done;                // [G] chunk workgroups of the running gradient launch that have published their partials
  float lambda;
  float beta1, beta2, omb1, omb2, eps, fisher_beta;
  double dbeta1, dbeta2;
  const float *traj;              // [G x maxT x rowlen] rows: s, pure_a, a, s2, r, goal
  const int *len;                 // [G]
  const uint8_t *endflag;         // [G x maxT] row is the last transition of its episode
  const float *gamma;             // [G]
  const float *sgate, *agate;     // developmental gates [G x S], [G x A], or null
  GateMode gm;
  const uint8_t *mask;            // [G] or null
  float *V, *nextV, *tgt, *deltas;   // [G x maxT]
  uint8_t *rowmode;               // [G x maxT] 0: no gradient, 1: target = a, 2: target = pure_a
  int *n_pos;                     // [G]
  float *gpart_a, *gpart_v;       // [G x maxchunks x stride] per net; pads are zero and stay zero
  float *loss_part;               // [G x maxchunks]
  float *loss_out;                // [G]
  const float *lw;                // [G x maxT] label weights of ddrl_nfac_learn_batch_lw (MLP::learn_batch_lw, EuclideanWSLoss), or null
};


// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute of a kernel: the size already
// configured is remembered per device (a handle per GPU may live in one process).  A concurrent
// first launch from two threads at worst sets the attribute twice.
struct LdsConfig { size_t bytes[64] = {}; };
// (a kernel's LdsConfig is shared by every handle of the process: handles driven from several host threads take turns here, so
// that a smaller request never lands on top of a larger one that another thread has just recorded)
inline std::mutex &ddrl_lds_turn() { static std::mutex m; return m; }
template <class K>
static inline hipError_t ddrl_set_lds(K kernel, size_t lds, LdsConfig &cfg) {
  if (lds <= 48 * 1024) return hipSuccess;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63;
  std::lock_guard<std::mutex> hold(ddrl_lds_turn());
  if (lds > cfg.bytes[dev] || dev == 63) {
    hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    cfg.bytes[dev] = lds;
  }
  return hipSuccess;
}
