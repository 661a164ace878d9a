// ddpg_device.h -- pieces shared by the fused DDPG update kernel (kernels_ddpg.hip, minibatches of up
// to 64 rows in one workgroup) and the chunked one (kernels_ddpg_big.hip, larger minibatches).
#pragma once
#include "net_device.h"

#define DDRL_GATHER_TPR (DDRL_NTHREADS / DDRL_MB)   // threads per row in gather_tile
// the thread that wrote rv[row] / dn[row] in gather_tile: it may read them back without a barrier
__device__ __forceinline__ bool gather_owns_row(int tid, int &row) {
  row = tid / DDRL_GATHER_TPR;
  return tid - row * DDRL_GATHER_TPR == 0;
}

// compile-time shapes: BASELINE.md section 3 "mid" (a=100,100 / q=100,100) and "small" (a=20 / q=50,7,
// agent/deepqn/data/ddpg-cp.ini:15-16) HalfCheetah nets (S=18, A=6), leaky-ReLU hidden, tanh actor output
typedef SNet<1, 2, 18, 100, 100, 6> MidActor;
typedef SNet<1, 0, 24, 100, 100, 1> MidCritic;
typedef SNet<1, 2, 18, 20, 6> SmallActor;
typedef SNet<1, 0, 24, 50, 7, 1> SmallCritic;

// does an instantiation park input-gradient tiles in LDS during the deep layers' optimiser loops (bwd_net)?
template <class NA, class NQ> struct ParkOf { static constexpr bool value = NA::kPark && NQ::kPark; };
template <> struct ParkOf<NetDesc, NetDesc> { static constexpr bool value = false; };

struct LdsMap {
  float *hbase; int hstride; // hidden activation l (1..L-1) of the pass in flight lives at hbase + (l-1)*hstride
  float *xc, *xa;            // critic input [s|a] tile, actor input s tile
  float *topq, *topa;        // output tiles: q [64 x 4], actions [64 x ld(A)]
  float *yv, *rv, *dn;       // TD target, reward, goal flag per row
  float *nrew; int *nvalid;  // n-step return: discounted reward prefix and horizon per row
  float *dump;               // DDRL_NWAVES x 64 floats nobody reads: where l2_touch's loads land (mlp_device.h)
};

// gather one 64-row tile (B <= 64 rows are real: idx[0..B)) of the minibatch from the agent's replay ring (DeepQNAg.hpp:353-377,
// nstep_return == 0).  next=true: s' into xa and xc[:, :S];  next=false: s into xa, [s|a] into xc,
// reward / goal into rv / dn.
__device__ __forceinline__ void gather_tile(const DdpgParams &P, int S, int A, int ldxc, int ldxa, int agent,
                                            const int *__restrict__ idx, int B, int head, bool next,
                                            const LdsMap &M, int tid, int u = 0, int row_base = 0) {      // row_base: the tile's first row in the minibatch (chunked updates: the masks are drawn per minibatch row)
  // DDRL_NTHREADS / 64 threads per row, each with its own column residue: a thread reads idx[row] once
  // and its row loads are independent of each other, so the whole gather costs two memory round trips
  // (index, then row) instead of two per element
  constexpr int TPR = DDRL_GATHER_TPR;
  const float *rep = P.rep + (size_t)agent * P.cap * P.rowlen;
  const int row = tid / TPR, q = tid - row * TPR;
  const bool live = row < B;
  int phys = 0;
  if (live) {
    phys = head + idx[row] + (next && P.nstep > 0 ? M.nvalid[row] : 0);   // s' of the last valid step (:361-372)
    if (phys >= P.cap) phys -= P.cap;
  }
  const float *rr = rep + (size_t)phys * P.rowlen;
  const float *sg = P.sgate ? P.sgate + (size_t)agent * S : nullptr, *ag = P.agate ? P.agate + (size_t)agent * A : nullptr;
#pragma unroll 8
  for (int col = q; col < ldxc; col += TPR) {
    float v = 0.f;
    if (live) {
      if (next) { if (col < S) v = rr[S + A + col]; }
      else if (col < S + A) v = rr[col];
    }
    // developmental gates on the state inputs of both nets (each net's own layer instance: its own Bernoulli draw) and on
    // the critic's action input
    float vc = v, va = v;
    if (live && (sg || ag)) {
      const GateMode &gm = P.gm;
      const unsigned draw = gm.draw + (unsigned)u;
      if (sg && col < S) {
        vc *= ddrl_gate_factor(sg[col], gm.st_bern, gm.st_scale, gm.seed, draw, agent, next ? GS_CRITIC_T_S : GS_CRITIC_S, row_base + row, col);
        va *= ddrl_gate_factor(sg[col], gm.st_bern, gm.st_scale, gm.seed, draw, agent, next ? GS_ACTOR_T_S : GS_ACTOR_S, row_base + row, col);
      }
      if (ag && col >= S && col < S + A)
        vc *= ddrl_gate_factor(ag[col - S], gm.ac_bern, gm.ac_scale, gm.seed, draw, agent, GS_CRITIC_A_REPLAY, row_base + row, col - S);
    }
    M.xc[row * ldxc + col] = vc;
    if (col < ldxa) M.xa[row * ldxa + col] = col < S ? va : 0.f;
  }
  if (!next && q == 0) {
    M.rv[row] = live ? rr[2 * S + A] : 0.f;
    M.dn[row] = live ? rr[2 * S + A + 1] : 0.f;
  }
}

// The same gather in two halves, for the kernels with compile-time shapes: gather_load only REQUESTS the thread's
// share of its row (idx[row], then NCOL row elements + reward / goal flag: two dependent memory round trips, 2-5 k
// cycles each under load) into registers; gather_store applies the gates and writes the LDS tiles.  The request can
// therefore be issued a whole pass -- or a whole quarter of an update -- before the tiles are needed: the replay ring
// and the index block are constant during a launch (the prep kernel drew the indices of all its updates), so nothing
// orders it behind the previous update's stores.  nstep_return > 0 keeps gather_tile (the row depends on the
// horizon computed in LDS).
#define DDRL_GATHER_MAXC 8
struct GatherRegs { float v[DDRL_GATHER_MAXC]; float r, dn; };

template <int NCOL>
__device__ __forceinline__ void gather_load(const DdpgParams &P, int S, int A, int ldxc, int agent,
                                            const int *__restrict__ idx, int B, int head, bool next, int tid, GatherRegs &g) {
  constexpr int TPR = DDRL_GATHER_TPR;
  static_assert(NCOL <= DDRL_GATHER_MAXC, "gather_load: more columns per thread than GatherRegs holds");
  const float *rep = P.rep + (size_t)agent * P.cap * P.rowlen;
  const int row = tid / TPR, q = tid - row * TPR;
  const bool live = row < B;
  int phys = 0;
  if (live) {
    phys = head + idx[row];
    if (phys >= P.cap) phys -= P.cap;
  }
  const float *rr = rep + (size_t)phys * P.rowlen;
#pragma unroll
  for (int j = 0; j < NCOL; ++j) {
    const int col = q + j * TPR;
    float v = 0.f;
    if (live && col < ldxc) {
      if (next) { if (col < S) v = rr[S + A + col]; }
      else if (col < S + A) v = rr[col];
    }
    g.v[j] = v;
  }
  g.r = 0.f; g.dn = 0.f;
  if (!next && q == 0 && live) { g.r = rr[2 * S + A]; g.dn = rr[2 * S + A + 1]; }
}

template <int NCOL>
__device__ __forceinline__ void gather_store(const DdpgParams &P, int S, int A, int ldxc, int ldxa, int agent, int B, bool next,
                                             const LdsMap &M, int tid, int u, const GatherRegs &g) {
  constexpr int TPR = DDRL_GATHER_TPR;
  const int row = tid / TPR, q = tid - row * TPR;
  const bool live = row < B;
  const float *sg = P.sgate ? P.sgate + (size_t)agent * S : nullptr, *ag = P.agate ? P.agate + (size_t)agent * A : nullptr;
#pragma unroll
  for (int j = 0; j < NCOL; ++j) {
    const int col = q + j * TPR;
    if (col < ldxc) {
      const float v = g.v[j];
      float vc = v, va = v;
      if (live && (sg || ag)) {          // the gates of gather_tile
        const GateMode &gm = P.gm;
        const unsigned draw = gm.draw + (unsigned)u;
        if (sg && col < S) {
          vc *= ddrl_gate_factor(sg[col], gm.st_bern, gm.st_scale, gm.seed, draw, agent, next ? GS_CRITIC_T_S : GS_CRITIC_S, row, col);
          va *= ddrl_gate_factor(sg[col], gm.st_bern, gm.st_scale, gm.seed, draw, agent, next ? GS_ACTOR_T_S : GS_ACTOR_S, row, col);
        }
        if (ag && col >= S && col < S + A)
          vc *= ddrl_gate_factor(ag[col - S], gm.ac_bern, gm.ac_scale, gm.seed, draw, agent, GS_CRITIC_A_REPLAY, row, col - S);
      }
      M.xc[row * ldxc + col] = vc;
      if (col < ldxa) M.xa[row * ldxa + col] = col < S ? va : 0.f;
    }
  }
  if (!next && q == 0) { M.rv[row] = g.r; M.dn[row] = g.dn; }
}

// Bernoulli gates (DODevMLP.hpp:276-304, probabilist = 0): every Developmental layer instance draws its own mask per forward
// pass, and TD3 has two more nets with such layers than DDPG (qnn2, qnn_target2).  The two critics of a pair read the same
// input tile, so for the second one -- and for the first critic's pass of the actor phase, after the second critic has had
// the tile -- the critic's input tile is built again from the replay rows with that net's sites (uniform branch: the
// deterministic gates are the same for every instance and never come here).
__device__ __forceinline__ void td3_regate_xc(const DdpgParams &P, int S, int A, int ldxc, int agent, const int *__restrict__ idx, int B,
                                              int head, bool next, const LdsMap &M, int tid, int site_s, int site_a, int row_base = 0) {
  const float *rep = P.rep + (size_t)agent * P.cap * P.rowlen;
  const float *sg = P.sgate ? P.sgate + (size_t)agent * S : nullptr, *ag = P.agate ? P.agate + (size_t)agent * A : nullptr;
  const GateMode &gm = P.gm;
  const int ncol = next ? S : S + A;
  for (int e = tid; e < DDRL_MB * ncol; e += DDRL_NTHREADS) {
    const int row = e / ncol, col = e - row * ncol;
    float v = 0.f;
    if (row < B) {
      int phys = head + idx[row];
      if (phys >= P.cap) phys -= P.cap;
      v = rep[(size_t)phys * P.rowlen + (next ? S + A + col : col)];
      if (col < S) { if (sg) v *= ddrl_gate_factor(sg[col], gm.st_bern, gm.st_scale, gm.seed, gm.draw, agent, site_s, row_base + row, col); }
      else if (ag) v *= ddrl_gate_factor(ag[col - S], gm.ac_bern, gm.ac_scale, gm.seed, gm.draw, agent, site_a, row_base + row, col - S);
    }
    M.xc[row * ldxc + col] = v;
  }
}


template <bool EWC>
__device__ __forceinline__ OptCtx make_opt(const DdpgParams &P, const NetState &s, int stride,
                                           int agent, float tau, float omtau, float nsample, int u = 0) {
  OptCtx o;
  const size_t off = P.resident ? 0 : (size_t)agent * stride;
  o.w = s.w + off; o.m = s.m + off; o.v = s.v + off;
  o.wt = s.wt + off;
  o.fisher = EWC ? s.fisher + off : nullptr;
  o.fhat = EWC ? s.fhat + off : nullptr;
  o.wstar = EWC ? s.wstar + off : nullptr;
  o.lr_t = s.lr_t[(size_t)agent * P.nupd + u];     // bias-corrected rate of update u of this launch, computed in fp64 by the prep kernel
  o.tau = tau; o.omtau = omtau;
  o.decay = s.decay[agent];
  o.ewc_coef = EWC ? s.ewc_coef[agent] : 0.f;
  o.fisher_beta = P.fisher_beta;
  o.nsample = nsample;
  o.beta1 = P.beta1; o.beta2 = P.beta2; o.omb1 = P.omb1; o.omb2 = P.omb2; o.eps = P.eps;
#ifdef DDRL_EXP_NOMSKIP
  o.mld = o.m;
#else
  o.mld = P.beta1 == 0.f ? o.w : o.m;
#endif
  o.dump = nullptr;
  o.handoff = P.nupd > 1 && !P.pair;     // the pair kernel keeps an agent's consecutive updates in one workgroup
  return o;
}


// n-step horizon (DeepQNAg.hpp:360-373, 394-402) of the tile's rows: it stops before a transition
// that reached the goal and at the end of the replay; rewards of steps [0, valid) are discounted
__device__ __forceinline__ void nstep_horizon(const DdpgParams &P, int S, int A, int agent, const int *__restrict__ idx,
                                              int B, int len, int head, float gamma, const LdsMap &M, int tid) {
  if (tid < DDRL_MB) {
    int valid = 0;
    float acc = 0.f;
    if (tid < B) {
      const float *rep = P.rep + (size_t)agent * P.cap * P.rowlen;
      const int i0 = idx[tid];
      int last = i0;
      for (int j = i0 + 1; j < i0 + P.nstep && j < len; ++j) {
        const int ph = head + j >= P.cap ? head + j - P.cap : head + j;
        if (rep[(size_t)ph * P.rowlen + 2 * S + A + 1] != 0.f) break;
        last = j;
      }
      valid = last - i0;
      float lg = 1.f;
      for (int j = 0; j < valid; ++j) {
        const int ph = head + i0 + j >= P.cap ? head + i0 + j - P.cap : head + i0 + j;
        acc += rep[(size_t)ph * P.rowlen + 2 * S + A] * lg;
        lg *= gamma;
      }
    }
    M.nvalid[tid] = valid;
    M.nrew[tid] = acc;
  }
  ddrl_sync();
}

// carve LDS (same order as ddpg_update_lds_bytes)
__device__ __forceinline__ LdsMap ddpg_carve(float *p, int La, int Lq, int ldh, int ldxc, int ldxa, int ldq, int lda) {
  LdsMap M;
  M.hbase = p; M.hstride = DDRL_MB * ldh;
  p += ((La > Lq ? La : Lq) - 1) * M.hstride;
  M.xc = p; p += DDRL_MB * ldxc;
  M.xa = p; p += DDRL_MB * ldxa;
  M.dump = p; p += DDRL_NWAVES * 64;     // before topq: from topq on the allocation doubles as the park area of bwd_net
  M.topq = p; p += DDRL_MB * ldq;
  M.topa = p; p += DDRL_MB * lda;
  M.yv = p; p += DDRL_MB;
  M.rv = p; p += DDRL_MB;
  M.dn = p; p += DDRL_MB;
  M.nrew = p; p += DDRL_MB;
  M.nvalid = (int *)p; p += DDRL_MB;
  return M;
}

// XCD-aware workgroup order (layer-by-layer forward / input-gradient launches, chunk launches of large minibatches).
// Workgroups go to the 8 XCDs round-robin by linear id.  The NB column blocks of one (agent, chunk) read the same 64
// activation rows, the chunks of one agent the same weights: such a group takes consecutive slots of ONE XCD, so
// the shared operand comes from HBM once and from that XCD's L2 afterwards (with x = block, the group sat on
// different XCDs and each member fetched it again).  NB blocks per group, C x G groups.
__host__ __device__ inline unsigned xcd_grid(int NB, int C, int G) { return (unsigned)((C * G + 7) / 8 * 8) * NB; }
__device__ __forceinline__ bool xcd_decode(int NB, int C, int G, int &nb, int &chunk, int &agent) {
  const unsigned id = blockIdx.x, xcd = id & 7, j = id >> 3;
  nb = j % NB;
  const unsigned pair = (j / NB) * 8 + xcd;
  chunk = pair % C; agent = pair / C;
  return agent < G;
}

