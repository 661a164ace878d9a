// net_device.h -- whole-net forward / backward passes over one 64-row LDS tile, built from the
// layer blocks of mlp_device.h.  Shared by the fused DDPG kernel and the NFAC kernels.
#pragma once
#include "mlp_device.h"

// dW / db of one layer for one 64-row chunk, written (not accumulated) to gout: the chunk's partial
// gradient, reduced over chunks in a fixed order by the apply kernel (deterministic, no atomics)
__device__ __forceinline__ void bwd_dw_store(const float *__restrict__ dYs, int ldy, int N,
                                             const float *__restrict__ Xs, int ldx, int kp,
                                             int offw, int offb, float *__restrict__ gout,
                                             int wave, int lane, int tid, bool sc1 = false) {   // sc1: read by another workgroup of this launch
  const int g = lane >> 4, c = lane & 15;
  const int NT = (N + 15) >> 4, KT = (kp + 15) >> 4;
  const int total = NT * KT;
  for (int p = wave; p < total; p += DDRL_NWAVES) {
    bool ok;
    const int i = dw_elem(p, total, KT, N, kp, offw, c, g, ok);
    const f32x4 acc = dw_tile(dYs, ldy, N, Xs, ldx, kp, p, total, KT, c, g);
    if (ok) { if (sc1) store_sc1(gout + i, acc); else *(f32x4 *)(gout + i) = acc; }
  }
  for (int n = tid; n < N; n += DDRL_NTHREADS) {
    const float bg = bias_grad(dYs, ldy, n);
    if (sc1) store_sc1(gout + offb + n, bg); else gout[offb + n] = bg;
  }
}

// a wave's LDS stores must have landed before the same wave's later loads of other lanes' values
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// forward through layers [0, nl) of a net; hidden activation l goes to hbase + (l-1)*hstride, the
// output of the last InnerProduct (when nl == L) to top.  Row-parallel: every wave walks its own 16
// rows through all layers; one workgroup barrier at the end publishes the results.
// The weight fragments a forward pass starts with (tiles 0 and 1 of its first layer) can be requested
// by the caller long before the pass -- at kernel start, or at the end of the previous pass -- and
// travel in a FwdCarry; `after` names the first layer of the pass that follows (null: none).
struct FwdCarry { BFrag b0, b1; };

template <class ND>
__device__ __forceinline__ FwdNext fwd_first(const ND &nd, const float *__restrict__ w) {
  FwdNext nx = {w + nd.offw(0), w + nd.offb(0), nd.kp(0), nd.d(1)};
  return nx;
}
// the column tile behind tile slot t of this wave's part in a layer of NT column tiles (mlp_device.h, DDRL_PARTS), clamped
__device__ __forceinline__ int fwd_slot_tile(int t, int NT, int part) {
  const int nt = part + DDRL_PARTS * t;
  return nt < NT ? nt : NT - 1;
}
__device__ __forceinline__ void fwd_request(const FwdNext nx, int lane, FwdCarry &cy, int wave = 0) {     // (compile-time shapes only: the short-tail fragment form)
  const float *W = opaque(nx.W), *bias = opaque(nx.bias);
  const int NT = (nx.N + 15) >> 4, part = ddrl_part(wave);
  load_bfrag_st(W, bias, nx.kp, nx.N, fwd_slot_tile(0, NT, part), lane, cy.b0);
  if (1 < ddrl_slots(NT)) load_bfrag_st(W, bias, nx.kp, nx.N, fwd_slot_tile(1, NT, part), lane, cy.b1);
}

// ---- compile-time shapes: the whole pass as ONE static schedule ----
// The rolled tile loop of fwd_layer keeps its weight-fragment loads under conditions, which costs the s_waitcnt
// bookkeeping its precision (the built kernel waited with vmcnt(8): ONE fragment set in flight, not two), and it
// requested a layer's first tiles only after the previous layer's last tile.  Here the tiles of all layers of the pass
// form one sequence: tile j sits in fragment set j % 3, the loads for tile j + 2 are issued before tile j is computed,
// across layer boundaries and -- HAS_AFTER -- into the first two tiles of the pass that follows.  Fully unrolled: every
// guard folds, every vector-memory operation is unconditional, the waits are exact.
// Hand-over convention (unchanged): a pass starts with its tiles 0 / 1 in cy.b0 / cy.b1; that is where this pass leaves
// them for its successor when its tile count is a multiple of three (true for every instantiated net), else the successor
// starts cold.
template <class ND> static __host__ __device__ constexpr int fwd_pass_tiles(int nl) {
  int t = 0;
  for (int l = 0; l < nl; ++l) t += ddrl_slots((ND::d(l + 1) + 15) / 16);      // tile SLOTS of one part
  return t;
}
template <class ND, bool HAS_AFTER>
__device__ __forceinline__ void fwd_net_ct(const ND &nd, int nl, const float *__restrict__ w,
                                           const float *X, int ldx, float *hbase, int hstride, float *top, int ldtop,
                                           int wave, int lane, FwdCarry &cy, bool preloaded, const FwdNext after) {
  const int g = lane >> 4, c = lane & 15;
  const int part = ddrl_part(wave), rt = ddrl_rt(wave);
  BFrag b2;
  const float *wo = opaque(w);
  const float *aW = HAS_AFTER ? opaque(after.W) : nullptr, *ab = HAS_AFTER ? opaque(after.bias) : nullptr;
  const bool hand = HAS_AFTER && fwd_pass_tiles<ND>(nl) % 3 == 0;
  // request tile SLOT `t` of layer `l` (or, past the last layer of the pass, of the successor) into fragment set `set`
  auto request = [&](int l, int t, int set) {
    // walk forward over layers whose slots are exhausted (folds: every bound is a constant)
    bool done = false;
#pragma unroll
    for (int step = 0; step <= DDRL_MAXL; ++step) {
      if (!done && l < nl) {
        const int NTl = (nd.d(l + 1) + 15) >> 4;
        if (t < ddrl_slots(NTl)) {
          BFrag &dst = set == 0 ? cy.b0 : set == 1 ? cy.b1 : b2;
          load_bfrag_st(wo + nd.offw(l), wo + nd.offb(l), nd.kp(l), nd.d(l + 1), fwd_slot_tile(t, NTl, part), lane, dst);
          done = true;
        } else { t -= ddrl_slots(NTl); ++l; }
      }
    }
    if (!done && hand && l >= nl && t < 2 && t < ddrl_slots((after.N + 15) >> 4)) {
      BFrag &dst = set == 0 ? cy.b0 : set == 1 ? cy.b1 : b2;
      load_bfrag_st(aW, ab, after.kp, after.N, fwd_slot_tile(t, (after.N + 15) >> 4, part), lane, dst);
    }
  };
  if (!preloaded) { request(0, 0, 0); request(0, 1, 1); }
  const float *in = X;
  int ldi = ldx, j = 0;
#pragma unroll
  for (int l = 0; l < nd.L(); ++l) {
    if (l < nl) {
      const bool last = l == nd.L() - 1;
      float *out = last ? top : hbase + l * hstride;
      const int ldo = last ? ldtop : nd.ld(l + 1);
      const int N = nd.d(l + 1), kp = nd.kp(l), act = last ? nd.out_act() : nd.hidden_act();
      const int NT = (N + 15) >> 4;
      AFrag a;
      load_afrag_st(in, ldi, kp, wave, lane, a);
      float *yrow = out + (16 * rt + 4 * g) * ldo;
#pragma unroll
      for (int t = 0; t < ddrl_slots(NT); ++t, ++j) {
        request(l, t + 2, (j + 2) % 3);
        const BFrag &b = j % 3 == 0 ? cy.b0 : j % 3 == 1 ? cy.b1 : b2;
        const f32x4 acc = row_tile_st(a, b, kp);
        const int n = fwd_slot_tile(t, NT, part) * 16 + c;
        if (n < ldo && (DDRL_PARTS == 1 || part + DDRL_PARTS * t < NT)) {      // (uniform) a slot beyond the last tile stores nothing
#pragma unroll
          for (int r = 0; r < 4; ++r) yrow[r * ldo + n] = n < N ? act_fwd(act, acc[r] + b.bias) : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // one wave per row tile: its own LDS ordering is enough; two: the row's other columns are the partner's
      if (DDRL_PARTS == 1) wave_lds_fence(); else ddrl_sync();
      in = out;
      ldi = ldo;
    }
  }
  if (DDRL_PARTS == 1) ddrl_sync();
}

template <class ND, bool HAS_AFTER = true>
__device__ __forceinline__ void fwd_net(const ND &nd, int nl, const float *__restrict__ w,
                                        const float *X, int ldx, float *hbase, int hstride, float *top, int ldtop,
                                        int wave, int lane, FwdCarry &cy, bool preloaded, const FwdNext after) {
#ifndef DDRL_NO_FWD_CT
  if constexpr (!__is_same(ND, NetDesc)) {
    fwd_net_ct<ND, HAS_AFTER>(nd, nl, w, X, ldx, hbase, hstride, top, ldtop, wave, lane, cy, preloaded, after);
    return;
  }
#endif
  const float *in = X;
  int ldi = ldx;
#pragma unroll
  for (int l = 0; l < nd.L(); ++l) {
    if (l < nl) {
      const bool last = l == nd.L() - 1;
      float *out = last ? top : hbase + l * hstride;
      const int ldo = last ? ldtop : nd.ld(l + 1);
      FwdNext nx = after;                                // the last layer of this pass hands over to the next pass
      // run-time shapes keep all DDRL_MAXKC chunks of every fragment set allocated: there the hand-over
      // costs more registers than it saves time (NFAC's 64-wide nets dropped to one wave per SIMD)
      constexpr bool handover = !__is_same(ND, NetDesc);
      if (!handover) nx.W = nullptr;
      if (handover && l + 1 < nl) {
        const int l1 = l + 1 < nd.L() ? l + 1 : l;       // (keeps the accessors in range when folded)
        nx.W = w + nd.offw(l1); nx.bias = w + nd.offb(l1); nx.kp = nd.kp(l1); nx.N = nd.d(l1 + 1);
      }
      fwd_layer(in, ldi, nd.kp(l), w + nd.offw(l), w + nd.offb(l), nd.d(l + 1),
                last ? nd.out_act() : nd.hidden_act(), out, ldo, wave, lane, cy.b0, cy.b1, handover && (preloaded || l > 0), nx);
      wave_lds_fence();
      in = out;
      ldi = ldo;
    }
  }
  ddrl_sync();
}

// a pass on its own: nothing requested ahead, nothing handed over
template <class ND>
__device__ __forceinline__ void fwd_net(const ND &nd, int nl, const float *__restrict__ w,
                                        const float *X, int ldx, float *hbase, int hstride, float *top, int ldtop,
                                        int wave, int lane) {
  FwdCarry cy;
  const FwdNext none = {nullptr, nullptr, 4, 1};
  fwd_net<ND, false>(nd, nl, w, X, ldx, hbase, hstride, top, ldtop, wave, lane, cy, false, none);
}

// Backward through a whole net from dZ_L held in `top` (diff w.r.t. the last InnerProduct's
// output; the caller has synchronised the workgroup after writing it).  The dX chain is row-
// parallel (no barrier); the dW products contract over all 64 rows and are bracketed by barriers.
// DW_OPT: parameter gradients go through the fused optimiser epilogue (with a target twin).
// DW_STORE: they are written to gout (one agent-net image, same layout as the weights) for a later
// reduction over batch chunks.  nc > 0: the input gradient columns [c0, c0 + nc) are handed to
// dx0_store(row, col - c0, value); the caller synchronises before other waves read what it wrote.
enum { DW_NONE = 0, DW_OPT = 1, DW_STORE = 2 };
template <class ND> struct KParkOf { static constexpr bool value = ND::kPark; };
template <> struct KParkOf<NetDesc> { static constexpr bool value = false; };
#define DDRL_PARK_TILES (4 / DDRL_PARTS)     // input-gradient tiles (of up to DDRL_MAXKC) parked in LDS during a deep layer's optimiser loop
// TGT: the optimiser epilogue also moves the net's target twin (false: TD3's critics between two policy steps)
template <int TRAIN, bool EWC, bool TGT = true, class ND, class DX0>
__device__ __forceinline__ void bwd_net(const ND &nd, const float *__restrict__ w, const OptCtx &opt, float *gout,
                                        const float *X, int ldx, float *hbase, int hstride, float *top, int ldtop,
                                        int c0, int nc, DX0 dx0_store, int wave, int lane, int tid,
                                        long long *bst = nullptr,     // bst: diagnostic s_memtime slots (dX done / dW done per layer)
                                        float *park = nullptr) {      // park: DDRL_PARK_TILES * 1 KiB of LDS per wave, dead during the dW loops below the top layer
  static_assert(DDRL_PARTS == 1 || !__is_same(ND, NetDesc), "column parts: compile-time shapes only");
  const int g = lane >> 4, c = lane & 15;
  const int part = ddrl_part(wave), rt = ddrl_rt(wave);
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
    f32x4 acc[DDRL_MAXKC];
    int kt0 = 0, kt1 = (K + 15) >> 4;
    if (l == 0) { kt0 = c0 >> 4; kt1 = (c0 + nc + 15) >> 4; }
    // compile-time shapes: the static optimiser schedule of mlp_device.h (bwd_dw_opt_ct); the state of the layer's first
    // tile slots is requested HERE, ahead of the input-gradient pass, so that the tile loop does not start on a cold wait
#ifdef DDRL_NO_OPT_CT
    constexpr bool CT = false;
#else
    constexpr bool CT = !__is_same(ND, NetDesc) && TRAIN == DW_OPT;
#endif
    constexpr bool deep_ct = CT && KParkOf<ND>::value;
    OptCarry<3> oc3;
    OptCarry<2> oc2;
    if constexpr (CT) {
      if (deep_ct && l < nd.L() - 1) dw_opt_request<EWC, TGT, 3>(N, kp, nd.offw(l), opt, wave, lane, oc3);
      else dw_opt_request<EWC, TGT, 2>(N, kp, nd.offw(l), opt, wave, lane, oc2);
    }
    if (need_dx) bwd_dx<!__is_same(ND, NetDesc)>(dZ, ldz, N, W, kp, kt0, kt1, acc, wave, lane);
    if (bst && tid == 0 && ll < 2) bst[2 * ll] = clock64();
    if (TRAIN != DW_NONE) {
      ddrl_sync();   // every wave's rows of dZ_l are final and every wave has read W_l
      if (TRAIN == DW_OPT) {
        // Below the top layer the first DDRL_PARK_TILES input-gradient tiles wait in LDS while the
        // optimiser loop runs, which makes room for its third state set (the first layer has no
        // input gradient at all).  Wave-private: no barrier, only the wave's own LDS ordering.
#ifdef DDRL_EXP_SETS2      // measurement: two optimiser state sets everywhere (how much does the third one buy?)
        const bool deep = false;
#else
        const bool deep = park != nullptr && l < nd.L() - 1;
#endif
        if (deep && need_dx) {
#pragma unroll
          for (int i = 0; i < DDRL_PARK_TILES; ++i)
            if (i < ddrl_slots(kt1 - kt0)) *(f32x4 *)(park + (wave * DDRL_PARK_TILES + i) * 256 + lane * 4) = acc[i];
        }
        if constexpr (CT) {
          if (deep_ct && l < nd.L() - 1) bwd_dw_opt_ct<EWC, TGT, 3>(dZ, ldz, N, in, ldi, kp, nd.offw(l), nd.offb(l), opt, wave, lane, tid, oc3, true);
          else bwd_dw_opt_ct<EWC, TGT, 2>(dZ, ldz, N, in, ldi, kp, nd.offw(l), nd.offb(l), opt, wave, lane, tid, oc2, true);
        } else {
          if (deep) bwd_dw_opt<EWC, TGT, true>(dZ, ldz, N, in, ldi, kp, nd.offw(l), nd.offb(l), opt, wave, lane, tid);
          else bwd_dw_opt<EWC, TGT, false>(dZ, ldz, N, in, ldi, kp, nd.offw(l), nd.offb(l), opt, wave, lane, tid);
        }
        if (deep && need_dx) {
#pragma unroll
          for (int i = 0; i < DDRL_PARK_TILES; ++i)
            if (i < ddrl_slots(kt1 - kt0)) acc[i] = *(const f32x4 *)(park + (wave * DDRL_PARK_TILES + i) * 256 + lane * 4);
        }
      }
      else bwd_dw_store(dZ, ldz, N, in, ldi, kp, nd.offw(l), nd.offb(l), gout, wave, lane, tid, opt.handoff != 0);
      if (bst && tid == 0 && ll < 2) bst[2 * ll + 1] = clock64();
      ddrl_sync();   // every wave has read H_{l-1} before its rows are overwritten by dZ_{l-1}
    }
    if (need_dx) {
#pragma unroll
      for (int i = 0; i < DDRL_MAXKC; ++i) {
        const int kt = kt0 + part + DDRL_PARTS * i;          // the column tile of this wave's slot i (bwd_dx)
        if (i < ddrl_slots(kt1 - kt0) && kt < kt1) {
          const int col = kt * 16 + c;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * rt + 4 * g + r;
            if (l > 0) {
              if (col < K) in[row * ldi + col] = act_bwd(nd.hidden_act(), in[row * ldi + col], acc[i][r]);
            } else if (col >= c0 && col < c0 + nc) {
              dx0_store(row, col - c0, acc[i][r]);
            }
          }
        }
      }
      if (DDRL_PARTS == 1) wave_lds_fence(); else ddrl_sync();     // (two waves per row tile: the partner wrote the other columns)
    }
    dZ = in;
    ldz = ldi;
  }
}

// run-time descriptors live in the kernel arguments; compile-time ones are empty tag objects
template <class ND> struct Pick { static __device__ __forceinline__ ND get(const NetDesc &) { return ND(); } };
template <> struct Pick<NetDesc> { static __device__ __forceinline__ const NetDesc &get(const NetDesc &n) { return n; } };

