// mlp_device.h -- workgroup-level fp32 MFMA building blocks shared by the fused DDPG / NFAC /
// batched-inference kernels (gfx950 only).
//
// One 256-thread workgroup (4 waves, one per SIMD) owns one agent.  A batch tile of MB = 64 rows
// lives in LDS; weights are read straight from global memory (L2/MALL resident between passes)
// with one 16-byte load per lane per 4 MFMAs; products run on v_mfma_f32_16x16x4_f32, which is
// bit-for-bit an fp32 fma chain (cdna_hip_programming.md section 3), so results do not depend on
// tile shapes beyond the order of the k-sum.
//
// Forward passes and input-gradient (dX) passes are ROW-PARALLEL: wave w owns batch rows
// 16w .. 16w+15 and carries them through every layer by itself (all output column tiles), so a
// whole pass needs no workgroup barrier, every wave issues the same number of MFMAs whatever the
// layer width (a 1-wide critic output costs a quarter tile per wave, not a full tile on one
// wave), and the activation fragments are read from LDS once per layer instead of once per
// column tile.  Only the dW products contract over all 64 rows and synchronise the workgroup.
//
// Every function takes the layer shape as plain ints.  Called from a kernel instantiated on a
// compile-time net (SNet<...>) they arrive as constants and all guards / loop bounds fold; called
// from the generic kernel (NetDesc) the same source handles any shape up to DDRL_MAXW.
//
// Lane roles for v_mfma_f32_16x16x4_f32 (lane l, c = l & 15, g = l >> 4):
//   A operand: A[i = c][kk = g]     B operand: B[kk = g][j = c]     D reg r: D[i = 4g + r][j = c]
// A 16-deep k-chunk is issued as 4 MFMAs; in MFMA jj the kk slot g carries k = k0 + 4g + jj, so
// each lane feeds the 4 MFMAs from ONE float4 of its operand row (k0 + 4g .. k0 + 4g + 3).
#pragma once
#include "pop_internal.h"

// The barrier of everything in these headers.  Default: the hardware workgroup barrier.  A translation unit that
// runs TWO independent 256-thread teams inside one 512-thread workgroup (kernels_ddpg_pair.hip: two agents per
// CU held in opposite phases) defines DDRL_TEAMS before including them: the barrier then joins the four waves of
// the caller's team only, through a counter in LDS (gfx950 has no named barriers).  The counter only grows: a
// wave arriving as number `old` of its team waits for the count to reach the next multiple of four.  The wait is
// bounded; on expiry team_err is raised and the kernel reports it (results are then undefined, but it ends).
#ifdef DDRL_TEAMS
__shared__ unsigned ddrl_team_cnt[2];
__shared__ int ddrl_team_err;
__device__ __forceinline__ void ddrl_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // this wave's LDS and global accesses are done before it arrives
  if ((threadIdx.x & 63) == 0) {
    unsigned *cnt = &ddrl_team_cnt[threadIdx.x >> 8];
    const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const unsigned target = (old & ~3u) + 4u;
    int spins = 0;
    while ((int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - target) < 0) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22)) { ddrl_team_err = 1; break; }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
#else
__device__ __forceinline__ void ddrl_sync() { __syncthreads(); }
#endif

// ROW TILES AND COLUMN PARTS.  A batch tile has four 16-row tiles.  With four waves per workgroup (the default) wave w owns
// row tile w and every column tile of it.  A translation unit built with DDRL_NWAVES = 8 (kernels_ddpg_w8.hip: one
// 512-thread workgroup per agent, two waves per SIMD, for populations that do not fill the CUs with 256-thread
// workgroups) gives each row tile to TWO waves, w and w + 4: the compile-time-shape passes deal a layer's column tiles to
// them alternately (`part` 0 takes tiles 0, 2, 4 ..., part 1 tiles 1, 3, 5 ...) and close every layer with a workgroup
// barrier instead of the wave's own LDS fence.  Both parts walk the same number of tile slots; a slot beyond the layer's
// last tile repeats that tile (clamped loads, products on valid data) and stores nothing.
#define DDRL_PARTS (DDRL_NWAVES / 4)
static_assert(DDRL_NWAVES % 4 == 0 && DDRL_MB == 64 && DDRL_PARTS >= 1 && DDRL_PARTS <= 2, "row tiles: four per batch tile, one or two waves each");
__device__ __forceinline__ int ddrl_rt(int wave) { return DDRL_PARTS == 1 ? wave : (wave & 3); }
__device__ __forceinline__ int ddrl_part(int wave) { return DDRL_PARTS == 1 ? 0 : (wave >> 2); }
static __host__ __device__ constexpr int ddrl_slots(int tiles) { return (tiles + DDRL_PARTS - 1) / DDRL_PARTS; }   // tile slots of one part
#ifdef DDRL_TEAMS
static_assert(DDRL_NWAVES == 4, "two-team workgroups: the team barrier counts four waves");
#endif

#ifndef DDRL_F64_DEBUG
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
#else
// fp64 debug build: the same 16 x 16 x 4 product as an fma chain over k = 0..3 in the MFMA's order, operands fetched
// from the lanes that hold them (lane (c, g): A[i = c][k = g], B[k = g][j = c]; result reg r: D[4g + r][c])
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  const int lane = threadIdx.x & 63, g = lane >> 4, col = lane & 15;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float acc = c[r];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc = fma(__shfl(a, k * 16 + 4 * g + r), __shfl(b, k * 16 + col), acc);
    c[r] = acc;
  }
  return c;
}
#endif
// leaky-ReLU slope (Caffe ReLULayer negative_slope, MLP.hpp:888-897) in the arithmetic type
constexpr float kLeaky = (float)0.01;
__device__ __forceinline__ f32x4 zero4() { return (f32x4){0.f, 0.f, 0.f, 0.f}; }
// Hides a (uniform) pointer's provenance from the optimiser: the same weights are walked by several
// passes of one kernel, and without this the per-lane addresses of ALL passes are computed once,
// kept alive across the whole kernel and spilled (150 64-bit values in the mid-size DDPG kernel).
__device__ __forceinline__ const float *opaque(const float *p) {
  int z = 0;
  asm volatile("" : "+s"(z));      // an offset of 0 the optimiser cannot see; the address space of p is kept
  return p + z;
}

// a uniform GLOBAL pointer whose value (constant offsets included) the optimiser cannot see into: it stays one scalar register
// pair instead of being re-associated into the lanes' side of an address
// (DDRL_STATE_IN_LDS: a translation unit whose kernels may hand these blocks weights that live in LDS -- the resident kernel of
// kernels_ddpg_pair.hip -- must not name an address space: there the pointer keeps its provenance and the older offset trick)
__device__ __forceinline__ const float *opaque_base(const float *p) {
#if defined(DDRL_F64_DEBUG)
  return p;
#elif defined(DDRL_STATE_IN_LDS)
  return opaque(p);
#else
  typedef const float __attribute__((address_space(1))) *gptr;
  gptr q = (gptr)p;
  asm volatile("" : "+s"(q));
  return (const float *)q;
#endif
}
// A global access as UNIFORM base + 32-bit per-lane byte offset: the form the hardware takes directly (scalar base register pair +
// vector offset), so the lane's 64-bit address is never built on the vector ALU (a v_lshl_add_u64 per access otherwise -- the
// fp32 MFMAs run on the same ALU, every vector instruction saved is MFMA time).  ubase must be wave-uniform.
// The hardware form is only selected when the offset's zero-extension is visible in the basic block of the access: callers pass
// offsets through fresh_lane() at the top of the block that uses them (no instruction; it only keeps the value from being widened
// once, far away, into a 64-bit register pair that every access then adds to its base on the vector ALU).
__device__ __forceinline__ unsigned fresh_lane(unsigned lane_floats) {
  asm volatile("" : "+v"(lane_floats));
  return lane_floats;
}
__device__ __forceinline__ const float *lane_at(const float *ubase, unsigned lane_floats) {
  return (const float *)((const char *)ubase + lane_floats * (unsigned)sizeof(float));
}
__device__ __forceinline__ float *lane_at(float *ubase, unsigned lane_floats) {
  return (float *)((char *)ubase + lane_floats * (unsigned)sizeof(float));
}

// Cache warming without registers (EXPERIMENT switches DDRL_EXP_TOUCHW / DDRL_EXP_TOUCHSTATE only, never defined in the
// product): an LDS-DMA load of ONE dword per lane whose destination is a 256-byte dump area of LDS that nobody reads.
// Lane i's address names the 128-byte line to pull toward this CU, so one wave instruction covers 8 KiB of a
// contiguous array.  Tried for the weights of the coming passes and for the optimiser state ahead of the fused
// epilogue; both made the update SLOWER (507 vs 496 us and 615 vs 496 us per update): the
// launch is bound by the bytes a CU can pull in, not by the latency of single loads, and a touch is one more request.
// The statement is invisible to the compiler's s_waitcnt bookkeeping, which only makes its waits conservative
// (vector memory operations complete in issue order); nothing ever waits for the dump itself.
__device__ __forceinline__ void l2_touch(const float *p, float *dump_wave) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float *)dump_wave);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(p), "s"(dst) : "memory");
}
// touch every 128-byte line of `floats` contiguous floats, the lines dealt to the DDRL_NTHREADS threads of the team
__device__ __forceinline__ void l2_touch_block(const float *base, int floats, float *dump, int tid) {
  float *dw = dump + (tid >> 6) * 64;
  for (int i0 = 0; i0 < floats; i0 += 32 * DDRL_NTHREADS) {
    int i = i0 + 32 * tid;
    if (i >= floats) i = floats - 32 > 0 ? floats - 32 : 0;       // clamped, never out of the block
    l2_touch(base + i, dw);
  }
}

// Caffe ReLULayer (negative_slope 0.01 for hidden_layer_type 1, MLP.hpp:888-897), TanHLayer
__device__ __forceinline__ float act_fwd(int t, float x) {
  switch (t) {
    case 1: return x > 0.f ? x : kLeaky * x;
    case 2: return tanhf(x);
    case 3: return x > 0.f ? x : 0.f;
    default: return x;
  }
}
// in-place layers (top == bottom, MLP.hpp:799-803): the derivative is taken from the OUTPUT y
__device__ __forceinline__ float act_bwd(int t, float y, float dy) {
  switch (t) {
    case 1: return y > 0.f ? dy : kLeaky * dy;
    case 2: return dy * (1.f - y * y);
    case 3: return y > 0.f ? dy : 0.f;
    default: return dy;
  }
}

// The weight fragments one lane feeds to the MFMAs of one 16-column output tile: row n = nt*16 + c
// of W, columns kc*16 + 4g .. + 3 for every 16-deep k-chunk kc.  Requested in one burst so the
// L2/HBM latency is paid once per tile; callers request the NEXT tile's fragments (also across a
// layer boundary: weights do not depend on activations) before computing the current one.
struct BFrag { f32x4 v[DDRL_MAXKC]; float bias; float t[3]; };   // + the tile's bias: waiting for it separately would drain the prefetches (loads retire in order); t: short-tail operands (below)

__device__ __forceinline__ void load_bfrag(const float *__restrict__ W, const float *__restrict__ bias, int kp, int N, int nt, int lane, BFrag &b) {
  const int g = lane >> 4, c = lane & 15;
#ifdef DDRL_EXP_NOWLOAD
  b.bias = 0.01f * nt;
#pragma unroll
  for (int kc = 0; kc < DDRL_MAXKC; ++kc) b.v[kc] = (f32x4){0.01f * (float)lane, 0.02f, -0.01f * (float)kc, 0.03f};
  return;
#endif
  b.bias = bias[nt * 16 + c < N ? nt * 16 + c : N - 1];
  // tile-major storage: block (nt, kc) is contiguous and lane (c, g)'s float4 sits at c*kw + 4g, so
  // a full block is one 1-KiB wave load.  Pad rows (n >= N) exist in memory and hold zeros.
  // No divergent branch around a load (the compiler would fall back to s_waitcnt vmcnt(0) at the
  // join): the address is clamped into the block.
#pragma unroll
  for (int kc = 0; kc < DDRL_MAXKC; ++kc) {
    if (kc * 16 < kp) {                       // uniform (folds for compile-time shapes)
      const int kw = ddrl_tile_kw(kc, kp);
      const float *blk = W + ddrl_tile_base(nt, kc, kp) + c * kw;
#ifdef DDRL_EXP_WNT
      b.v[kc] = __builtin_nontemporal_load((const f32x4 *)(blk + (4 * g < kw ? 4 * g : kw - 4)));
#else
      b.v[kc] = *(const f32x4 *)(blk + (4 * g < kw ? 4 * g : kw - 4));   // k >= kp: finite filler, meets a zero A column
#endif
    }
  }
}


// The activation fragments of this wave's 16 rows for a whole layer input: row 16w + c, columns
// kc*16 + 4g .. + 3 for every 16-deep k-chunk.  Columns >= kp are zeroed HERE (one LDS-side select per
// layer), so the weight fragments they meet need no masking: masking a prefetched global load would
// force a wait for it right where it is issued.
struct AFrag { f32x4 v[DDRL_MAXKC]; float t[3]; };
struct WCol { float v[DDRL_MAXKC][4]; float t[3]; };     // the weight fragments of an input-gradient tile (load_wcol)

__device__ __forceinline__ void load_afrag(const float *__restrict__ Xs, int ldx, int kp, int wave, int lane, AFrag &a) {
  const int g = lane >> 4, c = lane & 15;
  const float *row = Xs + (16 * ddrl_rt(wave) + c) * ldx;
#pragma unroll
  for (int kc = 0; kc < DDRL_MAXKC; ++kc)
    if (kc * 16 < kp) {
      const bool in = kc * 16 + 4 * g < kp;
      const f32x4 v = *(const f32x4 *)(row + (in ? kc * 16 + 4 * g : kp - 4));
      a.v[kc] = (kc * 16 + 16 <= kp || in) ? v : zero4();
    }
}

// one 16 x 16 tile: rows of this wave x output columns nt*16 .. +15.  Two accumulators over the
// parity of the k-chunk keep two independent MFMA chains in flight (dependent-issue latency of
// v_mfma_f32_16x16x4_f32 is 40 cycles against 32 for independent ones).
__device__ __forceinline__ f32x4 row_tile(const AFrag &a, const BFrag &b, int kp) {
  f32x4 acc0 = zero4(), acc1 = zero4();
#pragma unroll
  for (int kc = 0; kc < DDRL_MAXKC; kc += 2) {
    if (kc * 16 < kp) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc0 = mfma16(a.v[kc][e], b.v[kc][e], acc0);
        if ((kc + 1) * 16 < kp) acc1 = mfma16(a.v[kc + 1][e], b.v[kc + 1][e], acc1);
      }
    }
  }
  return acc0 + acc1;
}

// ---- SHORT TAIL (compile-time shapes) ----
// A contraction of length kp = 16 F + T (T = 4, 8 or 12) costs 4 (F + 1) MFMAs in the layout above: the last chunk's
// four products carry one valid k each in slot group g = 0 (or two, ...) and zeros elsewhere.  The short-tail forms give
// the last chunk its own slot mapping -- product e' of the tail, slot g <-> k = 16 F + 4 e' + g -- so that it takes T / 4
// products: kp / 4 in all, the minimum (100-deep: 25 instead of 28; 20-deep: 5 instead of 8).  The tail operands are
// single dwords: X[row c][16 F + 4 e' + g] from LDS, W[n c][16 F + 4 e' + g] (a contiguous 256-B wave access of the
// [16 x T] block) or, for input gradients, W[16 F + 4 e' + g][k c].  The k-sum of a tail runs over the four slots of one
// product instead of over four products: the same terms in another order (results move in the last bit).
__host__ __device__ constexpr int st_full(int kp) { return kp / 16; }
__host__ __device__ constexpr int st_ntail(int kp) { return (kp - 16 * (kp / 16)) / 4; }

__device__ __forceinline__ void load_afrag_st(const float *__restrict__ Xs, int ldx, int kp, int wave, int lane, AFrag &a) {
  const int g = lane >> 4, c = lane & 15;
  const float *row = Xs + (16 * ddrl_rt(wave) + c) * ldx;
#pragma unroll
  for (int kc = 0; kc < DDRL_MAXKC; ++kc)
    if (kc < st_full(kp)) a.v[kc] = *(const f32x4 *)(row + kc * 16 + 4 * g);
#pragma unroll
  for (int e = 0; e < 3; ++e)
    if (e < st_ntail(kp)) a.t[e] = row[16 * st_full(kp) + 4 * e + g];
}

__device__ __forceinline__ void load_bfrag_st(const float *__restrict__ W, const float *__restrict__ bias, int kp, int N, int nt, int lane, BFrag &b) {
  const int g = lane >> 4, c = lane & 15;
#ifdef DDRL_EXP_NOWLOAD
  b.bias = 0.01f * nt;
#pragma unroll
  for (int kc = 0; kc < DDRL_MAXKC; ++kc) b.v[kc] = (f32x4){0.01f * (float)lane, 0.02f, -0.01f * (float)kc, 0.03f};
  b.t[0] = b.t[1] = b.t[2] = 0.01f;
  return;
#endif
  b.bias = *lane_at(bias, fresh_lane((unsigned)(nt * 16 + c < N ? nt * 16 + c : N - 1)));
  const unsigned lo = fresh_lane((unsigned)(c * 16 + 4 * g));
  // scalar bases the compiler cannot fold back into the lanes' offsets (opaque), one per 4 KiB of the strip: every load is then
  // base register pair + lane offset + immediate
  const float *sb0 = opaque_base(W + ddrl_tile_base(nt, 0, kp)), *sb1 = st_full(kp) > 4 ? opaque_base(W + ddrl_tile_base(nt, 4, kp)) : sb0;
#pragma unroll
  for (int kc = 0; kc < DDRL_MAXKC; ++kc)
    if (kc < st_full(kp))
      b.v[kc] = *(const f32x4 *)lane_at((kc < 4 ? sb0 : sb1) + (ddrl_tile_base(nt, kc, kp) - ddrl_tile_base(nt, kc < 4 ? 0 : 4, kp)), lo);
  const int tw = kp - 16 * st_full(kp);
  const unsigned lt = st_ntail(kp) > 0 ? fresh_lane((unsigned)(c * tw + g)) : 0u;
  const float *sbt = st_ntail(kp) > 0 ? opaque_base(W + ddrl_tile_base(nt, st_full(kp), kp)) : sb0;
#pragma unroll
  for (int e = 0; e < 3; ++e)
    if (e < st_ntail(kp)) b.t[e] = *lane_at(sbt + 4 * e, lt);
}

__device__ __forceinline__ f32x4 row_tile_st(const AFrag &a, const BFrag &b, int kp) {
  f32x4 acc0 = zero4(), acc1 = zero4();
#pragma unroll
  for (int kc = 0; kc < DDRL_MAXKC; kc += 2) {
    if (kc < st_full(kp)) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc0 = mfma16(a.v[kc][e], b.v[kc][e], acc0);
        if (kc + 1 < st_full(kp)) acc1 = mfma16(a.v[kc + 1][e], b.v[kc + 1][e], acc1);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 3; ++e)
    if (e < st_ntail(kp)) { if (e & 1) acc0 = mfma16(a.t[e], b.t[e], acc0); else acc1 = mfma16(a.t[e], b.t[e], acc1); }
  return acc0 + acc1;
}

// input gradients: the contraction runs over the layer's N outputs (np = roundup(N, 4) = 16 F + T)
__device__ __forceinline__ void load_wcol_st(const float *__restrict__ W, int kp, int N, int kt, int lane, WCol &b) {
  const int g = lane >> 4, c = lane & 15;
#ifdef DDRL_EXP_NOWLOAD
#pragma unroll
  for (int nc = 0; nc < DDRL_MAXKC; ++nc)
#pragma unroll
    for (int j = 0; j < 4; ++j) b.v[nc][j] = 0.01f * (float)(lane + j) - 0.02f * (float)kt;
  b.t[0] = b.t[1] = b.t[2] = 0.01f;
  return;
#endif
  const int np = ddrl_roundup(N, 4);
  const int kw = ddrl_tile_kw(kt, kp);
  const int cc = c < kw ? c : kw - 1;
  const unsigned lo = fresh_lane((unsigned)(4 * g * kw + cc)), lt = fresh_lane((unsigned)(g * kw + cc));
#pragma unroll
  for (int nc = 0; nc < DDRL_MAXKC; ++nc)
    if (nc < st_full(np)) {
      const float *sb = opaque_base(W + ddrl_tile_base(nc, kt, kp));      // (see load_bfrag_st)
#pragma unroll
      for (int j = 0; j < 4; ++j) b.v[nc][j] = *lane_at(sb + j * kw, lo);
    }
  const float *sbt = st_ntail(np) > 0 ? opaque_base(W + ddrl_tile_base(st_full(np), kt, kp)) : W;
#pragma unroll
  for (int e = 0; e < 3; ++e)
    if (e < st_ntail(np)) b.t[e] = *lane_at(sbt + 4 * e * kw, lt);
}

__device__ __forceinline__ f32x4 dx_tile_st(const AFrag &a, const WCol &b, int N) {
  const int np = ddrl_roundup(N, 4);
  f32x4 acc0 = zero4(), acc1 = zero4();
#pragma unroll
  for (int nc = 0; nc < DDRL_MAXKC; nc += 2) {
    if (nc < st_full(np)) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc0 = mfma16(a.v[nc][e], b.v[nc][e], acc0);
        if (nc + 1 < st_full(np)) acc1 = mfma16(a.v[nc + 1][e], b.v[nc + 1][e], acc1);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 3; ++e)
    if (e < st_ntail(np)) { if (e & 1) acc0 = mfma16(a.t[e], b.t[e], acc0); else acc1 = mfma16(a.t[e], b.t[e], acc1); }
  return acc0 + acc1;
}

// Y[rows of this wave][N] = act(X W^T + b) for one layer.  Xs / Ys: LDS, 64 rows; columns
// [N, ldy) of Ys are zeroed.  Three weight-fragment sets rotate: the loads for tile nt + 2 are issued
// before tile nt is computed (one tile of MFMAs, ~900 cycles, does not cover an L2 round trip under
// load).  The rotation also runs across the layer boundary: the last iteration requests tiles 0
// and 1 of the NEXT layer (weights do not depend on activations) into b0 / b1, where the next call
// expects them (`preloaded`), so a layer no longer starts with a cold wait.  nextW == nullptr: no
// next layer.
struct FwdNext { const float *W, *bias; int kp, N; };

__device__ __forceinline__ void fwd_layer(const float *__restrict__ Xs, int ldx, int kp,
                                          const float *__restrict__ W, const float *__restrict__ bias,
                                          int N, int act, float *__restrict__ Ys, int ldy, int wave, int lane,
                                          BFrag &b0, BFrag &b1, bool preloaded, const FwdNext nx) {
  const int g = lane >> 4, c = lane & 15;
  const int NT = (N + 15) >> 4;
  W = opaque(W);
  bias = opaque(bias);
  const float *nW = nx.W ? opaque(nx.W) : nullptr, *nb = nx.W ? opaque(nx.bias) : nullptr;
  const int nNT = (nx.N + 15) >> 4;
  AFrag a;
  load_afrag(Xs, ldx, kp, wave, lane, a);
  BFrag b2;
  if (!preloaded) {
    load_bfrag(W, bias, kp, N, 0, lane, b0);
    if (1 < NT) load_bfrag(W, bias, kp, N, 1, lane, b1);
  }
  float *yrow = Ys + (16 * wave + 4 * g) * ldy;
  auto store = [&](int nt, const f32x4 acc, const float bv) {
    const int n = nt * 16 + c;
    if (n < ldy) {
#pragma unroll
      for (int r = 0; r < 4; ++r) yrow[r * ldy + n] = n < N ? act_fwd(act, acc[r] + bv) : 0.f;
    }
  };
#ifdef DDRL_FWD_SETS4
  // Four weight-fragment sets in rotation: the loads for tile nt + 3 are issued before tile nt is computed.  Under load
  // (a second team streaming its optimiser state through the same CU) a fragment load takes ~3.5 k cycles, two tiles of
  // MFMAs (~1.8 k) did not cover it: the forward passes ran at the loads' pace.
  {
    BFrag b3;
    if (2 < NT) load_bfrag(W, bias, kp, N, 2, lane, b2);
    if (3 < NT) load_bfrag(W, bias, kp, N, 3, lane, b3);
#pragma unroll 1
    for (int nt = 0; nt < NT; nt += 4) {
      const bool tail = nt + 4 >= NT;
      store(nt, row_tile(a, b0, kp), b0.bias);
      if (nt + 4 < NT) load_bfrag(W, bias, kp, N, nt + 4, lane, b0);
      else if (tail && nW) load_bfrag(nW, nb, nx.kp, nx.N, 0, lane, b0);
      if (nt + 1 < NT) store(nt + 1, row_tile(a, b1, kp), b1.bias);
      if (nt + 5 < NT) load_bfrag(W, bias, kp, N, nt + 5, lane, b1);
      else if (tail && nW && 1 < nNT) load_bfrag(nW, nb, nx.kp, nx.N, 1, lane, b1);
      if (nt + 2 < NT) store(nt + 2, row_tile(a, b2, kp), b2.bias);
      if (nt + 6 < NT) load_bfrag(W, bias, kp, N, nt + 6, lane, b2);
      if (nt + 3 < NT) store(nt + 3, row_tile(a, b3, kp), b3.bias);
      if (nt + 7 < NT) load_bfrag(W, bias, kp, N, nt + 7, lane, b3);
    }
    return;
  }
#endif
  // a real loop: fully unrolled, the scheduler hoists every tile's weight loads and spills
#pragma unroll 1
  for (int nt = 0; nt < NT; nt += 3) {
    const bool tail = nt + 3 >= NT;            // last iteration: b0 / b1 are handed to the next layer
    if (nt + 2 < NT) load_bfrag(W, bias, kp, N, nt + 2, lane, b2);
    store(nt, row_tile(a, b0, kp), b0.bias);
    if (nt + 3 < NT) load_bfrag(W, bias, kp, N, nt + 3, lane, b0);
    else if (tail && nW) load_bfrag(nW, nb, nx.kp, nx.N, 0, lane, b0);
    if (nt + 1 < NT) store(nt + 1, row_tile(a, b1, kp), b1.bias);
    if (nt + 4 < NT) load_bfrag(W, bias, kp, N, nt + 4, lane, b1);
    else if (tail && nW && 1 < nNT) load_bfrag(nW, nb, nx.kp, nx.N, 1, lane, b1);
    if (nt + 2 < NT) store(nt + 2, row_tile(a, b2, kp), b2.bias);
  }
}

// dX[rows of this wave][cols] = dY W for the column tiles kt in [kt0, kt1), kept in registers
// (acc[kt - kt0]).  dYs: LDS, columns [N, roundup(N,4)) hold zeros.  B operand of tile (nc, kt): lane
// (c, g) supplies W[nc*16 + 4g + j][kt*16 + c], four dword loads inside one contiguous 1-KiB block.

__device__ __forceinline__ void load_wcol(const float *__restrict__ W, int kp, int N, int kt, int lane, WCol &b) {
  const int g = lane >> 4, c = lane & 15;
#ifdef DDRL_EXP_NOWLOAD
#pragma unroll
  for (int nc = 0; nc < DDRL_MAXKC; ++nc)
#pragma unroll
    for (int j = 0; j < 4; ++j) b.v[nc][j] = 0.01f * (float)(lane + j) - 0.02f * (float)kt;
  return;
#endif
  const int kw = ddrl_tile_kw(kt, kp);
  const int cc = c < kw ? c : kw - 1;                  // clamp: columns >= kp are discarded by the caller
#pragma unroll
  for (int nc = 0; nc < DDRL_MAXKC; ++nc)
    if (nc * 16 < N) {
      const float *blk = W + ddrl_tile_base(nc, kt, kp) + 4 * g * kw + cc;   // pad rows (n >= N) hold zeros
#pragma unroll
#ifdef DDRL_EXP_WNT
      for (int j = 0; j < 4; ++j) b.v[nc][j] = __builtin_nontemporal_load(blk + j * kw);
#else
      for (int j = 0; j < 4; ++j) b.v[nc][j] = blk[j * kw];
#endif
    }
}

__device__ __forceinline__ f32x4 dx_tile(const AFrag &a, const WCol &b, int N) {
  f32x4 acc0 = zero4(), acc1 = zero4();
#pragma unroll
  for (int nc = 0; nc < DDRL_MAXKC; nc += 2) {
    if (nc * 16 < N) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc0 = mfma16(a.v[nc][e], b.v[nc][e], acc0);
        if ((nc + 1) * 16 < N) acc1 = mfma16(a.v[nc + 1][e], b.v[nc + 1][e], acc1);
      }
    }
  }
  return acc0 + acc1;
}

template <bool ST = false>     // ST: the short-tail forms (compile-time shapes)
__device__ __forceinline__ void bwd_dx(const float *__restrict__ dYs, int ldy, int N,
                                       const float *__restrict__ W, int kp, int kt0, int kt1,
                                       f32x4 (&acc)[DDRL_MAXKC], int wave, int lane) {
  W = opaque(W);
  AFrag a;
  if (ST) {
    // tile slot i of this wave's part is column tile kt0 + part + DDRL_PARTS i (one part: kt0 + i); a slot beyond the last
    // tile repeats it (its accumulator is never stored)
    load_afrag_st(dYs, ldy, ddrl_roundup(N, 4), wave, lane, a);
    const int part = ddrl_part(wave), ns = ddrl_slots(kt1 - kt0);
    auto tile = [&](int i) { const int kt = kt0 + part + DDRL_PARTS * i; return kt < kt1 ? kt : kt1 - 1; };
    WCol b0, b1;
    if (0 < ns) load_wcol_st(W, kp, N, tile(0), lane, b0);
#pragma unroll
    for (int i = 0; i < DDRL_MAXKC; i += 2) {
      if (i < ns) {
        if (i + 1 < ns) load_wcol_st(W, kp, N, tile(i + 1), lane, b1);
        acc[i] = dx_tile_st(a, b0, N);
        __builtin_amdgcn_sched_barrier(0);
        if (i + 1 < ns) {
          if (i + 2 < ns) load_wcol_st(W, kp, N, tile(i + 2), lane, b0);
          acc[i + 1] = dx_tile_st(a, b1, N);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    return;
  }
  load_afrag(dYs, ldy, ddrl_roundup(N, 4), wave, lane, a);
#ifdef DDRL_DX_SETS3
  {
    // three weight-column sets: tile kt + 2 is requested before tile kt is computed (see fwd_layer)
    WCol b[3];
    if (kt0 < kt1) load_wcol(W, kp, N, kt0, lane, b[0]);
    if (kt0 + 1 < kt1) load_wcol(W, kp, N, kt0 + 1, lane, b[1]);
#pragma unroll
    for (int i = 0; i < DDRL_MAXKC; ++i) {
      const int kt = kt0 + i;
      if (kt < kt1) {
        if (kt + 2 < kt1) load_wcol(W, kp, N, kt + 2, lane, b[(i + 2) % 3]);
        acc[i] = dx_tile(a, b[i % 3], N);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    return;
  }
#endif
  WCol b0, b1;
  if (kt0 < kt1) load_wcol(W, kp, N, kt0, lane, b0);
#pragma unroll
  for (int i = 0; i < DDRL_MAXKC; i += 2) {
    const int kt = kt0 + i;
    if (kt < kt1) {
      // acc[] needs static indices, so this loop is unrolled; the scheduling barriers keep the
      // loads of later tiles from being hoisted above the products of this one (spills otherwise)
      if (kt + 1 < kt1) load_wcol(W, kp, N, kt + 1, lane, b1);
      acc[i] = dx_tile(a, b0, N);
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < kt1) {
        if (kt + 2 < kt1) load_wcol(W, kp, N, kt + 2, lane, b0);
        acc[i + 1] = dx_tile(a, b1, N);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

// Everything the fused optimiser epilogue needs for one net of one agent (pointers already
// offset to the agent's block).
struct OptCtx {
  float *w, *m, *v, *wt, *fisher;
  const float *fhat, *wstar;
  // where the first moment is LOADED from.  With beta1 == 0 (agent.momentum 0 / 2, MLP.hpp:160-173) the new moment is
  // 1 * g + 0 * m: the old one is never needed, so the load is pointed at the weights' own lines (already on their way
  // for this tile: served by the L1, no HBM traffic) and its finite value is multiplied by zero.  No branch, no second
  // instantiation; 4 of the optimiser's 28 B/param of reads are gone.
  const float *mld;
  float lr_t, tau, omtau, decay, ewc_coef, fisher_beta, nsample;
  float beta1, beta2, omb1, omb2, eps;
  float *dump;       // LDS dump area for l2_touch (null: none)
  int handoff;       // the state is read by ANOTHER workgroup later in this launch (multi-update launches): sc1 stores
};

// 1 / (sqrt(v) + eps) on the hardware's 1-ulp v_sqrt_f32 / v_rcp_f32 instead of the ~25-instruction
// IEEE expansions of sqrtf and '/': v >= 0 and the divisor is >= eps = 1e-8, well inside the normal
// range, and a denormal v (where v_sqrt_f32 is inexact) is below eps^2 and changes nothing.
__device__ __forceinline__ float adam_rdenom(float v, float eps) {
#ifdef DDRL_F64_DEBUG
  return 1.0 / (sqrt(v) + eps);
#else
  return __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + eps);
#endif
}

// NT: optimiser state other than the weights is touched once per update and is streamed past the
// caches (nontemporal) so that it does not evict the weights the eight passes re-read -- the DDPG
// population, 1.2 GB of state per launch.  NFAC runs ~100 full-batch Adam steps per update on a state
// that fits the L2: there the ordinary cached accesses are the right ones (NT = false).
template <bool NT> __device__ __forceinline__ f32x4 stream_ld(const float *p) {
#ifndef DDRL_EXP_NONT
  if (NT) return __builtin_nontemporal_load((const f32x4 *)p);
#endif
  return *(const f32x4 *)p;
}
template <bool NT> __device__ __forceinline__ void stream_st(float *p, f32x4 x) {
#ifndef DDRL_EXP_NONT
  if (NT) { __builtin_nontemporal_store(x, (f32x4 *)p); return; }
#endif
  *(f32x4 *)p = x;
}

// Stores that another workgroup of the same launch will read (MI355X_MICROARCH.md, inter-workgroup
// visibility): `sc1` stores leave the XCD's L2 by themselves, so the hand-off needs no L2 write-back
// (`buffer_wbl2`) at its release -- only every storing wave's vmcnt(0), the barrier and the flag.  16-byte
// sc1 stores run at the plain rate; the 4-byte ones (biases, counters) are few.
__device__ __forceinline__ void store_sc1(float *p, f32x4 x) {
#ifdef DDRL_F64_DEBUG
  *(f32x4 *)p = x;     // (the debug build runs one update per launch: nothing is handed to another workgroup)
#else
  asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(x) : "memory");
#endif
}
__device__ __forceinline__ void store_sc1(float *p, float x) {
  __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Optimiser state of 4 consecutive parameters, requested before the products that produce their
// gradient so the HBM latency hides behind the MFMAs.
struct OptRegs { f32x4 w, m, v, wt, fisher, fhat, wstar; };

// EWC: the net has Fisher / anchor arrays (DODevMLP.hpp:604-707).  TGT: the net has a target twin
// that is soft-updated in the same pass (MLP.hpp:437-452).
// `at` maps an array's (uniform) base pointer to this lane's element: base + i, or lane_at(base + u, l) (element index split
// into its wave-uniform and per-lane parts)
struct AtIndex { int i; __device__ __forceinline__ const float *operator()(const float *b) const { return b + i; }
                 __device__ __forceinline__ float *operator()(float *b) const { return b + i; }
                 __device__ __forceinline__ AtIndex fresh() const { return *this; } };
struct AtSplit { int u; unsigned l; __device__ __forceinline__ const float *operator()(const float *b) const { return lane_at(b + u, l); }
                 __device__ __forceinline__ float *operator()(float *b) const { return lane_at(b + u, l); }
                 __device__ __forceinline__ AtSplit fresh() const { return AtSplit{u, fresh_lane(l)}; } };

template <bool EWC, bool TGT, bool NT = true, class AT>
__device__ __forceinline__ void opt_load4_at(const OptCtx &o, const AT at0, OptRegs &r) {
#if defined(DDRL_EXP_NOOPTMEM) || defined(DDRL_EXP_NOOPTLD)
  r.w = r.m = r.v = r.wt = r.fisher = r.fhat = r.wstar = (f32x4){1.5f, 1.f, 2.f, 3.f};
  return;
#endif
  const AT at = at0.fresh();
  r.w = *(const f32x4 *)at(o.w);
  r.m = stream_ld<NT>(at(o.mld));
  r.v = stream_ld<NT>(at((const float *)o.v));
  if (TGT) r.wt = *(const f32x4 *)at((const float *)o.wt);
  if (EWC) {
    r.fisher = stream_ld<NT>(at((const float *)o.fisher));
    r.fhat = stream_ld<NT>(at(o.fhat));
    r.wstar = stream_ld<NT>(at(o.wstar));
  }
}
template <bool EWC, bool TGT, bool NT = true>
__device__ __forceinline__ void opt_load4(const OptCtx &o, int i, OptRegs &r) { opt_load4_at<EWC, TGT, NT>(o, AtIndex{i}, r); }

// Four parameters: Fisher EMA (DODevMLP.hpp:651), EWC penalty (DODevMLP.hpp:702; ewc_coef is 0 and
// fhat / wstar are all-zero until the first ewc_setup), L2 decay (caffe SGDSolver::Regularize; decay
// <= 0 = off), Adam (caffe AdamSolver::ComputeUpdateValue + Net::Update) and the target net's soft
// update (MLP.hpp:449-450): one read and one write of each array per update, no branches.
// A per-agent scalar of the optimiser context, as the wave-uniform value it is.  Besides keeping it in a scalar register,
// this stands between the context's field and the vector expression it scales: a scalar load that feeds a vector splat
// directly is widened by the compiler into a 16-byte load spanning the neighbouring fields, after which the context can
// no longer be split into registers and lives in scratch memory (seen in wide_dw_kernel: 11 stores + 8 loads per workgroup).
__device__ __forceinline__ float uni(float x) {
#ifdef DDRL_F64_DEBUG
  return x;
#else
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
#endif
}

template <bool EWC, bool TGT, bool NT = true, bool PRED = true, class AT>     // PRED false: every lane stores (its index is a parameter's or the dump's)
__device__ __forceinline__ void opt_step4_at(const OptCtx &o, const AT at0, f32x4 grad, const OptRegs &r, bool ok) {
  f32x4 w = r.w, m, v, f, wt;
  if (EWC) {
    f = r.fisher * uni(o.fisher_beta) + (grad * grad) * uni(o.nsample);
    grad += uni(o.ewc_coef) * r.fhat * (w - r.wstar);
  }
  grad += fmaxf(uni(o.decay), 0.f) * w;
  m = uni(o.omb1) * grad + uni(o.beta1) * r.m;
  v = uni(o.omb2) * (grad * grad) + uni(o.beta2) * r.v;
  const float lr_t = uni(o.lr_t), eps = uni(o.eps);
#pragma unroll
  for (int e = 0; e < 4; ++e) w[e] -= lr_t * (m[e] * adam_rdenom(v[e], eps));
  if (TGT) wt = uni(o.tau) * w + uni(o.omtau) * r.wt;
#if defined(DDRL_EXP_NOOPTMEM) || defined(DDRL_EXP_NOOPTST)
  ok = ok && (w[0] + wt[1] + m[2] + v[3] + (EWC ? f[0] : 0.f) == 12345.678f);
#endif
  if (!PRED || ok) {                          // stores only: nothing waits on them
    const AT at = at0.fresh();
    if (o.handoff) {                          // uniform
      if (EWC) store_sc1(at(o.fisher), f);
      store_sc1(at(o.m), m);
      store_sc1(at(o.v), v);
      store_sc1(at(o.w), w);
      if (TGT) store_sc1(at(o.wt), wt);
    } else {
      const AT at = at0.fresh();
      if (EWC) stream_st<NT>(at(o.fisher), f);
      stream_st<NT>(at(o.m), m);
      stream_st<NT>(at(o.v), v);
      *(f32x4 *)at(o.w) = w;
      if (TGT) *(f32x4 *)at(o.wt) = wt;
    }
  }
}
template <bool EWC, bool TGT, bool NT = true, bool PRED = true>
__device__ __forceinline__ void opt_step4(const OptCtx &o, int i, f32x4 grad, const OptRegs &r, bool ok) {
  opt_step4_at<EWC, TGT, NT, PRED>(o, AtIndex{i}, grad, r, ok);
}

// scalar form (biases)
template <bool EWC, bool TGT>
__device__ __forceinline__ void opt_step(const OptCtx &o, int i, float grad) {
  float w = o.w[i];
  if (EWC) {
    o.fisher[i] = o.fisher[i] * o.fisher_beta + (grad * grad) * o.nsample;
    grad += o.ewc_coef * o.fhat[i] * (w - o.wstar[i]);
  }
  grad += fmaxf(o.decay, 0.f) * w;
  const float m = o.omb1 * grad + o.beta1 * o.mld[i];
  const float v = o.omb2 * (grad * grad) + o.beta2 * o.v[i];
  o.m[i] = m;
  o.v[i] = v;
  w -= o.lr_t * (m * adam_rdenom(v, o.eps));
  o.w[i] = w;
  if (TGT) o.wt[i] = o.tau * w + o.omtau * o.wt[i];
}

// the same step with the state already in registers (requested at the start of the phase)
struct OptRegs1 { float w, m, v, wt, fisher, fhat, wstar; };
template <bool EWC, bool TGT>
__device__ __forceinline__ void opt_load1(const OptCtx &o, int i, OptRegs1 &r) {
  r.w = o.w[i]; r.m = o.mld[i]; r.v = o.v[i];
  if (TGT) r.wt = o.wt[i];
  if (EWC) { r.fisher = o.fisher[i]; r.fhat = o.fhat[i]; r.wstar = o.wstar[i]; }
}
template <bool EWC, bool TGT>
__device__ __forceinline__ void opt_step1(const OptCtx &o, int i, float grad, const OptRegs1 &r, bool ok) {
  float w = r.w, f = 0.f;
  if (EWC) {
    f = r.fisher * o.fisher_beta + (grad * grad) * o.nsample;
    grad += o.ewc_coef * r.fhat * (w - r.wstar);
  }
  grad += fmaxf(o.decay, 0.f) * w;
  const float m = o.omb1 * grad + o.beta1 * r.m;
  const float v = o.omb2 * (grad * grad) + o.beta2 * r.v;
  w -= o.lr_t * (m * adam_rdenom(v, o.eps));
  if (ok) {
    if (o.handoff) {
      if (EWC) store_sc1(o.fisher + i, f);
      store_sc1(o.m + i, m);
      store_sc1(o.v + i, v);
      store_sc1(o.w + i, w);
      if (TGT) store_sc1(o.wt + i, o.tau * w + o.omtau * r.wt);
    } else {
      if (EWC) o.fisher[i] = f;
      o.m[i] = m;
      o.v[i] = v;
      o.w[i] = w;
      if (TGT) o.wt[i] = o.tau * w + o.omtau * r.wt;
    }
  }
}

// dW[N x K] = dY^T X  (+ db = dY^T 1), each element handed to the optimiser straight from the MFMA
// accumulator.  The product is issued transposed (A = X, B = dY) so that lane (c, g) ends up with
// dW[n = nt*16 + c][k = kt*16 + 4g .. + 3]: four consecutive parameters = one 16-byte access to
// each state array.  dYs [64 x N] and Xs [64 x K] in LDS; rows >= batch must be zero in dYs.
// Output tiles (nt, kt) are dealt to the waves in contiguous runs (balanced to within one tile).
// Pad columns [K, kp) take part: their gradient is exactly 0 and their state stays 0.
// element index of lane (c, g)'s 4 parameters in output tile p of a [N x kp] weight block; ok = the
// lane owns real parameters there (else the index is clamped to a valid, aligned element whose
// loaded state is ignored)
__device__ __forceinline__ int dw_elem(int p, int total, int KT, int N, int kp, int offw, int c, int g, bool &ok) {
  const int pc = p < total ? p : total - 1;
  const int nt = pc / KT, kt = pc - nt * KT;
  const int n = nt * 16 + c, k = kt * 16 + 4 * g;
  ok = p < total && n < N && k < kp;
  return ok ? offw + ddrl_tile_base(nt, kt, kp) + c * ddrl_tile_kw(kt, kp) + 4 * g : offw;
}

// one 16 x 16 tile of dW^T = X^T dY (64-deep), result lane layout: see bwd_dw_opt
__device__ __forceinline__ f32x4 dw_tile(const float *__restrict__ dYs, int ldy, int N,
                                         const float *__restrict__ Xs, int ldx, int kp,
                                         int p, int total, int KT, int c, int g) {
  const int pc = p < total ? p : total - 1;
  const int nt = pc / KT, kt = pc - nt * KT;
  const int bn = nt * 16 + c, ak = kt * 16 + c;
  // clamped operand columns: the products they feed land in elements that are never stored
  const float *xa = Xs + (ak < kp ? ak : kp - 1) + 4 * g * ldx;
  const float *yb = dYs + (bn < N ? bn : N - 1) + 4 * g * ldy;
  f32x4 acc0 = zero4(), acc1 = zero4();     // two independent chains: 32- instead of 40-cycle issue
#pragma unroll
  for (int mc = 0; mc < 4; mc += 2)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = mc * 16 + j;
      acc0 = mfma16(xa[row * ldx], yb[row * ldy], acc0);
      acc1 = mfma16(xa[(row + 16) * ldx], yb[(row + 16) * ldy], acc1);
    }
  return acc0 + acc1;
}

// the same tile with ALL 32 LDS operands requested before the first product (the compiler, left alone, puts every pair
// of products behind its own ds_read2 + wait: eight LDS round trips per tile instead of one).  Used by the static
// schedule of bwd_dw_opt_ct, where the tile's products are no longer hidden behind a vmcnt(0) wait.
__device__ __forceinline__ f32x4 dw_tile_pre(const float *__restrict__ dYs, int ldy, int N,
                                             const float *__restrict__ Xs, int ldx, int kp,
                                             int p, int total, int KT, int c, int g) {
  const int pc = p < total ? p : total - 1;
  const int nt = pc / KT, kt = pc - nt * KT;
  const int bn = nt * 16 + c, ak = kt * 16 + c;
  const float *xa = Xs + (ak < kp ? ak : kp - 1) + 4 * g * ldx;
  const float *yb = dYs + (bn < N ? bn : N - 1) + 4 * g * ldy;
  float xv[16], yv[16];
#pragma unroll
  for (int mc = 0; mc < 4; ++mc)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xv[mc * 4 + j] = xa[(mc * 16 + j) * ldx];
      yv[mc * 4 + j] = yb[(mc * 16 + j) * ldy];
    }
  __builtin_amdgcn_sched_barrier(0);
  f32x4 acc0 = zero4(), acc1 = zero4();
#pragma unroll
  for (int mc = 0; mc < 4; mc += 2)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc0 = mfma16(xv[mc * 4 + j], yv[mc * 4 + j], acc0);
      acc1 = mfma16(xv[(mc + 1) * 4 + j], yv[(mc + 1) * 4 + j], acc1);
    }
  return acc0 + acc1;
}

// db[n] = sum over the 64 rows of dY[:, n]: four interleaved partial sums so the LDS reads pipeline
__device__ __forceinline__ float bias_grad(const float *__restrict__ dYs, int ldy, int n) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
  for (int row = 0; row < DDRL_MB; row += 4) {
    s0 += dYs[row * ldy + n];
    s1 += dYs[(row + 1) * ldy + n];
    s2 += dYs[(row + 2) * ldy + n];
    s3 += dYs[(row + 3) * ldy + n];
  }
  return (s0 + s1) + (s2 + s3);
}

template <bool EWC, bool TGT, bool SETS3 = false>
__device__ __forceinline__ void bwd_dw_opt(const float *__restrict__ dYs, int ldy, int N,
                                           const float *__restrict__ Xs, int ldx, int kp,
                                           int offw, int offb, const OptCtx &o,
                                           int wave, int lane, int tid) {
  const int g = lane >> 4, c = lane & 15;
  const int NT = (N + 15) >> 4, KT = (kp + 15) >> 4;
  const int total = NT * KT;
  const int per = (total + DDRL_NWAVES - 1) / DDRL_NWAVES;   // tile slots per wave, the same for every wave
  const int p0 = wave * per;
  // Two statically named state sets: the next tile's optimiser state is requested BEFORE this
  // tile's stores are issued (vector memory operations retire in issue order, so a load queued
  // behind the stores would wait for them) and is never copied (a copy would wait for the load).
  // Every wave walks `per` slots and every lane loads in every slot (clamped addresses); only the
  // stores are predicated.  The two-tile body is kept as a real loop (unrolling it
  // fully makes the scheduler hoist every tile's loads and spill); inside it the outstanding
  // loads are counted exactly (s_waitcnt vmcnt(N), not 0).
  // the bias state first (every lane loads, clamped; N <= DDRL_MAXW <= 128 columns): its latency
  // hides behind the whole tile loop
  static_assert(DDRL_MAXW <= DDRL_NTHREADS, "one thread per bias");
  OptRegs1 rb;
  const int nb = tid < N ? tid : N - 1;
  opt_load1<EWC, TGT>(o, offb + nb, rb);
#ifdef DDRL_EXP_TOUCHSTATE
  if (o.dump) {
    // this wave's slice of the layer (tiles p0 .. p0 + per - 1 are contiguous in the tile-major layout) toward the caches
    int nt0 = (p0 < total ? p0 : total - 1) / KT, kt0 = (p0 < total ? p0 : total - 1) - nt0 * KT;
    const int lo = offw + ddrl_tile_base(nt0, kt0, kp);
    const int pe = p0 + per < total ? p0 + per : total;
    int hi = offw + ddrl_wsize(N, kp);
    if (pe < total) { const int nt1 = pe / KT, kt1 = pe - nt1 * KT; hi = offw + ddrl_tile_base(nt1, kt1, kp); }
    float *dw = o.dump + wave * 64;
    for (int i0 = lo; i0 < hi; i0 += 32 * 64) {
      int i = i0 + 32 * lane;
      if (i >= hi) i = hi - 32 > lo ? hi - 32 : lo;
      l2_touch(o.w + i, dw); l2_touch(o.m + i, dw); l2_touch(o.v + i, dw);
      if (TGT) l2_touch(o.wt + i, dw);
      if (EWC) { l2_touch(o.fisher + i, dw); l2_touch(o.fhat + i, dw); l2_touch(o.wstar + i, dw); }
    }
  }
#endif
  if (SETS3) {
    // Three sets in rotation: the state of tile s + 2 is requested before tile s is computed (a tile
    // of products + Adam is ~1.2k cycles, an HBM round trip 4-6k).  Used where the registers allow
    // it: the callers park the layer's input-gradient accumulators in LDS for the duration.
    OptRegs r0, r1, r2;
    bool ok0 = false, ok1 = false, ok2 = false;
    int i0 = dw_elem(p0, total, KT, N, kp, offw, c, g, ok0), i1 = 0, i2 = 0;
    opt_load4<EWC, TGT>(o, i0, r0);
    if (1 < per) { i1 = dw_elem(p0 + 1, total, KT, N, kp, offw, c, g, ok1); opt_load4<EWC, TGT>(o, i1, r1); }
#pragma unroll 1
    for (int s = 0; s < per; s += 3) {
      if (s + 2 < per) { i2 = dw_elem(p0 + s + 2, total, KT, N, kp, offw, c, g, ok2); opt_load4<EWC, TGT>(o, i2, r2); }
      {
        const f32x4 acc = dw_tile(dYs, ldy, N, Xs, ldx, kp, p0 + s, total, KT, c, g);
        opt_step4<EWC, TGT>(o, i0, acc, r0, ok0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (s + 1 < per) {
        if (s + 3 < per) { i0 = dw_elem(p0 + s + 3, total, KT, N, kp, offw, c, g, ok0); opt_load4<EWC, TGT>(o, i0, r0); }
        const f32x4 acc = dw_tile(dYs, ldy, N, Xs, ldx, kp, p0 + s + 1, total, KT, c, g);
        opt_step4<EWC, TGT>(o, i1, acc, r1, ok1);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (s + 2 < per) {
        if (s + 4 < per) { i1 = dw_elem(p0 + s + 4, total, KT, N, kp, offw, c, g, ok1); opt_load4<EWC, TGT>(o, i1, r1); }
        const f32x4 acc = dw_tile(dYs, ldy, N, Xs, ldx, kp, p0 + s + 2, total, KT, c, g);
        opt_step4<EWC, TGT>(o, i2, acc, r2, ok2);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    OptRegs r0, r1;
    bool ok0 = false, ok1 = false;
    int i0 = dw_elem(p0, total, KT, N, kp, offw, c, g, ok0), i1 = 0;
    opt_load4<EWC, TGT>(o, i0, r0);
  #pragma unroll 1
    for (int s = 0; s < per; s += 2) {
      if (s + 1 < per) { i1 = dw_elem(p0 + s + 1, total, KT, N, kp, offw, c, g, ok1); opt_load4<EWC, TGT>(o, i1, r1); }
      {
        const f32x4 acc = dw_tile(dYs, ldy, N, Xs, ldx, kp, p0 + s, total, KT, c, g);
        opt_step4<EWC, TGT>(o, i0, acc, r0, ok0);
      }
      if (s + 1 < per) {
        if (s + 2 < per) { i0 = dw_elem(p0 + s + 2, total, KT, N, kp, offw, c, g, ok0); opt_load4<EWC, TGT>(o, i0, r0); }
        const f32x4 acc = dw_tile(dYs, ldy, N, Xs, ldx, kp, p0 + s + 1, total, KT, c, g);
        opt_step4<EWC, TGT>(o, i1, acc, r1, ok1);
      }
    }
  }
  // bias gradient: column sums of dY (waves whose lanes all sit beyond N skip the LDS reads)
  if (wave * 64 < N) opt_step1<EWC, TGT>(o, offb + nb, bias_grad(dYs, ldy, nb), rb, tid < N);
}

// ---- the same epilogue with COMPILE-TIME trip counts (SNet shapes) ----
// What the loop above loses: its vector-memory operations sit under conditions (`s + 2 < per`, predicated stores, the
// tile-slot guards), so the compiler's s_waitcnt bookkeeping has to assume the path on which the fewest of them were
// issued -- in the built kernel every tile waited with vmcnt(0), i.e. for the prefetches of the NEXT tiles as well, and
// the two / three state sets bought nothing (round 3: the ISA showed it, and two sets measured like three).  Here the
// tile slots are fully unrolled into one static schedule in which every slot issues exactly 7 (4) loads and 5 (3) stores
// by ALL lanes:
//  * lanes that own no parameter -- pad rows (n >= N) of the last strip, and in a short last k-block (kw < 16) the lanes
//    beyond it -- take the place of a lane that does (the strip's last real row, lane group 0): same address, same state,
//    the gradient fetched from that lane, the very same values stored a second time.  Identical addresses inside one wave
//    access are one request, so the pad rows of the tile-major layout (12 of 112 rows of a 100-wide layer, 15 of 16 of a
//    1-wide one: a quarter more state than there are parameters) are no longer READ either;
//  * a wave's slots beyond the last tile are always its LAST ones: their loads go to a clamped address, products and stores
//    are skipped under one uniform branch, which only blurs the counts of what follows the loop.
// The waits inside the loop are then exact: before a tile's Adam step only ITS loads are awaited.
// requested: sets 0 .. NSETS-2 were requested by the caller (bwd_net: ahead of the layer's input-gradient pass).
template <int NSETS> struct OptCarry { OptRegs r[NSETS]; };      // (a slot's element index / source lane are recomputed at its use: cheaper than keeping them)

// element index of lane (c, g)'s four parameters in tile slot p, and the lane whose gradient they take (itself, or
// lane (c, 0) beyond a short k-block)
__device__ __forceinline__ AtSplit dw_elem_u(int p, int total, int KT, int N, int kp, int offw, int c, int g, int &src) {
  const int pc = p < total ? p : total - 1;
  const int nt = pc / KT, kt = pc - nt * KT;
  const int kw = ddrl_tile_kw(kt, kp);
  const int gg = 4 * g < kw ? g : 0;
  const int cc = nt * 16 + c < N ? c : N - 1 - nt * 16;      // pad rows of the last strip: the strip's last real row
  src = 16 * gg + cc;
  return AtSplit{offw + ddrl_tile_base(nt, kt, kp), (unsigned)(cc * kw + 4 * gg)};      // the tile (uniform) + the lane's place in it
}

template <bool EWC, bool TGT, int NSETS>
__device__ __forceinline__ void dw_opt_request(int N, int kp, int offw, const OptCtx &o, int wave, int lane, OptCarry<NSETS> &cy) {
  const int g = lane >> 4, c = lane & 15;
  const int NT = (N + 15) >> 4, KT = (kp + 15) >> 4, total = NT * KT;
  const int per = (total + DDRL_NWAVES - 1) / DDRL_NWAVES, p0 = wave * per;
#pragma unroll
  for (int q = 0; q < NSETS - 1; ++q)
    if (q < per) {
      int src;
      opt_load4_at<EWC, TGT>(o, dw_elem_u(p0 + q, total, KT, N, kp, offw, c, g, src), cy.r[q]);
    }
}

template <bool EWC, bool TGT, int NSETS>
__device__ __forceinline__ void bwd_dw_opt_ct(const float *__restrict__ dYs, int ldy, int N,
                                              const float *__restrict__ Xs, int ldx, int kp,
                                              int offw, int offb, const OptCtx &o,
                                              int wave, int lane, int tid, OptCarry<NSETS> &cy, bool requested) {
  const int g = lane >> 4, c = lane & 15;
  const int NT = (N + 15) >> 4, KT = (kp + 15) >> 4, total = NT * KT;
  const int per = (total + DDRL_NWAVES - 1) / DDRL_NWAVES, p0 = wave * per;
  OptRegs1 rb;
  const int nb = tid < N ? tid : N - 1;
  opt_load1<EWC, TGT>(o, offb + nb, rb);
  if (!requested) dw_opt_request<EWC, TGT, NSETS>(N, kp, offw, o, wave, lane, cy);
#pragma unroll
  for (int s = 0; s < per; ++s) {
    if (s + NSETS - 1 < per) {
      int src;
      opt_load4_at<EWC, TGT>(o, dw_elem_u(p0 + s + NSETS - 1, total, KT, N, kp, offw, c, g, src), cy.r[(s + NSETS - 1) % NSETS]);
    }
    const int q = s % NSETS;
    if (p0 + s < total) {                      // uniform; false only for a wave's trailing slots
#ifdef DDRL_DW_PRE
      f32x4 acc = dw_tile_pre(dYs, ldy, N, Xs, ldx, kp, p0 + s, total, KT, c, g);
#else
      f32x4 acc = dw_tile(dYs, ldy, N, Xs, ldx, kp, p0 + s, total, KT, c, g);
#endif
      int src;
      const AtSplit i = dw_elem_u(p0 + s, total, KT, N, kp, offw, c, g, src);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = __shfl(acc[e], src);
      opt_step4_at<EWC, TGT, true, false>(o, i, acc, cy.r[q], true);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (wave * 64 < N) opt_step1<EWC, TGT>(o, offb + nb, bias_grad(dYs, ldy, nb), rb, tid < N);
}
