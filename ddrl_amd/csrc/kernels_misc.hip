// kernels_misc.hip -- replay ring, bit-exact replay index sampling, batch-1 policy inference and
// the EWC bookkeeping kernels (gfx950).
#include "bn_device.h"

// ------------------------------------------------------------------------------------------
// Replay index sampling: DeepQNAg::sample_transition (DeepQNAg.hpp:275-288) draws
//   std::uniform_int_distribution<int>(0, len-1)(mt19937)
// B times.  One wave per agent keeps the agent's mt19937 (624 words + position) in LDS, twists it
// 64 words at a time (the recurrence x[i] <- f(x[i], x[i+1], x[i+397]) only reaches >= 64 words
// back/forward, so a 64-wide block is dependence-free), and evaluates libstdc++-11's Lemire
// reduction for 64 draws at once; a rejected draw (probability len / 2^32) consumes one engine
// output and is replayed for the same slot, exactly as the sequential loop does.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

// The same launch prepares the update's Adam step sizes: caffe AdamSolver uses t = iter + 1 and
// correction = sqrt(1 - beta2^t) / (1 - beta1^t); evaluated in fp64 because 1 - 0.999^t loses 4
// digits in fp32 at small t.  draw == 0: indices come from the host, only the step sizes are made.
extern "C" __global__ void __launch_bounds__(64)
replay_draw_kernel(uint32_t *__restrict__ mt, int *__restrict__ pos, const int *__restrict__ rep_len,
                   int B, int nupd, int min_len, const uint8_t *__restrict__ mask, int *__restrict__ idx_out, int draw,
                   double beta1, double beta2, const float *__restrict__ lr_a, const float *__restrict__ lr_q,
                   const int *__restrict__ iter_a, const int *__restrict__ iter_q,
                   float *__restrict__ lrt_a, float *__restrict__ lrt_q, int *__restrict__ upd_done) {
  __shared__ uint32_t st[624];
  const int agent = blockIdx.x, lane = threadIdx.x;
  if (upd_done && lane == 0) upd_done[agent] = 0;       // progress counter of the coming launch (ddpg_update_kernel)
  if (mask && !mask[agent]) return;
  const uint32_t n = (uint32_t)rep_len[agent];
  if ((int)n < min_len || (int)n < B) return;       // the guard of DeepQNAg.hpp:338: no draw consumed
  // nupd consecutive updates in one launch (the fupd loop, DeepQNAg.hpp:341): step sizes for
  // t = iter + 1 .. iter + nupd, and nupd * B indices drawn back to back like nupd calls would
  for (int j = lane; j < 2 * nupd; j += 64) {
    const int q = j & 1, u = j >> 1;
    const int t = (q ? iter_q : iter_a)[agent] + 1 + u;
    const double corr = sqrt(1.0 - pow(beta2, (double)t)) / (1.0 - pow(beta1, (double)t));
    (q ? lrt_q : lrt_a)[(size_t)agent * nupd + u] = (float)((double)(q ? lr_q : lr_a)[agent] * corr);
  }
  if (!draw) return;
  const int count = B * nupd;
  uint32_t *gst = mt + (size_t)agent * 624;
  for (int i = lane; i < 624; i += 64) st[i] = gst[i];
  int p = pos[agent];
  __syncthreads();
  const uint32_t thresh = (uint32_t)(0u - n) % n;
  int *out = idx_out + (size_t)agent * count;
  int drawn = 0;
  while (drawn < count) {
    if (p >= 624) {
      for (int b = 0; b < 624; b += 64) {
        const int i = b + lane;
        uint32_t x = 0;
        if (i < 624) {
          const uint32_t y = (st[i] & 0x80000000u) | (st[(i + 1) % 624] & 0x7fffffffu);
          x = st[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        __syncthreads();
        if (i < 624) st[i] = x;
        __syncthreads();
      }
      p = 0;
    }
    int chunk = 624 - p;
    if (chunk > count - drawn) chunk = count - drawn;
    if (chunk > 64) chunk = 64;
    uint64_t prod = 0;
    bool rej = false;
    if (lane < chunk) {
      prod = (uint64_t)mt_temper(st[p + lane]) * (uint64_t)n;
      const uint32_t low = (uint32_t)prod;
      rej = low < thresh;                            // low < range && low < threshold, threshold < range
    }
    const unsigned long long bal = __ballot(rej);
    const int ok = bal ? (__ffsll((long long)bal) - 1) : chunk;
    if (lane < ok) out[drawn + lane] = (int)(prod >> 32);
    drawn += ok;
    p += bal ? ok + 1 : ok;
  }
  __syncthreads();
  for (int i = lane; i < 624; i += 64) gst[i] = st[i];
  if (lane == 0) pos[agent] = p;
}

// std::mt19937(seed) seeding, one thread per agent
extern "C" __global__ void mt_seed_kernel(uint32_t *__restrict__ mt, int *__restrict__ pos,
                                          const uint32_t *__restrict__ seeds, int G) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= G) return;
  uint32_t *s = mt + (size_t)a * 624;
  uint32_t x = seeds[a];
  s[0] = x;
  for (int i = 1; i < 624; ++i) { x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i; s[i] = x; }
  pos[a] = 624;
}

// ------------------------------------------------------------------------------------------
// Replay ring push (std::deque push_back / pop_front at capacity, DeepQNAg.hpp:159-163)
// row = [s (S), a (A), s2 (S), r, goal]
// ------------------------------------------------------------------------------------------
extern "C" __global__ void replay_append_kernel(float *__restrict__ rep, int *__restrict__ rep_len,
                                                int *__restrict__ rep_head, int cap, int rowlen, int S, int A,
                                                const float *__restrict__ s, const float *__restrict__ a,
                                                const float *__restrict__ r, const float *__restrict__ s2,
                                                const uint8_t *__restrict__ goal, const uint8_t *__restrict__ mask) {
  const int agent = blockIdx.x, t = threadIdx.x;
  if (mask && !mask[agent]) return;
  const int len = rep_len[agent], head = rep_head[agent];
  int slot;
  if (len >= cap) slot = head; else { slot = head + len; if (slot >= cap) slot -= cap; }
  float *row = rep + ((size_t)agent * cap + slot) * rowlen;
  for (int e = t; e < rowlen; e += blockDim.x) {
    float v = 0.f;
    if (e < S) v = s[(size_t)agent * S + e];
    else if (e < S + A) v = a[(size_t)agent * A + e - S];
    else if (e < 2 * S + A) v = s2[(size_t)agent * S + e - S - A];
    else if (e == 2 * S + A) v = r[agent];
    else if (e == 2 * S + A + 1) v = goal[agent] ? 1.f : 0.f;
    row[e] = v;
  }
  __syncthreads();
  if (t == 0) {
    if (len >= cap) rep_head[agent] = head + 1 >= cap ? 0 : head + 1;
    else rep_len[agent] = len + 1;
  }
}

// ------------------------------------------------------------------------------------------
// Batch-1 policy inference for every agent: MLP::computeOut on ann_testing (DeepQNAg.hpp:126).
// One wave per agent; lane n owns output n of each layer; activations ping-pong through LDS.
// ------------------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(64)
infer_kernel(const NetDesc nd, const float *__restrict__ w, const float *__restrict__ states,
             float *__restrict__ actions, const float *__restrict__ in_gate, const float *__restrict__ out_gate,
             const GateMode gm) {
  __shared__ __align__(16) float buf[2][DDRL_WIDE_MAXW + 8];
  const int agent = blockIdx.x, lane = threadIdx.x;
  const float *wa = w + (size_t)agent * nd.stride();
  const int d0 = nd.d(0);
  for (int k = lane; k < nd.kp(0); k += 64) {
    float x = k < d0 ? states[(size_t)agent * d0 + k] : 0.f;
    if (in_gate && k < d0)
      x *= ddrl_gate_factor(in_gate[(size_t)agent * d0 + k], gm.st_bern, gm.st_scale, gm.seed, gm.draw, agent, GS_INFER_S, 0, k);
    if (nd.bn(0) && k < nd.bn_c(0)) x = bn_test_value(nd, wa, 0, k, x);       // ann_testing is a TEST-phase copy (DeepQNAg.hpp:221-222)
    buf[0][k] = x;
  }
  __syncthreads();
  int cur = 0;
  for (int l = 0; l < nd.L(); ++l) {
    const int kp = nd.kp(l), N = nd.d(l + 1);
    const bool last = l == nd.L() - 1;
    const int act = last ? nd.out_act() : nd.hidden_act();
    for (int n = lane; n < ddrl_roundup(N, 4); n += 64) {
      float y = 0.f;
      if (n < N) {
        const float *wl = wa + nd.offw(l);
        float s = 0.f;
        for (int k = 0; k < kp; k += 4) {
          const f32x4 wv = *(const f32x4 *)(wl + ddrl_widx(n, k, kp));
          const f32x4 xv = *(const f32x4 *)(&buf[cur][k]);
          s = fmaf(wv[0], xv[0], s); s = fmaf(wv[1], xv[1], s);
          s = fmaf(wv[2], xv[2], s); s = fmaf(wv[3], xv[3], s);
        }
        y = act_fwd(act, s + wa[nd.offb(l) + n]);
        if (nd.bn(l + 1)) y = bn_test_value(nd, wa, l + 1, n, y);
      }
      if (last) {
        if (n < N) {
          if (out_gate)
            y *= ddrl_gate_factor(out_gate[(size_t)agent * N + n], gm.ac_bern, gm.ac_scale, gm.seed, gm.draw, agent, GS_INFER_OUT, 0, n);
          actions[(size_t)agent * N + n] = y;
        }
      } else buf[cur ^ 1][n] = y;
    }
    __syncthreads();
    cur ^= 1;
  }
}

// ------------------------------------------------------------------------------------------
// EWC bookkeeping (DODevMLP.hpp:604-758), one workgroup per agent
// ------------------------------------------------------------------------------------------
// update_best_param_previous_task: best <- w for flagged agents (the score logic runs on the host)
extern "C" __global__ void snapshot_kernel(float *__restrict__ dst, const float *__restrict__ src, int stride,
                                           const uint8_t *__restrict__ flag) {
  const int agent = blockIdx.x;
  if (flag && !flag[agent]) return;
  const size_t off = (size_t)agent * stride;
  for (int i = threadIdx.x * 4; i < stride; i += blockDim.x * 4)
    *(f32x4 *)(dst + off + i) = *(const f32x4 *)(src + off + i);
}

// MLP::copyWeightsFrom (MLP.hpp:651-676) for every flagged agent, from device memory: row `agent` of
// flat [G x ld] (ld == 0: one row for all) holds the parameters in learnable_params() order
// [W1 (out x in), b1, W2, b2, ...]; they go to the tile-major image.  Pad elements are not touched.
extern "C" __global__ void __launch_bounds__(256)
load_flat_kernel(NetDesc nd, float *__restrict__ w, const float *__restrict__ flat, int ld, const uint8_t *__restrict__ flag) {
  const int agent = blockIdx.x;
  if (flag && !flag[agent]) return;
  const float *src = flat + (size_t)agent * ld;
  float *dst = w + (size_t)agent * nd.stride();
  int base = 0;
  for (int l = 0; l < nd.L(); ++l) {
    const int din = nd.d(l), dout = nd.d(l + 1), nw = din * dout;
    for (int i = threadIdx.x; i < nw; i += blockDim.x) {
      const int n = i / din, c = i - n * din;
      dst[nd.offw(l) + ddrl_widx(n, c, nd.kp(l))] = src[base + i];
    }
    for (int i = threadIdx.x; i < dout; i += blockDim.x) dst[nd.offb(l) + i] = src[base + nw + i];
    base += nw + dout;
  }
}

// ewc_setup (DODevMLP.hpp:604-633): wstar <- best; w <- wstar; fhat <- (F / nbr) / max(F / nbr);
// F <- 0 (the reference drops the vector: fisher = nullptr), nbr <- 0
extern "C" __global__ void __launch_bounds__(256)
ewc_setup_kernel(const NetDesc nd, float *__restrict__ w, const float *__restrict__ best, float *__restrict__ wstar,
                 float *__restrict__ fisher, float *__restrict__ fhat, float *__restrict__ fisher_nbr,
                 int stride, const uint8_t *__restrict__ flag) {
  __shared__ float red[256];
  const int agent = blockIdx.x, t = threadIdx.x;
  if (flag && !flag[agent]) return;
  const size_t off = (size_t)agent * stride;
  const float nbr = fisher_nbr[agent];
  float mx = 0.f;
  for (int i = t; i < stride; i += 256) {
    const float b = best[off + i];
    wstar[off + i] = b;
    // copyWeightsFrom(best, ignore_null_lr = true) (DODevMLP.hpp:616): a tower's running statistics (lr_mult 0) are not in the
    // vector and stay as they are
    bool stats = false;
    for (int r = 0; r <= nd.L(); ++r)
      stats = stats || (nd.bn(r) && i >= nd.bn_off(r) && i < nd.bn_off(r) + nd.bn_stats_floats(r));
    if (!stats) w[off + i] = b;
    const float f = fisher[off + i] / nbr;
    mx = fmaxf(mx, f);
  }
  red[t] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (t < s) red[t] = fmaxf(red[t], red[t + s]); __syncthreads(); }
  mx = red[0];
  for (int i = t; i < stride; i += 256) {
    const float f = fisher[off + i] / nbr;
    fhat[off + i] = f / mx;
    fisher[off + i] = 0.f;
  }
  __syncthreads();
  if (t == 0) fisher_nbr[agent] = 0.f;
}

// ewc_cost (DODevMLP.hpp:660-684): sum_k fhat_k (wstar_k - w_k)^2 per agent (scaled on the host)
extern "C" __global__ void __launch_bounds__(256)
ewc_cost_kernel(const float *__restrict__ w, const float *__restrict__ wstar, const float *__restrict__ fhat,
                int stride, float *__restrict__ out) {
  __shared__ float red[256];
  const int agent = blockIdx.x, t = threadIdx.x;
  const size_t off = (size_t)agent * stride;
  float s = 0.f;
  for (int i = t; i < stride; i += 256) {
    const float x = wstar[off + i] - w[off + i];
    s += x * x * fhat[off + i];
  }
  red[t] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) { if (t < k) red[t] += red[t + k]; __syncthreads(); }
  if (t == 0) out[agent] = red[0];
}

// ---- launch wrappers (called from the host side in pop.hip) ----
extern "C" hipError_t replay_draw_launch(uint32_t *mt, int *pos, const int *rep_len, int G, int B, int nupd,
                                         const uint8_t *mask, int *idx_out, int draw, double beta1, double beta2,
                                         const float *lr_a, const float *lr_q, const int *iter_a, const int *iter_q,
                                         float *lrt_a, float *lrt_q, int *upd_done, hipStream_t st) {
  hipLaunchKernelGGL(replay_draw_kernel, dim3(G), dim3(64), 0, st, mt, pos, rep_len, B, nupd, 250, mask, idx_out, draw,
                     beta1, beta2, lr_a, lr_q, iter_a, iter_q, lrt_a, lrt_q, upd_done);
  return hipGetLastError();
}
extern "C" hipError_t mt_seed_launch(uint32_t *mt, int *pos, const uint32_t *seeds, int G, hipStream_t st) {
  hipLaunchKernelGGL(mt_seed_kernel, dim3((G + 63) / 64), dim3(64), 0, st, mt, pos, seeds, G);
  return hipGetLastError();
}
extern "C" hipError_t replay_append_launch(float *rep, int *len, int *head, int G, int cap, int rowlen, int S, int A,
                                           const float *s, const float *a, const float *r, const float *s2,
                                           const uint8_t *goal, const uint8_t *mask, hipStream_t st) {
  hipLaunchKernelGGL(replay_append_kernel, dim3(G), dim3(64), 0, st, rep, len, head, cap, rowlen, S, A, s, a, r, s2, goal, mask);
  return hipGetLastError();
}
extern "C" hipError_t infer_launch(const NetDesc *nd, const float *w, const float *states, float *actions,
                                   const float *in_gate, const float *out_gate, const GateMode *gm, int G, hipStream_t st) {
  hipLaunchKernelGGL(infer_kernel, dim3(G), dim3(64), 0, st, *nd, w, states, actions, in_gate, out_gate, *gm);
  return hipGetLastError();
}
extern "C" hipError_t snapshot_launch(float *dst, const float *src, int stride, const uint8_t *flag, int G, hipStream_t st) {
  hipLaunchKernelGGL(snapshot_kernel, dim3(G), dim3(256), 0, st, dst, src, stride, flag);
  return hipGetLastError();
}
extern "C" hipError_t load_flat_launch(const NetDesc *nd, float *w, const float *flat, int ld, const uint8_t *flag, int G, hipStream_t st) {
  hipLaunchKernelGGL(load_flat_kernel, dim3(G), dim3(256), 0, st, *nd, w, flat, ld, flag);
  return hipGetLastError();
}
extern "C" hipError_t ewc_setup_launch(const NetDesc *nd, float *w, const float *best, float *wstar, float *fisher, float *fhat,
                                       float *nbr, int stride, const uint8_t *flag, int G, hipStream_t st) {
  hipLaunchKernelGGL(ewc_setup_kernel, dim3(G), dim3(256), 0, st, *nd, w, best, wstar, fisher, fhat, nbr, stride, flag);
  return hipGetLastError();
}
extern "C" hipError_t ewc_cost_launch(const float *w, const float *wstar, const float *fhat, int stride, float *out,
                                      int G, hipStream_t st) {
  hipLaunchKernelGGL(ewc_cost_kernel, dim3(G), dim3(256), 0, st, w, wstar, fhat, stride, out);
  return hipGetLastError();
}
