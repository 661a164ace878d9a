// cmaes.hip -- CMA-ES policy search on gfx950 behind the ddrl_cmaes_* C ABI.
//
// What it replaces: the vendored c-cmaes, /root/reference/agent/cmaes/src/cmaes.c, as driven by
// CMAESAg (/root/reference/agent/cmaes/include/CMAESAg.hpp:50-253).
//
// Device (fp32 matrices, fp64 vectors/scalars):
//   cmaes_gemm_nt       X = m + sigma * DZ * B^T            (cmaes.c:645-652)        fp32 MFMA GEMM
//   cmaes_rank_kernel   Sorted_index                        (cmaes.c:2989-3000)      exact LDS sort
//   cmaes_mean/btv/bv/path_kernel   mean, BDz, ps, hsig, pc, sigma   (cmaes.c:826-898, 952)   fp64, all CUs
//   cmaes_jacobi_*      C = B diag(D^2) B^T, one-sided Jacobi in fp64 (throughput mode; cmaes.c:1885-2261 is the host path)
//   cmaes_pack_y        y_k = sqrt(w_k) (x_{k:lambda} - m_old) / sigma  (cmaes.c:985-989)
//   cmaes_gemm_nt       C <- a C + c1 pc pc^T + cmu Y Y^T   (cmaes.c:979-991)        fp32 MFMA GEMM
// Host (fp64): parameter defaults, the reference's LCG + Box-Muller stream (sequential by
// construction: a shuffle table indexed by the previous output), the lazily scheduled eigen-
// decomposition (Householder + QL, the JAMA routines c-cmaes uses).
#include "../../include/ddrl_pop.h"
#include "pop_internal.h"
#include "host_util.h"
#include "ckpt.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#define HIPCHK(expr)                                                                        \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) return ddrl_fail(DDRL_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// device scalar block (doubles)
enum { SC_SIGMA = 0, SC_SIGMA_C, SC_PSXPS, SC_HSIG, SC_FBEST, SC_N };

__device__ __forceinline__ f32x4 cm_mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// C[M x N] (+epilogue) = A[M x K] * Bt[N x K]^T, all row-major with row strides lda/ldb/ldc (floats,
// multiples of 4); K a multiple of 16 (operands are zero-padded by their producers).
// 64 x 64 tile per 256-thread workgroup, 16-deep k-steps staged through LDS, each wave a 32 x 32
// quadrant = 2 x 2 MFMA tiles of v_mfma_f32_16x16x4_f32.
// MODE 0 (sampling):    C[r][c] = mean[c] + sigma * acc
// MODE 1 (covariance):  C[r][c] = cc * C[r][c] + c1 * pc[r] * pc[c] + cmu * acc, with
//                       cc = (1 - c1 - cmu) + c1 * (1 - hsig) * ccumcov * (2 - ccumcov)   (cmaes.c:981-984)
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(256)
cmaes_gemm_nt(const float *__restrict__ A, int lda, const float *__restrict__ Bt, int ldb,
              float *__restrict__ C, int ldc, int M, int N, int K,
              const double *__restrict__ vec, const double *__restrict__ sc,
              double c1, double cmu, double ccumcov, int only_row) {
  __shared__ __align__(16) float As[64][20];   // 16 + 4 pad: rows stay 16-B aligned
  __shared__ __align__(16) float Bs[64][20];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int g = lane >> 4, c = lane & 15;
  const int bm = blockIdx.y * 64, bn = blockIdx.x * 64;
  const int wr = wave >> 1, wc = wave & 1;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int lr = tid >> 2, lk = (tid & 3) * 4;          // staging: one float4 of A and of Bt per thread
  const int ar = min(bm + lr, M - 1), br = min(bn + lr, N - 1);
  for (int k0 = 0; k0 < K; k0 += 16) {
    const f32x4 av = *(const f32x4 *)(A + (size_t)ar * lda + k0 + lk);
    const f32x4 bv = *(const f32x4 *)(Bt + (size_t)br * ldb + k0 + lk);
    __syncthreads();
    *(f32x4 *)&As[lr][lk] = av;
    *(f32x4 *)&Bs[lr][lk] = bv;
    __syncthreads();
    f32x4 a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = *(const f32x4 *)&As[wr * 32 + i * 16 + c][4 * g];
#pragma unroll
    for (int j = 0; j < 2; ++j) b[j] = *(const f32x4 *)&Bs[wc * 32 + j * 16 + c][4 * g];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = cm_mfma(a[i][e], b[j][e], acc[i][j]);
  }
  double cc = 0.;
  float sigma = 0.f;
  if (MODE == 0) sigma = (float)sc[SC_SIGMA];
  else cc = (1. - c1 - cmu) + c1 * (1. - sc[SC_HSIG]) * ccumcov * (2. - ccumcov);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = bm + wr * 32 + i * 16 + 4 * g + r, col = bn + wc * 32 + j * 16 + c;
        if (row < M && col < N) {
          if (MODE == 0) {
            const int orow = only_row >= 0 ? only_row : row;
            C[(size_t)orow * ldc + col] = (float)vec[col] + sigma * acc[i][j][r];
          } else {
            float *p = C + (size_t)row * ldc + col;
            *p = (float)(cc * (double)*p + c1 * vec[row] * vec[col] + cmu * (double)acc[i][j][r]);
          }
        }
      }
}

// ---------------------------------------------------------------------------------------------
// Sorted_index (cmaes.c:2989-3000): ascending insertion sort with a strict '<' break, i.e. among
// equal values the LATER index comes first.  Equivalent total order: (value ascending, index
// descending); sorted here with a bitonic network over (fp64 key, index) pairs in LDS (one workgroup).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool rank_after(unsigned long long ka, unsigned int ia, unsigned long long kb, unsigned int ib) {
  return ka > kb || (ka == kb && ia < ib);         // equal values: the later index sorts first
}

__global__ void __launch_bounds__(1024)
cmaes_rank_kernel(const double *__restrict__ f, int n, int npow2, int *__restrict__ index) {
  extern __shared__ unsigned long long keys[];     // [npow2] keys, then [npow2] u32 indices
  unsigned int *ids = (unsigned int *)(keys + npow2);
  const int tid = threadIdx.x;
  for (int i = tid; i < npow2; i += 1024) {
    unsigned long long k = ~0ull;
    if (i < n) {
      double v = f[i];
      if (v == 0.) v = 0.;                         // -0 == +0 for the reference's '<'
      unsigned long long u = (unsigned long long)__double_as_longlong(v);
      k = (u >> 63) ? ~u : (u | 0x8000000000000000ull);
    }
    keys[i] = k;
    ids[i] = i < n ? (unsigned int)i : 0u;         // pads: max key, smallest index => sort last
  }
  __syncthreads();
  for (int size = 2; size <= npow2; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < npow2 / 2; t += 1024) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool up = (lo & size) == 0;
        const unsigned long long a = keys[lo], b = keys[hi];
        const unsigned int ia = ids[lo], ib = ids[hi];
        if (rank_after(a, ia, b, ib) == up && !(a == b && ia == ib)) {
          keys[lo] = b; keys[hi] = a; ids[lo] = ib; ids[hi] = ia;
        }
      }
      __syncthreads();
    }
  for (int i = tid; i < n; i += 1024) index[i] = (int)ids[i];
}

// ---------------------------------------------------------------------------------------------
// The O(N mu + N^2) vector part of cmaes_UpdateDistribution (cmaes.c:825-898, 952) in fp64, spread over the chip
// (it was one 1024-thread workgroup: 1.04 ms = 85 % of a generation's GPU time at lambda = 4096):
//   cmaes_mean_kernel   flat-fitness sigma bump, best-ever, xmean = sum_k w_k x_{k:lambda}, BDz       ceil(N / 64) workgroups
//   cmaes_btv_kernel    z = D^-1 B^T BDz: column sums of B, coalesced across columns                    ceil(N / 64) workgroups
//   cmaes_bv_kernel     ps = (1 - cs) ps + sqrt(cs (2 - cs)) B z: one wave per row of B                 ceil(N / 4) workgroups
//   cmaes_path_kernel   |ps|^2, hsig, pc, sigma                                                          one workgroup
// Every sum runs in a fixed order (no atomics): results do not depend on scheduling.
// sc[SC_SIGMA] is read by the first three and rewritten by the last one only; the bumped value travels in sc[SC_SIGMA_C].
// ---------------------------------------------------------------------------------------------
struct CmaesVecArgs {
  int N, ldx, ldn, lambda, mu;
  double mueff, cs, damps, ccumcov, chiN, gen;
  const float *X, *B;
  const double *fit;
  const int *index;
  const double *weights;
  double *xmean, *xold, *ps, *pc, *BDz, *tmp, *D, *sc, *xbest;
};

// sigma after the flat-fitness bump (cmaes.c:826-831): every workgroup derives it from the same inputs
__device__ __forceinline__ double cmaes_sigma_bumped(const CmaesVecArgs &a) {
  double sigma = a.sc[SC_SIGMA];
  if (a.fit[a.index[0]] == a.fit[a.index[a.lambda / 2]]) sigma *= exp(0.2 + a.cs / a.damps);
  return sigma;
}

__global__ void __launch_bounds__(256)
cmaes_mean_kernel(const CmaesVecArgs a) {
  __shared__ double part[4][64];
  const int tid = threadIdx.x, c = tid & 63, q = tid >> 6;
  const int i = blockIdx.x * 64 + c, N = a.N;
  const double sigma = cmaes_sigma_bumped(a);
  const int ibest = a.index[0];
  const bool newbest = a.sc[SC_FBEST] > a.fit[ibest] || a.gen == 1.;       // best-ever solution (cmaes.c:834-841)
  // xmean_i = sum_k w_k x_{k:lambda, i}: rows k = q, q + 4, ... per quarter, the quarters added in order
  double m = 0.;
  if (i < N) {
    // eight rows in flight per thread (the rows are scattered by index: a dependent chain of single loads is latency-bound)
    double acc[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
    int k = q;
    for (; k + 28 < a.mu; k += 32) {
      float x[8]; double w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { x[u] = a.X[(size_t)a.index[k + 4 * u] * a.ldx + i]; w[u] = a.weights[k + 4 * u]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += w[u] * (double)x[u];
    }
    for (; k < a.mu; k += 4) acc[0] += a.weights[k] * (double)a.X[(size_t)a.index[k] * a.ldx + i];
    m = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  }
  part[q][c] = m;
  __syncthreads();
  if (q == 0 && i < N) {
    m = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
    const double old = a.xmean[i];
    a.xold[i] = old;
    a.xmean[i] = m;
    a.BDz[i] = sqrt(a.mueff) * (m - old) / sigma;                            // cmaes.c:846-852
    if (newbest) a.xbest[i] = (double)a.X[(size_t)ibest * a.ldx + i];
  }
  if (blockIdx.x == 0 && tid == 0) a.sc[SC_SIGMA_C] = sigma;                 // Adapt_C2 uses sigma before its final update
}

// z = D^-1 B^T BDz   (cmaes.c:855-862)
__global__ void __launch_bounds__(256)
cmaes_btv_kernel(const CmaesVecArgs a) {
  __shared__ double part[4][64];
  const int tid = threadIdx.x, c = tid & 63, q = tid >> 6;
  const int i = blockIdx.x * 64 + c, N = a.N;
  double s = 0.;
  if (i < N)
    for (int j = q; j < N; j += 4) s += (double)a.B[(size_t)j * a.ldn + i] * a.BDz[j];
  part[q][c] = s;
  __syncthreads();
  if (q == 0 && i < N) a.tmp[i] = ((part[0][c] + part[1][c]) + (part[2][c] + part[3][c])) / a.D[i];
}

// ps = (1 - cs) ps + sqrt(cs (2 - cs)) B z   (cmaes.c:878-886): one wave per row, lanes stride the row
__global__ void __launch_bounds__(256)
cmaes_bv_kernel(const CmaesVecArgs a) {
  const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6), N = a.N;
  if (i >= N) return;
  double s = 0.;
  for (int j = lane; j < N; j += 64) s += (double)a.B[(size_t)i * a.ldn + j] * a.tmp[j];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) a.ps[i] = (1. - a.cs) * a.ps[i] + sqrt(a.cs * (2. - a.cs)) * s;
}

// |ps|^2, hsig, pc, sigma   (cmaes.c:893-898, 952); the best value travels here too (the mean kernel's workgroups still read it)
__global__ void __launch_bounds__(1024)
cmaes_path_kernel(const CmaesVecArgs a) {
  __shared__ double red[1024];
  const int tid = threadIdx.x, N = a.N;
  double part = 0.;
  for (int i = tid; i < N; i += 1024) { const double p = a.ps[i]; part += p * p; }
  red[tid] = part;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
  const double psxps = red[0];
  const double hsig = (sqrt(psxps) / sqrt(1. - pow(1. - a.cs, 2. * a.gen)) / a.chiN < 1.4 + 2. / (N + 1)) ? 1. : 0.;
  for (int i = tid; i < N; i += 1024)
    a.pc[i] = (1. - a.ccumcov) * a.pc[i] + hsig * sqrt(a.ccumcov * (2. - a.ccumcov)) * a.BDz[i];
  if (tid == 0) {
    const int ibest = a.index[0];
    if (a.sc[SC_FBEST] > a.fit[ibest] || a.gen == 1.) a.sc[SC_FBEST] = a.fit[ibest];
    a.sc[SC_PSXPS] = psxps;
    a.sc[SC_HSIG] = hsig;
    a.sc[SC_SIGMA] = a.sc[SC_SIGMA_C] * exp(((sqrt(psxps) / a.chiN) - 1.) * a.cs / a.damps);   // cmaes.c:952
  }
}

// Y[i][k] = sqrt(w_k) (x_{k:lambda,i} - xold_i) / sigma  (N x mu16, zero padded): the rank-mu term
// sum_k w_k y_k y_k^T / sigma^2 (cmaes.c:985-989) becomes Y Y^T, one k-contiguous operand used on both
// sides of the product, which keeps C exactly symmetric in fp32
__global__ void cmaes_pack_y(const float *__restrict__ X, int ldx, const int *__restrict__ index,
                             const double *__restrict__ weights, const double *__restrict__ xold,
                             const double *__restrict__ sc, int N, int mu, int mu16, float *__restrict__ Y) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (k >= mu16 || i >= N) return;
  float y = 0.f;
  if (k < mu) y = (float)(sqrt(weights[k]) * ((double)X[(size_t)index[k] * ldx + i] - xold[i]) / sc[SC_SIGMA_C]);
  Y[(size_t)i * mu16 + k] = y;
}

// Throughput mode (ddrl_cmaes_set_option "device_rng"): D o z drawn on the device from a counter-based
// generator (Philox4x32-10 + Box-Muller), one counter per 4 values, so rows can be produced in any
// order on any rank.  NOT c-cmaes' stream: reproducible against itself only.
__device__ __forceinline__ void philox4x32(unsigned int c[4], unsigned int k0, unsigned int k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c[1] ^ k0, n2 = (unsigned int)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (unsigned int)p1; c[3] = (unsigned int)p0; c[0] = n0; c[2] = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
__global__ void cmaes_fill_dz_kernel(float *__restrict__ DZ, int ldn, const double *__restrict__ D, int rows, int N,
                                     unsigned int seed, unsigned int draw, int row_base) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;      // 4 columns per thread
  if (4 * q >= ldn || k >= rows) return;
  unsigned int c[4] = {(unsigned int)q, (unsigned int)(row_base + k), draw, 0u};
  philox4x32(c, seed, 0x5EEDu);
  float z[4];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float u1 = ((float)c[2 * p] + 1.0f) * 2.3283064e-10f;             // (0, 1]
    const float u2 = (float)c[2 * p + 1] * 2.3283064e-10f;
    const float r = sqrtf(-2.0f * logf(u1));
    z[2 * p] = r * cospif(2.0f * u2);
    z[2 * p + 1] = r * sinpif(2.0f * u2);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int i = 4 * q + e;
    DZ[(size_t)k * ldn + i] = i < N ? (float)(D[i] * (double)z[e]) : 0.f;
  }
}

// any |x_i| >= bound per individual (CMAESAg::is_feasible, CMAESAg.hpp:135-142)
__global__ void cmaes_infeasible_kernel(const float *__restrict__ X, int ldx, int N, float bound, int *__restrict__ flag) {
  const int k = blockIdx.x;
  int bad = 0;
  for (int i = threadIdx.x; i < N; i += blockDim.x) bad |= fabsf(X[(size_t)k * ldx + i]) >= bound;
  bad = __syncthreads_or(bad);
  if (threadIdx.x == 0) flag[k] = bad;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct ddrl_cmaes {
  int N = 0, lambda = 0, mu = 0, device = 0;
  int ldn = 0, mu16 = 0;                 // padded row strides (multiples of 16)
  double mueff, mucov, cs, ccumcov, ccov, damps, modulo, chiN;
  std::vector<double> weights;
  // reference random stream (cmaes.c:2421-2482)
  long aktseed = 0, aktrand = 0, rgrand[32];
  int flgstored = 0;
  double hold = 0;
  // schedule
  double gen = 0, countevals = 0, genOfEigensysUpdate = 0;
  int state = 0, eigen_uptodate = 1;
  int device_eigen = 0;                  // throughput mode: one-sided Jacobi on the GPU instead of the host JAMA routines
  int eigen_block = 1;                   // ... with blocks of rows as the tournament's players (cmaes_jacobi_block_round) instead of single rows
  int sweep_graph_block = 0;             // which of the two the captured sweep holds
  int eigen_persist = 1;                 // ... and all sweeps in one launch with barriers in memory (cmaes_jacobi_sweeps_fast) where the one-trip round serves
  int eigen_persist_timeouts = 0;        // launches that gave up at a barrier (and were finished round by round)
  int eigen_persist_wait_ms = 2000;      // a workgroup's patience at a hand-over
  int eigen_persist_drop = 0;            // (tests) workgroups left out of the launch
  int eigen_sweeps = 0;                  // sweeps the last device eigendecomposition took
  hipGraphExec_t sweep_graph = nullptr;  // one Jacobi sweep (N - 1 rounds) as a captured graph
  bool sweep_graph_failed = false;
  int eig_warm = 0;                      // d_W holds the previous eigenvectors: the next decomposition starts from them
  double *d_A = nullptr, *d_ev = nullptr, *d_W = nullptr, *d_Cs = nullptr;
  int *d_info = nullptr;
  int device_rng = 0;                    // throughput mode: Philox on the device instead of c-cmaes' stream
  unsigned int rng_seed = 0, rng_draw = 0;
  std::vector<double> D, B;              // host copies (fp64): D [N], B [N x N]
  // cmaes_TestForTermination inputs (cmaes.c:322-325, 347-349, 379-380, 817-836)
  std::vector<double> hist, fitcur, init_stds;
  double minEW = 0., maxEW = 0., dMaxSignifKond = 0., stopMaxFunEvals = 0., stopMaxIter = 0.;
  hipStream_t stream = nullptr;
  // device
  float *d_X = nullptr, *d_DZ = nullptr, *d_B = nullptr, *d_C = nullptr, *d_Y = nullptr;
  double *d_fit = nullptr;
  double *d_xmean = nullptr, *d_xold = nullptr, *d_ps = nullptr, *d_pc = nullptr, *d_BDz = nullptr, *d_tmp = nullptr,
         *d_D = nullptr, *d_w = nullptr, *d_sc = nullptr, *d_xbest = nullptr;
  int *d_index = nullptr, *d_flag = nullptr;
  std::vector<void *> allocs;
};

template <class T>
static int cm_alloc(ddrl_cmaes *h, T **p, size_t n) {
  void *q = nullptr;
  hipError_t e = hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T));
  if (e != hipSuccess) return ddrl_fail(DDRL_ENOMEM, "hipMalloc(%zu bytes): %s", n * sizeof(T), hipGetErrorString(e));
  h->allocs.push_back(q);
  e = hipMemsetAsync(q, 0, std::max<size_t>(n, 1) * sizeof(T), h->stream);
  if (e != hipSuccess) return ddrl_fail(DDRL_EHIP, "hipMemset: %s", hipGetErrorString(e));
  *p = (T *)q;
  return DDRL_OK;
}

// ATTRIBUTION.  rnd_start / rnd_uniform / rnd_gauss and eig::tridiag / eig::ql below follow, operation for operation,
// cmaes_random_Start / _Uniform / _Gauss and Householder2 / QLalgo2 of N. Hansen's c-cmaes (agent/cmaes/src/cmaes.c:
// 2421-2482 and 1986-2261 in the reference tree; (c) 1996, 2003, 2007 Nikolaus Hansen, dual-licensed Apache-2.0 /
// LGPL-2.1+, cmaes.c:14-26), the latter two being c-cmaes' port of JAMA's tred2 / tql2 (public domain, NIST / MathWorks,
// from EISPACK).  A bit-identical z stream and eigen schedule require the same operations in the same order, so
// these ~170 lines are a restatement by necessity (parity mode); the throughput mode uses this library's own device
// routines instead.


// cmaes_random_Start / _Uniform / _Gauss (cmaes.c:2421-2482)
static void rnd_start(ddrl_cmaes *t, unsigned long inseed) {
  t->flgstored = 0;
  while (inseed > 2e9) inseed /= 2;
  if (inseed < 1) inseed = 1;
  t->aktseed = (long)inseed;
  for (int i = 39; i >= 0; --i) {
    long tmp = t->aktseed / 127773;
    t->aktseed = 16807 * (t->aktseed - tmp * 127773) - 2836 * tmp;
    if (t->aktseed < 0) t->aktseed += 2147483647;
    if (i < 32) t->rgrand[i] = t->aktseed;
  }
  t->aktrand = t->rgrand[0];
}
static double rnd_uniform(ddrl_cmaes *t) {
  long tmp = t->aktseed / 127773;
  t->aktseed = 16807 * (t->aktseed - tmp * 127773) - 2836 * tmp;
  if (t->aktseed < 0) t->aktseed += 2147483647;
  tmp = t->aktrand / 67108865;
  t->aktrand = t->rgrand[tmp];
  t->rgrand[tmp] = t->aktseed;
  return (double)(t->aktrand) / (2.147483647e9);
}
static double rnd_gauss(ddrl_cmaes *t) {
  if (t->flgstored) { t->flgstored = 0; return t->hold; }
  double x1, x2, rquad;
  do {
    x1 = 2.0 * rnd_uniform(t) - 1.0;
    x2 = 2.0 * rnd_uniform(t) - 1.0;
    rquad = x1 * x1 + x2 * x2;
  } while (rquad >= 1 || rquad <= 0);
  const double fac = sqrt(-2.0 * log(rquad) / rquad);
  t->flgstored = 1;
  t->hold = fac * x1;
  return fac * x2;
}

// Symmetric eigendecomposition as c-cmaes does it: Householder tridiagonalisation followed by QL
// with implicit shifts, eigenvalues sorted ascending (cmaes.c:1951-2261; the JAMA tred2/tql2).
namespace eig {
static inline double hyp(double a, double b) {
  double r = 0;
  if (std::fabs(a) > std::fabs(b)) { r = b / a; r = std::fabs(a) * std::sqrt(1 + r * r); }
  else if (b != 0) { r = a / b; r = std::fabs(b) * std::sqrt(1 + r * r); }
  return r;
}
static void tridiag(int n, std::vector<double> &Vm, double *d, double *e) {
  auto V = [&](int i, int j) -> double & { return Vm[(size_t)i * n + j]; };
  for (int j = 0; j < n; ++j) d[j] = V(n - 1, j);
  for (int i = n - 1; i > 0; --i) {
    double scale = 0., h = 0.;
    for (int k = 0; k < i; ++k) scale += std::fabs(d[k]);
    if (scale == 0.) {
      e[i] = d[i - 1];
      for (int j = 0; j < i; ++j) { d[j] = V(i - 1, j); V(i, j) = 0.; V(j, i) = 0.; }
    } else {
      for (int k = 0; k < i; ++k) { d[k] /= scale; h += d[k] * d[k]; }
      double f = d[i - 1], g = std::sqrt(h);
      if (f > 0) g = -g;
      e[i] = scale * g;
      h -= f * g;
      d[i - 1] = f - g;
      for (int j = 0; j < i; ++j) e[j] = 0.;
      for (int j = 0; j < i; ++j) {
        f = d[j];
        V(j, i) = f;
        g = e[j] + V(j, j) * f;
        for (int k = j + 1; k <= i - 1; ++k) { g += V(k, j) * d[k]; e[k] += V(k, j) * f; }
        e[j] = g;
      }
      f = 0.;
      for (int j = 0; j < i; ++j) { e[j] /= h; f += e[j] * d[j]; }
      const double hh = f / (h + h);
      for (int j = 0; j < i; ++j) e[j] -= hh * d[j];
      for (int j = 0; j < i; ++j) {
        f = d[j]; g = e[j];
        for (int k = j; k <= i - 1; ++k) V(k, j) -= (f * e[k] + g * d[k]);
        d[j] = V(i - 1, j);
        V(i, j) = 0.;
      }
    }
    d[i] = h;
  }
  for (int i = 0; i < n - 1; ++i) {
    V(n - 1, i) = V(i, i);
    V(i, i) = 1.;
    const double h = d[i + 1];
    if (h != 0.) {
      for (int k = 0; k <= i; ++k) d[k] = V(k, i + 1) / h;
      for (int j = 0; j <= i; ++j) {
        double g = 0.;
        for (int k = 0; k <= i; ++k) g += V(k, i + 1) * V(k, j);
        for (int k = 0; k <= i; ++k) V(k, j) -= g * d[k];
      }
    }
    for (int k = 0; k <= i; ++k) V(k, i + 1) = 0.;
  }
  for (int j = 0; j < n; ++j) { d[j] = V(n - 1, j); V(n - 1, j) = 0.; }
  V(n - 1, n - 1) = 1.;
  e[0] = 0.;
}
static void ql(int n, double *d, double *e, std::vector<double> &Vm) {
  auto V = [&](int i, int j) -> double & { return Vm[(size_t)i * n + j]; };
  double f = 0., tst1 = 0.;
  const double eps = 2.22e-16;
  for (int i = 1; i < n; ++i) e[i - 1] = e[i];
  e[n - 1] = 0.;
  for (int l = 0; l < n; ++l) {
    tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
    int m = l;
    while (m < n) { if (std::fabs(e[m]) <= eps * tst1) break; ++m; }
    if (m > l) {
      do {
        double g = d[l];
        double p = (d[l + 1] - g) / (2.0 * e[l]);
        double r = hyp(p, 1.);
        if (p < 0) r = -r;
        d[l] = e[l] / (p + r);
        d[l + 1] = e[l] * (p + r);
        const double dl1 = d[l + 1];
        double h = g - d[l];
        for (int i = l + 2; i < n; ++i) d[i] -= h;
        f += h;
        p = d[m];
        double c = 1., c2 = c, c3 = c, s = 0., s2 = 0.;
        const double el1 = e[l + 1];
        for (int i = m - 1; i >= l; --i) {
          c3 = c2; c2 = c; s2 = s;
          g = c * e[i];
          h = c * p;
          r = hyp(p, e[i]);
          e[i + 1] = s * r;
          s = e[i] / r;
          c = p / r;
          p = c * d[i] - s * g;
          d[i + 1] = h + s * (c * g + s * d[i]);
          for (int k = 0; k < n; ++k) {
            h = V(k, i + 1);
            V(k, i + 1) = s * V(k, i) + c * h;
            V(k, i) = c * V(k, i) - s * h;
          }
        }
        p = -s * s2 * c3 * el1 * e[l] / dl1;
        e[l] = s * p;
        d[l] = c * p;
      } while (std::fabs(e[l]) > eps * tst1);
    }
    d[l] += f;
    e[l] = 0.;
  }
  for (int i = 0; i < n - 1; ++i) {
    int k = i;
    double p = d[i];
    for (int j = i + 1; j < n; ++j) if (d[j] < p) { k = j; p = d[j]; }
    if (k != i) {
      d[k] = d[i]; d[i] = p;
      for (int j = 0; j < n; ++j) std::swap(V(j, i), V(j, k));
    }
  }
}
}  // namespace eig

static int upload_BD(ddrl_cmaes *h) {
  const int N = h->N;
  std::vector<float> Bf((size_t)N * h->ldn, 0.f);
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) Bf[(size_t)i * h->ldn + j] = (float)h->B[(size_t)i * N + j];
  HIPCHK(hipMemcpyAsync(h->d_B, Bf.data(), Bf.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->d_D, h->D.data(), N * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return DDRL_OK;
}

// ---------------------------------------------------------------------------------------------
// "device_eigen" (throughput mode): C = B diag(D^2) B^T on the GPU in fp64 by the parallel cyclic ONE-SIDED JACOBI
// method (Hestenes), this library's own routine -- no vendor solver.
// G starts as C (rows g_i), W as the identity.  A Givens rotation in the (p, q) plane is applied to the rows of both,
// chosen so that g_p . g_q becomes 0; a round rotates N / 2 disjoint row pairs at once (round-robin tournament
// order, one workgroup per pair), N - 1 rounds make a sweep.  When all rows of G = W C are mutually orthogonal, W C^2 W^T
// is diagonal: the rows of W are the eigenvectors and lambda_i = w_i C w_i^T = g_i . w_i.  Convergence is quadratic;
// it stops when a whole sweep found |g_p . g_q| <= 1e-9 |g_p| |g_q| for every pair (the eigenvalue error is second
// order in that, and B is stored in fp32): 6-9 sweeps, each replayed as ONE captured HIP graph of N - 1 rounds.
// Any orthonormal eigenbasis samples the same distribution; this one differs from c-cmaes' Householder / QL basis in
// signs, order and inside (near-)degenerate eigenspaces, so the mode is for throughput, not for seed-for-seed
// comparison (parity mode keeps the host routines above).  Eigenvalues agree with them to ~1e-13 relative
// (tests/test_cmaes_gpu.py).
// ---------------------------------------------------------------------------------------------
// lower triangle of the fp32 C -> full fp64 G; W = I
__global__ void cmaes_jacobi_init(const float *__restrict__ C, int ldn, int N, double *__restrict__ G, double *__restrict__ W) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j >= N || i >= N) return;
  G[(size_t)i * N + j] = (double)(j <= i ? C[(size_t)i * ldn + j] : C[(size_t)j * ldn + i]);
  if (W) W[(size_t)i * N + j] = i == j ? 1. : 0.;
}

__device__ __forceinline__ double block_sum256(double v, double *red, int tid) {   // all threads get the sum
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ void block_sum256x3(double &a, double &b, double &c, double *red /* [12] */, int tid) {
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); c += __shfl_xor(c, o); }
  if ((tid & 63) == 0) { red[tid >> 6] = a; red[4 + (tid >> 6)] = b; red[8 + (tid >> 6)] = c; }
  __syncthreads();
  a = (red[0] + red[1]) + (red[2] + red[3]);
  b = (red[4] + red[5]) + (red[6] + red[7]);
  c = (red[8] + red[9]) + (red[10] + red[11]);
}

// warm start: G = W Cs with W the previous generation's eigenvectors (rows) -- C moved by O(ccov) since, so the rows of G
// are nearly orthogonal already and the iteration needs 2-4 sweeps instead of 9-11 from W = I
__global__ void __launch_bounds__(256)
cmaes_jacobi_warm(const double *__restrict__ W, const double *__restrict__ Cs, int N, double *__restrict__ G) {
  const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
  if (j >= N) return;
  const double *w = W + (size_t)i * N;
  double s0 = 0., s1 = 0.;
  int k = 0;
  for (; k + 1 < N; k += 2) { s0 += w[k] * Cs[(size_t)k * N + j]; s1 += w[k + 1] * Cs[(size_t)(k + 1) * N + j]; }
  if (k < N) s0 += w[k] * Cs[(size_t)k * N + j];
  G[(size_t)i * N + j] = s0 + s1;
}

// round `step` of a sweep over n2 = N rounded up to even players: workgroup b rotates the pair it is dealt
__global__ void __launch_bounds__(256)
cmaes_jacobi_round(double *__restrict__ G, double *__restrict__ W, int N, int n2, int step, int *__restrict__ rotated) {
  __shared__ double red[12];
  const int tid = threadIdx.x, b = blockIdx.x, m = n2 - 1;
  int p, q;
  if (b == 0) { p = m; q = step; }
  else { p = (step + b) % m; q = (step - b + m) % m; }
  if (p >= N || q >= N) return;                     // the bye of an odd N
  if (p > q) { const int t = p; p = q; q = t; }
  double *gp = G + (size_t)p * N, *gq = G + (size_t)q * N, *wp = W + (size_t)p * N, *wq = W + (size_t)q * N;
  // up to JR elements of each of the four rows per thread stay in registers from the dot products to the rotation:
  // one memory round trip per round instead of two (rows longer than 256 JR take the tail from memory again)
  constexpr int JR = 3;
  double xg[JR], yg[JR], xw[JR], yw[JR];
  double al = 0., be = 0., ga = 0.;
#pragma unroll
  for (int r = 0; r < JR; ++r) {
    const int i = tid + 256 * r;
    const bool in = i < N;
    xg[r] = in ? gp[i] : 0.; yg[r] = in ? gq[i] : 0.; xw[r] = in ? wp[i] : 0.; yw[r] = in ? wq[i] : 0.;
  }
#pragma unroll
  for (int r = 0; r < JR; ++r) { al += xg[r] * xg[r]; be += yg[r] * yg[r]; ga += xg[r] * yg[r]; }
  for (int i = tid + 256 * JR; i < N; i += 256) { const double x = gp[i], y = gq[i]; al += x * x; be += y * y; ga += x * y; }
  block_sum256x3(al, be, ga, red, tid);
  if (fabs(ga) <= 1e-9 * sqrt(al * be)) return;     // uniform: orthogonal enough (eigenvalue error is second order: 1e-18)
  const double zeta = (be - al) / (2. * ga);
  const double t = (zeta >= 0. ? 1. : -1.) / (fabs(zeta) + sqrt(1. + zeta * zeta));
  const double c = 1. / sqrt(1. + t * t), sn = c * t;
#pragma unroll
  for (int r = 0; r < JR; ++r) {
    const int i = tid + 256 * r;
    if (i < N) {
      gp[i] = c * xg[r] - sn * yg[r]; gq[i] = sn * xg[r] + c * yg[r];
      wp[i] = c * xw[r] - sn * yw[r]; wq[i] = sn * xw[r] + c * yw[r];
    }
  }
  for (int i = tid + 256 * JR; i < N; i += 256) {
    const double x = gp[i], y = gq[i];
    gp[i] = c * x - sn * y; gq[i] = sn * x + c * y;
    const double u = wp[i], v = wq[i];
    wp[i] = c * u - sn * v; wq[i] = sn * u + c * v;
  }
  if (tid == 0) atomicAdd(rotated, 1);
}

// BLOCKED rounds (round 3; option "eigen_block", default on).  The rounds above are launch-bound: a workgroup rotates two rows
// of 631 doubles and is gone in 6.9 us, 630 times a sweep.  Here the players of the tournament are blocks of JB = 8 rows: a
// workgroup takes the two blocks it is dealt (16 rows), forms their 16 x 16 Gram matrix S = A A^T (the only pass over the rows:
// staged through LDS in 64-column chunks), runs ONE cyclic sweep of the same rotations on S itself -- 15 steps of 8 disjoint
// pairs, S <- R S R^T, accumulating T <- R T, every rotation chosen exactly as the row kernel chooses it from (|g_p|^2, |g_q|^2,
// g_p . g_q) -- and then applies T to the 16 rows of G and W in one more pass.  A sweep is ceil(N / 8) - 1 rounds (78 instead of
// 630 at N = 631); every pair of rows still meets once per sweep (pairs inside a block more often), and convergence is judged
// by the same per-pair test, counted over a whole sweep.
#define JB 8
#define JT 1024      // threads of a blocked round: the two passes over the rows use all of them, the 16 x 16 sweep the first 256
__global__ void __launch_bounds__(JT)
cmaes_jacobi_block_round(double *__restrict__ G, double *__restrict__ W, int N, int nb2, int step, int *__restrict__ rotated) {
  __shared__ double tile[2 * JB][257];              // a 256-column chunk of the 16 rows
  __shared__ double part[3][256];                   // partial Gram sums of the column quarters 1..3
  __shared__ double S[2 * JB][2 * JB + 1], T[2 * JB][2 * JB + 1];
  __shared__ double cc[2 * JB], ss[2 * JB];
  __shared__ int partner[2 * JB], rows[2 * JB], nrot;
  const int tid = threadIdx.x, b = blockIdx.x, m = nb2 - 1;
  int P, Q;
  if (b == 0) { P = m; Q = step; }
  else { P = (step + b) % m; Q = (step - b + m) % m; }
  if (P * JB >= N && Q * JB >= N) return;          // (a block beyond N is a bye: its partner still rotates inside itself)
  if (P > Q) { const int t = P; P = Q; Q = t; }
  if (tid < 2 * JB) { const int ri = tid < JB ? P * JB + tid : Q * JB + tid - JB; rows[tid] = ri < N ? ri : -1; }
  if (tid == 0) nrot = 0;
  __syncthreads();
  const int e = tid & 255, x = e >> 4, y = e & 15, kq = tid >> 8;     // this thread's entry of the 16 x 16 matrices, its quarter of a chunk
  // ---- S = A A^T over the 16 rows ----
  double acc = 0.;
#pragma unroll 1
  for (int c0 = 0; c0 < N; c0 += 256) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f = tid + JT * q, r = f >> 8, k = f & 255, ri = rows[r];
      tile[r][k] = ri >= 0 && c0 + k < N ? G[(size_t)ri * N + c0 + k] : 0.;
    }
    __syncthreads();
#pragma unroll 16
    for (int k = 64 * kq; k < 64 * kq + 64; ++k) acc += tile[x][k] * tile[y][k];
    __syncthreads();
  }
  if (kq > 0) part[kq - 1][e] = acc;
  __syncthreads();
  if (kq == 0) {
    S[x][y] = (acc + part[0][e]) + (part[1][e] + part[2][e]);
    T[x][y] = x == y ? 1. : 0.;
  }
  __syncthreads();
  // ---- one cyclic sweep on S: 15 steps of 8 disjoint pairs (the same tournament over 16 players) ----
#pragma unroll 1
  for (int st = 0; st < 2 * JB - 1; ++st) {
    if (tid < JB) {
      const int mm = 2 * JB - 1;
      int a, c;
      if (tid == 0) { a = mm; c = st; }
      else { a = (st + tid) % mm; c = (st - tid + mm) % mm; }
      if (a > c) { const int t = a; a = c; c = t; }
      const double al = S[a][a], be = S[c][c], ga = S[a][c];
      double co = 1., sn = 0.;
      if (fabs(ga) > 1e-9 * sqrt(al * be)) {        // the row kernel's test and rotation (rows a, c of the pair)
        const double zeta = (be - al) / (2. * ga);
        const double t = (zeta >= 0. ? 1. : -1.) / (fabs(zeta) + sqrt(1. + zeta * zeta));
        co = 1. / sqrt(1. + t * t); sn = co * t;
        atomicAdd(&nrot, 1);
      }
      // new_a = co a - sn c, new_c = sn a + co c
      partner[a] = c; partner[c] = a;
      cc[a] = co; ss[a] = -sn; cc[c] = co; ss[c] = sn;
    }
    __syncthreads();
    double s_new = 0., t_new = 0.;
    if (kq == 0) {
      const int px = partner[x], py = partner[y];
      s_new = cc[x] * (cc[y] * S[x][y] + ss[y] * S[x][py]) + ss[x] * (cc[y] * S[px][y] + ss[y] * S[px][py]);
      t_new = cc[x] * T[x][y] + ss[x] * T[px][y];
    }
    __syncthreads();
    if (kq == 0) { S[x][y] = s_new; T[x][y] = t_new; }
    __syncthreads();
  }
  if (nrot == 0) return;                            // uniform: these 16 rows are orthogonal enough already
  if (tid == 0) atomicAdd(rotated, nrot);
  // ---- rows <- T rows, for G and W ----
#pragma unroll 1
  for (int which = 0; which < 2; ++which) {
    double *M = which ? W : G;
#pragma unroll 1
    for (int k = tid; k < N; k += JT) {
      double v[2 * JB];
#pragma unroll
      for (int r = 0; r < 2 * JB; ++r) { const int ri = rows[r]; v[r] = ri >= 0 ? M[(size_t)ri * N + k] : 0.; }
#pragma unroll 1                                     // (rolled: unrolled, the 256 entries of T are hoisted out of the column loop into 512 registers)
      for (int r = 0; r < 2 * JB; ++r) {
        const int ri = rows[r];
        if (ri >= 0) {
          double o = 0.;
#pragma unroll
          for (int u = 0; u < 2 * JB; ++u) o += T[r][u] * v[u];
          M[(size_t)ri * N + k] = o;
        }
      }
    }
  }
}

// The same round for N <= 1024 (every policy of BASELINE.md): ONE trip to memory.  A workgroup fetches its 16 rows of G and W at
// once (the rows were written by other workgroups in the previous round, so every dependent access costs a device-level round
// trip -- the round above makes five of them in a row) into the registers of its 8 waves, laid out as the fp64 matrix cores'
// operand; parks the G values in LDS for the Gram matrix (v_mfma_f64_16x16x4, both operands from one LDS read); runs the 16 x 16
// sweep in the first WAVE alone (its own LDS ordering instead of 45 workgroup barriers); and applies T to both matrices from the
// registers, on the matrix cores again.  The rotation's tangent is computed in fp32 (it only steers the iteration), its cosine
// from the tangent in fp64 (a Newton-refined reciprocal square root: c^2 + s^2 = 1 to fp64 rounding).
__device__ __forceinline__ void jacobi_wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#define JF 512       // threads of the one-trip round
#define JFG_MAX 8    // column groups (of 16) a wave holds at most: 8 waves x 8 x 16 = 1024 columns (template parameter JFG: as many as N needs)
typedef double jf_f64x4 __attribute__((ext_vector_type(4)));

// One cyclic sweep on the 16 x 16 Gram matrix S (LDS), rotations accumulated in T (LDS); one wave.  FULL: all 120 pairs of the 16
// rows (15 steps: player 15 meets st, the others p <-> (2 st - p) mod 15); otherwise only the 64 pairs ACROSS the two blocks
// (8 steps: row p of the first block meets row (p + st) mod 8 of the second) -- the pairs inside a block have had their turn
// in round 0 of the sweep, where every block plays (a block with a bye plays alone).
// A step is one wave's chain of dependent instructions, so it is kept short: lane (xg, yy) owns rows 4 xg .. 4 xg + 3 of column
// yy and computes the rotation of ROW yy's pair itself (four lanes each; the column's coefficients are then at hand), fetches
// everything it will combine before that arithmetic starts, and gets its rows' coefficients from the lanes that hold them
// (no trip through LDS).  The tangent only steers the iteration: hardware reciprocal / square root in fp32 (|zeta| beyond
// 1e15: t = 1 / (2 zeta), where 1 + zeta^2 leaves fp32); the cosine is 1 / sqrt(1 + t^2) in fp64 (fp32 seed, two Newton steps).
template <bool FULL>
__device__ __forceinline__ void jacobi_tile_sweep(double *S, double *T, int lane, int *nrot, int *worst) {
  constexpr int LS = 2 * JB + 1, LT = 2 * JB, mm = 2 * JB - 1, nst = FULL ? 2 * JB - 1 : JB;
  const int yy = lane & 15, xg = lane >> 4;
  auto partner_of = [&](int p, int st) {
    if (!FULL) return p < JB ? JB + ((p + st) & (JB - 1)) : ((p - st) & (JB - 1));
    int q = (2 * st - p + 2 * mm) % mm;             // (p, st < 15: the sum is positive)
    q = p == st ? mm : q;
    return p == mm ? st : q;
  };
  int mine = 0;                                     // rotations this lane counted (lane a < 16 of each pair), their worst ratio
  float bad = 0.f;
#pragma unroll 1
  for (int st = 0; st < nst; ++st) {
    const int py = partner_of(yy, st), a = yy < py ? yy : py, c = yy < py ? py : yy;
    const double al = S[a * LS + a], be = S[c * LS + c], ga = S[a * LS + c];
    double sxy[4], sxp[4], spy[4], spp[4], tx[4], tp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int xx = 4 * xg + i, px = partner_of(xx, st);
      sxy[i] = S[xx * LS + yy]; sxp[i] = S[xx * LS + py]; spy[i] = S[px * LS + yy]; spp[i] = S[px * LS + py];
      tx[i] = T[xx * LT + yy]; tp[i] = T[px * LT + yy];
    }
    const bool rot = ga * ga > 1e-18 * (al * be);   // |g_a . g_c| > 1e-9 |g_a| |g_c|
    const float zeta = (float)((be - al) * __builtin_amdgcn_rcp(2. * ga)), az = fabsf(zeta);
    float tf = az > 1e15f ? 0.5f * __builtin_amdgcn_rcpf(az) : __builtin_amdgcn_rcpf(az + __builtin_amdgcn_sqrtf(1.f + az * az));
    tf = zeta >= 0.f ? tf : -tf;
    const double t = (double)tf, qq = 1. + t * t;
    double r = (double)__builtin_amdgcn_rsqf((float)qq);
    r = r * (1.5 - 0.5 * qq * r * r);
    r = r * (1.5 - 0.5 * qq * r * r);
    const double cy = rot ? r : 1., sy = rot ? (yy == a ? -(r * t) : r * t) : 0.;      // new_a = co a - sn c, new_c = sn a + co c
    if (rot && yy == a) { ++mine; bad = fmaxf(bad, (float)(ga * ga) * __builtin_amdgcn_rcpf((float)(al * be))); }
    double sn4[4], tn4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double cx = __shfl(cy, 4 * xg + i), sx = __shfl(sy, 4 * xg + i);       // row (4 xg + i)'s pair: lane 4 xg + i computed it
      sn4[i] = cx * (cy * sxy[i] + sy * sxp[i]) + sx * (cy * spy[i] + sy * spp[i]);
      tn4[i] = cx * tx[i] + sx * tp[i];
    }
    jacobi_wave_fence();
#pragma unroll
    for (int i = 0; i < 4; ++i) { S[(4 * xg + i) * LS + yy] = sn4[i]; T[(4 * xg + i) * LT + yy] = tn4[i]; }
    jacobi_wave_fence();
  }
  if (lane < 2 * JB && mine) { atomicAdd(nrot, mine); atomicMax(worst, __float_as_int(bad)); }
}

// COH: the rows are read and written with agent-scope accesses (they by-pass the caches that are not shared by all CUs), for the
// one-launch decomposition, whose workgroups hand rows to each other inside a kernel
template <bool COH>
__device__ __forceinline__ double jacobi_ld(const double *p) { return COH ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p; }
template <bool COH>
__device__ __forceinline__ void jacobi_st(double *p, double v) {
  if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <bool COH, int JFG>
__device__ __forceinline__ void jacobi_fast_round(double *G, double *W, int N, int ldt, int nb2, int step, int *rotated, double *jsm) {
  double *tile = jsm;                               // [16][ldt]: the G rows
  double *S = tile + 2 * JB * ldt, *T = S + 2 * JB * (2 * JB + 1);     // [16][17], [16][16]
  double *part = T + 2 * JB * 2 * JB;               // [8 waves][256]: the waves' partial Gram matrices
  __shared__ int nrot, worst;
  const int tid = threadIdx.x, b = blockIdx.x, m = nb2 - 1, LS = 2 * JB + 1, LT = 2 * JB;
  int P, Q;
  if (b == 0) { P = m; Q = step; }
  else { P = (step + b) % m; Q = (step - b + m) % m; }
  if (P * JB >= N && Q * JB >= N) return;
  if (P > Q) { const int t = P; P = Q; Q = t; }
#ifdef DDRL_CMAES_STAMPS
  const long long t0 = wall_clock64();
  long long t1 = 0, t2 = 0, t3 = 0;
#endif
  auto row_of = [&](int r) { const int ri = r < JB ? P * JB + r : Q * JB + r - JB; return ri < N ? ri : -1; };
  // The rows are held as the fp64 matrix core's B operand of the product T x rows: lane (lr = l >> 4, lc = l & 15) has element
  // [row 4 ks + lr][column 16 cg + lc] of k-step ks of column group cg; wave wv owns the groups wv, wv + 8, ... (at most 8: N <= 1024).
  // Every fetch is issued before the first is waited for: the addresses are clamped into the matrix instead of branching round
  // the load (a conditional load waits for its own data before the next is issued: 32 round trips in a row).
  const int lane = tid & 63, wv = tid >> 6, lr = lane >> 4, lc = lane & 15, ncg = (N + 15) >> 4;
  double vg[JFG][4], vw[JFG][4];
  int rix[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) rix[ks] = row_of(4 * ks + lr);
#pragma unroll
  for (int g = 0; g < JFG; ++g) {
    const int cg = wv + (JF / 64) * g;
    if (cg < ncg) {                                 // (wave-uniform)
      const int col = 16 * cg + lc, colc = col < N ? col : N - 1;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const size_t at = (size_t)(rix[ks] >= 0 ? rix[ks] : 0) * N + colc;
        vg[g][ks] = jacobi_ld<COH>(G + at);
        vw[g][ks] = jacobi_ld<COH>(W + at);
      }
    }
  }
#pragma unroll
  for (int g = 0; g < JFG; ++g) {
    const int cg = wv + (JF / 64) * g;
    if (cg < ncg) {
      const int col = 16 * cg + lc;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bool in = rix[ks] >= 0 && col < N;
        vg[g][ks] = in ? vg[g][ks] : 0.;
        vw[g][ks] = in ? vw[g][ks] : 0.;
        if (col < N) tile[(4 * ks + lr) * ldt + col] = vg[g][ks];
      }
    }
  }
  if (tid == 0) { nrot = 0; worst = 0; }
  __syncthreads();
#ifdef DDRL_CMAES_STAMPS
  t1 = wall_clock64();
#endif
  {
    // S = A A^T on the fp64 matrix cores: v_mfma_f64_16x16x4 sums 4 columns a step, A[i][k] on lane (i = l & 15, k = l >> 4) and
    // B[k][j] = A[j][k] on the same lane -- one LDS read feeds both operands.  The 8 waves take the 4-column steps round robin
    // into two accumulators each; their partial matrices meet in LDS (D[row = (l >> 4) + 4 v][col = l & 15] in register v).
    const double *src = tile + lc * ldt + lr;
    jf_f64x4 acc0 = {0., 0., 0., 0.}, acc1 = {0., 0., 0., 0.};
    const int nk = (N + 3) >> 2;
    for (int ks = wv; ks < nk; ks += 32) {          // four steps' operands are read before the first product waits for its own
      double av[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) av[u] = 4 * (ks + 8 * u) + lr < N ? src[4 * (ks + 8 * u)] : 0.;
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], av[0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], av[1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], av[2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], av[3], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) part[wv * 256 + v * 64 + lane] = acc0[v] + acc1[v];
    if (tid < 256) T[tid] = (tid >> 4) == (tid & 15) ? 1. : 0.;
  }
  __syncthreads();
  if (tid < 256) {
    double sum = 0.;
#pragma unroll
    for (int w = 0; w < JF / 64; ++w) sum += part[w * 256 + tid];
    S[(((tid & 63) >> 4) + 4 * (tid >> 6)) * LS + (tid & 15)] = sum;
  }
  __syncthreads();
#ifdef DDRL_CMAES_STAMPS
  t2 = wall_clock64();
#endif
  if (tid < 64) {                                   // ---- the 16 x 16 sweep, first wave only (round 0 of a sweep: all pairs) ----
    if (step == 0) jacobi_tile_sweep<true>(S, T, tid, &nrot, &worst);
    else jacobi_tile_sweep<false>(S, T, tid, &nrot, &worst);
  }
  __syncthreads();
#ifdef DDRL_CMAES_STAMPS
  t3 = wall_clock64();
#endif
  if (nrot == 0) return;                            // uniform
  // rows <- T x rows, G and W alike, on the matrix cores: A[i][k] = T[i][4 ks + k] on lane (i = lc, k = lr), four k-steps per
  // column group; the result has row lr + 4 v, column lc in register v
  double ta[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) ta[ks] = T[lc * LT + 4 * ks + lr];
  int rox[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) rox[v] = row_of(lr + 4 * v);
#pragma unroll
  for (int g = 0; g < JFG; ++g) {
    const int cg = wv + (JF / 64) * g;
    if (cg < ncg) {
      jf_f64x4 og = {0., 0., 0., 0.}, ow = {0., 0., 0., 0.};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        og = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[ks], vg[g][ks], og, 0, 0, 0);
        ow = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[ks], vw[g][ks], ow, 0, 0, 0);
      }
      const int col = 16 * cg + lc;
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (rox[v] >= 0 && col < N) {
          jacobi_st<COH>(G + (size_t)rox[v] * N + col, og[v]);
          jacobi_st<COH>(W + (size_t)rox[v] * N + col, ow[v]);
        }
    }
  }
  // (after the stores: the compiler waits for these two before whatever follows them, and this is the wave the others wait for)
  if (tid == 0) { atomicAdd(rotated, nrot); atomicMax(rotated + 1, worst); }
#ifdef DDRL_CMAES_STAMPS
  if (b == 1 && tid == 0) {
    const long long t4 = wall_clock64();
    __builtin_amdgcn_s_waitcnt(0);
    const long long t5 = wall_clock64();
    rotated[2] = (int)(t1 - t0); rotated[3] = (int)(t2 - t1); rotated[4] = (int)(t3 - t2); rotated[5] = (int)(t4 - t3); rotated[6] = (int)(t5 - t4); rotated[7] = step;
  }
#endif
}

template <int JFG>
__global__ void __launch_bounds__(JF)
cmaes_jacobi_block_fast(double *__restrict__ G, double *__restrict__ W, int N, int ldt, int nb2, int step, int *__restrict__ rotated) {
  extern __shared__ __align__(16) double jsm[];
  jacobi_fast_round<false, JFG>(G, W, N, ldt, nb2, step, rotated, jsm);
}

// The WHOLE decomposition in one launch: the nb2 / 2 workgroups (at most 64, one per CU: all resident at once) loop over the sweeps'
// rounds themselves -- a round's kernel boundary (drain, cache write-back, dispatch: ~7 us against ~12 us of work) and the host's
// look at the rotation count after every sweep go.  Inside a sweep there is no barrier at all: a block's rows are touched by
// its owner of the round only, so a workgroup entering round e + 1 needs nothing but its two blocks' rows as their owners of
// round e left them.  Every block has a counter of the rounds it has been through (`done`, behind the sweeps' counters in
// ctl): a workgroup publishes e + 1 for its two blocks once every wave's stores have been acknowledged, and waits for its next
// two blocks to show e + 1 before it fetches them -- the tournament's data flow, nothing more; fast pairs run ahead of slow
// ones.  At the end of a sweep the workgroups meet once (ctl[0] counts arrivals, never reset inside a launch) so that all
// read the same rotation count.  A workgroup's rows reach the others through agent-scope stores and loads (`sc1`: no
// write-back / invalidation of a whole L2, which costs what the kernel boundary did).  ctl[1] is raised by a workgroup that has
// waited longer than its patience (2 s of the 100 MHz clock; option "eigen_persist_wait_ms") (a workgroup that is not resident: another process holds
// CUs) -- every workgroup then leaves at its next look (no wave spins for ever), and the host carries on from that state with
// the launch-per-round path (whole rounds of disjoint row pairs: any such state is a valid iterate).  ctl[2] = sweeps done,
// ctl[8 + 2 s], ctl[9 + 2 s]: sweep s's rotation count and worst ratio.
#define JACOBI_MAX_SWEEPS 60
#define JACOBI_MAX_BLOCKS (2 * JF / JB)           // blocks of the largest N the one-trip round serves (1024 / 8)
#define JACOBI_CTL_INTS (8 + 2 * JACOBI_MAX_SWEEPS + JACOBI_MAX_BLOCKS)
#define JACOBI_PERSIST_JFG 5                      // widest instantiation of the one-launch kernel (N <= 640): beyond, its loop's registers spill
// thread 0 of a workgroup: wait until *p >= want (agent scope); false: gave up (own time limit, or another workgroup's)
__device__ __forceinline__ bool jacobi_wait_for(int *p, int want, int *abort_flag, long long ticks) {
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
    if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
    if (wall_clock64() - t0 > ticks) { __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
    __builtin_amdgcn_s_sleep(1);
  }
  return true;
}
template <int JFG>
__global__ void __launch_bounds__(JF)
cmaes_jacobi_sweeps_fast(double *G, double *W, int N, int ldt, int nb2, int *ctl, long long ticks) {
  extern __shared__ __align__(16) double jsm[];
  __shared__ int go;
  const int tid = threadIdx.x, b = blockIdx.x, nwg = nb2 / 2, m = nb2 - 1;      // (nwg: the grid the rounds need, whatever was launched)
  int *done = ctl + 8 + 2 * JACOBI_MAX_SWEEPS;      // [nb2]: rounds each block has been through
  int e = 0;                                        // rounds this workgroup has finished
  for (int sweep = 0; sweep < JACOBI_MAX_SWEEPS; ++sweep) {
    int *rotated = ctl + 8 + 2 * sweep;
    for (int step = 0; step < nb2 - 1; ++step) {
      const int P = b == 0 ? m : (step + b) % m, Q = b == 0 ? step : (step - b + m) % m;      // jacobi_fast_round's pairing
      if (e > 0) {
        if (tid == 0) go = jacobi_wait_for(done + P, e, ctl + 1, ticks) && jacobi_wait_for(done + Q, e, ctl + 1, ticks);
        __syncthreads();
        if (!go) return;                            // uniform
      }
      jacobi_fast_round<true, JFG>(G, W, N, ldt, nb2, step, rotated, jsm);
      __builtin_amdgcn_s_waitcnt(0);                // this wave's stores have been acknowledged
      __syncthreads();                              // (also: `go` has been read by every thread before it is written again)
      ++e;
      if (tid == 0) {
        __hip_atomic_store(done + P, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(done + Q, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // the sweep's end: every workgroup reads the same count once all have arrived (their counts were added before)
    if (tid == 0) {
      __hip_atomic_fetch_add(ctl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      go = jacobi_wait_for(ctl, (sweep + 1) * nwg, ctl + 1, ticks);
    }
    __syncthreads();
    if (!go) return;
    const int rot = __hip_atomic_load(rotated, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (b == 0 && tid == 0) ctl[2] = sweep + 1;
    if (rot == 0) return;
    __syncthreads();                                // `go` is written again at the next round's wait
  }
}
extern "C" size_t cmaes_jacobi_fast_lds(int ldt) {
  return ((size_t)2 * JB * ldt + (2 * JB) * (2 * JB + 1) + (2 * JB) * (2 * JB) + (JF / 64) * 256) * sizeof(double);
}

// lambda_i = g_i . w_i, D_i = sqrt(max(lambda_i, 0)), B[:, i] = w_i (fp32, the sampling GEMM's operand)
__global__ void __launch_bounds__(256)
cmaes_jacobi_finish(const double *__restrict__ G, const double *__restrict__ W, int N, int ldn, double *__restrict__ ev,
                    double *__restrict__ D, float *__restrict__ B) {
  __shared__ double red[4];
  const int tid = threadIdx.x, i = blockIdx.x;
  double l = 0.;
  for (int j = tid; j < N; j += 256) {
    const double w = W[(size_t)i * N + j];
    l += G[(size_t)i * N + j] * w;
    B[(size_t)j * ldn + i] = (float)w;
  }
  l = block_sum256(l, red, tid);
  if (tid == 0) { ev[i] = l; D[i] = sqrt(l > 0. ? l : 0.); }
}

static int device_eigen(ddrl_cmaes *h) {
  const int N = h->N;
  // players of a sweep's tournament: blocks of JB rows (eigen_block) or single rows
  const int nplayers = h->eigen_block ? (N + JB - 1) / JB : N, n2 = h->eigen_block ? (nplayers < 2 ? 2 : (nplayers + 1) / 2 * 2) : (N + 1) / 2 * 2;
  const int ldt = N | 1;                            // LDS row stride of the one-trip kernel (odd: the Gram sums' two row reads spread over the banks)
  const bool fast = h->eigen_block && N <= 2 * JF && cmaes_jacobi_fast_lds(ldt) <= 160 * 1024 - 64;
  // the one-trip kernels are compiled per number of 16-column groups a wave holds (their registers): 1 .. JFG_MAX
  const int jfg = ((N + 15) / 16 + JF / 64 - 1) / (JF / 64);
  typedef void (*round_fn)(double *, double *, int, int, int, int, int *);
  typedef void (*sweeps_fn)(double *, double *, int, int, int, int *, long long);
  static const round_fn round_k[JFG_MAX] = {cmaes_jacobi_block_fast<1>, cmaes_jacobi_block_fast<2>, cmaes_jacobi_block_fast<3>, cmaes_jacobi_block_fast<4>,
                                            cmaes_jacobi_block_fast<5>, cmaes_jacobi_block_fast<6>, cmaes_jacobi_block_fast<7>, cmaes_jacobi_block_fast<8>};
  static const sweeps_fn sweeps_k[JACOBI_PERSIST_JFG] = {cmaes_jacobi_sweeps_fast<1>, cmaes_jacobi_sweeps_fast<2>, cmaes_jacobi_sweeps_fast<3>,
                                                         cmaes_jacobi_sweeps_fast<4>, cmaes_jacobi_sweeps_fast<5>};
  if (fast) {
    static LdsConfig conf[JFG_MAX];
    HIPCHK(ddrl_set_lds(round_k[jfg - 1], cmaes_jacobi_fast_lds(ldt), conf[jfg - 1]));
  }
  auto round = [&](int step) {
    if (fast) hipLaunchKernelGGL(round_k[jfg - 1], dim3(n2 / 2), dim3(JF), cmaes_jacobi_fast_lds(ldt), h->stream, h->d_A, h->d_W, N, ldt, n2, step, h->d_info);
    else if (h->eigen_block) hipLaunchKernelGGL(cmaes_jacobi_block_round, dim3(n2 / 2), dim3(JT), 0, h->stream, h->d_A, h->d_W, N, n2, step, h->d_info);
    else hipLaunchKernelGGL(cmaes_jacobi_round, dim3(n2 / 2), dim3(256), 0, h->stream, h->d_A, h->d_W, N, n2, step, h->d_info);
  };
  if (h->sweep_graph && h->sweep_graph_block != h->eigen_block) {     // the option changed: the captured sweep is the other kernel's
    (void)hipGraphExecDestroy(h->sweep_graph);
    h->sweep_graph = nullptr;
  }
  if (h->eig_warm) {       // Cs = C in fp64 (d_Cs), G = W_prev Cs; W keeps the previous eigenvectors
    hipLaunchKernelGGL(cmaes_jacobi_init, dim3((N + 63) / 64, N), dim3(64), 0, h->stream, h->d_C, h->ldn, N, h->d_Cs, (double *)nullptr);
    hipLaunchKernelGGL(cmaes_jacobi_warm, dim3((N + 255) / 256, N), dim3(256), 0, h->stream, h->d_W, h->d_Cs, N, h->d_A);
  } else {
    hipLaunchKernelGGL(cmaes_jacobi_init, dim3((N + 63) / 64, N), dim3(64), 0, h->stream, h->d_C, h->ldn, N, h->d_A, h->d_W);
  }
  HIPCHK(hipGetLastError());
  int sweeps = 0;
  if (N > 1) {
    // one sweep = n2 - 1 dependent launches: captured once as a HIP graph and replayed (the rounds are launch-bound)
    if (!h->sweep_graph && !h->sweep_graph_failed) {
      hipGraph_t g = nullptr;
      bool ok = hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
      if (ok) {
        for (int step = 0; step < n2 - 1; ++step) round(step);
        ok = hipStreamEndCapture(h->stream, &g) == hipSuccess && g;
      }
      if (ok) ok = hipGraphInstantiate(&h->sweep_graph, g, nullptr, nullptr, 0) == hipSuccess;
      h->sweep_graph_block = h->eigen_block;
      if (g) (void)hipGraphDestroy(g);
      if (!ok) { h->sweep_graph = nullptr; h->sweep_graph_failed = true; (void)hipGetLastError(); }
    }
    bool persist = fast && h->eigen_persist && jfg <= JACOBI_PERSIST_JFG;
#ifdef DDRL_CMAES_STAMPS
    persist = false;                                // the stamps are the launch-per-round kernel's
#endif
    static const bool trace = getenv("DDRL_CMAES_TRACE") != nullptr;
    auto say = [&](int sweep, int rotations, int worst_bits) {
      float w;
      memcpy(&w, &worst_bits, 4);
      fprintf(stderr, "[cmaes] gen %g sweep %d: %d rotations, worst (g_a.g_c)^2/(|g_a|^2|g_c|^2) %.3g\n", (double)h->gen, sweep, rotations, (double)w);
    };
    bool done = false;
    if (persist) {                                  // every sweep in ONE launch (cmaes_jacobi_sweeps_fast)
      static LdsConfig conf2[JACOBI_PERSIST_JFG];
      HIPCHK(ddrl_set_lds(sweeps_k[jfg - 1], cmaes_jacobi_fast_lds(ldt), conf2[jfg - 1]));
      std::vector<int> ctl(JACOBI_CTL_INTS);
      HIPCHK(hipMemsetAsync(h->d_info, 0, ctl.size() * sizeof(int), h->stream));
      // (eigen_persist_drop: a test's way to a workgroup that never arrives -- the grid is launched short)
      const int grid = std::max(1, n2 / 2 - h->eigen_persist_drop);
      hipLaunchKernelGGL(sweeps_k[jfg - 1], dim3(grid), dim3(JF), cmaes_jacobi_fast_lds(ldt), h->stream, h->d_A, h->d_W, N, ldt, n2, h->d_info,
                         (long long)h->eigen_persist_wait_ms * 100000ll);
      HIPCHK(hipGetLastError());
      HIPCHK(hipMemcpyAsync(ctl.data(), h->d_info, ctl.size() * sizeof(int), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
      if (trace) for (int s2 = 0; s2 < ctl[2]; ++s2) say(s2, ctl[8 + 2 * s2], ctl[9 + 2 * s2]);
      if (ctl[1]) {                                 // a workgroup was not resident within the barrier's bound: the matrices are a valid
        if (++h->eigen_persist_timeouts >= 2) h->eigen_persist = 0;     // intermediate state (whole rounds only), the rounds below carry on from it; twice: this GPU is shared, no third wait
        if (trace) fprintf(stderr, "[cmaes] the one-launch decomposition gave up at a barrier after %d sweeps: launch per round from here\n", ctl[2]);
      } else if (ctl[2] > 0 && ctl[8 + 2 * (ctl[2] - 1)] == 0) {
        sweeps = ctl[2] - 1;
        done = true;
      } else {
        sweeps = JACOBI_MAX_SWEEPS;
        done = true;
      }
    }
    for (; !done && sweeps < JACOBI_MAX_SWEEPS; ++sweeps) {
      HIPCHK(hipMemsetAsync(h->d_info, 0, 8 * sizeof(int), h->stream));
      if (h->sweep_graph) {
        HIPCHK(hipGraphLaunch(h->sweep_graph, h->stream));
      } else {
        for (int step = 0; step < n2 - 1; ++step) round(step);
      }
      HIPCHK(hipGetLastError());
      int info[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      HIPCHK(hipMemcpyAsync(info, h->d_info, sizeof(info), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
      const int rotated = info[0];
      if (trace) {
        say(sweeps, rotated, info[1]);
#ifdef DDRL_CMAES_STAMPS
        fprintf(stderr, "[cmaes]   stamps (10 ns units) of workgroup 1, round %d: load %d gram %d sweep %d apply %d drain %d\n", info[7], info[2], info[3], info[4], info[5], info[6]);
#endif
      }
      if (rotated == 0) break;
    }
    if (sweeps >= JACOBI_MAX_SWEEPS) return ddrl_fail(DDRL_EHIP, "device_eigen: the Jacobi iteration did not converge in %d sweeps", JACOBI_MAX_SWEEPS);
  }
  h->eigen_sweeps = sweeps;
  h->eig_warm = 1;
  hipLaunchKernelGGL(cmaes_jacobi_finish, dim3(N), dim3(256), 0, h->stream, h->d_A, h->d_W, N, h->ldn, h->d_ev, h->d_D, h->d_B);
  HIPCHK(hipGetLastError());
  // host mirrors (TestForTermination, ddrl_cmaes_get_vec): eigenvalues now, B = W^T
  std::vector<double> Wh((size_t)N * N);
  HIPCHK(hipMemcpyAsync(Wh.data(), h->d_W, Wh.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(h->D.data(), h->d_ev, N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) h->B[(size_t)j * N + i] = Wh[(size_t)i * N + j];
  return DDRL_OK;
}

// cmaes_UpdateEigensystem(t, 0) (cmaes.c:1885-1947) with the CPU-time gate off (maxtime = 1.0)
static int update_eigensystem(ddrl_cmaes *h) {
  if (h->eigen_uptodate) return DDRL_OK;
  if (h->gen < h->genOfEigensysUpdate + h->modulo) return DDRL_OK;
  const int N = h->N;
  if (h->device_eigen) {
    int rc = device_eigen(h);
    if (rc) return rc;
    h->minEW = *std::min_element(h->D.begin(), h->D.end());
    h->maxEW = *std::max_element(h->D.begin(), h->D.end());
    for (int i = 0; i < N; ++i) h->D[i] = std::sqrt(std::max(h->D[i], 0.));
    h->eigen_uptodate = 1;
    h->genOfEigensysUpdate = h->gen;
    return DDRL_OK;                         // d_B and d_D were written by cmaes_jacobi_finish
  }
  std::vector<float> Cf((size_t)N * h->ldn);
  HIPCHK(hipMemcpyAsync(Cf.data(), h->d_C, Cf.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  std::vector<double> Q((size_t)N * N), e(N + 1);
  for (int i = 0; i < N; ++i)
    for (int j = 0; j <= i; ++j) Q[(size_t)i * N + j] = Q[(size_t)j * N + i] = (double)Cf[(size_t)i * h->ldn + j];
  eig::tridiag(N, Q, h->D.data(), e.data());
  eig::ql(N, h->D.data(), e.data(), Q);
  h->minEW = *std::min_element(h->D.begin(), h->D.end());      // cmaes.c:1915-1916
  h->maxEW = *std::max_element(h->D.begin(), h->D.end());
  for (int i = 0; i < N; ++i) h->D[i] = std::sqrt(h->D[i]);
  h->B = Q;
  h->eigen_uptodate = 1;
  h->genOfEigensysUpdate = h->gen;
  return upload_BD(h);
}

extern "C" int ddrl_cmaes_create(int N, const double *xstart, const double *stds, long seed, int lambda, int device,
                                 ddrl_cmaes_t **out) {
  if (!xstart || !stds || !out || N < 1) return ddrl_fail(DDRL_EINVAL, "bad argument");
  if (seed < 1) return ddrl_fail(DDRL_EINVAL, "seed must be >= 1 (the reference's seed 0 = wall clock, cmaes.c:2406-2410)");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return ddrl_fail(DDRL_EHIP, "no HIP device: libddrl_pop has no CPU path");
  ddrl_cmaes *h = new ddrl_cmaes();
  auto bail = [&](int code) { ddrl_cmaes_destroy(h); return code; };
  if (device >= 0) h->device = device; else if (hipGetDevice(&h->device) != hipSuccess) h->device = 0;
  if (hipSetDevice(h->device) != hipSuccess) return bail(ddrl_fail(DDRL_EHIP, "hipSetDevice failed"));
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return bail(ddrl_fail(DDRL_EHIP, "hipStreamCreate failed"));
  // parameter defaults: cmaes_readpara_init (cmaes.c:2545-2578) + SupplementDefaults (:2772-2851)
  // + SetWeights "log" (:2857-2891)
  h->N = N;
  if (lambda < 2) lambda = 4 + (int)(3 * log((double)N));
  if (lambda > 8192) return bail(ddrl_fail(DDRL_EINVAL, "lambda > 8192 not supported by the LDS ranking kernel"));
  h->lambda = lambda;
  h->mu = lambda / 2;
  h->weights.resize(h->mu);
  double s1 = 0, s2 = 0;
  for (int i = 0; i < h->mu; ++i) h->weights[i] = log(h->mu + 1.) - log(i + 1.);
  for (int i = 0; i < h->mu; ++i) { s1 += h->weights[i]; s2 += h->weights[i] * h->weights[i]; }
  h->mueff = s1 * s1 / s2;
  for (int i = 0; i < h->mu; ++i) h->weights[i] /= s1;
  h->cs = (h->mueff + 2.) / (N + h->mueff + 3.);
  h->ccumcov = 4. / (N + 4);
  h->mucov = h->mueff;
  const double t1 = 2. / ((N + 1.4142) * (N + 1.4142));
  double t2 = (2. * h->mueff - 1.) / ((N + 2.) * (N + 2.) + h->mueff);
  t2 = (t2 > 1) ? 1 : t2;
  t2 = (1. / h->mucov) * t1 + (1. - 1. / h->mucov) * t2;
  h->ccov = t2;
  const double stopMaxFunEvals = 1 * 900 * (N + 3) * (N + 3);
  const double stopMaxIter = ceil((double)(stopMaxFunEvals / lambda));
  h->stopMaxFunEvals = stopMaxFunEvals;
  h->stopMaxIter = stopMaxIter;
  h->damps = 1 * (1 + 2 * std::max(0., sqrt((h->mueff - 1.) / (N + 1.)) - 1))
      * std::max(0.3, 1. - (double)N / (1e-6 + std::min(stopMaxIter, stopMaxFunEvals / lambda))) + h->cs;
  h->modulo = 1. / h->ccov / (double)(N) / 10.;
  rnd_start(h, (unsigned long)seed);
  h->rng_seed = (unsigned int)seed;
  // cmaes_init_final (cmaes.c:279-398)
  double trace = 0.;
  for (int i = 0; i < N; ++i) trace += stds[i] * stds[i];
  const double sigma = sqrt(trace / N);
  h->chiN = sqrt((double)N) * (1. - 1. / (4. * N) + 1. / (21. * N * N));
  h->ldn = ddrl_roundup(N, 16);
  h->mu16 = ddrl_roundup(h->mu, 16);
  h->D.assign(N, 0.);
  h->B.assign((size_t)N * N, 0.);
  std::vector<float> Cf((size_t)N * h->ldn, 0.f);
  for (int i = 0; i < N; ++i) {
    h->B[(size_t)i * N + i] = 1.;
    h->D[i] = stds[i] * sqrt(N / trace);
    Cf[(size_t)i * h->ldn + i] = (float)(h->D[i] * h->D[i]);
  }
  {
    double dtest;
    for (dtest = 1.; dtest && dtest < 1.1 * dtest; dtest *= 2.) if (dtest == dtest + 1.) break;
    h->dMaxSignifKond = dtest / 1000.;
  }
  h->hist.assign(10 + (int)ceil(3. * 10. * N / lambda), 0.);
  h->fitcur.assign(lambda, 0.);
  h->init_stds.assign(stds, stds + N);
  h->minEW = *std::min_element(h->D.begin(), h->D.end()); h->minEW *= h->minEW;
  h->maxEW = *std::max_element(h->D.begin(), h->D.end()); h->maxEW *= h->maxEW;
  int rc;
  const size_t L = lambda, ldn = h->ldn;
  if ((rc = cm_alloc(h, &h->d_X, L * ldn))) return bail(rc);
  if ((rc = cm_alloc(h, &h->d_DZ, L * ldn))) return bail(rc);
  if ((rc = cm_alloc(h, &h->d_B, (size_t)N * ldn))) return bail(rc);
  if ((rc = cm_alloc(h, &h->d_C, (size_t)N * ldn))) return bail(rc);
  if ((rc = cm_alloc(h, &h->d_Y, (size_t)N * h->mu16))) return bail(rc);
  if ((rc = cm_alloc(h, &h->d_fit, L))) return bail(rc);
  if ((rc = cm_alloc(h, &h->d_xmean, N))) return bail(rc);
  if ((rc = cm_alloc(h, &h->d_xold, N))) return bail(rc);
  if ((rc = cm_alloc(h, &h->d_ps, N))) return bail(rc);
  if ((rc = cm_alloc(h, &h->d_pc, N))) return bail(rc);
  if ((rc = cm_alloc(h, &h->d_BDz, N))) return bail(rc);
  if ((rc = cm_alloc(h, &h->d_tmp, N))) return bail(rc);
  if ((rc = cm_alloc(h, &h->d_D, N))) return bail(rc);
  if ((rc = cm_alloc(h, &h->d_w, h->mu))) return bail(rc);
  if ((rc = cm_alloc(h, &h->d_sc, SC_N))) return bail(rc);
  if ((rc = cm_alloc(h, &h->d_xbest, N))) return bail(rc);
  if ((rc = cm_alloc(h, &h->d_index, L))) return bail(rc);
  if ((rc = cm_alloc(h, &h->d_flag, L))) return bail(rc);
  double sc[SC_N] = {sigma, sigma, 0., 0., 0.};
  auto up = [&](void *dst, const void *src, size_t n) { return hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, h->stream); };
  if (up(h->d_C, Cf.data(), Cf.size() * 4) != hipSuccess || up(h->d_xmean, xstart, N * 8) != hipSuccess ||
      up(h->d_xold, xstart, N * 8) != hipSuccess || up(h->d_w, h->weights.data(), h->mu * 8) != hipSuccess ||
      up(h->d_sc, sc, sizeof(sc)) != hipSuccess)
    return bail(ddrl_fail(DDRL_EHIP, "initial upload failed"));
  if ((rc = upload_BD(h))) return bail(rc);
  *out = h;
  return DDRL_OK;
}

extern "C" int ddrl_cmaes_destroy(ddrl_cmaes_t *h) {
  if (!h) return DDRL_OK;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->sweep_graph) (void)hipGraphExecDestroy(h->sweep_graph);
  for (void *p : h->allocs) (void)hipFree(p);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return DDRL_OK;
}

// rows [k0, k0 + n) of DZ <- D o z from the reference stream, then X rows = m + sigma * DZ * B^T
static int sample_rows(ddrl_cmaes *h, int k0, int n) {
  const int N = h->N;
  std::vector<float> dz;
  if (h->device_rng) {
    const dim3 grid((h->ldn / 4 + 63) / 64, n);
    hipLaunchKernelGGL(cmaes_fill_dz_kernel, grid, dim3(64), 0, h->stream, h->d_DZ + (size_t)k0 * h->ldn, h->ldn, h->d_D, n, N,
                       h->rng_seed, h->rng_draw++, k0);
    HIPCHK(hipGetLastError());
  } else {
    dz.assign((size_t)n * h->ldn, 0.f);
    for (int k = 0; k < n; ++k)
      for (int i = 0; i < N; ++i) dz[(size_t)k * h->ldn + i] = (float)(h->D[i] * rnd_gauss(h));   // cmaes.c:645
    HIPCHK(hipMemcpyAsync(h->d_DZ + (size_t)k0 * h->ldn, dz.data(), dz.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
  }
  dim3 grid((N + 63) / 64, (n + 63) / 64);
  hipLaunchKernelGGL(cmaes_gemm_nt<0>, grid, dim3(256), 0, h->stream, h->d_DZ + (size_t)k0 * h->ldn, h->ldn, h->d_B, h->ldn,
                     h->d_X + (size_t)k0 * h->ldn, h->ldn, n, N, h->ldn, h->d_xmean, h->d_sc, 0., 0., 0., -1);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));   // dz is a local buffer
  return DDRL_OK;
}

extern "C" int ddrl_cmaes_sample(ddrl_cmaes_t *h) {
  if (!h) return ddrl_fail(DDRL_EINVAL, "NULL handle");
  HIPCHK(hipSetDevice(h->device));
  int rc = update_eigensystem(h);
  if (rc) return rc;
  if ((rc = sample_rows(h, 0, h->lambda))) return rc;
  if (h->state == 3 || h->gen == 0) h->gen += 1;   // cmaes.c:654-656
  h->state = 1;
  return DDRL_OK;
}

extern "C" int ddrl_cmaes_resample(ddrl_cmaes_t *h, int k) {
  if (!h || k < 0 || k >= h->lambda) return ddrl_fail(DDRL_EINVAL, "index==%d must be between 0 and %d", k, h ? h->lambda : 0);
  HIPCHK(hipSetDevice(h->device));
  return sample_rows(h, k, 1);
}

extern "C" int ddrl_cmaes_make_feasible(ddrl_cmaes_t *h, double bound, int *n_resampled) {
  if (!h) return ddrl_fail(DDRL_EINVAL, "NULL handle");
  HIPCHK(hipSetDevice(h->device));
  int count = 0;
  std::vector<int> flag(h->lambda);
  auto check = [&]() -> int {
    hipLaunchKernelGGL(cmaes_infeasible_kernel, dim3(h->lambda), dim3(256), 0, h->stream, h->d_X, h->ldn, h->N, (float)bound, h->d_flag);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(flag.data(), h->d_flag, h->lambda * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return DDRL_OK;
  };
  int rc = check();
  if (rc) return rc;
  for (int i = 0; i < h->lambda; ++i) {          // CMAESAg.hpp:163-167: in index order, until feasible
    while (flag[i]) {
      if ((rc = sample_rows(h, i, 1))) return rc;
      ++count;
      if (count > 100000) return ddrl_fail(DDRL_ESTATE, "feasibility resampling does not terminate");
      if ((rc = check())) return rc;
    }
  }
  if (n_resampled) *n_resampled = count;
  return DDRL_OK;
}

extern "C" int ddrl_cmaes_get_population(ddrl_cmaes_t *h, double *X) {
  if (!h || !X) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  HIPCHK(hipSetDevice(h->device));
  std::vector<float> xf((size_t)h->lambda * h->ldn);
  HIPCHK(hipMemcpyAsync(xf.data(), h->d_X, xf.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int k = 0; k < h->lambda; ++k)
    for (int i = 0; i < h->N; ++i) X[(size_t)k * h->N + i] = xf[(size_t)k * h->ldn + i];
  return DDRL_OK;
}

extern "C" int ddrl_cmaes_population_dev(ddrl_cmaes_t *h, const float **d_X, int *ld) {
  if (!h || !d_X || !ld) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  *d_X = h->d_X;
  *ld = h->ldn;
  return DDRL_OK;
}

static int rank_dev(ddrl_cmaes *h, const double *d_f, int n, int *d_index) {
  int p2 = 2;
  while (p2 < n) p2 <<= 1;
  const size_t lds = (size_t)p2 * (sizeof(unsigned long long) + sizeof(unsigned int));
  static LdsConfig configured;
  HIPCHK(ddrl_set_lds(cmaes_rank_kernel, lds, configured));
  hipLaunchKernelGGL(cmaes_rank_kernel, dim3(1), dim3(1024), lds, h->stream, d_f, n, p2, d_index);
  HIPCHK(hipGetLastError());
  return DDRL_OK;
}

extern "C" int ddrl_cmaes_rank(ddrl_cmaes_t *h, const double *fitness, int n, int32_t *index_out) {
  if (!h || !fitness || !index_out || n < 1 || n > h->lambda) return ddrl_fail(DDRL_EINVAL, "bad argument (n must be 1..lambda)");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemcpyAsync(h->d_fit, fitness, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  int rc = rank_dev(h, h->d_fit, n, h->d_flag);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(index_out, h->d_flag, n * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return DDRL_OK;
}

// ---------------------------------------------------------------------------------------------
// ddrl_cmaes_rank_noalloc: ddrl_cmaes_rank for callers that cannot allocate (code under HIP graph capture,
// serving loops, harnesses that forbid hipMalloc).  No handle, no hipMalloc / hipFree: the device staging of
// the fitness values and of the resulting index -- h->d_fit and h->d_flag in ddrl_cmaes_rank, buffers the
// library allocates -- lives in a file-scope device array instead.  Each GPU holds its own copy of the array
// at its own address: the address is looked up once per device (hipGetSymbolAddress) with that device current
// and only that pointer, never the array's name, is handed to copies and kernels.
// Every call on a GPU shares that GPU's array: calls that overlap on one GPU race (the caller serialises them).
// ---------------------------------------------------------------------------------------------
#define DDRL_RANK_NOALLOC_MAX 4096           // = the largest population the LDS sort of cmaes_rank_kernel takes
#define DDRL_RANK_SCRATCH_BYTES (DDRL_RANK_NOALLOC_MAX * (sizeof(double) + sizeof(int)))   // 48 KiB per GPU
static __device__ __align__(256) unsigned char scratch[DDRL_RANK_SCRATCH_BYTES];

static int rank_scratch_ptr(unsigned char **out) {
  static unsigned char *per_device[64] = {};
  int dev = -1;
  HIPCHK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return ddrl_fail(DDRL_EINVAL, "device ordinal %d out of range", dev);
  unsigned char *p = __atomic_load_n(&per_device[dev], __ATOMIC_ACQUIRE);
  if (!p) {
    HIPCHK(hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch)));
    if (!p) return ddrl_fail(DDRL_EHIP, "hipGetSymbolAddress(scratch) returned a null address");
    __atomic_store_n(&per_device[dev], p, __ATOMIC_RELEASE);     // two first calls at once look up the same address
  }
  *out = p;
  return DDRL_OK;
}

extern "C" int ddrl_cmaes_rank_noalloc(const double *fitness, int n, int32_t *index_out, void *stream) {
  if (!fitness || !index_out) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  if (n < 1 || n > DDRL_RANK_NOALLOC_MAX)
    return ddrl_fail(DDRL_EINVAL, "ddrl_cmaes_rank_noalloc: n = %d outside 1..%d (the fixed scratch holds %d values)", n,
                     DDRL_RANK_NOALLOC_MAX, DDRL_RANK_NOALLOC_MAX);
  unsigned char *sp = nullptr;
  int rc = rank_scratch_ptr(&sp);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  double *d_f = (double *)sp;                                          // [n] keys
  int *d_idx = (int *)(sp + DDRL_RANK_NOALLOC_MAX * sizeof(double));   // [n] result
  // nothing here must start at zero: the kernel reads d_f[0..n) (just copied) and writes d_idx[0..n) in full
  HIPCHK(hipMemcpyAsync(d_f, fitness, (size_t)n * sizeof(double), hipMemcpyDefault, st));
  int p2 = 2;
  while (p2 < n) p2 <<= 1;
  const size_t lds = (size_t)p2 * (sizeof(unsigned long long) + sizeof(unsigned int));
  static LdsConfig configured;
  HIPCHK(ddrl_set_lds(cmaes_rank_kernel, lds, configured));
  hipLaunchKernelGGL(cmaes_rank_kernel, dim3(1), dim3(1024), lds, st, (const double *)d_f, n, p2, d_idx);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(index_out, d_idx, (size_t)n * sizeof(int), hipMemcpyDefault, st));
  HIPCHK(hipStreamSynchronize(st));
  return DDRL_OK;
}

extern "C" int ddrl_cmaes_update_dev(ddrl_cmaes_t *h, const double *d_fitness) {
  if (!h || !d_fitness) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  if (h->state != 1) return ddrl_fail(DDRL_ESTATE, "cmaes_UpdateDistribution(): call ddrl_cmaes_sample() before the update");
  HIPCHK(hipSetDevice(h->device));
  const int N = h->N;
  h->countevals += h->lambda;
  if (d_fitness != h->d_fit)
    HIPCHK(hipMemcpyAsync(h->d_fit, d_fitness, h->lambda * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  int rc = rank_dev(h, h->d_fit, h->lambda, h->d_index);
  if (rc) return rc;
  {  // rgFuncValue and the best-value history (cmaes.c:817-836), kept on the host for TestForTermination
    int best = 0;
    HIPCHK(hipMemcpyAsync(h->fitcur.data(), h->d_fit, h->lambda * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(&best, h->d_index, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (size_t i = h->hist.size() - 1; i > 0; --i) h->hist[i] = h->hist[i - 1];
    h->hist[0] = h->fitcur[best];
  }
  CmaesVecArgs a;
  a.N = N; a.ldx = h->ldn; a.ldn = h->ldn; a.lambda = h->lambda; a.mu = h->mu;
  a.mueff = h->mueff; a.cs = h->cs; a.damps = h->damps; a.ccumcov = h->ccumcov; a.chiN = h->chiN; a.gen = h->gen;
  a.X = h->d_X; a.B = h->d_B; a.fit = h->d_fit; a.index = h->d_index; a.weights = h->d_w;
  a.xmean = h->d_xmean; a.xold = h->d_xold; a.ps = h->d_ps; a.pc = h->d_pc; a.BDz = h->d_BDz; a.tmp = h->d_tmp;
  a.D = h->d_D; a.sc = h->d_sc; a.xbest = h->d_xbest;
  hipLaunchKernelGGL(cmaes_mean_kernel, dim3((N + 63) / 64), dim3(256), 0, h->stream, a);
  hipLaunchKernelGGL(cmaes_btv_kernel, dim3((N + 63) / 64), dim3(256), 0, h->stream, a);
  hipLaunchKernelGGL(cmaes_bv_kernel, dim3((N + 3) / 4), dim3(256), 0, h->stream, a);
  hipLaunchKernelGGL(cmaes_path_kernel, dim3(1), dim3(1024), 0, h->stream, a);
  HIPCHK(hipGetLastError());
  // Adapt_C2 (cmaes.c:963-1001), diagonalCov == 0
  if (h->ccov != 0.) {
    const double ccov1 = std::min(h->ccov * (1. / h->mucov), 1.);
    const double ccovmu = std::min(h->ccov * (1 - 1. / h->mucov), 1. - ccov1);
    dim3 pg((h->mu16 + 255) / 256, N);
    hipLaunchKernelGGL(cmaes_pack_y, pg, dim3(256), 0, h->stream, h->d_X, h->ldn, h->d_index, h->d_w, h->d_xold, h->d_sc,
                       N, h->mu, h->mu16, h->d_Y);
    HIPCHK(hipGetLastError());
    dim3 grid((N + 63) / 64, (N + 63) / 64);
    hipLaunchKernelGGL(cmaes_gemm_nt<1>, grid, dim3(256), 0, h->stream, h->d_Y, h->mu16, h->d_Y, h->mu16, h->d_C, h->ldn,
                       N, N, h->mu16, h->d_pc, h->d_sc, ccov1, ccovmu, h->ccumcov, -1);
    HIPCHK(hipGetLastError());
    h->eigen_uptodate = 0;
  }
  h->state = 3;
  return DDRL_OK;
}

extern "C" int ddrl_cmaes_update(ddrl_cmaes_t *h, const double *fitness) {
  if (!h || !fitness) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemcpyAsync(h->d_fit, fitness, h->lambda * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return ddrl_cmaes_update_dev(h, h->d_fit);
}

extern "C" int ddrl_cmaes_get(ddrl_cmaes_t *h, const char *key, double *out) {
  if (!h || !key || !out) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  const std::string k(key);
  if (k == "sigma" || k == "fbestever") {
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(out, h->d_sc + (k == "sigma" ? SC_SIGMA : SC_FBEST), sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return DDRL_OK;
  }
  if (k == "eigen_sweeps") *out = h->eigen_sweeps; else if (k == "eigen_persist_timeouts") *out = h->eigen_persist_timeouts; else if (k == "eigen_persist") *out = h->eigen_persist; else if (k == "gen") *out = h->gen; else if (k == "lambda") *out = h->lambda; else if (k == "mu") *out = h->mu;
  else if (k == "N" || k == "dim") *out = h->N; else if (k == "mueff") *out = h->mueff; else if (k == "cs") *out = h->cs;
  else if (k == "damps") *out = h->damps; else if (k == "ccumcov") *out = h->ccumcov; else if (k == "ccov") *out = h->ccov;
  else if (k == "mucov") *out = h->mucov; else if (k == "chiN") *out = h->chiN; else if (k == "modulo") *out = h->modulo;
  else if (k == "eigen_uptodate") *out = h->eigen_uptodate; else if (k == "countevals") *out = h->countevals;
  else return ddrl_fail(DDRL_EINVAL, "cmaes_Get: unknown key '%s'", key);
  return DDRL_OK;
}

extern "C" int ddrl_cmaes_get_vec(ddrl_cmaes_t *h, const char *key, double *out) {
  if (!h || !key || !out) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  HIPCHK(hipSetDevice(h->device));
  const std::string k(key);
  const int N = h->N;
  const double *src = nullptr;
  if (k == "xmean") src = h->d_xmean; else if (k == "ps") src = h->d_ps; else if (k == "pc") src = h->d_pc;
  else if (k == "D") src = h->d_D; else if (k == "xbestever") src = h->d_xbest;
  if (src) {
    HIPCHK(hipMemcpyAsync(out, src, N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return DDRL_OK;
  }
  if (k == "weights") { std::copy(h->weights.begin(), h->weights.end(), out); return DDRL_OK; }
  if (k == "C" || k == "diagC" || k == "B") {
    std::vector<float> M((size_t)N * h->ldn);
    HIPCHK(hipMemcpyAsync(M.data(), k == "B" ? h->d_B : h->d_C, M.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (k == "diagC") for (int i = 0; i < N; ++i) out[i] = M[(size_t)i * h->ldn + i];
    else for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) out[(size_t)i * N + j] = M[(size_t)i * h->ldn + j];
    return DDRL_OK;
  }
  return ddrl_fail(DDRL_EINVAL, "cmaes_GetPtr: unknown key '%s'", key);
}

extern "C" int ddrl_cmaes_get_index(ddrl_cmaes_t *h, int32_t *index) {
  if (!h || !index) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemcpyAsync(index, h->d_index, h->lambda * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return DDRL_OK;
}

// cmaes_TestForTermination (cmaes.c:1536-1667) with the defaults of cmaes_readpara_init (:2555-2559:
// stopFitness off, stopTolFun 1e-12, stopTolFunHist 1e-13, stopTolX 0, stopTolUpXFactor 1e3).  CMAESAg
// only logs the result (CMAESAg.hpp:145-154).  msg receives the '|'-separated names of the criteria
// met; returns their number (0 = keep going) or a negative error code.
extern "C" int ddrl_cmaes_test_for_termination(ddrl_cmaes_t *h, char *msg, int msg_len) {
  if (!h || (msg_len > 0 && !msg)) return ddrl_fail(DDRL_EINVAL, "bad argument");
  HIPCHK(hipSetDevice(h->device));
  const int N = h->N;
  std::vector<double> xmean(N), pc(N), diagC(N);
  double sigma = 0.;
  int rc;
  if ((rc = ddrl_cmaes_get_vec(h, "xmean", xmean.data())) || (rc = ddrl_cmaes_get_vec(h, "pc", pc.data())) ||
      (rc = ddrl_cmaes_get_vec(h, "diagC", diagC.data())) || (rc = ddrl_cmaes_get(h, "sigma", &sigma))) return rc;
  std::string out;
  int cnt = 0;
  auto add = [&](const char *name) { if (cnt++) out += "|"; out += name; };
  const double stopTolFun = 1e-12, stopTolFunHist = 1e-13, stopTolX = 0., stopTolUpXFactor = 1e3;
  const int nhist = (int)h->hist.size();
  const int nh = std::max(1, (int)std::min(h->gen, (double)nhist));     // rgdouMax over 0 elements reads element 0
  double range = std::max(*std::max_element(h->hist.begin(), h->hist.begin() + nh), *std::max_element(h->fitcur.begin(), h->fitcur.end()))
               - std::min(*std::min_element(h->hist.begin(), h->hist.begin() + nh), *std::min_element(h->fitcur.begin(), h->fitcur.end()));
  if (h->gen > 0 && range <= stopTolFun) add("TolFun");
  if (h->gen > nhist) {
    range = *std::max_element(h->hist.begin(), h->hist.end()) - *std::min_element(h->hist.begin(), h->hist.end());
    if (range <= stopTolFunHist) add("TolFunHist");
  }
  int cTemp = 0;
  for (int i = 0; i < N; ++i) {
    cTemp += (sigma * std::sqrt(diagC[i]) < stopTolX) ? 1 : 0;
    cTemp += (sigma * pc[i] < stopTolX) ? 1 : 0;
  }
  if (cTemp == 2 * N) add("TolX");
  for (int i = 0; i < N; ++i)
    if (sigma * std::sqrt(diagC[i]) > stopTolUpXFactor * h->init_stds[i]) { add("TolUpX"); break; }
  if (h->maxEW >= h->minEW * h->dMaxSignifKond) add("ConditionNumber");
  if (!(0. >= h->gen)) {                                    // !flgdiag
    for (int ax = 0; ax < N; ++ax) {
      const double fac = 0.1 * sigma * h->D[ax];
      int k = 0;
      for (; k < N; ++k) if (xmean[k] != xmean[k] + fac * h->B[(size_t)k * N + ax]) break;
      if (k == N) { add("NoEffectAxis"); break; }
    }
  }
  for (int k = 0; k < N; ++k)
    if (xmean[k] == xmean[k] + 0.2 * sigma * std::sqrt(diagC[k])) { add("NoEffectCoordinate"); break; }
  if (h->countevals >= h->stopMaxFunEvals) add("MaxFunEvals");
  if (h->gen >= h->stopMaxIter) add("MaxIter");
  if (msg_len > 0) {
    strncpy(msg, out.c_str(), msg_len - 1);
    msg[msg_len - 1] = 0;
  }
  return cnt;
}

// options (throughput modes; same algorithm, not seed-for-seed comparable with c-cmaes):
//   "device_rng" (0/1): draw z on the device (Philox) instead of evaluating c-cmaes' sequential
//   generator on the host;  "device_eigen" (0/1): eigendecomposition by one-sided Jacobi on the GPU (device_eigen above);
//   "eigen_block" (1/0): its rounds over blocks of rows (default) or over single rows.
//   "eigen_persist" (1/0): all sweeps of a decomposition in one launch (default, N <= 640) or one launch per round.
extern "C" int ddrl_cmaes_set_option(ddrl_cmaes_t *h, const char *key, int value) {
  if (!h || !key) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  if (std::string(key) == "device_rng") { h->device_rng = value != 0; return DDRL_OK; }
  if (std::string(key) == "device_eigen") {
    if (value && !h->d_A) {
      HIPCHK(hipSetDevice(h->device));
      int rc;
      if ((rc = cm_alloc(h, &h->d_A, (size_t)h->N * h->N)) || (rc = cm_alloc(h, &h->d_ev, h->N)) ||
          (rc = cm_alloc(h, &h->d_W, (size_t)h->N * h->N)) || (rc = cm_alloc(h, &h->d_Cs, (size_t)h->N * h->N)) ||
          (rc = cm_alloc(h, &h->d_info, JACOBI_CTL_INTS))) return rc;
    }
    h->device_eigen = value != 0;
    h->eig_warm = 0;
    return DDRL_OK;
  }
  if (std::string(key) == "eigen_persist_wait_ms") { h->eigen_persist_wait_ms = std::max(1, std::min(value, 10000)); return DDRL_OK; }
  if (std::string(key) == "eigen_persist_drop") { h->eigen_persist_drop = std::max(0, value); return DDRL_OK; }     // tests only
  if (std::string(key) == "eigen_persist") { h->eigen_persist = value != 0; return DDRL_OK; }  // 1 (default): one launch per decomposition; 0: one per round
  if (std::string(key) == "eigen_block") { h->eigen_block = value != 0; return DDRL_OK; }      // 1 (default): blocked rounds; 0: one row pair per workgroup
  return ddrl_fail(DDRL_EINVAL, "unknown option '%s'", key);
}

// ---- run state: what cmaes_WriteToFile(evo, "resume", ...) / cmaes_resume_distribution (cmaes.c:426-545;
// CMAESAg.hpp:279-308) carry over, plus what they leave out and a bit-identical continuation needs: the
// random stream, the current generation's samples, the eigensystem schedule and the termination history
int ddrl_cmaes_checkpoint(ddrl_cmaes_t *h, Ckpt &c) {
  c.st = h->stream;
  int rc;
  const int meta[3] = {h->N, h->lambda, h->mu};
  if ((rc = c.same("cmaes.meta", meta, sizeof(meta)))) return rc;
  const size_t N = h->N, L = h->lambda, ldn = h->ldn;
  if ((rc = c.val("cmaes.aktseed", h->aktseed)) || (rc = c.val("cmaes.aktrand", h->aktrand)) || (rc = c.host("cmaes.rgrand", h->rgrand, sizeof(h->rgrand))) ||
      (rc = c.val("cmaes.flgstored", h->flgstored)) || (rc = c.val("cmaes.hold", h->hold)) || (rc = c.val("cmaes.gen", h->gen)) ||
      (rc = c.val("cmaes.countevals", h->countevals)) || (rc = c.val("cmaes.genOfEigensysUpdate", h->genOfEigensysUpdate)) ||
      (rc = c.val("cmaes.state", h->state)) || (rc = c.val("cmaes.eigen_uptodate", h->eigen_uptodate)) ||
      (rc = c.val("cmaes.device_rng", h->device_rng)) || (rc = c.val("cmaes.rng_seed", h->rng_seed)) || (rc = c.val("cmaes.rng_draw", h->rng_draw)) ||
      (rc = c.val("cmaes.minEW", h->minEW)) || (rc = c.val("cmaes.maxEW", h->maxEW)) || (rc = c.val("cmaes.dMaxSignifKond", h->dMaxSignifKond)) ||
      (rc = c.vec("cmaes.D", h->D)) || (rc = c.vec("cmaes.B", h->B)) || (rc = c.vec("cmaes.hist", h->hist)) || (rc = c.vec("cmaes.fitcur", h->fitcur)))
    return rc;
  if (h->D.size() != N || h->B.size() != N * N) return ddrl_fail(DDRL_EIO, "%s: eigensystem of the wrong size", c.path.c_str());
  if ((rc = c.dev("cmaes.X", h->d_X, L * ldn * sizeof(float))) || (rc = c.dev("cmaes.DZ", h->d_DZ, L * ldn * sizeof(float))) ||
      (rc = c.dev("cmaes.Bdev", h->d_B, N * ldn * sizeof(float))) || (rc = c.dev("cmaes.C", h->d_C, N * ldn * sizeof(float))) ||
      (rc = c.dev("cmaes.fit", h->d_fit, L * sizeof(double))) || (rc = c.dev("cmaes.xmean", h->d_xmean, N * sizeof(double))) ||
      (rc = c.dev("cmaes.xold", h->d_xold, N * sizeof(double))) || (rc = c.dev("cmaes.ps", h->d_ps, N * sizeof(double))) ||
      (rc = c.dev("cmaes.pc", h->d_pc, N * sizeof(double))) || (rc = c.dev("cmaes.BDz", h->d_BDz, N * sizeof(double))) ||
      (rc = c.dev("cmaes.Ddev", h->d_D, N * sizeof(double))) || (rc = c.dev("cmaes.sc", h->d_sc, SC_N * sizeof(double))) ||
      (rc = c.dev("cmaes.xbest", h->d_xbest, N * sizeof(double))) || (rc = c.dev("cmaes.index", h->d_index, L * sizeof(int))))
    return rc;
  // throughput mode: the eigenvectors the next Jacobi iteration starts from (a cold start would give another, equally
  // valid basis: the continuation would no longer be bit-identical)
  int dev_eig = h->device_eigen;
  if ((rc = c.same("cmaes.device_eigen", &dev_eig, sizeof(dev_eig)))) return rc;
  if (h->device_eigen) {
    if ((rc = c.val("cmaes.eig_warm", h->eig_warm)) || (rc = c.dev("cmaes.W", h->d_W, N * N * sizeof(double)))) return rc;
  }
  return DDRL_OK;
}

static int cmaes_run_file(ddrl_cmaes_t *h, const char *path, bool save) {
  if (!h || !path) return ddrl_fail(DDRL_EINVAL, "NULL argument");
  if (hipSetDevice(h->device) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return ddrl_fail(DDRL_EHIP, "device not available");
  Ckpt c;
  int rc;
  if ((rc = c.open(path, save)) || (rc = ddrl_cmaes_checkpoint(h, c))) return rc;
  return c.finish();
}
extern "C" int ddrl_cmaes_save_run(ddrl_cmaes_t *h, const char *path) { return cmaes_run_file(h, path, true); }
extern "C" int ddrl_cmaes_load_previous_run(ddrl_cmaes_t *h, const char *path) { return cmaes_run_file(h, path, false); }
