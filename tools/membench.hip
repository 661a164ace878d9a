// tools/membench.hip -- what HBM rate can the optimiser epilogue's access pattern reach, and how do
// FETCH_SIZE / WRITE_SIZE count it?  (development tool, not part of the library)
//
// Emulates one optimiser sweep over G agent blocks of NP floats: 7 arrays read, 5 written per
// parameter (w, m, v, wt, fisher rewritten; fhat, wstar read only), one 256-thread workgroup per
// agent, each wave walking 16 x 16 tiles of a [rows x kp] weight matrix.
//   mode 0: the MFMA accumulator layout: lane (c, g) owns row c, floats 4g..4g+3 of a 16-float
//           segment  -> 16 rows x 64 B per wave instruction
//   mode 1: two adjacent tiles exchanged between lane pairs (c, c^8): 8 rows x 128 B per instruction
//   mode 2: flat, 64 lanes x 16 B contiguous = 1 KiB per instruction
//   mode 3: the seven arrays interleaved per KiB block ([w|m|v|wt|fisher|fhat|wstar] = 7 KiB contiguous
//           per 256 parameters): does a longer contiguous run per DRAM page visit raise the rate?
//   mode 4: as mode 2 with two blocks (14 KiB of loads) in flight per wave
//   mode 5: plain copy, one array read, one written (the ceiling for mixed read / write traffic)
// usage: membench <mode> [G] [iters]      prints GB/s of algorithmic bytes (48 B per parameter)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ROWS = 100, KP = 100, NP = 12800;   // floats per agent block (padded to 256 B)

struct Arrays { float *a[7]; };

__device__ __forceinline__ void sweep4(const Arrays A, size_t i, bool ok) {
  f32x4 r[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) r[k] = *(const f32x4 *)(A.a[k] + i);
  f32x4 g = r[5] * (r[0] - r[6]);
  f32x4 m = 0.1f * g + 0.9f * r[1], v = 0.001f * g * g + 0.999f * r[2];
  f32x4 w = r[0] - 1e-3f * m / (v + 1e-8f);
  f32x4 wt = 0.01f * w + 0.99f * r[3], f = r[4] * 0.999f + g * g;
  if (ok) {
    *(f32x4 *)(A.a[0] + i) = w; *(f32x4 *)(A.a[1] + i) = m; *(f32x4 *)(A.a[2] + i) = v;
    *(f32x4 *)(A.a[3] + i) = wt; *(f32x4 *)(A.a[4] + i) = f;
  }
}

template <int MODE>
__global__ void __launch_bounds__(256) sweep(const Arrays A) {
  const int agent = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int g = lane >> 4, c = lane & 15;
  const size_t base = (size_t)agent * NP;
  if (MODE == 2) {
    for (int i = tid * 4; i < ROWS * KP; i += 1024) sweep4(A, base + i, true);
    return;
  }
  if (MODE == 3) {
    // block b of the agent: floats [b*7*256, (b+1)*7*256) of a[0]; array k of the block at + k*256
    float *blk = A.a[0] + (size_t)agent * 7 * NP;
    for (int b = wave; b * 256 < ROWS * KP; b += 4) {
      float *p = blk + (size_t)b * 7 * 256 + lane * 4;
      f32x4 r[7];
#pragma unroll
      for (int k = 0; k < 7; ++k) r[k] = *(const f32x4 *)(p + k * 256);
      f32x4 g = r[5] * (r[0] - r[6]);
      f32x4 m = 0.1f * g + 0.9f * r[1], v = 0.001f * g * g + 0.999f * r[2];
      f32x4 w = r[0] - 1e-3f * m / (v + 1e-8f);
      f32x4 wt = 0.01f * w + 0.99f * r[3], f = r[4] * 0.999f + g * g;
      *(f32x4 *)(p) = w; *(f32x4 *)(p + 256) = m; *(f32x4 *)(p + 512) = v; *(f32x4 *)(p + 768) = wt; *(f32x4 *)(p + 1024) = f;
    }
    return;
  }
  if (MODE == 4) {
    int i = tid * 4;
    for (; i + 1024 < ROWS * KP; i += 2048) {
      f32x4 r[14];
#pragma unroll
      for (int k = 0; k < 7; ++k) { r[k] = *(const f32x4 *)(A.a[k] + base + i); r[7 + k] = *(const f32x4 *)(A.a[k] + base + i + 1024); }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x4 *q = r + 7 * h;
        f32x4 g = q[5] * (q[0] - q[6]);
        f32x4 m = 0.1f * g + 0.9f * q[1], v = 0.001f * g * g + 0.999f * q[2];
        f32x4 w = q[0] - 1e-3f * m / (v + 1e-8f);
        f32x4 wt = 0.01f * w + 0.99f * q[3], f = q[4] * 0.999f + g * g;
        const size_t j = base + i + 1024 * h;
        *(f32x4 *)(A.a[0] + j) = w; *(f32x4 *)(A.a[1] + j) = m; *(f32x4 *)(A.a[2] + j) = v;
        *(f32x4 *)(A.a[3] + j) = wt; *(f32x4 *)(A.a[4] + j) = f;
      }
    }
    for (; i < ROWS * KP; i += 1024) sweep4(A, base + i, true);
    return;
  }
  if (MODE == 5) {
    for (int k = 0; k < 6; ++k)     // 6 arrays' worth of bytes copied a[k] -> a[(k+1)%7]: 24 B read + 24 B written per parameter
      for (int i = tid * 4; i < ROWS * KP; i += 1024) *(f32x4 *)(A.a[6] + base + i) = *(const f32x4 *)(A.a[k] + base + i);
    return;
  }
  const int NT = (ROWS + 15) / 16, KT = (KP + 15) / 16;
  if (MODE == 0) {
    for (int p = wave; p < NT * KT; p += 4) {
      const int nt = p / KT, kt = p - nt * KT;
      const int n = nt * 16 + c, k = kt * 16 + 4 * g;
      const bool ok = n < ROWS && k < KP;
      sweep4(A, base + (ok ? n * KP + k : 0), ok);
    }
  } else {
    const int KT2 = (KT + 1) / 2;
    for (int p = wave; p < NT * KT2; p += 4) {
      const int nt = p / KT2, kp2 = p - nt * KT2;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int n = nt * 16 + half * 8 + (c & 7), k = kp2 * 32 + (c >> 3) * 16 + 4 * g;
        const bool ok = n < ROWS && k < KP;
        sweep4(A, base + (ok ? n * KP + k : 0), ok);
      }
    }
  }
}

int main(int argc, char **argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const int G = argc > 2 ? atoi(argv[2]) : 2048;
  const int iters = argc > 3 ? atoi(argv[3]) : 20;
  Arrays A;
  const size_t n = (size_t)G * NP;
  if (mode == 3) {
    if (hipMalloc(&A.a[0], n * 4 * 7) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(A.a[0], 0, n * 4 * 7);
    for (int k = 1; k < 7; ++k) A.a[k] = A.a[0];
  } else
  for (int k = 0; k < 7; ++k) {
    if (hipMalloc(&A.a[k], n * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(A.a[k], 0, n * 4);
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto launch = [&]() {
    if (mode == 0) hipLaunchKernelGGL(sweep<0>, dim3(G), dim3(256), 0, 0, A);
    else if (mode == 1) hipLaunchKernelGGL(sweep<1>, dim3(G), dim3(256), 0, 0, A);
    else if (mode == 3) hipLaunchKernelGGL(sweep<3>, dim3(G), dim3(256), 0, 0, A);
    else if (mode == 4) hipLaunchKernelGGL(sweep<4>, dim3(G), dim3(256), 0, 0, A);
    else if (mode == 5) hipLaunchKernelGGL(sweep<5>, dim3(G), dim3(256), 0, 0, A);
    else hipLaunchKernelGGL(sweep<2>, dim3(G), dim3(256), 0, 0, A);
  };
  for (int i = 0; i < 3; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)G * ROWS * KP * 48.0;
  printf("mode %d G %d: %.1f us/launch, %.0f GB/s algorithmic (read %.1f MB, write %.1f MB per launch)\n", mode, G,
         ms * 1e3 / iters, bytes / (ms * 1e-3 / iters) / 1e9, (double)G * ROWS * KP * 28 / 1e6, (double)G * ROWS * KP * 20 / 1e6);
  return hipGetLastError() != hipSuccess;
}
