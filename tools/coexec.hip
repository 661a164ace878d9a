// tools/coexec.hip -- do fp32 MFMAs and fp32 VALU work of two waves on ONE SIMD overlap on gfx950?
// (development tool)  One 512-thread workgroup per CU: waves 0-3 (one per SIMD) run role A, waves 4-7 role B.
//   role 0: idle     role 1: v_mfma_f32_16x16x4_f32 chain pairs     role 2: independent v_fma_f32     role 3: v_pk_fma_f32
//   role 4: mfma + valu interleaved in ONE wave (8 fma per mfma)
// prints cycles per loop iteration for each role (s_memtime), alone and together.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int ROLE>
__device__ __forceinline__ float work(int iters, float seed) {
  f32x4 a0 = {seed, 1, 2, 3}, a1 = {seed, 2, 3, 4};
  float x0 = seed, x1 = seed + 1, x2 = seed + 2, x3 = seed + 3, x4 = seed + 4, x5 = seed + 5, x6 = seed + 6, x7 = seed + 7;
  f32x2 p0 = {seed, 1}, p1 = {seed, 2}, p2 = {seed, 3}, p3 = {seed, 4};
  for (int i = 0; i < iters; ++i) {
    if (ROLE == 1 || ROLE == 4) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, 1.0001f, a0, 0, 0, 0);
        if (ROLE == 4) {
          x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0001f, 0.5f); x2 = fmaf(x2, 1.0001f, 0.5f); x3 = fmaf(x3, 1.0001f, 0.5f);
        }
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, 1.0002f, a1, 0, 0, 0);
        if (ROLE == 4) {
          x4 = fmaf(x4, 1.0001f, 0.5f); x5 = fmaf(x5, 1.0001f, 0.5f); x6 = fmaf(x6, 1.0001f, 0.5f); x7 = fmaf(x7, 1.0001f, 0.5f);
        }
      }
    }
    if (ROLE == 2) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0001f, 0.5f); x2 = fmaf(x2, 1.0001f, 0.5f); x3 = fmaf(x3, 1.0001f, 0.5f);
        x4 = fmaf(x4, 1.0001f, 0.5f); x5 = fmaf(x5, 1.0001f, 0.5f); x6 = fmaf(x6, 1.0001f, 0.5f); x7 = fmaf(x7, 1.0001f, 0.5f);
      }
    }
    if (ROLE == 3) {
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        p0 = p0 * 1.0001f + 0.5f; p1 = p1 * 1.0001f + 0.5f; p2 = p2 * 1.0001f + 0.5f; p3 = p3 * 1.0001f + 0.5f;
      }
    }
  }
  return a0[0] + a1[1] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0[0] + p1[1] + p2[0] + p3[1];
}

template <int RA, int RB>
__global__ void __launch_bounds__(512, 2) k(int iters, float *out, long long *cyc) {
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  const long long t0 = clock64();
  float r = wave < 4 ? work<RA>(iters, (float)threadIdx.x) : work<RB>(iters, (float)threadIdx.x);
  const long long t1 = clock64();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
  if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int RA, int RB>
void run(const char *name, int iters, float *out, long long *cyc) {
  hipLaunchKernelGGL((k<RA, RB>), dim3(256), dim3(512), 0, 0, iters, out, cyc);
  hipLaunchKernelGGL((k<RA, RB>), dim3(256), dim3(512), 0, 0, iters, out, cyc);
  (void)hipDeviceSynchronize();
  long long h[2048];
  (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  double a = 0, b = 0;
  for (int i = 0; i < 256; ++i) for (int w = 0; w < 8; ++w) (w < 4 ? a : b) += (double)h[i * 8 + w];
  printf("%-34s waves 0-3: %8.1f cycles/iter   waves 4-7: %8.1f cycles/iter\n", name, a / 1024 / iters, b / 1024 / iters);
}

int main() {
  float *out; long long *cyc;
  (void)hipMalloc(&out, 4096); (void)hipMalloc(&cyc, 2048 * 8);
  const int it = 2000;
  run<1, 0>("mfma (16 per iter) | idle", it, out, cyc);
  run<2, 0>("128 v_fma | idle", it, out, cyc);
  run<3, 0>("128 v_pk_fma | idle", it, out, cyc);
  run<1, 1>("mfma | mfma", it, out, cyc);
  run<2, 2>("v_fma | v_fma", it, out, cyc);
  run<1, 2>("mfma | 128 v_fma", it, out, cyc);
  run<1, 3>("mfma | 128 v_pk_fma", it, out, cyc);
  run<4, 0>("mfma + 64 v_fma interleaved | idle", it, out, cyc);
  run<4, 4>("interleaved | interleaved", it, out, cyc);
  return 0;
}
