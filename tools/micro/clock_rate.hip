// Is s_memtime (clock64) the shader clock?  Ticks of s_memtime per microsecond of s_memrealtime (wall_clock64, 100 MHz) in
// a kernel that only spins, and in kernels that run an MFMA chain on 1 / 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ void k(float* out, long long* res, int iters) {
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.37f * (threadIdx.x % 7 + e)); b[e] = (_Float16)(0.11f * (threadIdx.x % 5 - e)); }
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const long long w0 = wall_clock64(), t0 = clock64();
  if (MODE == 0) {
    for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_sleep(16);
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 24; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m & 3], 0, 0, 0);
    }
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1];
  if (blockIdx.x == 7 && threadIdx.x == 0) { res[0] = t1 - t0; res[1] = w1 - w0; }
}
// streaming copy (16 bytes per lane and iteration), block 7 wave 0 reports
__global__ void kcopy(const float4* src, float4* dst, long long* res, int iters, size_t stride) {
  const long long w0 = wall_clock64(), t0 = clock64();
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int it = 0; it < iters; ++it, i += stride) dst[i] = src[i];
  const long long t1 = clock64(), w1 = wall_clock64();
  if (blockIdx.x == 7 && threadIdx.x == 0) { res[0] = t1 - t0; res[1] = w1 - w0; }
}
int main() {
  {
    const size_t n = (size_t)1 << 26;  // float4 elements: 1 GiB per buffer
    float4 *a, *b; long long* r; long long hh[2];
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&r, 16);
    hipMemset(a, 1, n * 16);
    for (int rep = 0; rep < 2; ++rep) {
      const int blocks = 2048, threads = 256, iters = (int)(n / (blocks * threads));
      kcopy<<<blocks, threads>>>(a, b, r, iters, (size_t)blocks * threads);
      hipMemcpy(hh, r, 16, hipMemcpyDeviceToHost);
      printf("copy 1 GiB (8 blocks/CU): %lld ticks in %.1f us -> %.3f GHz by s_memtime;  %.2f TB/s read+write\n", hh[0], hh[1] / 100.0, hh[0] / (hh[1] / 100.0) / 1e3,
             2.0 * n * 16 / (hh[1] / 100.0 * 1e-6) / 1e12);
    }
  }
  float* out; long long* res; long long h[2];
  hipMalloc(&out, 4 * 256 * 512); hipMalloc(&res, 16);
  for (int rep = 0; rep < 2; ++rep) {
    k<0><<<256, 256>>>(out, res, 20000); hipMemcpy(h, res, 16, hipMemcpyDeviceToHost);
    printf("spin (s_sleep), 1 wave/SIMD: %lld ticks in %.1f us -> %.3f GHz\n", h[0], h[1] / 100.0, h[0] / (h[1] / 100.0) / 1e3);
    k<1><<<256, 256>>>(out, res, 3000); hipMemcpy(h, res, 16, hipMemcpyDeviceToHost);
    printf("mfma chain, 1 wave/SIMD:     %lld ticks in %.1f us -> %.3f GHz, %.2f ticks per MFMA\n", h[0], h[1] / 100.0, h[0] / (h[1] / 100.0) / 1e3, h[0] / 72000.0);
    k<1><<<256, 512>>>(out, res, 3000); hipMemcpy(h, res, 16, hipMemcpyDeviceToHost);
    printf("mfma chain, 2 waves/SIMD:    %lld ticks in %.1f us -> %.3f GHz, %.2f ticks per MFMA and wave\n", h[0], h[1] / 100.0, h[0] / (h[1] / 100.0) / 1e3, h[0] / 72000.0);
    k<1><<<256, 1024>>>(out, res, 3000); hipMemcpy(h, res, 16, hipMemcpyDeviceToHost);
    printf("mfma chain, 4 waves/SIMD:    %lld ticks in %.1f us -> %.3f GHz, %.2f ticks per MFMA and wave\n", h[0], h[1] / 100.0, h[0] / (h[1] / 100.0) / 1e3, h[0] / 72000.0);
  }
  return 0;
}
