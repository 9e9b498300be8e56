// Issue rate of v_mfma_f32_16x16x32_f16 against the distance between two MFMAs on the same accumulator and the number
// of waves per SIMD (round 5: what paces the consumers' tap loop?).  hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int D, int MODE>
__global__ void k(float* out, long long* clk, int iters, float seed) {
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(seed * (threadIdx.x % 7 + e)); b[e] = (_Float16)(seed * (threadIdx.x % 5 - e)); }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 24; ++m) {
      if (MODE == 0) {            // round-robin over D accumulators
        acc[m % D] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m % D], 0, 0, 0);
      } else {                    // k_conv_bfr's order: groups of 6 on 2 accumulators (a0 a1 a0 a1 a0 a1), groups rotate over D/2 pairs
        const int g = (m / 6) % (D / 2), i = m & 1;
        acc[2 * g + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[2 * g + i], 0, 0, 0);
      }
    }
  }
  const long long t1 = clock64();
  f32x4 s = acc[0];
  for (int i = 1; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if ((threadIdx.x & 63) == 0) clk[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int D, int MODE>
void run(int waves_per_simd, float seed) {
  const int threads = 256 * waves_per_simd, blocks = 256, iters = 2000;
  float* out; long long* clk;
  hipMalloc(&out, sizeof(float) * threads * blocks);
  hipMalloc(&clk, sizeof(long long) * blocks * threads / 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<D, MODE><<<blocks, threads>>>(out, clk, 10, seed);
  hipEventRecord(e0);
  k<D, MODE><<<blocks, threads>>>(out, clk, iters, seed);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks * threads / 64);
  hipMemcpy(h.data(), clk, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
  double sum = 0; for (auto v : h) sum += v;
  const double per = sum / h.size() / (iters * 24.0);
  const double tf = 2.0 * 16 * 16 * 32 * 24.0 * iters * (threads / 64) * blocks / (ms * 1e-3) / 1e12;
  printf("mode %d dist %d waves/SIMD %d data %s: %6.2f clk per MFMA and wave, %6.2f per SIMD, %7.1f TF, %.3f GHz eff\n", MODE, D, waves_per_simd,
         seed == 0.f ? "zero" : "rand", per, per / waves_per_simd, tf, sum / h.size() / (ms * 1e-3) / 1e9);
  hipFree(out); hipFree(clk);
}

int main() {
  for (float seed : {0.f, 0.37f}) {
    for (int w : {1, 2}) {
      run<1, 0>(w, seed); run<2, 0>(w, seed); run<3, 0>(w, seed); run<4, 0>(w, seed); run<6, 0>(w, seed); run<8, 0>(w, seed);
      run<2, 1>(w, seed); run<4, 1>(w, seed); run<8, 1>(w, seed);
    }
  }
  return 0;
}
