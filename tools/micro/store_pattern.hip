// Store-pattern probe: how fast can a grid of waves write a [pixels][64 channels] fp32 tensor when each store
// instruction covers (a) 4 pixels x 256 contiguous bytes (the LDS-staged epilogues), (b) 16 pixels x 64-byte segments
// (stores straight from the transposed MFMA accumulator layout, four instructions complete a pixel's 256 bytes),
// (c) like (b) with the four instructions of a pixel issued back to back, (d) a plain linear fill.
//   hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern && ./store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kC = 64;

template <int MODE>
__global__ __launch_bounds__(256) void k_store(float* __restrict__ out, long npix, int tiles_per_block) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const f4 v = {1.f, 2.f, 3.f, (float)lane};
  for (int t = 0; t < tiles_per_block; ++t) {
    const long tile = (long)blockIdx.x * tiles_per_block + t;  // 256 pixels per tile, 64 per wave
    const long p0 = tile * 256 + wave * 64;
    if (p0 >= npix) return;
    if (MODE == 0) {        // 4 pixels x 256 B per instruction
      const int q4 = lane & 15, r = lane >> 4;
#pragma unroll
      for (int i = 0; i < 16; ++i) *reinterpret_cast<f4*>(out + (p0 + i * 4 + r) * kC + q4 * 4) = v;
    } else if (MODE == 1) { // 16 pixels x 64 B per instruction; nt outer (a pixel's line is completed 4 x 16 stores later)
      const int j = lane & 15, kq = lane >> 4;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<f4*>(out + (p0 + mt * 16 + j) * kC + nt * 16 + kq * 4) = v;
    } else if (MODE == 2) { // same segments, the 4 instructions of a pixel group back to back
      const int j = lane & 15, kq = lane >> 4;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<f4*>(out + (p0 + mt * 16 + j) * kC + nt * 16 + kq * 4) = v;
    } else {                // linear
#pragma unroll
      for (int i = 0; i < 16; ++i) *reinterpret_cast<f4*>(out + p0 * kC + (i * 64 + lane) * 4) = v;
    }
  }
}

template <int MODE>
static void run(float* d, long npix, const char* name, int tpb) {
  const long tiles = npix / 256;
  const int grid = (int)((tiles + tpb - 1) / tpb);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int it = 0; it < 12; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_store<MODE>, dim3(grid), dim3(256), 0, 0, d, npix, tpb);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (it >= 2 && ms < best) best = ms;
  }
  printf("%-40s tiles/block %3d  %.3f ms  %.2f TB/s\n", name, tpb, best, npix * kC * 4.0 / best / 1e9);
}

int main() {
  const long npix = 64L * 256 * 256;  // the c2 first layer's output: 1.07 GB
  float* d;
  hipMalloc(&d, npix * kC * 4);
  for (int tpb : {1, 64}) {
    run<3>(d, npix, "linear fill", tpb);
    run<0>(d, npix, "4 px x 256 B per instruction", tpb);
    run<1>(d, npix, "16 px x 64 B, channel tile outer", tpb);
    run<2>(d, npix, "16 px x 64 B, pixel tile outer", tpb);
  }
  return 0;
}
