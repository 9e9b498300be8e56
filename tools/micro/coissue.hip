// What does a wave of plain VALU / LDS / VMEM work cost a dense MFMA wave on the SAME SIMD (and vice versa)?
// 512-thread blocks, one per CU: waves 0-3 run an MFMA chain (one per SIMD), waves 4-7 a partner loop of one kind.
// Both loops run for a fixed count; every wave reports its own elapsed s_memtime ticks.
//   hipcc --offload-arch=gfx950 -O3 coissue.hip -o coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, long long* clk, const float* src, int mf_iters, int pt_iters, float seed) {
  __shared__ uint4 lds[4096];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  long long t0, t1;
  float res = 0.f;
  __syncthreads();
  if (wave < 4) {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(seed * (lane % 7 + e)); b[e] = (_Float16)(seed * (lane % 5 - e)); }
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    t0 = clock64();
    for (int it = 0; it < mf_iters; ++it) {
#pragma unroll
      for (int m = 0; m < 24; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m & 3], 0, 0, 0);
    }
    t1 = clock64();
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
    res = s[0] + s[1] + s[2] + s[3];
  } else {
    float x[8];
    for (int e = 0; e < 8; ++e) x[e] = seed * (lane + e) + 1.f;
    uint4 w = {1u, 2u, 3u, 4u};
    const float* p = src + (size_t)blockIdx.x * 65536 + (wave - 4) * 16384 + lane * 4;
    t0 = clock64();
    for (int it = 0; it < pt_iters; ++it) {
      if (KIND == 1) {          // 16 plain fp32 fma
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = __builtin_fmaf(x[e], 1.0001f, 0.5f);
      } else if (KIND == 2) {   // 16 packed fp32 fma (v_pk_fma_f32)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            f32x2 v = {x[e], x[e + 1]};
            v = __builtin_elementwise_fma(v, (f32x2){1.0001f, 1.0001f}, (f32x2){0.5f, 0.5f});
            x[e] = v[0]; x[e + 1] = v[1];
          }
      } else if (KIND == 3) {   // the f16x3 split of 8 values (split8h)
        f16x8 h, m;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xs = x[e] * seed;
          const _Float16 hh = (_Float16)xs;
          h[e] = hh;
          m[e] = (_Float16)(xs - (float)hh);
        }
        w = __builtin_bit_cast(uint4, h);
        const uint4 w2 = __builtin_bit_cast(uint4, m);
        x[0] += (float)(w.x ^ w2.y); x[3] += (float)(w.z ^ w2.w); x[5] += (float)(w.y ^ w2.x);
      } else if (KIND == 4) {   // 4 ds_write_b128
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[((wave - 4) * 1024 + r * 64 + lane) & 4095] = w;
        w.x += 1u;
      } else if (KIND == 5) {   // 4 ds_read_b128
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint4 v = lds[((wave - 4) * 1024 + r * 64 + lane + it) & 4095];
          w.x ^= v.x; w.y ^= v.w;
        }
      } else if (KIND == 6) {   // 4 global loads of 16 bytes per lane, waited for
        f32x4 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = *reinterpret_cast<const f32x4*>(p + ((it * 4 + r) & 63) * 256);
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] += v[r][0];
      } else if (KIND == 7) {   // 16 v_max_f32
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = fmaxf(x[e], x[(e + 1) & 7] - 1.f);
      } else {
        __builtin_amdgcn_s_sleep(8);
      }
    }
    t1 = clock64();
    for (int e = 0; e < 8; ++e) res += x[e];
    res += (float)(w.x + w.y);
  }
  out[blockIdx.x * 512 + threadIdx.x] = res;
  if (lane == 0) clk[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND>
void run(const char* what, int mf_iters, int pt_iters, float* out, long long* clk, const float* src, double ops_per_iter) {
  k<KIND><<<256, 512>>>(out, clk, src, 10, 10, 0.37f);
  hipDeviceSynchronize();
  k<KIND><<<256, 512>>>(out, clk, src, mf_iters, pt_iters, 0.37f);
  hipDeviceSynchronize();
  std::vector<long long> h(256 * 8);
  hipMemcpy(h.data(), clk, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
  double mf = 0, pt = 0;
  for (int b = 0; b < 256; ++b)
    for (int w = 0; w < 8; ++w) (w < 4 ? mf : pt) += h[b * 8 + w];
  mf /= 1024; pt /= 1024;
  printf("%-34s mfma iters %5d partner iters %6d: %6.2f ticks per MFMA, %7.1f ticks per partner iteration (%5.1f per op)\n", what, mf_iters,
         pt_iters, mf_iters ? mf / (mf_iters * 24.0) : 0.0, pt_iters ? pt / pt_iters : 0.0, pt_iters ? pt / pt_iters / ops_per_iter : 0.0);
}

int main() {
  float* out; long long* clk; float* src;
  hipMalloc(&out, sizeof(float) * 256 * 512);
  hipMalloc(&clk, sizeof(long long) * 256 * 8);
  hipMalloc(&src, sizeof(float) * 256 * 65536);
  hipMemset(src, 0, sizeof(float) * 256 * 65536);
  // each partner alone, the MFMA wave alone, then both (partner iterations sized to last about as long as the MFMA loop)
  run<0>("mfma alone (partner sleeps)", 2000, 100, out, clk, src, 1);
  run<1>("16 v_fma_f32 alone", 0, 20000, out, clk, src, 16);
  run<1>("16 v_fma_f32 + mfma", 2000, 20000, out, clk, src, 16);
  run<2>("16 v_pk_fma_f32 alone", 0, 20000, out, clk, src, 16);
  run<2>("16 v_pk_fma_f32 + mfma", 2000, 20000, out, clk, src, 16);
  run<7>("16 v_max_f32 alone", 0, 20000, out, clk, src, 16);
  run<7>("16 v_max_f32 + mfma", 2000, 20000, out, clk, src, 16);
  run<3>("f16x3 split of 8 alone", 0, 10000, out, clk, src, 1);
  run<3>("f16x3 split of 8 + mfma", 2000, 10000, out, clk, src, 1);
  run<4>("4 ds_write_b128 alone", 0, 10000, out, clk, src, 4);
  run<4>("4 ds_write_b128 + mfma", 2000, 10000, out, clk, src, 4);
  run<5>("4 ds_read_b128 alone", 0, 10000, out, clk, src, 4);
  run<5>("4 ds_read_b128 + mfma", 2000, 10000, out, clk, src, 4);
  run<6>("4 global_load_b128 alone", 0, 3000, out, clk, src, 4);
  run<6>("4 global_load_b128 + mfma", 2000, 3000, out, clk, src, 4);
  return 0;
}
