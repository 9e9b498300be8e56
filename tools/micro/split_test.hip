#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split8h(const float (&f)[8], float s, uint4 (&pl)[2]) {
  f16x8 h, m;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float x = f[e] * s;
    const _Float16 hh = (_Float16)x;
    h[e] = hh;
    m[e] = (_Float16)(x - (float)hh);
  }
  pl[0] = __builtin_bit_cast(uint4, h);
  pl[1] = __builtin_bit_cast(uint4, m);
}
__device__ __forceinline__ void split2_mix(float x0, float x1, float s, unsigned& h, unsigned& m) {
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(s));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(m) : "v"(x0), "v"(s), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(m) : "v"(x1), "v"(s), "v"(h));
}
__device__ __forceinline__ void split8h_mix(const float (&f)[8], float s, uint4 (&pl)[2]) {
  split2_mix(f[0], f[1], s, pl[0].x, pl[1].x);
  split2_mix(f[2], f[3], s, pl[0].y, pl[1].y);
  split2_mix(f[4], f[5], s, pl[0].z, pl[1].z);
  split2_mix(f[6], f[7], s, pl[0].w, pl[1].w);
}
__global__ void k(const f32x4* in, uint4* out, uint4* out2, float s) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  f32x4 a = in[t * 2], b = in[t * 2 + 1];
  float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  uint4 p[2], q[2];
  split8h(f, s, p);
  split8h_mix(f, s, q);
  out[t * 2] = p[0]; out[t * 2 + 1] = p[1];
  out2[t * 2] = q[0]; out2[t * 2 + 1] = q[1];
}
int main() {
  const int n = 1 << 20;
  std::vector<float> h(n * 8);
  srand(1);
  for (size_t i = 0; i < h.size(); ++i) {
    const int e = rand() % 40 - 30;
    h[i] = (float)((rand() / (double)RAND_MAX * 2 - 1) * pow(2.0, e));
    if (i % 97 == 0) h[i] = 0.f;
    if (i % 1013 == 0) h[i] = 1e-30f;
  }
  float* din; uint4 *o1, *o2;
  hipMalloc(&din, h.size() * 4); hipMalloc(&o1, n * 32); hipMalloc(&o2, n * 32);
  hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (float s : {1.f, 1024.f, 8192.f * 1024.f, 1.f / 4096}) {
    k<<<n / 256, 256>>>((const f32x4*)din, o1, o2, s);
    std::vector<unsigned> a(n * 8), b(n * 8);
    hipMemcpy(a.data(), o1, n * 32, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), o2, n * 32, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < a.size(); ++i) bad += a[i] != b[i];
    printf("scale %g: %zu of %zu words differ\n", s, bad, a.size());
  }
  return 0;
}
