// bf16 fragment helpers shared by the bf16x3 / bf16x6 convolution kernels: exact 2- or 3-way split of 8 fp32
// values into bf16 planes (x = h + m [+ l]) and the 16x16x32 bf16 MFMA on packed 8-element fragments.
#pragma once
#include <hip/hip_runtime.h>

namespace srk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NP>
__device__ __forceinline__ void split8n(const float (&f)[8], uint4 (&pl)[NP]) {
  bf16x8 h, m, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 hh = (__bf16)f[e];
    const float r1 = f[e] - (float)hh;
    const __bf16 mm = (__bf16)r1;
    h[e] = hh;
    m[e] = mm;
    if (NP == 3) l[e] = (__bf16)(r1 - (float)mm);
  }
  pl[0] = __builtin_bit_cast(uint4, h);
  pl[1] = __builtin_bit_cast(uint4, m);
  if (NP == 3) pl[NP - 1] = __builtin_bit_cast(uint4, l);
}

__device__ __forceinline__ f32x4 mfma16(const uint4& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0,
                                                 0);
}

}  // namespace srk
