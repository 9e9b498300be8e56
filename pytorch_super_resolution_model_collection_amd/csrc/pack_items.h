// Per-item bodies of the weight-packing kernels, shared by the per-layer kernels (conv_generic.hip,
// conv_mfma_bf16.hip) and the batched whole-model packer (k_pack_batched).
#pragma once
#include "srk_common.h"

namespace srk {

typedef __bf16 pk_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void pk_split8(const float (&f)[8], uint4& hi, uint4& lo) {
  pk_bf16x8 h, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 hh = (__bf16)f[e];
    h[e] = hh;
    l[e] = (__bf16)(f[e] - (float)hh);
  }
  hi = __builtin_bit_cast(uint4, h);
  lo = __builtin_bit_cast(uint4, l);
}

// exact 3-way split x = h + m + l (h, m as in pk_split8; l = bf16(x - h - m) is exact)
__device__ __forceinline__ void pk_split8x3(const float (&f)[8], uint4& hi, uint4& mid, uint4& lo) {
  pk_bf16x8 h, m, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 hh = (__bf16)f[e];
    const float r1 = f[e] - (float)hh;
    const __bf16 mm = (__bf16)r1;
    h[e] = hh;
    m[e] = mm;
    l[e] = (__bf16)(r1 - (float)mm);
  }
  hi = __builtin_bit_cast(uint4, h);
  mid = __builtin_bit_cast(uint4, m);
  lo = __builtin_bit_cast(uint4, l);
}

__device__ __forceinline__ size_t pk_src(int ci, int co, int kh, int kw, int Cout, int Cin, int KH, int KW, int transposed) {
  return transposed ? ((((size_t)ci * Cout + co) * KH + kh) * KW + kw) : ((((size_t)co * Cin + ci) * KH + kh) * KW + kw);
}

// fp32 layouts: fwd wp[kh][kw][ci][co'] (co' = pixel-shuffle-permuted), bwd wp[kh][kw][co][ci]; e = packed index
__device__ __forceinline__ void pack_f32_item(int e, const float* __restrict__ w, float* __restrict__ wp, int Cout,
                                              int Cin, int KH, int KW, int transposed, int ps_r, int bwd) {
  int ci, co_p, tap;
  if (!bwd) {
    co_p = e % Cout;
    ci = (e / Cout) % Cin;
    tap = e / (Cout * Cin);
  } else {
    ci = e % Cin;
    co_p = (e / Cin) % Cout;
    tap = e / (Cout * Cin);
  }
  const int kh = tap / KW, kw = tap % KW;
  int co = co_p;
  if (ps_r > 1) {  // packed order (i, j, c) -> torch order c*r*r + i*r + j
    const int C = Cout / (ps_r * ps_r);
    const int q = co_p / C, c = co_p % C;
    co = c * ps_r * ps_r + q;
  }
  wp[e] = w[pk_src(ci, co, kh, kw, Cout, Cin, KH, KW, transposed)];
}

// bf16 main layout [tap][chunk][ocb][plane h|m][group][co][8], followed by the third plane (bf16x6
// kernels) [tap][chunk][ocb][group][co][8]
__device__ __forceinline__ void pack_bf3_item(long it, const float* __restrict__ w, uint4* __restrict__ dst, int Cout,
                                              int Cin, int KH, int KW, int transposed, int ps_r, int bwd, int IC, int OC,
                                              int ICc, int OCb, int NB) {
  const int col = (int)(it % NB);
  long r = it / NB;
  const int g = (int)(r % 4);
  r /= 4;
  const int ocb = (int)(r % OCb);
  r /= OCb;
  const int cc = (int)(r % ICc);
  const int tap = (int)(r / ICc);
  const int kh = tap / KW, kw = tap - kh * KW;
  const int oc = ocb * 64 + col;
  float f[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ic = cc * 32 + g * 8 + e;
    float v = 0.f;
    if (ic < IC && oc < OC) {
      int ci, co;
      if (!bwd) {
        ci = ic;
        co = oc;
        if (ps_r > 1) {
          const int C = Cout / (ps_r * ps_r);
          const int q = oc / C, c = oc - q * C;
          co = c * ps_r * ps_r + q;
        }
      } else {
        ci = oc;
        co = ic;
        if (ps_r > 1) {  // data gradient of a fused conv + pixel shuffle: K runs over dy's packed (i, j, c) channels
          const int C = Cout / (ps_r * ps_r);
          const int q = ic / C, c = ic - q * C;
          co = c * ps_r * ps_r + q;
        }
      }
      v = w[pk_src(ci, co, kh, kw, Cout, Cin, KH, KW, transposed)];
    }
    f[e] = v;
  }
  uint4 hi, mid, lo;
  pk_split8x3(f, hi, mid, lo);
  const size_t slot = (size_t)(tap * ICc + cc) * OCb + ocb;
  uint4* blk = dst + slot * (size_t)(8 * NB);
  blk[(0 * 4 + g) * NB + col] = hi;
  blk[(1 * 4 + g) * NB + col] = mid;
  uint4* third = dst + (size_t)KH * KW * ICc * OCb * (size_t)(8 * NB);
  third[slot * (size_t)(4 * NB) + g * NB + col] = lo;
}

// bf16x3 row-packed layout (gather IC <= 4) [kh][ks][ocb][plane][group][co][8]
__device__ __forceinline__ void pack_bf3_rows_item(long it, const float* __restrict__ w, uint4* __restrict__ dst,
                                                   int Cout, int Cin, int KH, int KW, int transposed, int ps_r, int bwd,
                                                   int IC, int OC, int KS, int OCb, int NB) {
  const int col = (int)(it % NB);
  long r = it / NB;
  const int g = (int)(r % 4);
  r /= 4;
  const int ocb = (int)(r % OCb);
  r /= OCb;
  const int ks = (int)(r % KS);
  const int kh = (int)(r / KS);
  const int oc = ocb * 64 + col;
  float f[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int kw = ks * 8 + 2 * g + (e >> 2), ic = e & 3;
    float v = 0.f;
    if (kw < KW && ic < IC && oc < OC) {
      int ci = ic, co = oc;
      if (bwd) {
        ci = oc;
        co = ic;
      } else if (ps_r > 1) {
        const int C = Cout / (ps_r * ps_r);
        const int q = oc / C, c = oc - q * C;
        co = c * ps_r * ps_r + q;
      }
      v = w[pk_src(ci, co, kh, kw, Cout, Cin, KH, KW, transposed)];
    }
    f[e] = v;
  }
  uint4 hi, lo;
  pk_split8(f, hi, lo);
  uint4* blk = dst + ((size_t)(kh * KS + ks) * OCb + ocb) * (size_t)(8 * NB);
  blk[(0 * 4 + g) * NB + col] = hi;
  blk[(1 * 4 + g) * NB + col] = lo;
}

}  // namespace srk
