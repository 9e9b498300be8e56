// Per-item bodies of the weight-packing kernels, shared by the per-layer kernels (conv_generic.hip,
// conv_mfma_bf16.hip) and the batched whole-model packer (k_pack_batched).
#pragma once
#include "srk_common.h"

namespace srk {

constexpr int kPackCols = 14;  // int64 columns per row of the whole-model pack table (include/srk.h)
typedef __bf16 pk_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 pk_f16x8 __attribute__((ext_vector_type(8)));

// Sizes of the prepared sections of a packed buffer (host and device): [fp32 layout | 256-aligned: bf16 planes h, m
// (main) | bf16 plane l (main / 2) | forward buffers only, 256-aligned: fp16 planes h, m of w * 2^kw (main) | trailer
// 256 B: {float 2^-kw, float 2^kw}]
__host__ __device__ inline int pk_nb(int OC) { return OC >= 64 ? 64 : ((OC + 15) / 16) * 16; }
__host__ __device__ inline size_t pk_prepared_offset(size_t elems) { return (elems * sizeof(float) + 255) & ~(size_t)255; }
__host__ __device__ inline size_t pk_main_bytes(int IC, int OC, int T) {
  const int ICc = (IC + 31) / 32, OCb = (OC + 63) / 64;
  return (size_t)T * ICc * OCb * 8 * pk_nb(OC) * 16;
}
__host__ __device__ inline size_t pk_f16_offset(int IC, int OC, int T) {
  return (pk_main_bytes(IC, OC, T) / 2 * 3 + 255) & ~(size_t)255;
}
__host__ __device__ inline size_t pk_f16_bytes(int IC, int OC, int T) { return pk_main_bytes(IC, OC, T) + 256; }

// fp16 planes of 8 scaled values (h = RN(x s), m = RN(x s - h))
__device__ __forceinline__ void pk_split8h(const float (&f)[8], float s, uint4& hi, uint4& lo) {
  pk_f16x8 h, m;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float x = f[e] * s;
    const _Float16 hh = (_Float16)x;
    h[e] = hh;
    m[e] = (_Float16)(x - (float)hh);
  }
  hi = __builtin_bit_cast(uint4, h);
  lo = __builtin_bit_cast(uint4, m);
}

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
                                              int ICc, int OCb, int NB, uint4* __restrict__ f16dst = nullptr,
                                              float f16scale = 1.f) {
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
  if (f16dst) {  // fp16 planes of w * 2^kw, same [plane][group][co] slots as the bf16 main layout
    uint4 fh, fm;
    pk_split8h(f, f16scale, fh, fm);
    uint4* fb = f16dst + slot * (size_t)(8 * NB);
    fb[(0 * 4 + g) * NB + col] = fh;
    fb[(1 * 4 + g) * NB + col] = fm;
  }
}

// max|w| of one layer -> trailer {2^-kw, 2^kw} with max|w| * 2^kw in [2^13, 2^14) (kw = 0 for an all-zero filter).
// One 256-thread block per layer.
__device__ __forceinline__ void pack_f16_trailer(const float* __restrict__ w, long elems, float* __restrict__ trailer) {
  __shared__ float sm_amax[4];
  float a = 0.f;
  for (long e = threadIdx.x; e < elems; e += 256) a = fmaxf(a, fabsf(w[e]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a = fmaxf(a, __shfl_xor(a, o, 64));
  if ((threadIdx.x & 63) == 0) sm_amax[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    a = fmaxf(fmaxf(sm_amax[0], sm_amax[1]), fmaxf(sm_amax[2], sm_amax[3]));
    int k = 0;
    if (a != 0.f) k = 140 - (int)((__float_as_uint(a) >> 23) & 0xff);
    k = k < -126 ? -126 : (k > 126 ? 126 : k);
    trailer[0] = __uint_as_float((unsigned)(127 - k) << 23);  // 2^-kw
    trailer[1] = __uint_as_float((unsigned)(127 + k) << 23);  // 2^kw
  }
}

// bf16x3 row-packed layout (gather IC <= 4) [kh][ks][ocb][plane][group][co][8]
__device__ __forceinline__ void pack_bf3_rows_item(long it, const float* __restrict__ w, uint4* __restrict__ dst,
                                                   int Cout, int Cin, int KH, int KW, int transposed, int ps_r, int bwd,
                                                   int IC, int OC, int KS, int OCb, int NB,
                                                   uint4* __restrict__ f16dst = nullptr, float f16scale = 1.f) {
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
  if (f16dst) {  // fp16 planes of w * 2^kw in the same row-packed slots (f16x3 first layers)
    uint4 fh, fm;
    pk_split8h(f, f16scale, fh, fm);
    uint4* fb = f16dst + ((size_t)(kh * KS + ks) * OCb + ocb) * (size_t)(8 * NB);
    fb[(0 * 4 + g) * NB + col] = fh;
    fb[(1 * 4 + g) * NB + col] = fm;
  }
}

}  // namespace srk
