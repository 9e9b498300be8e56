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

// ---- f16x3: x * 2^k split into two fp16 planes (h = RN(x s), m = RN(x s - h)); see SRK_ALGO_MFMA_F16X3 -------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// Two values per call, four v_fma_mix: h = f16(x s) and m = f16(x s - h) straight from the fp32 operands (the mixed-precision
// fma rounds once to fp16; x s is exact -- s is a power of two -- and x s - h has at most 13 significant bits, so both
// results equal the cvt / sub / cvt sequence bit for bit: tools/micro/split_test.hip, 8.4 M words at four scales).  The
// compiler's own choice for the scalar source was 23 instructions per 8 values, half of them packed fp32 operations that are
// slow beside an MFMA stream (tools/micro/coissue.hip); this is 16.
__device__ __forceinline__ void split2h(float x0, float x1, float s, unsigned& h, unsigned& m) {
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(s));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(m) : "v"(x0), "v"(s), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(m) : "v"(x1), "v"(s), "v"(h));
}
__device__ __forceinline__ void split8h(const float (&f)[8], float s, uint4 (&pl)[2]) {
  split2h(f[0], f[1], s, pl[0].x, pl[1].x);
  split2h(f[2], f[3], s, pl[0].y, pl[1].y);
  split2h(f[4], f[5], s, pl[0].z, pl[1].z);
  split2h(f[6], f[7], s, pl[0].w, pl[1].w);
}

__device__ __forceinline__ f32x4 mfma16h(const uint4& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <bool F16>
__device__ __forceinline__ f32x4 mfma16x(const uint4& a, const uint4& b, f32x4 c) {
  if constexpr (F16) return mfma16h(a, b, c); else return mfma16(a, b, c);
}

// Running maximum of |values| (SRK_AMAX_SLOTS = 16 slots, one per 64-byte line; non-negative floats order like their
// bit patterns).
// amax_read: wave-uniform maximum over the slots.  amax_scale_exp: k with amax * 2^k in [2^13, 2^14) (0 for amax = 0;
// inf / nan inputs give k = -114: the products then overflow to inf / nan like the fp32 ones would).
__device__ __forceinline__ float amax_read(const float* __restrict__ slots) {
  float a = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) a = fmaxf(a, slots[i * 16]);
  return a;
}
__device__ __forceinline__ int amax_scale_exp(float amax) {
  const int E = (int)((__float_as_uint(amax) >> 23) & 0xff);
  if (amax == 0.f) return 0;
  return 140 - E;  // amax in [2^(E-127), 2^(E-126))  ->  amax * 2^(140-E) in [2^13, 2^14)
}
__device__ __forceinline__ float exp2i(int k) {  // 2^k for -126 <= k <= 127 (clamped)
  k = k < -126 ? -126 : (k > 127 ? 127 : k);
  return __uint_as_float((unsigned)(k + 127) << 23);
}
// Maximum over the 64 lanes, the same value in every lane.  Rotations inside the rows of 16 lanes as DPP operands of
// v_max (no LDS traffic), then the four row results through v_readlane.  (Until round 4 this was six dependent
// __shfl_xor = ds_bpermute_b32 round trips, ~400 clocks in front of every running-maximum commit and inside the fused
// residual block's intermediate scale, tools/res2_prof.py.)
// PRECONDITION: all 64 lanes active (full EXEC) -- v_readlane of an inactive lane returns whatever its register holds.
// Every caller (amax_commit, amax_commit_block, the fused residual block's intermediate scale) runs convergent; a call under
// lane divergence or behind a per-lane early return would feed garbage into the running maximum.
template <int CTRL>
__device__ __forceinline__ float dpp_rot(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_rot<0x121>(v));   // row_ror:1
  v = fmaxf(v, dpp_rot<0x122>(v));   // row_ror:2
  v = fmaxf(v, dpp_rot<0x124>(v));   // row_ror:4
  v = fmaxf(v, dpp_rot<0x128>(v));   // row_ror:8 -> every lane holds its row's maximum
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// Committing a maximum: the slot is read EARLY (amax_peek, before the stores of the epilogue, so that the tail of a
// short-lived block does not wait for a memory round trip that nothing overlaps) and the atomic is issued only when the
// maximum still raises that (possibly stale, never too high) value.  Whole bookkeeping on the c2 first layer
// (15876 blocks): ~10 us of 320 (tools/time_yamax.py).
__device__ __forceinline__ float amax_peek(const float* __restrict__ slots, int slot_hint) {
  // agent-scope atomic load: served by L2, where the atomics land (a plain or nontemporal load may hit a stale line
  // of the CU's vector L1 for the whole kernel, and then every wave issues its atomic)
  return slots ? __hip_atomic_load(slots + (slot_hint & 15) * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
}
// one atomic per wave (all 64 lanes must be active: wave_max)
__device__ __forceinline__ void amax_commit(float* __restrict__ slots, float lane_max, int slot_hint, float peeked) {
  const float m = wave_max(lane_max);
  if ((threadIdx.x & 63) == 0 && m > peeked)
    atomicMax(reinterpret_cast<unsigned*>(slots + (slot_hint & 15) * 16), __float_as_uint(m));
}
// one atomic per BLOCK (kernels whose epilogue may use a barrier; all lanes of every wave active): `sm` = one float per wave
__device__ __forceinline__ void amax_commit_block(float* __restrict__ slots, float lane_max, int slot_hint, float* sm,
                                                  int nwaves, float peeked) {
  const float m = wave_max(lane_max);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float b = sm[0];
    for (int i = 1; i < nwaves; ++i) b = fmaxf(b, sm[i]);
    if (b > peeked) atomicMax(reinterpret_cast<unsigned*>(slots + (slot_hint & 15) * 16), __float_as_uint(b));
  }
}
__device__ __forceinline__ float abs_max4(float cur, const f32x4& v) {
  return fmaxf(fmaxf(cur, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
}

// ---- LDS layout helpers of the 2 x 8 pixel-tile kernels (conv_res2.hip, conv_c64.hip; the scheme is explained at the top
// of conv_res2.hip): offset of 8-channel group g in a [4 groups] plane whose group stride S is 8 (mod 16), and the lane
// column -> tile pixel maps for rows of pitch 12 / 10 slots.
__device__ __forceinline__ constexpr int lds_goff(int g, int S) { return g * S + (g >> 1) * 4; }

// lane column (lane & 15) -> pixel index within a 2 x 8 tile.  conv1 (row pitch 12): lanes {0-3, 12-15} hold the pixels
// {0-5, 8, 9}; conv2 (row pitch 10): {0-4, 8-10}.
__device__ __forceinline__ int lds_pix_p12(int col) {
  return col < 4 ? col : col < 6 ? col + 2 : col < 12 ? col + 4 : col < 14 ? col - 8 : col - 6;
}
__device__ __forceinline__ int lds_pix_p10(int col) {
  return col < 4 ? col : col < 7 ? col + 1 : col < 12 ? col + 4 : col == 12 ? 4 : col - 5;
}

}  // namespace srk
