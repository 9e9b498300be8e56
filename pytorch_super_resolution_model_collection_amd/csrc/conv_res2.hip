// Both 3x3 convolutions of a residual block in ONE launch per 8x8 output tile ("res2").
//
//   forward  (base_networks.py:128-150, norm=None):  mid = relu(conv1(x) + b1)        out = conv2(mid) + b2 + x
//   backward (data gradient of the same block):      dmid = conv2^T(dy) * (mid > 0)   dx  = conv1^T(dmid) + dy
//
// Why: on a strong-scaled shard (EDSR x4, 16 patches per GPU: one 64-pixel tile per CU) a body convolution is a
// latency chain -- launch, halo load, 9 taps behind global filter fragments, K-split reduce, store -- of ~11-14 us of
// which the matrix work is 1.4 us, and a block's second conv cannot start before the first one's stores have landed
// and been re-read.  Here the block's tile keeps the intermediate in LDS: the first conv is evaluated on the 10x10
// region the second one needs (1.56x its matrix work, which the tile has time for), split into bf16 planes straight
// from the accumulators, and consumed by the second conv after one barrier.  One launch, one global halo load and one
// dependent store -> load round trip less per block; the intermediate's centre 8x8 is still written out (the weight
// gradients and the backward mask need it), but nothing waits for that store.
//
// Same arithmetic as conv_bfd.hip (fp32 operands split into bf16 planes, v_mfma_f32_16x16x32_bf16, fp32
// accumulation; NP = 3: bf16x6 fp32-faithful forward, NP = 2: bf16x3 data gradient), same prepared filter layout
// [tap][chunk][plane][group][co][8 x bf16] read as MFMA operands straight from global memory, same transposed product
// (a lane ends up with 4 consecutive channels of one pixel).  Block = 8 waves: wave (ow, kgrp) owns output channels
// [16 ow, 16 ow + 16) and the 32-channel input chunk kgrp; the two chunk groups swap half of their partial
// accumulators through LDS and each finishes half of the pixels.
#include "srk_common.h"
#include "conv_problem.h"
#include "bf16_frag.h"
#include <stdlib.h>

namespace srk {

constexpr int R2_C = 64;       // channels in = mid = out
constexpr int R2_TS = 8;       // output tile width (and the height of the standard tile)
constexpr int R2_H1 = 12;      // input halo width (tile + 2 + 2)
constexpr int R2_MW = 10;      // mid region width (tile + 1 + 1)
// Tile = 8 x 8 outputs: 12 x 12 input halo, 10 x 10 intermediate in 7 MFMA pixel tiles, 4 output pixel tiles.
// (Round 4 experiment, removed: 4-row half tiles so that the 16-patch EDSR shard -- one 8 x 8 tile per CU -- has two
// resident blocks per CU.  tools/res2_prof.py: forward 16.2 -> 16.5 us, backward 15.2 -> 14.8 us, step 1.199 -> 1.212 ms:
// the two blocks stretch each other's tap phases by what they hide of each other's barriers.)
constexpr int R2_NPIX1 = R2_H1 * R2_H1;   // 144 input halo pixels
constexpr int R2_MT1 = 7;                 // 16-pixel tiles of the intermediate (r2_mid_rc)
constexpr int R2_MT1A = 4;                // ... finished by chunk group 0 (the rest by group 1)
constexpr int R2_MT2 = 4;                 // 16-pixel tiles of the output: rows 2 mt, 2 mt + 1
constexpr int R2_MT2A = 2;                // ... finished by each chunk group

// ---- LDS layout of the operand planes, in 16-byte slots (8 channels of one pixel): [chunk][plane][group g][pixel slot]
// with the group offset lds_goff(g) = g * S + (g >> 1) * 4 and S = 8 (mod 16).
//  * ds_read_b128 serves the lane sets {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (and + 32) in one clock each when
//    their 16 slots differ mod 16; a set mixes 8 lanes of group kq with 8 lanes of group kq + 1.  Round 3 read 16
//    row-major pixels of a 10-wide region from rows of pitch 12 (conv1) / 10 (conv2) with S = 0 (mod 16): the pixels of
//    the tile's second row alias the first row's (16 == 0, 17 == 1), EVERY operand read took two clocks per set
//    (SQ_LDS_BANK_CONFLICT = 35 % of the kernel's cycles at batch 128), and the LDS, not the matrix pipe, paced the taps.
//    Now a pixel tile is 2 rows x 8 columns, the lane -> pixel maps lds_pix_p12 / lds_pix_p10 put slots on the lanes {0-3, 12-15}
//    whose residues, shifted by S = 8, are the complement of the other 8 lanes' residues, for every tap (a tap shifts all
//    16 alike): one clock per set.  The 10 x 10 intermediate is five 2 x 8 tiles (columns 0-7), the 8 x 2 strip of columns
//    8-9 above row 8 (rows of pitch 12 give a column strip only 8 residues: this one tile reads in two clocks), and the
//    2 x 2 corner evaluated as the 2 x 8 tile at columns 2-9 of rows 8-9 (columns 2-7 recomputed and dropped).
//  * ds_write_b128 serves 8 adjacent lanes per clock when their slots differ mod 8: the halo staging puts 4 adjacent
//    pixels x the groups {g, g + 2} on 8 lanes (lds_goff(g + 2) - lds_goff(g) = 4 mod 8); 16 lanes still read 4 x 128
//    contiguous bytes of global memory.
constexpr int R2_S1 = 152, R2_S2 = 104;           // >= 144 / >= 100, 8 mod 16
constexpr int R2_PL1 = 4 * R2_S1 + 4, R2_PL2 = 4 * R2_S2 + 4;   // one [4 groups] plane
// pixel (r, c) of the 10 x 10 intermediate that lane column `col` holds in tile mt (0..6); false: a dropped duplicate
__device__ __forceinline__ bool r2_mid_rc(int mt, int col, int& r, int& c) {
  const int i = lds_pix_p12(col);
  if (mt == 5) {   // strip: rows 0-3 on the lanes {0-3, 12-15}, rows 4-7 on {4-11}
    const bool a = col < 4 || col >= 12;
    const int k = a ? (col < 4 ? col : col - 8) : col - 4;
    r = (a ? 0 : 4) + (k >> 1);
    c = 8 + (k & 1);
    return true;
  }
  if (mt >= 6) {
    r = 8 + (i >> 3);
    c = 2 + (i & 7);
    return mt == 6 && (i & 7) >= 6;
  }
  r = 2 * mt + (i >> 3);
  c = i & 7;
  return true;
}

struct Res2Params {
  const float* in;    // [N, H, W, 64]: x (forward) / dy (backward); also the residual added to `out`
  const uint4* wq1;   // first conv: prepared filter planes h, m
  const uint4* wq1l;  //             third plane (NP = 3)
  const uint4* wq2;   // second conv
  const uint4* wq2l;
  const float* bias1;  // forward only (may be NULL)
  const float* bias2;
  const float* gate;   // backward: saved forward `mid`; dmid is zeroed where it is <= 0
  float* mid;          // [N, H, W, 64] centre store of the intermediate
  float* out;
  int N, H, W, tiles_y, tiles_x;
  int dbg;
  // f16x3 forward (SRK_ALGO_MFMA_F16X3): wq1 / wq2 point at the fp16 planes, wd1 / wd2 at their trailers {2^-kw, 2^kw}
  const float* x_amax;
  float* y_amax;   // optional (any arithmetic): running max of |out|
  const float* wd1;
  const float* wd2;
  long long* prof;   // experiments build only (srk_debug_res2_prof): 16 clock64() stamps per block, thread 0
};

#ifdef SRK_EXPERIMENTS
#define R2_PROF(i) do { if (R.prof && threadIdx.x == 0) R.prof[(size_t)blockIdx.x * 16 + (i)] = clock64(); } while (0)
static long long* g_res2_prof = nullptr;
#else
#define R2_PROF(i) do { } while (0)
#endif

template <int NP>
__device__ __forceinline__ void r2_split4(const f32x4& v, uint2 (&pl)[NP]) {
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  bf16x4 h, m, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const __bf16 hh = (__bf16)v[e];
    const float r1 = v[e] - (float)hh;
    const __bf16 mm = (__bf16)r1;
    h[e] = hh;
    m[e] = mm;
    if (NP == 3) l[e] = (__bf16)(r1 - (float)mm);
  }
  pl[0] = __builtin_bit_cast(uint2, h);
  pl[1] = __builtin_bit_cast(uint2, m);
  if (NP == 3) pl[NP - 1] = __builtin_bit_cast(uint2, l);
}

__device__ __forceinline__ void r2_split4h(const f32x4& v, float s, uint2 (&pl)[2]) {
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  f16x4 h, m;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float x = v[e] * s;
    const _Float16 hh = (_Float16)x;
    h[e] = hh;
    m[e] = (_Float16)(x - (float)hh);
  }
  pl[0] = __builtin_bit_cast(uint2, h);
  pl[1] = __builtin_bit_cast(uint2, m);
}

#define mfma16 mfma16x<F16>
#define SRK_R2_PASSES(ACC, A, BF, MT)                                                                   \
  {                                                                                                     \
    if (NP == 3) {                                                                                      \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) ACC[mt] = mfma16(BF[0], A[NP - 1][mt], ACC[mt]); \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) ACC[mt] = mfma16(BF[NP - 1], A[0][mt], ACC[mt]); \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) ACC[mt] = mfma16(BF[1], A[1][mt], ACC[mt]);      \
    }                                                                                                   \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) ACC[mt] = mfma16(BF[0], A[1][mt], ACC[mt]);        \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) ACC[mt] = mfma16(BF[1], A[0][mt], ACC[mt]);        \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) ACC[mt] = mfma16(BF[0], A[0][mt], ACC[mt]);        \
  }

// PF: the NEXT tap's operand reads are issued in front of the current tap's MFMAs (two fragment sets: ~180 VGPRs, one
// block per CU) -- for grids of at most one block per CU (the 16-patch EDSR shard), where nothing else hides the LDS
// latency at the start of every tap.
template <int NP, bool BWD, bool F16 = false, bool PF = false>
__global__ __launch_bounds__(512, 2) void k_res2(Res2Params R) {
  static_assert(!F16 || (NP == 2 && !BWD), "f16x3 is the two-plane forward arithmetic");
  constexpr int TH = R2_TS, MT1A = R2_MT1A, MT2 = R2_MT2, MT2A = R2_MT2A;
  constexpr int S1 = R2_S1, S2 = R2_S2, PL1 = R2_PL1, PL2 = R2_PL2;
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
  __shared__ float r2_amx[8];
  // f16x3: input scale 2^kx from the tensor's running maximum; the intermediate gets its own scale from the tile's maximum
  float sx = 1.f, dsc1 = 1.f;
  int kx = 0;
  if constexpr (F16) {
    kx = amax_scale_exp(amax_read(R.x_amax));
    sx = exp2i(kx);
    dsc1 = exp2i(-kx) * R.wd1[0];
  }
  uint4* hal1 = smem4;                 // [2 chunks][NP][PL1]: 12 x 12 input halo
  uint4* hal2 = smem4 + 2 * NP * PL1;  // [2][NP][PL2]: 10 x 10 intermediate
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ow = wave & 3, kgrp = wave >> 2;
  const int col = lane & 15, kq = lane >> 4;
  // XCD-aware block order (round 6): hardware deals consecutive block ids round-robin over the 8 XCDs, each with its own L2;
  // a tile's 12 x 12 halo overlaps its eight neighbours' (2.25x the tile), and with b = blockIdx.x every neighbour sat on
  // another XCD -- the overlap was fetched once per L2 (counter traffic 1.34x forward / 1.45x backward, the only large
  // kernel that re-read).  Now every XCD walks a contiguous range of tiles (whole patches of the 16-patch shard).
  int b = xcd_tile_index((int)blockIdx.x, (int)gridDim.x);
  const int txi = b % R.tiles_x;
  b /= R.tiles_x;
  const int tyi = b % R.tiles_y;
  const int n = b / R.tiles_y;
  const int r0 = tyi * TH, c0 = txi * R2_TS;
  const size_t img = (size_t)n * R.H * R.W * R2_C;
  const float* __restrict__ inb = R.in + img;

  // ---- filter fragments: position seq = conv * 9 + tap of the 18-tap sequence (the prefetch runs across the two convs)
  const int wlane = kq * 64 + col + ow * 16;
  auto load_b = [&](int seq, uint4(&dst)[NP]) {
    const int cv = seq >= 9 ? 1 : 0, t = seq - 9 * cv;
    const int wt = BWD ? 8 - t : t;  // data gradient: the taps run flipped (conv_tile.h, TRANS gather with stride 1)
    const size_t slot = (size_t)(wt * 2 + kgrp);
    const uint4* w = (cv ? R.wq2 : R.wq1) + slot * 512 + wlane;
    dst[0] = w[0];
    dst[1] = w[256];
    if (NP == 3) dst[NP - 1] = ((cv ? R.wq2l : R.wq1l) + slot * 256)[wlane];
  };
  R2_PROF(0);
  uint4 bq[3][NP];
  load_b(0, bq[0]);
  load_b(1, bq[1]);

  // ---- input halo -> planes in LDS.  item = (pixel, 8-channel group of the 64); lane bits: [pixel & 3][g >> 1][g & 1]
  // [chunk][pixel >> 2] (see lds_goff); every global load is issued before the first conversion
  constexpr int NIT = (R2_NPIX1 * 8 + 511) / 512;
  f32x4 v0[NIT], v1[NIT];
  if (!(SRK_KDBG(R.dbg) & 1)) {
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int item = tid + k * 512;
      const int g8 = ((item >> 1) & 2) | ((item >> 3) & 1) | ((item >> 2) & 4), hp = ((item >> 5) << 2) | (item & 3);
      const int hy = hp / R2_H1, hx = hp - hy * R2_H1;
      const int iy = r0 - 2 + hy, ix = c0 - 2 + hx;
      v0[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
      v1[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (hp < R2_NPIX1 && (unsigned)iy < (unsigned)R.H && (unsigned)ix < (unsigned)R.W) {
        const float* p = inb + ((size_t)iy * R.W + ix) * R2_C + g8 * 8;
        v0[k] = *reinterpret_cast<const f32x4*>(p);
        v1[k] = *reinterpret_cast<const f32x4*>(p + 4);
      }
    }
  }
  // ---- per-lane geometry of the exchange after the first conv (group 0 finishes mid tiles 0-3, group 1 tiles 4-6),
  // computed HERE, behind the halo loads and in front of their first use: after the first conv's taps this index
  // arithmetic was part of the ~2.3 k clocks the block's first wave then waits at the barrier (tools/res2_prof.py).
  // (The gate / bias loads stay behind the taps: 16 + 8 more live registers across them are 132 - 138 VGPRs, one
  //  block per CU.)
  const int ch4 = ow * 16 + kq * 4;
  int moff[MT1A];  // element offset of the tile's pixel in the image (-1: outside the image or the region)
  int mpos[MT1A];  // its slot in the 10 x 10 intermediate (-1: a dropped duplicate)
  bool mcen[MT1A];
#pragma unroll
  for (int q = 0; q < MT1A; ++q) {
    const int mt = kgrp ? MT1A + q : q;
    int r, c;
    const bool valid = r2_mid_rc(mt, col, r, c);
    mpos[q] = valid ? r * R2_MW + c : -1;
    const int iy = r0 - 1 + r, ix = c0 - 1 + c;
    const bool inimg = valid && (unsigned)iy < (unsigned)R.H && (unsigned)ix < (unsigned)R.W;
    moff[q] = inimg ? (int)(((size_t)iy * R.W + ix) * R2_C) + ch4 : -1;
    mcen[q] = inimg && r >= 1 && r <= TH && c >= 1 && c <= R2_TS;
  }
  if (!(SRK_KDBG(R.dbg) & 1)) {
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int item = tid + k * 512;
      const int g8 = ((item >> 1) & 2) | ((item >> 3) & 1) | ((item >> 2) & 4), hp = ((item >> 5) << 2) | (item & 3);
      if (hp < R2_NPIX1) {
        float f[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          f[e] = v0[k][e];
          f[4 + e] = v1[k][e];
        }
        uint4 pl[NP];
        if constexpr (F16) split8h(f, sx, pl); else split8n<NP>(f, pl);
        const int chunk = g8 >> 2, g = g8 & 3;
#pragma unroll
        for (int p = 0; p < NP; ++p) hal1[(chunk * NP + p) * PL1 + lds_goff(g, S1) + hp] = pl[p];
      }
    }
  }
  R2_PROF(1);
  __syncthreads();
  R2_PROF(2);

  // ---- first conv on the 10x10 mid region: 7 pixel tiles x this wave's 16 channels x chunk kgrp
  f32x4 acc1[R2_MT1];
#pragma unroll
  for (int mt = 0; mt < R2_MT1; ++mt) acc1[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    int hpA[R2_MT1];
#pragma unroll
    for (int mt = 0; mt < R2_MT1; ++mt) {
      int r, c;
      r2_mid_rc(mt, col, r, c);
      hpA[mt] = r * R2_H1 + c + lds_goff(kq, S1);
    }
    constexpr int plane1 = PL1;
    const uint4* h1c = hal1 + kgrp * NP * plane1;
    if constexpr (PF) {
      uint4 a2[2][NP][R2_MT1];
      auto lda = [&](int t, uint4 (&a)[NP][R2_MT1]) {
        const int toff = (t / 3) * R2_H1 + (t % 3);
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
          for (int mt = 0; mt < R2_MT1; ++mt) a[p][mt] = h1c[p * plane1 + hpA[mt] + toff];
      };
      lda(0, a2[0]);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        load_b(t + 2, bq[(t + 2) % 3]);
        if (t + 1 < 9) lda(t + 1, a2[(t + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        SRK_R2_PASSES(acc1, a2[t & 1], bq[t % 3], R2_MT1)
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      load_b(t + 2, bq[(t + 2) % 3]);
      // (a scheduling fence per tap: without one the unrolled taps' fragment reads are hoisted over each other -- 248 VGPRs,
      //  one block per CU; until round 4 a run-time ablation branch around the tap body had that effect by accident)
      __builtin_amdgcn_sched_barrier(0);
      if (!(SRK_KDBG(R.dbg) & 4)) {
        const int toff = (t / 3) * R2_H1 + (t % 3);
        uint4 a[NP][R2_MT1];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
          for (int mt = 0; mt < R2_MT1; ++mt) a[p][mt] = h1c[p * plane1 + hpA[mt] + toff];
        SRK_R2_PASSES(acc1, a, bq[t % 3], R2_MT1)
      }
    }
    }
  }

  R2_PROF(3);
  // ---- the chunk groups swap halves
  f32x4 gt[MT1A];  // backward: forward mid values of this wave's tiles (issued before the barriers)
#pragma unroll
  for (int q = 0; q < MT1A; ++q) {
    // (unconditional, from a clamped address: a load under a divergent branch is followed by s_waitcnt vmcnt(0) at the
    //  join, which would serialise the latencies of these loads; a pixel outside the image is zeroed below anyway)
    gt[q] = (f32x4){1.f, 1.f, 1.f, 1.f};
    if constexpr (BWD) gt[q] = *reinterpret_cast<const f32x4*>(R.gate + img + (moff[q] >= 0 ? moff[q] : ch4));
  }
  // biases: requested here, complete before the epilogues that use them (a first use inside those conditional store
  // sequences put an s_waitcnt vmcnt(0) -- a wait for the previous store's acknowledgement -- in front of every store)
  f32x4 b1 = {0.f, 0.f, 0.f, 0.f}, b2 = {0.f, 0.f, 0.f, 0.f};
  if (!BWD && R.bias1) b1 = *reinterpret_cast<const f32x4*>(R.bias1 + ch4);
  if (!BWD && R.bias2) b2 = *reinterpret_cast<const f32x4*>(R.bias2 + ch4);
  __syncthreads();  // every wave is done with the input halo
  R2_PROF(10);
  f32x4* red = reinterpret_cast<f32x4*>(smem4) + (size_t)(ow * R2_MT1) * 64 + lane;  // [4 ow][7 tiles][64 lanes]
  if (kgrp == 0) {
#pragma unroll
    for (int mt = MT1A; mt < R2_MT1; ++mt) red[mt * 64] = acc1[mt];
  } else {
#pragma unroll
    for (int mt = 0; mt < MT1A; ++mt) red[mt * 64] = acc1[mt];
  }
  __syncthreads();
  R2_PROF(11);
  float smid = 1.f, dsc2 = 1.f;
  {
    asm volatile("" ::"v"(b1), "v"(b2));
    if constexpr (BWD) {
#pragma unroll
      for (int q = 0; q < MT1A; ++q) asm volatile("" ::"v"(gt[q]));
    }
    const int ch2 = ow >> 1, g2 = (ow & 1) * 2 + (kq >> 1);
    f32x4 vv[MT1A];
    float lmax = 0.f;
#pragma unroll
    for (int mt = 0; mt < R2_MT1; ++mt) {
      if ((mt < MT1A) != (kgrp == 0)) continue;
      const int q = mt < MT1A ? mt : mt - MT1A;
      f32x4 v = acc1[mt] + red[mt * 64];
      if (!BWD) {
        if constexpr (F16) v *= dsc1;
        v += b1;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gt[q][e] > 0.f ? v[e] : 0.f;
      }
      if (moff[q] < 0) v = (f32x4){0.f, 0.f, 0.f, 0.f};  // outside the image: the second conv's zero padding
      if (mcen[q]) *reinterpret_cast<f32x4*>(R.mid + img + moff[q]) = v;
      vv[q] = v;
      if constexpr (F16) lmax = abs_max4(lmax, v);
    }
    R2_PROF(12);
    if constexpr (F16) {
      // the intermediate's scale: maximum over the whole 10 x 10 x 64 tile (the second conv mixes all of it)
      lmax = wave_max(lmax);
      if (lane == 0) r2_amx[wave] = lmax;
      __syncthreads();
      float m8 = r2_amx[0];
#pragma unroll
      for (int i = 1; i < 8; ++i) m8 = fmaxf(m8, r2_amx[i]);
      const int km = amax_scale_exp(m8);
      smid = exp2i(km);
      dsc2 = exp2i(-km) * R.wd2[0];
    }
    R2_PROF(13);
#pragma unroll
    for (int mt = 0; mt < R2_MT1; ++mt) {
      if ((mt < MT1A) != (kgrp == 0)) continue;
      const int q = mt < MT1A ? mt : mt - MT1A;
      if (mpos[q] >= 0) {
        uint2 pl[NP];
        if constexpr (F16) r2_split4h(vv[q], smid, pl); else r2_split4<NP>(vv[q], pl);
#pragma unroll
        for (int p = 0; p < NP; ++p)
          reinterpret_cast<uint2*>(hal2 + (ch2 * NP + p) * PL2 + lds_goff(g2, S2) + mpos[q])[kq & 1] = pl[p];
      }
    }
  }
  // residual (= the centre of the input) for the tiles this wave finishes: group 0 the first half, group 1 the second
  f32x4 res[MT2A];
  int ooff[MT2A];
#pragma unroll
  for (int q = 0; q < MT2A; ++q) {
    const int m = (kgrp * MT2A + q) * 16 + lds_pix_p10(col);
    const int iy = r0 + (m >> 3), ix = c0 + (m & 7);
    const bool ok = iy < R.H && ix < R.W;
    ooff[q] = ok ? (int)(((size_t)iy * R.W + ix) * R2_C) + ch4 : -1;
    res[q] = *reinterpret_cast<const f32x4*>(inb + (ok ? ooff[q] : ch4));  // (unconditional: see gt; unused when !ok)
  }
  R2_PROF(4);
  __syncthreads();  // mid planes complete
  R2_PROF(5);

  // ---- second conv on the TH x 8 centre: MT2 pixel tiles
  f32x4 acc2[MT2];
#pragma unroll
  for (int mt = 0; mt < MT2; ++mt) acc2[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    int hpB[MT2];
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt) {
      const int m = mt * 16 + lds_pix_p10(col);
      hpB[mt] = (m >> 3) * R2_MW + (m & 7) + lds_goff(kq, S2);
    }
    constexpr int plane2 = PL2;
    const uint4* h2c = hal2 + kgrp * NP * plane2;
    if constexpr (PF) {
      uint4 a2[2][NP][MT2];
      auto lda = [&](int t, uint4 (&a)[NP][MT2]) {
        const int toff = (t / 3) * R2_MW + (t % 3);
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
          for (int mt = 0; mt < MT2; ++mt) a[p][mt] = h2c[p * plane2 + hpB[mt] + toff];
      };
      lda(0, a2[0]);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (t + 2 < 9) load_b(9 + t + 2, bq[(t + 2) % 3]);
        if (t + 1 < 9) lda(t + 1, a2[(t + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        SRK_R2_PASSES(acc2, a2[t & 1], bq[t % 3], MT2)
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (t + 2 < 9) load_b(9 + t + 2, bq[(t + 2) % 3]);
      __builtin_amdgcn_sched_barrier(0);
      if (!(SRK_KDBG(R.dbg) & 4)) {
        const int toff = (t / 3) * R2_MW + (t % 3);
        uint4 a[NP][MT2];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
          for (int mt = 0; mt < MT2; ++mt) a[p][mt] = h2c[p * plane2 + hpB[mt] + toff];
        SRK_R2_PASSES(acc2, a, bq[t % 3], MT2)
      }
    }
    }
  }
  R2_PROF(6);
  // swap halves again (the region of the input halo is free: nothing reads it after the barrier above)
  f32x4* red2 = reinterpret_cast<f32x4*>(smem4) + (size_t)(ow * MT2) * 64 + lane;  // [4 ow][MT2 tiles][64 lanes]
  const float peeked = amax_peek(R.y_amax, blockIdx.x);  // (early: see amax_commit)
  if (kgrp == 0) {
#pragma unroll
    for (int mt = MT2A; mt < MT2; ++mt) red2[mt * 64] = acc2[mt];
  } else {
#pragma unroll
    for (int mt = 0; mt < MT2A; ++mt) red2[mt * 64] = acc2[mt];
  }
  __syncthreads();
  R2_PROF(7);
#pragma unroll
  for (int q = 0; q < MT2A; ++q) asm volatile("" ::"v"(res[q]));  // (landed during the second conv: no wait between the stores below)
  float oamax = 0.f;
#pragma unroll
  for (int q = 0; q < MT2A; ++q) {
    const int mt = kgrp * MT2A + q;
    if (ooff[q] >= 0) {
      // (element-wise select of VALUES: a select between two elements of the register array is folded into one load with
      //  a dynamic index, which sends the whole accumulator array to scratch memory)
      const f32x4 own_lo = acc2[q], own_hi = acc2[MT2A + q];
      f32x4 own;
#pragma unroll
      for (int e = 0; e < 4; ++e) own[e] = kgrp ? own_hi[e] : own_lo[e];
      f32x4 v = own + red2[mt * 64];
      if constexpr (F16) v *= dsc2;
      v = v + b2 + res[q];
      *reinterpret_cast<f32x4*>(R.out + img + ooff[q]) = v;
      if (R.y_amax) oamax = abs_max4(oamax, v);
    }
  }
  R2_PROF(8);
  if (R.y_amax) amax_commit_block(R.y_amax, oamax, blockIdx.x, r2_amx, 8, peeked);
  R2_PROF(9);
}
#undef SRK_R2_PASSES
#undef mfma16

static int r2_dbg() {
  return SRK_EXP_INT("SRK_DBG", 0);
}

// Small problems only: the tile does 1.56x the first conv's matrix work, which is free while a CU has a few tiles and
// mostly waits, and a loss once the separate kernels run many tiles per CU at their matrix rate.  Measured on the EDSR
// x4 train step (tools/shard_step.py, fused vs separate): 16 patches 1.36 vs 1.63 ms, 32: 2.27 vs 2.56, 64: 4.04 vs
// 4.57, 128 (8 tiles per CU): 7.57 vs 7.08 -> the limit sits at 5 tiles per CU.
bool conv_res2_supported(int N, int H, int W, int C) {
  // (round 3: with the f16x3 forward the fused block also won at 8 tiles per CU: EDSR batch 128 6.46 -> 6.33 ms.  Round 6: from
  //  the size at which the wave-specialised family takes a 64 -> 64 layer -- 512 pixels per CU, conv_bfw_applicable -- the block's
  //  two convs run faster as two launches of the ring kernel's canvas variant, which adds the skip / the gradient fan-in itself:
  //  EDSR batch 128 5.56 -> 5.40 ms same-box; at batch 64 those kernels do not apply and the fused block stays 16 % ahead)
  const int max_tiles = env_int("SRK_RES2_MAX_TILES", 8 * kNumCU - 1);
  if (C != R2_C || N < 1 || H < 1 || W < 1) return false;
  if ((long)H * W * R2_C >= (1L << 29)) return false;  // 32-bit element offsets inside an image
  const long tiles = (long)N * ((H + R2_TS - 1) / R2_TS) * ((W + R2_TS - 1) / R2_TS);
  return tiles <= max_tiles;
}

template <int NP, bool BWD, bool F16 = false>
static int r2_launch(const Res2Params& R, hipStream_t s) {
  const size_t lds = (size_t)2 * NP * (R2_PL1 + R2_PL2) * 16;
  const size_t grid = (size_t)R.N * R.tiles_y * R.tiles_x;
  if constexpr (NP == 2) {
    // at most one block per CU: the variant that reads the next tap's operands ahead (SRK_RES2_PF=0: off)
    if (grid <= (size_t)kNumCU && env_int("SRK_RES2_PF", 1) != 0) {
      static LdsLimit limp;
      limp.ensure(reinterpret_cast<const void*>(&k_res2<NP, BWD, F16, true>), lds);
      note_kernel("k_res2<%d,%d%s,pf>", NP, (int)BWD, F16 ? ",f16" : "");
      hipLaunchKernelGGL((k_res2<NP, BWD, F16, true>), dim3((unsigned)grid), dim3(512), lds, s, R);
      return check_launch("conv_res2");
    }
  }
  static LdsLimit lim;
  lim.ensure(reinterpret_cast<const void*>(&k_res2<NP, BWD, F16>), lds);
  note_kernel("k_res2<%d,%d%s>", NP, (int)BWD, F16 ? ",f16" : "");
  hipLaunchKernelGGL((k_res2<NP, BWD, F16>), dim3((unsigned)grid), dim3(512), lds, s, R);
  return check_launch("conv_res2");
}

// `wp1` / `wp2`: packed filter buffers of srk_pack_weight_fwd (forward) / srk_pack_weight_bwd (backward) of the conv
// that runs first / second in this direction.  planes = 3: bf16x6, 2: bf16x3, 4: f16x3 (forward only, x_amax required).
int conv_res2(const float* in, const float* wp1, const float* wp2, const float* bias1, const float* bias2,
              const float* gate, float* mid, float* out, int N, int H, int W, int planes, bool bwd, hipStream_t s,
              const float* x_amax, float* y_amax) {
  const size_t elems = (size_t)9 * R2_C * R2_C;
  const char* b1 = reinterpret_cast<const char*>(wp1) + bf3_prepared_offset(elems);
  const char* b2 = reinterpret_cast<const char*>(wp2) + bf3_prepared_offset(elems);
  const size_t main_bytes = bf3_main_bytes(R2_C, R2_C, 9);
  Res2Params R{};
  R.in = in;
  R.wq1 = reinterpret_cast<const uint4*>(b1);
  R.wq1l = reinterpret_cast<const uint4*>(b1 + main_bytes);
  R.wq2 = reinterpret_cast<const uint4*>(b2);
  R.wq2l = reinterpret_cast<const uint4*>(b2 + main_bytes);
  R.bias1 = bias1;
  R.bias2 = bias2;
  R.gate = gate;
  R.mid = mid;
  R.out = out;
  R.N = N; R.H = H; R.W = W;
  R.tiles_y = (H + R2_TS - 1) / R2_TS;
  R.tiles_x = (W + R2_TS - 1) / R2_TS;
  R.dbg = r2_dbg();
  R.x_amax = x_amax;
  R.y_amax = y_amax;
#ifdef SRK_EXPERIMENTS
  R.prof = g_res2_prof;
#endif
  if (planes == 4) {
    if (bwd || !x_amax) {
      set_error("conv_res2: f16x3 is a forward arithmetic and needs x_amax");
      return SRK_ERR_BAD_ARG;
    }
    const size_t foff = f16_section_offset(R2_C, R2_C, 9);
    R.wq1 = reinterpret_cast<const uint4*>(b1 + foff);
    R.wq2 = reinterpret_cast<const uint4*>(b2 + foff);
    R.wd1 = reinterpret_cast<const float*>(b1 + foff + main_bytes);
    R.wd2 = reinterpret_cast<const float*>(b2 + foff + main_bytes);
    return r2_launch<2, false, true>(R, s);
  }
  if (bwd) return planes == 3 ? r2_launch<3, true>(R, s) : r2_launch<2, true>(R, s);
  return planes == 3 ? r2_launch<3, false>(R, s) : r2_launch<2, false>(R, s);
}

}  // namespace srk

#ifdef SRK_EXPERIMENTS
// experiments build only: device buffer of 16 int64 per block that the next fused-block launches stamp (R2_PROF); NULL: off
extern "C" void srk_debug_res2_prof(void* p) { srk::g_res2_prof = static_cast<long long*>(p); }
#endif
