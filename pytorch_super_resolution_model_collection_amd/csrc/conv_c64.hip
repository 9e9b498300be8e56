// One 3x3, stride-1, pad-1 convolution 64 -> 64 channels on a SMALL problem, per 8x8 output tile ("c64"): the single-conv
// form of the fused residual-block kernel of conv_res2.hip.
//
// Why: SRGAN's generator (srgan.py:14-46: conv -> BatchNorm -> PReLU -> conv -> BatchNorm -> + skip, 16 blocks) cannot
// fuse its two convolutions -- a BatchNorm needs the statistics of the whole batch in between -- so an adversarial step
// at the reference's batch of 16 runs ~140 launches of exactly this layer (two generator forwards, two data-gradient
// passes), each a latency chain on one 64-pixel tile per CU: 16 - 20 us on the channel-split blocks of conv_bfd.hip
// (halo staged chunk by chunk, K-split reduce through LDS, LDS-staged epilogue, 4 barriers) for 1.4 us of matrix work.
// The fused residual kernel runs TWO such convolutions in ~16 us with a leaner skeleton; this is that skeleton for one:
//   * block = one 8x8 output tile, 8 waves: wave (ow, kgrp) owns output channels [16 ow, 16 ow + 16) and the 32-channel
//     input chunk kgrp; both chunks of the 10x10 input halo are staged at once (planes h, m [, l] in LDS);
//   * filter fragments of the 9 taps come straight from global memory in the prepared MFMA layout, two taps ahead;
//   * transposed product (A = filter, B = pixels): a lane ends up with 4 consecutive channels of one pixel, so the
//     epilogue (bias, residual / gradient fan-in add, running maximum) stores 16-byte vectors from registers;
//   * the two chunk groups swap half of their partial sums through a separate LDS region and each finishes two of the
//     four pixel tiles: two barriers per launch in all.
// Arithmetic: as conv_bfd.hip -- NP = 2 bf16x3 (data gradients), NP = 3 bf16x6 (fp32-faithful forward of small problems),
// F16: f16x3 with the input's running maximum.  BWD: the data gradient (flipped taps, filters of srk_pack_weight_bwd).
#include "srk_common.h"
#include "conv_problem.h"
#include "bf16_frag.h"
#include <stdlib.h>

namespace srk {

constexpr int C64_C = 64;
constexpr int C64_TS = 8;       // output tile side
constexpr int C64_HS = 10;      // halo side
constexpr int C64_S = 104;      // LDS stride of an 8-channel group: >= 100 halo pixels, 8 (mod 16) -- conflict-free reads and
constexpr int C64_PL = 4 * C64_S + 4;   // writes as in conv_res2.hip (lds_goff, lds_pix_p10); one [4 groups] plane

struct C64Params {
  const float* in;     // [N, H, W, 64]
  const uint4* wq;     // prepared filter planes h, m (or the fp16 section)
  const uint4* wql;    // third plane (NP = 3)
  const float* bias;   // may be NULL
  const float* add;    // optional tensor of the output's shape added to it (residual / gradient fan-in)
  float* out;
  int N, H, W, tiles_y, tiles_x;
  const float* x_amax;  // F16
  float* y_amax;        // optional
  const float* wd;      // F16: trailer {2^-kw, 2^kw}
  double* bn_partial;   // optional (forward): row blockIdx.x of [tiles][2 * 64] = {sum, sum of squares} of this tile's output
};

// sum over the 16 lanes of a DPP row (lanes 16 k .. 16 k + 15) of a double, by four row rotations on its two halves; every
// lane gets a total (lane 0 of the row is the one that is used: a fixed order of additions)
template <int CTRL>
__device__ __forceinline__ double c64_rot_d(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double c64_row_sum_d(double v) {
  v += c64_rot_d<0x121>(v);   // row_ror:1
  v += c64_rot_d<0x122>(v);   // row_ror:2
  v += c64_rot_d<0x124>(v);   // row_ror:4
  v += c64_rot_d<0x128>(v);   // row_ror:8
  return v;
}

#define c64_mfma mfma16x<F16>
template <int NP, bool BWD, bool F16 = false>
__global__ __launch_bounds__(512, 2) void k_c64(C64Params R) {
  static_assert(!F16 || NP == 2, "f16x3 is a two-plane arithmetic");
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
  __shared__ float c64_amx[8];
  __shared__ double c64_bn[2][2][C64_C];   // [chunk group][sum | sum of squares][channel] (bn_partial)
  float sx = 1.f, dsc = 1.f;
  if constexpr (F16) {
    const int kx = amax_scale_exp(amax_read(R.x_amax));
    sx = exp2i(kx);
    dsc = exp2i(-kx) * R.wd[0];
  }
  uint4* hal = smem4;                                               // [2 chunks][NP][C64_PL]
  f32x4* redb = reinterpret_cast<f32x4*>(smem4 + 2 * NP * C64_PL);  // [4 ow][4 tiles][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ow = wave & 3, kgrp = wave >> 2;
  const int col = lane & 15, kq = lane >> 4;
  const int j = lds_pix_p10(col);  // this lane's pixel within a 2 x 8 pixel tile
  int b = xcd_tile_index((int)blockIdx.x, (int)gridDim.x);   // (contiguous tile ranges per XCD: the 10 x 10 halos overlap)
  const int txi = b % R.tiles_x;
  b /= R.tiles_x;
  const int tyi = b % R.tiles_y;
  const int n = b / R.tiles_y;
  const int r0 = tyi * C64_TS, c0 = txi * C64_TS;
  const size_t img = (size_t)n * R.H * R.W * C64_C;
  const float* __restrict__ inb = R.in + img;

  const int wlane = kq * 64 + col + ow * 16;
  auto load_b = [&](int t, uint4(&dst)[NP]) {
    const int wt = BWD ? 8 - t : t;  // data gradient: flipped taps (TRANS gather with stride 1)
    const size_t slot = (size_t)(wt * 2 + kgrp);
    const uint4* w = R.wq + slot * 512 + wlane;
    dst[0] = w[0];
    dst[1] = w[256];
    if (NP == 3) dst[NP - 1] = (R.wql + slot * 256)[wlane];
  };
  uint4 bq[3][NP];
  load_b(0, bq[0]);
  load_b(1, bq[1]);

  // ---- input halo -> planes in LDS: item = (pixel, 8-channel group), lane bits [pixel & 3][g >> 1][g & 1][chunk]
  // [pixel >> 2] (conflict-free ds_write_b128, conv_res2.hip); unconditional loads from clamped addresses (a
  // load under a divergent branch is followed by s_waitcnt vmcnt(0) at the join), zeroed by a select for the padding
  {
    constexpr int NIT = (100 * 8 + 511) / 512;
    f32x4 v0[NIT], v1[NIT];
    bool ok[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int item = tid + k * 512;
      const int g8 = ((item >> 1) & 2) | ((item >> 3) & 1) | ((item >> 2) & 4), hp = ((item >> 5) << 2) | (item & 3);
      const int hy = hp / C64_HS, hx = hp - hy * C64_HS;
      const int iy = r0 - 1 + hy, ix = c0 - 1 + hx;
      ok[k] = hp < 100 && (unsigned)iy < (unsigned)R.H && (unsigned)ix < (unsigned)R.W;
      const float* p = inb + (ok[k] ? ((size_t)iy * R.W + ix) * C64_C : 0) + g8 * 8;
      v0[k] = *reinterpret_cast<const f32x4*>(p);
      v1[k] = *reinterpret_cast<const f32x4*>(p + 4);
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int item = tid + k * 512;
      const int g8 = ((item >> 1) & 2) | ((item >> 3) & 1) | ((item >> 2) & 4), hp = ((item >> 5) << 2) | (item & 3);
      if (hp < 100) {
        float f[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          f[e] = ok[k] ? v0[k][e] : 0.f;
          f[4 + e] = ok[k] ? v1[k][e] : 0.f;
        }
        uint4 pl[NP];
        if constexpr (F16) split8h(f, sx, pl); else split8n<NP>(f, pl);
        const int chunk = g8 >> 2, g = g8 & 3;
#pragma unroll
        for (int p = 0; p < NP; ++p) hal[(chunk * NP + p) * C64_PL + lds_goff(g, C64_S) + hp] = pl[p];
      }
    }
  }
  // epilogue operands requested before the taps: they land under the matrix work
  const int ch4 = ow * 16 + kq * 4;
  f32x4 res[2], bias4 = {0.f, 0.f, 0.f, 0.f};
  int ooff[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int m = (kgrp * 2 + q) * 16 + j;
    const int iy = r0 + (m >> 3), ix = c0 + (m & 7);
    const bool okp = iy < R.H && ix < R.W;
    ooff[q] = okp ? (int)(((size_t)iy * R.W + ix) * C64_C) + ch4 : -1;
    res[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (R.add) res[q] = *reinterpret_cast<const f32x4*>(R.add + img + (okp ? ooff[q] : ch4));
  }
  if (R.bias) bias4 = *reinterpret_cast<const f32x4*>(R.bias + ch4);
  const float peeked = amax_peek(R.y_amax, blockIdx.x);
  __syncthreads();

  // ---- 9 taps: 4 pixel tiles x this wave's 16 channels x chunk kgrp
  f32x4 acc[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    int hp[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int m = mt * 16 + j;
      hp[mt] = (m >> 3) * C64_HS + (m & 7) + lds_goff(kq, C64_S);
    }
    constexpr int plane = C64_PL;
    const uint4* hc = hal + kgrp * NP * plane;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (t + 2 < 9) load_b(t + 2, bq[(t + 2) % 3]);
      const int toff = (t / 3) * C64_HS + (t % 3);
      uint4 a[NP][4];
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a[p][mt] = hc[p * plane + hp[mt] + toff];
      const uint4(&bf)[NP] = bq[t % 3];
      if (NP == 3) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = c64_mfma(bf[0], a[NP - 1][mt], acc[mt]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = c64_mfma(bf[NP - 1], a[0][mt], acc[mt]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = c64_mfma(bf[1], a[1][mt], acc[mt]);
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = c64_mfma(bf[0], a[1][mt], acc[mt]);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = c64_mfma(bf[1], a[0][mt], acc[mt]);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = c64_mfma(bf[0], a[0][mt], acc[mt]);
    }
  }
  // ---- the chunk groups swap halves (own LDS region: nobody waits for the halo readers): group 0 finishes pixel tiles
  // 0, 1 and group 1 tiles 2, 3
  f32x4* red = redb + (size_t)(ow * 4) * 64 + lane;
  if (kgrp == 0) {
    red[2 * 64] = acc[2];
    red[3 * 64] = acc[3];
  } else {
    red[0 * 64] = acc[0];
    red[1 * 64] = acc[1];
  }
  __syncthreads();
  float oamax = 0.f;
  double bs[4] = {0.0, 0.0, 0.0, 0.0}, bq2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int mt = kgrp * 2 + q;
    // (element-wise select of VALUES: see conv_res2.hip -- a select between two array elements becomes a dynamic index)
    const f32x4 own_lo = acc[q], own_hi = acc[2 + q];
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = kgrp ? own_hi[e] : own_lo[e];
    v += red[mt * 64];
    if constexpr (F16) v *= dsc;
    v = v + bias4 + res[q];
    if (ooff[q] >= 0) {
      *reinterpret_cast<f32x4*>(R.out + img + ooff[q]) = v;
      if (R.y_amax) oamax = abs_max4(oamax, v);
      if (!BWD && R.bn_partial) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const double d = (double)v[e];
          bs[e] += d;
          bq2[e] += d * d;   // (exact products, as in k_bn_colsum's forward mode)
        }
      }
    }
  }
  // Column sums of the stored tile for the BatchNorm behind this conv (srk_epilogue.bn_partial): the lane's two pixels,
  // the 16 pixel columns of its DPP row, then the two chunk groups (which finish different pixel tiles) through LDS --
  // a fixed order, one row of the partial slab per block, no atomics.
  if (!BWD && R.bn_partial) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bs[e] = c64_row_sum_d(bs[e]);
      bq2[e] = c64_row_sum_d(bq2[e]);
    }
    if (col == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        c64_bn[kgrp][0][ow * 16 + kq * 4 + e] = bs[e];
        c64_bn[kgrp][1][ow * 16 + kq * 4 + e] = bq2[e];
      }
    }
    __syncthreads();
    if (tid < 2 * C64_C) {
      const int which = tid >> 6, c = tid & 63;
      R.bn_partial[(size_t)blockIdx.x * (2 * C64_C) + which * C64_C + c] = c64_bn[0][which][c] + c64_bn[1][which][c];
    }
  }
  if (R.y_amax) amax_commit_block(R.y_amax, oamax, blockIdx.x, c64_amx, 8, peeked);
}
#undef c64_mfma

// The layer this kernel is for, on a problem small enough that per-tile latency -- not matrix throughput -- is what
// the separate kernels spend their time on (the same limit as conv_bfd.hip's small-problem blocks): plain epilogue (bias,
// one added tensor), no activation, no gradient mask, no pixel shuffle, 16-byte aligned tensors.  SRK_C64=0: never.
bool conv_c64_applicable(const GatherConv& g, const Epi& ep, const float* in, const float* out, const float* mask_y) {
  if (env_int("SRK_C64", 1) == 0) return false;
  if (g.IC != C64_C || g.OC != C64_C || g.KH != 3 || g.KW != 3 || g.stride != 1 || g.pad != 1) return false;
  if (g.in_nchw || g.in_ps_r > 1 || g.IH != g.OH || g.IW != g.OW) return false;
  if (mask_y || ep.act != SRK_ACT_NONE || ep.ps_r > 1 || ep.out_relu) return false;
  if ((((uintptr_t)in | (uintptr_t)out | (uintptr_t)ep.bias | (uintptr_t)ep.residual) & 15) != 0) return false;
  if ((long)g.OH * g.OW * C64_C >= (1L << 29)) return false;  // 32-bit element offsets inside an image
  return conv_bfd_small_problem(g);
}

template <int NP, bool BWD, bool F16>
static int c64_launch(const C64Params& R, hipStream_t s) {
  const size_t lds = (size_t)2 * NP * C64_PL * 16 + (size_t)4 * 4 * 64 * 16;
  static LdsLimit lim;
  lim.ensure(reinterpret_cast<const void*>(&k_c64<NP, BWD, F16>), lds);
  note_kernel("k_c64<%d,%d%s>", NP, (int)BWD, F16 ? ",f16" : "");
  note_amax_written(R.y_amax != nullptr);
  hipLaunchKernelGGL((k_c64<NP, BWD, F16>), dim3((unsigned)((size_t)R.N * R.tiles_y * R.tiles_x)), dim3(512), lds, s, R);
  return check_launch("conv_c64");
}

// planes: 2 bf16x3, 3 bf16x6, 4 f16x3 (needs ep.x_amax and a forward packed buffer).  g.trans selects the data gradient.
int conv_c64_gather(const GatherConv& g, const float* in, const float* wp, float* out, const Epi& ep, int planes,
                    hipStream_t s) {
  const size_t elems = (size_t)9 * C64_C * C64_C;
  const char* b1 = reinterpret_cast<const char*>(wp) + bf3_prepared_offset(elems);
  const size_t main_bytes = bf3_main_bytes(C64_C, C64_C, 9);
  C64Params R{};
  R.in = in;
  R.wq = reinterpret_cast<const uint4*>(b1);
  R.wql = reinterpret_cast<const uint4*>(b1 + main_bytes);
  R.bias = ep.bias;
  R.add = ep.residual;
  R.out = out;
  R.N = g.N; R.H = g.OH; R.W = g.OW;
  R.tiles_y = (g.OH + C64_TS - 1) / C64_TS;
  R.tiles_x = (g.OW + C64_TS - 1) / C64_TS;
  R.x_amax = ep.x_amax;
  R.y_amax = ep.y_amax;
  const bool bwd = g.trans != 0;
  R.bn_partial = bwd ? nullptr : ep.bn_partial;
  if (R.bn_partial) note_bn_partial_rows(R.N * R.tiles_y * R.tiles_x);
  if (planes == 4) {
    if (bwd || !ep.x_amax) {
      set_error("conv_c64: f16x3 is a forward arithmetic and needs x_amax");
      return SRK_ERR_BAD_ARG;
    }
    const size_t foff = f16_section_offset(C64_C, C64_C, 9);
    R.wq = reinterpret_cast<const uint4*>(b1 + foff);
    R.wd = reinterpret_cast<const float*>(b1 + foff + main_bytes);
    return c64_launch<2, false, true>(R, s);
  }
  if (planes == 3) return bwd ? c64_launch<3, true, false>(R, s) : c64_launch<3, false, false>(R, s);
  return bwd ? c64_launch<2, true, false>(R, s) : c64_launch<2, false, false>(R, s);
}

}  // namespace srk
