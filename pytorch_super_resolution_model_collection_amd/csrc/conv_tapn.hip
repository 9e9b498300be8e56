// Output convolutions with a handful of output channels (the 64 -> 3 reconstruction convs that end EDSR, VDSR,
// SRResNet-style generators; reference base_networks.py ConvBlock with output_size = num_channels), stride 1,
// KH*KW*OC <= 32.
//
// An MFMA tile over (pixels x output channels) would be >= 80 % padding, and the plain-VALU kernel
// (k_conv_direct) is latency-bound on interleaved LDS / scalar-cache reads (1.2 TB/s at 128 x 64 x 128 x 128).
// This kernel makes the TAPS part of the matrix N dimension instead:
//
//   z[p][t*OC + oc] = sum_c x[p][c] * w[t][c][oc]          one GEMM [halo pixels x IC] x [IC x (T*OC <= 32)]
//   y[q][oc]        = sum_t z[q + offset(t)][t*OC + oc]     a 9-term shifted sum per output pixel
//
//   * the A operand (16 pixels x 32 channels, lane = (pixel j, group kq)) is read STRAIGHT from global memory
//     into fragment registers, every input pixel of the tile exactly once, all loads of a wave (<= 6 pixel
//     groups) in flight together, then split to bf16 planes in registers.  The contraction index is permuted
//     (fragment slot (kq, e) = channel (e/4)*16 + kq*4 + e%4, same permutation on the filter side) so that
//     the 4 lanes of a pixel read 64 CONTIGUOUS bytes per load instruction;
//   * the B operand (the whole filter, 64 x 27 values) is built once per wave from the fp32 filter and stays
//     in registers as bf16 planes;
//   * always the exact 3-way split (bf16x6: rms error 3.5e-7, below an fp32-accumulating FMA chain) -- the
//     kernel is bound by the activation read, the MFMAs are free;
//   * z goes through LDS ([pixel][36] floats: conflict-free fragment writes), then one thread per output pixel
//     sums its taps and runs the scalar epilogue (bias / activation / residual / pixel shuffle).
//
// HBM traffic = the activation once (+ halo overlap served by L2) + OC/IC of it written.
#include "srk_common.h"
#include "conv_problem.h"
#include "conv_tile.h"
#include "bf16_frag.h"
#include <stdlib.h>

namespace srk {

constexpr int TAPN_ZS = 36;    // z row stride in floats: 4*ZS = 16 (mod 64) spreads the 4 row groups of a fragment over the banks
constexpr int TAPN_MAXI = 6;   // pixel groups (16 pixels each) per wave: halo <= 4 * 6 * 16 = 384 pixels
constexpr int TAPN_NP = 3;     // bf16 planes

template <int KS, int OCT>  // KS = IC / 32, OCT = output channels
__global__ __launch_bounds__(256, 3) void k_conv_tapn(MfmaConvParams P) {
  extern __shared__ __attribute__((aligned(16))) float zbuf[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kq = lane >> 4;
  // XCD-aware block order: hardware deals consecutive block ids round-robin over the 8 XCDs (each with its own
  // L2); give every XCD a contiguous range of tiles so that neighbouring tiles share their halo rows in ONE L2.
  int b;
  {
    const int nb = gridDim.x, per = nb >> 3, rem = nb & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    b = xcd < rem ? xcd * (per + 1) + idx : rem * (per + 1) + (xcd - rem) * per + idx;
  }
  const int txi = b % P.tiles_x;
  b /= P.tiles_x;
  const int tyi = b % P.tiles_y;
  const int n = b / P.tiles_y;
  const int r0 = tyi * P.TH, c0 = txi * P.TW;
  const int T = P.KHv * P.KWv;
  const int NN = T * OCT;
  const int npix = P.HH * P.HW;
  const int MT = (npix + 15) >> 4;

  if (T > 0) {
    // activation fragments of this wave's pixel groups: all global loads issued before anything else
    f32x4 raw[TAPN_MAXI][KS * 2];
    const float* __restrict__ inb = P.in + (size_t)n * P.IH * P.IW * P.IC + kq * 4;
    const int iyb = r0 + P.iy0, ixb = c0 + P.ix0;
#pragma unroll
    for (int i = 0; i < TAPN_MAXI; ++i) {
      const int hp = (wave + 4 * i) * 16 + j;
      const int hy = hp / P.HW, hx = hp - hy * P.HW;
      const int iy = iyb + hy, ix = ixb + hx;
      const bool ok = hp < npix && (unsigned)iy < (unsigned)P.IH && (unsigned)ix < (unsigned)P.IW;
      const float* src = inb + ((size_t)iy * P.IW + ix) * P.IC;
#pragma unroll
      for (int q = 0; q < KS * 2; ++q) {
        raw[i][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (ok) raw[i][q] = *reinterpret_cast<const f32x4*>(src + q * 16);
      }
    }
    // filter fragments: column nn = t*OC + oc of the [IC x 32] matrix, rows in the permuted channel order above
    uint4 bf[KS][2][TAPN_NP];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int nn = nt * 16 + j;
      const bool on = nn < NN;
      const int t = on ? nn / OCT : 0, oc = nn - t * OCT;
      const int u = t / P.KWv, v = t - u * P.KWv;
      const int tapw = (P.wh0 + P.wdh * u) * P.KW_full + (P.ww0 + P.wdw * v);
      const float* __restrict__ w = P.wp + ((size_t)tapw * P.IC + kq * 4) * OCT + oc;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = on ? w[(s * 32 + (e >> 2) * 16 + (e & 3)) * OCT] : 0.f;
        split8n<TAPN_NP>(f, bf[s][nt]);
      }
    }
    const bool two = NN > 16;
#pragma unroll
    for (int i = 0; i < TAPN_MAXI; ++i) {
      const int mt = wave + 4 * i;
      if (mt < MT) {  // wave-uniform
        f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
        uint4 a[KS][TAPN_NP];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          float f[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f[e] = raw[i][2 * s][e];
            f[4 + e] = raw[i][2 * s + 1][e];
          }
          split8n<TAPN_NP>(f, a[s]);
        }
        // smallest products first (as in k_conv_bfd)
#define SRK_TAPN_PASS(pa, pb)                                      \
  _Pragma("unroll") for (int s = 0; s < KS; ++s) {                 \
    acc[0] = mfma16(a[s][pa], bf[s][0][pb], acc[0]);               \
    if (two) acc[1] = mfma16(a[s][pa], bf[s][1][pb], acc[1]);      \
  }
        SRK_TAPN_PASS(2, 0)
        SRK_TAPN_PASS(0, 2)
        SRK_TAPN_PASS(1, 1)
        SRK_TAPN_PASS(1, 0)
        SRK_TAPN_PASS(0, 1)
        SRK_TAPN_PASS(0, 0)
#undef SRK_TAPN_PASS
        // C/D layout: col = lane & 15 (nn), row = (lane >> 4) * 4 + reg (pixel of the group)
        float* zr = zbuf + (size_t)(mt * 16 + kq * 4) * TAPN_ZS + j;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          zr[e * TAPN_ZS] = acc[0][e];
          if (two) zr[e * TAPN_ZS + 16] = acc[1][e];
        }
      }
    }
  }
  __syncthreads();
  if (tid >= P.TH * P.TW) return;
  const int r = tid / P.TW, c = tid - r * P.TW;
  const int pr = r0 + r, pc = c0 + c;
  if (pr >= P.PH || pc >= P.PW) return;
  float sum[OCT];
#pragma unroll
  for (int o = 0; o < OCT; ++o) sum[o] = 0.f;
  const float* zp = zbuf + (size_t)(r * P.HW + c) * TAPN_ZS;
  int t = 0;
  for (int u = 0; u < P.KHv; ++u)
    for (int v = 0; v < P.KWv; ++v, ++t) {
      const float* zt = zp + (u * P.HW + v) * TAPN_ZS + t * OCT;
#pragma unroll
      for (int o = 0; o < OCT; ++o) sum[o] += zt[o];
    }
  GatherConv g{};
  g.OH = P.OH; g.OW = P.OW; g.OC = P.OC;
  const int oy = P.oy0 + pr * P.os, ox = P.ox0 + pc * P.os;
#pragma unroll
  for (int o = 0; o < OCT; ++o) epi_store(P.ep, g, sum[o], n, oy, ox, o, P.out);
}

bool conv_tapn_gather_supported(const GatherConv& g, const float* in, const float* mask_y) {
  static const int off = getenv("SRK_TAPN") ? !atoi(getenv("SRK_TAPN")) : 0;  // SRK_TAPN=0: use k_conv_direct
  if (off) return false;
  if (g.OC < 1 || g.OC > 3 || g.KH * g.KW * g.OC > 32) return false;
  if (g.IC != 32 && g.IC != 64) return false;
  if (!g.trans && g.stride != 1) return false;
  if (mask_y || g.in_nchw || g.in_ps_r > 1) return false;
  if ((uintptr_t)in % 16 != 0) return false;
  return true;
}

template <int KS, int OCT>
static int tapn_launch(MfmaConvParams P, hipStream_t s) {
  TilePick best{};
  const int kh = P.KHv > 0 ? P.KHv : 1, kw = P.KWv > 0 ? P.KWv : 1;
  if (!pick_tile(256, P.PH, P.PW, 1, kh, kw, TAPN_ZS, 4 * TAPN_MAXI * 16 * TAPN_ZS, best)) {
    set_error("conv_tapn: no tile fits");
    return SRK_ERR_UNSUPPORTED;
  }
  P.TH = best.TH; P.TW = best.TW; P.tiles_y = best.tiles_y; P.tiles_x = best.tiles_x; P.HH = best.HH; P.HW = best.HW;
  const size_t lds = (size_t)(((best.HH * best.HW + 15) & ~15)) * TAPN_ZS * sizeof(float);
  dim3 grid((unsigned)((size_t)P.tiles_x * P.tiles_y * P.N));
  hipLaunchKernelGGL((k_conv_tapn<KS, OCT>), grid, dim3(256), lds, s, P);
  return check_launch("conv_tapn");
}

int conv_tapn_gather(const GatherConv& g, const float* in, const float* wp, float* out, const Epi& ep, hipStream_t s) {
  return for_each_phase(g, in, wp, out, ep, nullptr, 0.f, [&](const MfmaConvParams& P) {
    if (P.is != 1) {
      set_error("conv_tapn: strided gather");
      return (int)SRK_ERR_UNSUPPORTED;
    }
    const int key = (g.IC / 32) * 10 + g.OC;
    switch (key) {
      case 11: return tapn_launch<1, 1>(P, s);
      case 12: return tapn_launch<1, 2>(P, s);
      case 13: return tapn_launch<1, 3>(P, s);
      case 21: return tapn_launch<2, 1>(P, s);
      case 22: return tapn_launch<2, 2>(P, s);
      default: return tapn_launch<2, 3>(P, s);
    }
  });
}

}  // namespace srk
