// Output convolutions with a handful of output channels (the 64 -> 3 reconstruction convs that end EDSR, VDSR,
// SRResNet-style generators, the 9x9 / 5x5 output convs of SRGAN-G and SRCNN; reference base_networks.py ConvBlock with
// output_size = num_channels), stride 1.
//
// An MFMA tile over (pixels x output channels) would be >= 80 % padding, and the plain-VALU kernel
// (k_conv_direct) is latency-bound on interleaved LDS / scalar-cache reads (1.2 TB/s at 128 x 64 x 128 x 128).
// This kernel makes the TAPS part of the matrix N dimension instead:
//
//   z[p][t*OC + oc] = sum_c x[p][c] * w[t][c][oc]          one GEMM [halo pixels x IC] x [IC x (T*OC <= 32)]
//   y[q][oc]        = sum_t z[q + offset(t)][t*OC + oc]     a 9-term shifted sum per output pixel
// (kernels with more than 32/OC taps run the GEMM once per group of 32/OC taps -- 9 passes for 9x9 x 3 channels --
//  over the SAME activation registers, each pass adding its taps into the output pixel's running sums)
//
//   * the A operand (16 pixels x 32 channels, lane = (pixel j, group kq)) is read STRAIGHT from global memory
//     into fragment registers, every input pixel of the tile exactly once, all loads of a wave (<= 6 pixel
//     groups) in flight together, then split to bf16 planes in registers.  The contraction index is permuted
//     (fragment slot (kq, e) = channel (e/4)*16 + kq*4 + e%4, same permutation on the filter side) so that
//     the 4 lanes of a pixel read 64 CONTIGUOUS bytes per load instruction;
//   * the B operand (the whole filter, 64 x 27 values) is built once per wave from the fp32 filter and stays
//     in registers as bf16 planes;
//   * the exact 3-way split (bf16x6: rms error 3.5e-7, below an fp32-accumulating FMA chain) in every class for <= 32
//     columns; kernels with several tap groups run bf16x3 in the bf16x3 class (inference, gradients) -- the
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
#include <type_traits>

namespace srk {

constexpr int TAPN_ZS = 36;    // z row stride in floats: 4*ZS = 16 (mod 64) spreads the 4 row groups of a fragment over the banks
constexpr int TAPN_MAXI = 6;   // pixel groups (16 pixels each) per wave: halo <= 4 * 6 * 16 = 384 pixels

// KS = IC / 32, OCT = output channels, MULTI = more than one tap group, NP = bf16 planes (3: bf16x6; 2: bf16x3, only for
// MULTI kernels in the bf16x3 class -- there the GEMM runs once per tap group over a halo 2-3x the tile and the MFMAs
// are no longer free)
// MASK: activation-gradient prologue of a data gradient (srk_bwd_mask: x <- x * (mask_y > 0 ? 1 : slope) while loading;
// the data gradient of SRGAN-D's first layer, 3 -> 64 3x3 + LeakyReLU, srgan.py:51, is a 64 -> 3 TRANS gather with a mask and
// ran k_conv_direct<3> at 100 us)
template <int KS, int OCT, bool MULTI, int NP, bool MASK = false>
__global__ __launch_bounds__(256, MULTI ? 2 : 3) void k_conv_tapn(MfmaConvParams P) {
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
  constexpr int TPG = 32 / OCT;             // taps per group: one pass of the [pixels x IC] x [IC x 32] GEMM
  const int NG = MULTI ? (T + TPG - 1) / TPG : 1;  // passes (1 for 3x3; 3 for 5x5; 9 for 9x9 with 3 output channels)
  const int npix = P.HH * P.HW;
  const int MT = (npix + 15) >> 4;
  const bool live = tid < P.TH * P.TW;
  const int m = live ? tid : 0;
  const int r = m / P.TW, c = m - r * P.TW;
  float sum[OCT];
#pragma unroll
  for (int o = 0; o < OCT; ++o) sum[o] = 0.f;

  if (T > 0) {
    // activation fragments of this wave's pixel groups: all global loads issued before anything else, kept in
    // registers across the tap-group passes
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
    if constexpr (MASK) {
      // second pass, UNCONDITIONAL loads from clamped addresses (a load under the divergent `if (ok)` is followed by an
      // s_waitcnt vmcnt(0) at the join: six serialised round trips -- the first version of this prologue ran at 96 us
      // against 31 us for the same layer without a mask); outside the image raw is zero and stays zero
      const float* __restrict__ mkb = P.mask_y + (size_t)n * P.IH * P.IW * P.IC + kq * 4;
#pragma unroll
      for (int i = 0; i < TAPN_MAXI; ++i) {
        const int hp = (wave + 4 * i) * 16 + j;
        const int hy = hp / P.HW, hx = hp - hy * P.HW;
        const int iy = iyb + hy, ix = ixb + hx;
        const bool ok = hp < npix && (unsigned)iy < (unsigned)P.IH && (unsigned)ix < (unsigned)P.IW;
        const float* msk = mkb + (ok ? ((size_t)iy * P.IW + ix) * P.IC : 0);
        f32x4 m[KS * 2];
#pragma unroll
        for (int q = 0; q < KS * 2; ++q) m[q] = *reinterpret_cast<const f32x4*>(msk + q * 16);
#pragma unroll
        for (int q = 0; q < KS * 2; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) raw[i][q][e] = m[q][e] > 0.f ? raw[i][q][e] : raw[i][q][e] * P.mask_slope;
      }
    }
    for (int tg = 0; tg < NG; ++tg) {
      const int t0 = tg * TPG;
      const int tcount = (T - t0) < TPG ? (T - t0) : TPG;
      const int NN = tcount * OCT;
      // filter fragments of this tap group: column nn = (t - t0)*OC + oc of the [IC x 32] matrix, rows in the
      // permuted channel order above
      uint4 bf[KS][2][NP];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int nn = nt * 16 + j;
        const bool on = nn < NN;
        const int t = t0 + (on ? nn / OCT : 0), oc = on ? nn % OCT : 0;
        const int u = t / P.KWv, v = t - u * P.KWv;
        const int tapw = (P.wh0 + P.wdh * u) * P.KW_full + (P.ww0 + P.wdw * v);
        const float* __restrict__ w = P.wp + ((size_t)tapw * P.IC + kq * 4) * OCT + oc;
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) {
          float f[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = on ? w[(s2 * 32 + (e >> 2) * 16 + (e & 3)) * OCT] : 0.f;
          split8n<NP>(f, bf[s2][nt]);
        }
      }
      const bool two = NN > 16;
      if (tg) __syncthreads();  // the previous group's z fully consumed
#pragma unroll
      for (int i = 0; i < TAPN_MAXI; ++i) {
        const int mt = wave + 4 * i;
        if (mt < MT) {  // wave-uniform
          f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
          uint4 a[KS][NP];
#pragma unroll
          for (int s2 = 0; s2 < KS; ++s2) {
            float f[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              f[e] = raw[i][2 * s2][e];
              f[4 + e] = raw[i][2 * s2 + 1][e];
            }
            split8n<NP>(f, a[s2]);
          }
          // smallest products first (as in k_conv_bfd)
#define SRK_TAPN_PASS(pa, pb)                                        \
  _Pragma("unroll") for (int s2 = 0; s2 < KS; ++s2) {                \
    acc[0] = mfma16(a[s2][pa], bf[s2][0][pb], acc[0]);               \
    if (two) acc[1] = mfma16(a[s2][pa], bf[s2][1][pb], acc[1]);      \
  }
          if (NP == 3) {
            SRK_TAPN_PASS(NP - 1, 0)
            SRK_TAPN_PASS(0, NP - 1)
            SRK_TAPN_PASS(1, 1)
          }
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
      __syncthreads();
      if (live) {  // this thread's output pixel: add the taps of the group
        int u = t0 / P.KWv, v = t0 - u * P.KWv;
        for (int q = 0; q < tcount; ++q) {
          const float* zt = zbuf + (size_t)((r + u) * P.HW + c + v) * TAPN_ZS + q * OCT;
#pragma unroll
          for (int o = 0; o < OCT; ++o) sum[o] += zt[o];
          if (++v == P.KWv) {
            v = 0;
            ++u;
          }
        }
      }
    }
  }
  if (!live) return;
  const int pr = r0 + r, pc = c0 + c;
  if (pr >= P.PH || pc >= P.PW) return;
  GatherConv g{};
  g.OH = P.OH; g.OW = P.OW; g.OC = P.OC;
  const int oy = P.oy0 + pr * P.os, ox = P.ox0 + pc * P.os;
#pragma unroll
  for (int o = 0; o < OCT; ++o) epi_store(P.ep, g, sum[o], n, oy, ox, o, P.out);
}

bool conv_tapn_gather_supported(const GatherConv& g, const float* in, const float* mask_y) {
  const bool off = env_int("SRK_TAPN", 1) == 0;  // SRK_TAPN=0: use k_conv_direct
  if (off) return false;
  if (g.OC < 1 || g.OC > 3 || g.KH * g.KW > 121) return false;  // larger kernels: several 32-column tap groups
  if (g.IC != 32 && g.IC != 64) return false;
  if (!g.trans && g.stride != 1) return false;
  if (g.in_nchw || g.in_ps_r > 1) return false;
  if (mask_y && (g.KH * g.KW * g.OC > 32 || (uintptr_t)mask_y % 16 != 0)) return false;   // mask prologue: single-pass kernel only
  if ((uintptr_t)in % 16 != 0) return false;
  return true;
}

template <int KS, int OCT>
static int tapn_launch(MfmaConvParams P, bool x6, hipStream_t s) {
  TilePick best{};
  const int kh = P.KHv > 0 ? P.KHv : 1, kw = P.KWv > 0 ? P.KWv : 1;
  if (!pick_tile(256, P.PH, P.PW, 1, kh, kw, TAPN_ZS, 4 * TAPN_MAXI * 16 * TAPN_ZS, best)) {
    set_error("conv_tapn: no tile fits");
    return SRK_ERR_UNSUPPORTED;
  }
  P.TH = best.TH; P.TW = best.TW; P.tiles_y = best.tiles_y; P.tiles_x = best.tiles_x; P.HH = best.HH; P.HW = best.HW;
  const size_t lds = (size_t)(((best.HH * best.HW + 15) & ~15)) * TAPN_ZS * sizeof(float);
  dim3 grid((unsigned)((size_t)P.tiles_x * P.tiles_y * P.N));
  if (P.KHv * P.KWv * OCT > 32) {
    if (x6)
      hipLaunchKernelGGL((k_conv_tapn<KS, OCT, true, 3>), grid, dim3(256), lds, s, P);
    else
      hipLaunchKernelGGL((k_conv_tapn<KS, OCT, true, 2>), grid, dim3(256), lds, s, P);
  } else if (P.mask_y) {
    note_kernel("k_conv_tapn<%d,%d,mask>", KS, OCT);
    hipLaunchKernelGGL((k_conv_tapn<KS, OCT, false, 3, true>), grid, dim3(256), lds, s, P);
  } else {
    note_kernel("k_conv_tapn<%d,%d>", KS, OCT);
    hipLaunchKernelGGL((k_conv_tapn<KS, OCT, false, 3>), grid, dim3(256), lds, s, P);
  }
  return check_launch("conv_tapn");
}

int conv_tapn_gather(const GatherConv& g, const float* in, const float* wp, float* out, const Epi& ep, bool x6,
                     hipStream_t s, const float* mask_y, float mask_slope) {
  return for_each_phase(g, in, wp, out, ep, mask_y, mask_slope, [&](const MfmaConvParams& P) {
    if (P.is != 1) {
      set_error("conv_tapn: strided gather");
      return (int)SRK_ERR_UNSUPPORTED;
    }
    const int key = (g.IC / 32) * 10 + g.OC;
    switch (key) {
      case 11: return tapn_launch<1, 1>(P, x6, s);
      case 12: return tapn_launch<1, 2>(P, x6, s);
      case 13: return tapn_launch<1, 3>(P, x6, s);
      case 21: return tapn_launch<2, 1>(P, x6, s);
      case 22: return tapn_launch<2, 2>(P, x6, s);
      default: return tapn_launch<2, 3>(P, x6, s);
    }
  });
}

// ---------------------------------------------------------------------------------------------
// Weight gradient of the same layers (Cout <= 3, KH*KW*Cout <= 32, stride 1, 2*pad <= K-1):
//
//   dW[t][ci][oc] = sum over input pixels p of  x[p][ci] * dyc[p][t*OC + oc],   dyc[p][(t, oc)] = dy[p + pad - t][oc]
//
// one GEMM with M = ci (<= 64), N = t*OC + oc (<= 32), K = pixels.  A K step is 32 consecutive pixels of one image
// row.  The MFMA wants 8 consecutive K values of one M row per lane; with lane (j, kq) loading the float4
// x[pixel kq*8 + e][channels 4j .. 4j+3] for e < 8 (fully coalesced 256-byte pixel rows) the lane ends up holding
// exactly that for FOUR rows -- channel 4j + i is row j of M tile i -- so the big operand needs no transposition
// through LDS at all (M rows are merely labelled in a permuted order).  The small operand dyc is gathered from dy
// (12 bytes per pixel, L1-resident).  Waves walk the K steps of the whole batch with a grid stride, keep the full
// 64 x 32 accumulator in registers (8 tiles), and every block writes ONE partial slab at the end (4 waves summed
// through LDS); the deterministic slab reduction and bias finish are the shared k_wgrad_reduce.  The bias gradient
// partials come for free: the centre tap's dyc column IS dy.
// bf16x3 arithmetic (as the other weight-gradient kernels of the bf16 class).
// ---------------------------------------------------------------------------------------------
int conv_wgrad_reduce_launch(const float* ws, float* dw, int G, int Cout, int Cin, int KH, int KW, int transposed,
                             float beta, const float* bias_partial, float* db, int bias_cout, int out_ps_r,
                             hipStream_t s);  // conv_wgrad_mfma.hip

struct WgTapnParams {
  const float* x;
  const float* dy;
  float* ws;            // [G][T][Cin][Cout]
  float* bias_partial;  // [G][Cout] or NULL
  int N, H, W, Cin, OH, OW, pad, KH, KW;
  int S;      // 32-pixel K steps per image row
  int units;  // N * H * S
};

constexpr int WGT_RS = 33;  // LDS row stride of the per-wave partial [64][32]

template <int OCT>
__global__ __launch_bounds__(256, 3) void k_wgrad_tapn(WgTapnParams P) {
  __shared__ float red[4][64 * WGT_RS];
  __shared__ float bred[4][2][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kq = lane >> 4;
  const int T = P.KH * P.KW, NN = T * OCT;
  // this lane's two dyc columns
  bool on[2], centre[2];
  int tu[2], tv[2], toc[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int nn = nt * 16 + j;
    on[nt] = nn < NN;
    const int t = on[nt] ? nn / OCT : 0;
    toc[nt] = nn - t * OCT;
    tu[nt] = t / P.KW;
    tv[nt] = t - tu[nt] * P.KW;
    centre[nt] = on[nt] && tu[nt] == P.pad && tv[nt] == P.pad;
  }
  const bool two = NN > 16;
  const bool ch_on = 4 * j < P.Cin;
  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[i][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum[2] = {0.f, 0.f};
  const int nw = gridDim.x * 4;
  for (int unit = blockIdx.x * 4 + wave; unit < P.units; unit += nw) {
    const int rr = unit / P.S, sg = unit - rr * P.S;
    const int n = rr / P.H, r = rr - n * P.H;
    const int cb = sg * 32 + kq * 8;  // first pixel column of this lane's 8
    // dyc fragments (gather; zero outside dy)
    float braw[2][8];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int qy = r + P.pad - tu[nt];
      const bool oky = on[nt] && (unsigned)qy < (unsigned)P.OH && (nt == 0 || two);
      const float* __restrict__ src = P.dy + ((size_t)(n * P.OH + qy) * P.OW) * OCT + toc[nt];
      const int qx0 = cb + P.pad - tv[nt];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int qx = qx0 + e;
        braw[nt][e] = (oky && (unsigned)qx < (unsigned)P.OW) ? src[(size_t)qx * OCT] : 0.f;
      }
    }
    // x fragments: float4 = channels 4j..4j+3 of pixel cb + e
    // (unconditional loads from clamped addresses, zeroed afterwards: under the divergent branch each load was followed
    //  by an s_waitcnt vmcnt(0) -- eight serialised round trips per K step)
    f32x4 araw[8];
    const float* __restrict__ xs = P.x + ((size_t)(n * P.H + r) * P.W) * P.Cin + (ch_on ? 4 * j : 0);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int cx = cb + e < P.W ? cb + e : P.W - 1;
      araw[e] = *reinterpret_cast<const f32x4*>(xs + (size_t)cx * P.Cin);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (!(ch_on && cb + e < P.W)) araw[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint4 bfr[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      if (centre[nt]) {
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) t += braw[nt][e];
        bsum[nt] += t;
      }
      split8n<2>(braw[nt], bfr[nt]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = araw[e][i];
      uint4 a[2];
      split8n<2>(f, a);
      acc[i][0] = mfma16(a[1], bfr[0][0], acc[i][0]);
      acc[i][0] = mfma16(a[0], bfr[0][1], acc[i][0]);
      acc[i][0] = mfma16(a[0], bfr[0][0], acc[i][0]);
      if (two) {
        acc[i][1] = mfma16(a[1], bfr[1][0], acc[i][1]);
        acc[i][1] = mfma16(a[0], bfr[1][1], acc[i][1]);
        acc[i][1] = mfma16(a[0], bfr[1][0], acc[i][1]);
      }
    }
  }
  // block partial: C/D layout col = lane & 15 (nn), row = (lane >> 4) * 4 + reg; row m of tile i is channel 4m + i
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wave][(4 * (kq * 4 + e) + i) * WGT_RS + nt * 16 + j] = acc[i][nt][e];
  bred[wave][0][lane] = bsum[0];
  bred[wave][1][lane] = bsum[1];
  __syncthreads();
  float* slab = P.ws + (size_t)blockIdx.x * T * P.Cin * OCT;
  for (int idx = tid; idx < 64 * 32; idx += 256) {
    const int ci = idx >> 5, nn = idx & 31;
    if (ci < P.Cin && nn < NN) {
      const float v = (red[0][ci * WGT_RS + nn] + red[1][ci * WGT_RS + nn]) + (red[2][ci * WGT_RS + nn] + red[3][ci * WGT_RS + nn]);
      const int t = nn / OCT, oc = nn - t * OCT;
      slab[((size_t)t * P.Cin + ci) * OCT + oc] = v;
    }
  }
  if (P.bias_partial && tid < OCT) {
    const int nn = (P.pad * P.KW + P.pad) * OCT + tid;  // centre tap column of channel tid
    const int nt = nn >> 4, jj = nn & 15;
    float t = 0.f;
    for (int w = 0; w < 4; ++w)
      for (int q = 0; q < 4; ++q) t += bred[w][nt][jj + 16 * q];
    P.bias_partial[(size_t)blockIdx.x * OCT + tid] = t;
  }
}

// The same gradient for MORE than 32 columns (Cout <= 3 with up to 85 taps: SRGAN-G's 9x9 64 -> 3 output conv ran the
// role-swapped exact-fp32 k_wgrad_mfma_smallcin<16> at 159 us): the columns are split over the 4 waves of a block -- wave w
// owns columns [64 w, 64 w + 64), four 16-column tiles -- and all four walk the SAME K steps (the x fragments are loaded by
// every wave: L1 hits; the dyc gathers differ).  Nothing is combined across waves: each stores its own columns of the slab.
template <int OCT>
__global__ __launch_bounds__(256, 2) void k_wgrad_tapnw(WgTapnParams P) {
  __shared__ float bred[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kq = lane >> 4;
  const int T = P.KH * P.KW, NN = T * OCT;
  constexpr int NTW = 4;
  bool on[NTW], centre[NTW];
  int tu[NTW], tv[NTW], toc[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const int nn = (wave * NTW + nt) * 16 + j;
    on[nt] = nn < NN;
    const int t = on[nt] ? nn / OCT : 0;
    toc[nt] = on[nt] ? nn - t * OCT : 0;
    tu[nt] = t / P.KW;
    tv[nt] = t - tu[nt] * P.KW;
    centre[nt] = on[nt] && tu[nt] == P.pad && tv[nt] == P.pad;
  }
  const bool ch_on = 4 * j < P.Cin;
  f32x4 acc[4][NTW];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[i][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;   // (at most one of this lane's columns is a centre-tap column)
  for (int unit = blockIdx.x; unit < P.units; unit += gridDim.x) {
    const int rr = unit / P.S, sg = unit - rr * P.S;
    const int n = rr / P.H, r = rr - n * P.H;
    const int cb = sg * 32 + kq * 8;  // first pixel column of this lane's 8
    // x fragments: float4 = channels 4j..4j+3 of pixel cb + e (unconditional loads from clamped addresses, zeroed afterwards)
    f32x4 araw[8];
    const float* __restrict__ xs = P.x + ((size_t)(n * P.H + r) * P.W) * P.Cin + (ch_on ? 4 * j : 0);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int cx = cb + e < P.W ? cb + e : P.W - 1;
      araw[e] = *reinterpret_cast<const f32x4*>(xs + (size_t)cx * P.Cin);
    }
    // dyc fragments of this wave's columns (gather; zero outside dy), from clamped addresses as well
    float braw[NTW][8];
    bool bok[NTW][8];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int qy = r + P.pad - tu[nt];
      const bool oky = on[nt] && (unsigned)qy < (unsigned)P.OH;
      const int qyc = oky ? qy : 0;
      const float* __restrict__ src = P.dy + ((size_t)(n * P.OH + qyc) * P.OW) * OCT + toc[nt];
      const int qx0 = cb + P.pad - tv[nt];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int qx = qx0 + e;
        bok[nt][e] = oky && (unsigned)qx < (unsigned)P.OW;
        braw[nt][e] = src[(size_t)(bok[nt][e] ? qx : 0) * OCT];
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (!(ch_on && cb + e < P.W)) araw[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint4 a[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = araw[e][i];
      split8n<2>(f, a[i]);
    }
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = bok[nt][e] ? braw[nt][e] : 0.f;
      if (centre[nt]) {
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) t += f[e];
        bsum += t;
      }
      uint4 bfr[2];
      split8n<2>(f, bfr);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][nt] = mfma16(a[i][1], bfr[0], acc[i][nt]);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][nt] = mfma16(a[i][0], bfr[1], acc[i][nt]);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][nt] = mfma16(a[i][0], bfr[0], acc[i][nt]);
    }
  }
  // the block's slab [t][ci][oc]: C/D layout col = lane & 15 (nn), row = (lane >> 4) * 4 + reg; row m of tile i is channel 4m + i
  float* slab = P.ws + (size_t)blockIdx.x * T * P.Cin * OCT;
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    if (!on[nt]) continue;
    const int t = tu[nt] * P.KW + tv[nt];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ci = 4 * (kq * 4 + e) + i;
        if (ci < P.Cin) slab[((size_t)t * P.Cin + ci) * OCT + toc[nt]] = acc[i][nt][e];
      }
  }
  if (P.bias_partial) {   // the centre tap's dyc column IS dy: its K sums are the bias gradient partials
    // a centre column (u = v = pad, channel oc) sits on the 4 lanes (j, kq = 0..3) of ONE wave: sum over kq through LDS
    bred[wave][lane] = bsum;
    __syncthreads();
    if (tid < OCT) {
      const int nn = (P.pad * P.KW + P.pad) * OCT + tid;
      const int w = nn >> 6, jj = nn & 15;
      // (which of the wave's four column tiles holds it does not matter: a lane has at most one centre column)
      float t = 0.f;
      for (int q = 0; q < 4; ++q) t += bred[w][jj + 16 * q];
      P.bias_partial[(size_t)blockIdx.x * OCT + tid] = t;
    }
  }
}

static bool wgrad_tapn_wide(const srk_conv_desc& d) { return d.KH * d.KW * d.Cout > 32; }

static int wgrad_tapn_blocks(const srk_conv_desc& d) {
  if (wgrad_tapn_wide(d)) {   // all waves of a block walk the same K steps: >= 8 per block, two blocks per CU
    const long units = (long)d.N * d.H * ((d.W + 31) / 32);
    long g = (units + 7) / 8;
    if (g > 2 * kNumCU) g = 2 * kNumCU;
    return g < 1 ? 1 : (int)g;
  }
  const long units = (long)d.N * d.H * ((d.W + 31) / 32);
  long g = (units + 15) / 16;  // >= 4 K steps per wave (measured: 2..4 equal, 8 slower on the 16-image shard)
  if (g > 4 * kNumCU) g = 4 * kNumCU;  // 4 blocks per CU resident (36 KB LDS, 123 VGPRs)
  return g < 1 ? 1 : (int)g;
}

bool conv_wgrad_tapn_supported(const srk_conv_desc& d, const float* x, const srk_bwd_mask* mask) {
  const bool off = env_int("SRK_TAPN", 1) == 0;
  if (off) return false;
  if (d.transposed || d.stride != 1 || d.dy_ps_r > 1 || (mask && mask->y)) return false;
  if (d.Cout < 1 || d.Cout > 3 || d.KH * d.KW * d.Cout > 256) return false;
  if (d.KH * d.KW * d.Cout > 32) {   // the column-split kernel: 0 never, 2 on problems of any size (tests); small ones keep the old path
    const int mode = env_int("SRK_WGRAD_TAPNW", 1);
    if (mode == 0 || (mode != 2 && (long)d.N * d.H * d.W < 64L * 1024)) return false;
  }
  if (d.Cin < 8 || d.Cin > 64 || d.Cin % 4 != 0) return false;
  if (2 * d.pad > d.KH - 1 || 2 * d.pad > d.KW - 1) return false;
  if ((long)d.N * d.H * ((d.W + 31) / 32) >= (1L << 30)) return false;
  if (x && (uintptr_t)x % 16 != 0) return false;
  return true;
}

size_t conv_wgrad_tapn_ws(const srk_conv_desc& d) {
  return (size_t)wgrad_tapn_blocks(d) * ((size_t)d.KH * d.KW * d.Cin * d.Cout + d.Cout) * sizeof(float);
}

int conv_wgrad_tapn(const srk_conv_desc& d, const float* x, const float* dy, float* dw, float* db, float beta, void* ws,
                    size_t ws_bytes, hipStream_t s) {
  const int G = wgrad_tapn_blocks(d);
  const size_t need = conv_wgrad_tapn_ws(d);
  if (!ws || ws_bytes < need) {
    set_error("conv_wgrad_tapn: workspace %zu < %zu", ws_bytes, need);
    return SRK_ERR_WORKSPACE;
  }
  WgTapnParams P{};
  P.x = x; P.dy = dy; P.ws = (float*)ws;
  const size_t slab_floats = (size_t)G * d.KH * d.KW * d.Cin * d.Cout;
  P.bias_partial = db ? P.ws + slab_floats : nullptr;
  P.N = d.N; P.H = d.H; P.W = d.W; P.Cin = d.Cin; P.OH = d.OH; P.OW = d.OW; P.pad = d.pad; P.KH = d.KH; P.KW = d.KW;
  P.S = (d.W + 31) / 32;
  P.units = d.N * d.H * P.S;
  if (wgrad_tapn_wide(d)) {
    switch (d.Cout) {
      case 1: hipLaunchKernelGGL(k_wgrad_tapnw<1>, dim3(G), dim3(256), 0, s, P); break;
      case 2: hipLaunchKernelGGL(k_wgrad_tapnw<2>, dim3(G), dim3(256), 0, s, P); break;
      default: hipLaunchKernelGGL(k_wgrad_tapnw<3>, dim3(G), dim3(256), 0, s, P); break;
    }
  } else {
    switch (d.Cout) {
      case 1: hipLaunchKernelGGL(k_wgrad_tapn<1>, dim3(G), dim3(256), 0, s, P); break;
      case 2: hipLaunchKernelGGL(k_wgrad_tapn<2>, dim3(G), dim3(256), 0, s, P); break;
      default: hipLaunchKernelGGL(k_wgrad_tapn<3>, dim3(G), dim3(256), 0, s, P); break;
    }
  }
  int rc = check_launch("conv_wgrad_tapn");
  if (rc) return rc;
  return conv_wgrad_reduce_launch(P.ws, dw, G, d.Cout, d.Cin, d.KH, d.KW, 0, beta, P.bias_partial, db, d.Cout, 0, s);
}

// ---------------------------------------------------------------------------------------------
// The mirror case: gathers with <= 3 INPUT channels and KH*KW*IC <= 32 (the data gradient of the reconstruction
// convs: dx[64] from dy[3]), written for the TRANS gathers that the row-packed bf16x3 first-layer kernel
// (k_conv_bf3_rows) does not take.  Here the taps join the contraction index:
//
//   out[p][oc] = sum over k = (t, c) of  inc[p][k] * w[k][oc],   inc[p][(t, c)] = in[p + offset(t)][c]    (K <= 32)
//
// computed transposed (M = oc in 16-channel tiles, N = 16 consecutive pixels of a row, ONE K step): the filter
// (A operand) stays in registers as bf16 planes, the B operand is 8 gathered floats per lane and 16 pixels, and the
// C/D layout (lane = pixel, 4 registers = 4 consecutive channels) stores float4s straight to global memory --
// no LDS, no barriers; waves walk the 16-pixel groups of the batch with a grid stride.  bf16x3 arithmetic.
// The kernel is bound by its output write (e.g. 537 MB for 128 x 64 x 128 x 128).
// ---------------------------------------------------------------------------------------------
template <int MTN>  // 16-channel output tiles (OC = 16 * MTN)
__global__ __launch_bounds__(256) void k_conv_tapk(MfmaConvParams P, int groups_per_row, int units) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kq = lane >> 4;
  const int T = P.KHv * P.KWv, KK = T * P.IC;
  // filter fragments: row m = j of tile i is output channel 16 i + j; k = kq*8 + e = t*IC + c
  uint4 af[MTN][2];
  int koff[8];       // gather offset of contraction slot e relative to the pixel's base address (elements)
  int kuv[8];        // (u << 16) | v of slot e, -1 = padding slot
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = kq * 8 + e;
    const bool on = k < KK;
    const int t = on ? k / P.IC : 0, c = k - t * P.IC;
    const int u = t / P.KWv, v = t - u * P.KWv;
    koff[e] = (u * P.IW + v) * P.IC + c;
    kuv[e] = on ? ((u << 16) | v) : -1;
  }
#pragma unroll
  for (int i = 0; i < MTN; ++i) {
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      // (unconditional load of slot 0 for padding slots + select: loads under a divergent branch are each followed by an
      //  s_waitcnt vmcnt(0) at the join -- 6 serialised round trips in the prologue of every block)
      const int k = kuv[e] >= 0 ? kq * 8 + e : 0;
      const int t = k / P.IC, c = k - t * P.IC;
      const int u = t / P.KWv, v = t - u * P.KWv;
      const int tapw = (P.wh0 + P.wdh * u) * P.KW_full + (P.ww0 + P.wdw * v);
      f[e] = P.wp[((size_t)tapw * P.IC + c) * P.OC + 16 * i + j];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = kuv[e] >= 0 ? f[e] : 0.f;
    split8n<2>(f, af[i]);
  }
  // two pixel groups per iteration: 16 independent gathers in flight before the first conversion
  constexpr int U = 2;
  const int nw = gridDim.x * 4;
  const float* __restrict__ res = P.ep.residual;
  for (int unit0 = (blockIdx.x * 4 + wave) * U; unit0 < units; unit0 += nw * U) {
    float f[U][8];
    size_t ooff[U];
    bool px_on[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int unit = unit0 + q < units ? unit0 + q : units - 1;
      const int rr = unit / groups_per_row, gx = unit - rr * groups_per_row;
      const int n = rr / P.PH, pr = rr - n * P.PH;
      const int pc = gx * 16 + j;  // this lane's pixel (phase coordinates)
      px_on[q] = pc < P.PW && unit0 + q < units;
      const int iyb = pr * P.is + P.iy0, ixb = pc * P.is + P.ix0;
      const float* __restrict__ src = P.in + (((size_t)n * P.IH + iyb) * P.IW + ixb) * P.IC;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int u = kuv[e] >> 16, v = kuv[e] & 0xffff;
        const bool ok = px_on[q] && kuv[e] >= 0 && (unsigned)(iyb + u) < (unsigned)P.IH && (unsigned)(ixb + v) < (unsigned)P.IW;
        f[q][e] = ok ? src[koff[e]] : 0.f;
      }
      const int oy = P.oy0 + pr * P.os, ox = P.ox0 + pc * P.os;
      ooff[q] = (((size_t)n * P.OH + oy) * P.OW + ox) * P.OC + kq * 4;
    }
#pragma unroll
    for (int q = 0; q < U; ++q) {
      uint4 bf[2];
      split8n<2>(f[q], bf);
#pragma unroll
      for (int i = 0; i < MTN; ++i) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc = mfma16(af[i][1], bf[0], acc);
        acc = mfma16(af[i][0], bf[1], acc);
        acc = mfma16(af[i][0], bf[0], acc);
        // C/D layout: col = lane & 15 = pixel j, rows kq*4 + reg = channels 16 i + kq*4 .. +3
        if (px_on[q]) {
          if (res) acc += *reinterpret_cast<const f32x4*>(res + ooff[q] + 16 * i);
          *reinterpret_cast<f32x4*>(P.out + ooff[q] + 16 * i) = acc;
        }
      }
    }
  }
}

// (epilogue: optional "+ residual" only -- what a data gradient needs)
bool conv_tapk_gather_supported(const GatherConv& g, const Epi& ep, const float* out, const float* mask_y) {
  if (ep.bias || ep.act != SRK_ACT_NONE || ep.ps_r > 1 || (uintptr_t)out % 16 != 0 || (uintptr_t)ep.residual % 16 != 0)
    return false;
  const bool off = env_int("SRK_TAPN", 1) == 0;
  if (off) return false;
  if (!g.trans || g.stride != 1) return false;  // CONV gathers with IC <= 4 have the row-packed kernel
  if (g.IC < 1 || g.IC > 3 || g.KH * g.KW * g.IC > 32) return false;
  if (g.OC % 16 != 0 || g.OC > 64) return false;
  if (mask_y || g.in_nchw || g.in_ps_r > 1) return false;
  if ((long)g.IH * g.IW * g.IC >= (1L << 30) || (long)g.N * g.OH * ((g.OW + 15) / 16) >= (1L << 30)) return false;
  return true;
}

int conv_tapk_gather(const GatherConv& g, const float* in, const float* wp, float* out, const Epi& ep, hipStream_t s) {
  return for_each_phase(g, in, wp, out, ep, nullptr, 0.f, [&](const MfmaConvParams& P) {
    const int gpr = (P.PW + 15) / 16;
    const int units = P.N * P.PH * gpr;
    long nb = (units + 15) / 16;  // >= 4 pixel groups per wave
    if (nb > 8 * kNumCU) nb = 8 * kNumCU;
    if (nb < 1) nb = 1;
    switch (P.OC / 16) {
      case 1: hipLaunchKernelGGL(k_conv_tapk<1>, dim3((unsigned)nb), dim3(256), 0, s, P, gpr, units); break;
      case 2: hipLaunchKernelGGL(k_conv_tapk<2>, dim3((unsigned)nb), dim3(256), 0, s, P, gpr, units); break;
      case 3: hipLaunchKernelGGL(k_conv_tapk<3>, dim3((unsigned)nb), dim3(256), 0, s, P, gpr, units); break;
      default: hipLaunchKernelGGL(k_conv_tapk<4>, dim3((unsigned)nb), dim3(256), 0, s, P, gpr, units); break;
    }
    return check_launch("conv_tapk");
  });
}

// ---------------------------------------------------------------------------------------------
// The same mirror case with MORE than 32 contraction slots: the data gradient of a many-tap few-channel output conv
// (SRGAN-G's 9x9 64 -> 3, srgan.py:32: dx[64] from dy[3], a stride-1 TRANS gather with IC = 3, OC = 64, 81 taps).  It ran
// the exact-fp32 k_conv_mfma_tg<4> -- one 128-pixel tile per block walking 81 taps: 118 - 127 us for 8.15 GFLOP.
//
// K is packed ONE KERNEL ROW PER STEP: slot k' = v * IC + c of step u (k' < KW * IC <= 32; the rest of the step is zero
// filter) -- the KW * IC values of a row are CONTIGUOUS in an NHWC tensor with IC channels, so lane (pixel j, kq) gathers
// the 8 consecutive floats k' = 8 kq .. 8 kq + 7 at (iy + u, ix) * IC + k'.  Transposed product as k_conv_tapk (M = oc in
// 16-channel tiles, N = 16 consecutive pixels of a row; the C/D layout stores float4s of 4 channels per pixel, no LDS
// epilogue).  The filter fragments [kernel row][channel tile][plane] (73 KB for 9 x 4 x 2) live in LDS: the fp32 filter is
// copied there in whole lines, turned into fragments through registers, and read back as one ds_read_b128 per MFMA
// operand; a wave keeps the gathered rows of TWO pixel groups in flight (the loads of step u + 1 -- two 16-byte loads per
// lane and group -- are issued in front of the MFMAs of step u).  bf16x3 arithmetic, products smallest-first; 9 steps x 4 tiles x 3 MFMAs per 16 pixels.
// ---------------------------------------------------------------------------------------------
template <int MTN>  // 16-channel output tiles (OC = 16 * MTN)
__global__ __launch_bounds__(256, 2) void k_conv_tapkm(MfmaConvParams P, int groups_per_row, int units) {
  extern __shared__ __attribute__((aligned(16))) uint4 fsm4[];   // fragments [KHv][MTN][2 planes][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kq = lane >> 4;
  const int KH = P.KHv, RK = P.KWv * P.IC;     // K slots of a kernel row that carry filter values (<= 32)
  // ---- filter: global fp32 [tap][IC][OC] -> LDS (coalesced) -> fragments through registers -> LDS
  {
    float* wsm = reinterpret_cast<float*>(fsm4);
    const int nflt = P.KHv * P.KW_full * P.IC * P.OC;   // (stride-1 gather: KHv x KWv is the whole filter)
    if ((nflt & 3) == 0 && (reinterpret_cast<uintptr_t>(P.wp) & 15) == 0) {
      for (int i = tid; i < (nflt >> 2); i += 256) reinterpret_cast<f32x4*>(wsm)[i] = reinterpret_cast<const f32x4*>(P.wp)[i];
    } else {
      for (int i = tid; i < nflt; i += 256) wsm[i] = P.wp[i];
    }
    __syncthreads();
    constexpr int MAXF = (9 * MTN * 64 + 255) / 256;   // fragments (u, tile, lane) per thread, KH <= 9
    uint4 fr[MAXF][2];
    const int nfrag = KH * MTN * 64;
#pragma unroll
    for (int r = 0; r < MAXF; ++r) {
      const int f = tid + 256 * r;
      const int fl = f & 63, fi = (f >> 6) % MTN, fu = (f >> 6) / MTN;
      const int fj = fl & 15, fkq = fl >> 4;
      float v8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = fkq * 8 + e;
        const bool on = f < nfrag && k < RK;
        const int kk = on ? k : 0, uu = f < nfrag ? fu : 0;
        const int v = kk / P.IC, c = kk - v * P.IC;
        const int tapw = (P.wh0 + P.wdh * uu) * P.KW_full + (P.ww0 + P.wdw * v);
        const float w = wsm[((size_t)tapw * P.IC + c) * P.OC + 16 * fi + fj];
        v8[e] = on ? w : 0.f;
      }
      split8n<2>(v8, fr[r]);
    }
    __syncthreads();   // every source value is in registers
#pragma unroll
    for (int r = 0; r < MAXF; ++r) {
      const int f = tid + 256 * r;
      if (f < nfrag) {
        const int fl = f & 63, fq = f >> 6;   // fq = u * MTN + tile
        fsm4[(size_t)(fq * 2 + 0) * 64 + fl] = fr[r][0];
        fsm4[(size_t)(fq * 2 + 1) * 64 + fl] = fr[r][1];
      }
    }
    __syncthreads();
  }
  // this lane's K slots: column shift and channel are the same for every step
  int slot_v[8];
  bool slot_on[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = kq * 8 + e;
    slot_on[e] = k < RK;
    slot_v[e] = (slot_on[e] ? k : 0) / P.IC;
  }
  constexpr int U = 2;
  const int nw = gridDim.x * 4;
  const float* __restrict__ res = P.ep.residual;
  for (int unit0 = (blockIdx.x * 4 + wave) * U; unit0 < units; unit0 += nw * U) {
    size_t ooff[U];
    bool px_on[U];
    int iyb[U], ixb[U];
    const float* src[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int unit = unit0 + q < units ? unit0 + q : units - 1;
      const int rr = unit / groups_per_row, gx = unit - rr * groups_per_row;
      const int n = rr / P.PH, pr = rr - n * P.PH;
      const int pc = gx * 16 + j;  // this lane's pixel (phase coordinates)
      px_on[q] = pc < P.PW && unit0 + q < units;
      iyb[q] = pr * P.is + P.iy0;
      ixb[q] = pc * P.is + P.ix0;
      src[q] = P.in + (size_t)n * P.IH * P.IW * P.IC + kq * 8;
      const int oy = P.oy0 + pr * P.os, ox = P.ox0 + pc * P.os;
      ooff[q] = (((size_t)n * P.OH + oy) * P.OW + ox) * P.OC + kq * 4;
    }
    f32x4 acc[MTN][U];
#pragma unroll
    for (int i = 0; i < MTN; ++i)
#pragma unroll
      for (int q = 0; q < U; ++q) acc[i][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float raw[2][U][8];
    unsigned okm[2][U];
    // The 8 floats of kernel row u for both pixel groups.  They are CONTIGUOUS in memory (k' = v * IC + c), so a lane issues
    // two 16-byte loads (4-byte aligned) instead of eight dword gathers -- with dword gathers the kernel was bound by the
    // address path: 72 wave-loads of 64 scattered dwords per 16 pixels, 88 us.  The vector may start left of its row (image
    // edge: those slots are masked, the bytes belong to the row above) -- only a vector that leaves the TENSOR falls back to
    // clamped scalar loads (first / last rows of the batch).
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    const long in_elems = (long)P.N * P.IH * P.IW * P.IC;
    auto gather = [&](int buf, int u) {
#pragma unroll
      for (int q = 0; q < U; ++q) {
        const int iy = iyb[q] + u;
        const bool rok = px_on[q] && u < KH && (unsigned)iy < (unsigned)P.IH;
        const int iyc = rok ? iy : 0;
        unsigned m = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int ix = ixb[q] + slot_v[e];
          const bool ok = rok && slot_on[e] && (unsigned)ix < (unsigned)P.IW;
          m |= ok ? (1u << e) : 0u;
        }
        okm[buf][q] = m;
        // element (iy, ixb, k' = 8 kq + e) = src + (iy * IW + ixb) * IC + e   (src already holds image n and + 8 kq)
        const long start = ((long)iyc * P.IW + ixb[q]) * P.IC;
        const long abs0 = (src[q] - P.in) + start;
        if (m != 0u && abs0 >= 0 && abs0 + 8 <= in_elems) {
          const f32x4u v0 = *reinterpret_cast<const f32x4u*>(src[q] + start);
          const f32x4u v1 = *reinterpret_cast<const f32x4u*>(src[q] + start + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            raw[buf][q][e] = v0[e];
            raw[buf][q][4 + e] = v1[e];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) raw[buf][q][e] = ((m >> e) & 1u) ? src[q][start + e] : 0.f;
        }
      }
    };
    gather(0, 0);
    for (int u = 0; u < KH; ++u) {
      const int cur = u & 1;
      // (dynamic buffer index would spill: two copies of the body, selected by the wave-uniform parity)
      auto body = [&](auto curc) {
        constexpr int cb = decltype(curc)::value;
        gather(cb ^ 1, u + 1);
        uint4 bf[U][2];
#pragma unroll
        for (int q = 0; q < U; ++q) {
          float f8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) f8[e] = ((okm[cb][q] >> e) & 1u) ? raw[cb][q][e] : 0.f;
          split8n<2>(f8, bf[q]);
        }
#pragma unroll
        for (int i = 0; i < MTN; ++i) {
          const uint4 ah = fsm4[(size_t)((u * MTN + i) * 2 + 0) * 64 + lane];
          const uint4 am = fsm4[(size_t)((u * MTN + i) * 2 + 1) * 64 + lane];
#pragma unroll
          for (int q = 0; q < U; ++q) acc[i][q] = mfma16(am, bf[q][0], acc[i][q]);
#pragma unroll
          for (int q = 0; q < U; ++q) acc[i][q] = mfma16(ah, bf[q][1], acc[i][q]);
#pragma unroll
          for (int q = 0; q < U; ++q) acc[i][q] = mfma16(ah, bf[q][0], acc[i][q]);
        }
      };
      if (cur == 0) body(std::integral_constant<int, 0>{}); else body(std::integral_constant<int, 1>{});
    }
    // C/D layout: col = lane & 15 = pixel j, rows kq*4 + reg = channels 16 i + kq*4 .. +3
#pragma unroll
    for (int q = 0; q < U; ++q) {
      if (px_on[q]) {
#pragma unroll
        for (int i = 0; i < MTN; ++i) {
          f32x4 v = acc[i][q];
          if (res) v += *reinterpret_cast<const f32x4*>(res + ooff[q] + 16 * i);
          *reinterpret_cast<f32x4*>(P.out + ooff[q] + 16 * i) = v;
        }
      }
    }
  }
}

// (epilogue: optional "+ residual" only -- what a data gradient needs)
bool conv_tapkm_gather_supported(const GatherConv& g, const Epi& ep, const float* out, const float* mask_y) {
  if (ep.bias || ep.act != SRK_ACT_NONE || ep.ps_r > 1 || (uintptr_t)out % 16 != 0 || (uintptr_t)ep.residual % 16 != 0)
    return false;
  if (env_int("SRK_TAPKM", 1) == 0) return false;
  if (!g.trans || g.stride != 1) return false;
  if (g.IC < 1 || g.IC > 4 || g.KH * g.KW * g.IC <= 32 || g.KW * g.IC > 32 || g.KH > 9) return false;
  if (g.OC % 16 != 0 || g.OC > 64) return false;
  if (mask_y || g.in_nchw || g.in_ps_r > 1) return false;
  if ((long)g.IH * g.IW * g.IC >= (1L << 30) || (long)g.N * g.OH * ((g.OW + 15) / 16) >= (1L << 30)) return false;
  return true;
}

template <int MTN>
static int tapkm_launch(const MfmaConvParams& P, int gpr, int units, hipStream_t s) {
  size_t lds = (size_t)P.KHv * MTN * 2 * 64 * 16;
  const size_t flt = (size_t)P.KHv * P.KW_full * P.IC * P.OC * 4;
  if (lds < flt) lds = flt;
  static LdsLimit lim;
  lim.ensure(reinterpret_cast<const void*>(&k_conv_tapkm<MTN>), lds);
  long nb = (units + 4 * 2 * 4 - 1) / (4 * 2 * 4);   // >= 4 pairs of pixel groups per wave (the filter prologue is per block)
  if (nb > 2 * kNumCU) nb = 2 * kNumCU;
  if (nb < 1) nb = 1;
  note_kernel("k_conv_tapkm<%d>", MTN);
  hipLaunchKernelGGL(k_conv_tapkm<MTN>, dim3((unsigned)nb), dim3(256), lds, s, P, gpr, units);
  return check_launch("conv_tapkm");
}

int conv_tapkm_gather(const GatherConv& g, const float* in, const float* wp, float* out, const Epi& ep, hipStream_t s) {
  return for_each_phase(g, in, wp, out, ep, nullptr, 0.f, [&](const MfmaConvParams& P) {
    if (P.is != 1 || P.KHv != g.KH || P.KWv != g.KW) {
      set_error("conv_tapkm: strided gather");
      return (int)SRK_ERR_UNSUPPORTED;
    }
    const int gpr = (P.PW + 15) / 16;
    const int units = P.N * P.PH * gpr;
    switch (P.OC / 16) {
      case 1: return tapkm_launch<1>(P, gpr, units, s);
      case 2: return tapkm_launch<2>(P, gpr, units, s);
      case 3: return tapkm_launch<3>(P, gpr, units, s);
      default: return tapkm_launch<4>(P, gpr, units, s);
    }
  });
}


}  // namespace srk
