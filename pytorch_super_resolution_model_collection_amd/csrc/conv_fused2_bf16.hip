// Two convolutions fused through LDS (bf16x3 arithmetic, see conv_mfma_bf16.hip):
//
//     y = act2( conv2( act1( conv1(x) ) ) )         conv1: Cin <= 4 -> C1 (multiple of 64), stride 1
//                                                   conv2: C1 -> C2, stride 1
//
// e.g. ESPCN's first two layers (espcn.py:18-19: conv5 3->64 + ReLU, conv3 64->32 + ReLU).  The
// C1-channel intermediate is the largest tensor of the network (16.3 MB per 256x256 image, written
// once and read once = 53 % of the net's compulsory HBM traffic); here it never leaves the CU: for
// every output tile of conv2 the block
//   1. stages the tiny Cin<=4 input halo ((TH+KH2+KH1-2) x (TW+KW2+KW1-2) pixels) in LDS,
//   2. computes conv1 + bias + act1 for the (TH+KH2-1) x (TW+KW2-1) halo of conv2 on the matrix
//      cores (row-packed K: 8 kw x 4 channels per K step), 32 output channels at a time, and writes
//      the result — already split into bf16 hi/lo — into the [plane][8-channel group][pixel] LDS
//      layout that the conv2 MFMA loop reads (the MFMAs run transposed, D^T = W^T A^T, so a lane
//      holds 4 consecutive channels of a pixel = one 8-byte LDS store per plane),
//   3. runs the conv2 taps for that 32-channel chunk exactly like k_conv_bf3.
// The conv1 work is split by output-channel halves, so nothing is recomputed except the halo
// overlap between neighbouring tiles (1.3x for 10x25 tiles).
#include "srk_common.h"
#include "conv_problem.h"
#include "conv_tile.h"
#include <stdlib.h>

namespace srk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int F2_MT1 = 6;          // conv1 M-tiles (16 halo pixels each) per wave -> halo <= 384 pixels
constexpr int F2_EPI_STRIDE = 68;  // same staged-epilogue slab as conv_mfma_bf16.hip

struct Fused2Params {
  MfmaConvParams P;   // conv2 geometry / epilogue (in = unused)
  const float* x;     // network input [N, IH0, IW0, IC1] (NHWC) or [N, IC1, IH0, IW0] (x_nchw)
  const uint4* wq1;   // conv1 filters, row-packed prepared layout  [KH1][KS1][ocb][plane][g][64][8]
  const uint4* wq2;   // conv2 filters, prepared layout             [tap][chunk][ocb][plane][g][NB2][8]
  const float* bias1;
  float slope1;
  int act1;
  int x_nchw;
  int IH0, IW0, IC1;
  int C1, KH1, KW1, pad1, KS1, OCb1;
  int H1, W1;         // conv1 output size (= conv2 input size)
  int HH0, HW0;       // input halo of the tile
  int NPIX0p, NPIX2p;
  int ICc2, OCb2, NB2;
};

__device__ __forceinline__ f32x4 f2_mfma(const uint4& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0,
                                                 0);
}

template <int NT>
__global__ __launch_bounds__(256, 2) void k_conv_fused2(Fused2Params F) {
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
  const MfmaConvParams& P = F.P;
  uint2* l0 = reinterpret_cast<uint2*>(smem4);        // [2 planes][NPIX0p] x (4 bf16)
  uint4* hal = smem4 + F.NPIX0p;                      // [2][4][NPIX2p]
  uint4* wl = hal + 8 * F.NPIX2p;                     // [2 bufs][wslot]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kq = lane >> 4;
  int b = blockIdx.x;
  const int txi = b % P.tiles_x;
  b /= P.tiles_x;
  const int tyi = b % P.tiles_y;
  const int n = b / P.tiles_y;
  const int r0 = tyi * P.TH, c0 = txi * P.TW;   // conv2 output tile origin
  const int ocbi = blockIdx.y;
  const int ocb = ocbi * 64;
  const int npx = P.TH * P.TW;
  const int NB2 = F.NB2;
  const int wslot2 = 8 * NB2;
  const int wslot = wslot2 > 256 ? wslot2 : 256;  // conv1 half-slices are 256 uint4
  const int T2 = P.KHv * P.KWv;
  const int Q1 = F.KH1 * F.KS1;
  const int npix2 = P.HH * P.HW;                // conv2 halo pixels = conv1 outputs needed
  const int npix0 = F.HH0 * F.HW0;
  // conv2 halo origin in conv1-output coordinates, input halo origin in input coordinates
  const int h2y = r0 + P.iy0, h2x = c0 + P.ix0;   // (P.is == 1)
  const int h0y = h2y - F.pad1, h0x = h2x - F.pad1;

  // ---- phase A: input halo (<= 4 channels) -> hi/lo planes, pad pixels zeroed
  for (int hp = tid; hp < F.NPIX0p; hp += 256) {
    float f[4] = {0.f, 0.f, 0.f, 0.f};
    if (hp < npix0) {
      const int hy = hp / F.HW0, hx = hp - hy * F.HW0;
      const int iy = h0y + hy, ix = h0x + hx;
      if (iy >= 0 && iy < F.IH0 && ix >= 0 && ix < F.IW0) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (e < F.IC1)
            f[e] = F.x_nchw ? F.x[(((size_t)n * F.IC1 + e) * F.IH0 + iy) * F.IW0 + ix]
                            : F.x[(((size_t)n * F.IH0 + iy) * F.IW0 + ix) * F.IC1 + e];
      }
    }
    bf16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const __bf16 hh = (__bf16)f[e];
      h[e] = hh;
      l[e] = (__bf16)(f[e] - (float)hh);
    }
    l0[hp] = __builtin_bit_cast(uint2, h);
    l0[F.NPIX0p + hp] = __builtin_bit_cast(uint2, l);
  }

  // conv1 M-tiles of this wave: tile index wave + 4*i ; lane's pixel = tile*16 + j
  int p1[F2_MT1], off1[F2_MT1];
  bool v1[F2_MT1];
#pragma unroll
  for (int i = 0; i < F2_MT1; ++i) {
    const int p = (wave + 4 * i) * 16 + j;
    p1[i] = p;
    const int pp = p < npix2 ? p : 0;
    const int r = pp / P.HW, c = pp - r * P.HW;
    off1[i] = r * F.HW0 + c + 2 * kq;
    const int y1 = h2y + r, x1 = h2x + c;  // position in the conv1 output image
    v1[i] = p < npix2 && y1 >= 0 && y1 < F.H1 && x1 >= 0 && x1 < F.W1;
  }
  // conv2 fragment offsets
  int hp2[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    int m = wave * 64 + mt * 16 + j;
    if (m >= npx) m = 0;
    const int r = m / P.TW, c = m - r * P.TW;
    hp2[mt] = r * P.HW + c + kq * F.NPIX2p;
  }
  f32x4 acc[4][NT];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool wave_live = wave * 64 < npx;
  const int lo_plane = 4 * F.NPIX2p;
  const int mt1_count = (npix2 + 15) >> 4;

  for (int cc = 0; cc < F.ICc2; ++cc) {
    // ================= conv1: output channels [32cc, 32cc+32) for every conv2-halo pixel =========
    // filter half-slice for K step q: [plane][g][32 co][8] gathered from the 64-wide prepared block
    auto w1src = [&](int q) -> uint4 {
      const int pl = tid >> 7, g = (tid >> 5) & 3, col = tid & 31;
      const int co = cc * 32 + col;
      const uint4* blk = F.wq1 + ((size_t)q * F.OCb1 + (co >> 6)) * (size_t)512;
      return blk[(pl * 4 + g) * 64 + (co & 63)];
    };
    f32x4 a1[F2_MT1][2];
#pragma unroll
    for (int i = 0; i < F2_MT1; ++i) {
      a1[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      a1[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();  // l0 staged (cc == 0) / previous chunk's conv2 taps done with wl and hal
    wl[tid] = w1src(0);
    __syncthreads();
    {
      int u = 0, ks = 0;
      for (int q = 0; q < Q1; ++q) {
        uint4 wr = {0, 0, 0, 0};
        if (q + 1 < Q1) wr = w1src(q + 1);
        const uint4* wb = wl + (q & 1) * wslot;
        const int toff = u * F.HW0 + ks * 8;
        // filter fragments (row operand): lane (j = co within tile, kq = kw pair)
        uint4 bh[2], bl[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          bh[nt] = wb[(0 * 4 + kq) * 32 + nt * 16 + j];
          bl[nt] = wb[(1 * 4 + kq) * 32 + nt * 16 + j];
        }
#pragma unroll
        for (int i = 0; i < F2_MT1; ++i) {
          if ((wave + 4 * i) < mt1_count) {   // wave-uniform
            const uint2* p = l0 + off1[i] + toff;
            const uint2 h0 = p[0], h1 = p[1];
            const uint2 q0 = p[F.NPIX0p], q1 = p[F.NPIX0p + 1];
            const uint4 ah = make_uint4(h0.x, h0.y, h1.x, h1.y);
            const uint4 al = make_uint4(q0.x, q0.y, q1.x, q1.y);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
              // transposed product: rows = output channels, cols = pixels
              a1[i][nt] = f2_mfma(bh[nt], al, a1[i][nt]);
              a1[i][nt] = f2_mfma(bl[nt], ah, a1[i][nt]);
              a1[i][nt] = f2_mfma(bh[nt], ah, a1[i][nt]);
            }
          }
        }
        if (q + 1 < Q1) wl[((q + 1) & 1) * wslot + tid] = wr;
        __syncthreads();
        if (++ks == F.KS1) {
          ks = 0;
          ++u;
        }
      }
    }
    // conv1 epilogue: bias + act1, split, 8-byte stores into the conv2 halo planes.
    // C/D layout (transposed): col = lane&15 = pixel, row = (lane>>4)*4 + reg = channel in tile.
    {
      uint2* hal2 = reinterpret_cast<uint2*>(hal);
#pragma unroll
      for (int i = 0; i < F2_MT1; ++i) {
        if ((wave + 4 * i) < mt1_count && p1[i] < npix2) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const int col = nt * 16 + kq * 4;          // channel within the 32-chunk
            const int co = cc * 32 + col;
            bf16x4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float v = 0.f;
              if (v1[i]) {  // outside the conv1 output image the conv2 input is zero padding
                v = a1[i][nt][e] + (F.bias1 ? F.bias1[co + e] : 0.f);
                v = act_apply(v, F.act1, F.slope1);
              }
              const __bf16 hh = (__bf16)v;
              h[e] = hh;
              l[e] = (__bf16)(v - (float)hh);
            }
            const int g = col >> 3, half = (col >> 2) & 1;
            hal2[((0 * 4 + g) * F.NPIX2p + p1[i]) * 2 + half] = __builtin_bit_cast(uint2, h);
            hal2[((1 * 4 + g) * F.NPIX2p + p1[i]) * 2 + half] = __builtin_bit_cast(uint2, l);
          }
        }
      }
    }
    // ================= conv2 taps for this 32-channel chunk ======================================
    auto w2src = [&](int t) -> const uint4* {
      const int u = t / P.KWv, v = t - u * P.KWv;
      const int tapw = (P.wh0 + P.wdh * u) * P.KW_full + (P.ww0 + P.wdw * v);
      return F.wq2 + ((size_t)(tapw * F.ICc2 + cc) * F.OCb2 + ocbi) * (size_t)wslot2;
    };
    const bool w0_ok = tid < wslot2, w1_ok = tid + 256 < wslot2;
    {
      const uint4* src = w2src(0);
      if (w0_ok) wl[tid] = src[tid];
      if (w1_ok) wl[tid + 256] = src[tid + 256];
    }
    __syncthreads();  // hal (conv1 result) and the first conv2 filter slice visible
    for (int t = 0; t < T2; ++t) {
      uint4 wr0 = {0, 0, 0, 0}, wr1 = {0, 0, 0, 0};
      if (t + 1 < T2) {
        const uint4* src = w2src(t + 1);
        if (w0_ok) wr0 = src[tid];
        if (w1_ok) wr1 = src[tid + 256];
      }
      if (wave_live) {
        const int u = t / P.KWv, v = t - u * P.KWv;
        const uint4* hb = hal + u * P.HW + v;
        const uint4* wb = wl + (t & 1) * wslot + kq * NB2 + j;
        uint4 ah[4], al[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          ah[mt] = hb[hp2[mt]];
          al[mt] = hb[hp2[mt] + lo_plane];
        }
        uint4 bh[NT], bl[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          bh[nt] = wb[nt * 16];
          bl[nt] = wb[4 * NB2 + nt * 16];
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f2_mfma(al[mt], bh[nt], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f2_mfma(ah[mt], bl[nt], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f2_mfma(ah[mt], bh[nt], acc[mt][nt]);
      }
      if (t + 1 < T2) {
        uint4* wn = wl + ((t + 1) & 1) * wslot;
        if (w0_ok) wn[tid] = wr0;
        if (w1_ok) wn[tid + 256] = wr1;
      }
      __syncthreads();
    }
  }

  // ---- conv2 epilogue: LDS-staged 16-byte stores (see bf3_epilogue in conv_mfma_bf16.hip)
  {
    float* st = reinterpret_cast<float*>(smem4) + wave * (32 * F2_EPI_STRIDE);
    constexpr int Q4 = NT * 4;
    constexpr int RPI = 64 / Q4;
    const int row0 = lane / Q4, q4 = lane - row0 * Q4;
    const int oc4 = ocb + q4 * 4;
    const bool lane_on = row0 < RPI && oc4 < P.OC;
    const int tw_magic = div_small_magic(P.TW);
    EpiCol col{};
    if (lane_on) col = epi_col_setup(P.ep, P.OW, P.OC, oc4);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg)
            st[(mh * 16 + kq * 4 + reg) * F2_EPI_STRIDE + nt * 16 + j] = acc[2 * h + mh][nt][reg];
      __syncthreads();
      if (lane_on) {
#pragma unroll 2
        for (int row = row0; row < 32; row += RPI) {
          const int m = wave * 64 + h * 32 + row;
          if (m < npx) {
            const int r = div_small(m, tw_magic), c = m - r * P.TW;
            const int pr = r0 + r, pc = c0 + c;
            if (pr < P.PH && pc < P.PW) {
              const epi_f4 v = *reinterpret_cast<const epi_f4*>(st + row * F2_EPI_STRIDE + q4 * 4);
              epi_store4_col(P.ep, col, P.OH, P.OW, P.OC, n, pr, pc, v, P.out);
            }
          }
        }
      }
      __syncthreads();
    }
  }
}

template <int NT>
static void f2_launch(const Fused2Params& F, dim3 grid, size_t lds, hipStream_t s) {
  static LdsLimit lim;
  lim.ensure(reinterpret_cast<const void*>(&k_conv_fused2<NT>), lds);
  hipLaunchKernelGGL(k_conv_fused2<NT>, grid, dim3(256), lds, s, F);
}

// Returns SRK_ERR_UNSUPPORTED (with a message) for anything outside the fused kernel's envelope; the
// caller then runs the two layers separately.
int conv_fused2_forward(const srk_conv_desc& d1, const srk_conv_desc& d2, const float* x, int x_nchw,
                        const float* wp1, const float* wp2, float* y, const Epi& ep1, const Epi& ep2, hipStream_t s) {
  if (d1.transposed || d2.transposed || d1.stride != 1 || d2.stride != 1 || d1.Cin > 4 || (d1.Cout % 64) != 0 ||
      d1.Cout != d2.Cin || d2.Cout < 8 || d1.OH != d2.H || d1.OW != d2.W || d1.N != d2.N || ep1.residual ||
      ep1.ps_r > 1 || (ep1.act == SRK_ACT_PRELU) || d1.KH * d1.KW > 1024 || d2.KH * d2.KW > 1024) {
    set_error("conv_fused2: configuration outside the fused kernel's envelope");
    return SRK_ERR_UNSUPPORTED;
  }
  Fused2Params F{};
  MfmaConvParams& P = F.P;
  P.in = nullptr; P.out = y; P.ep = ep2;
  P.N = d2.N; P.IH = d2.H; P.IW = d2.W; P.IC = d2.Cin; P.OH = d2.OH; P.OW = d2.OW; P.OC = d2.Cout;
  P.PH = d2.OH; P.PW = d2.OW; P.oy0 = 0; P.ox0 = 0; P.os = 1; P.iy0 = -d2.pad; P.ix0 = -d2.pad; P.is = 1;
  P.KHv = d2.KH; P.KWv = d2.KW; P.wh0 = 0; P.wdh = 1; P.ww0 = 0; P.wdw = 1; P.KW_full = d2.KW;
  F.x = x; F.x_nchw = x_nchw; F.bias1 = ep1.bias; F.act1 = ep1.act; F.slope1 = ep1.slope;
  F.IH0 = d1.H; F.IW0 = d1.W; F.IC1 = d1.Cin; F.C1 = d1.Cout; F.KH1 = d1.KH; F.KW1 = d1.KW; F.pad1 = d1.pad;
  F.KS1 = (d1.KW + 7) / 8; F.OCb1 = (d1.Cout + 63) / 64; F.H1 = d1.OH; F.W1 = d1.OW;
  const int NT = d2.Cout >= 64 ? 4 : (d2.Cout + 15) / 16;
  F.NB2 = NT * 16; F.ICc2 = d1.Cout / 32; F.OCb2 = (d2.Cout + 63) / 64;
  const size_t e1 = (size_t)d1.KH * d1.KW * d1.Cin * d1.Cout, e2 = (size_t)d2.KH * d2.KW * d2.Cin * d2.Cout;
  F.wq1 = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(wp1) + bf3_prepared_offset(e1));
  F.wq2 = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(wp2) + bf3_prepared_offset(e2));
  // tile: conv2 halo <= 384 pixels (6 conv1 M-tiles per wave); LDS = input halo + conv2 halo planes + 2 filter slots
  const int wslot = 8 * F.NB2 > 256 ? 8 * F.NB2 : 256;
  TilePick best{};
  bool found = false;
  for (int maxhalo = 64 * F2_MT1; maxhalo >= 64 && !found; maxhalo -= 64) {
    // pick_tile budgets in floats per halo pixel (32 = 128 B); cap the halo pixel count through the budget
    TilePick tp{};
    if (!pick_tile(256, P.PH, P.PW, 1, d2.KH, d2.KW, 32, maxhalo * 32, tp)) continue;
    const int hh0 = tp.HH + d1.KH - 1, hw0 = tp.HW + d1.KW - 1;
    const size_t npix0p = ((size_t)hh0 * hw0 + 16 + 15) & ~(size_t)15;
    const size_t npix2p = ((size_t)tp.HH * tp.HW + 15) & ~(size_t)15;
    const size_t lds = (npix0p + 8 * npix2p + 2 * (size_t)wslot) * 16;
    if (lds <= (size_t)kLdsBudgetBytes) {
      best = tp;
      found = true;
    }
  }
  if (!found) {
    set_error("conv_fused2: no tile fits LDS");
    return SRK_ERR_UNSUPPORTED;
  }
  P.TH = best.TH; P.TW = best.TW; P.tiles_y = best.tiles_y; P.tiles_x = best.tiles_x; P.HH = best.HH; P.HW = best.HW;
  F.HH0 = best.HH + d1.KH - 1; F.HW0 = best.HW + d1.KW - 1;
  F.NPIX0p = (F.HH0 * F.HW0 + 16 + 15) & ~15;
  F.NPIX2p = (best.HH * best.HW + 15) & ~15;
  size_t lds = ((size_t)F.NPIX0p + 8 * (size_t)F.NPIX2p + 2 * (size_t)wslot) * 16;
  const size_t epi_bytes = (size_t)4 * 32 * F2_EPI_STRIDE * sizeof(float);
  if (lds < epi_bytes) lds = epi_bytes;
  dim3 grid((unsigned)((size_t)P.tiles_x * P.tiles_y * P.N), F.OCb2);
  switch (NT) {
    case 1: f2_launch<1>(F, grid, lds, s); break;
    case 2: f2_launch<2>(F, grid, lds, s); break;
    case 3: f2_launch<3>(F, grid, lds, s); break;
    default: f2_launch<4>(F, grid, lds, s); break;
  }
  return check_launch("conv_fused2");
}

}  // namespace srk
