// Weight gradient of Conv2d / ConvTranspose2d on the CDNA4 matrix cores (exact fp32,
// v_mfma_f32_16x16x4_f32).
//
//   dW[tap][ci][co] = sum over pixels q of  X[.][ci] * dY[.][co]
//     Conv2d:          X at q*s - p + tap (shifted, "big"),  dY at q (anchor, "small")
//     ConvTranspose2d: X at q (anchor),  dY at q*s - p + tap (shifted)
//
// GEMM view per tap: M = ci, N = co, K = pixels (hundreds of thousands) -> split-K:
//   * persistent blocks (2 per CU) walk spatial tiles of <=64 anchor pixels; for each tile the
//     anchor tile and the shifted halo are staged in LDS once (coalesced 16-byte reads, zero
//     fill, ReLU/LeakyReLU gradient mask applied to dY on the fly);
//   * wave w owns the 16 input channels [16w,16w+16) of the block's 64-channel chunk and all
//     (<=9 taps) x (<=4 co tiles) 16x16 accumulators of the current tap group (<=144 VGPRs), so a
//     4-pixel K step costs <=13 ds_read_b32 for 36 MFMAs;
//   * accumulators live in registers across ALL tiles of the block; each block writes one
//     partial slab, a second kernel reduces the slabs in a fixed order (deterministic) into the
//     torch layout with beta-accumulate.
// Cin <= 4 (first layers) uses the flattened variant: M = (tap, ci) dense, so a 3x3x3 filter is
// 2 MFMA row tiles instead of 9 mostly-empty ones.
#include "srk_common.h"
#include "conv_problem.h"
#include <stdlib.h>

namespace srk {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WG_TP = 64;       // anchor pixels per tile
constexpr int WG_TP_SC = 192;   // ... of k_wgrad_mfma_smallcin (round 5: a 64-pixel tile was 7 us of barriers and LDS round trips for 16 MFMA quads)
constexpr int WG_TAPS = 9;      // taps per register pass
constexpr int WG_MAXBLOCKS = 512;

struct WgradParams {
  const float* x;
  const float* dy;
  const float* mask_y;
  float mask_slope;
  float* ws;  // [G][T][Cin][Cout]
  float* bias_partial;  // [G][Cout] column sums of dY (NULL: not requested / transposed)
  int N, Cin, Cout;
  int XH, XW, YH, YW;  // spatial dims of x and dy
  int KH, KW, stride, pad, transposed;
  int AH, AW;          // anchor spatial dims (dy for conv, x for transposed)
  int BH, BW;          // shifted spatial dims
  int TH, TW, tiles_y, tiles_x, HH, HW;
  int ntiles, G;
  int PSX, PSY;        // LDS pixel strides (floats) of the x / dy regions
  int xs_floats;       // size of the x region (dy region follows)
  int vec_x, vec_y;
};

__device__ __forceinline__ f32x4 mfma16w(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Stage rows [y0,y0+ny) x cols [x0,x0+nx) x channels [cb,cb+cc) of an NHWC tensor into
// lds[pixel][ps] (channels zero-padded to ccp, OOB pixels zero). `rows_total` >= ny*nx rows are
// written (extra rows zero) so K can be padded to a multiple of 4.
__device__ __forceinline__ void stage_region(const float* __restrict__ src, const float* __restrict__ mask,
                                             float mslope, float* lds, int ps, int n, int TH_, int TW_, int C, int y0,
                                             int x0, int ny, int nx, int rows_total, int cb, int cc, int ccp, int vec) {
  const int nvec = ccp >> 2;
  const int items = rows_total * nvec;
  for (int it = threadIdx.x; it < items; it += 256) {
    const int hp = it / nvec, q = it - hp * nvec;
    const int hy = hp / nx, hx = hp - hy * nx;
    const int iy = y0 + hy, ix = x0 + hx;
    const int ch = q * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (hy < ny && iy >= 0 && iy < TH_ && ix >= 0 && ix < TW_ && ch < cc) {
      const size_t off = (((size_t)n * TH_ + iy) * TW_ + ix) * C + cb + ch;
      if (vec && ch + 3 < cc) {
        v = *reinterpret_cast<const f32x4*>(src + off);
        if (mask) {
          const f32x4 m = *reinterpret_cast<const f32x4*>(mask + off);
          v.x = m.x > 0.f ? v.x : v.x * mslope;
          v.y = m.y > 0.f ? v.y : v.y * mslope;
          v.z = m.z > 0.f ? v.z : v.z * mslope;
          v.w = m.w > 0.f ? v.w : v.w * mslope;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (ch + e < cc) {
            float t = src[off + e];
            if (mask) t = mask[off + e] > 0.f ? t : t * mslope;
            v[e] = t;
          }
      }
    }
    *reinterpret_cast<f32x4*>(lds + (size_t)hp * ps + ch) = v;
  }
}

// ---------------------------------------------------------------------------------------------
// W1: general channel counts.  grid = (G, ci chunks of 64, co chunks of NTC*16).
// ---------------------------------------------------------------------------------------------
template <int NTC, bool TRANS>
__global__ __launch_bounds__(256, 2) void k_wgrad_mfma(WgradParams P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;
  float* ys = smem + P.xs_floats;
  __shared__ int hoff[WG_TP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int cib = blockIdx.y * 64;
  const int cob = blockIdx.z * (NTC * 16);
  const int cic = (P.Cin - cib) < 64 ? (P.Cin - cib) : 64;
  const int cicp = (cic + 15) & ~15;
  const int coc = (P.Cout - cob) < NTC * 16 ? (P.Cout - cob) : NTC * 16;
  const int cocp = NTC * 16;
  const int T = P.KH * P.KW;
  const int npx = P.TH * P.TW;
  const int npx4 = (npx + 3) & ~3;
  const bool wave_live = wave * 16 < cicp;
  const bool want_bias = P.bias_partial != nullptr && blockIdx.y == 0;
  float bsum = 0.f;

  for (int p = tid; p < WG_TP; p += 256) {
    int h = 0;
    if (p < npx) {
      const int r = p / P.TW, c = p - r * P.TW;
      h = (r * P.stride) * P.HW + c * P.stride;
    }
    hoff[p] = h;
  }

  for (int t0 = 0; t0 < T; t0 += WG_TAPS) {
    const int tg = (T - t0) < WG_TAPS ? (T - t0) : WG_TAPS;
    int toff[WG_TAPS];
#pragma unroll
    for (int t = 0; t < WG_TAPS; ++t) {
      const int tt = (t < tg) ? (t0 + t) : t0;
      const int u = tt / P.KW, v = tt - u * P.KW;
      toff[t] = (u * P.HW + v) * (TRANS ? P.PSY : P.PSX);
    }
    f32x4 acc[WG_TAPS][NTC];
#pragma unroll
    for (int t = 0; t < WG_TAPS; ++t)
#pragma unroll
      for (int nt = 0; nt < NTC; ++nt) acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int tile = blockIdx.x; tile < P.ntiles; tile += P.G) {
      int b = tile;
      const int txi = b % P.tiles_x;
      b /= P.tiles_x;
      const int tyi = b % P.tiles_y;
      const int n = b / P.tiles_y;
      const int r0 = tyi * P.TH, c0 = txi * P.TW;
      // anchor rows actually inside the tensor for this tile (ragged last tiles)
      __syncthreads();  // previous tile fully consumed (and hoff visible on the first pass)
      const int by0 = r0 * P.stride - P.pad, bx0 = c0 * P.stride - P.pad;
      if (!TRANS) {
        stage_region(P.x, nullptr, 0.f, xs, P.PSX, n, P.XH, P.XW, P.Cin, by0, bx0, P.HH, P.HW, P.HH * P.HW, cib, cic,
                     cicp, P.vec_x);
        stage_region(P.dy, P.mask_y, P.mask_slope, ys, P.PSY, n, P.YH, P.YW, P.Cout, r0, c0, P.TH, P.TW, npx4, cob,
                     coc, cocp, P.vec_y);
      } else {
        stage_region(P.x, nullptr, 0.f, xs, P.PSX, n, P.XH, P.XW, P.Cin, r0, c0, P.TH, P.TW, npx4, cib, cic, cicp,
                     P.vec_x);
        stage_region(P.dy, P.mask_y, P.mask_slope, ys, P.PSY, n, P.YH, P.YW, P.Cout, by0, bx0, P.HH, P.HW,
                     P.HH * P.HW, cob, coc, cocp, P.vec_y);
      }
      __syncthreads();
      if (!TRANS && want_bias && t0 == 0 && tid < cocp) {
        // db partial: column sums of the LDS-resident (masked) dY tile; zero rows pad the tile
        float bs0 = 0.f, bs1 = 0.f;
        for (int p = 0; p + 1 < npx4; p += 2) {
          bs0 += ys[p * P.PSY + tid];
          bs1 += ys[(p + 1) * P.PSY + tid];
        }
        bsum += bs0 + bs1;
      }
      if (wave_live) {
        for (int k4 = 0; k4 < npx4; k4 += 4) {
          const int p = k4 + kq;
          const int ho = hoff[p];
          if (!TRANS) {
            const float* ap = xs + ho * P.PSX + wave * 16 + i;
            const float* bp = ys + p * P.PSY + i;
            float bq[NTC];
#pragma unroll
            for (int nt = 0; nt < NTC; ++nt) bq[nt] = bp[nt * 16];
#pragma unroll
            for (int t = 0; t < WG_TAPS; ++t) {
              if (t < tg) {
                const float a = ap[toff[t]];
#pragma unroll
                for (int nt = 0; nt < NTC; ++nt) acc[t][nt] = mfma16w(a, bq[nt], acc[t][nt]);
              }
            }
          } else {
            const float a = xs[p * P.PSX + wave * 16 + i];
            const float* bp = ys + ho * P.PSY + i;
#pragma unroll
            for (int t = 0; t < WG_TAPS; ++t) {
              if (t < tg) {
#pragma unroll
                for (int nt = 0; nt < NTC; ++nt) acc[t][nt] = mfma16w(a, bp[toff[t] + nt * 16], acc[t][nt]);
              }
            }
          }
        }
      }
    }
    if (!TRANS && want_bias && t0 == 0 && tid < cocp && cob + tid < P.Cout)
      P.bias_partial[(size_t)blockIdx.x * P.Cout + cob + tid] = bsum;
    // partial slab: ws[g][t][ci][co]; C/D layout col = lane&15 (co), row = (lane>>4)*4 + reg (ci)
    if (wave_live) {
      float* slab = P.ws + (size_t)blockIdx.x * T * P.Cin * P.Cout;
#pragma unroll
      for (int t = 0; t < WG_TAPS; ++t) {
        if (t < tg) {
#pragma unroll
          for (int nt = 0; nt < NTC; ++nt) {
            const int co = cob + nt * 16 + i;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
              const int ci = cib + wave * 16 + kq * 4 + reg;
              if (ci < P.Cin && co < P.Cout) slab[((size_t)(t0 + t) * P.Cin + ci) * P.Cout + co] = acc[t][nt][reg];
            }
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// W2: Cin <= 4 (Conv2d only).  M = (tap, ci) flattened dense: m = tap*Cin + ci, MT row tiles.
// Wave w owns output-channel tile w of a 64-channel chunk and all MT row tiles.
// LDS: x halo [pixel][4], dy tile [pixel][PSY].
// ---------------------------------------------------------------------------------------------
// stage_region in two halves for a register prefetch: item `it` of the region (what stage_region's thread loop loads in
// iteration it / 256) as a value, and its LDS store
__device__ __forceinline__ f32x4 stage_item_load(const float* __restrict__ src, const float* __restrict__ mask, float mslope,
                                                 int n, int TH_, int TW_, int C, int y0, int x0, int ny, int nx,
                                                 int rows_total, int cb, int cc, int ccp, int vec, int it) {
  const int nvec = ccp >> 2;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (it >= rows_total * nvec) return v;
  const int hp = it / nvec, q = it - hp * nvec;
  const int hy = hp / nx, hx = hp - hy * nx;
  const int iy = y0 + hy, ix = x0 + hx;
  const int ch = q * 4;
  if (hy < ny && iy >= 0 && iy < TH_ && ix >= 0 && ix < TW_ && ch < cc) {
    const size_t off = (((size_t)n * TH_ + iy) * TW_ + ix) * C + cb + ch;
    if (vec && ch + 3 < cc) {
      v = *reinterpret_cast<const f32x4*>(src + off);
      if (mask) {
        const f32x4 m = *reinterpret_cast<const f32x4*>(mask + off);
        v.x = m.x > 0.f ? v.x : v.x * mslope;
        v.y = m.y > 0.f ? v.y : v.y * mslope;
        v.z = m.z > 0.f ? v.z : v.z * mslope;
        v.w = m.w > 0.f ? v.w : v.w * mslope;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (ch + e < cc) {
          float t = src[off + e];
          if (mask) t = mask[off + e] > 0.f ? t : t * mslope;
          v[e] = t;
        }
    }
  }
  return v;
}
__device__ __forceinline__ void stage_item_store(float* lds, int ps, int rows_total, int ccp, int it, const f32x4& v) {
  const int nvec = ccp >> 2;
  if (it >= rows_total * nvec) return;
  const int hp = it / nvec, q = it - hp * nvec;
  *reinterpret_cast<f32x4*>(lds + (size_t)hp * ps + q * 4) = v;
}

#ifndef SCIN_UNROLL
#define SCIN_UNROLL 1
#endif
template <int MT>
__global__ __launch_bounds__(256, 2) void k_wgrad_mfma_smallcin(WgradParams P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;
  float* ys = smem + P.xs_floats;
  __shared__ int hoff[WG_TP_SC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int cob = blockIdx.z * 64;
  const int coc = (P.Cout - cob) < 64 ? (P.Cout - cob) : 64;
  const int T = P.KH * P.KW;
  const int M = T * P.Cin;
  const int npx = P.TH * P.TW;
  const int npx4 = (npx + 3) & ~3;
  const bool wave_live = wave * 16 < coc;

  for (int p = tid; p < WG_TP_SC; p += 256) {
    int h = 0;
    if (p < npx) {
      const int r = p / P.TW, c = p - r * P.TW;
      h = ((r * P.stride) * P.HW + c * P.stride) * 4;
    }
    hoff[p] = h;
  }
  // per-lane gather offset of row m = mt*16 + i inside the halo: tap shift * 4 + ci
  int moff[MT];
  bool mval[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mt * 16 + i;
    mval[mt] = m < M;
    const int mm = mval[mt] ? m : 0;
    const int t = mm / P.Cin, ci = mm - t * P.Cin;
    const int u = t / P.KW, v = t - u * P.KW;
    moff[mt] = (u * P.HW + v) * 4 + ci;
  }
  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;

  // Register prefetch of the next tile (round 5): a tile is 64 anchor pixels -- 16 KB of dY and a few hundred halo values --
  // and the block used to load it, wait, multiply 16 quads and start over: a latency chain per tile with two blocks per CU
  // to hide it (VDSR's first layer: 6724 tiles, 103 us for 115 MB).  Now tile t + 1's loads are issued in front of tile t's
  // MFMA loop and stored to LDS behind it.  Regions whose items exceed the register batches (XIT / YIT per thread) keep
  // the staged form.
  constexpr bool PFOK = MT <= 4;   // (the 9x9 problems hold 64 accumulator registers: no room for a 48-register batch at two blocks per CU)
  constexpr int XIT = PFOK ? 2 : 1, YIT = PFOK ? WG_TP_SC * 16 / 256 : 1;
  const bool pf = PFOK && P.HH * P.HW <= XIT * 256 && npx4 * 16 <= YIT * 256;
  f32x4 xr[XIT], yr[YIT];
  auto tile_rc = [&](int tile, int& n, int& r0, int& c0) {
    int b = tile;
    const int txi = b % P.tiles_x;
    b /= P.tiles_x;
    const int tyi = b % P.tiles_y;
    n = b / P.tiles_y;
    r0 = tyi * P.TH;
    c0 = txi * P.TW;
  };
  auto prefetch = [&](int tile) {
    int n, r0, c0;
    tile_rc(tile, n, r0, c0);
#pragma unroll
    for (int k = 0; k < XIT; ++k)
      xr[k] = stage_item_load(P.x, nullptr, 0.f, n, P.XH, P.XW, P.Cin, r0 * P.stride - P.pad, c0 * P.stride - P.pad, P.HH, P.HW,
                              P.HH * P.HW, 0, P.Cin, 4, P.vec_x, tid + 256 * k);
#pragma unroll
    for (int k = 0; k < YIT; ++k)
      yr[k] = stage_item_load(P.dy, P.mask_y, P.mask_slope, n, P.YH, P.YW, P.Cout, r0, c0, P.TH, P.TW, npx4, cob, coc, 64,
                              P.vec_y, tid + 256 * k);
  };
  if (pf && (int)blockIdx.x < P.ntiles) prefetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < P.ntiles; tile += P.G) {
    int n, r0, c0;
    tile_rc(tile, n, r0, c0);
    __syncthreads();
    if (pf) {
#pragma unroll
      for (int k = 0; k < XIT; ++k) stage_item_store(xs, 4, P.HH * P.HW, 4, tid + 256 * k, xr[k]);
#pragma unroll
      for (int k = 0; k < YIT; ++k) stage_item_store(ys, P.PSY, npx4, 64, tid + 256 * k, yr[k]);
    } else {
      stage_region(P.x, nullptr, 0.f, xs, 4, n, P.XH, P.XW, P.Cin, r0 * P.stride - P.pad, c0 * P.stride - P.pad, P.HH,
                   P.HW, P.HH * P.HW, 0, P.Cin, 4, P.vec_x);
      stage_region(P.dy, P.mask_y, P.mask_slope, ys, P.PSY, n, P.YH, P.YW, P.Cout, r0, c0, P.TH, P.TW, npx4, cob, coc,
                   64, P.vec_y);
    }
    __syncthreads();
    if (pf && tile + P.G < P.ntiles) prefetch(tile + P.G);
    if (P.bias_partial) {   // column sums of dY: wave w takes the pixels w, w + 4, ... (combined once, behind the last tile)
      float bs0 = 0.f, bs1 = 0.f;
      for (int p = wave; p + 4 < npx4; p += 8) {
        bs0 += ys[p * P.PSY + lane];
        bs1 += ys[(p + 4) * P.PSY + lane];
      }
      if (((npx4 - 1 - wave) >> 2) % 2 == 0 && wave < npx4) bs0 += ys[(wave + ((npx4 - 1 - wave) >> 2) * 4) * P.PSY + lane];
      bsum += bs0 + bs1;
    }
    if (wave_live) {
      // (SCIN_UNROLL pixel quads per trip: their LDS reads -- 1 + MT per quad, each behind a table lookup -- are independent and
      //  issue together; one quad per trip left every MFMA behind a full LDS round trip)
#pragma unroll SCIN_UNROLL
      for (int k4 = 0; k4 < npx4; k4 += 4) {
        const int p = k4 + kq;
        const float bv = ys[p * P.PSY + wave * 16 + i];
        const float* ap = xs + hoff[p];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const float a = mval[mt] ? ap[moff[mt]] : 0.f;
          acc[mt] = mfma16w(a, bv, acc[mt]);
        }
      }
    }
  }
  if (P.bias_partial) {
    __syncthreads();
    float* bred = smem;   // [4 waves][64]
    bred[wave * 64 + lane] = bsum;
    __syncthreads();
    if (tid < 64 && cob + tid < P.Cout)
      P.bias_partial[(size_t)blockIdx.x * P.Cout + cob + tid] = (bred[tid] + bred[64 + tid]) + (bred[128 + tid] + bred[192 + tid]);
  }
  if (wave_live) {
    float* slab = P.ws + (size_t)blockIdx.x * T * P.Cin * P.Cout;
    const int co = cob + wave * 16 + i;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int m = mt * 16 + kq * 4 + reg;  // = tap*Cin + ci -> ws index (tap*Cin + ci)*Cout + co
        if (m < M && co < P.Cout) slab[(size_t)m * P.Cout + co] = acc[mt][reg];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Deterministic slab reduction + layout change:  dw(torch layout) = beta*dw + sum_g ws[g][t][ci][co]
// ---------------------------------------------------------------------------------------------
// `blk`: block index inside this reduction (blockIdx.x of a launch of its own).  WIDE: the summation order of
// k_wgrad_reduce (4 accumulators, slab stride 16); !WIDE: that of the grouped reduce of conv_wgrad_bf16.hip (2
// accumulators, stride 8) -- k_wgrad_reduce_multi reproduces either bit for bit.
template <bool WIDE>
__device__ __forceinline__ void wgrad_reduce_body(float (&sm)[4][64], int blk, const float* __restrict__ ws,
                                                  float* __restrict__ dw, int G, int Cout, int Cin, int KH, int KW,
                                                  int transposed, float beta, const float* __restrict__ bias_partial,
                                                  float* __restrict__ db, int bias_cout, int out_ps_r) {
  // (round 5: 256 elements per block with one 16-byte load per lane -- what the grouped reduce of conv_wgrad_bf16.hip now does:
  //  14.2 -> 11.3 us over 21 layers -- was measured SLOWER here: a single layer's reduce is a few hundred blocks at most, and
  //  four times fewer of them leave CUs idle: VDSR layer 7.0 -> 9.1 us, SRGAN-D 512 -> 512 layer 47 -> 56 us per call.)
  // 64 consecutive slab elements per block (coalesced 256-byte rows); the 4 waves each sum a quarter
  // of the G slabs with 4 independent accumulators (16 loads in flight per lane), combined through
  // LDS in a fixed order => deterministic.  The blocks past the last slab element finish the bias
  // gradient the same way (bias_partial[G][bias_cout] -> db), saving a launch per layer.
  const int elems = KH * KW * Cin * Cout;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nwb = (elems + 63) / 64;
  if (blk >= nwb) {
    if (!db || !bias_partial) return;
    const int co = (blk - nwb) * 64 + lane;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    if (co < bias_cout) {
      int g = w;
      if (WIDE) {
        for (; g + 12 < G; g += 16) {
          t0 += bias_partial[(size_t)g * bias_cout + co];
          t1 += bias_partial[(size_t)(g + 4) * bias_cout + co];
          t2 += bias_partial[(size_t)(g + 8) * bias_cout + co];
          t3 += bias_partial[(size_t)(g + 12) * bias_cout + co];
        }
      } else {
        for (; g + 4 < G; g += 8) {
          t0 += bias_partial[(size_t)g * bias_cout + co];
          t1 += bias_partial[(size_t)(g + 4) * bias_cout + co];
        }
      }
      for (; g < G; g += 4) t0 += bias_partial[(size_t)g * bias_cout + co];
    }
    sm[w][lane] = WIDE ? (t0 + t1) + (t2 + t3) : t0 + t1;
    __syncthreads();
    if (w == 0 && co < bias_cout) {
      const float t = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
      int cot = co;
      if (out_ps_r > 1) {  // slab channel order (i, j, c) -> torch channel c*r*r + i*r + j
        const int C = bias_cout / (out_ps_r * out_ps_r);
        const int q = co / C, c = co - q * C;
        cot = c * out_ps_r * out_ps_r + q;
      }
      db[cot] = beta != 0.f ? beta * db[cot] + t : t;
    }
    return;
  }
  const int e = blk * 64 + lane;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < elems) {
    int g = w;
    if (WIDE) {
      for (; g + 12 < G; g += 16) {
        s0 += ws[(size_t)g * elems + e];
        s1 += ws[(size_t)(g + 4) * elems + e];
        s2 += ws[(size_t)(g + 8) * elems + e];
        s3 += ws[(size_t)(g + 12) * elems + e];
      }
    } else {
      for (; g + 4 < G; g += 8) {
        s0 += ws[(size_t)g * elems + e];
        s1 += ws[(size_t)(g + 4) * elems + e];
      }
    }
    for (; g < G; g += 4) s0 += ws[(size_t)g * elems + e];
  }
  sm[w][lane] = WIDE ? (s0 + s1) + (s2 + s3) : s0 + s1;
  __syncthreads();
  if (w != 0 || e >= elems) return;
  const float v = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
  int co = e % Cout;
  if (out_ps_r > 1) {
    const int C = Cout / (out_ps_r * out_ps_r);
    const int q = co / C, c = co - q * C;
    co = c * out_ps_r * out_ps_r + q;
  }
  const int ci = (e / Cout) % Cin;
  const int tap = e / (Cout * Cin);
  const int kh = tap / KW, kw = tap - kh * KW;
  // transposed == 2: role-swapped small-Cout problem (see swap_small_cout): ConvTranspose-style index with the
  // taps mirrored
  const size_t o = transposed == 2 ? ((((size_t)ci * Cout + co) * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw))
                   : transposed    ? ((((size_t)ci * Cout + co) * KH + kh) * KW + kw)
                                   : ((((size_t)co * Cin + ci) * KH + kh) * KW + kw);
  dw[o] = beta != 0.f ? beta * dw[o] + v : v;
}

__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ ws, float* __restrict__ dw, int G,
                                                      int Cout, int Cin, int KH, int KW, int transposed, float beta,
                                                      const float* __restrict__ bias_partial, float* __restrict__ db,
                                                      int bias_cout, int out_ps_r) {
  __shared__ float sm[4][64];
  wgrad_reduce_body<true>(sm, (int)blockIdx.x, ws, dw, G, Cout, Cin, KH, KW, transposed, beta, bias_partial, db, bias_cout,
                          out_ps_r);
}

// ---- deferred reductions: the slab reductions of SEVERAL weight-gradient launches in one launch ----------------------
// A strong-scaled shard or an SRGAN step ends its backward pass with a handful to dozens of weight-gradient launches, each
// followed by a reduce launch of a few microseconds of work (EDSR, 16 patches: 6 reduce launches = 60 us of a 1.2 ms step;
// SRGAN: 32 per step).  Between srk_wgrad_reduce_defer(1) and srk_wgrad_reduce_flush() the reduce of every weight-gradient
// call of the calling thread is queued instead (the caller keeps the workspaces alive and distinct), and the flush runs
// them all as ONE launch: block -> (job, block inside the job) by a scan over <= 40 jobs passed by value (graph-capturable),
// each job summed in exactly the order of the kernel it replaces.
struct RedJob {
  const float* ws;
  float* dw;
  const float* bias_partial;
  float* db;
  int G, blk0;            // slabs; first block of this job in the merged grid
  short Cout, Cin, bias_cout;
  unsigned char KH, KW, transposed, out_ps_r, wide, pad_;
  float beta;
};
constexpr int kMaxRedJobs = 40;
struct RedJobs {
  RedJob j[kMaxRedJobs];
  int n, blocks;
};

__global__ __launch_bounds__(256) void k_wgrad_reduce_multi(RedJobs J) {
  __shared__ float sm[4][64];
  int k = 0;
  for (int i = 1; i < J.n; ++i)
    if ((int)blockIdx.x >= J.j[i].blk0) k = i;
  const RedJob& r = J.j[k];
  const int blk = (int)blockIdx.x - r.blk0;
  if (r.wide)
    wgrad_reduce_body<true>(sm, blk, r.ws, r.dw, r.G, r.Cout, r.Cin, r.KH, r.KW, r.transposed, r.beta, r.bias_partial, r.db,
                            r.bias_cout, r.out_ps_r);
  else
    wgrad_reduce_body<false>(sm, blk, r.ws, r.dw, r.G, r.Cout, r.Cin, r.KH, r.KW, r.transposed, r.beta, r.bias_partial, r.db,
                             r.bias_cout, r.out_ps_r);
}

static thread_local RedJobs g_red{};
static thread_local int g_red_defer = 0;

bool wgrad_reduce_deferring() { return g_red_defer > 0; }

// Every weight-gradient entry point calls this first: a queued reduction that updates `dw` / `db` must land before
// anything else touches those tensors (a shared weight's second use may run a kernel that accumulates in place, without
// a reduction of its own -- the order of the two updates is part of the result), and big partial slabs are not worth
// keeping: read back right behind their kernel they come from the Infinity Cache, at the end of a pass from HBM (SRGAN's
// discriminator: 9 merged launches of 64 us against 32 of 14 us).
constexpr size_t kRedPendingBytes = (size_t)24 << 20;
static thread_local size_t g_red_bytes = 0;
int wgrad_reduce_flush(hipStream_t s) {
  if (g_red.n == 0) return SRK_OK;
  hipLaunchKernelGGL(k_wgrad_reduce_multi, dim3((unsigned)g_red.blocks), dim3(256), 0, s, g_red);
  g_red.n = 0;
  g_red.blocks = 0;
  g_red_bytes = 0;
  return check_launch("wgrad_reduce_multi");
}

int wgrad_reduce_before_update(const float* dw, const float* db, hipStream_t s) {
  if (g_red.n == 0) return SRK_OK;
  bool clash = g_red_bytes > kRedPendingBytes;
  for (int i = 0; i < g_red.n && !clash; ++i) clash = g_red.j[i].dw == dw || (db && g_red.j[i].db == db);
  return clash ? wgrad_reduce_flush(s) : SRK_OK;
}


// queue one reduction (or run it now when nothing is being deferred / it does not fit a job record)
int wgrad_reduce_submit(bool wide, const float* ws, float* dw, int G, int Cout, int Cin, int KH, int KW, int transposed,
                        float beta, const float* bias_partial, float* db, int bias_cout, int out_ps_r, hipStream_t s) {
  const int elems = KH * KW * Cin * Cout;
  const int bias_blocks = (bias_partial && db) ? cdiv(bias_cout, 64) : 0;
  // (a reduction over more than a few MB of slabs runs right behind its kernel, while the slabs are still cached)
  const bool fits = Cout < 32768 && Cin < 32768 && bias_cout < 32768 && KH < 256 && KW < 256 &&
                    (size_t)G * elems * sizeof(float) <= ((size_t)6 << 20);
  if (!g_red_defer || !fits) {
    if (!wide) return -100;   // (the grouped caller launches its own kernel)
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(cdiv(elems, 64) + bias_blocks), dim3(256), 0, s, ws, dw, G, Cout, Cin, KH, KW,
                       transposed, beta, bias_partial, db, bias_cout, out_ps_r);
    return check_launch("conv_wgrad_reduce");
  }
  // two queued jobs must not update the same tensor (beta = 1 accumulates: shared weights), nor may the queue overflow
  bool clash = g_red.n >= kMaxRedJobs;
  for (int i = 0; i < g_red.n && !clash; ++i) clash = g_red.j[i].dw == dw || (db && g_red.j[i].db == db);
  if (clash) {
    const int rc = wgrad_reduce_flush(s);
    if (rc) return rc;
  }
  RedJob& r = g_red.j[g_red.n++];
  r.ws = ws; r.dw = dw; r.bias_partial = bias_blocks ? bias_partial : nullptr; r.db = bias_blocks ? db : nullptr;
  r.G = G; r.blk0 = g_red.blocks;
  r.Cout = (short)Cout; r.Cin = (short)Cin; r.bias_cout = (short)bias_cout;
  r.KH = (unsigned char)KH; r.KW = (unsigned char)KW; r.transposed = (unsigned char)transposed;
  r.out_ps_r = (unsigned char)out_ps_r; r.wide = wide ? 1 : 0; r.pad_ = 0;
  r.beta = beta;
  g_red.blocks += cdiv(elems, 64) + bias_blocks;
  g_red_bytes += (size_t)G * elems * sizeof(float);
  return SRK_OK;
}

// bias_partial / db may be NULL (no bias, or the bias gradient is produced elsewhere); bias_cout = channels of db
int conv_wgrad_reduce_launch(const float* ws, float* dw, int G, int Cout, int Cin, int KH, int KW, int transposed,
                             float beta, const float* bias_partial, float* db, int bias_cout, int out_ps_r,
                             hipStream_t s) {
  return wgrad_reduce_submit(true, ws, dw, G, Cout, Cin, KH, KW, transposed, beta, bias_partial, db, bias_cout, out_ps_r, s);
}

// ---------------------------------------------------------------------------------------------
// Host
// ---------------------------------------------------------------------------------------------
static constexpr int kWgLdsBudget = 78 * 1024;

struct WgPlan {
  bool ok;
  bool smallcin;
  int NTC;
  int TH, TW, tiles_y, tiles_x, HH, HW;
  int PSX, PSY, xs_floats;
  size_t lds;
  int G, ntiles;
};

static WgPlan plan(const srk_conv_desc& d) {
  WgPlan pl{};
  pl.ok = false;
  const int T = d.KH * d.KW;
  if (T > 32 * 32 || d.Cout < 1) return pl;
  pl.smallcin = (d.Cin <= 4) && !d.transposed && (T * d.Cin <= 256);
  if (!pl.smallcin && d.Cin <= 4) return pl;  // transposed / huge kernels with tiny Cin: generic kernel
  const int AH = d.transposed ? d.H : d.OH, AW = d.transposed ? d.W : d.OW;
  int ntc = (d.Cout + 15) / 16;
  if (ntc > 4) ntc = 4;
  pl.NTC = pl.smallcin ? 4 : ntc;
  // LDS pixel strides: lanes of one half-wave (kq = 0,1) must land on disjoint banks -> stride = 16 (mod 32)
  const int cip = pl.smallcin ? 4 : (d.Cin >= 64 ? 64 : ((d.Cin + 15) & ~15));
  const int psx_full = pl.smallcin ? 4 : (((cip % 32) == 16) ? cip : cip + 16);
  const int cop = pl.NTC * 16;
  const int psy_full = ((cop % 32) == 16) ? cop : cop + 16;
  // conv: x is the halo, dy the tile; transposed: the other way round
  const int ps_halo = d.transposed ? psy_full : psx_full;
  const int ps_tile = d.transposed ? psx_full : psy_full;
  bool found = false;
  long best_tiles = 0, best_halo = 0;
  // smallcin: 192-pixel tiles when the problem still has >= 2 of them per block slot (VDSR's first layer, SRGAN-D's at
  // 128x128), 64-pixel tiles otherwise (SRGAN-G's 9x9 first layer at 32x32: 86 large tiles would leave two thirds of the CUs idle)
  int tp = WG_TP;
  if (pl.smallcin && (long)d.N * cdiv(AH * AW, WG_TP_SC) >= 2L * WG_MAXBLOCKS && T * d.Cin <= 64) tp = WG_TP_SC;
  for (int TW = 1; TW <= (AW < tp ? AW : tp); ++TW) {
    int TH = tp / TW;
    if (TH > AH) TH = AH;
    for (; TH >= 1; --TH) {
      const int HH = (TH - 1) * d.stride + d.KH, HW = (TW - 1) * d.stride + d.KW;
      const int tile_rows = (TH * TW + 3) & ~3;
      const long floats = (long)HH * HW * ps_halo + (long)tile_rows * ps_tile;
      if (floats * 4 <= kWgLdsBudget) {
        const long tiles = (long)cdiv(AH, TH) * cdiv(AW, TW);
        const long halo = (long)HH * HW * tiles;
        if (!found || tiles < best_tiles || (tiles == best_tiles && halo < best_halo)) {
          found = true;
          best_tiles = tiles;
          best_halo = halo;
          pl.TH = TH; pl.TW = TW; pl.tiles_y = cdiv(AH, TH); pl.tiles_x = cdiv(AW, TW); pl.HH = HH; pl.HW = HW;
        }
        break;
      }
    }
  }
  if (!found) return pl;
  pl.PSX = psx_full;
  pl.PSY = psy_full;
  const int tile_rows = (pl.TH * pl.TW + 3) & ~3;
  pl.xs_floats = d.transposed ? tile_rows * psx_full : pl.HH * pl.HW * psx_full;
  const int ys_floats = d.transposed ? pl.HH * pl.HW * psy_full : tile_rows * psy_full;
  pl.lds = ((size_t)pl.xs_floats + ys_floats) * 4;
  const long nt = (long)d.N * pl.tiles_y * pl.tiles_x;
  if (nt > (1L << 30)) return pl;
  pl.ntiles = (int)nt;
  pl.G = pl.ntiles < WG_MAXBLOCKS ? pl.ntiles : WG_MAXBLOCKS;
  {
    // deep, small layers (SRGAN discriminator: 512 -> 512 at 6x6): every split-K slab is a full filter (9.4 MB), and
    // one slab per tile makes the slabs -- written here, read back by the reduce -- the dominant traffic.  Cap the
    // slab count at what fills the GPU (2 blocks per CU) together with the channel blocks of the grid: SRGAN adversarial
    // step 16.49 -> 15.78 ms (0 = one slab per tile: 16.49, 1: 16.12, 4: 15.99).
    const int per_cu = env_int("SRK_WG_MFMA_BLOCKS", 2);
    if (per_cu > 0 && !pl.smallcin) {
      const int chan_blocks = cdiv(d.Cin, 64) * cdiv(d.Cout, pl.NTC * 16);
      int g = (per_cu * kNumCU + chan_blocks - 1) / chan_blocks;
      if (g < 1) g = 1;
      if (pl.G > g) pl.G = g;
    }
  }
  pl.ok = true;
  return pl;
}

// Small Cout (the 64->3 output layers: EDSR / VDSR / SRCNN / LapSRN tails, SRGAN's 9x9 64->3), stride 1:
//   dW[tap][ci][co] = sum_q x[q + tap - p][ci] dy[q][co] = sum_q' dy[q' + (K-1-tap) - (K-1-p)][co] x[q'][ci]
// i.e. the weight gradient of a convolution with input dy (Cout channels), output-gradient x (Cin channels), padding
// K-1-p and mirrored taps.  That problem has a tiny channel count on its *input* side, which is exactly what the
// flattened (tap, ci) kernel k_wgrad_mfma_smallcin is built for (one pass over the pixels instead of one pass per
// group of 9 taps with a single 16-wide channel tile 3/16 full).  The reduce kernel undoes the swap (mode 2).
static bool swap_small_cout(const srk_conv_desc& d, srk_conv_desc& ds) {
  if (d.transposed || d.stride != 1 || d.Cout > 4 || d.Cin < 8 || d.KH != d.KW) return false;
  if (d.KH * d.KW * d.Cout > 256 || d.KH - 1 - d.pad < 0) return false;
  ds = d;
  ds.H = d.OH; ds.W = d.OW; ds.Cin = d.Cout;
  ds.OH = d.H; ds.OW = d.W; ds.Cout = d.Cin;
  ds.pad = d.KH - 1 - d.pad;
  return true;
}

bool conv_wgrad_mfma_supported(const srk_conv_desc& d) { return plan(d).ok; }

size_t conv_wgrad_mfma_ws(const srk_conv_desc& d) {
  WgPlan pl = plan(d);
  if (!pl.ok) return 0;
  size_t need = (size_t)pl.G * d.KH * d.KW * d.Cin * d.Cout * sizeof(float) + conv_bias_grad_ws(d);
  srk_conv_desc ds;
  if (swap_small_cout(d, ds)) {
    WgPlan ps = plan(ds);
    if (ps.ok) {
      const size_t n2 = (size_t)ps.G * d.KH * d.KW * d.Cin * d.Cout * sizeof(float) + conv_bias_grad_ws(d);
      if (n2 > need) need = n2;
    }
  }
  return need;
}

template <typename K>
static void wg_set_lds(K kern, LdsLimit& cur, size_t lds) {
  cur.ensure(reinterpret_cast<const void*>(kern), lds);
}

template <int NTC, bool TRANS>
static void launch_w1(const WgradParams& P, dim3 grid, size_t lds, hipStream_t s) {
  static LdsLimit cur;
  wg_set_lds(&k_wgrad_mfma<NTC, TRANS>, cur, lds);
  hipLaunchKernelGGL((k_wgrad_mfma<NTC, TRANS>), grid, dim3(256), lds, s, P);
}
template <int MT>
static void launch_w2(const WgradParams& P, dim3 grid, size_t lds, hipStream_t s) {
  static LdsLimit cur;
  wg_set_lds(&k_wgrad_mfma_smallcin<MT>, cur, lds);
  hipLaunchKernelGGL((k_wgrad_mfma_smallcin<MT>), grid, dim3(256), lds, s, P);
}

int conv_wgrad_mfma(const srk_conv_desc& d, const float* x, const float* dy, const srk_bwd_mask* mask, float* dw,
                    float* db, float beta, void* ws, size_t ws_bytes, hipStream_t s);

// role-swapped launch for small Cout (no activation mask on these layers)
static int conv_wgrad_small_cout(const srk_conv_desc& d, const srk_conv_desc& ds, const float* x, const float* dy,
                                 float* dw, float* db, float beta, void* ws, size_t ws_bytes, hipStream_t s) {
  WgPlan pl = plan(ds);
  const size_t slab_bytes = (size_t)pl.G * d.KH * d.KW * d.Cin * d.Cout * sizeof(float);
  const size_t need = slab_bytes + conv_bias_grad_ws(d);
  if (!ws || ws_bytes < need) {
    set_error("conv_wgrad_mfma: workspace %zu < %zu", ws_bytes, need);
    return SRK_ERR_WORKSPACE;
  }
  WgradParams P{};
  P.x = dy; P.dy = x; P.mask_y = nullptr; P.mask_slope = 0.f;
  P.ws = (float*)ws;
  P.bias_partial = nullptr;
  P.N = ds.N; P.Cin = ds.Cin; P.Cout = ds.Cout;
  P.XH = ds.H; P.XW = ds.W; P.YH = ds.OH; P.YW = ds.OW;
  P.KH = ds.KH; P.KW = ds.KW; P.stride = 1; P.pad = ds.pad; P.transposed = 0;
  P.AH = ds.OH; P.AW = ds.OW; P.BH = ds.H; P.BW = ds.W;
  P.TH = pl.TH; P.TW = pl.TW; P.tiles_y = pl.tiles_y; P.tiles_x = pl.tiles_x; P.HH = pl.HH; P.HW = pl.HW;
  P.ntiles = pl.ntiles; P.G = pl.G; P.PSX = pl.PSX; P.PSY = pl.PSY; P.xs_floats = pl.xs_floats;
  P.vec_x = (ds.Cin % 4 == 0) && ((uintptr_t)dy % 16 == 0);
  P.vec_y = (ds.Cout % 4 == 0) && ((uintptr_t)x % 16 == 0);
  const int MT = (ds.KH * ds.KW * ds.Cin + 15) / 16;
  dim3 grid(pl.G, 1, cdiv(ds.Cout, 64));
  if (MT <= 2) launch_w2<2>(P, grid, pl.lds, s);
  else if (MT <= 5) launch_w2<5>(P, grid, pl.lds, s);
  else launch_w2<16>(P, grid, pl.lds, s);
  int rc = check_launch("conv_wgrad_small_cout");
  if (rc) return rc;
  rc = conv_wgrad_reduce_launch((const float*)ws, dw, pl.G, ds.Cout, ds.Cin, ds.KH, ds.KW, 2, beta, nullptr, nullptr, 0, 0, s);
  if (rc) return rc;
  if (db) rc = conv_bias_grad(d, dy, nullptr, db, beta, reinterpret_cast<float*>(static_cast<char*>(ws) + slab_bytes), s);
  return rc;
}

int conv_wgrad_mfma(const srk_conv_desc& d, const float* x, const float* dy, const srk_bwd_mask* mask, float* dw,
                    float* db, float beta, void* ws, size_t ws_bytes, hipStream_t s) {
  {
    srk_conv_desc ds;
    const char* e = env_str("SRK_WGRAD_SWAP");  // 0 disables the role-swapped small-Cout path
    if (!(e && atoi(e) == 0) && !(mask && mask->y) && swap_small_cout(d, ds) && plan(ds).ok && plan(ds).smallcin)
      return conv_wgrad_small_cout(d, ds, x, dy, dw, db, beta, ws, ws_bytes, s);
  }
  WgPlan pl = plan(d);
  if (!pl.ok) {
    set_error("conv_wgrad_mfma: shape not covered");
    return SRK_ERR_UNSUPPORTED;
  }
  const size_t slab_bytes = (size_t)pl.G * d.KH * d.KW * d.Cin * d.Cout * sizeof(float);
  const size_t need = slab_bytes + conv_bias_grad_ws(d);
  if (!ws || ws_bytes < need) {
    set_error("conv_wgrad_mfma: workspace %zu < %zu", ws_bytes, need);
    return SRK_ERR_WORKSPACE;
  }
  WgradParams P{};
  P.x = x; P.dy = dy; P.mask_y = mask ? mask->y : nullptr; P.mask_slope = mask ? mask->slope : 0.f;
  P.ws = (float*)ws;
  float* bias_ws = reinterpret_cast<float*>(static_cast<char*>(ws) + slab_bytes);
  P.bias_partial = (db && !d.transposed) ? bias_ws : nullptr;
  P.N = d.N; P.Cin = d.Cin; P.Cout = d.Cout;
  P.XH = d.H; P.XW = d.W; P.YH = d.OH; P.YW = d.OW;
  P.KH = d.KH; P.KW = d.KW; P.stride = d.stride; P.pad = d.pad; P.transposed = d.transposed;
  P.AH = d.transposed ? d.H : d.OH; P.AW = d.transposed ? d.W : d.OW;
  P.BH = d.transposed ? d.OH : d.H; P.BW = d.transposed ? d.OW : d.W;
  P.TH = pl.TH; P.TW = pl.TW; P.tiles_y = pl.tiles_y; P.tiles_x = pl.tiles_x; P.HH = pl.HH; P.HW = pl.HW;
  P.ntiles = pl.ntiles; P.G = pl.G; P.PSX = pl.PSX; P.PSY = pl.PSY; P.xs_floats = pl.xs_floats;
  P.vec_x = (d.Cin % 4 == 0) && ((uintptr_t)x % 16 == 0);
  P.vec_y = (d.Cout % 4 == 0) && ((uintptr_t)dy % 16 == 0) && (!P.mask_y || (uintptr_t)P.mask_y % 16 == 0);
  // every (t, ci, co) of every slab is written exactly once by exactly one block => no memset needed
  if (pl.smallcin) {
    const int MT = (d.KH * d.KW * d.Cin + 15) / 16;
    dim3 grid(pl.G, 1, cdiv(d.Cout, 64));
    if (MT <= 2) launch_w2<2>(P, grid, pl.lds, s);
    else if (MT <= 5) launch_w2<5>(P, grid, pl.lds, s);
    else launch_w2<16>(P, grid, pl.lds, s);
  } else {
    dim3 grid(pl.G, cdiv(d.Cin, 64), cdiv(d.Cout, pl.NTC * 16));
    if (!d.transposed) {
      switch (pl.NTC) {
        case 1: launch_w1<1, false>(P, grid, pl.lds, s); break;
        case 2: launch_w1<2, false>(P, grid, pl.lds, s); break;
        case 3: launch_w1<3, false>(P, grid, pl.lds, s); break;
        default: launch_w1<4, false>(P, grid, pl.lds, s); break;
      }
    } else {
      switch (pl.NTC) {
        case 1: launch_w1<1, true>(P, grid, pl.lds, s); break;
        case 2: launch_w1<2, true>(P, grid, pl.lds, s); break;
        case 3: launch_w1<3, true>(P, grid, pl.lds, s); break;
        default: launch_w1<4, true>(P, grid, pl.lds, s); break;
      }
    }
  }
  int rc = check_launch("conv_wgrad_mfma");
  if (rc) return rc;
  const bool fused_bias = db && P.bias_partial;
  rc = conv_wgrad_reduce_launch((const float*)ws, dw, pl.G, d.Cout, d.Cin, d.KH, d.KW, d.transposed, beta,
                                fused_bias ? bias_ws : nullptr, fused_bias ? db : nullptr, d.Cout, 0, s);
  if (rc) return rc;
  if (db && !fused_bias) rc = conv_bias_grad(d, dy, mask, db, beta, bias_ws, s);
  return rc;
}

}  // namespace srk

extern "C" int srk_wgrad_reduce_defer(int on) {
  const int prev = srk::g_red_defer;
  srk::g_red_defer = on ? 1 : 0;
  return prev;
}

extern "C" int srk_wgrad_reduce_flush(void* stream) { return srk::wgrad_reduce_flush((hipStream_t)stream); }
