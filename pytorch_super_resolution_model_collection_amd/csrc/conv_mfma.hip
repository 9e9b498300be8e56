// Implicit-GEMM convolution on the CDNA4 matrix cores, exact fp32 (v_mfma_f32_16x16x4_f32).
//
// GEMM view: M = output pixels, N = output channels, K = taps x input channels.
// One 256-thread block (4 waves) owns a tile of <=128 output pixels x <=64 output channels:
//   * the NHWC input halo of the tile ((TH-1)*is+KH) x ((TW-1)*is+KW) pixels x CK channels is
//     staged ONCE in LDS with coalesced 16-byte global reads (zero-filled outside the image, the
//     activation-gradient mask applied on the fly for backward-data); every tap then re-reads it
//     from LDS — HBM sees each input element ~once (+halo overlap, L2-absorbed);
//   * the filter slice of the current tap [CK][<=64] is staged in LDS, the next tap's slice is
//     prefetched into registers while the current one is being multiplied;
//   * each wave owns 32 pixels x 64 channels = 2 x 4 MFMA tiles (32 accumulator VGPRs); per
//     16-channel K step a lane issues 2 ds_read_b128 (A) + 16 ds_read_b32 (B) for 32 MFMAs;
//   * epilogue: bias + activation + residual add + optional pixel-shuffle scatter, straight
//     from the accumulators (C/D layout: col = lane&15, row = (lane>>4)*4 + reg).
//
// One kernel serves Conv2d forward (any stride), its data gradient (stride 1: flipped taps;
// stride s: s*s phase launches), ConvTranspose2d forward and its data gradient, through the
// "phase space" parametrisation below.  Input-channel counts <= 4 (first layers of the forward
// nets, last layers in backward-data) use the tap-group variant, which packs 4 taps x 4 padded
// channels into each 16-deep K step instead of wasting 13/16 of the MFMA on zero channels.
#include "srk_common.h"
#include "conv_problem.h"
#include "conv_tile.h"
#include "bf16_frag.h"   // amax_peek / amax_commit / abs_max4
#include <type_traits>

namespace srk {

typedef float f32x4 __attribute__((ext_vector_type(4)));


__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// Halo staging: global NHWC -> LDS [pixel][PSA], channels [cb, cb+ck) zero-padded to ckp.
// ---------------------------------------------------------------------------------------------
// All loads of a batch (<= HALO_BATCH float4 per thread) are issued before the first LDS store, so
// the HBM/L2 latency is paid once per batch, not once per element.
constexpr int HALO_BATCH = 8;

template <bool MASK>
__device__ __forceinline__ void load_halo_t(const MfmaConvParams& P, float* halo, int n, int r0, int c0, int cb,
                                            int ck, int ckp) {
  const int nvec = ckp >> 2;  // float4 slots per pixel
  const int items = P.HH * P.HW * nvec;
  const int iyb = r0 * P.is + P.iy0, ixb = c0 * P.is + P.ix0;
  for (int base = threadIdx.x; base < items; base += 256 * HALO_BATCH) {
    f32x4 v[HALO_BATCH], m[HALO_BATCH];
    int dst[HALO_BATCH];
#pragma unroll
    for (int k = 0; k < HALO_BATCH; ++k) {
      const int it = base + 256 * k;
      v[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (MASK) m[k] = (f32x4){1.f, 1.f, 1.f, 1.f};
      dst[k] = -1;
      if (it < items) {
        const int hp = it / nvec, q = it - hp * nvec;
        const int hy = hp / P.HW, hx = hp - hy * P.HW;
        const int iy = iyb + hy, ix = ixb + hx;
        const int ch = q * 4;
        dst[k] = hp * P.PSA + ch;
        if (iy >= 0 && iy < P.IH && ix >= 0 && ix < P.IW && ch < ck) {
          const size_t off = (((size_t)n * P.IH + iy) * P.IW + ix) * P.IC + cb + ch;
          if (P.vec_in && ch + 3 < ck) {
            v[k] = *reinterpret_cast<const f32x4*>(P.in + off);
            if (MASK) m[k] = *reinterpret_cast<const f32x4*>(P.mask_y + off);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (ch + e < ck) {
                v[k][e] = P.in[off + e];
                if (MASK) m[k][e] = P.mask_y[off + e];
              }
            }
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < HALO_BATCH; ++k) {
      if (dst[k] >= 0) {
        f32x4 x = v[k];
        if (MASK) {
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = m[k][e] > 0.f ? x[e] : x[e] * P.mask_slope;
        }
        *reinterpret_cast<f32x4*>(halo + dst[k]) = x;
      }
    }
  }
}

__device__ __forceinline__ void load_halo(const MfmaConvParams& P, float* halo, int n, int r0, int c0, int cb, int ck,
                                          int ckp) {
  if (P.mask_y)
    load_halo_t<true>(P, halo, n, r0, c0, cb, ck, ckp);
  else
    load_halo_t<false>(P, halo, n, r0, c0, cb, ck, ckp);
}

// Epilogue shared by both MFMA variants: accumulators -> wave-private LDS slab (32 pixels x 64
// channels, row stride 68 floats) -> 16-byte stores where a wave writes 4 pixels x 256 contiguous
// bytes, with the fused bias / activation / residual / pixel-shuffle.  C/D layout: col = lane&15
// (channel), row = (lane>>4)*4 + reg (pixel); per-channel-group math hoisted into EpiCol.
constexpr int EPI_STRIDE = 68;

template <int NT>
__device__ __forceinline__ void store_tile(const MfmaConvParams& P, float* smem_f, const f32x4 (&acc)[2][NT], int n,
                                           int r0, int c0, int ocb, int wave, int lane) {
  const int j = lane & 15, kq = lane >> 4;
  const int npx = P.TH * P.TW;
  constexpr int Q4 = NT * 4;
  __syncthreads();  // every wave is done reading the halo / filter regions that the slab overlays
  float* st = smem_f + wave * (32 * EPI_STRIDE);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) st[(mt * 16 + kq * 4 + reg) * EPI_STRIDE + nt * 16 + j] = acc[mt][nt][reg];
  __syncthreads();
  constexpr int RPI = 64 / Q4;
  const int row0 = lane / Q4, q4 = lane - row0 * Q4;
  const int oc4 = ocb + q4 * 4;
  // running maximum of what the block stores (ep.y_amax: the next layer's f16x3 scale -- a first layer on this kernel used
  // to cost the trunk behind it an srk_absmax pass); valid when every channel group takes the 16-byte path, which the
  // host checks before it reports the maximum as written (conv_mfma_gather)
  float amax = 0.f;
  const float peeked = amax_peek(P.ep.y_amax, blockIdx.x + wave);
  if (row0 < RPI && oc4 < P.OC) {
    const int tw_magic = div_small_magic(P.TW);
    const EpiCol col = epi_col_setup(P.ep, P.OW, P.OC, oc4);
#pragma unroll 2
    for (int row = row0; row < 32; row += RPI) {
      const int m = wave * 32 + row;
      if (m < npx) {
        const int r = div_small(m, tw_magic), c = m - r * P.TW;
        const int pr = r0 + r, pc = c0 + c;
        if (pr < P.PH && pc < P.PW) {
          const epi_f4 v = *reinterpret_cast<const epi_f4*>(st + row * EPI_STRIDE + q4 * 4);
          const epi_f4 o = epi_store4_col(P.ep, col, P.OH, P.OW, P.OC, n, P.oy0 + pr * P.os, P.ox0 + pc * P.os, v, P.out);
          if (P.ep.y_amax) amax = abs_max4(amax, o);
        }
      }
    }
  }
  if (P.ep.y_amax) amax_commit(P.ep.y_amax, amax, blockIdx.x + wave, peeked);
}

// ---------------------------------------------------------------------------------------------
// Main variant: IC >= 5 (channels padded to 16 inside LDS).
// ---------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256, 2) void k_conv_mfma(MfmaConvParams P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* halo = smem;
  float* wl = smem + P.halo_floats;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kq = lane >> 4;
  int b = blockIdx.x;
  const int txi = b % P.tiles_x;
  b /= P.tiles_x;
  const int tyi = b % P.tiles_y;
  const int n = b / P.tiles_y;
  const int r0 = tyi * P.TH, c0 = txi * P.TW;
  const int ocb = blockIdx.y * 64;
  const int npx = P.TH * P.TW;

  int aoff[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    int m = wave * 32 + mt * 16 + j;
    if (m >= npx) m = 0;
    const int r = m / P.TW, c = m - r * P.TW;
    aoff[mt] = ((r * P.is) * P.HW + c * P.is) * P.PSA + 4 * kq;
  }

  f32x4 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int T = P.KHv * P.KWv;
  const int BNp = P.BNp;
  constexpr int WV = NT * 4;  // float4 slots per filter row

  if (T > 0) {
    for (int cb = 0; cb < P.IC; cb += P.CK) {
      const int ck = (P.IC - cb) < P.CK ? (P.IC - cb) : P.CK;
      const int ckp = (ck + 15) & ~15;
      const int witems = ckp * WV;
      __syncthreads();  // previous chunk fully consumed
      load_halo(P, halo, n, r0, c0, cb, ck, ckp);

      f32x4 wr[4];
      auto fetch_w = [&](int t) {
        const int u = t / P.KWv, v = t - u * P.KWv;
        const int tapw = (P.wh0 + P.wdh * u) * P.KW_full + (P.ww0 + P.wdw * v);
        const float* base = P.wp + ((size_t)tapw * P.IC + cb) * P.OC + ocb;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int item = tid + it * 256;
          f32x4 w = {0.f, 0.f, 0.f, 0.f};
          if (item < witems) {
            const int ci = item / WV, co = (item - ci * WV) * 4;
            if (ci < ck) {
              const float* p = base + (size_t)ci * P.OC + co;
              if (P.vec_w && ocb + co + 3 < P.OC) {
                w = *reinterpret_cast<const f32x4*>(p);
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  if (ocb + co + e < P.OC) w[e] = p[e];
              }
            }
          }
          wr[it] = w;
        }
      };
      auto stash_w = [&]() {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int item = tid + it * 256;
          if (item < witems) {
            const int ci = item / WV, co = (item - ci * WV) * 4;
            *reinterpret_cast<f32x4*>(wl + ci * BNp + co) = wr[it];
          }
        }
      };

      fetch_w(0);
      for (int t = 0; t < T; ++t) {
        __syncthreads();  // previous tap's reads of wl done (and halo stores visible for t == 0)
        stash_w();
        if (t + 1 < T) fetch_w(t + 1);  // in flight while this tap is multiplied
        __syncthreads();
        const int u = t / P.KWv, v = t - u * P.KWv;
        const int toff = (u * P.HW + v) * P.PSA;
        const float* ha0 = halo + aoff[0] + toff;
        const float* ha1 = halo + aoff[1] + toff;
        for (int c16 = 0; c16 < ckp; c16 += 16) {
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(ha0 + c16);
          const f32x4 a1 = *reinterpret_cast<const f32x4*>(ha1 + c16);
          float bq[NT][4];
          const float* wrow = wl + (c16 + 4 * kq) * BNp + j;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[nt][q] = wrow[q * BNp + nt * 16];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              acc[0][nt] = mfma16(a0[q], bq[nt][q], acc[0][nt]);
              acc[1][nt] = mfma16(a1[q], bq[nt][q], acc[1][nt]);
            }
          }
        }
      }
    }
  }
  store_tile<NT>(P, smem, acc, n, r0, c0, ocb, wave, lane);
}

// ---------------------------------------------------------------------------------------------
// Tap-group variant: IC <= 4.  LDS halo is [pixel][4] (channels zero-padded); a 16-deep K step
// covers 4 taps x 4 channels: k-slot kq <-> tap 4*tg + kq, element q <-> channel q.
// The filter for up to TG_STAGE tap groups is staged per pass as rows (tap_local*4 + ci).
// ---------------------------------------------------------------------------------------------
constexpr int TG_STAGE = 8;  // 32 taps per filter stage

template <int NT>
__global__ __launch_bounds__(256, 2) void k_conv_mfma_tg(MfmaConvParams P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* halo = smem;
  float* wl = smem + P.halo_floats;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kq = lane >> 4;
  int b = blockIdx.x;
  const int txi = b % P.tiles_x;
  b /= P.tiles_x;
  const int tyi = b % P.tiles_y;
  const int n = b / P.tiles_y;
  const int r0 = tyi * P.TH, c0 = txi * P.TW;
  const int ocb = blockIdx.y * 64;
  const int npx = P.TH * P.TW;

  int aoff[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    int m = wave * 32 + mt * 16 + j;
    if (m >= npx) m = 0;
    const int r = m / P.TW, c = m - r * P.TW;
    aoff[mt] = ((r * P.is) * P.HW + c * P.is) * 4;
  }
  f32x4 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int T = P.KHv * P.KWv;
  const int BNp = P.BNp;
  constexpr int WV = NT * 4;

  if (T > 0) {
    load_halo(P, halo, n, r0, c0, 0, P.IC, 4);
    const int ngroups = (T + 3) >> 2;
    for (int g0 = 0; g0 < ngroups; g0 += TG_STAGE) {
      const int ng = (ngroups - g0) < TG_STAGE ? (ngroups - g0) : TG_STAGE;
      const int rows = ng * 16;  // (tap_local*4 + ci)
      __syncthreads();           // previous stage consumed (halo visible on first pass)
      for (int item = tid; item < rows * WV; item += 256) {
        const int row = item / WV, co = (item - row * WV) * 4;
        const int t = g0 * 4 + (row >> 2), ci = row & 3;
        f32x4 w = {0.f, 0.f, 0.f, 0.f};
        if (t < T && ci < P.IC) {
          const int u = t / P.KWv, v = t - u * P.KWv;
          const int tapw = (P.wh0 + P.wdh * u) * P.KW_full + (P.ww0 + P.wdw * v);
          const float* p = P.wp + ((size_t)tapw * P.IC + ci) * P.OC + ocb + co;
          if (P.vec_w && ocb + co + 3 < P.OC) {
            w = *reinterpret_cast<const f32x4*>(p);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (ocb + co + e < P.OC) w[e] = p[e];
          }
        }
        *reinterpret_cast<f32x4*>(wl + row * BNp + co) = w;
      }
      __syncthreads();
      for (int tg = 0; tg < ng; ++tg) {
        int t = (g0 + tg) * 4 + kq;
        if (t >= T) t = 0;  // its filter rows are zero
        const int u = t / P.KWv, v = t - u * P.KWv;
        const int toff = (u * P.HW + v) * 4;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(halo + aoff[0] + toff);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(halo + aoff[1] + toff);
        float bq[NT][4];
        const float* wrow = wl + ((tg * 4 + kq) * 4) * BNp + j;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int q = 0; q < 4; ++q) bq[nt][q] = wrow[q * BNp + nt * 16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            acc[0][nt] = mfma16(a0[q], bq[nt][q], acc[0][nt]);
            acc[1][nt] = mfma16(a1[q], bq[nt][q], acc[1][nt]);
          }
        }
      }
    }
  }
  store_tile<NT>(P, smem, acc, n, r0, c0, ocb, wave, lane);
}

// ---------------------------------------------------------------------------------------------
// Direct variant: OC <= 4 (the 64->3 output convs; 3->3 LapSRN image deconvs).  An MFMA tile
// would be >= 75% padding, so this is plain VALU: one thread per output pixel (<=256 per block),
// the input halo staged in LDS exactly as above, the filter read through the scalar cache
// (its address is wave-uniform), OC fp32 FMAs per LDS element.
// ---------------------------------------------------------------------------------------------
template <int OCT>
__global__ __launch_bounds__(256, 2) void k_conv_direct(MfmaConvParams P) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* halo = smem;
  const int tid = threadIdx.x;
  int b = blockIdx.x;
  const int txi = b % P.tiles_x;
  b /= P.tiles_x;
  const int tyi = b % P.tiles_y;
  const int n = b / P.tiles_y;
  const int r0 = tyi * P.TH, c0 = txi * P.TW;
  const int npx = P.TH * P.TW;
  const bool live = tid < npx;
  const int m = live ? tid : 0;
  const int r = m / P.TW, c = m - r * P.TW;
  const int aoff = ((r * P.is) * P.HW + c * P.is) * P.PSA;
  float acc[OCT];
#pragma unroll
  for (int o = 0; o < OCT; ++o) acc[o] = 0.f;
  const int T = P.KHv * P.KWv;
  const float* __restrict__ wp = P.wp;
  if (T > 0) {
    for (int cb = 0; cb < P.IC; cb += P.CK) {
      const int ck = (P.IC - cb) < P.CK ? (P.IC - cb) : P.CK;
      const int ckp = (ck + 3) & ~3;
      __syncthreads();
      load_halo(P, halo, n, r0, c0, cb, ck, ckp);
      __syncthreads();
      for (int t = 0; t < T; ++t) {
        const int u = t / P.KWv, v = t - u * P.KWv;
        const int tapw = (P.wh0 + P.wdh * u) * P.KW_full + (P.ww0 + P.wdw * v);
        const float* __restrict__ wt = wp + ((size_t)tapw * P.IC + cb) * P.OC;
        const float* hp = halo + aoff + (u * P.HW + v) * P.PSA;
        const int full = ck & ~3;
        for (int c4 = 0; c4 < full; c4 += 4) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(hp + c4);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int o = 0; o < OCT; ++o) acc[o] = fmaf(a[e], wt[(c4 + e) * OCT + o], acc[o]);
        }
        if (full < ck) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(hp + full);
#pragma unroll
          for (int e = 0; e < 3; ++e)
            if (full + e < ck) {
#pragma unroll
              for (int o = 0; o < OCT; ++o) acc[o] = fmaf(a[e], wt[(full + e) * OCT + o], acc[o]);
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
  for (int o = 0; o < OCT; ++o) epi_store(P.ep, g, acc[o], n, oy, ox, o, P.out);
}

// ---------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------
bool conv_mfma_gather_supported(const GatherConv& g, const Epi& ep) {
  (void)ep;
  if (g.OC < 8) return false;                       // <=4: direct variant; 5..7: generic kernel
  if (g.KH * g.KW > 32 * 32) return false;
  if ((long)g.N * g.OH * g.OW > (1L << 30)) return false;
  return true;
}

// Raise the dynamic-LDS limit of a kernel once per device (host call, not a stream op): srk_common.h LdsLimit.
static void ensure_lds(const void* fn, LdsLimit& lim, size_t lds) { lim.ensure(fn, lds); }

template <int NT>
static void launch_variant(bool tapgroup, const MfmaConvParams& P, dim3 grid, size_t lds, hipStream_t s) {
  static LdsLimit cur_tg, cur_main;
  note_amax_written(P.ep.y_amax != nullptr && epi_all_vector(P));   // (store_tile: the vector path keeps the maximum)
  if (tapgroup) {
    ensure_lds(reinterpret_cast<const void*>(&k_conv_mfma_tg<NT>), cur_tg, lds);
    note_kernel("k_conv_mfma_tg<%d>", NT);
    hipLaunchKernelGGL(k_conv_mfma_tg<NT>, grid, dim3(256), lds, s, P);
  } else {
    ensure_lds(reinterpret_cast<const void*>(&k_conv_mfma<NT>), cur_main, lds);
    note_kernel("k_conv_mfma<%d>", NT);
    hipLaunchKernelGGL(k_conv_mfma<NT>, grid, dim3(256), lds, s, P);
  }
}

static void apply_pick(MfmaConvParams& P, const TilePick& t) {
  P.TH = t.TH; P.TW = t.TW; P.tiles_y = t.tiles_y; P.tiles_x = t.tiles_x; P.HH = t.HH; P.HW = t.HW;
}

template <int OCT>
static void launch_direct(const MfmaConvParams& P, dim3 grid, size_t lds, hipStream_t s) {
  static LdsLimit cur;
  ensure_lds(reinterpret_cast<const void*>(&k_conv_direct<OCT>), cur, lds);
  note_kernel("k_conv_direct<%d>", OCT);
  hipLaunchKernelGGL(k_conv_direct<OCT>, grid, dim3(256), lds, s, P);
}

// Channel-chunk / tile choice for the variants that stage [pixel][CK+4] halos: try chunks of
// 64 / 32 / 16 channels, keep the largest unless a smaller one tiles the phase grid >15% more
// efficiently (strided convs and 9x9 kernels have large halos).
static bool pick_chunk(int maxpix, const MfmaConvParams& P, int icp, int wfloats_per_ci, int& CKout, TilePick& out) {
  int bestCK = 0;
  for (int CK = 64; CK >= 16; CK >>= 1) {
    const int ck = icp < CK ? icp : CK;
    if (ck == bestCK) continue;
    TilePick tp{};
    if (!pick_tile(maxpix, P.PH, P.PW, P.is, P.KHv > 0 ? P.KHv : 1, P.KWv > 0 ? P.KWv : 1, ck + 4,
                   kLdsBudgetBytes / 4 - ck * wfloats_per_ci, tp))
      continue;
    if (bestCK == 0 || tp.eff > out.eff * 1.15) {
      bestCK = ck;
      out = tp;
    }
  }
  CKout = bestCK;
  return bestCK != 0;
}

static int launch_phase(MfmaConvParams P, hipStream_t s) {
  const int T = P.KHv * P.KWv;
  TilePick best{};
  if (P.OC <= 4) {  // direct VALU variant
    const int icp = (P.IC + 3) & ~3;
    int CK = 0;
    if (!pick_chunk(256, P, icp, 0, CK, best)) {
      set_error("conv_direct: no tile fits LDS");
      return SRK_ERR_UNSUPPORTED;
    }
    P.CK = CK;
    P.PSA = CK + 4;
    apply_pick(P, best);
    P.halo_floats = best.HH * best.HW * P.PSA;
    const size_t lds = (size_t)P.halo_floats * 4;
    dim3 grid((unsigned)((size_t)P.tiles_x * P.tiles_y * P.N), 1);
    switch (P.OC) {
      case 1: launch_direct<1>(P, grid, lds, s); break;
      case 2: launch_direct<2>(P, grid, lds, s); break;
      case 3: launch_direct<3>(P, grid, lds, s); break;
      default: launch_direct<4>(P, grid, lds, s); break;
    }
    return check_launch("conv_direct");
  }
  const bool tapgroup = P.IC <= 4;
  const int NT = P.OC >= 64 ? 4 : (P.OC + 15) / 16;
  P.BNp = NT * 16 + 4;
  size_t lds;
  if (tapgroup) {
    P.CK = 4;
    P.PSA = 4;
    int ng = (T + 3) / 4;
    if (ng > TG_STAGE) ng = TG_STAGE;
    if (ng < 1) ng = 1;
    const int wfloats = ng * 16 * P.BNp;
    if (!pick_tile(128, P.PH, P.PW, P.is, P.KHv > 0 ? P.KHv : 1, P.KWv > 0 ? P.KWv : 1, 4,
                   kLdsBudgetBytes / 4 - wfloats, best)) {
      set_error("conv_mfma: no tile fits LDS (tap-group)");
      return SRK_ERR_UNSUPPORTED;
    }
    apply_pick(P, best);
    P.halo_floats = best.HH * best.HW * 4;
    lds = ((size_t)P.halo_floats + wfloats) * 4;
  } else {
    const int icp = (P.IC + 15) & ~15;
    int CK = 0;
    if (!pick_chunk(128, P, icp, P.BNp, CK, best)) {
      set_error("conv_mfma: no tile fits LDS");
      return SRK_ERR_UNSUPPORTED;
    }
    P.CK = CK;
    P.PSA = CK + 4;
    apply_pick(P, best);
    P.halo_floats = best.HH * best.HW * P.PSA;
    lds = ((size_t)P.halo_floats + (size_t)P.CK * P.BNp) * 4;
  }
  if (lds < (size_t)4 * 32 * EPI_STRIDE * sizeof(float)) lds = (size_t)4 * 32 * EPI_STRIDE * sizeof(float);
  dim3 grid((unsigned)((size_t)P.tiles_x * P.tiles_y * P.N), cdiv(P.OC, 64));
  switch (NT) {
    case 1: launch_variant<1>(tapgroup, P, grid, lds, s); break;
    case 2: launch_variant<2>(tapgroup, P, grid, lds, s); break;
    case 3: launch_variant<3>(tapgroup, P, grid, lds, s); break;
    default: launch_variant<4>(tapgroup, P, grid, lds, s); break;
  }
  return check_launch(tapgroup ? "conv_mfma_tg" : "conv_mfma");
}

// The direct variant is reached through the same gather entry (OC <= 4).
bool conv_direct_gather_supported(const GatherConv& g, const Epi& ep) {
  (void)ep;
  return g.OC <= 4 && g.KH * g.KW <= 32 * 32;
}
int conv_direct_gather(const GatherConv& g, const float* in, const float* wp, float* out, const Epi& ep,
                       const float* mask_y, float mask_slope, hipStream_t s) {
  return conv_mfma_gather(g, in, wp, out, ep, mask_y, mask_slope, s);
}

int conv_mfma_gather(const GatherConv& g, const float* in, const float* wp, float* out, const Epi& ep,
                     const float* mask_y, float mask_slope, hipStream_t s) {
  return for_each_phase(g, in, wp, out, ep, mask_y, mask_slope, [&](const MfmaConvParams& P) { return launch_phase(P, s); });
}

}  // namespace srk
