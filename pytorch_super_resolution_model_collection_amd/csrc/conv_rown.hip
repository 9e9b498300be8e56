// Few-output-channel convolutions with MANY taps (the 9x9 64 -> 3 output conv of SRGAN's generator, srgan.py:32; the 5x5
// 32 -> 3 layer that ends SRCNN, srcnn.py:20), stride 1: kernel ROWS on the matrix N axis, kernel COLUMNS in K ("rown").
//
// k_conv_tapn puts (tap, oc) on N and runs one [pixels x IC] x [IC x 32] GEMM per group of 32 / OC taps: nine passes for
// 9x9 x 3, each with its own z round trip through LDS, two barriers and a filter rebuild from fp32 scalars, over a halo that
// is 3x the 8 x 16 tile -- 213 - 231 us for 8.15 GFLOP on the SRGAN step (16 x 128 x 128), 4 % of the pipe.  Here:
//
//   Z[h][x][(u, oc)] = sum over v, c of  X[h][x + v][c] * W[u][v][c][oc]        one GEMM per INPUT row h:
//                                                                                M = 64 output columns, N = KH * OC <= 32,
//                                                                                K = KW * IC (576 for 9 x 64)
//   Y[h - u][x][oc] += Z[h][x][(u, oc)]                                          a KH-term sum down the columns
//
//   * a block owns TH x 64 outputs of one image and walks the TH + KH - 1 input rows of its halo.  Per row: the fp32 NHWC
//     row segment (64 + KW - 1 pixels) is split into bf16 planes and staged [plane][8-channel group][pixel] (16-byte slots:
//     one ds_read_b128 per MFMA operand, the tap shift v is an address offset; stores and loads conflict-free with a group
//     stride of 0 mod 16 slots); the loads of the NEXT row are issued before the matrix phase and committed after it;
//   * the K dimension is split over the 4 waves by (v, 32-channel step) pairs -- 18 pairs for 9 x 64: 5 / 5 / 4 / 4 -- so that
//     a wave's filter fragments (its pairs x 2 column tiles x NP planes) stay in REGISTERS for the life of the block;
//     every wave covers all four 16-pixel tiles of the row and leaves its partial Z in its own LDS slab;
//   * thread (x, oc) adds the four partials in a fixed order (deterministic) into the KH running column sums of its output
//     column, which live in its REGISTERS as a window that shifts by one row per input row; the oldest entry is complete
//     after each row: bias / activation / residual (bias and slope read once per block -- a load inside the per-row
//     epilogue was a global round trip per row: 3.0 of 5.5 us) and one coalesced 768-byte store per row;
//   * two barriers per input row; two blocks per CU (57 - 68 KB of LDS) hide each other's staging and column sums.
// Work beside the algorithmic FLOPs: N padding 27 -> 32, and the (TH + KH - 1) / TH halo rows (1.5x at TH = 16).
// Arithmetic: bf16x3 (NP = 2) or the exact three-way split bf16x6 (NP = 3), products smallest-first, as k_conv_tapn.
#include "srk_common.h"
#include "conv_problem.h"
#include "conv_tile.h"
#include "bf16_frag.h"
#include <stdlib.h>
#include <type_traits>

namespace srk {

// constant-ablation builds (tools/tu_variant.sh RN<bits> conv_rown.hip -DRN_ABL=<bits>; 0 in the release library):
// 1 no column sums, 2 no matrix phase, 4 no commit (split + LDS stores), 8 no global loads, 16 no z stores
#ifndef RN_ABL
#define RN_ABL 0
#endif
constexpr int RN_TW = 64;             // output columns per block
constexpr int RN_ZS = 36;             // z row stride in floats (conflict-free C/D fragment writes, as TAPN_ZS)

template <int I0, int I1, typename F>
__device__ __forceinline__ void rn_static_for(F&& f) {
  if constexpr (I0 < I1) {
    f(std::integral_constant<int, I0>{});
    rn_static_for<I0 + 1, I1>(f);
  }
}

// KS = IC / 32, OCT = output channels, KW = kernel width (compile time: the filter registers are indexed by it),
// NP = bf16 planes, NT = 16-column tiles of N (KH * OCT <= 16 * NT)
template <int KS, int OCT, int KW, int NP, int NT>
__global__ __launch_bounds__(256, 2) void k_conv_rown(MfmaConvParams P) {
  constexpr int NQ = KW * KS;                        // (v, channel step) pairs of K
  constexpr int QW = (NQ + 3) / 4;                   // ... per wave
  constexpr int NG = KS * 4;                         // 8-channel groups of a pixel
  constexpr int XV = RN_TW + KW - 1;                 // halo pixels of a row
  constexpr int XP = (XV + 15) & ~15;                // slots per group: 0 mod 16 (conflict-free ds_read_b128 across kq)
  constexpr int NIT = (XV * NG + 255) / 256;         // staging items (pixel, group) per thread
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
  uint4* xs = smem4;                                                  // [NP][NG][XP]
  float* zs = reinterpret_cast<float*>(smem4 + NP * NG * XP);         // [4 waves][64 px][RN_ZS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kq = lane >> 4;
  const int KH = P.KHv;
  const int NN = KH * OCT;
  // XCD-aware block order (as k_conv_tapn): neighbouring tiles share their halo rows in ONE L2
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
  const int r0 = tyi * P.TH, c0 = txi * RN_TW;
  const int th = P.PH - r0 < P.TH ? P.PH - r0 : P.TH;    // output rows of this tile
  const int nrows = th + KH - 1;                         // input rows it walks

  // ---- this wave's filter fragments: pairs q = wave, wave + 4, ...; column nn = nt*16 + j = (u, oc); K slot e of lane
  // (j, kq) = channel ks*32 + kq*8 + e (the order of the staged A operand).  The fp32 filter [tap][IC][OC] (62 KB for
  // 9 x 9 x 64 x 3) goes through LDS first, read from global memory ONCE per block in whole lines: gathered straight from
  // global memory a lane's 80 values are 80 wave-loads of 64 different lines each -- ~20 k L2 requests per block.
  {
    float* wsm = reinterpret_cast<float*>(smem4);
    const int nflt = P.KHv * P.KW_full * P.IC * OCT;   // (stride-1 gathers: every tap of the filter is a valid one)
    if ((nflt & 3) == 0 && (reinterpret_cast<uintptr_t>(P.wp) & 15) == 0) {
      for (int i = tid; i < (nflt >> 2); i += 256) reinterpret_cast<f32x4*>(wsm)[i] = reinterpret_cast<const f32x4*>(P.wp)[i];
    } else {
      for (int i = tid; i < nflt; i += 256) wsm[i] = P.wp[i];
    }
    __syncthreads();
  }
  uint4 bw[QW][NT][NP];
#pragma unroll
  for (int qi = 0; qi < QW; ++qi) {
    const int q = wave + 4 * qi;
    const bool qon = q < NQ;
    const int qq = qon ? q : 0;
    const int v = qq / KS, ks = qq - v * KS;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int nn = nt * 16 + j;
      const bool on = qon && nn < NN;
      const int u = on ? nn / OCT : 0, oc = on ? nn - u * OCT : 0;
      const int tapw = (P.wh0 + P.wdh * u) * P.KW_full + (P.ww0 + P.wdw * v);
      const float* w = reinterpret_cast<const float*>(smem4) + ((size_t)tapw * P.IC + ks * 32 + kq * 8) * OCT + oc;
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = w[e * OCT];   // (unconditional, in-bounds reads; padding columns zeroed below)
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = on ? f[e] : 0.f;
      split8n<NP>(f, bw[qi][nt]);
    }
  }
  __syncthreads();   // the filter image is dead: xs / zs take its place

  // ---- staging items of this thread: 8 consecutive pixels x the groups of a pixel per 8 * NG consecutive items (a wave
  // reads whole pixels: 256-byte rows; 8 adjacent lanes store 8 adjacent slots of one group: conflict-free)
  int it_px[NIT], it_g[NIT];
  bool it_on[NIT];
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int i = tid + 256 * k;
    const int oct = i / (8 * NG), rem = i - oct * (8 * NG);
    it_g[k] = rem >> 3;
    it_px[k] = oct * 8 + (rem & 7);
    it_on[k] = it_px[k] < XV;
  }
  const float* __restrict__ inb = P.in + (size_t)n * P.IH * P.IW * P.IC;
  f32x4 raw[NIT][2];
  unsigned okmask = 0;   // bit k: item k of the row in `raw` lies inside the image (applied at commit: nothing waits at issue)
  auto issue = [&](int hh) {   // global loads of input row hh of the halo, from clamped addresses (no load under a branch)
    const int iy = r0 + hh + P.iy0;
    const bool rok = hh < nrows && (unsigned)iy < (unsigned)P.IH;
    const int iyc = rok ? iy : 0;
    okmask = 0;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int ix = c0 + it_px[k] + P.ix0;
      const bool ok = rok && it_on[k] && (unsigned)ix < (unsigned)P.IW;
      const int ixc = ok ? ix : 0;
      const float* src = inb + ((size_t)iyc * P.IW + ixc) * P.IC + it_g[k] * 8;
      raw[k][0] = *reinterpret_cast<const f32x4*>(src);
      raw[k][1] = *reinterpret_cast<const f32x4*>(src + 4);
      okmask |= ok ? (1u << k) : 0u;
    }
  };
  auto commit = [&]() {        // zeros outside the image, split into planes, store [plane][group][pixel]
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      if (it_on[k]) {
        const bool ok = (okmask >> k) & 1u;
        float f[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          f[e] = ok ? raw[k][0][e] : 0.f;
          f[4 + e] = ok ? raw[k][1][e] : 0.f;
        }
        uint4 pl[NP];
        split8n<NP>(f, pl);
#pragma unroll
        for (int p = 0; p < NP; ++p) xs[(p * NG + it_g[k]) * XP + it_px[k]] = pl[p];
      }
    }
  };
  // rows of the input that lie outside the image contribute zeros: their staging and matrix phases are skipped
  auto row_in_image = [&](int hh) { return (unsigned)(r0 + hh + P.iy0) < (unsigned)P.IH; };

  // column-sum role of this thread: output column x = tid / OCT, channel oc = tid % OCT (the first 64 * OCT threads);
  // bias / activation slope are read ONCE here -- a load inside the per-row epilogue is a global round trip per row
  const bool cs_on = tid < RN_TW * OCT;
  const int cs_x = cs_on ? tid / OCT : 0, cs_oc = cs_on ? tid - cs_x * OCT : 0;
  const bool cs_px_ok = cs_on && c0 + cs_x < P.PW;
  const float cs_bias = P.ep.bias ? P.ep.bias[cs_oc] : 0.f;
  const float cs_slope = P.ep.act == SRK_ACT_PRELU ? P.ep.prelu_w[P.ep.prelu_n > 1 ? cs_oc : 0] : P.ep.slope;
  float col[KW];
#pragma unroll
  for (int u = 0; u < KW; ++u) col[u] = 0.f;
  issue(0);
  for (int hh = 0; hh < nrows; ++hh) {
    const bool rok = row_in_image(hh);   // block-uniform
    if (rok && !(RN_ABL & 4)) commit();
    if (!(RN_ABL & 8)) issue(hh + 1);    // in flight across the matrix phase and the column sums
    __syncthreads();
    if (rok && !(RN_ABL & 2)) {
      // ---- matrix phase: this wave's K pairs over the four 16-pixel tiles of the row
#pragma unroll 1
      for (int mh = 0; mh < RN_TW / 32; ++mh) {   // two 16-pixel tiles at a time: four independent accumulator chains
        f32x4 acc[2][NT];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // Every wave runs QW pairs: a pair past the end of K (waves 2, 3 of 18 pairs) multiplies zero filter fragments by the
        // operand of pair 0 -- no branch in the stream, so the reads of pair qi + 1 are issued in front of the MFMAs of pair
        // qi (with the wave-uniform `if (q < NQ)` of the first version every pair waited for its own reads: 28 clocks per
        // MFMA); the block waits for its 5-pair waves either way.
        uint4 a[2][2][NP];
        auto aread = [&](int buf, int qi) {
          const int q = wave + 4 * qi;
          const int qq = q < NQ ? q : 0;
          const int v = qq / KS, ks = qq - v * KS;
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int p = 0; p < NP; ++p) a[buf][m][p] = xs[(p * NG + ks * 4 + kq) * XP + (mh * 2 + m) * 16 + j + v];
        };
        aread(0, 0);
        rn_static_for<0, QW>([&](auto qic) {
          constexpr int qi = decltype(qic)::value, cur = qi & 1;
          if constexpr (qi + 1 < QW) aread(cur ^ 1, qi + 1);
          // smallest products first (as k_conv_tapn / k_conv_bfd)
#define SRK_ROWN_PASS(pa, pb)                                                    \
  _Pragma("unroll") for (int m = 0; m < 2; ++m) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) \
      acc[m][nt] = mfma16(a[cur][m][pa], bw[qi][nt][pb], acc[m][nt]);
          if constexpr (NP == 3) {
            SRK_ROWN_PASS(2, 0)
            SRK_ROWN_PASS(0, 2)
            SRK_ROWN_PASS(1, 1)
          }
          SRK_ROWN_PASS(1, 0)
          SRK_ROWN_PASS(0, 1)
          SRK_ROWN_PASS(0, 0)
#undef SRK_ROWN_PASS
        });
        // C/D layout: col = lane & 15 (nn), row = (lane >> 4) * 4 + reg (pixel of the tile)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          float* zr = zs + (size_t)(wave * RN_TW + (mh * 2 + m) * 16 + kq * 4) * RN_ZS + j;
          if (!(RN_ABL & 16)) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
              for (int e = 0; e < 4; ++e) zr[e * RN_ZS + nt * 16] = acc[m][nt][e];
          } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) asm volatile("" ::"v"(acc[m][nt]));
          }
        }
      }
    }
    __syncthreads();
    // ---- column sums: thread (x, oc) keeps the KH running sums of its output column in REGISTERS -- col[u] belongs to
    // output row hh - u; after this row's terms the oldest one (u = KH - 1) is complete, then the window shifts by one row
    if (cs_on && !(RN_ABL & 1)) {
      if (rok) {
        const float* z = zs + (size_t)cs_x * RN_ZS + cs_oc;
        float zv[KW][4];
#pragma unroll
        for (int u = 0; u < KW; ++u)
#pragma unroll
          for (int w = 0; w < 4; ++w) zv[u][w] = z[w * RN_TW * RN_ZS + u * OCT];
#pragma unroll
        for (int u = 0; u < KW; ++u) col[u] += (zv[u][0] + zv[u][1]) + (zv[u][2] + zv[u][3]);   // fixed order: deterministic
      }
      const int rr = hh - (KW - 1);
      if (rr >= 0 && cs_px_ok) {   // (rr < th by the loop bound)
        float vout = col[KW - 1] + cs_bias;
        if (P.ep.act != SRK_ACT_NONE) vout = act_apply(vout, P.ep.act, cs_slope);
        const size_t o = (((size_t)n * P.OH + (P.oy0 + (r0 + rr) * P.os)) * P.OW + (P.ox0 + (c0 + cs_x) * P.os)) * P.OC + cs_oc;
        if (P.ep.residual) vout += P.ep.residual[o];
        P.out[o] = vout;
      }
#pragma unroll
      for (int u = KW - 1; u > 0; --u) col[u] = col[u - 1];
      col[0] = 0.f;
    }
    // (no barrier here: the next commit writes xs, last read before the barrier above; the next matrix phase writes zs
    //  behind the next iteration's first barrier, which every thread reaches only after its column sums)
  }
}

// what the kernel covers: stride-1 gathers with <= 3 output channels whose taps do not fit one 32-column group (those stay
// on k_conv_tapn's single pass), 32 or 64 input channels, a kernel of <= 32 / OC rows and 3 .. 9 columns
bool conv_rown_gather_supported(const GatherConv& g, const float* in, const float* mask_y) {
  const int mode = env_int("SRK_ROWN", 1);   // 0: never (k_conv_tapn's tap groups), 2: whenever the shape is covered (tests)
  if (mode == 0) return false;
  // small problems stay on k_conv_tapn: a 64-column tile per block leaves most CUs idle (SRCNN's 5x5 32 -> 3 layer on
  // 16 x 48 x 48 outputs: 18.2 us here, 16.3 us there; SRGAN's 9x9 on 16 x 128 x 128: 69 - 95 us against 202 - 223)
  if (mode != 2 && (long)g.N * g.OH * g.OW < 128L * 1024) return false;
  if (g.OC < 1 || g.OC > 3 || g.KH * g.KW * g.OC <= 32) return false;
  if (g.IC != 32 && g.IC != 64) return false;
  if (g.stride != 1 || mask_y || g.in_nchw || g.in_ps_r > 1) return false;
  if (g.KH != g.KW || g.KH * g.OC > 32 || (g.KW != 9 && g.KW != 5 && g.KW != 7)) return false;   // (square: KW is the template argument)
  if ((uintptr_t)in % 16 != 0) return false;
  if ((long)g.IH * g.IW * g.IC >= (1L << 30)) return false;
  return true;
}

template <int KS, int OCT, int KW, int NP, int NT>
static int rown_launch_t(MfmaConvParams P, hipStream_t s) {
  constexpr int NG = KS * 4, XP = ((RN_TW + KW - 1) + 15) & ~15;
  size_t lds = (size_t)NP * NG * XP * 16 + (size_t)4 * RN_TW * RN_ZS * 4;
  const size_t flt = (size_t)P.KHv * P.KW_full * P.IC * OCT * 4;   // the fp32 filter passes through the same memory first
  if (lds < flt) lds = flt;
  static LdsLimit lim;
  lim.ensure(reinterpret_cast<const void*>(&k_conv_rown<KS, OCT, KW, NP, NT>), lds);
  note_kernel("k_conv_rown<%d,%d,%d,%d,%d>", KS, OCT, KW, NP, NT);
  dim3 grid((unsigned)((size_t)P.tiles_x * P.tiles_y * P.N));
  hipLaunchKernelGGL((k_conv_rown<KS, OCT, KW, NP, NT>), grid, dim3(256), lds, s, P);
  return check_launch("conv_rown");
}

template <int KS, int OCT, int KW>
static int rown_launch_kw(const MfmaConvParams& P, bool x6, hipStream_t s) {
  const int nt = (P.KHv * OCT + 15) / 16;
  if (nt == 1) return x6 ? rown_launch_t<KS, OCT, KW, 3, 1>(P, s) : rown_launch_t<KS, OCT, KW, 2, 1>(P, s);
  return x6 ? rown_launch_t<KS, OCT, KW, 3, 2>(P, s) : rown_launch_t<KS, OCT, KW, 2, 2>(P, s);
}

template <int KS, int OCT>
static int rown_launch(MfmaConvParams P, bool x6, hipStream_t s) {
  // tile height: the block count that fills the CUs (two resident blocks each) against the (TH + KH - 1) / TH halo rows;
  // a lone block per CU cannot hide its staging and column sums behind another block's matrix phase (x 1.25)
  const int kh = P.KHv, tx = (P.PW + RN_TW - 1) / RN_TW;
  int best = 8;
  double best_cost = 0.;
  const int forced = env_int("SRK_ROWN_TH", 0);
  for (int th = 8; th <= 64; th *= 2) {
    const long blocks = (long)P.N * ((P.PH + th - 1) / th) * tx;
    const long slots = 2L * kNumCU;
    const long waves = (blocks + slots - 1) / slots;          // rounds of two blocks per CU
    double cost = (double)waves * 2.0 * (th + kh - 1);
    if (blocks <= kNumCU) cost = 1.25 * (th + kh - 1);         // one block per CU (or fewer): no partner
    else if (blocks < slots) cost = (double)(th + kh - 1) * (1.0 + (double)(blocks - kNumCU) / kNumCU);
    if (best_cost == 0. || cost < best_cost) {
      best_cost = cost;
      best = th;
    }
    if (th >= P.PH) break;
  }
  if (forced > 0) best = forced;
  P.TH = best; P.TW = RN_TW;
  P.tiles_y = (P.PH + best - 1) / best;
  P.tiles_x = tx;
  P.HH = best + kh - 1; P.HW = RN_TW + P.KWv - 1;
  if ((long)P.tiles_x * P.tiles_y * P.N >= (1L << 30)) {
    set_error("conv_rown: too many tiles");
    return SRK_ERR_UNSUPPORTED;
  }
  switch (P.KWv) {
    case 9: return rown_launch_kw<KS, OCT, 9>(P, x6, s);
    case 7: return rown_launch_kw<KS, OCT, 7>(P, x6, s);
    case 5: return rown_launch_kw<KS, OCT, 5>(P, x6, s);
    default: return rown_launch_kw<KS, OCT, 3>(P, x6, s);
  }
}

int conv_rown_gather(const GatherConv& g, const float* in, const float* wp, float* out, const Epi& ep, bool x6,
                     hipStream_t s) {
  return for_each_phase(g, in, wp, out, ep, nullptr, 0.f, [&](const MfmaConvParams& P) {
    if (P.is != 1 || P.KWv != g.KW || P.KHv != g.KH) {
      set_error("conv_rown: strided gather");
      return (int)SRK_ERR_UNSUPPORTED;
    }
    const int key = (g.IC / 32) * 10 + g.OC;
    switch (key) {
      case 11: return rown_launch<1, 1>(P, x6, s);
      case 12: return rown_launch<1, 2>(P, x6, s);
      case 13: return rown_launch<1, 3>(P, x6, s);
      case 21: return rown_launch<2, 1>(P, x6, s);
      case 22: return rown_launch<2, 2>(P, x6, s);
      default: return rown_launch<2, 3>(P, x6, s);
    }
  });
}

}  // namespace srk
