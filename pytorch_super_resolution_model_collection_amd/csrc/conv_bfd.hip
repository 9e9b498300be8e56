// Implicit-GEMM convolution on the bf16 matrix cores, filters straight from global memory ("bfd").
//
// Same arithmetic family as conv_mfma_bf16.hip — fp32 operands split into bf16 planes, products on
// v_mfma_f32_16x16x32_bf16, fp32 accumulation — in two precisions:
//
//   NP = 2 (bf16x3)  a = h + m            a*b ~= m*h + h*m + h*h               (~5e-6 rel, 3 MFMAs)
//   NP = 3 (bf16x6)  a = h + m + l EXACT  a*b ~= l*h + h*l + m*m + m*h + h*m + h*h
//                    (dropped m*l, l*m, l*l <= 2^-23 |a*b|, typically ~1e-8: fp32-faithful, 6 MFMAs —
//                     2.7x the fp32-MFMA rate.  This is the training-forward path of the default
//                     "mixed" precision: ReLU masks must match the fp32 reference's.)
//
// Structure:
//   * the prepared filter layout [tap][chunk][ocb][plane][group][co][8 x bf16] IS the MFMA
//     B-operand layout, so every lane loads its own 16-byte fragments of the NEXT taps into
//     registers (prefetch depth PF) while the MFMAs of the current tap run.  Filters of a layer
//     are <= 221 KB and shared by every block: L1/L2 hits.  No weight LDS slots and NO barrier in
//     the tap loop — the waves of a block drift apart and overlap each other's loads and MFMAs;
//   * the fp32 NHWC halo of the block's pixel tile is read once, split into planes and staged in
//     LDS as [plane][8-channel group][pixel][8 x bf16] (conflict-free ds_read_b128 A fragments);
//   * a block is NPW pixel-waves x NOW channel-waves: wave (pw, ow) owns 64 pixels x NTW*16 output
//     channels.  Large problems use NOW = 1 (64 x 64 per wave: fewest LDS bytes per MFMA); small
//     problems (a strong-scaled shard of a batch) use 64-pixel blocks whose waves split the output
//     channels, so that 4x more SIMDs share the serial MFMA chain of a tile;
//   * epilogue through an LDS slab shared by the channel-waves of a pixel group: 16-byte stores,
//     a pixel's channels contiguous (bias / activation / residual / pixel-shuffle as everywhere).
#include "srk_common.h"
#include "conv_problem.h"
#include "conv_tile.h"
#include "bf16_frag.h"
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace srk {

struct BfdParams {
  MfmaConvParams P;
  const uint4* wq;   // prepared filters, planes h, m
  const uint4* wq3;  // third plane (l) [tap][chunk][ocb][group][co]
  int ICc;           // 32-channel chunks
  int OCb;           // 64-channel output blocks
  int NB;            // output channels per block in the prepared layout
  int NPIXp;         // LDS stride of an 8-channel group in 16-byte slots: >= the halo pixels, chosen with `perm` (bfd_lds_plan)
  unsigned long long perm;  // lane column (lane & 15) -> pixel of a 16-pixel M tile, 4 bits each
  int dbg;
  int allc;          // small problems: every channel chunk of the halo staged up front (one load latency, one barrier)
  int cpr;           // chunks staged per barrier round: ICc (allc), 2 (K-split without allc) or 1
  const float* w_descale;  // F16 kernels: trailer of the fp16 filter section {2^-kw, 2^kw}
};

// halo chunk: channels [cb, cb+32) of every halo pixel -> NP planes.  thread -> (8-channel group
// g = tid&3, pixel slot tid>>2); all global loads of a batch are issued before the first conversion.
constexpr int BFD_STAGE_IT = 6;

template <bool MASK, int NTHR, int NP, int IT, bool F16 = false>
__device__ __forceinline__ void bfd_stage_halo_t(const BfdParams& B, uint4* hal, int n, int r0, int c0, int cb,
                                                 float sx = 1.f) {
  const MfmaConvParams& P = B.P;
  const int npix = P.HH * P.HW;
  // (4 adjacent lanes = the 4 channel groups of one pixel: 128 contiguous bytes per pixel for the global loads.
  //  The resulting ds_write_b128 pattern is 4-way bank-conflicted; remapping lanes to make the LDS writes
  //  conflict-free breaks the adjacent-lane coalescing of the loads and measured 0.64 -> 0.81 ms on the c2 layer.)
  const int g = threadIdx.x & 3;
  const int hp0 = threadIdx.x >> 2;
  int hy = hp0 / P.HW, hx = hp0 - hy * P.HW;
  constexpr int PPP = NTHR / 4;  // pixels per pass
  const int dyp = PPP / P.HW, dxp = PPP - dyp * P.HW;
  const int iyb = r0 * P.is + P.iy0, ixb = c0 * P.is + P.ix0;
  const int ch = cb + g * 8;
  const bool ch_any = ch < P.IC, ch_vec = P.vec_in && ch + 7 < P.IC;
  const InAddr ia = conv_in_addr(P, ch);
  const size_t img_off = (size_t)n * P.IH * ia.sA;  // wave-uniform: scalar base pointers
  const float* __restrict__ inb = P.in + img_off;
  const float* __restrict__ mkb = MASK ? P.mask_y + img_off : nullptr;
  for (int base = hp0; base < npix; base += PPP * IT) {
    f32x4 v0[IT], v1[IT], m0[IT], m1[IT];
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      v0[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
      v1[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (MASK) {
        m0[k] = (f32x4){1.f, 1.f, 1.f, 1.f};
        m1[k] = (f32x4){1.f, 1.f, 1.f, 1.f};
      }
      const int hp = base + PPP * k;
      const int iy = iyb + hy, ix = ixb + hx;
      if (hp < npix && ch_any && (unsigned)iy < (unsigned)P.IH && (unsigned)ix < (unsigned)P.IW) {
        const unsigned off = (unsigned)iy * ia.sA + (unsigned)ix * ia.sB + ia.K;
        if (ch_vec) {
          v0[k] = *reinterpret_cast<const f32x4*>(inb + off);
          v1[k] = *reinterpret_cast<const f32x4*>(inb + off + 4);
          if (MASK) {
            m0[k] = *reinterpret_cast<const f32x4*>(mkb + off);
            m1[k] = *reinterpret_cast<const f32x4*>(mkb + off + 4);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (ch + e < P.IC) {
              v0[k][e] = inb[off + e];
              if (MASK) m0[k][e] = mkb[off + e];
            }
            if (ch + 4 + e < P.IC) {
              v1[k][e] = inb[off + 4 + e];
              if (MASK) m1[k][e] = mkb[off + 4 + e];
            }
          }
        }
      }
      hy += dyp;
      hx += dxp;
      if (hx >= P.HW) {
        hx -= P.HW;
        ++hy;
      }
    }
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      const int hp = base + PPP * k;
      if (hp < npix) {
        float f[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          f[e] = v0[k][e];
          f[4 + e] = v1[k][e];
          if (MASK) {
            f[e] = m0[k][e] > 0.f ? f[e] : f[e] * P.mask_slope;
            f[4 + e] = m1[k][e] > 0.f ? f[4 + e] : f[4 + e] * P.mask_slope;
          }
        }
        uint4 pl[NP];
        if constexpr (F16) split8h(f, sx, pl); else split8n<NP>(f, pl);
#pragma unroll
        for (int p = 0; p < NP; ++p) hal[p * (4 * B.NPIXp + 4) + lds_goff(g, B.NPIXp) + hp] = pl[p];
      }
    }
  }
}

constexpr int BFD_MAXTAPS = 128;
#ifndef BFD_OCC3
#define BFD_OCC3 1
#endif
#ifndef BFD_ROT
#define BFD_ROT 1
#endif
constexpr int BFD_EPI_STRIDE = 68;  // floats per staged output row (64 + 4: conflict-free float4 rows)

// accumulators -> LDS slab of the pixel group (32 pixels x block channels, two halves) -> 16-byte
// stores by all NOW waves of the group.  C/D layout: col = lane&15 (channel), row = (lane>>4)*4+reg.
template <int NTW, int NPW, int NOW>
__device__ __forceinline__ void bfd_epilogue(const MfmaConvParams& P, unsigned long long perm, float* smem_f,
                                             const f32x4 (&acc)[4][NTW], int n, int r0, int c0, int ocb, int pw, int ow,
                                             int lane, bool active = true) {
  float amax = 0.f;
  const float peeked = amax_peek(P.ep.y_amax, blockIdx.x + (threadIdx.x >> 6));
  const int j = lane & 15, kq = lane >> 4;
  const int npx = P.TH * P.TW;
  __syncthreads();
  float* st = smem_f + pw * (32 * BFD_EPI_STRIDE);
  constexpr int Q4 = NTW * NOW * 4;   // float4 columns per row
  constexpr int RPI = (64 * NOW) / Q4;  // rows per pass of the group
  const int gi = ow * 64 + lane;
  const int row0 = gi / Q4, q4 = gi - row0 * Q4;
  const int oc4 = ocb + q4 * 4;
  const bool lane_on = active && row0 < RPI && oc4 < P.OC;  // (inactive waves -- K-split partners -- only keep the barriers)
  const int tw_magic = div_small_magic(P.TW);
  EpiCol col{};
  if (lane_on) col = epi_col_setup(P.ep, P.OW, P.OC, oc4);
  const EpiTile et = epi_tile_setup(P, n, r0, c0);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (active) {
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg)
            st[(mh * 16 + (int)((perm >> (4 * (kq * 4 + reg))) & 15)) * BFD_EPI_STRIDE + (ow * NTW + nt) * 16 + j] =
                acc[2 * h + mh][nt][reg];   // (accumulator row i = the pixel lane column i read: perm)
    }
    __syncthreads();
    if (lane_on) {
#pragma unroll 2
      for (int row = row0; row < 32; row += RPI) {
        const int m = pw * 64 + h * 32 + row;
        if (m < npx) {
          const int r = div_small(m, tw_magic), c = m - r * P.TW;
          const int pr = r0 + r, pc = c0 + c;
          if (pr < P.PH && pc < P.PW) {
            const epi_f4 v = *reinterpret_cast<const epi_f4*>(st + row * BFD_EPI_STRIDE + q4 * 4);
            // vector path only: the host routes layers that could need the scalar fallback to other kernels
            // (conv_epi_all_vector), which keeps its hoisted index divisions out of these kernels
            const epi_f4 o = epi_store4_tile(P.ep, col, et, r, c, v, P.out);
            if (P.ep.y_amax) amax = abs_max4(amax, o);
          }
        }
      }
    }
    __syncthreads();
  }
  if (P.ep.y_amax) amax_commit(P.ep.y_amax, amax, blockIdx.x + (threadIdx.x >> 6), peeked);
}

// Direct epilogue of the transposed product (C/D: col = lane & 15 = pixel of the M tile, rows kq*4 + reg = 4
// consecutive output channels of the N tile): one 16-byte store per (pixel tile, channel tile) and lane, the 4 kq
// lanes of a pixel cover 64 contiguous bytes — no LDS slab, no barrier.  On a small problem (one tile per CU, the
// kernel a latency chain) the LDS-staged epilogue was 2.2 of the 12.4 us; its better store coalescing only pays
// when the epilogue is bandwidth-bound (large problems keep it).
template <int NTW>
__device__ __forceinline__ void bfd_epilogue_direct(const MfmaConvParams& P, unsigned long long perm,
                                                    const f32x4 (&acc)[4][NTW], int n, int r0, int c0, int ocb, int pw,
                                                    int ow, int lane, bool active) {
  if (!active) return;
  const int j = (int)((perm >> (4 * (lane & 15))) & 15), kq = lane >> 4;
  const int npx = P.TH * P.TW;
  const int tw_magic = div_small_magic(P.TW);
  const EpiTile et = epi_tile_setup(P, n, r0, c0);
  float amax = 0.f;
  const float peeked = amax_peek(P.ep.y_amax, blockIdx.x + (threadIdx.x >> 6));
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const int oc4 = ocb + (ow * NTW + nt) * 16 + kq * 4;
    if (oc4 >= P.OC) continue;
    const EpiCol col = epi_col_setup(P.ep, P.OW, P.OC, oc4);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int m = pw * 64 + mt * 16 + j;
      if (m < npx) {
        const int r = div_small(m, tw_magic), c = m - r * P.TW;
        if (r0 + r < P.PH && c0 + c < P.PW) {
          const epi_f4 o = epi_store4_tile(P.ep, col, et, r, c, acc[mt][nt], P.out);
          if (P.ep.y_amax) amax = abs_max4(amax, o);
        }
      }
    }
  }
  if (P.ep.y_amax) amax_commit(P.ep.y_amax, amax, blockIdx.x + (threadIdx.x >> 6), peeked);
}

// KS = 2 (small problems with every chunk staged up front): a second set of NPW*NOW waves takes the odd channel
// chunks -- two waves per SIMD on a block that is pure latency otherwise -- and the partial accumulators meet in LDS
// before the epilogue.
#ifndef BFD_OCC_SMALL
#define BFD_OCC_SMALL 1
#endif
// waves per SIMD the kernel is compiled for (0: the default bound of 2)
template <int NTW, int NPW, int NOW, int NP, int PF, int KS>
constexpr int bfd_occ() {
  if (BFD_OCC3 && NP == 3 && NTW == 2 && NPW == 2 && NOW == 2 && KS == 1 && PF == 1) return 3;  // exactly <2,2,2,3,1,1>
  if (BFD_OCC_SMALL && NPW == 1 && NOW == 4 && KS == 1) return NP == 2 ? 4 : 3;               // <1,1,4,NP,*,1>
  return 0;
}

// The kernel body; `bx` = the block's tile index (blockIdx.x of a single-phase launch).
template <int NTW, int NPW, int NOW, int NP, int PF, int KS, bool F16 = false, int OCCX = 0>
__device__ __forceinline__ void bfd_body(const BfdParams& B, const int bx) {
  static_assert(!F16 || NP == 2, "f16x3 has two planes");
  constexpr int NTHR = 64 * NPW * NOW * KS;
  constexpr bool TEPI = NPW == 1;
  // the bf16x6 4-wave block (the training forward of every 64-channel layer at benchmark batch sizes) is compiled for
  // 3 waves per SIMD (launch bound 168 VGPRs; it took 233 = 2 waves per SIMD): its halo staging keeps 2 pixels per
  // thread in flight instead of 6, which is what the register budget was spent on.  VDSR step 8.30 -> 7.96 ms, EDSR
  // 6.66 -> 6.62 ms (BFD_OCC3=0 restores the old build)
  // the 4-wave 64-pixel block without the K split (2 .. 8 tiles per CU: the up-sampler and the discriminator's middle
  // layers at 16 patches) likewise: 187 / 208 -> 116 / 137 VGPRs = 4 / 3 resident blocks per CU instead of 2
  constexpr bool OCC3 = bfd_occ<NTW, NPW, NOW, NP, PF, KS>() != 0 || OCCX != 0;
  constexpr int SIT = OCC3 ? 2 : BFD_STAGE_IT;
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
  const MfmaConvParams& P = B.P;
  uint4* hal = smem4;  // [NP][4][NPIXp]
  const int tid = threadIdx.x, lane = tid & 63, wave0 = tid >> 6;
  const int kgrp = wave0 / (NPW * NOW), wave = wave0 - kgrp * (NPW * NOW);  // K-split group, wave inside it
  const int pw = wave % NPW, ow = wave / NPW;
  const int col = lane & 15, kq = lane >> 4;
  const int j = (int)((B.perm >> (4 * col)) & 15);   // this lane's pixel within a 16-pixel M tile
  int b = bx;
  const int txi = b % P.tiles_x;
  b /= P.tiles_x;
  const int tyi = b % P.tiles_y;
  const int n = b / P.tiles_y;
  const int r0 = tyi * P.TH, c0 = txi * P.TW;
  const int ocbi = blockIdx.y;
  const int ocb = ocbi * 64;
  const int npx = P.TH * P.TW;
  const int NB = B.NB;
  const int T = P.KHv * P.KWv;

  int hp[4];
  {
    const int tw_magic = div_small_magic(P.TW);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      int m = pw * 64 + mt * 16 + j;
      if (m >= npx) m = 0;
      const int r = div_small(m, tw_magic), c = m - r * P.TW;
      hp[mt] = (r * P.is) * P.HW + c * P.is + lds_goff(kq, B.NPIXp);
    }
  }
  f32x4 acc[4][NTW];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const bool wave_live = pw * 64 < npx;
  const int plane = 4 * B.NPIXp + 4;
  const int wlane = kq * NB + col + ow * NTW * 16;
  // f16x3: activation scale 2^kx from the running maximum of the input tensor, descale 2^-(kx + kw)
  float sx = 1.f, dsc = 1.f;
  if constexpr (F16) {
    const int kx = amax_scale_exp(amax_read(P.ep.x_amax));
    sx = exp2i(kx);
    dsc = exp2i(-kx) * B.w_descale[0];
  }

  // Tap walks in scalar registers (no LDS tables, no per-tap division): virtual tap (u, v) reads halo offset
  // u*HW + v and weight tap (wh0 + wdh*u)*KW_full + ww0 + wdw*v.  `Walk` is the prefetch head over the flat
  // (chunk, tap) sequence; the compute loop keeps its own (toff, tv).
  const int wtap0 = P.wh0 * P.KW_full + P.ww0;
  const int wtap_row = P.wdh * P.KW_full - P.wdw * P.KWv;  // extra weight-tap step at a row wrap
  struct Walk {
    int cc, t, tv, wt;
  };
  auto walk_next = [&](Walk& h) {
    h.wt += P.wdw;
    if (++h.tv == P.KWv) {
      h.tv = 0;
      h.wt += wtap_row;
    }
    if (++h.t == T) {
      h.t = 0;
      h.tv = 0;
      h.wt = wtap0;
      h.cc += KS;
    }
  };
  // filter fragments of the head position -> registers (past the end the walk re-reads the last chunk: valid
  // memory, never used)
  auto load_b = [&](const Walk& h, uint4 (&dst)[NP][NTW]) {
    if (SRK_KDBG(B.dbg) & 8) return;
    const int hc = h.cc < B.ICc ? h.cc : B.ICc - 1;
    const size_t slot = (size_t)(h.wt * B.ICc + hc) * B.OCb + ocbi;
    const uint4* w = B.wq + slot * (size_t)(8 * NB) + wlane;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      dst[0][nt] = w[nt * 16];
      dst[1][nt] = w[4 * NB + nt * 16];
    }
    if (NP == 3) {
      const uint4* w3 = B.wq3 + slot * (size_t)(4 * NB) + wlane;
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) dst[NP - 1][nt] = w3[nt * 16];
    }
  };
  // one tap: A fragments from LDS, 3 or 6 MFMA passes against the given filter fragments
  auto tap_mfma = [&](const uint4* halc, int toff, const uint4 (&bf)[NP][NTW]) {
    if (wave_live && !(SRK_KDBG(B.dbg) & 4)) {
      const uint4* hb = halc + toff;
      uint4 a[NP][4];
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a[p][mt] = hb[hp[mt] + p * plane];
      // smallest products first; every pass runs over 4*NTW independent accumulators
      // TEPI (single pixel-wave blocks = the small-problem configuration): the product runs transposed (A = filter,
      // B = pixels; the fragment layouts of the two operands are the same), so that a lane ends up with 4 consecutive
      // output channels of one pixel and stores them directly (bfd_epilogue_direct)
#define SRK_BFD_PASS(pa, pb)                                                              \
  _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) _Pragma("unroll") for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = \
      TEPI ? mfma16x<F16>(bf[pb][nt], a[pa][mt], acc[mt][nt]) : mfma16x<F16>(a[pa][mt], bf[pb][nt], acc[mt][nt]);
      if (NP == 3) {
        SRK_BFD_PASS(NP - 1, 0)
        SRK_BFD_PASS(0, NP - 1)
        SRK_BFD_PASS(1, 1)
      }
      SRK_BFD_PASS(1, 0)
      SRK_BFD_PASS(0, 1)
      SRK_BFD_PASS(0, 0)
#undef SRK_BFD_PASS
    }
  };

  if (T > 0) {
    uint4 bq[PF + 1][NP][NTW];  // bq[0] = current tap, bq[1..PF] = the following taps
    Walk head{kgrp, 0, 0, wtap0};
#pragma unroll
    for (int d = 0; d < PF; ++d) {
      load_b(head, bq[d]);
      walk_next(head);
    }
    const int cstride = NP * plane;  // uint4 per staged chunk
    // Rounds of `cpr` chunks between barriers: 1 chunk, 2 chunks (K-split: one per wave group), or all of them (allc,
    // a single round so that the rotation below runs across chunk boundaries).  Group kgrp computes the chunks
    // kgrp, kgrp + KS, ... of its round.
    const int cpr = B.cpr;
    const int nseg = (B.ICc + cpr - 1) / cpr;
    const int cadv = KS * cstride;
    int ct = 0, ctv = 0, cbase = kgrp * cstride, coff = cbase;  // compute walk: halo offset of the current tap
    auto cwalk_next = [&]() {
      ++coff;
      if (++ctv == P.KWv) {
        ctv = 0;
        coff += P.HW - P.KWv;
      }
      if (++ct == T) {
        ct = 0;
        ctv = 0;
        cbase += cadv;
        coff = cbase;
      }
    };
    for (int seg = 0; seg < nseg; ++seg) {
      const int cfirst = seg * cpr;
      const int cend = cfirst + cpr < B.ICc ? cfirst + cpr : B.ICc;
      if (seg) __syncthreads();  // previous round's halo fully consumed
      for (int c2 = cfirst; c2 < cend && !(SRK_KDBG(B.dbg) & 1); ++c2) {
        if (P.mask_y)
          bfd_stage_halo_t<true, NTHR, NP, SIT, F16>(B, hal + (c2 - cfirst) * cstride, n, r0, c0, c2 * 32, sx);
        else
          bfd_stage_halo_t<false, NTHR, NP, SIT, F16>(B, hal + (c2 - cfirst) * cstride, n, r0, c0, c2 * 32, sx);
      }
      __syncthreads();
      const int mine = cfirst + kgrp < cend ? (cend - cfirst - kgrp + KS - 1) / KS : 0;  // chunks of this group in the round
      const int seg_len = mine * T;
      cbase = kgrp * cstride;
      coff = cbase;
      int it = 0;
      if (BFD_ROT && NTW * NP <= 6) {
        // rotate through the PF+1 register sets instead of shifting them: after PF+1 taps the roles are back
        // where they started (wider tiles spill when unrolled like this and take the shifting loop below)
        for (; it + PF + 1 <= seg_len; it += PF + 1) {
#pragma unroll
          for (int k = 0; k <= PF; ++k) {
            load_b(head, bq[(k + PF) % (PF + 1)]);
            walk_next(head);
            tap_mfma(hal, coff, bq[k]);
            cwalk_next();
          }
        }
      }
      for (; it < seg_len; ++it) {
        load_b(head, bq[PF]);
        walk_next(head);
        tap_mfma(hal, coff, bq[0]);
        cwalk_next();
#pragma unroll
        for (int d = 0; d < PF; ++d)
#pragma unroll
          for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) bq[d][p][nt] = bq[d + 1][p][nt];
      }
    }
  }
  if (SRK_KDBG(B.dbg) & 2) {
    if (acc[0][0][0] == 123.456f) P.out[0] = 1.f;  // keep the accumulators live
    return;
  }
  if (KS > 1) {
    // partial sums of the odd-chunk group -> LDS (the halo is dead) -> added by the even-chunk group, in a fixed order
    f32x4* red = reinterpret_cast<f32x4*>(smem4) + (size_t)wave * (4 * NTW * 64) + lane;
    __syncthreads();
    if (kgrp == 1) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) red[(mt * NTW + nt) * 64] = acc[mt][nt];
    }
    __syncthreads();
    if (kgrp == 0) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] += red[(mt * NTW + nt) * 64];
    }
  }
  if constexpr (F16) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] *= dsc;
  }
  if constexpr (TEPI)
    bfd_epilogue_direct<NTW>(P, B.perm, acc, n, r0, c0, ocb, pw, ow, lane, kgrp == 0);
  else
    bfd_epilogue<NTW, NPW, NOW>(P, B.perm, reinterpret_cast<float*>(smem4), acc, n, r0, c0, ocb, pw, ow, lane, kgrp == 0);
}

template <int NTW, int NPW, int NOW, int NP, int PF, int KS, bool F16 = false, int OCCX = 0>
__global__ __launch_bounds__(64 * NPW * NOW * KS, OCCX ? OCCX : (bfd_occ<NTW, NPW, NOW, NP, PF, KS>() ? bfd_occ<NTW, NPW, NOW, NP, PF, KS>() : 2)) void k_conv_bfd(
    BfdParams B) {
  bfd_body<NTW, NPW, NOW, NP, PF, KS, F16, OCCX>(B, (int)blockIdx.x);
}

// All phases of a strided TRANS gather in ONE launch (round 6).  for_each_phase turns a stride-s transposed gather (the data
// gradient of a stride-s conv, a deconv forward) into s * s stride-1 problems with their own tap subsets; they used to be
// s * s launches -- on the SRGAN discriminator's deep layers (512 -> 512 at 8 x 8 per phase, 16 patches) four latency chains
// of 128 blocks each, back to back: 113 us.  Here the phases' parameter blocks travel together in the kernel arguments, a
// block finds its phase from its index (wave-uniform: scalar loads with a register offset) and the chains run side by side.
constexpr int BFD_MAXPH = 4;
struct BfdMulti {
  BfdParams ph[BFD_MAXPH];
  int start[BFD_MAXPH + 1];   // first block of phase p
  int nph;
};
template <int NTW, int NPW, int NOW, int NP, int PF, int KS, bool F16 = false, int OCCX = 0>
__global__ __launch_bounds__(64 * NPW * NOW * KS, OCCX ? OCCX : (bfd_occ<NTW, NPW, NOW, NP, PF, KS>() ? bfd_occ<NTW, NPW, NOW, NP, PF, KS>() : 2)) void k_conv_bfd_mp(
    BfdMulti M) {
  int p = 0;
#pragma unroll
  for (int q = 1; q < BFD_MAXPH; ++q)
    if (q < M.nph && (int)blockIdx.x >= M.start[q]) p = q;
  p = __builtin_amdgcn_readfirstlane(p);
  bfd_body<NTW, NPW, NOW, NP, PF, KS, F16, OCCX>(M.ph[p], (int)blockIdx.x - M.start[p]);
}

// ---------------------------------------------------------------------------------------------
// Host
// ---------------------------------------------------------------------------------------------
// LDS plan of the operand reads (round 4).  ds_read_b128 serves the lane sets {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}
// (+ 32) in one clock each when their 16 slots differ mod 16; a set mixes 8 lanes of channel group kq with 8 lanes of
// group kq + 1, whose slots lie D = stride mod 16 further (lds_goff).  Pixel i of an M tile sits at slot
// (m / TW) * is * HW + (m % TW) * is: 16 consecutive slots when a tile row is >= 16 pixels wide (D = 0, identity), but the
// 8 x 8 tiles of the small-problem blocks put pixels 8-15 one halo row further, where they alias pixels 0-7 -- every
// operand read took two clocks per set (SQ_LDS_BANK_CONFLICT: 19-22 % of the cycles of k_conv_bfd<1,1,4,*>).  This
// searches D and the split of the 16 pixels over the two lane classes for the fewest clocks over the block's M tiles
// (12870 splits x 16 strides, cached per geometry); conv_res2.hip documents the two hand-derived instances.
static void bfd_lds_plan(int TW, int is, int HW, int npx, int npix, int NPW, int& stride, unsigned long long& perm) {
  struct Key {
    int TW, is, HW, npx, NPW;
    bool operator<(const Key& o) const {
      return std::tie(TW, is, HW, npx, NPW) < std::tie(o.TW, o.is, o.HW, o.npx, o.NPW);
    }
  };
  struct Val {
    int D;
    unsigned long long perm;
  };
  static std::mutex mu;
  static std::map<Key, Val> cache;
  const Key key{TW, is, HW, npx, NPW};
  Val v{};
  bool hit = false;
  {
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) {
      v = it->second;
      hit = true;
    }
  }
  if (!hit) {
    // the search runs OUTSIDE the lock (other threads -- DP ranks of one process, the autograd thread -- keep dispatching;
    // two threads that miss on the same geometry compute the same plan twice)
    const int ntile = NPW * 4;
    std::vector<int> pos((size_t)ntile * 16, -1);
    int live = 0;
    for (int t = 0; t < ntile; ++t)
      for (int i = 0; i < 16; ++i) {
        const int m = t * 16 + i;
        if (m < npx) pos[(size_t)t * 16 + i] = (m / TW) * is * HW + (m % TW) * is;
        if (i == 0 && m < npx) live = t + 1;
      }
    static const int colA[8] = {0, 1, 2, 3, 12, 13, 14, 15}, colB[8] = {4, 5, 6, 7, 8, 9, 10, 11};
    long best_cost = -1;
    int best_D = 0;
    unsigned best_mask = 0;
    const bool off = env_int("SRK_BFD_LDS_PLAN", 1) == 0;
    auto cost_of = [&](int D, unsigned mk) {
      long cost = 0;
      for (int t = 0; t < live; ++t) {
        int cnt0[16] = {0}, cnt1[16] = {0};
        int m0 = 0, m1 = 0;
        for (int i = 0; i < 16; ++i) {
          const int ps = pos[(size_t)t * 16 + i];
          if (ps < 0) continue;   // dead pixels all read slot 0 of their group: one address, no conflict of their own
          const bool a = (mk >> i) & 1;
          const int r0 = (a ? ps : ps + D) & 15, r1 = (a ? ps + D : ps) & 15;
          m0 = std::max(m0, ++cnt0[r0]);
          m1 = std::max(m1, ++cnt1[r1]);
        }
        cost += m0 + m1;
      }
      return cost;
    };
    // candidate order: the round-3 layout first (ties keep it), then the rest
    for (int di = 0; di < (off ? 1 : 16) && best_cost != 2L * live; ++di) {
      const int D = di == 0 ? 0 : (di == 1 ? 8 : (di <= 8 ? di - 1 : di));   // 0, 8, 1 .. 7, 9 .. 15
      // mask: pixels on the lane class A; the first candidate is the identity (pixels {0-3, 12-15}), then every other
      // 8-of-16 subset in increasing order (Gosper's hack: 12870 masks, not 65536 popcount tests)
      for (unsigned mk = 0xF00Fu, first = 1; mk < 65536u;) {
        if (first || mk != 0xF00Fu) {
          const long cost = cost_of(D, mk);
          if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best_D = D;
            best_mask = mk;
            if (best_cost == 2L * live) break;   // one clock per set everywhere
          }
        }
        if (off) break;
        if (first) {
          first = 0;
          mk = 0x00FFu;
          continue;
        }
        const unsigned c = mk & (0u - mk), r = mk + c;
        mk = (((r ^ mk) >> 2) / c) | r;
      }
    }
    v.D = best_D;
    v.perm = 0;
    int na = 0, nb = 0;
    for (int i = 0; i < 16; ++i) {
      const int colx = ((best_mask >> i) & 1) ? colA[na++] : colB[nb++];
      v.perm |= (unsigned long long)i << (4 * colx);
    }
    std::lock_guard<std::mutex> lock(mu);
    cache[key] = v;
  }
  stride = ((npix - v.D + 15) & ~15) + v.D;   // smallest value >= npix that is D (mod 16)
  perm = v.perm;
}

template <int NTW, int NPW, int NOW, int NP, int PF, bool F16 = false, int OCCX = 0>
static int bfd_launch(BfdParams B, int budget_bytes, hipStream_t s) {
  note_amax_written(B.P.ep.y_amax != nullptr);   // both epilogues keep the running maximum of what they store
  MfmaConvParams& P = B.P;
  const int maxpix = 64 * NPW;
  TilePick best{};
  const int kh = P.KHv > 0 ? P.KHv : 1, kw = P.KWv > 0 ? P.KWv : 1;
  bool ok = pick_tile(maxpix, P.PH, P.PW, P.is, kh, kw, NP * 16, budget_bytes / 4 - 16 * NP * 16, best);
  if (!ok || best.eff < 0.6) {  // huge halos (large kernels / strides): one block per CU
    TilePick big{};
    if (pick_tile(maxpix, P.PH, P.PW, P.is, kh, kw, NP * 16, (156 * 1024) / 4 - 16 * NP * 16, big) &&
        (!ok || big.eff > best.eff * 1.2)) {
      best = big;
      ok = true;
    }
  }
  if (!ok) {
    set_error("conv_bfd: no tile fits LDS");
    return SRK_ERR_UNSUPPORTED;
  }
  P.TH = best.TH; P.TW = best.TW; P.tiles_y = best.tiles_y; P.tiles_x = best.tiles_x; P.HH = best.HH; P.HW = best.HW;
  bfd_lds_plan(P.TW, P.is, P.HW, P.TH * P.TW, P.HH * P.HW, NPW, B.NPIXp, B.perm);
  size_t lds = (size_t)NP * (4 * B.NPIXp + 4) * 16;
  B.allc = 0;
  B.cpr = 1;
  const size_t lds_chunk = lds;
  if (NPW == 1 && B.ICc > 1 && lds * B.ICc <= 64 * 1024) {  // small-problem blocks: stage all chunks at once
    B.allc = 1;
    B.cpr = B.ICc;
    lds *= B.ICc;
  }
  const size_t epi_bytes = (size_t)NPW * 32 * BFD_EPI_STRIDE * sizeof(float);
  if (lds < epi_bytes) lds = epi_bytes;
  dim3 grid((unsigned)((size_t)P.tiles_x * P.tiles_y * P.N), B.OCb);
  if constexpr (NPW == 1) {
    // small-problem blocks with >= 2 staged chunks: split the chunks over two wave groups (SRK_BFD_KSPLIT=0: off)
    const int ksplit = env_int("SRK_BFD_KSPLIT", 1);
    const size_t red_bytes = (size_t)NOW * 4 * NTW * 64 * 16;
    // only while the grid leaves the CUs with one block each: with two resident blocks the other block already hides the
    // latency and the split just adds the reduction (B = 32 EDSR shard: 3.28 -> 3.55 ms with it, B = 16: 2.63 -> 2.37 ms)
    if (ksplit && B.ICc >= 2 && (B.allc || 2 * lds_chunk <= (size_t)150 * 1024) &&
        ((long)grid.x * grid.y <= kNumCU + kNumCU / 4 || ksplit > 1)) {
      if (!B.allc) {  // many chunks (deep layers of the SRGAN discriminator): two chunks per barrier round, one per group
        B.cpr = 2;
        lds = 2 * lds_chunk;
        if (lds < epi_bytes) lds = epi_bytes;
      }
      if (lds < red_bytes) lds = red_bytes;
      static LdsLimit lim2;
      lim2.ensure(reinterpret_cast<const void*>(&k_conv_bfd<NTW, NPW, NOW, NP, PF, 2, F16, OCCX>), lds);
      if (SRK_KDBG(B.dbg) & 32)
        fprintf(stderr, "[srk] k_conv_bfd<%d,%d,%d,%d,%d> K-split 2: lds %zu B, grid %u x %u, tile %dx%d halo %dx%d\n", NTW, NPW,
                NOW, NP, PF, lds, grid.x, grid.y, P.TH, P.TW, P.HH, P.HW);
      note_kernel("k_conv_bfd<%d,%d,%d,%d,%d,2%s>", NTW, NPW, NOW, NP, PF, F16 ? ",f16" : "");
      hipLaunchKernelGGL((k_conv_bfd<NTW, NPW, NOW, NP, PF, 2, F16, OCCX>), grid, dim3(64 * NPW * NOW * 2), lds, s, B);
      return check_launch("conv_bfd");
    }
  }
  static LdsLimit lim;
  const void* fn = reinterpret_cast<const void*>(&k_conv_bfd<NTW, NPW, NOW, NP, PF, 1, F16, OCCX>);
  lim.ensure(fn, lds);
  if (SRK_KDBG(B.dbg) & 32) {
    int nb = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64 * NPW * NOW, lds);
    fprintf(stderr, "[srk] k_conv_bfd<%d,%d,%d,%d,%d>: lds %zu B, grid %u x %u, occupancy %d blocks/CU, tile %dx%d halo %dx%d\n",
            NTW, NPW, NOW, NP, PF, lds, grid.x, grid.y, nb, P.TH, P.TW, P.HH, P.HW);
  }
  note_kernel("k_conv_bfd<%d,%d,%d,%d,%d,1%s%s>", NTW, NPW, NOW, NP, PF, F16 ? ",f16" : "", OCCX == 3 ? ",occ3" : (OCCX == 4 ? ",occ4" : ""));
  hipLaunchKernelGGL((k_conv_bfd<NTW, NPW, NOW, NP, PF, 1, F16, OCCX>), grid, dim3(64 * NPW * NOW), lds, s, B);
  return check_launch("conv_bfd");
}

// The phases of a stride-2 TRANS gather as one launch of the small-problem block (64 pixels x NOW channel-waves): every
// phase keeps the tile, LDS plan and staging mode bfd_launch would give it; the K-split decision is taken on the blocks of
// the WHOLE launch.  -1: not taken (the caller launches the phases one by one).
template <int NOW, int NP>
static int bfd_launch_small_multi(const MfmaConvParams* phases, int nph, const BfdParams& base, hipStream_t s) {
  constexpr int NTW = 1, NPW = 1, PF = 2;
  constexpr int SMALL = 36 * 1024;
  if (nph < 2 || nph > BFD_MAXPH) return -1;
  BfdMulti M{};
  M.nph = nph;
  size_t lds_chunk[BFD_MAXPH], lds[BFD_MAXPH];
  long total = 0;
  const size_t epi_bytes = (size_t)NPW * 32 * BFD_EPI_STRIDE * sizeof(float);
  for (int i = 0; i < nph; ++i) {
    BfdParams B = base;
    B.P = phases[i];
    MfmaConvParams& P = B.P;
    TilePick best{};
    const int kh = P.KHv > 0 ? P.KHv : 1, kw = P.KWv > 0 ? P.KWv : 1;
    bool ok = pick_tile(64 * NPW, P.PH, P.PW, P.is, kh, kw, NP * 16, SMALL / 4 - 16 * NP * 16, best);
    if (!ok || best.eff < 0.6) {
      TilePick big{};
      if (pick_tile(64 * NPW, P.PH, P.PW, P.is, kh, kw, NP * 16, (156 * 1024) / 4 - 16 * NP * 16, big) &&
          (!ok || big.eff > best.eff * 1.2)) {
        best = big;
        ok = true;
      }
    }
    if (!ok) return -1;
    P.TH = best.TH; P.TW = best.TW; P.tiles_y = best.tiles_y; P.tiles_x = best.tiles_x; P.HH = best.HH; P.HW = best.HW;
    bfd_lds_plan(P.TW, P.is, P.HW, P.TH * P.TW, P.HH * P.HW, NPW, B.NPIXp, B.perm);
    size_t l = (size_t)NP * (4 * B.NPIXp + 4) * 16;
    B.allc = 0;
    B.cpr = 1;
    lds_chunk[i] = l;
    if (B.ICc > 1 && l * B.ICc <= 64 * 1024) {
      B.allc = 1;
      B.cpr = B.ICc;
      l *= B.ICc;
    }
    if (l < epi_bytes) l = epi_bytes;
    lds[i] = l;
    M.ph[i] = B;
    M.start[i] = (int)total;
    total += (long)P.tiles_x * P.tiles_y * P.N;
    if (total >= (1L << 30)) return -1;
  }
  M.start[nph] = (int)total;
  const int OCb = base.OCb;
  // K split over two wave groups while the whole launch leaves the CUs with about one block each (as bfd_launch)
  const int ksplit = env_int("SRK_BFD_KSPLIT", 1);
  bool ks2 = ksplit && base.ICc >= 2 && (total * OCb <= kNumCU + kNumCU / 4 || ksplit > 1);
  for (int i = 0; i < nph && ks2; ++i) ks2 = M.ph[i].allc || 2 * lds_chunk[i] <= (size_t)150 * 1024;
  size_t lds_max = 0;
  const size_t red_bytes = (size_t)NOW * 4 * NTW * 64 * 16;
  for (int i = 0; i < nph; ++i) {
    if (ks2) {
      if (!M.ph[i].allc) {
        M.ph[i].cpr = 2;
        lds[i] = 2 * lds_chunk[i];
        if (lds[i] < epi_bytes) lds[i] = epi_bytes;
      }
      if (lds[i] < red_bytes) lds[i] = red_bytes;
    }
    if (lds[i] > lds_max) lds_max = lds[i];
  }
  note_amax_written(base.P.ep.y_amax != nullptr);
  dim3 grid((unsigned)total, (unsigned)OCb);
  if (ks2) {
    static LdsLimit lim2;
    lim2.ensure(reinterpret_cast<const void*>(&k_conv_bfd_mp<NTW, NPW, NOW, NP, PF, 2>), lds_max);
    note_kernel("k_conv_bfd_mp<%d,%d,%d,%d,%d,2>x%d", NTW, NPW, NOW, NP, PF, nph);
    hipLaunchKernelGGL((k_conv_bfd_mp<NTW, NPW, NOW, NP, PF, 2>), grid, dim3(64 * NPW * NOW * 2), lds_max, s, M);
  } else {
    static LdsLimit lim;
    lim.ensure(reinterpret_cast<const void*>(&k_conv_bfd_mp<NTW, NPW, NOW, NP, PF, 1>), lds_max);
    note_kernel("k_conv_bfd_mp<%d,%d,%d,%d,%d,1>x%d", NTW, NPW, NOW, NP, PF, nph);
    hipLaunchKernelGGL((k_conv_bfd_mp<NTW, NPW, NOW, NP, PF, 1>), grid, dim3(64 * NPW * NOW), lds_max, s, M);
  }
  return check_launch("conv_bfd_mp");
}

bool conv_bfd_gather_supported(const GatherConv& g, const Epi& ep) {
  (void)ep;
  if (g.OC < 8 || g.IC < 8) return false;
  if (g.KH * g.KW > 32 * 32) return false;
  if ((long)g.N * g.OH * g.OW > (1L << 30)) return false;
  if ((long)g.IH * g.IW * g.IC >= (1L << 30)) return false;  // 32-bit in-image offsets in the staging loops
  return true;
}

static int bfd_dbg() {
  return SRK_EXP_INT("SRK_DBG", 0);
}

// small problem: fewer pixels than two resident 256-pixel tiles per CU -> 64-pixel blocks whose waves
// split the output channels
bool conv_bfd_small_problem(const GatherConv& g) {
  const char* e = env_str("SRK_BFD_SMALL");  // tests: 0 / 1 force the large / small block configuration
  if (e) return atoi(e) != 0;
  const int st = g.trans ? g.stride : 1;
  const long px = (long)g.N * ((g.OH + st - 1) / st) * ((g.OW + st - 1) / st) * ((g.OC + 63) / 64);
  return px < 256L * 2 * kNumCU;
}

template <int NP, bool F16 = false>
static int bfd_launch_phase(MfmaConvParams P, const uint4* wq, const uint4* wq3, bool small, hipStream_t s,
                            const float* w_descale = nullptr) {
  BfdParams B{};
  B.w_descale = w_descale;
  const int NT = P.OC >= 64 ? 4 : (P.OC + 15) / 16;
  B.NB = NT * 16;
  B.ICc = (P.IC + 31) / 32;
  B.OCb = (P.OC + 63) / 64;
  B.wq = wq;
  B.wq3 = wq3;
  B.dbg = bfd_dbg();
  B.P = P;
  constexpr int BIG = kLdsBudgetBytes, SMALL = 36 * 1024;
  if (small) {
    switch (NT) {
      case 1: return bfd_launch<1, 1, 1, NP, 2, F16>(B, SMALL, s);
      case 2: return bfd_launch<1, 1, 2, NP, 2, F16>(B, SMALL, s);
      case 3: return bfd_launch<1, 1, 3, NP, 2, F16>(B, SMALL, s);
      default: return bfd_launch<1, 1, 4, NP, 2, F16>(B, SMALL, s);  // (filter prefetch depth 5 / 8 measured: no gain)
    }
  }
  switch (NT) {
    case 1: return bfd_launch<1, 4, 1, NP, 2, F16>(B, BIG, s);
    case 2: return bfd_launch<2, 4, 1, NP, 2, F16>(B, BIG, s);
    case 3: return bfd_launch<3, 4, 1, NP, 1, F16>(B, BIG, s);
    default:
      if (NP == 3) return bfd_launch<2, 2, 2, NP, 1, F16>(B, BIG, s);
      if constexpr (F16) {
        // experiment switch (SRK_BFD_F16_CFG): 0 = the bf16x3 block (4 pixel-waves x 64 channels, 2 waves per SIMD),
        // 1 / 2 = the bf16x6 block shape (2 x 2 waves of 64 px x 32 ch) compiled for 3 / 4 waves per SIMD
        const int cfg = env_int("SRK_BFD_F16_CFG", 1);
        if (cfg == 1) return bfd_launch<2, 2, 2, NP, 1, true, 3>(B, BIG, s);
        if (cfg == 2) return bfd_launch<2, 2, 2, NP, 1, true, 4>(B, BIG, s);
      } else if constexpr (NP == 2) {
        const int cfg3 = SRK_EXP_INT("SRK_BFD_X3_CFG", 0);   // experiment: the same block for bf16x3
        if (cfg3 == 1) return bfd_launch<2, 2, 2, NP, 1, false, 3>(B, BIG, s);
      }
      return bfd_launch<4, 4, 1, NP, 1, F16>(B, BIG, s);
  }
}

// planes = 2: bf16x3, planes = 3: bf16x6, planes = 4: f16x3 (forward buffers only; ep.x_amax required).  `wp` is the
// packed filter buffer of srk_pack_weight_*.
int conv_bfd_gather(const GatherConv& g, const float* in, const float* wp, float* out, const Epi& ep,
                    const float* mask_y, float mask_slope, int planes, hipStream_t s) {
  const size_t elems = (size_t)g.KH * g.KW * g.IC * g.OC;
  const char* base = reinterpret_cast<const char*>(wp) + bf3_prepared_offset(elems);
  const uint4* wq = reinterpret_cast<const uint4*>(base);
  const uint4* wq3 = reinterpret_cast<const uint4*>(base + bf3_main_bytes(g.IC, g.OC, g.KH * g.KW));
  const bool small = conv_bfd_small_problem(g);
  if (planes == 4) {
    if (!ep.x_amax) {
      set_error("conv_bfd: the f16x3 kernels need srk_epilogue.x_amax");
      return SRK_ERR_BAD_ARG;
    }
    const char* fbase = base + f16_section_offset(g.IC, g.OC, g.KH * g.KW);
    const uint4* wh = reinterpret_cast<const uint4*>(fbase);
    const float* trailer = reinterpret_cast<const float*>(fbase + bf3_main_bytes(g.IC, g.OC, g.KH * g.KW));
    return for_each_phase(g, in, wp, out, ep, mask_y, mask_slope, [&](const MfmaConvParams& P) {
      return bfd_launch_phase<2, true>(P, wh, nullptr, small, s, trailer);
    });
  }
  // strided TRANS gathers on the small-problem block: every phase in ONE launch (k_conv_bfd_mp); SRK_BFD_MP=0: one by one
  if (g.trans && g.stride == 2 && small && g.OC >= 64 && env_int("SRK_BFD_MP", 1) != 0) {
    MfmaConvParams phases[BFD_MAXPH];
    int nph = 0;
    bool fits = true;
    const int rc0 = for_each_phase(g, in, wp, out, ep, mask_y, mask_slope, [&](const MfmaConvParams& P) {
      if (nph < BFD_MAXPH) phases[nph] = P; else fits = false;
      ++nph;
      return (int)SRK_OK;
    });
    if (rc0 == SRK_OK && fits && nph >= 2) {
      BfdParams base{};
      base.NB = 64;
      base.ICc = (g.IC + 31) / 32;
      base.OCb = (g.OC + 63) / 64;
      base.wq = wq;
      base.wq3 = wq3;
      base.dbg = bfd_dbg();
      const int rc = planes == 3 ? bfd_launch_small_multi<4, 3>(phases, nph, base, s) : bfd_launch_small_multi<4, 2>(phases, nph, base, s);
      if (rc != -1) return rc;
    }
  }
  return for_each_phase(g, in, wp, out, ep, mask_y, mask_slope, [&](const MfmaConvParams& P) {
    return planes == 3 ? bfd_launch_phase<3>(P, wq, wq3, small, s) : bfd_launch_phase<2>(P, wq, wq3, small, s);
  });
}

}  // namespace srk
