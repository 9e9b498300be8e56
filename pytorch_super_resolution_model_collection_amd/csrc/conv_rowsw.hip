// Wave-specialised persistent form of the row-packed first-layer kernel (k_conv_bf3_rows: Cin <= 4, the K dimension
// of an MFMA step = 8 taps of one kernel row x 4 channels) for problems of benchmark size (the 3 -> 64 5x5 layer of
// ESPCN on 64 x 256x256: 16384 tiles, 1.04 GB of output).
//
// Why: the per-tile kernel's phases do not overlap.  SRK_ROWS_DBG ablations on that layer (320 us): without the halo
// loads 232 us (the 19 MB input costs 90 us of pure latency: every block waits for its own few loads), without the
// epilogue 170, without the MFMAs 243, loop skeleton + per-step filter copies alone 63; a linear fill of the output
// runs in 180 us on the same box (tools/micro/store_pattern.hip: 64-byte segments per pixel store as fast as whole
// lines).  Here, as in k_conv_bfw:
//   * one persistent block per CU, the filter planes of its 32-channel slice copied to LDS once;
//   * 4 producer waves stage the halo of tile s+1 (one thread per pixel: <= 4 floats -> {h, l} 16-bit planes, 16 bytes
//     per pixel) while 8 consumer waves run the MFMAs of tile s; the loads of tile s+2 are already in flight;
//   * consumers (32 pixels x 32 channels each, transposed product: a lane holds 4 consecutive channels of a pixel)
//     store the finished tile from registers, one or two 16-byte stores between the K steps of the next tile.
// Output-channel slices (nsl = OC / 32) sit on neighbouring blocks of one XCD; the input is so small that staging it
// once per slice costs nothing.  Arithmetic and summation order are those of k_conv_bf3_rows (bf16x3 / f16x3).
#include "srk_common.h"
#include "conv_problem.h"
#include "conv_tile.h"
#include "bf16_frag.h"
#include <stdlib.h>
#include <type_traits>

namespace srk {

template <int I0, int I1, typename F>
__device__ __forceinline__ void rw_static_for(F&& f) {
  if constexpr (I0 < I1) {
    f(std::integral_constant<int, I0>{});
    rw_static_for<I0 + 1, I1>(f);
  }
}

constexpr int RW_IT = 2;  // staging register batches: halos of <= 512 * 2 pixels

struct RowswParams {
  MfmaConvParams P;
  const uint4* wq;         // row-packed planes [kh][ks][64-channel block][plane][group][NBfull] (bf16, or the fp16 section)
  const float* w_descale;  // F16: trailer {2^-kw, 2^kw} of the fp16 section
  int KS, NBfull, OCb, NPIXp, ntiles, nsl;
  unsigned out_bytes;
  int dbg;  // ablation (SRK_ROWSW_DBG): 1 no global loads, 2 no stores, 4 no MFMA loop, 16 no LDS commit
#ifdef SRK_ROWSR_PROF
  unsigned* prof;  // phase-clock build (tools/rowsr_prof.py): 8 counters per wave
#endif
};
#ifdef SRK_ROWSR_PROF
static unsigned* g_rowsr_prof = nullptr;
#define RS_T() ((unsigned)clock64())  /* s_memtime (gfx950 has no SHADER_CYCLES register) */
#define RS_ACC(i, a, b) pacc[i] += (b) - (a)
#else
#define RS_T() 0u
#define RS_ACC(i, a, b) do { } while (0)
#endif

// add-and-carry walk over the tiles first, first + step, ... of a block (wave-uniform, scalar registers): a 32-bit
// division is ~40 VALU instructions = 160 SIMD cycles per wave, and the per-stage decode had three of them
struct RowswWalk {
  int n, y, x;
  __device__ __forceinline__ void init(int tile, int tiles_x, int img_tiles) {
    n = tile / img_tiles;
    const int q = tile - n * img_tiles;
    y = q / tiles_x;
    x = q - y * tiles_x;
    n = __builtin_amdgcn_readfirstlane(n);
    y = __builtin_amdgcn_readfirstlane(y);
    x = __builtin_amdgcn_readfirstlane(x);
  }
  __device__ __forceinline__ void advance(int st_n, int st_y, int st_x, int tiles_x, int tiles_y) {
    x += st_x;
    y += st_y;
    n += st_n;
    if (x >= tiles_x) {
      x -= tiles_x;
      ++y;
    }
    if (y >= tiles_y) {
      y -= tiles_y;
      ++n;
    }
  }
};

template <int NTW, int QT, bool F16>
__global__ __launch_bounds__(512, 2) void k_conv_rowsw(RowswParams B) {
  constexpr bool WREG = NTW * QT <= 12;
  constexpr int NCW = 8, MTW = 2, NTHR = 512, NB = 16 * NTW;
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
  const MfmaConvParams& P = B.P;
  constexpr int wslot = 8 * NB;  // uint4 per K step: [plane 2][group 4][NB]
  uint4* wl = smem4;
  uint4* hal0 = smem4 + QT * wslot;  // 2 buffers x NPIXp pixels x {h: 4 x 16 bit, l: 4 x 16 bit}
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kq = lane >> 4;
  const int npx = P.TH * P.TW, npix = P.HH * P.HW;
  float sx = 1.f, dsc = 1.f;
  if constexpr (F16) {
    const int kx = amax_scale_exp(amax_read(P.ep.x_amax));
    sx = exp2i(kx);
    dsc = exp2i(-kx) * B.w_descale[0];
  }
  const int nsl = B.nsl;
  const int xcd = blockIdx.x & 7;
  const int sl = (blockIdx.x >> 3) % nsl, bi = (blockIdx.x >> 3) / nsl;
  for (int e = tid; e < QT * wslot; e += NTHR) {
    const int q = e / wslot, w = e - q * wslot;
    const int pg = w / NB, o = w - pg * NB;
    const int oc = sl * NB + o, ocb = oc / B.NBfull;
    wl[e] = B.wq[((size_t)q * B.OCb + ocb) * (size_t)(8 * B.NBfull) + pg * B.NBfull + (oc - ocb * B.NBfull)];
  }
  // pixels past the halo (the padded tap slots of the last row read them; they meet zero filter taps and must be finite)
  for (int e = tid; e < 2 * (B.NPIXp - npix); e += NTHR) {
    const int b = e / (B.NPIXp - npix), i = e - b * (B.NPIXp - npix);
    hal0[(size_t)b * B.NPIXp + npix + i] = make_uint4(0, 0, 0, 0);
  }
  // tiles of this block: XCD-aware contiguous ranges (see k_conv_bfw)
  const int nblk = gridDim.x;
  int first, count;
  const int nb_x = ((nblk + 7 - xcd) >> 3) / nsl;
  {
    const int per_x = B.ntiles >> 3, rem_x = B.ntiles & 7;
    const int tiles_x = per_x + (xcd < rem_x ? 1 : 0);
    const int start_x = xcd * per_x + (xcd < rem_x ? xcd : rem_x);
    first = start_x + bi;
    count = bi < tiles_x ? (tiles_x - bi + nb_x - 1) / nb_x : 0;
  }
  const int S = count;
  const int img_tiles = P.tiles_x * P.tiles_y;
  const int st_n = nb_x / img_tiles, st_y = (nb_x - st_n * img_tiles) / P.tiles_x, st_x = nb_x - st_n * img_tiles - st_y * P.tiles_x;
  RowswWalk wi, wc;  // tile of the next issue() / of the next compute stage
  wi.init(first, P.tiles_x, img_tiles);
  wc = wi;

  // ---- staging (every thread: pixels tid, tid + 512 of the halo): loads one stage ahead of the LDS commit ----------
  float pv[RW_IT][4];
  int hy0[RW_IT], hx0[RW_IT];
#pragma unroll
  for (int k = 0; k < RW_IT; ++k) {
    const int hq = tid + NTHR * k;
    hy0[k] = hq / P.HW;
    hx0[k] = hq - hy0[k] * P.HW;
  }
  // Loads are unconditional and branch-free (clamped addresses; out-of-image pixels are zeroed by a select at commit
  // time): a load under a divergent branch gets an s_waitcnt vmcnt(0) at the join, which serialises the memory
  // latency of every pixel slot into the stage.
  const size_t plane = (size_t)P.IH * P.IW;
  const size_t est = P.in_nchw ? plane : 1;  // element stride between the channels of a pixel
  const size_t e1 = P.IC > 1 ? est : 0, e2 = P.IC > 2 ? 2 * est : 0, e3 = P.IC > 3 ? 3 * est : 0;
  int vmask = 0;  // bit k: slot k is a pixel of the halo inside the image
  auto issue = [&]() {
    const int n = wi.n, iyb = wi.y * P.TH + P.iy0, ixb = wi.x * P.TW + P.ix0;
    wi.advance(st_n, st_y, st_x, P.tiles_x, P.tiles_y);
    if (SRK_KDBG(B.dbg) & 1) return;
    vmask = 0;
    const float* img = P.in + (size_t)n * P.IC * plane;
#pragma unroll
    for (int k = 0; k < RW_IT; ++k) {
      const int iy = iyb + hy0[k], ix = ixb + hx0[k];
      const bool ok = tid + NTHR * k < npix && (unsigned)iy < (unsigned)P.IH && (unsigned)ix < (unsigned)P.IW;
      vmask |= ok ? (1 << k) : 0;
      const int cy = min(max(iy, 0), P.IH - 1), cx = min(max(ix, 0), P.IW - 1);
      const size_t pix = (size_t)cy * P.IW + cx;
      const float* src = img + (P.in_nchw ? pix : pix * P.IC);
      pv[k][0] = src[0];
      pv[k][1] = src[e1];
      pv[k][2] = src[e2];
      pv[k][3] = src[e3];
    }
  };
  auto commit = [&](uint4* hal) {
    if (SRK_KDBG(B.dbg) & 16) return;
#pragma unroll
    for (int k = 0; k < RW_IT; ++k) {
      const int hq = tid + NTHR * k;
      if (hq < npix) {
        uint2 hu, lu;
        if constexpr (F16) {
          typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
          f16x4 h, l;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xs = (((vmask >> k) & 1) && e < P.IC) ? pv[k][e] * sx : 0.f;
            const _Float16 hh = (_Float16)xs;
            h[e] = hh;
            l[e] = (_Float16)(xs - (float)hh);
          }
          hu = __builtin_bit_cast(uint2, h);
          lu = __builtin_bit_cast(uint2, l);
        } else {
          typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
          bf16x4 h, l;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xv = (((vmask >> k) & 1) && e < P.IC) ? pv[k][e] : 0.f;
            const __bf16 hh = (__bf16)xv;
            h[e] = hh;
            l[e] = (__bf16)(xv - (float)hh);
          }
          hu = __builtin_bit_cast(uint2, h);
          lu = __builtin_bit_cast(uint2, l);
        }
        hal[hq] = make_uint4(hu.x, hu.y, lu.x, lu.y);
      }
    }
  };

  // ---- MFMA side ---------------------------------------------------------------------------------------------------
  const int pw = wave;
  int hp[MTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) {
    int m = pw * (16 * MTW) + mt * 16 + j;
    if (m >= npx) m = 0;
    const int r = m / P.TW, c = m - r * P.TW;
    hp[mt] = r * P.HW + c + 2 * kq;  // K slots kq*8 .. kq*8+7 = taps 2kq, 2kq+1 of the row x 4 channels
  }
  int wrow[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) wrow[nt] = kq * NB + nt * 16 + j;
  f32x4 bias4[NTW];
  int coff[NTW], poff[MTW], pix_ok[MTW];
  {
    const EpiTile e0 = epi_tile_setup(P, 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
      const int m = pw * (16 * MTW) + mt * 16 + j;
      const int r = m / P.TW, c = m - r * P.TW;
      pix_ok[mt] = m < npx ? ((r << 16) | c) : -1;
      poff[mt] = r * e0.RS + c * e0.CS;
    }
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const EpiCol cl = epi_col_setup(P.ep, P.OW, P.OC, sl * NB + nt * 16 + kq * 4);
      coff[nt] = (int)cl.off_oc;
      bias4[nt] = cl.bias;
    }
  }
  const float act_slope = P.ep.act == SRK_ACT_NONE ? 1.f
                          : P.ep.act == SRK_ACT_RELU ? 0.f
                          : P.ep.act == SRK_ACT_PRELU ? P.ep.prelu_w[0] : P.ep.slope;
  f32x4 acc[NTW][MTW], pend[NTW][MTW];
  float amax = 0.f;
  // Deferred stores are UNCONDITIONAL buffer stores (pixels outside the tile / the image, and the first stage with nothing
  // parked, carry an out-of-range offset that the buffer unit drops): a store under a branch is one the compiler cannot
  // count, and the staging loads' s_waitcnt then degrades to vmcnt(0) -- every stage waited for the write
  // acknowledgements of the previous tile (measured: 300 -> ... us on the c2 first layer).
  constexpr unsigned kDrop = 0x80000000u;  // (host: the output tensor is smaller than 2 GiB)
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(P.out, 0, B.out_bytes, 0x00020000);
  unsigned pend_voff[MTW];  // byte offset of the lane's pixel mt (channel group 0 of the slice), or kDrop
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) pend_voff[mt] = kDrop;
  constexpr int NST = NTW * MTW;
  // the parked tile leaves in the FIRST K steps of the next one: loads and stores share vmcnt and complete out of order
  // with each other, so the staging commit at the end of the stage waits for every store issued before it -- by then
  // they are two K steps old
  constexpr int SPREAD = QT > 2 ? QT - 2 : 1;
  constexpr int PER_STEP = (NST + SPREAD - 1) / SPREAD;
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  auto store_slot = [&](auto qc) {
    constexpr int q = decltype(qc)::value;
    constexpr int mt = q / NTW, nt = q - mt * NTW;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, pend[nt][mt]), orsrc, (int)(pend_voff[mt] + 4u * (unsigned)coff[nt]), 0, 0);
  };
  if (S > 0) {
    issue();
    commit(hal0);
    if (S > 1) issue();
  }
  // everything loaded before the loop must have landed before it: a first use inside the loop puts an
  // s_waitcnt vmcnt(0) into every iteration (it would wait for the previous tile's store acknowledgements)
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) asm volatile("" ::"v"(bias4[nt]));
  asm volatile("" ::"v"(act_slope));
  __syncthreads();  // filter and tile 0 visible
  // WREG: the lane's filter fragments of every K step stay in registers for the whole kernel (QT * 2 * NTW x 16 bytes);
  // with them in LDS each wave re-reads the whole filter per 32 pixels and the LDS pipe is as busy as the MFMA pipe
  uint4 wreg[WREG ? QT : 1][2][NTW];
  if constexpr (WREG) {
#pragma unroll
    for (int q = 0; q < QT; ++q)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        wreg[q][0][nt] = wl[q * wslot + wrow[nt]];
        wreg[q][1][nt] = wl[q * wslot + 4 * NB + wrow[nt]];
      }
  }
  for (int s = 0; s < S; ++s) {
    const int n = wc.n, r0 = wc.y * P.TH, c0 = wc.x * P.TW;
    wc.advance(st_n, st_y, st_x, P.tiles_x, P.tiles_y);
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {  // (waves past the end of a ragged tile run on pixel 0 and drop their stores: no branch around the stores)
      const uint4* hal = hal0 + (size_t)(s & 1) * B.NPIXp;
      uint4 fa[WREG ? 1 : 2][2][NTW], fb[2][2][MTW];  // [buffer][plane][tile]
      int toff = 0, ks = 0, qq = 0;
      auto load_frags = [&](uint4 (&a)[2][NTW], uint4 (&b)[2][MTW]) {
        const uint4* hb = hal + toff;
        const uint4* wt = wl + qq * wslot;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
          const uint4 p0 = hb[hp[mt]], p1 = hb[hp[mt] + 1];
          b[0][mt] = make_uint4(p0.x, p0.y, p1.x, p1.y);
          b[1][mt] = make_uint4(p0.z, p0.w, p1.z, p1.w);
        }
        if constexpr (!WREG) {
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) {
            a[0][nt] = wt[wrow[nt]];
            a[1][nt] = wt[4 * NB + wrow[nt]];
          }
        }
        ++qq;
        toff += 8;
        if (++ks == B.KS) {
          ks = 0;
          toff += P.HW - 8 * B.KS;
        }
      };
      auto mfmas = [&](const uint4 (&a)[2][NTW], const uint4 (&b)[2][MTW]) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
          for (int mt = 0; mt < MTW; ++mt) acc[nt][mt] = mfma16x<F16>(a[0][nt], b[1][mt], acc[nt][mt]);  // w_h * x_l
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
          for (int mt = 0; mt < MTW; ++mt) acc[nt][mt] = mfma16x<F16>(a[1][nt], b[0][mt], acc[nt][mt]);  // w_l * x_h
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
          for (int mt = 0; mt < MTW; ++mt) acc[nt][mt] = mfma16x<F16>(a[0][nt], b[0][mt], acc[nt][mt]);  // w_h * x_h
      };
      if (!(SRK_KDBG(B.dbg) & 4)) load_frags(fa[0], fb[0]);
      if (!(SRK_KDBG(B.dbg) & 4)) rw_static_for<0, QT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        if (t + 1 < QT) load_frags(fa[WREG ? 0 : ((t + 1) & 1)], fb[(t + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (WREG) mfmas(wreg[t], fb[t & 1]); else mfmas(fa[t & 1], fb[t & 1]);
        __builtin_amdgcn_sched_barrier(0);
        rw_static_for<(t * PER_STEP < NST ? t * PER_STEP : NST), ((t + 1) * PER_STEP < NST ? (t + 1) * PER_STEP : NST)>(
            [&](auto qc) { store_slot(qc); });
      });
      if (s + 1 < S) commit(hal0 + (size_t)((s + 1) & 1) * B.NPIXp);  // (that buffer was last read in stage s - 1)
      if (s + 2 < S) issue();
      // tile finished: park it (C/D col = lane & 15 = pixel, rows kq*4 + reg = 4 consecutive channels per M tile)
      const unsigned tile_off = 4u * (unsigned)epi_tile_setup(P, n, r0, c0).off0;
      int pend_mask = 0;
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) {
        const int r = pix_ok[mt] >> 16, c = pix_ok[mt] & 0xffff;
        const bool ok = pix_ok[mt] >= 0 && r0 + r < P.PH && c0 + c < P.PW && !(SRK_KDBG(B.dbg) & 2);
        if (ok) pend_mask |= 1 << mt;
        pend_voff[mt] = ok ? tile_off + 4u * (unsigned)poff[mt] : kDrop;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          f32x4 v = acc[nt][mt];
          if constexpr (F16) v *= dsc;
          v += bias4[nt];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : act_slope * v[e];
          pend[nt][mt] = v;
          if (P.ep.y_amax && ((pend_mask >> mt) & 1)) amax = abs_max4(amax, v);
        }
      }
    }
    __syncthreads();
  }
  rw_static_for<0, NST>([&](auto qc) { store_slot(qc); });  // the last tile (S = 0: nothing parked, dropped)
  if (P.ep.y_amax) amax_commit(P.ep.y_amax, amax, blockIdx.x + wave, amax_peek(P.ep.y_amax, blockIdx.x + wave));
}


// ---------------------------------------------------------------------------------------------------------------------
// k_conv_rowsr: the same layer with ROW-REUSED pixel fragments and the filter in registers (round 4).
//
// k_conv_rowsw on the c2 first layer (3 -> 64, 5x5, 64 x 256x256; builds with the ablation word as a compile-time
// constant, -DSRK_KDBG_CONST, 310 us): skeleton (staging, parking, barriers) alone 66 us, + the MFMA loop 219 us, + the
// stores 310 us, where the layer's 7.9 M MFMAs are 102 us of matrix issue.  Two things: (1) per 32 pixels a wave read the
// whole filter (40 x ds_read_b128) and 20 pixel fragments for 120 MFMAs -- 480 reads = 1920 LDS cycles per 256-pixel
// stage and CU beside 3840 clocks of matrix issue per SIMD, every first read of a K step in front of its MFMAs; (2) all
// eight waves of the one block per CU ran their matrix phase and their VALU phases (staging conversion, parking
// arithmetic) in lockstep behind one barrier per stage, so the matrix pipe idled through every VALU phase.
// Here:
//   * a wave owns 4 consecutive tile rows x 16 columns x 32 output channels.  The K slots of a step are 8 taps of ONE
//     kernel row, so the fragment of halo row R serves (tile row mt, kernel row ky) for every mt + ky = R: the 4 rows x KH
//     kernel rows need KH + 3 row fragments (2 x ds_read_b128 each) instead of 4 KH -- 16 reads per 64 pixels for the 5x5
//     layer instead of 40 -- and each is consumed as soon as it lands (walk over R; per accumulator the kernel rows still
//     arrive in ascending order with the passes w_h x_l, w_l x_h, w_h x_h: the summation order of k_conv_bf3_rows);
//   * the wave's filter fragments (KH x 2 planes x 2 channel tiles = 80 registers for 5x5) are loaded once from global
//     memory and stay in registers: no filter in LDS at all, LDS reads per 256 pixels 480 -> 64 + 64;
//   * blocks are 4 waves (2 row groups x 2 channel halves: an 8 x 16 tile x 64 channels), TWO persistent blocks per CU, one
//     wave of each per SIMD: the two blocks of a CU drift apart and one's matrix phase covers the other's VALU phases and
//     barrier.  Register budget per wave is the same 256 as with one 8-wave block.
// Applies when the kernel row fits one K step (KW <= 8) and the 8 x 16 tile's halo fits one pixel per thread; anything
// else stays with k_conv_rowsw.
//
// BAND (round 6, the "sliding window" of DESIGN 10.9 item 2): a block walks whole BANDS -- the tiles_x tiles of one tile
// row, left to right -- and stages per stage ONE aligned chunk of 16 columns x (7 + KH) halo rows (chunk i = the columns of
// tile i: one 64-byte line per row and plane) into a ring of four chunk slots; tile i reads chunks i - 1, i, i + 1 (a
// lane's two pixels: slot and column fixed per lane, the slot index rotates with the stage).  The chunks left of a band's
// first tile and right of its last are all padding: a fifth, zero slot.  Read requests per tile: 3 (7 + KH) = 36 lines
// against ~60 - 70 for the 20-column halo segments that straddle three lines; the products and their order are unchanged.
template <int KH, bool F16, bool RELU, int ICN, bool BAND = false>
__global__ __launch_bounds__(256, 2) void k_conv_rowsr(RowswParams B) {
  constexpr int NTW = 2, MTW = 4, NTHR = 256, NR = KH + MTW - 1, TH = 8, TW = 16;
  constexpr int BHH = 7 + KH, BSLOT = BHH * 16;  // BAND: halo rows, uint4 per chunk slot
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
  const MfmaConvParams& P = B.P;
  uint4* hal0 = smem4;  // 2 buffers x NPIXp pixels x {h: 4 x 16 bit, l: 4 x 16 bit}
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kq = lane >> 4;
  const int chh = wave & 1, rg = wave >> 1;
  const int npix = P.HH * P.HW;
  float sx = 1.f, dsc = 1.f;
  if constexpr (F16) {
    const int kx = amax_scale_exp(amax_read(P.ep.x_amax));
    sx = exp2i(kx);
    dsc = exp2i(-kx) * B.w_descale[0];
  }
  const int nsl = B.nsl;
  const int xcd = blockIdx.x & 7;
  const int sl = (blockIdx.x >> 3) % nsl, bi = (blockIdx.x >> 3) / nsl;
  // filter fragments of this wave: [kernel row][plane][channel tile], lane (j, kq) = channel j of the tile, K slots 8 kq ..
  uint4 wreg[KH][2][NTW];
#pragma unroll
  for (int q = 0; q < KH; ++q)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int oc = sl * 64 + chh * 32 + nt * 16 + j, ocb = oc / B.NBfull;
      const uint4* w = B.wq + ((size_t)q * B.OCb + ocb) * (size_t)(8 * B.NBfull) + (oc - ocb * B.NBfull);
      wreg[q][0][nt] = w[kq * B.NBfull];
      wreg[q][1][nt] = w[(4 + kq) * B.NBfull];
    }
  // pixels past the halo (the padded tap slots of the last row read them; they meet zero filter taps and must be finite)
  if constexpr (BAND) {
    if (tid < BSLOT) hal0[4 * BSLOT + tid] = make_uint4(0, 0, 0, 0);  // the zero slot
  } else {
    for (int e = tid; e < 2 * (B.NPIXp - npix); e += NTHR) {
      const int b = e / (B.NPIXp - npix), i = e - b * (B.NPIXp - npix);
      hal0[(size_t)b * B.NPIXp + npix + i] = make_uint4(0, 0, 0, 0);
    }
  }
  // tiles of this block: XCD-aware contiguous ranges (see k_conv_bfw)
  const int nblk = gridDim.x;
  int first, count;
  const int nb_x = ((nblk + 7 - xcd) >> 3) / nsl;
  {
    const int per_x = B.ntiles >> 3, rem_x = B.ntiles & 7;
    const int tiles_x = per_x + (xcd < rem_x ? 1 : 0);
    const int start_x = xcd * per_x + (xcd < rem_x ? xcd : rem_x);
    first = start_x + bi;
    count = bi < tiles_x ? (tiles_x - bi + nb_x - 1) / nb_x : 0;
  }
  // BAND: B.ntiles counts bands (image, tile row); the walk's x is the tile inside the band and steps by one
  const int S = BAND ? count * P.tiles_x : count;
  const int img_tiles = BAND ? P.tiles_y : P.tiles_x * P.tiles_y;
  const int st_n = nb_x / img_tiles;
  const int st_y = BAND ? nb_x - st_n * img_tiles : (nb_x - st_n * img_tiles) / P.tiles_x;
  const int st_x = BAND ? 0 : nb_x - st_n * img_tiles - st_y * P.tiles_x;
  RowswWalk wi, wc;
  if constexpr (BAND) {
    wi.n = __builtin_amdgcn_readfirstlane(first / img_tiles);
    wi.y = __builtin_amdgcn_readfirstlane(first - wi.n * img_tiles);
    wi.x = 0;
  } else {
    wi.init(first, P.tiles_x, img_tiles);
  }
  wc = wi;
  auto walk = [&](RowswWalk& w) {
    if constexpr (BAND) {
      if (++w.x == P.tiles_x) {
        w.x = 0;
        w.y += st_y;
        w.n += st_n;
        if (w.y >= P.tiles_y) {
          w.y -= P.tiles_y;
          ++w.n;
        }
      }
    } else {
      w.advance(st_n, st_y, st_x, P.tiles_x, P.tiles_y);
    }
  };

  // ---- staging: thread tid owns halo pixel tid (npix <= 256).  Loads are unconditional and branch-free as in
  // k_conv_rowsw, and run TWO stages ahead of the LDS commit in two register sets (tile t in set t & 1): vmcnt retires in
  // order, so the wait for a tile's pixels is also a wait for every store issued before those loads -- with the loads one
  // stage ahead that was the previous stage's eight stores, one stage old, and the write acknowledgements of a chip that
  // writes 3.5 TB/s take longer than that (constant-ablation builds, tools/build_variant.sh -DSRK_KDBG_CONST: the layer in
  // 199 us without the loads, 176 us without the stores, 302 us with both).  Now the wait covers stores two stages old.
  // Everything is issued unconditionally -- tiles past the end of the block's range are dummies (clamped loads, a commit
  // nobody reads, stores dropped) and the stage loop runs in pairs -- so that the compiler can count: a path on which an
  // operation was not issued makes its s_waitcnt pass assume the smaller distance everywhere.
  float pv[2][4];
  bool vok[2] = {false, false};
  const int hy0 = BAND ? tid >> 4 : tid / P.HW, hx0 = BAND ? tid & 15 : tid - hy0 * P.HW;
  const int nstage = BAND ? BSLOT : npix;  // threads that stage a pixel
  const size_t plane = (size_t)P.IH * P.IW;
  const size_t est = P.in_nchw ? plane : 1;
  const size_t e1 = P.IC > 1 ? est : 0, e2 = P.IC > 2 ? 2 * est : 0, e3 = P.IC > 3 ? 3 * est : 0;
  int issued = 0;  // tiles issued so far (wave-uniform)
  auto issue = [&](auto setc) {
    constexpr int st = decltype(setc)::value;
    const bool live = issued < S;
    ++issued;
    const int n = live ? wi.n : 0, ty = live ? wi.y : 0, tx = live ? wi.x : 0;
    const int iy = ty * TH + P.iy0 + hy0, ix = tx * TW + (BAND ? 0 : P.ix0) + hx0;
    walk(wi);
    if (SRK_KDBG(B.dbg) & 1) return;
    vok[st] = (BAND ? live : true) && tid < nstage && (unsigned)iy < (unsigned)P.IH && (unsigned)ix < (unsigned)P.IW;
    const int cy = min(max(iy, 0), P.IH - 1), cx = min(max(ix, 0), P.IW - 1);
    const size_t pix = (size_t)cy * P.IW + cx;
    // (one load instruction per channel the input HAS: every VMEM instruction in the consumers' store stream costs issue
    //  time of the stores -- the layer ran in 199 us without the loads, 252 us with all of them hitting L1, 300 us as it is)
    const float* src = P.in + (size_t)n * P.IC * plane + (P.in_nchw ? pix : pix * P.IC);
    pv[st][0] = src[0];
    pv[st][1] = ICN > 1 ? src[e1] : 0.f;
    pv[st][2] = ICN > 2 ? src[e2] : 0.f;
    pv[st][3] = ICN > 3 ? src[e3] : 0.f;
  };
  auto commit = [&](uint4* hal, auto setc) {
    constexpr int st = decltype(setc)::value;
    if (SRK_KDBG(B.dbg) & 16) return;
    uint2 hu, lu;
    if constexpr (F16) {
      typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
      f16x4 h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xs = (vok[st] && e < P.IC) ? pv[st][e] * sx : 0.f;
        const _Float16 hh = (_Float16)xs;
        h[e] = hh;
        l[e] = (_Float16)(xs - (float)hh);
      }
      hu = __builtin_bit_cast(uint2, h);
      lu = __builtin_bit_cast(uint2, l);
    } else {
      typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
      bf16x4 h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xv = (vok[st] && e < P.IC) ? pv[st][e] : 0.f;
        const __bf16 hh = (__bf16)xv;
        h[e] = hh;
        l[e] = (__bf16)(xv - (float)hh);
      }
      hu = __builtin_bit_cast(uint2, h);
      lu = __builtin_bit_cast(uint2, l);
    }
    if (tid < nstage) hal[tid] = make_uint4(hu.x, hu.y, lu.x, lu.y);  // (the wait for pv sits in front of the conversions)
  };

  // ---- MFMA side: halo slot of (row rg * 4 + R, column j, taps 2 kq, 2 kq + 1) = hp0 + R * HW
  const int hp0 = (rg * MTW) * P.HW + j + 2 * kq;
  // BAND: the lane's two pixels as (chunk relative to the tile's own: -1 / 0 / +1, column inside the chunk)
  const int bo0 = j + 2 * kq + P.ix0, bo1 = bo0 + 1;
  const int bd0 = bo0 >> 4, bd1 = bo1 >> 4;
  const int bc0 = (rg * MTW) * 16 + (bo0 & 15), bc1 = (rg * MTW) * 16 + (bo1 & 15);
  f32x4 bias4[NTW];
  int coff[NTW];
  const EpiTile e0 = epi_tile_setup(P, 0, 0, 0);
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const EpiCol cl = epi_col_setup(P.ep, P.OW, P.OC, sl * 64 + chh * 32 + nt * 16 + kq * 4);
    coff[nt] = (int)cl.off_oc;
    bias4[nt] = cl.bias;
  }
  const float act_slope = P.ep.act == SRK_ACT_NONE ? 1.f
                          : P.ep.act == SRK_ACT_RELU ? 0.f
                          : P.ep.act == SRK_ACT_PRELU ? P.ep.prelu_w[0] : P.ep.slope;
  f32x4 acc[NTW][MTW], pend[NTW][MTW];
  float amax = 0.f;
  constexpr unsigned kDrop = 0x80000000u;
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(P.out, 0, B.out_bytes, 0x00020000);
  unsigned pend_voff[MTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) pend_voff[mt] = kDrop;
  constexpr int NST = NTW * MTW;
  constexpr int SPREAD = NR - 2;
  constexpr int PER_STEP = (NST + SPREAD - 1) / SPREAD;
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  auto store_slot = [&](auto qc) {
    constexpr int q = decltype(qc)::value;
    constexpr int mt = q / NTW, nt = q - mt * NTW;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, pend[nt][mt]), orsrc, (int)(pend_voff[mt] + 4u * (unsigned)coff[nt]), 0, 0);
  };
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;
  if constexpr (BAND) {
    if (S > 0) {  // chunks 0 and 1 committed, 2 and 3 in flight: stage s commits chunk s + 2 and issues chunk s + 4
      issue(Set0{});
      issue(Set1{});
      commit(hal0, Set0{});
      issue(Set0{});
      commit(hal0 + BSLOT, Set1{});
      rw_static_for<0, NST>([&](auto qc) { store_slot(qc); });  // (dropped: see below)
      issue(Set1{});
    }
  } else if (S > 0) {
    issue(Set0{});                  // tile 0
    commit(hal0, Set0{});
    issue(Set1{});                  // tile 1
    // eight dropped stores (nothing is parked yet): the steady state has a stage's eight stores between the two sets'
    // loads, and the first wait inside the loop is counted for the smaller of the two distances
    rw_static_for<0, NST>([&](auto qc) { store_slot(qc); });
    issue(Set0{});                  // tile 2
  }
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) asm volatile("" ::"v"(bias4[nt]));
#pragma unroll
  for (int q = 0; q < KH; ++q)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      typedef unsigned u4v __attribute__((ext_vector_type(4)));
      asm volatile("" ::"v"(__builtin_bit_cast(u4v, wreg[q][0][nt])), "v"(__builtin_bit_cast(u4v, wreg[q][1][nt])));
    }
  asm volatile("" ::"v"(act_slope));
  __syncthreads();  // tile 0 visible
#ifdef SRK_ROWSR_PROF
  unsigned pacc[5] = {0, 0, 0, 0, 0};
  unsigned pgrp[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // clocks per row group R of the matrix loop (SRK_ROWSR_PROF=2)
  unsigned pprev = 0;
#endif
  auto stage = [&](const int s, auto nset) {  // nset: register set of tile s + 1 (= of tile s + 3)
    const unsigned pt0 = RS_T();
    unsigned pt1 = 0, pt2 = 0, pt3 = 0;
    const int n = wc.n, r0 = wc.y * TH, c0 = wc.x * TW;
    const int bt = wc.x;
    walk(wc);
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
      const uint4* hal = hal0 + (size_t)(s & 1) * B.NPIXp + hp0;
      const uint4 *ha = hal, *hb = hal + 1;
      int rstep = P.HW;
      if constexpr (BAND) {
        const bool z0 = (bd0 < 0 && bt == 0) || (bd0 > 0 && bt == P.tiles_x - 1);
        const bool z1 = (bd1 < 0 && bt == 0) || (bd1 > 0 && bt == P.tiles_x - 1);
        ha = hal0 + (z0 ? 4 : (s + bd0) & 3) * BSLOT + bc0;
        hb = hal0 + (z1 ? 4 : (s + bd1) & 3) * BSLOT + bc1;
        rstep = 16;
      }
      uint4 fb[2][2];  // [buffer][plane]
      int roff = 0;
      auto load_row = [&](uint4 (&b)[2]) {
        const uint4 p0 = ha[roff], p1 = hb[roff];
        b[0] = make_uint4(p0.x, p0.y, p1.x, p1.y);
        b[1] = make_uint4(p0.z, p0.w, p1.z, p1.w);
        roff += rstep;
      };
      if (!(SRK_KDBG(B.dbg) & 4)) load_row(fb[0]);
      if (!(SRK_KDBG(B.dbg) & 4)) rw_static_for<0, NR>([&](auto rc) {
        constexpr int R = decltype(rc)::value;
        if (R + 1 < NR) load_row(fb[(R + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const uint4(&b)[2] = fb[R & 1];
        // (tile row mt, kernel row ky = R - mt) for every valid pair; pass-major like k_conv_rowsw's K step
        rw_static_for<0, MTW>([&](auto mc) {
          constexpr int mt = decltype(mc)::value, ky = R - mt;
          if constexpr (ky >= 0 && ky < KH) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[nt][mt] = mfma16x<F16>(wreg[ky][0][nt], b[1], acc[nt][mt]);  // w_h * x_l
          }
        });
        rw_static_for<0, MTW>([&](auto mc) {
          constexpr int mt = decltype(mc)::value, ky = R - mt;
          if constexpr (ky >= 0 && ky < KH) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[nt][mt] = mfma16x<F16>(wreg[ky][1][nt], b[0], acc[nt][mt]);  // w_l * x_h
          }
        });
        rw_static_for<0, MTW>([&](auto mc) {
          constexpr int mt = decltype(mc)::value, ky = R - mt;
          if constexpr (ky >= 0 && ky < KH) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[nt][mt] = mfma16x<F16>(wreg[ky][0][nt], b[0], acc[nt][mt]);  // w_h * x_h
          }
        });
        __builtin_amdgcn_sched_barrier(0);
        rw_static_for<(R * PER_STEP < NST ? R * PER_STEP : NST), ((R + 1) * PER_STEP < NST ? (R + 1) * PER_STEP : NST)>(
            [&](auto qc) { store_slot(qc); });
#if defined(SRK_ROWSR_PROF) && SRK_ROWSR_PROF == 2
        {
          const unsigned tn = RS_T();
          pgrp[R & 7] += tn - (R == 0 ? pt0 : pprev);
          pprev = tn;
        }
#endif
      });
      pt1 = RS_T();
      if constexpr (BAND) commit(hal0 + (size_t)((s + 2) & 3) * BSLOT, nset);  // chunk s + 2 (slot last read in stage s - 1)
      else commit(hal0 + (size_t)((s + 1) & 1) * B.NPIXp, nset);  // tile s + 1 (that buffer was last read in stage s - 1)
      pt2 = RS_T();
      issue(nset);                                            // tile s + 3 (BAND: chunk s + 4)
      pt3 = RS_T();
      // tile finished: park it (C/D col = lane & 15 = pixel column, rows kq*4 + reg = 4 consecutive channels)
      const unsigned tile_off = 4u * (unsigned)epi_tile_setup(P, n, r0, c0).off0;
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) {
        const int r = rg * MTW + mt;
        const bool ok = s < S && r0 + r < P.PH && c0 + j < P.PW && !(SRK_KDBG(B.dbg) & 2);
        pend_voff[mt] = ok ? tile_off + 4u * (unsigned)(r * e0.RS + j * e0.CS) : kDrop;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          f32x4 v = acc[nt][mt];
          if constexpr (F16) v *= dsc;
          v += bias4[nt];
          if constexpr (RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : act_slope * v[e];
          }
          pend[nt][mt] = v;
          if (P.ep.y_amax && ok) amax = abs_max4(amax, v);
        }
      }
    }
    const unsigned pt4 = RS_T();
    __syncthreads();
    const unsigned pt5 = RS_T();
    (void)pt0; (void)pt1; (void)pt2; (void)pt3; (void)pt4; (void)pt5;
    RS_ACC(0, pt0, pt1); RS_ACC(1, pt1, pt2); RS_ACC(2, pt2, pt3); RS_ACC(3, pt3, pt4); RS_ACC(4, pt4, pt5);
  };
  for (int s0 = 0; s0 < S; s0 += 2) {  // (an odd S ends with one dummy stage)
    if constexpr (BAND) {
      stage(s0, Set0{});
      stage(s0 + 1, Set1{});
    } else {
      stage(s0, Set1{});
      stage(s0 + 1, Set0{});
    }
  }
  rw_static_for<0, NST>([&](auto qc) { store_slot(qc); });  // the last tile (S = 0: nothing parked, dropped)
  if (P.ep.y_amax) amax_commit(P.ep.y_amax, amax, blockIdx.x + wave, amax_peek(P.ep.y_amax, blockIdx.x + wave));
#ifdef SRK_ROWSR_PROF
  if (B.prof && lane == 0) {
#pragma unroll
    for (int i = 0; i < 5; ++i) B.prof[((size_t)blockIdx.x * 4 + wave) * 8 + i] = pacc[i];
    B.prof[((size_t)blockIdx.x * 4 + wave) * 8 + 5] = (unsigned)S;
#if SRK_ROWSR_PROF == 2
#pragma unroll
    for (int i = 0; i < 8; ++i) B.prof[(size_t)gridDim.x * 32 + ((size_t)blockIdx.x * 4 + wave) * 8 + i] = pgrp[i];
#endif
  }
#endif
}

// the tile (<= 256 pixels) that covers PH x PW with the fewest tiles and a halo of at most cap_px pixels
static bool rowsw_pick_tile(int PH, int PW, int KHv, int KWv, long cap_px, TilePick& best) {
  bool found = false;
  long best_tiles = 0, best_halo = 0;
  const int maxTW = PW < 256 ? PW : 256;
  for (int TW = 1; TW <= maxTW; ++TW) {
    int TH = 256 / TW;
    if (TH > PH) TH = PH;
    for (; TH >= 1; --TH) {
      const int HH = TH - 1 + KHv, HWd = TW - 1 + KWv;
      if ((long)HH * HWd > cap_px) continue;
      const long tiles = (long)cdiv(PH, TH) * cdiv(PW, TW);
      const long halo = (long)HH * HWd * tiles;
      if (!found || tiles < best_tiles || (tiles == best_tiles && halo < best_halo)) {
        found = true;
        best_tiles = tiles;
        best_halo = halo;
        best = TilePick{TH, TW, (int)cdiv(PH, TH), (int)cdiv(PW, TW), HH, HWd, (double)PH * PW / ((double)tiles * 256.0)};
      }
      break;
    }
  }
  return found;
}

static inline int rowsw_steps(const GatherConv& g) { return g.KH * ((g.KW + 7) / 8); }

// Applicability: stride-1 CONV gathers with Cin <= 4 (NHWC or NCHW in place), OC a multiple of 64, 3 or 5 K steps
// (3x3 and 5x5 filters), every output group on the 16-byte store path, the branch-free activations, an output below
// 2 GiB, and at least two 256-pixel tiles per CU.  SRK_ROWSW: 0 never, 1 whenever applicable, unset = automatic.
bool conv_rowsw_applicable(const GatherConv& g, const Epi& ep, const float* in, const float* out, const float* mask_y) {
  const char* e = env_str("SRK_ROWSW");
  const int mode = e ? atoi(e) : 2;
  if (mode == 0 || mask_y || g.trans || g.stride != 1 || g.in_ps_r > 1 || g.IC > 4 || g.IC < 1) return false;
  if (g.OC % 64 != 0 || g.OC < 64) return false;
  const int Q = rowsw_steps(g);
  if (Q != 3 && Q != 5) return false;
  if (!conv_epi_all_vector(g.OC, ep, out)) return false;
  if (ep.residual || (ep.act == SRK_ACT_PRELU && ep.prelu_n > 1)) return false;
  if (ep.act != SRK_ACT_NONE && ep.act != SRK_ACT_RELU && ep.act != SRK_ACT_LRELU && ep.act != SRK_ACT_PRELU) return false;
  if ((long)g.N * g.OH * g.OW * g.OC >= (1L << 29)) return false;  // 32-bit byte offsets into an output below 2 GiB (kDrop)
  if ((long)g.N * g.OH * g.OW >= (1L << 30) || (long)g.IH * g.IW * g.IC >= (1L << 30)) return false;
  (void)in;
  if (mode == 1) return true;
  return (long)g.N * g.OH * g.OW >= 256L * 2 * kNumCU;
}

template <int NTW, int QT>
static int rowsw_launch(const RowswParams& B, size_t lds, int grid, hipStream_t s) {
  note_amax_written(B.P.ep.y_amax != nullptr);
  if (B.w_descale) {
    static LdsLimit limh;
    limh.ensure(reinterpret_cast<const void*>(&k_conv_rowsw<NTW, QT, true>), lds);
    note_kernel("k_conv_rowsw<%d,%d,f16>", NTW, QT);
    hipLaunchKernelGGL((k_conv_rowsw<NTW, QT, true>), dim3(grid), dim3(512), lds, s, B);
    return check_launch("conv_rowsw");
  }
  static LdsLimit lim;
  lim.ensure(reinterpret_cast<const void*>(&k_conv_rowsw<NTW, QT, false>), lds);
  note_kernel("k_conv_rowsw<%d,%d>", NTW, QT);
  hipLaunchKernelGGL((k_conv_rowsw<NTW, QT, false>), dim3(grid), dim3(512), lds, s, B);
  return check_launch("conv_rowsw");
}

template <int KH, bool BAND = false>
static int rowsr_launch(const RowswParams& B, size_t lds, int grid, hipStream_t s) {
  note_amax_written(B.P.ep.y_amax != nullptr);
  const bool relu = B.P.ep.act == SRK_ACT_RELU;
  const bool f16 = B.w_descale != nullptr;
  note_kernel("k_conv_rowsr<%d%s%s%s>", KH, f16 ? ",f16" : "", relu ? ",relu" : "", BAND ? ",band" : "");
  auto go = [&](auto f16c, auto reluc) {
    constexpr bool F = decltype(f16c)::value, R = decltype(reluc)::value;
    switch (B.P.IC) {
      case 1: hipLaunchKernelGGL((k_conv_rowsr<KH, F, R, 1, BAND>), dim3(grid), dim3(256), lds, s, B); break;
      case 2: hipLaunchKernelGGL((k_conv_rowsr<KH, F, R, 2, BAND>), dim3(grid), dim3(256), lds, s, B); break;
      case 3: hipLaunchKernelGGL((k_conv_rowsr<KH, F, R, 3, BAND>), dim3(grid), dim3(256), lds, s, B); break;
      default: hipLaunchKernelGGL((k_conv_rowsr<KH, F, R, 4, BAND>), dim3(grid), dim3(256), lds, s, B); break;
    }
  };
  if (f16 && relu) go(std::true_type{}, std::true_type{});
  else if (f16) go(std::true_type{}, std::false_type{});
  else if (relu) go(std::false_type{}, std::true_type{});
  else go(std::false_type{}, std::false_type{});
  return check_launch("conv_rowsr");
}

// returns -1 when no tile fits (the caller falls back to k_conv_bf3_rows).  f16: the f16x3 arithmetic (ep.x_amax set)
int conv_rowsw_gather(const GatherConv& g, const float* in, const float* wp, float* out, const Epi& ep, hipStream_t s,
                      bool f16) {
  const size_t elems = (size_t)g.KH * g.KW * g.IC * g.OC;
  const char* prepared = reinterpret_cast<const char*>(wp) + bf3_prepared_offset(elems);
  const char* fsec = prepared + f16_section_offset(g.IC, g.OC, g.KH * g.KW);
  const uint4* wq = reinterpret_cast<const uint4*>(f16 ? fsec : prepared);
  const float* w_descale = f16 ? reinterpret_cast<const float*>(fsec + bf3_main_bytes(g.IC, g.OC, g.KH * g.KW)) : nullptr;
  return for_each_phase(g, in, wp, out, ep, nullptr, 0.f, [&](const MfmaConvParams& P0) {
    RowswParams B{};
    B.P = P0;
    MfmaConvParams& P = B.P;
    B.wq = wq;
    B.w_descale = w_descale;
    B.KS = (P.KWv + 7) / 8;
    B.NBfull = P.OC >= 64 ? 64 : P.OC;
    B.OCb = (P.OC + 63) / 64;
    const int Q = P.KHv * B.KS;
    // 64-channel slices.  (32-channel slices double the number of stages, and with them the per-stage park / staging /
    // barrier work: measured 366 us against 289 us on the c2 first layer, even with the filter in registers.)
    constexpr int ntw = 4;
    B.nsl = P.OC / (16 * ntw);
    const int dbg = SRK_EXP_INT("SRK_ROWSW_DBG", 0);
    B.dbg = dbg;
    // row-reused fragments, filter in registers, two 4-wave blocks per CU (k_conv_rowsr): one K step per kernel row and the
    // fixed 8 x 16 tile's halo on one pixel per thread.  SRK_ROWSR=0: the 8-wave kernel below.
    if (P.is == 1 && B.KS == 1 && (Q == 3 || Q == 5) && (7 + P.KHv) * (15 + P.KWv) <= 256 && env_int("SRK_ROWSR", 1) != 0) {
      P.TH = 8; P.TW = 16;
      P.tiles_y = (P.PH + 7) / 8; P.tiles_x = (P.PW + 15) / 16;
      P.HH = 7 + P.KHv; P.HW = 15 + P.KWv;
      B.NPIXp = P.HH * P.HW + 16;
      B.nsl = P.OC / 64;
      const long ntiles = (long)P.tiles_x * P.tiles_y * P.N;
      // Whole bands per block with one aligned 16-column chunk staged per tile (k_conv_rowsr<.., band>): where the input is no
      // wider than the tiles' columns (the chunks beside a band are all padding), the left overhang fits one chunk, and the
      // bands divide evenly enough among the blocks.  SRK_ROWSB: 0 never, 2 whenever applicable, unset = that rule.
      const int band_mode = env_int("SRK_ROWSB", 1);
      const long nbands = (long)P.tiles_y * P.N;
      if (band_mode != 0 && ntiles < (1L << 30) && P.IW <= 16 * P.tiles_x && P.ix0 <= 0 && P.ix0 >= -16 &&
          P.ix0 + 15 + 7 < 32) {
        int grid = 2 * kNumCU - (2 * kNumCU) % (8 * B.nsl);
        const long want = ((nbands + 7) / 8) * 8 * B.nsl;
        if (want < grid) grid = (int)want;
        const long nblk = grid / B.nsl;
        const long rounds = nblk > 0 ? (nbands + nblk - 1) / nblk : 0;
        const bool even = nblk > 0 && rounds * nblk * 100 <= nbands * 107 && nbands >= 2 * nblk;
        if (grid > 0 && (band_mode == 2 || even)) {
          B.ntiles = (int)nbands;
          B.out_bytes = (unsigned)((size_t)P.N * P.OH * P.OW * P.OC * sizeof(float));
          const size_t lds = (size_t)5 * (7 + P.KHv) * 16 * 16;
#ifdef SRK_ROWSR_PROF
          B.prof = g_rowsr_prof;
#endif
          return Q == 3 ? rowsr_launch<3, true>(B, lds, grid, s) : rowsr_launch<5, true>(B, lds, grid, s);
        }
      }
      if (ntiles < (1L << 30)) {
        B.ntiles = (int)ntiles;
        B.out_bytes = (unsigned)((size_t)P.N * P.OH * P.OW * P.OC * sizeof(float));
        const size_t lds = (size_t)2 * B.NPIXp * 16;
#ifdef SRK_ROWSR_PROF
        B.prof = g_rowsr_prof;
#endif
        int grid = 2 * kNumCU - (2 * kNumCU) % (8 * B.nsl);
        const long want = ((ntiles + 7) / 8) * 8 * B.nsl;
        if (want < grid) grid = (int)want;
        if (grid > 0) return Q == 3 ? rowsr_launch<3>(B, lds, grid, s) : rowsr_launch<5>(B, lds, grid, s);
      }
    }
    TilePick best{};
    if (P.is != 1 || !rowsw_pick_tile(P.PH, P.PW, P.KHv, P.KWv, 768, best)) return -1;
    P.TH = best.TH; P.TW = best.TW; P.tiles_y = best.tiles_y; P.tiles_x = best.tiles_x; P.HH = best.HH; P.HW = best.HW;
    B.NPIXp = best.HH * best.HW + 16;  // +16: the last row's padded tap slots read past the halo
    const size_t lds = ((size_t)Q * 8 * 16 * ntw + (size_t)2 * B.NPIXp) * 16;
    const long ntiles = (long)P.tiles_x * P.tiles_y * P.N;
    if (ntiles >= (1L << 30)) return -1;
    B.ntiles = (int)ntiles;
    B.out_bytes = (unsigned)((size_t)P.N * P.OH * P.OW * P.OC * sizeof(float));
    int grid = kNumCU - kNumCU % (8 * B.nsl);
    const long want = ((ntiles + 7) / 8) * 8 * B.nsl;
    if (grid == 0) return -1;
    if (want < grid) grid = (int)want;
    if (Q == 3) return rowsw_launch<4, 3>(B, lds, grid, s);
    if (Q == 5) return rowsw_launch<4, 5>(B, lds, grid, s);
    return -1;
  });
}

}  // namespace srk

#ifdef SRK_ROWSR_PROF
// phase-clock build only: device buffer of 8 x uint32 per wave (4 waves per block) that the next k_conv_rowsr launches fill
extern "C" void srk_debug_rowsr_prof(void* p) { srk::g_rowsr_prof = static_cast<unsigned*>(p); }
#endif
