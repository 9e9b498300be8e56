// Ring form of the wave-specialised persistent 3x3 convolution (k_conv_bfw, conv_bfw.hip): the same roles, arithmetic and
// accumulation order (bit-equal outputs), but producers and consumers no longer meet at a workgroup barrier every stage.
//
// Why (DESIGN 10.3 / 10.9 #1, round-4 profile): with two halo buffers and one s_barrier per (tile, chunk) stage every wave
// of the block runs in lockstep -- all consumer waves park a finished tile at the same moment (the matrix pipe idles), a
// stage lasts as long as its slowest wave, and the producers spend a third of it waiting.  Neither pipe was saturated
// (MFMA ~50 % busy, LDS ~30 %, HBM 3.4 TB/s).  Here:
//
//   * tiles are 128 pixels; the 8 consumer waves form TWO groups of 4 (waves 0-3 / 4-7; wave w and w + 4 share a SIMD)
//     that take alternate tiles of the block's list, so the two waves of a SIMD are in different phases of different
//     tiles: one wave's parking (descale / bias / activation, no MFMA) runs under the other's taps;
//   * the halo buffers form a ring of NBUF >= 3 slots in LDS, filled in the fixed global stage order
//     (pair p of tiles, chunk cc, group g): stage s lives in slot s % NBUF;
//   * per slot two monotonic counters in LDS: full[slot] += 1 per producer wave once its ds_writes have landed,
//     free[slot] += 1 per consumer wave once its fragment reads of the slot have returned.  Use k of a slot may be
//     filled when free >= 4 k and read when full >= 4 (k + 1).  A wave waits for exactly the data it needs
//     (ds_read_b32 + s_sleep polling); nobody waits for the block;
//   * producers keep TWO register sets of loads in flight (stages s + 1 and s + 2) and run ahead as far as the ring
//     allows; every load is issued unconditionally (stages past the end read through an out-of-range offset), so the
//     compiler counts the waits of a commit exactly (vmcnt = the other set's loads).
//
// Every poll is bounded (BFR_SPIN_CAP): a protocol error ends as wrong numbers and a non-zero srk_ring_timeouts(), not as
// a hung GPU.
#include "conv_bfw.h"

namespace srk {

// Ablation builds (tools/ring_ablate.sh; release: 0): 1 no global loads, 2 no stores, 4 no MFMAs, 8 no fragment reads,
// 16 no producer split + LDS commit, 32 no parking arithmetic
#ifndef BFR_ABL
#define BFR_ABL 0
#endif
constexpr int BFR_MAXBUF = 6;
constexpr int BFR_PIT = 3;  // producer register batches per set: halos of <= 64 * 3 = 192 pixels
constexpr unsigned BFR_SPIN_CAP = 1u << 18;

__device__ unsigned g_bfr_timeouts = 0;

typedef __attribute__((address_space(3))) unsigned bfr_cnt_t;

// (ablation builds) a fragment the compiler must treat as written / as read
__device__ __forceinline__ void bfr_touch(uint4& u) { asm volatile("" : "+v"(u.x), "+v"(u.y), "+v"(u.z), "+v"(u.w)); }
__device__ __forceinline__ void bfr_use(const uint4& u) { asm volatile("" ::"v"(u.x), "v"(u.y), "v"(u.z), "v"(u.w)); }

__device__ __forceinline__ unsigned bfr_peek(bfr_cnt_t* p) {
  return (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
// wait until *p >= target (monotonic counters: the difference stays far below 2^31)
__device__ __forceinline__ void bfr_wait(bfr_cnt_t* p, unsigned target, bool& dead) {
  if (!dead) {
    unsigned spins = 0;
    while ((int)(bfr_peek(p) - target) < 0) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > BFR_SPIN_CAP) {
        dead = true;
        break;
      }
    }
  }
  asm volatile("" ::: "memory");
}
// every LDS access this wave issued so far has completed, then one count
__device__ __forceinline__ void bfr_signal(bfr_cnt_t* p) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
}

template <int NTW, bool F16 = false, bool MASK = false, bool OMASK = false>
__global__ __launch_bounds__(768, 3) void k_conv_bfr(BfwParams B) {
  constexpr int TT = 9, MTW = 2;
  constexpr int NCW = 8, NGW = 4, NPW = 4;  // consumer waves (two groups of NGW), producer waves
  constexpr int NTHR = 64 * (NCW + NPW);
  constexpr int PSTEP = 16 * NPW, PIT = BFR_PIT;
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
  const MfmaConvParams& P = B.P;
  const int NB = B.NB;
  const int wslot = 8 * NB;  // uint4 per (tap, chunk): [plane 2][group 4][NB]
  const int hbuf = 8 * B.NPIXp;  // uint4 per halo buffer: [plane 2][group 4][NPIXp]
  uint4* wl = smem4;
  uint4* hal0 = smem4 + (size_t)TT * B.ICc * wslot;
  bfr_cnt_t* cnt = (bfr_cnt_t*)(hal0 + (size_t)B.nbuf * hbuf);  // full[BFR_MAXBUF], free[BFR_MAXBUF]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave >= NCW;
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
  const int sl = (blockIdx.x >> 3) % nsl, bi = (blockIdx.x >> 3) / nsl;  // slice, block index inside the XCD
  for (int e = tid; e < TT * B.ICc * wslot; e += NTHR) {
    const int slot = e / wslot, w = e - slot * wslot;
    const int t = slot / B.ICc, cc = slot - t * B.ICc;
    const int u = t / P.KWv, v = t - u * P.KWv;
    const int tapw = (P.wh0 + P.wdh * u) * P.KW_full + (P.ww0 + P.wdw * v);
    const int pg = w / NB, o = w - pg * NB;
    const int oc = sl * NB + o, ocb = oc / B.NBfull;
    wl[e] = B.wq[((size_t)(tapw * B.ICc + cc) * B.OCb + ocb) * (size_t)(8 * B.NBfull) + pg * B.NBfull + (oc - ocb * B.NBfull)];
  }
  if (tid < 2 * BFR_MAXBUF) cnt[tid] = 0u;
  // tiles of this block (XCD-aware contiguous ranges, as in k_conv_bfw): first, first + tstride, ... (count of them)
  const int nblk = gridDim.x;
  int first, count;
  {
    const int per_x = B.ntiles >> 3, rem_x = B.ntiles & 7;
    const int nb_x = ((nblk + 7 - xcd) >> 3) / nsl;
    const int tiles_x = per_x + (xcd < rem_x ? 1 : 0);
    const int start_x = xcd * per_x + (xcd < rem_x ? xcd : rem_x);
    first = start_x + bi;
    count = bi < tiles_x ? (tiles_x - bi + nb_x - 1) / nb_x : 0;
  }
  const int tstride = ((nblk + 7 - xcd) >> 3) / nsl;
  count = __builtin_amdgcn_readfirstlane(count);
  const int ICc = __builtin_amdgcn_readfirstlane(B.ICc);
  const int nbuf = __builtin_amdgcn_readfirstlane(B.nbuf);

  // Tile coordinates (image, tile row, tile column) of the two tiles of a pair, advanced by TWO list steps per pair:
  // add + carry on wave-uniform values, no divisions in the loops.
  const int img_tiles = P.tiles_x * P.tiles_y;
  int s2n, s2y, s2x;  // 2 * tstride as (images, tile rows, tile columns)
  int a_n, a_y, a_x, b_n, b_y, b_x;
  {
    auto split = [&](int t, int& n, int& y, int& x) {
      n = t / img_tiles;
      const int q = t - n * img_tiles;
      y = q / P.tiles_x;
      x = q - y * P.tiles_x;
      n = __builtin_amdgcn_readfirstlane(n);
      y = __builtin_amdgcn_readfirstlane(y);
      x = __builtin_amdgcn_readfirstlane(x);
    };
    split(2 * tstride, s2n, s2y, s2x);
    split(first, a_n, a_y, a_x);
    split(first + tstride, b_n, b_y, b_x);
  }
  auto adv2 = [&](int& n, int& y, int& x) {
    x += s2x;
    y += s2y;
    n += s2n;
    if (x >= P.tiles_x) {
      x -= P.tiles_x;
      ++y;
    }
    if (y >= P.tiles_y) {
      y -= P.tiles_y;
      ++n;
    }
  };
  bool dead = false;

  if (producer) {
    // ------------------------------------------------------------------ producers
    if (NTW <= 2 && __builtin_amdgcn_readfirstlane(tid) >= 64 * NCW) __builtin_amdgcn_s_setprio(1);  // (see k_conv_bfw)
    const int ptid = tid - 64 * NCW;
    const int g = ptid & 3, hp0 = ptid >> 2;
    const int hy0 = hp0 / P.HW, hx0 = hp0 - hy0 * P.HW;
    const int dyp = PSTEP / P.HW, dxp = PSTEP - dyp * P.HW;
    f32x4 pvA[PIT][2], pvB[PIT][2];
    f32x4 mkA[MASK ? PIT : 1][2], mkB[MASK ? PIT : 1][2];
    int it_rel[PIT], it_hx[PIT];
    {
      int hy = hy0, hx = hx0;
#pragma unroll
      for (int k = 0; k < PIT; ++k) {
        it_rel[k] = hp0 + PSTEP * k < npix ? ((hy * P.IW + hx) * P.IC + g * 8) * 4 : -1;
        it_hx[k] = hx;
        hy += dyp;
        hx += dxp;
        if (hx >= P.HW) {
          hx -= P.HW;
          ++hy;
        }
      }
    }
    const unsigned img_bytes = (unsigned)((size_t)P.IH * P.IW * P.IC * 4);
    auto bload = [](__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
      return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
    };
    auto rsrc_of = [](const float* base, unsigned bytes) {
      const unsigned long long a = reinterpret_cast<unsigned long long>(base);
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
      void* p = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
      return __builtin_amdgcn_make_buffer_rsrc(p, (short)0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
    };
    // global stage order: for every pair of tiles, for every chunk, group 0 then group 1 (a last single tile: group 0 only)
    int left = count, wcc = 0, wg = 0;
    auto issue = [&](f32x4 (&pv)[PIT][2], f32x4 (&mk)[MASK ? PIT : 1][2]) {
      const bool valid = left > 0;
      const bool second = wg == 1;
      const int n = valid ? (second ? b_n : a_n) : 0;
      const int r0 = (second ? b_y : a_y) * P.TH, c0 = (second ? b_x : a_x) * P.TW;
      const int cc = wcc;
      if (valid) {
        const int ng = left >= 2 ? 2 : 1;
        if (++wg == ng) {
          wg = 0;
          if (++wcc == ICc) {
            wcc = 0;
            left -= ng;
            adv2(a_n, a_y, a_x);
            adv2(b_n, b_y, b_x);
          }
        }
      }
      constexpr unsigned OOB = 0x80000000u;
      const int iyb = r0 * P.is + P.iy0, ixb = c0 * P.is + P.ix0;
      const bool ch_on = valid && cc * 32 + g * 8 + 7 < P.IC;
      const size_t img = (size_t)n * P.IH * P.IW * P.IC;
      const __amdgpu_buffer_rsrc_t rin = rsrc_of(P.in + img, img_bytes);
      const __amdgpu_buffer_rsrc_t rmk = rsrc_of(MASK ? P.mask_y + img : P.in, MASK ? img_bytes : 0u);
      (void)rmk;
      const int obase = ((iyb * P.IW + ixb) * P.IC + cc * 32) * 4;  // may be negative: rows above the image wrap out of range
#pragma unroll
      for (int k = 0; k < PIT; ++k) {
        const bool ok = it_rel[k] >= 0 && ch_on && (unsigned)(ixb + it_hx[k]) < (unsigned)P.IW;
        const unsigned o = ok ? (unsigned)(obase + it_rel[k]) : OOB;
        if constexpr (BFR_ABL & 1) {
          pv[k][0] = pv[k][1] = (f32x4){(float)o, 1.f, 2.f, 3.f};
          if constexpr (MASK) mk[k][0] = mk[k][1] = pv[k][0];
        } else {
          pv[k][0] = bload(rin, o);
          pv[k][1] = bload(rin, o + 16u);
          if constexpr (MASK) {
            mk[k][0] = bload(rmk, o);
            mk[k][1] = bload(rmk, o + 16u);
          }
        }
      }
    };
    auto commit = [&](const f32x4 (&pv)[PIT][2], const f32x4 (&mk)[MASK ? PIT : 1][2], uint4* hal) {
      if constexpr (BFR_ABL & 16) {
#pragma unroll
        for (int k = 0; k < PIT; ++k) asm volatile("" ::"v"(pv[k][0]), "v"(pv[k][1]));
        return;
      }
#pragma unroll
      for (int k = 0; k < PIT; ++k) {
        const int hq = hp0 + PSTEP * k;
        if (hq < npix) {
          float f[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f[e] = pv[k][0][e];
            f[4 + e] = pv[k][1][e];
            if constexpr (MASK) {
              f[e] = mk[k][0][e] > 0.f ? f[e] : f[e] * P.mask_slope;
              f[4 + e] = mk[k][1][e] > 0.f ? f[4 + e] : f[4 + e] * P.mask_slope;
            }
          }
          uint4 pl[2];
          if constexpr (F16) split8h(f, sx, pl); else split8n<2>(f, pl);
          hal[(0 * 4 + g) * B.NPIXp + hq] = pl[0];
          hal[(1 * 4 + g) * B.NPIXp + hq] = pl[1];
        }
      }
    };
    const int S = count * ICc;
    issue(pvA, mkA);
    issue(pvB, mkB);
    __syncthreads();  // filter and counters visible
    int b = 0;
    unsigned k = 0;  // slot and use count of the stage about to be committed
    for (int s = 0; s < S; s += 2) {
      bfr_wait(cnt + BFR_MAXBUF + b, NGW * k, dead);
      commit(pvA, mkA, hal0 + (size_t)b * hbuf);
      bfr_signal(cnt + b);
      if (++b == nbuf) {
        b = 0;
        ++k;
      }
      issue(pvA, mkA);  // stage s + 2
      if (s + 1 < S) {
        bfr_wait(cnt + BFR_MAXBUF + b, NGW * k, dead);
        commit(pvB, mkB, hal0 + (size_t)b * hbuf);
        bfr_signal(cnt + b);
        if (++b == nbuf) {
          b = 0;
          ++k;
        }
      }
      issue(pvB, mkB);  // stage s + 3
    }
    if (dead && lane == 0) atomicAdd(&g_bfr_timeouts, 1u);
    return;
  }

  // -------------------------------------------------------------------- consumers
  const int grp = __builtin_amdgcn_readfirstlane(wave >> 2), gw = wave & 3;
  const int pj = !B.perm ? j : (j < 4 ? 2 * j : (j < 12 ? 2 * j - 7 : 2 * j - 16));
  int hp[MTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) {
    int m = gw * (16 * MTW) + mt * 16 + pj;
    if (m >= npx) m = 0;
    const int r = m / P.TW, c = m - r * P.TW;
    hp[mt] = (r * P.is) * P.HW + c * P.is + kq * B.NPIXp;
  }
  const bool wave_live = gw * (16 * MTW) < npx;
  const int plane = 4 * B.NPIXp;
  int wrow[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) wrow[nt] = kq * NB + nt * 16 + j;
  f32x4 bias4[NTW];
  int coff[NTW], poff[MTW];
  int pix_ok[MTW];
  {
    const EpiTile e0 = epi_tile_setup(P, 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
      const int m = gw * (16 * MTW) + mt * 16 + pj;
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
  f32x4 acc[NTW][MTW];
  f32x4 pend[NTW][MTW];  // the finished tile, stored one slot per tap under the next stage's MFMAs (see k_conv_bfw)
  f32x4 om[OMASK ? NTW : 1][OMASK ? MTW : 1];
  float amax = 0.f;
  constexpr unsigned kDrop = 0x80000000u;
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(P.out, 0, B.out_bytes, 0x00020000);
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  unsigned pend_voff[MTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) pend_voff[mt] = kDrop;
  int pend_mask = 0;
  bool pend_live = false;
  constexpr int NST = NTW * MTW;  // stores per tile and lane, slot q = mt * NTW + nt
  static_assert(NST <= TT, "every pending store must find a tap");
  auto store_slot = [&](auto qc) {
    constexpr int q = decltype(qc)::value;
    constexpr int mt = q / NTW, nt = q - mt * NTW;
    const f32x4 v = pend[nt][mt];
    if constexpr (BFR_ABL & 2) {
      asm volatile("" ::"v"(v));
      return;
    }
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), orsrc, (int)(pend_voff[mt] + 4u * (unsigned)coff[nt]), 0, 0);
  };
  auto park = [&](int n, int r0, int c0) {
    const unsigned tile_off = 4u * (unsigned)epi_tile_setup(P, n, r0, c0).off0;
    pend_mask = 0;
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
      const int r = pix_ok[mt] >> 16, c = pix_ok[mt] & 0xffff;
      const bool pok = pix_ok[mt] >= 0 && r0 + r < P.PH && c0 + c < P.PW;
      if (pok) pend_mask |= 1 << mt;
      pend_voff[mt] = pok ? tile_off + 4u * (unsigned)poff[mt] : kDrop;
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        f32x4 v = acc[nt][mt];
        if constexpr (BFR_ABL & 32) {
          pend[nt][mt] = v;
          continue;
        }
        if constexpr (F16) v *= dsc;
        v += bias4[nt];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : act_slope * v[e];
        if constexpr (OMASK) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = om[nt][mt][e] > 0.f ? v[e] : 0.f;
        }
        pend[nt][mt] = v;
        if (P.ep.y_amax && ((pend_mask >> mt) & 1)) amax = abs_max4(amax, v);
      }
    }
    pend_live = true;
  };
  // (everything this wave loaded so far has landed before the loop: see k_conv_bfw)
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) asm volatile("" ::"v"(bias4[nt]));
  asm volatile("" ::"v"(act_slope));
  __syncthreads();  // filter and counters visible

  // own tiles: list entries grp, grp + 2, ...; stage (entry i, chunk cc) is number (i >> 1) * 2 ICc + cc * ng + (i & 1) of
  // the global order, ng = tiles of the pair
  int o_n = grp ? b_n : a_n, o_y = grp ? b_y : a_y, o_x = grp ? b_x : a_x;
  int b = grp;       // ring slot of the next own stage (nbuf >= 3 > grp)
  unsigned k = 0;    // ... and how often that slot was used before
  for (int ti = grp; ti < count; ti += 2) {
    const int ng = (ti | 1) < count ? 2 : 1;
    const int n = o_n, r0 = o_y * P.TH, c0 = o_x * P.TW;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int cc = 0; cc < ICc; ++cc) {
      if constexpr (OMASK) {
        if (cc == ICc - 1 && wave_live) {
          const float* ob = P.ep.out_relu + epi_tile_setup(P, n, r0, c0).off0;
#pragma unroll
          for (int mt = 0; mt < MTW; ++mt) {
            const int r = pix_ok[mt] >> 16, c = pix_ok[mt] & 0xffff;
            const bool ok = pix_ok[mt] >= 0 && r0 + r < P.PH && c0 + c < P.PW;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
              om[nt][mt] = *reinterpret_cast<const f32x4*>(ok ? ob + (coff[nt] + poff[mt]) : P.ep.out_relu);
          }
        }
      }
      bfr_wait(cnt + b, NPW * (k + 1), dead);
      if (wave_live) {
        const uint4* hal = hal0 + (size_t)b * hbuf;
        const uint4* wb = wl + (size_t)cc * wslot;
        const size_t wstep = (size_t)ICc * wslot;
        uint4 fa[2][2][NTW], fb[2][2][MTW];  // [buffer][plane][tile]
        int wt_toff = 0, wt_tv = 0, wt_t = 0;
        auto load_frags = [&](uint4 (&a)[2][NTW], uint4 (&bb)[2][MTW]) {
          if constexpr (BFR_ABL & 8) {
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) { bfr_touch(bb[0][mt]); bfr_touch(bb[1][mt]); }
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) { bfr_touch(a[0][nt]); bfr_touch(a[1][nt]); }
            return;
          }
          const uint4* hb = hal + wt_toff;
          const uint4* wt = wb + (size_t)wt_t * wstep;
#pragma unroll
          for (int mt = 0; mt < MTW; ++mt) {
            bb[0][mt] = hb[hp[mt]];
            bb[1][mt] = hb[hp[mt] + plane];
          }
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) {
            a[0][nt] = wt[wrow[nt]];
            a[1][nt] = wt[4 * NB + wrow[nt]];
          }
          ++wt_t;
          ++wt_toff;
          if (++wt_tv == P.KWv) {
            wt_tv = 0;
            wt_toff += P.HW - P.KWv;
          }
        };
        auto mfmas = [&](const uint4 (&a)[2][NTW], const uint4 (&bb)[2][MTW]) {
          if constexpr (BFR_ABL & 4) {
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) { bfr_use(bb[0][mt]); bfr_use(bb[1][mt]); }
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) { bfr_use(a[0][nt]); bfr_use(a[1][nt]); }
            return;
          }
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) acc[nt][mt] = mfma16x<F16>(a[0][nt], bb[1][mt], acc[nt][mt]);  // w_h * x_m
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) acc[nt][mt] = mfma16x<F16>(a[1][nt], bb[0][mt], acc[nt][mt]);  // w_m * x_h
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) acc[nt][mt] = mfma16x<F16>(a[0][nt], bb[0][mt], acc[nt][mt]);  // w_h * x_h
        };
        load_frags(fa[0], fb[0]);
        srk_static_for<0, TT>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          if (t + 1 < TT) load_frags(fa[(t + 1) & 1], fb[(t + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
          mfmas(fa[t & 1], fb[t & 1]);
          __builtin_amdgcn_sched_barrier(0);
          if (pend_live) {
            srk_static_for<t, (t + 1 < NST ? t + 1 : NST)>([&](auto qc) { store_slot(qc); });
          }
        });
        pend_live = false;
      }
      bfr_signal(cnt + BFR_MAXBUF + b);  // (the last tap's fragments were operands of MFMAs issued above: the reads have returned)
      b += cc + 1 < ICc ? ng : 2 * ICc - (ICc - 1) * ng;
      while (b >= nbuf) {
        b -= nbuf;
        ++k;
      }
    }
    if (wave_live) park(n, r0, c0);
    adv2(o_n, o_y, o_x);
  }
  if (pend_live) srk_static_for<0, NST>([&](auto qc) { store_slot(qc); });
  if (P.ep.y_amax) amax_commit(P.ep.y_amax, amax, blockIdx.x + wave, amax_peek(P.ep.y_amax, blockIdx.x + wave));
  if (dead && lane == 0) atomicAdd(&g_bfr_timeouts, 1u);
}

template <int NTW>
static int bfr_launch_t(const BfwParams& B, size_t lds, int grid, hipStream_t s) {
  note_amax_written(B.P.ep.y_amax != nullptr);
  const dim3 blk(768);
  if constexpr (NTW == 2) {  // data gradients (bf16x3): mask on dy and / or ReLU gradient on dx
    if (B.P.mask_y && B.P.ep.out_relu) {
      static LdsLimit limb;
      limb.ensure(reinterpret_cast<const void*>(&k_conv_bfr<NTW, false, true, true>), lds);
      note_kernel("k_conv_bfr<%d,mask,relu>", NTW);
      hipLaunchKernelGGL((k_conv_bfr<NTW, false, true, true>), dim3(grid), blk, lds, s, B);
      return check_launch("conv_bfr");
    }
    if (B.P.mask_y) {
      static LdsLimit limm;
      limm.ensure(reinterpret_cast<const void*>(&k_conv_bfr<NTW, false, true, false>), lds);
      note_kernel("k_conv_bfr<%d,mask>", NTW);
      hipLaunchKernelGGL((k_conv_bfr<NTW, false, true, false>), dim3(grid), blk, lds, s, B);
      return check_launch("conv_bfr");
    }
    if (B.P.ep.out_relu) {
      static LdsLimit limo;
      limo.ensure(reinterpret_cast<const void*>(&k_conv_bfr<NTW, false, false, true>), lds);
      note_kernel("k_conv_bfr<%d,relu>", NTW);
      hipLaunchKernelGGL((k_conv_bfr<NTW, false, false, true>), dim3(grid), blk, lds, s, B);
      return check_launch("conv_bfr");
    }
  }
  if (B.P.mask_y || B.P.ep.out_relu) return -1;
  if (B.w_descale) {  // f16x3 arithmetic
    static LdsLimit limh;
    limh.ensure(reinterpret_cast<const void*>(&k_conv_bfr<NTW, true>), lds);
    note_kernel("k_conv_bfr<%d,f16>", NTW);
    hipLaunchKernelGGL((k_conv_bfr<NTW, true>), dim3(grid), blk, lds, s, B);
    return check_launch("conv_bfr");
  }
  static LdsLimit lim;
  lim.ensure(reinterpret_cast<const void*>(&k_conv_bfr<NTW>), lds);
  note_kernel("k_conv_bfr<%d>", NTW);
  hipLaunchKernelGGL((k_conv_bfr<NTW>), dim3(grid), blk, lds, s, B);
  return check_launch("conv_bfr");
}

// B: the launch as conv_bfw_gather prepared it up to the tile choice (P, wq, ICc, NB, nsl, ...).  Picks the 128-pixel
// tile and the ring depth; -1 when the layer is not one of the ring kernel's (3x3, 32 or 48 channels per slice, a ring of
// at least three slots beside the filter).  SRK_BFR=0: never.
int conv_bfr_launch(const BfwParams& B0, hipStream_t s) {
  if (env_int("SRK_BFR", 1) == 0) return -1;
  BfwParams B = B0;
  MfmaConvParams& P = B.P;
  const int ntw = B.NB / 16;
  if (P.KHv != 3 || P.KWv != 3 || P.is != 1 || (ntw != 2 && ntw != 3) || B.NB % 16 != 0) return -1;
  if ((P.mask_y || P.ep.out_relu) && ntw != 2) return -1;
  const size_t wbytes = (size_t)9 * B.ICc * 8 * B.NB * 16;
  const long lds_cap = 160L * 1024 - 512 - 2 * BFR_MAXBUF * 4;
  long px_cap = (lds_cap - (long)wbytes) / (3 * 128);  // three slots at least
  if (px_cap > 64 * BFR_PIT) px_cap = 64 * BFR_PIT;
  if (px_cap < 64) return -1;
  TilePick best{};
  if (!bfw_pick_tile(128, P.PH, P.PW, 3, 3, px_cap, B.perm, 1, best)) return -1;
  P.TH = best.TH; P.TW = best.TW; P.tiles_y = best.tiles_y; P.tiles_x = best.tiles_x; P.HH = best.HH; P.HW = best.HW;
  B.NPIXp = bfw_group_stride(best.HH * best.HW, B.perm);
  const size_t slot_bytes = (size_t)8 * B.NPIXp * 16;
  long nbuf = (lds_cap - (long)wbytes) / (long)slot_bytes;
  const int want = env_int("SRK_BFR_NBUF", 0);
  if (want >= 3 && want < nbuf) nbuf = want;
  if (nbuf > BFR_MAXBUF) nbuf = BFR_MAXBUF;
  if (nbuf < 3) return -1;
  B.nbuf = (int)nbuf;
  const size_t lds = wbytes + (size_t)nbuf * slot_bytes + 2 * BFR_MAXBUF * 4;
  const long ntiles = (long)P.tiles_x * P.tiles_y * P.N;
  if (ntiles >= (1L << 29)) return -1;
  B.ntiles = (int)ntiles;
  B.out_bytes = (unsigned)((size_t)P.N * P.OH * P.OW * P.OC * sizeof(float));
  const int nsl = B.nsl;
  int grid = kNumCU;
  if (nsl > 1) {
    grid -= grid % (8 * nsl);
    const long want_g = ((ntiles + 7) / 8) * 8 * nsl;
    if (grid == 0) return -1;
    if (want_g < grid) grid = (int)want_g;
  } else if (grid > ntiles) {
    grid = (int)ntiles;
  }
  if (B.dbg & 32)
    fprintf(stderr, "[srk] k_conv_bfr<%d>: lds %zu B (filter %zu), ring %d x %zu B, grid %d of %ld tiles x %d slices, tile %dx%d halo %dx%d\n",
            ntw, lds, wbytes, B.nbuf, slot_bytes, grid, ntiles, nsl, P.TH, P.TW, P.HH, P.HW);
  return ntw == 2 ? bfr_launch_t<2>(B, lds, grid, s) : bfr_launch_t<3>(B, lds, grid, s);
}

}  // namespace srk

// Polls of k_conv_bfr that ran into their iteration cap since the last reset (0 in a correct library; a diagnostic for
// tests and fuzzers -- synchronises the device).
extern "C" int srk_ring_timeouts(int reset) {
  unsigned v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(srk::g_bfr_timeouts), sizeof(v)) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  if (reset && v) {
    const unsigned z = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(srk::g_bfr_timeouts), &z, sizeof(z));
  }
  return (int)v;
}
