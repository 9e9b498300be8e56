// Wave-specialised persistent bf16x3 convolution for layers whose whole prepared filter fits in LDS next to TWO
// halo-chunk buffers (the ESPCN 64->32 and 32->48 3x3 layers: the c2 benchmark).
//
// Why: on those layers every phase of the per-tile kernels is short and the phases do not overlap -- measured on
// the 64->32 layer (SRK_DBG ablations): halo staging 0.19 ms (= the activation read at HBM speed), MFMAs 0.17 ms,
// epilogue 0.08 ms, per-tap filter copies 0.07 ms, loop/barrier skeleton 0.11 ms, total 0.57 ms; co-resident blocks
// run the same phases in lockstep.  Here the phases run CONCURRENTLY on different waves of one block:
//
//   * one block per CU: 8 (or 4) CONSUMER waves of 32 (64) pixels x all 16*NTW output channels each, plus 4 PRODUCER
//     waves;
//   * the filter is copied to LDS once per block (once, not per tap); the block then walks (tile, 32-channel chunk)
//     stages; stage s computes from halo buffer s&1 while the producers fill buffer (s+1)&1 -- one barrier per stage;
//   * producers: global fp32 loads of stage s+2 are issued right after the LDS commit of stage s+1 (a full stage
//     of latency cover), commit = bf16 split + ds_write_b128: their VALU work interleaves with the consumers'
//     MFMAs on the same SIMD;
//   * consumers: fragment reads + MFMAs only (software-pipelined tap loop, no global-memory waits), then -- after
//     the last chunk of a tile -- the epilogue straight from the accumulators, while the producers already stage the
//     next tile.  The MFMA runs transposed (D[channel][pixel]: A = filter, B = pixels): lane (pixel j, group kq)
//     holds 4 consecutive output channels per 16-channel tile, so its stores are 16-byte vectors and the 4 kq lanes
//     of a pixel write 64 contiguous bytes per instruction -- no LDS transpose (there is no LDS left for one:
//     filter 72 KB + 2 x 42 KB halo buffers).
//
// Arithmetic identical to k_conv_bf3 (x = h + m, products m*h + h*m + h*h, fp32 accumulate; the
// accumulation order over (chunk, tap) is the same).
#include "conv_bfw.h"

namespace srk {

#ifdef SRK_EXPERIMENTS
#define BFW_CLK() clock64()
static long long* g_bfw_prof = nullptr;
#else
#define BFW_CLK() 0ll
#endif

template <int NTW, int TT, int MTW, bool F16 = false, bool MASK = false, bool OMASK = false, int NPW = 4>
__global__ __launch_bounds__(64 * (16 / MTW + NPW), (16 / MTW + NPW) / 4) void k_conv_bfw(BfwParams B) {
  constexpr int NCW = 16 / MTW;          // consumer waves: MTW 16-pixel groups each, 256 pixels per block
  constexpr int NTHR = 64 * (NCW + NPW);   // + NPW producer waves (4; 8 measured in round 4: 481 -> 489 us on the c2 64->32 layer)
  constexpr int PSTEP = 16 * NPW;        // halo pixels the producers cover per register batch
  constexpr int PIT = BFW_IT * 4 / NPW;  // register batches per producer thread
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
  const MfmaConvParams& P = B.P;
  const int NB = B.NB;
  const int wslot = 8 * NB;  // uint4 per (tap, chunk): [plane 2][group 4][NB]
  const int T = P.KHv * P.KWv;
  const int hbuf = 8 * B.NPIXp;  // uint4 per halo buffer: [plane 2][group 4][NPIXp]
  uint4* wl = smem4;
  uint4* hal0 = smem4 + (size_t)T * B.ICc * wslot;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave >= NCW;
  const int j = lane & 15, kq = lane >> 4;
  const int npx = P.TH * P.TW, npix = P.HH * P.HW;
  // f16x3 (SRK_ALGO_MFMA_F16X3): activation scale 2^kx from the input's running maximum, descale 2^-(kx + kw)
  float sx = 1.f, dsc = 1.f;
  if constexpr (F16) {
    const int kx = amax_scale_exp(amax_read(P.ep.x_amax));
    sx = exp2i(kx);
    dsc = exp2i(-kx) * B.w_descale[0];
  }

  const int nsl = B.nsl;
  const int xcd = blockIdx.x & 7;
  const int sl = (blockIdx.x >> 3) % nsl, bi = (blockIdx.x >> 3) / nsl;  // slice, block index inside the XCD
  for (int e = tid; e < T * B.ICc * wslot; e += NTHR) {
    const int slot = e / wslot, w = e - slot * wslot;
    const int t = slot / B.ICc, cc = slot - t * B.ICc;
    const int u = t / P.KWv, v = t - u * P.KWv;
    const int tapw = (P.wh0 + P.wdh * u) * P.KW_full + (P.ww0 + P.wdw * v);
    const int pg = w / NB, o = w - pg * NB;  // (plane, group) row of the slot, channel inside the slice
    // packed layout [tap][chunk][64-channel block][plane][group][NBfull channels]
    const int oc = sl * NB + o, ocb = oc / B.NBfull;
    wl[e] = B.wq[((size_t)(tapw * B.ICc + cc) * B.OCb + ocb) * (size_t)(8 * B.NBfull) + pg * B.NBfull + (oc - ocb * B.NBfull)];
  }
  // tiles of this block: XCD-aware order (consecutive block ids go round-robin over the 8 XCDs / L2s; give each XCD
  // a contiguous range of tiles so neighbouring tiles share halo rows in one L2)
  const int nblk = gridDim.x;
  int first, count;
  {
    const int per_x = B.ntiles >> 3, rem_x = B.ntiles & 7;  // tiles per XCD
    const int nb_x = ((nblk + 7 - xcd) >> 3) / nsl;         // blocks per slice on this XCD (host: grid % (8 nsl) == 0)
    const int tiles_x = per_x + (xcd < rem_x ? 1 : 0);
    const int start_x = xcd * per_x + (xcd < rem_x ? xcd : rem_x);
    first = start_x + bi;  // block bi takes tiles start_x + bi, + nb_x, ...
    count = bi < tiles_x ? (tiles_x - bi + nb_x - 1) / nb_x : 0;
  }
  const int tstride = ((nblk + 7 - xcd) >> 3) / nsl;
  const int S = count * B.ICc;  // stages of this block

  // Stage walk (tile first + i * tstride, chunk cc) without divisions in the loop: a 32-bit division is ~40 VALU
  // instructions, and the per-stage decode had four of them in every wave.  The tile step is split once into (images,
  // tile rows, tile columns); an advance is add + carry on wave-uniform values.  decode() returns the stages in order.
  const int img_tiles = P.tiles_x * P.tiles_y;
  const int st_n = tstride / img_tiles, st_y = (tstride - st_n * img_tiles) / P.tiles_x,
            st_x = tstride - st_n * img_tiles - st_y * P.tiles_x;
  int w_n, w_y, w_x, w_cc = 0;
  {
    w_n = first / img_tiles;
    const int q = first - w_n * img_tiles;
    w_y = q / P.tiles_x;
    w_x = q - w_y * P.tiles_x;
    w_n = __builtin_amdgcn_readfirstlane(w_n);
    w_y = __builtin_amdgcn_readfirstlane(w_y);
    w_x = __builtin_amdgcn_readfirstlane(w_x);
  }
  auto decode = [&](int& n, int& r0, int& c0, int& cc) {
    n = w_n;
    r0 = w_y * P.TH;
    c0 = w_x * P.TW;
    cc = w_cc;
    if (++w_cc == B.ICc) {
      w_cc = 0;
      w_x += st_x;
      w_y += st_y;
      w_n += st_n;
      if (w_x >= P.tiles_x) {
        w_x -= P.tiles_x;
        ++w_y;
      }
      if (w_y >= P.tiles_y) {
        w_y -= P.tiles_y;
        ++w_n;
      }
    }
  };

  if (producer) {
    // ------------------------------------------------------------------ producers
    // Narrow layers (<= 32 output channels: few MFMAs per staged byte) run closer to the HBM side of the kernel: their
    // producer waves get issue priority over the MFMA waves of the SIMD, so that the next stage's loads leave as soon as
    // the halo buffer is free (64->32 layer of c2: 0.471 -> 0.437 ms; the 48-channel pixel-shuffle layer LOSES 9 % with
    // it -- its consumers also carry the 756 MB store stream -- and priority on the consumers changes nothing).
    // s_setprio ignores EXEC, so the branch must be provably wave-uniform (readfirstlane).  SRK_DBG & 2048: off.
    if (NTW <= 2 && __builtin_amdgcn_readfirstlane(tid) >= 64 * NCW && !(SRK_KDBG(B.dbg) & 2048)) __builtin_amdgcn_s_setprio(1);
    const int ptid = tid - 64 * NCW;
    const int g = ptid & 3, hp0 = ptid >> 2;
    const int hy0 = hp0 / P.HW, hx0 = hp0 - hy0 * P.HW;
    const int dyp = PSTEP / P.HW, dxp = PSTEP - dyp * P.HW;
    // (Two register sets -- the loads of stage s+2 issued BEFORE stage s+1 is committed, a whole stage to land instead of
    // issue time + barrier wait -- were measured again in round 4 on top of the counted waits: 473 vs 474 us and 356 vs 356
    // us on the two c2 layers.  The producers are not what a stage waits for: after the rewrite of issue() below their wave
    // spends 35 % of a stage at the barrier, tools/bfw_prof.py.)
    f32x4 pv0[PIT], pv1[PIT];
    f32x4 mk0[MASK ? PIT : 1], mk1[MASK ? PIT : 1];  // MASK: y of the forward layer (dx = conv^T(dy * act'(y)))
    // Round 4: everything about the thread's items that does not depend on the stage is computed ONCE -- byte offset
    // relative to the halo origin (-1: no item), halo column -- and the loads go, unconditionally, through a buffer
    // descriptor of the image: a row above / below the image is an offset outside the descriptor (reads zero), a column
    // left / right of it, a missing item or a channel group beyond IC gets one by a select.  Per stage and item that
    // leaves one add, one compare and one select in front of two loads.  (Before: a 64-bit multiply chain and a
    // divergent branch around every pair of loads -- ~35 VALU instructions per item, 1450 of the producer wave's 6300
    // clocks per stage by the clock64 stamps of tools/bfw_prof.py, and an s_waitcnt vmcnt(0) in front of the first
    // conversion of the commit because loads under a branch cannot be counted.)
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
    const unsigned img_bytes = (unsigned)((size_t)P.IH * P.IW * P.IC * 4);   // (host: below 2 GiB)
    auto bload = [](__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
      typedef unsigned bfw_u32x4 __attribute__((ext_vector_type(4)));
      return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
    };
    auto rsrc_of = [](const float* base, unsigned bytes) {
      const unsigned long long a = reinterpret_cast<unsigned long long>(base);
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
      void* p = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
      return __builtin_amdgcn_make_buffer_rsrc(p, (short)0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
    };
    auto issue = [&]() {
      int n, r0, c0, cc;
      decode(n, r0, c0, cc);
      if (SRK_KDBG(B.dbg) & 1) return;
      constexpr unsigned OOB = 0x80000000u;
      const int iyb = r0 * P.is + P.iy0, ixb = c0 * P.is + P.ix0;
      const bool ch_on = cc * 32 + g * 8 + 7 < P.IC;
      const size_t img = (size_t)n * P.IH * P.IW * P.IC;
      const __amdgpu_buffer_rsrc_t rin = rsrc_of(P.in + img, img_bytes);
      const __amdgpu_buffer_rsrc_t rmk = rsrc_of(MASK ? P.mask_y + img : P.in, MASK ? img_bytes : 0u);
      (void)rmk;
      const int obase = ((iyb * P.IW + ixb) * P.IC + cc * 32) * 4;   // may be negative: rows above the image wrap out of range
#pragma unroll
      for (int k = 0; k < PIT; ++k) {
        const bool ok = it_rel[k] >= 0 && ch_on && (unsigned)(ixb + it_hx[k]) < (unsigned)P.IW;
        const unsigned o = ok ? (unsigned)(obase + it_rel[k]) : OOB;
        pv0[k] = bload(rin, o);
        pv1[k] = bload(rin, o + 16u);
        if constexpr (MASK) {
          mk0[k] = bload(rmk, o);
          mk1[k] = bload(rmk, o + 16u);
        }
      }
    };
    auto commit = [&](uint4* hal) {
      if (SRK_KDBG(B.dbg) & 16) return;
#pragma unroll
      for (int k = 0; k < PIT; ++k) {
        const int hq = hp0 + PSTEP * k;
        if (hq < npix) {
          float f[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f[e] = pv0[k][e];
            f[4 + e] = pv1[k][e];
            if constexpr (MASK) {
              f[e] = mk0[k][e] > 0.f ? f[e] : f[e] * P.mask_slope;
              f[4 + e] = mk1[k][e] > 0.f ? f[4 + e] : f[4 + e] * P.mask_slope;
            }
          }
          uint4 pl[2];
          if constexpr (F16) split8h(f, sx, pl); else split8n<2>(f, pl);
          hal[(0 * 4 + g) * B.NPIXp + hq] = pl[0];
          hal[(1 * 4 + g) * B.NPIXp + hq] = pl[1];
        }
      }
    };
    long long pt_commit = 0, pt_issue = 0, pt_wait = 0;
    if (S > 0 && T > 0) {
      issue();
      commit(hal0);
      if (S > 1) issue();
    }
    __syncthreads();  // filter, tap table and stage 0 visible
    const long long pt_begin = BFW_CLK();
    for (int s = 0; s < S; ++s) {
      const long long c0 = BFW_CLK();
      if (T > 0) {
        if (s + 1 < S) commit(hal0 + (size_t)((s + 1) & 1) * hbuf);
      }
      const long long c1 = BFW_CLK();
      if (T > 0) {
        if (s + 2 < S) issue();
      }
      const long long c2 = BFW_CLK();
      __syncthreads();
      const long long c3 = BFW_CLK();
      pt_commit += c1 - c0;
      pt_issue += c2 - c1;
      pt_wait += c3 - c2;
    }
#ifdef SRK_EXPERIMENTS
    if (B.prof && tid == 64 * NCW) {
      long long* pr = B.prof + (size_t)blockIdx.x * 16;
      pr[0] = pt_commit; pr[1] = pt_issue; pr[2] = pt_wait; pr[3] = S; pr[4] = BFW_CLK() - pt_begin;
    }
#endif
    (void)pt_commit; (void)pt_issue; (void)pt_wait; (void)pt_begin;
    return;
  }

  // -------------------------------------------------------------------- consumers
  const int pw = wave;
  // pixel of the M tile that this lane's MFMA column holds (see bfw_group_stride)
  const int pj = !B.perm ? j : (j < 4 ? 2 * j : (j < 12 ? 2 * j - 7 : 2 * j - 16));
  int hp[MTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) {
    int m = pw * (16 * MTW) + mt * 16 + pj;
    if (m >= npx) m = 0;
    const int r = m / P.TW, c = m - r * P.TW;
    hp[mt] = (r * P.is) * P.HW + c * P.is + kq * B.NPIXp;
  }
  const bool wave_live = pw * (16 * MTW) < npx;
  const int plane = 4 * B.NPIXp;
  // transposed MFMA (A = filter, B = pixels): C/D col = lane & 15 = pixel, rows kq*4 + reg of M tile nt = the 4
  // consecutive output channels nt*16 + kq*4 + reg -> one 16-byte store per (pixel, tile); the 4 kq lanes of a pixel
  // cover 64 contiguous bytes per store instruction
  int wrow[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) wrow[nt] = kq * NB + nt * 16 + j;
  // Epilogue data that does not depend on the tile: bias of the lane's 4 channels per M tile, and the element
  // offset of (pixel mt, channel group nt) relative to the tile origin (epi_tile_setup / epi_col_setup arithmetic:
  // plain NHWC or the fused pixel shuffle).  Only none / ReLU / leaky / scalar-PReLU activations reach this kernel
  // (conv_bfw_applicable), which makes the activation one branch-free select: v > 0 ? v : slope * v.
  f32x4 bias4[NTW];
  int coff[NTW], poff[MTW];  // element offset of slot (mt, nt) from the tile origin = coff[nt] + poff[mt]
  int pix_ok[MTW];  // (r << 16) | c of the lane's pixel in tile coordinates, -1 = beyond the tile
  {
    const EpiTile e0 = epi_tile_setup(P, 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
      const int m = pw * (16 * MTW) + mt * 16 + pj;
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
  // Deferred epilogue: the finished tile's values wait in `pend` and are stored one or two at a time between the taps
  // of the NEXT stage.  (A burst of 4*NTW KB per wave right after the last tap stalls the wave on the CU's write path
  // for ~2000 cycles per tile -- measured 0.06-0.09 ms per layer; spread out, the stores ride under the MFMAs.)
  // Needs the tap loop unrolled at compile time (TT = taps per chunk; TT = 0: dynamic loop, immediate epilogue).
  f32x4 pend[NTW][MTW];
  // OMASK (ep.out_relu: the data gradient leaves already multiplied by the ReLU gradient of the layer that produced this
  // conv's input): that tensor's values at the tile's output positions are requested when the tile's LAST chunk stage
  // begins -- unconditional loads, clamped to the tensor's first element for positions outside the tile -- and meet the
  // accumulators when the tile is parked, nine taps later.
  f32x4 om[OMASK ? NTW : 1][OMASK ? MTW : 1];
  float amax = 0.f;  // running maximum of what this lane stores (ep.y_amax)
  // Stores go through a buffer descriptor of the output with 32-bit byte offsets, unconditionally: a pixel outside the
  // tile / the image carries an offset beyond the tensor, which the buffer unit drops.  (With 64-bit addresses under
  // exec-mask branches the 48-channel f16 variant spilled two offset registers; the reload in front of a deferred store
  // came with an s_waitcnt vmcnt(0), i.e. a wait for the acknowledgement of every earlier store, in every stage.)
  constexpr unsigned kDrop = 0x80000000u;  // (host: outputs below 2 GiB)
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(P.out, 0, B.out_bytes, 0x00020000);
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  unsigned pend_voff[MTW];  // byte offset of the lane's pixel mt of the parked tile, or kDrop
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) pend_voff[mt] = kDrop;
  int pend_mask = 0;
  bool pend_live = false;
  constexpr int NST = NTW * MTW;             // stores per tile and lane, slot q = mt * NTW + nt
  constexpr int PER_TAP = NST > 9 ? 2 : 1;   // slots slipped in after each tap
  static_assert(TT == 0 || NST <= TT * PER_TAP, "every pending store must find a tap");
  auto store_slot = [&](auto qc) {
    constexpr int q = decltype(qc)::value;
    constexpr int mt = q / NTW, nt = q - mt * NTW;
    // (Round 4 measured the epilogue arithmetic moved here -- raw accumulators parked, descale / bias / activation / running
    //  maximum right in front of each store, between the taps of the next stage: same-box A/B 469 -> 504 us and 340 -> 348 us
    //  on the c2 layers, VDSR step 6.64 -> 6.84 ms.  The VALU work delays the wave's own next MFMAs more than the idle
    //  parking phase costs; not kept.)
    const f32x4 v = pend[nt][mt];
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), orsrc, (int)(pend_voff[mt] + 4u * (unsigned)coff[nt]), 0, 0);
  };
  auto flush_from = [&](auto q0c) {  // slots q0 .. NST-1
    constexpr int q0 = decltype(q0c)::value;
    srk_static_for<q0, NST>([&](auto qc) { store_slot(qc); });
  };
  // Parking (descale, bias, activation, output mask, running maximum: accumulators -> `pend`) takes ~1 k clocks in which
  // the wave issues no MFMA, and every consumer wave of the block did it at the same moment, at the end of a tile's last
  // stage: the matrix pipe idled through it (13 - 15 % of a consumer's stage).  With B.late the waves 4 - 7
  // -- each shares its SIMD with one of the waves 0 - 3 -- park a finished tile at the START of the next stage instead, from
  // the accumulators they are about to clear: one wave's parking runs under the other wave's taps at both ends of a
  // stage.  (Pays for single-chunk layers, where every stage ends a tile; the host sets it for those.)
  const bool late_wave = B.late && TT > 0 && NCW == 8 && (__builtin_amdgcn_readfirstlane(wave) & 4);
  bool park_due = false;
  int pk_n = 0, pk_r0 = 0, pk_c0 = 0;
  auto park = [&](int n, int r0, int c0) {
    // tile finished (C/D col = lane & 15 = pixel, rows kq*4 + reg = 4 consecutive channels per M tile): park it
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
  // everything this wave loaded so far (biases, activation slope) has landed BEFORE the loop: a first use inside it is
  // an s_waitcnt vmcnt(0) in every iteration -- the parking phase of every stage then waited for the write
  // acknowledgements of the stores issued under that stage's taps (found in the compiled code in round 4; k_conv_rowsw had
  // the fix, this kernel did not)
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) asm volatile("" ::"v"(bias4[nt]));
  asm volatile("" ::"v"(act_slope));
  __syncthreads();  // filter and stage 0 visible
  long long ct_taps = 0, ct_park = 0, ct_wait = 0;
  const long long ct_begin = BFW_CLK();
  for (int s = 0; s < S; ++s) {
    const long long k0 = BFW_CLK();
    int n, r0, c0, cc;
    decode(n, r0, c0, cc);
    if (park_due) {  // (late waves: the previous stage ended a tile; cc == 0 here)
      park(pk_n, pk_r0, pk_c0);
      park_due = false;
    }
    if (cc == 0) {
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (OMASK) {
      if (cc == B.ICc - 1 && wave_live) {
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
    if (wave_live && T > 0 && !(SRK_KDBG(B.dbg) & 4)) {
      const uint4* hal = hal0 + (size_t)(s & 1) * hbuf;
      const uint4* wb = wl + (size_t)cc * wslot;
      const size_t wstep = (size_t)B.ICc * wslot;
      uint4 fa[2][2][NTW], fb[2][2][MTW];  // [buffer][plane][tile]
      // tap walk in scalar registers (no LDS table: a table lookup would put a dependent LDS round trip and an
      // lgkmcnt(0) in front of every tap's fragment reads): halo offset u*HW + v, filter slot t
      int wt_toff = 0, wt_tv = 0, wt_t = 0;
      auto load_frags = [&](uint4 (&a)[2][NTW], uint4 (&b)[2][MTW]) {
        const uint4* hb = hal + wt_toff;
        const uint4* wt = wb + (size_t)wt_t * wstep;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
          b[0][mt] = hb[hp[mt]];
          b[1][mt] = hb[hp[mt] + plane];
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
      auto mfmas = [&](const uint4 (&a)[2][NTW], const uint4 (&b)[2][MTW]) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
          for (int mt = 0; mt < MTW; ++mt) acc[nt][mt] = mfma16x<F16>(a[0][nt], b[1][mt], acc[nt][mt]);  // w_h * x_m
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
          for (int mt = 0; mt < MTW; ++mt) acc[nt][mt] = mfma16x<F16>(a[1][nt], b[0][mt], acc[nt][mt]);  // w_m * x_h
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
          for (int mt = 0; mt < MTW; ++mt) acc[nt][mt] = mfma16x<F16>(a[0][nt], b[0][mt], acc[nt][mt]);  // w_h * x_h
      };
      // (sched_barrier: keep the next tap's ds_read_b128s IN FRONT of the current tap's MFMAs -- the machine
      //  scheduler otherwise sinks them behind ~16 MFMAs)
      load_frags(fa[0], fb[0]);
      if (TT > 0) {
        srk_static_for<0, (TT > 0 ? TT : 1)>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          if (t + 1 < TT) load_frags(fa[(t + 1) & 1], fb[(t + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
          mfmas(fa[t & 1], fb[t & 1]);
          __builtin_amdgcn_sched_barrier(0);
          if (pend_live) {
            srk_static_for<t * PER_TAP, ((t + 1) * PER_TAP < NST ? (t + 1) * PER_TAP : NST)>([&](auto qc) { store_slot(qc); });
          }
        });
        pend_live = false;
      } else {
        int t = 0;
        for (; t + 2 <= T; t += 2) {
          load_frags(fa[1], fb[1]);
          __builtin_amdgcn_sched_barrier(0);
          mfmas(fa[0], fb[0]);
          __builtin_amdgcn_sched_barrier(0);
          if (t + 2 < T) load_frags(fa[0], fb[0]);
          __builtin_amdgcn_sched_barrier(0);
          mfmas(fa[1], fb[1]);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (t < T) mfmas(fa[0], fb[0]);
      }
    }
    const long long k1 = BFW_CLK();
    if (cc == B.ICc - 1 && wave_live && !(SRK_KDBG(B.dbg) & 2)) {
      if (late_wave && s != S - 1) {  // parked at the top of the next stage, while the SIMD's other consumer wave runs its taps
        park_due = true;
        pk_n = n; pk_r0 = r0; pk_c0 = c0;
      } else {
        park(n, r0, c0);
        if (TT == 0 || s == S - 1 || (SRK_KDBG(B.dbg) & 1024)) {  // no unrolled tap loop to ride under / last tile of this block
          flush_from(std::integral_constant<int, 0>{});
          pend_live = false;
        }
      }
    }
    const long long k2 = BFW_CLK();
    __syncthreads();
    const long long k3 = BFW_CLK();
    ct_taps += k1 - k0;
    ct_park += k2 - k1;
    ct_wait += k3 - k2;
  }
#ifdef SRK_EXPERIMENTS
  if (B.prof && tid == 0) {
    long long* pr = B.prof + (size_t)blockIdx.x * 16 + 8;
    pr[0] = ct_taps; pr[1] = ct_park; pr[2] = ct_wait; pr[3] = S; pr[4] = BFW_CLK() - ct_begin;
  }
#endif
  (void)ct_taps; (void)ct_park; (void)ct_wait; (void)ct_begin;
  if (P.ep.y_amax) amax_commit(P.ep.y_amax, amax, blockIdx.x + wave, amax_peek(P.ep.y_amax, blockIdx.x + wave));  // (persistent: once)
}

// Applicability: stride-1 gathers (CONV, or TRANS = the data gradient of a stride-1 conv: flipped taps), IC a multiple
// of 8 (16-byte channel groups), OC = 16 * {1..4} or 32-channel slices of a wider 3x3 layer, filter planes (of a slice)
// + two halo buffers within the LDS, every output group on the 16-byte store path, and a problem large enough to keep
// one persistent block per CU busy for several tiles.  The activation-gradient mask (mask_y) is applied by the
// producers of the 32-channel 3x3 variant.  SRK_BFW: 0 never, 1 whenever applicable, unset = automatic.
bool conv_bfw_applicable(const GatherConv& g, const Epi& ep, const float* in, const float* out, const float* mask_y) {
  const char* e = env_str("SRK_BFW");
  const int mode = e ? atoi(e) : 2;
  if (mode == 0 || g.stride != 1 || g.in_nchw || g.in_ps_r > 1) return false;
  if (g.IC < 8 || g.IC % 8 != 0 || (uintptr_t)in % 16 != 0) return false;
  if (g.OC % 16 != 0 || g.OC < 16) return false;
  if ((long)g.IH * g.IW * g.IC * 4 >= (1L << 31)) return false;   // the producers' per-image buffer descriptor (32-bit byte offsets)
  const int T = g.KH * g.KW;
  const int nsl = bfw_slices(g);
  if (nsl == 0) return false;
  const int NB = g.OC / nsl;
  if (T > BFW_MAXTAPS - 1 || (T != 9 && NB > 48)) return false;  // (the 64-channel dynamic-tap variant spills)
  if (mask_y && (NB != 32 || T != 9 || (uintptr_t)mask_y % 16 != 0)) return false;
  if (ep.out_relu && (NB != 32 || T != 9 || ep.ps_r > 1 || ep.act != SRK_ACT_NONE || (uintptr_t)ep.out_relu % 16 != 0)) return false;
  if (!conv_epi_all_vector(g.OC, ep, out)) return false;
  if (ep.act == SRK_ACT_PRELU && ep.prelu_n > 1) return false;
  // a residual / gradient fan-in: only the ring kernel's canvas variant adds one (conv_bfr.hip: 3x3 pad-1 layers with 64 input
  // channels and 32-channel slices, no mask, plain stores); conv_bfw_gather returns -1 where that variant declines
  if (ep.residual && (T != 9 || NB != 32 || g.IC != 64 || mask_y || ep.out_relu || ep.ps_r > 1 || (uintptr_t)ep.residual % 16 != 0))
    return false;
  if (ep.act != SRK_ACT_NONE && ep.act != SRK_ACT_RELU && ep.act != SRK_ACT_LRELU && ep.act != SRK_ACT_PRELU) return false;
  if ((long)g.N * g.OH * g.OW * g.OC >= (1L << 29)) return false;  // 32-bit byte offsets into an output below 2 GiB (kDrop)
  const size_t wbytes = (size_t)T * ((g.IC + 31) / 32) * 8 * NB * 16;
  if (wbytes > 100 * 1024) return false;
  if ((long)g.N * g.OH * g.OW >= (1L << 30) || (long)g.IH * g.IW * g.IC >= (1L << 30)) return false;
  if (mode == 1) return true;
  // from two 256-pixel tiles per CU upwards (below that the small-problem blocks of k_conv_bfd take the layer); measured
  // on ESPCN x4 at 256x256: batch 4 40.6 k vs 34.5 k images/s, batch 8 48.1 k vs 38.8 k with the per-tile kernels
  return (long)g.N * g.OH * g.OW >= 256L * 2 * kNumCU;
}

template <int NTW, int TT, int MTW>
static int bfw_launch_t(const BfwParams& B, size_t lds, int grid, hipStream_t s) {
  note_amax_written(B.P.ep.y_amax != nullptr);
  if constexpr (NTW == 2 && TT == 9 && MTW == 2) {  // data gradients (bf16x3): mask on dy and / or ReLU gradient on dx
    const dim3 blk(64 * (16 / MTW + 4));
    if (B.P.mask_y && B.P.ep.out_relu) {
      static LdsLimit limb;
      limb.ensure(reinterpret_cast<const void*>(&k_conv_bfw<NTW, TT, MTW, false, true, true>), lds);
      note_kernel("k_conv_bfw<%d,%d,%d,mask,relu>", NTW, TT, MTW);
      hipLaunchKernelGGL((k_conv_bfw<NTW, TT, MTW, false, true, true>), dim3(grid), blk, lds, s, B);
      return check_launch("conv_bfw");
    }
    if (B.P.mask_y) {
      static LdsLimit limm;
      limm.ensure(reinterpret_cast<const void*>(&k_conv_bfw<NTW, TT, MTW, false, true, false>), lds);
      note_kernel("k_conv_bfw<%d,%d,%d,mask>", NTW, TT, MTW);
      hipLaunchKernelGGL((k_conv_bfw<NTW, TT, MTW, false, true, false>), dim3(grid), blk, lds, s, B);
      return check_launch("conv_bfw");
    }
    if (B.P.ep.out_relu) {
      static LdsLimit limo;
      limo.ensure(reinterpret_cast<const void*>(&k_conv_bfw<NTW, TT, MTW, false, false, true>), lds);
      note_kernel("k_conv_bfw<%d,%d,%d,relu>", NTW, TT, MTW);
      hipLaunchKernelGGL((k_conv_bfw<NTW, TT, MTW, false, false, true>), dim3(grid), blk, lds, s, B);
      return check_launch("conv_bfw");
    }
  }
  if (B.P.mask_y || B.P.ep.out_relu) return -1;
  if (B.w_descale) {  // f16x3 arithmetic
    static LdsLimit limh;
    limh.ensure(reinterpret_cast<const void*>(&k_conv_bfw<NTW, TT, MTW, true>), lds);
    note_kernel("k_conv_bfw<%d,%d,%d,f16>", NTW, TT, MTW);
    hipLaunchKernelGGL((k_conv_bfw<NTW, TT, MTW, true>), dim3(grid), dim3(64 * (16 / MTW + 4)), lds, s, B);
    return check_launch("conv_bfw");
  }
  static LdsLimit lim;
  lim.ensure(reinterpret_cast<const void*>(&k_conv_bfw<NTW, TT, MTW>), lds);
  note_kernel("k_conv_bfw<%d,%d,%d>", NTW, TT, MTW);
  hipLaunchKernelGGL((k_conv_bfw<NTW, TT, MTW>), dim3(grid), dim3(64 * (16 / MTW + 4)), lds, s, B);
  return check_launch("conv_bfw");
}
template <int NTW>
static int bfw_launch(const BfwParams& B, size_t lds, int grid, hipStream_t s) {
  // Consumer waves: 8 x 32 pixels (two per SIMD: LDS latency of one hides under the MFMAs of the other; measured
  // -1 % / -3.5 % on the c2 64->32 / 32->48 layers vs 4 x 64 pixels) while the kernel fits 168 VGPRs (<= 48 output
  // channels); 64 channels: 4 x 64 pixels.  3x3 kernels get the unrolled tap loop with deferred stores; with 64-pixel
  // consumer waves and > 32 channels that variant spills (256 VGPRs) and the dynamic loop is used.
  const int mtw_env = SRK_EXP_INT("SRK_BFW_MTW", 0);  // experiment: 2 or 4
  const bool t9 = B.P.KHv * B.P.KWv == 9;
  const int mtw = mtw_env ? mtw_env : (NTW <= 3 ? 2 : 4);
  if (mtw == 2 && NTW <= 3) {
    if (t9) return bfw_launch_t<NTW, 9, 2>(B, lds, grid, s);
    return bfw_launch_t<NTW, 0, 2>(B, lds, grid, s);
  }
  if (NTW <= 2 && t9) return bfw_launch_t<NTW, 9, 4>(B, lds, grid, s);
  return bfw_launch_t<NTW, 0, 4>(B, lds, grid, s);
}

// returns -1 when no tile fits (the caller falls back to the other kernels).  f16: the f16x3 arithmetic (ep.x_amax set)
int conv_bfw_gather(const GatherConv& g, const float* in, const float* wp, float* out, const Epi& ep,
                    const float* mask_y, float mask_slope, hipStream_t s, bool f16) {
  const size_t elems = (size_t)g.KH * g.KW * g.IC * g.OC;
  const char* prepared = reinterpret_cast<const char*>(wp) + bf3_prepared_offset(elems);
  const char* fsec = prepared + f16_section_offset(g.IC, g.OC, g.KH * g.KW);
  const uint4* wq = reinterpret_cast<const uint4*>(f16 ? fsec : prepared);
  const float* w_descale = f16 ? reinterpret_cast<const float*>(fsec + bf3_main_bytes(g.IC, g.OC, g.KH * g.KW)) : nullptr;
  const int dbg = SRK_EXP_INT("SRK_DBG", 0);
  const int nsl = bfw_slices(g);
  if (nsl == 0) return -1;
  return for_each_phase(g, in, wp, out, ep, mask_y, mask_slope, [&](const MfmaConvParams& P0) {
    BfwParams B{};
    B.P = P0;
    MfmaConvParams& P = B.P;
    B.wq = wq;
    B.w_descale = w_descale;
    B.nsl = nsl;
    B.NBfull = P.OC >= 64 ? 64 : P.OC;   // (pack_items.h: pk_nb; OC is a multiple of 16 here)
    B.OCb = (P.OC + 63) / 64;
    B.NB = P.OC / nsl;
    B.ICc = (P.IC + 31) / 32;
    B.dbg = dbg;
#ifdef SRK_EXPERIMENTS
    B.prof = g_bfw_prof;
#endif
    const int T = P.KHv * P.KWv;
    const size_t wbytes = (size_t)T * B.ICc * 8 * B.NB * 16;
    const long lds_cap = 160L * 1024 - 512;
    long px_cap = (lds_cap - (long)wbytes) / (2 * 128);  // halo pixels per buffer (128 bytes each)
    if (px_cap > 64 * BFW_IT) px_cap = 64 * BFW_IT;
    const int perm = SRK_EXP_INT("SRK_BFW_PERM", 1);
    const int w16 = SRK_EXP_INT("SRK_BFW_W16", 0);
    B.perm = perm;
    B.late = B.ICc == 1 && env_int("SRK_BFW_LATE", 1) != 0;   // (B.ICc is set above)
    if (T == 9 && P.is == 1) {  // ring of halo buffers instead of the per-stage barrier (conv_bfr.hip) where it applies
      const int rc = conv_bfr_launch(B, s);
      if (rc != -1) return rc;
    }
    if (P.ep.residual) return -1;   // (this kernel's epilogue has no residual: the caller's other kernels)
    TilePick best{};
    if (px_cap < 64 || P.is != 1) return -1;
    if (!bfw_pick_tile(256, P.PH, P.PW, P.KHv, P.KWv, px_cap, perm, w16 ? 16 : 1, best)) return -1;
    P.TH = best.TH; P.TW = best.TW; P.tiles_y = best.tiles_y; P.tiles_x = best.tiles_x; P.HH = best.HH; P.HW = best.HW;
    B.NPIXp = bfw_group_stride(best.HH * best.HW, perm);
    const size_t lds = wbytes + (size_t)2 * 8 * B.NPIXp * 16;
    const long ntiles = (long)P.tiles_x * P.tiles_y * P.N;
    if (ntiles >= (1L << 30)) return -1;
    B.ntiles = (int)ntiles;
    B.out_bytes = (unsigned)((size_t)P.N * P.OH * P.OW * P.OC * sizeof(float));
    int grid = kNumCU;
    if (nsl > 1) {  // whole groups of nsl neighbouring blocks on every XCD (a group without tiles just exits)
      grid -= grid % (8 * nsl);
      const long want = ((ntiles + 7) / 8) * 8 * nsl;
      if (grid == 0) return -1;
      if (want < grid) grid = (int)want;
    } else if (grid > ntiles) {
      grid = (int)ntiles;
    }
    if (dbg & 32)
      fprintf(stderr, "[srk] k_conv_bfw<%d>: lds %zu B (filter %zu), grid %d of %ld tiles x %d slices, tile %dx%d halo %dx%d\n",
              B.NB / 16, lds, wbytes, grid, ntiles, nsl, P.TH, P.TW, P.HH, P.HW);
    switch (B.NB / 16) {
      case 1: return bfw_launch<1>(B, lds, grid, s);
      case 2: return bfw_launch<2>(B, lds, grid, s);
      case 3: return bfw_launch<3>(B, lds, grid, s);
      default: return bfw_launch<4>(B, lds, grid, s);
    }
  });
}

}  // namespace srk

#ifdef SRK_EXPERIMENTS
// experiments build only: device buffer of 16 int64 per block for the role-time sums of the next k_conv_bfw launches
extern "C" void srk_debug_bfw_prof(void* p) { srk::g_bfw_prof = static_cast<long long*>(p); }
#endif
