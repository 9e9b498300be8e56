// Ring form of the wave-specialised persistent 3x3 convolution (k_conv_bfw, conv_bfw.hip) for the large layers whose
// filter stays in LDS (c2: ESPCN 64 -> 32 and 32 -> 48 + pixel shuffle).  Same roles and products (f16x3 / bf16x3:
// x = h + m, w_h x_m + w_m x_h + w_h x_h, fp32 accumulate, chunks in order), two structural changes:
//
// 1. Half the LDS bytes per MFMA.  Round 5's constant-ablation builds of the barrier kernel's tap loop (tools/
//    ring_ablate.sh) on the 64 -> 32 layer: its fragment reads alone run 207 us, its MFMAs alone 261 us, both
//    together 303 - 326 us with no global memory traffic at all -- a 32-pixel x 32-channel wave tile reads 8 fragments
//    per 12 MFMAs (0.67 KB per MFMA: 2/3 of the LDS's peak at full matrix rate, and the LDS reaches ~2/3 of its peak), so
//    the consumers were LDS-bound before the first byte left HBM.  Here a consumer wave owns 4 ROWS x 16 columns of an
//    8 x 16 tile and walks (kernel column v, halo row R): the pixel fragment of halo row R, columns v .. v + 15 serves
//    every (output row r, kernel row u) with r + u = R -- up to three MFMA groups per fragment -- and the filter
//    fragments of one kernel column (u = 0 .. 2) stay in registers while the six halo rows pass, each reloaded for the
//    next column right behind its last use: 72 fragment reads per 216 MFMAs (0.33 KB per MFMA).  The tile is fixed, so
//    every LDS address is lane base + immediate.  Accumulation order per output: chunk, then kernel column, then kernel
//    row (k_conv_bfw: row, then column) -- equal to the barrier kernel up to fp32 summation order, not bit-equal.
//
// 2. No workgroup barrier per stage.  4 consumer waves (one per SIMD, beside one producer wave each: matrix beside
//    memory) form two groups of two that take alternate tiles of the block's list; the halo buffers are a ring of
//    NBUF >= 3 slots filled in the fixed global stage order (pair of tiles, chunk, group): stage s lives in slot s % NBUF.
//    Per slot two monotonic counters in LDS: full[slot] += 1 per producer wave once its ds_writes have landed,
//    free[slot] += 1 per consumer wave once its fragment reads of the slot have returned; use k of a slot may be filled
//    when free >= 2 k and read when full >= 4 (k + 1).  A wave waits for exactly the data it needs (ds_read_b32 +
//    s_sleep polling, every poll capped: a protocol error ends as wrong numbers and a non-zero srk_ring_timeouts(), not as
//    a hung GPU).  Producers keep NSET register sets of loads in flight and run ahead as far as the ring allows; every
//    load is issued unconditionally (stages past the end read through an out-of-range offset), so the compiler counts a
//    commit's wait exactly (vmcnt = the loads of the other sets).
//    (The ring alone, on the barrier kernel's 32-pixel consumers, changed nothing: 472 vs 465 us and 332 vs 330 us on the
//    two c2 layers -- the lockstep was not what that kernel waited for)
#include "conv_bfw.h"

namespace srk {

// Ablation builds (tools/ring_ablate.sh; release: 0): 1 no global loads, 2 no stores, 4 no MFMAs, 8 no fragment reads,
// 16 no producer split + LDS commit, 32 no parking arithmetic
#ifndef BFR_ABL
#define BFR_ABL 0
#endif
#ifndef BFR_NSET
#define BFR_NSET 3
#endif
constexpr int BFR_MAXBUF = 6;
constexpr int BFR_CNT_BYTES = 128;  // counters behind the ring: full[6], free[6]; fused: scale exponents, staging / compute counts, tile maxima
constexpr int BFR_PIT = 3;  // producer register batches per set: the 180-pixel halo in 64-pixel batches
constexpr int BFR_NSET_C = BFR_NSET;  // register sets = stages of loads in flight per producer wave
constexpr unsigned BFR_SPIN_CAP = 1u << 18;
constexpr int BFR_TH = 8, BFR_TW = 16, BFR_HH = 10, BFR_HW = 18, BFR_NPIX = 180;
constexpr int BFR_NPIXP = 190;  // = bfw_group_stride(180, perm): +-2 (mod 16), conflict-free fragment reads and halo writes

__device__ unsigned g_bfr_timeouts = 0;

// Profiling builds (-DBFR_PROF, tools/ring_prof.py): s_memtime sums per role -- 16 int64 per block:
// [0..5] first producer wave {wait free, wait loads, split + commit, signal, issue, total}, [8..13] consumer wave 0
// {wait full, steps, signal, park, total, stages}
// (-DBFR_PROF=2: only the loop totals -- two stamps per wave, the stream itself is the release one)
#ifdef BFR_PROF
static long long* g_bfr_prof = nullptr;
#define BFR_CLK() (BFR_PROF == 2 ? 0ll : clock64())
#define BFR_CLK_TOTAL() clock64()
#else
#define BFR_CLK() 0ll
#define BFR_CLK_TOTAL() 0ll
#endif

typedef __attribute__((address_space(3))) unsigned bfr_cnt_t;

// A poll that ran into its cap means the ring protocol slipped (or the wave was starved beyond anything a profiler does):
// whatever this tile holds is wrong.  Count it (srk_ring_timeouts, diagnostics) and ABORT the dispatch: the stream then
// reports hipErrorLaunchFailure at its next synchronisation / srk_* call instead of SRK_OK with corrupt output.
__device__ __noinline__ void bfr_fail(int lane) {
  if (lane == 0) {
    atomicAdd(&g_bfr_timeouts, 1u);
    __threadfence_system();
  }
  __builtin_trap();
}

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
      __builtin_amdgcn_s_sleep(1);
      if (++spins > BFR_SPIN_CAP) {
        dead = true;
        break;
      }
    }
  }
  asm volatile("" ::: "memory");
}
// One count behind every LDS access this wave has issued so far.  No s_waitcnt: the LDS executes the operations of one wave
// in the order they were issued, so whoever sees the count sees the writes (or finds the reads done) that precede it.
// (BFR_SIGNAL_WAIT=1 builds drain the wave's LDS queue first.)
#ifndef BFR_SIGNAL_WAIT
#define BFR_SIGNAL_WAIT 0
#endif
__device__ __forceinline__ void bfr_signal(bfr_cnt_t* p) {
  if constexpr (BFR_SIGNAL_WAIT != 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else asm volatile("" ::: "memory");
  if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
}

// (Round 5 carried a FUSE variant here -- the Cin <= 4 first layer computed by the producers straight into the ring,
//  srk_conv2d_pair_forward.  Correct but slower than the two launches (0.93 vs 0.70 ms on the c2 pair: its producers carried
//  ~930 instructions per tile against the consumers' ~365); retired in round 6, DESIGN 13.  The last tree with it: b92a5aa.)
//
// CV ("canvas", round 6): the layers of VDSR -- 64 -> 64 on 41 x 41 patches -- as this kernel's fixed 8 x 16 tiles.  41 is
// 2.6 tiles wide (8 x 16 tiles over one patch: 73 % full), so the tiles are laid over a CANVAS instead: the batch as a grid
// of cells of (PH + 1) x (PW + 1) pixels, B.cv_kx patches side by side, each followed by one separator row and column that
// reads as zero -- the 3x3 "same" convolution's own padding, shared by neighbouring patches -- and whose outputs are
// dropped.  The host picks cv_kx for the fullest tiles (256 patches of 41 x 41: 8 across, 95 %).  Producers map a halo
// pixel to (patch, row, column) or to an out-of-range offset, consumers map an output pixel the same way; nothing else
// changes.  CV blocks also take output-channel SLICES (B.nsl, as k_conv_bfw does: the 147 KB filter of 64 -> 64 as two
// 32-channel halves on neighbouring blocks of one XCD).  OMASK: ep.out_relu as in k_conv_bfw (the data gradient leaves
// multiplied by the ReLU gradient of the layer below), requested when the tile starts, applied when it is parked.
// RES: ep.residual (conv + skip of a residual block, `/root/reference/base_networks.py:148`; the gradient fan-in of its first
// conv's data gradient) is requested the same way and added after the activation, the epilogue order of every other kernel.
template <int NTW, int ICC, bool F16, bool CV = false, bool OMASK = false, bool RES = false>
__global__ __launch_bounds__(512, 2) void k_conv_bfr(BfwParams B) {
  static_assert(CV || !(OMASK || RES), "the output mask and the residual come with the canvas variant");
  static_assert(!(OMASK && RES), "one tensor read per output tile");
  constexpr int NB = 16 * NTW;
  constexpr int NCW = 4, NGW = 2, NPW = 4;  // consumer waves (two groups of NGW), producer waves
  constexpr int NPS = NPW;                  // producer waves that fill one slot
  constexpr int NTHR = 64 * (NCW + NPW);
  constexpr int PSTEP = 16 * NPW, PIT = BFR_PIT, NSET = BFR_NSET_C;
  constexpr int TH = BFR_TH, TW = BFR_TW, HW = BFR_HW, NPIX = BFR_NPIX, NPIXP = BFR_NPIXP;
  constexpr int MR = TH / NGW;          // output rows per consumer wave (4)
  constexpr int WSLOT = 8 * NB;         // uint4 per (tap, chunk): [plane 2][group 4][NB]
  constexpr int HBUF = 8 * NPIXP;       // uint4 per halo slot: [plane 2][group 4][NPIXP]
  constexpr int PLANE_B = 4 * NPIXP, PLANE_A = 4 * NB;
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
#ifdef BFR_PROF
  const long long cw_entry = wall_clock64();
#endif
  const MfmaConvParams& P = B.P;
  uint4* wl = smem4;
  uint4* hal0 = smem4 + 9 * ICC * WSLOT;
  const int nbuf = __builtin_amdgcn_readfirstlane(B.nbuf);
  bfr_cnt_t* cnt = (bfr_cnt_t*)(hal0 + (size_t)nbuf * HBUF);  // full[BFR_MAXBUF], free[BFR_MAXBUF]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave >= NCW;
  const int j = lane & 15, kq = lane >> 4;
  float sx = 1.f, dsc = 1.f;
  if constexpr (F16) {
    const int kx = amax_scale_exp(amax_read(P.ep.x_amax));
    sx = exp2i(kx);
    dsc = exp2i(-kx) * B.w_descale[0];
  }
  (void)sx;

#ifdef BFR_PROF
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const long long cw_p1 = wall_clock64();   // arguments, input maximum, descale read
#endif
  const int xcd = blockIdx.x & 7;
  int bi = blockIdx.x >> 3, sl = 0;  // (not CV: one slice, the whole filter is resident)
  if constexpr (CV) {
    sl = bi % B.nsl;
    bi = bi / B.nsl;
  }
  // The filter into LDS: every load of a thread in flight before its first LDS store.  (As a plain loop the compiler waits for
  // each 16-byte load in turn -- global_load; s_waitcnt vmcnt(0); ds_write -- nine L2 round trips in a row, 4.1 - 4.7 us of the
  // 6.5 - 8.5 us between a block's first instruction and its first tile: tools/ring_prof.py, round 6.  Native vectors: an
  // array of HIP's uint4 structs ends up in scratch memory here.)
  {
    constexpr int FW = 9 * ICC * WSLOT, NLD = (FW + NTHR - 1) / NTHR;
    typedef unsigned fv4 __attribute__((ext_vector_type(4)));
    fv4 wv[NLD];
    srk_static_for<0, NLD>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int e = tid + i * NTHR;
      const int ec = e < FW ? e : FW - 1;
      const int slot = ec / WSLOT, w = ec - slot * WSLOT;
      const int t = slot / ICC, cc = slot - t * ICC;
      const int u = t / 3, v = t - u * 3;
      const int tapw = (P.wh0 + P.wdh * u) * P.KW_full + (P.ww0 + P.wdw * v);
      // packed layout [tap][chunk][64-channel block][plane][group][NBfull channels] with NBfull = NB where OC <= 48
      if constexpr (CV) {
        const int pg = w / NB, o = w - pg * NB;
        const int oc = sl * NB + o, ocb = oc / B.NBfull;
        wv[i] = *reinterpret_cast<const fv4*>(B.wq + ((size_t)(tapw * ICC + cc) * B.OCb + ocb) * (size_t)(8 * B.NBfull) + pg * B.NBfull +
                                              (oc - ocb * B.NBfull));
      } else {
        wv[i] = *reinterpret_cast<const fv4*>(B.wq + (size_t)(tapw * ICC + cc) * WSLOT + w);
      }
    });
    srk_static_for<0, NLD>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int e = tid + i * NTHR;
      if (e < FW) *reinterpret_cast<fv4*>(wl + e) = wv[i];
    });
  }
  if (tid < 2 * BFR_MAXBUF) cnt[tid] = 0u;
#ifdef BFR_PROF
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const long long cw_p2 = wall_clock64();   // filter in LDS
#endif
  // tiles of this block (XCD-aware contiguous ranges, as in k_conv_bfw): first, first + tstride, ... (count of them)
  const int nblk = gridDim.x;
  int first, count;
  {
    const int per_x = B.ntiles >> 3, rem_x = B.ntiles & 7;
    const int nb_x = CV ? ((nblk + 7 - xcd) >> 3) / B.nsl : (nblk + 7 - xcd) >> 3;
    const int tiles_x = per_x + (xcd < rem_x ? 1 : 0);
    const int start_x = xcd * per_x + (xcd < rem_x ? xcd : rem_x);
    first = start_x + bi;
    count = bi < tiles_x ? (tiles_x - bi + nb_x - 1) / nb_x : 0;
  }
  const int tstride = CV ? ((nblk + 7 - xcd) >> 3) / B.nsl : (nblk + 7 - xcd) >> 3;
  // CV: the tile coordinates below are those of the canvas (one "image" of tiles_y x tiles_x tiles)
  const int cvS = CV ? B.cv_sep : 1;
  const int cvH = P.PH + cvS, cvW = P.PW + cvS, cvK = CV ? B.cv_kx : 1;
  // v / cvH and v / cvW for the small non-negative v below: one multiply-high by ceil(2^32 / d) (host: exact for v d < 2^32)
  auto div_h = [&](int v) { return (int)__umulhi((unsigned)v, B.cv_mh); };
  auto div_w = [&](int v) { return (int)__umulhi((unsigned)v, B.cv_mw); };
  count = __builtin_amdgcn_readfirstlane(count);

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
  const int prio = __builtin_amdgcn_readfirstlane(B.late);  // 1: producers, 2: consumers issue first on their SIMD

  if (producer) {
    // ------------------------------------------------------------------ producers
    if (prio == 1) __builtin_amdgcn_s_setprio(1);
    const int ptid = tid - 64 * NCW;
    const int g = ptid & 3, hp0 = ptid >> 2;
    f32x4 pv[NSET][PIT][2];
    int it_rel[PIT], it_hx[PIT], it_hy[PIT];
#pragma unroll
    for (int k = 0; k < PIT; ++k) {
      const int hq = hp0 + PSTEP * k;
      const int hy = hq / HW, hx = hq - hy * HW;
      it_rel[k] = hq < NPIX ? ((hy * P.IW + hx) * P.IC + g * 8) * 4 : -1;
      it_hx[k] = hx;
      it_hy[k] = hy;
    }
    // CV: byte strides of an input row / pixel / patch, and what a step into the next cell adds to an offset
    const int cv_row = P.IW * P.IC * 4, cv_px = P.IC * 4, cv_img = P.IH * cv_row;
    const int cv_wrap_y = cvK * cv_img - cvH * cv_row, cv_wrap_x = cv_img - cvW * cv_px;
    int cv_rel[PIT];
#pragma unroll
    for (int k = 0; k < PIT; ++k) cv_rel[k] = it_hy[k] * cv_row + it_hx[k] * cv_px + g * 32;
    const unsigned img_bytes = (unsigned)((size_t)P.IH * P.IW * P.IC * 4);
    const unsigned in_bytes = (unsigned)((size_t)P.N * P.IH * P.IW * P.IC * 4);  // (CV; host: below 2 GiB)
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
    auto issue = [&](f32x4 (&v)[PIT][2]) {
      const bool valid = left > 0;
      const bool second = wg == 1;
      const int n = valid ? (second ? b_n : a_n) : 0;
      const int r0 = (second ? b_y : a_y) * TH, c0 = (second ? b_x : a_x) * TW;
      const int cc = wcc;
      if (valid) {
        const int ng = left >= 2 ? 2 : 1;
        if (++wg == ng) {
          wg = 0;
          if (++wcc == ICC) {
            wcc = 0;
            left -= ng;
            adv2(a_n, a_y, a_x);
            adv2(b_n, b_y, b_x);
          }
        }
      }
      constexpr unsigned OOB = 0x80000000u;
      const int iyb = r0 + P.iy0, ixb = c0 + P.ix0;
      const bool ch_on = valid && cc * 32 + g * 8 + 7 < P.IC;
      const size_t img = CV ? 0 : (size_t)n * P.IH * P.IW * P.IC;
      const __amdgpu_buffer_rsrc_t rin = rsrc_of(P.in + img, CV ? in_bytes : img_bytes);
      const int obase = ((iyb * P.IW + ixb) * P.IC + cc * 32) * 4;  // may be negative: rows above the image wrap out of range
      // CV: cell (cy0, cx0) and in-cell position (ry0, rx0) of the halo's first row / column -- cell -1 with the position of
      // the separator for the row above / the column left of the canvas.  A halo pixel (hy, hx) lies in that cell or, from
      // hy >= ty / hx >= tx on, in the next one; hy == ty - 1 / hx == tx - 1 is the separator; one cell past the last of a
      // canvas row is padding too.  Patches below the batch's last lie beyond the buffer: the descriptor returns zeros.
      // Without separators (cvS == 0: patches of whole tiles) the tile lies inside ONE cell and a halo pixel of another cell is
      // padding: valid where "one cell on" equals "the halo's first row / column lies one cell before the tile's".
      int ty = 0, tx = 0, cv_base = 0;
      bool x_last = false, y_prev = false, x_prev = false;
      if constexpr (CV) {
        const int cy0 = div_h(iyb + cvH) - 1, ry0 = iyb + cvH - (cy0 + 1) * cvH;
        const int cx0 = div_w(ixb + cvW) - 1, rx0 = ixb + cvW - (cx0 + 1) * cvW;
        ty = cvH - ry0;
        tx = cvW - rx0;
        x_last = cx0 == cvK - 1;
        y_prev = ty == 1;   // (the halo's first row is the last row of the cell above the tile's)
        x_prev = tx == 1;
        cv_base = (cy0 * cvK + cx0) * cv_img + ry0 * cv_row + rx0 * cv_px + cc * 128;
      }
#pragma unroll
      for (int k = 0; k < PIT; ++k) {
        bool ok;
        unsigned o;
        if constexpr (CV) {
          const bool wy = it_hy[k] >= ty, wx = it_hx[k] >= tx;
          ok = it_rel[k] >= 0 && ch_on &&
               (cvS ? it_hy[k] != ty - 1 && it_hx[k] != tx - 1 && !(wx && x_last) : wy == y_prev && wx == x_prev);
          o = ok ? (unsigned)(cv_base + cv_rel[k] + (wy ? cv_wrap_y : 0) + (wx ? cv_wrap_x : 0)) : OOB;
        } else {
          ok = it_rel[k] >= 0 && ch_on && (unsigned)(ixb + it_hx[k]) < (unsigned)P.IW;
          o = ok ? (unsigned)(obase + it_rel[k]) : OOB;
        }
        if constexpr (BFR_ABL & 1) {
          v[k][0] = v[k][1] = (f32x4){(float)o, 1.f, 2.f, 3.f};
        } else {
          v[k][0] = bload(rin, o);
          v[k][1] = bload(rin, o + 16u);
        }
      }
    };
    auto commit = [&](const f32x4 (&v)[PIT][2], uint4* hal) {
      if constexpr (BFR_ABL & 16) {
#pragma unroll
        for (int k = 0; k < PIT; ++k) asm volatile("" ::"v"(v[k][0]), "v"(v[k][1]));
        return;
      }
#pragma unroll
      for (int k = 0; k < PIT; ++k) {
        const int hq = hp0 + PSTEP * k;
        if (hq < NPIX) {
          float f[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f[e] = v[k][0][e];
            f[4 + e] = v[k][1][e];
          }
          uint4 pl[2];
          if constexpr (F16) split8h(f, sx, pl); else split8n<2>(f, pl);
          hal[(0 * 4 + g) * NPIXP + hq] = pl[0];
          hal[(1 * 4 + g) * NPIXP + hq] = pl[1];
        }
      }
    };
    const int S = count * ICC;
    srk_static_for<0, NSET>([&](auto ic) { issue(pv[decltype(ic)::value]); });
    __syncthreads();  // filter and counters visible
    int b = 0;
    unsigned k = 0;  // slot and use count of the stage about to be committed
    long long pt[5] = {0, 0, 0, 0, 0};
    const long long pt_begin = BFR_CLK_TOTAL();
    for (int s = 0; s < S; s += NSET) {
      srk_static_for<0, NSET>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const long long c0 = BFR_CLK();
        if (s + i < S) {
          bfr_wait(cnt + BFR_MAXBUF + b, NGW * k, dead);
          const long long c1 = BFR_CLK();
#ifdef BFR_PROF
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIT * (NSET - 1)) : "memory");
#endif
          const long long c2 = BFR_CLK();
          commit(pv[i], hal0 + (size_t)b * HBUF);
          const long long c3 = BFR_CLK();
          bfr_signal(cnt + b);
          const long long c4 = BFR_CLK();
          pt[0] += c1 - c0; pt[1] += c2 - c1; pt[2] += c3 - c2; pt[3] += c4 - c3;
          if (++b == nbuf) {
            b = 0;
            ++k;
          }
        }
        const long long c5 = BFR_CLK();
        issue(pv[i]);  // stage s + i + NSET (unconditional: the waits of the next commits stay countable)
        pt[4] += BFR_CLK() - c5;
      });
    }
#ifdef BFR_PROF
    if (B.prof && tid == 64 * NCW) {
      long long* pr = B.prof + (size_t)blockIdx.x * 16;
      for (int i = 0; i < 5; ++i) pr[i] = pt[i];
      pr[5] = BFR_CLK_TOTAL() - pt_begin;
    }
#endif
    (void)pt; (void)pt_begin;
    if (dead) bfr_fail(lane);
    return;
  }

  // -------------------------------------------------------------------- consumers
  // One consumer wave per SIMD: every cycle in which this wave has no MFMA to issue is a cycle its matrix pipe idles (a
  // lone MFMA wave reaches 2.25 - 2.4 PF at any accumulator distance, beside a VALU / LDS / VMEM partner wave too:
  // tools/micro/mfma_chain.hip, coissue.hip).  So the stream below has no stage boundaries: fragments are requested two
  // steps ahead ACROSS stages (the next stage's filter fragments are in LDS anyway, its first pixel fragments are
  // requested once its slot is known to be full: counter peeked at step 8, checked at step 16), the slot is handed back
  // at step 15 right behind its last read, the finished tile's stores go out unconditionally one per `SPER` steps of the
  // next tile, and parking is a handful of packed operations per accumulator.
  if (prio == 2) __builtin_amdgcn_s_setprio(1);
  const int grp = __builtin_amdgcn_readfirstlane(wave >> 1), gw = wave & 1;
  // lane column -> pixel column of the 16-pixel M tile: lanes {0-3, 12-15} hold the even columns, {4-11} the odd ones
  // (bfw_group_stride: with a group stride of +-2 (mod 16) every ds_read_b128 lane set covers 16 distinct slots)
  const int pj = j < 4 ? 2 * j : (j < 12 ? 2 * j - 7 : 2 * j - 16);
  const int lane_b = (MR * gw) * HW + pj + kq * NPIXP;  // pixel fragment (halo row R, column shift v): + R * HW + v
  const int lane_a = kq * NB + j;                        // filter fragment of tap t, tile nt: + t * ICC * WSLOT + nt * 16
  f32x4 bias4[NTW];
  int coff[NTW], poff[MR];
  {
    const EpiTile e0 = epi_tile_setup(P, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < MR; ++r) poff[r] = (MR * gw + r) * e0.RS + pj * e0.CS;
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
  const int act_kind = __builtin_amdgcn_readfirstlane(P.ep.act == SRK_ACT_NONE ? 0 : (P.ep.act == SRK_ACT_RELU ? 1 : 2));
  const bool want_amax = __builtin_amdgcn_readfirstlane(P.ep.y_amax != nullptr);
  f32x4 acc[NTW][MR];
  f32x4 pend[NTW][MR];  // the finished tile, stored under the next tile's MFMAs
  float amax = 0.f;
  const float dsc_t = dsc;
  constexpr unsigned kDrop = 0x80000000u;
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(P.out, 0, B.out_bytes, 0x00020000);
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  unsigned pend_voff[MR];  // (kDrop: the buffer unit drops the store -- before the first tile, and pixels beyond the image)
#pragma unroll
  for (int r = 0; r < MR; ++r) pend_voff[r] = kDrop;
  // CV: byte offsets of the lane's output pixels of the tile being computed (kDrop: separator / beyond the batch), set when
  // the tile starts; OMASK: the mask tensor's values there, requested at that moment through the same offsets
  unsigned cur_voff[CV ? MR : 1];
  f32x4 om[(OMASK || RES) ? NTW : 1][(OMASK || RES) ? MR : 1];   // (RES: the residual's values)
  auto cv_offsets = [&](int r0, int c0) {
    if constexpr (CV) {
      const int cy0 = div_h(r0), ry0 = r0 - cy0 * cvH;
      const int cx0 = div_w(c0), rx0 = c0 - cx0 * cvW;
      int rx = rx0 + pj, cx = cx0;
      if (rx >= cvW) {
        rx -= cvW;
        ++cx;
      }
      const bool col_ok = rx < P.PW && cx < cvK;
#pragma unroll
      for (int r = 0; r < MR; ++r) {
        int ry = ry0 + MR * gw + r, cy = cy0;
        if (ry >= cvH) {
          ry -= cvH;
          ++cy;
        }
        const int pn = cy * cvK + cx;
        const bool pok = col_ok && ry < P.PH && pn < P.N;
        cur_voff[r] = pok ? 4u * (unsigned)(((pn * P.OH + ry) * P.OW + rx) * P.OC) : kDrop;
      }
      if constexpr (OMASK || RES) {
        const __amdgpu_buffer_rsrc_t mrsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(RES ? P.ep.residual : P.ep.out_relu), 0, B.out_bytes, 0x00020000);
#pragma unroll
        for (int r = 0; r < MR; ++r)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt)
            om[nt][r] = __builtin_bit_cast(
                f32x4, __builtin_amdgcn_raw_buffer_load_b128(mrsrc, (int)(cur_voff[r] + 4u * (unsigned)coff[nt]), 0, 0));
      }
    }
  };
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int r = 0; r < MR; ++r) pend[nt][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int NST = NTW * MR;               // stores per tile and lane, slot q = r * NTW + nt
  constexpr int SPER = (18 * ICC) / NST;      // one store every SPER steps of the next tile
  static_assert(SPER >= 1, "every pending store must find a step");
  auto store_slot = [&](auto qc) {
    constexpr int q = decltype(qc)::value;
    constexpr int r = q / NTW, nt = q - r * NTW;
    const f32x4 v = pend[nt][r];
    if constexpr (BFR_ABL & 2) {
      asm volatile("" ::"v"(v));
      return;
    }
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), orsrc, (int)(pend_voff[r] + 4u * (unsigned)coff[nt]), 0, 0);
  };
  // accumulators -> pend: v = act(acc * 2^-k + bias).  One packed fma per two values, the activation by kind (ReLU = one
  // max; none = nothing; leaky / PReLU = max + min + fma), the running maximum as max3 with |.| modifiers
  auto park_kind = [&](auto kind_c, int n, int r0, int c0) {
    constexpr int KIND = decltype(kind_c)::value;
    const unsigned tile_off = CV ? 0u : 4u * (unsigned)epi_tile_setup(P, n, r0, c0).off0;
    const bool col_ok = c0 + pj < P.PW;
#pragma unroll
    for (int r = 0; r < MR; ++r) {
      bool pok;
      if constexpr (CV) {
        pok = cur_voff[r] != kDrop;
        pend_voff[r] = cur_voff[r];
      } else {
        pok = col_ok && r0 + MR * gw + r < P.PH;
        pend_voff[r] = pok ? tile_off + 4u * (unsigned)poff[r] : kDrop;
      }
      float rmax = 0.f;
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        f32x4 v = acc[nt][r];
        if constexpr (!(BFR_ABL & 32)) {
          if constexpr (F16) v = __builtin_elementwise_fma(v, (f32x4){dsc_t, dsc_t, dsc_t, dsc_t}, bias4[nt]); else v += bias4[nt];
          if constexpr (KIND == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          } else if constexpr (KIND == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(act_slope, fminf(v[e], 0.f), fmaxf(v[e], 0.f));
          }
          if constexpr (OMASK) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = om[nt][r][e] > 0.f ? v[e] : 0.f;
          }
          if constexpr (RES) v += om[nt][r];
          rmax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), rmax);
          rmax = fmaxf(fmaxf(fabsf(v[2]), fabsf(v[3])), rmax);
        }
        pend[nt][r] = v;
      }
      if (want_amax) amax = pok ? fmaxf(amax, rmax) : amax;
    }
  };
  auto park = [&](int n, int r0, int c0) {
    if (act_kind == 1) park_kind(std::integral_constant<int, 1>{}, n, r0, c0);
    else if (act_kind == 0) park_kind(std::integral_constant<int, 0>{}, n, r0, c0);
    else park_kind(std::integral_constant<int, 2>{}, n, r0, c0);
  };
  // (everything this wave loaded so far has landed before the loop: see k_conv_bfw)
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) asm volatile("" ::"v"(bias4[nt]));
  asm volatile("" ::"v"(act_slope));
#ifdef BFR_PROF
  const long long cw_p3 = wall_clock64();   // consumer set-up (bias, offsets) done, before the barrier
#endif
  __syncthreads();  // filter and counters visible

  // own tiles: list entries grp, grp + 2, ...; stage (entry i, chunk cc) is number (i >> 1) * 2 ICC + cc * ng + (i & 1) of
  // the global order, ng = tiles of the pair
  int o_n = grp ? b_n : a_n, o_y = grp ? b_y : a_y, o_x = grp ? b_x : a_x;
  long long ct[6] = {0, 0, 0, 0, 0, 0};
  const long long ct_begin = BFR_CLK_TOTAL();
#ifdef BFR_PROF
  const long long cw_begin = wall_clock64();
#endif
  int b = grp;     // ring slot of the current own stage (nbuf >= 3 > grp)
  unsigned k = 0;  // ... and how often that slot was used before
  uint4 fa[3][2][NTW];  // filter fragments of the current kernel column: [kernel row u][plane][tile]
  uint4 fb[3][2];       // pixel fragments of three consecutive steps: [step % 3][plane]
  const uint4* hb = hal0 + (size_t)b * HBUF + lane_b;
  auto ldA = [&](auto uc, auto vc, auto ccc) {
    constexpr int u = decltype(uc)::value, v = decltype(vc)::value, cc = decltype(ccc)::value;
    const uint4* wb = wl + lane_a;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      if constexpr (BFR_ABL & 8) {
        bfr_touch(fa[u][0][nt]);
        bfr_touch(fa[u][1][nt]);
      } else {
        fa[u][0][nt] = wb[((u * 3 + v) * ICC + cc) * WSLOT + nt * 16];
        fa[u][1][nt] = wb[((u * 3 + v) * ICC + cc) * WSLOT + PLANE_A + nt * 16];
      }
    }
  };
  auto ldB = [&](const uint4* base, auto sc) {  // step s = v * 6 + R of the stage whose slot `base` points into
    constexpr int s = decltype(sc)::value, v = s / 6, R = s - 6 * v;
    if constexpr (BFR_ABL & 8) {
      bfr_touch(fb[s % 3][0]);
      bfr_touch(fb[s % 3][1]);
    } else {
      fb[s % 3][0] = base[R * HW + v];
      fb[s % 3][1] = base[R * HW + v + PLANE_B];
    }
  };
  auto mfma3 = [&](auto uc, auto sc) {  // kernel row u against the pixel fragment of step s: output row R - u
    constexpr int u = decltype(uc)::value, s = decltype(sc)::value, R = s % 6, r = R - u;
    const uint4(&a)[2][NTW] = fa[u];
    const uint4(&bb)[2] = fb[s % 3];
    if constexpr (BFR_ABL & 4) {
      bfr_use(bb[0]);
      bfr_use(bb[1]);
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        bfr_use(a[0][nt]);
        bfr_use(a[1][nt]);
      }
      return;
    }
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[nt][r] = mfma16x<F16>(a[0][nt], bb[1], acc[nt][r]);  // w_h * x_m
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[nt][r] = mfma16x<F16>(a[1][nt], bb[0], acc[nt][r]);  // w_m * x_h
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[nt][r] = mfma16x<F16>(a[0][nt], bb[0], acc[nt][r]);  // w_h * x_h
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  if (grp < count) {  // the first own stage: nothing to overlap the wait and the first fragments with
    bfr_wait(cnt + b, NPS * (k + 1), dead);
    ldA(I0{}, I0{}, I0{});
    ldB(hb, I0{});
    ldA(I1{}, I0{}, I0{});
    ldB(hb, I1{});
  }
  for (int ti = grp; ti < count; ti += 2) {
    const int ng = (ti | 1) < count ? 2 : 1;
    const int n = o_n, r0 = o_y * TH, c0 = o_x * TW;
    cv_offsets(r0, c0);
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int r = 0; r < MR; ++r) acc[nt][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    srk_static_for<0, ICC>([&](auto ccc) {
      constexpr int cc = decltype(ccc)::value, ccn = cc + 1 < ICC ? cc + 1 : 0;
      const long long k1 = BFR_CLK();
      // the next own stage: its slot, its use count, whether it exists
      const bool has_next = cc + 1 < ICC || ti + 2 < count;
      int nb = b + (cc + 1 < ICC ? ng : 2 * ICC - (ICC - 1) * ng);
      unsigned nk = k;
      while (nb >= nbuf) {
        nb -= nbuf;
        ++nk;
      }
      const uint4* hbn = hal0 + (size_t)nb * HBUF + lane_b;
      unsigned peek_v = 0;
      // 18 steps (kernel column v, halo row R); a step's groups in ascending kernel row; behind the first group: the pixel
      // fragment of step s + 2 (of the NEXT stage from step 16 on), a pending store, the ring bookkeeping; behind the
      // first group as well: one kernel row of filter fragments (see below)
      srk_static_for<0, 18>([&](auto sc) {
        constexpr int s = decltype(sc)::value, v = s / 6, R = s - 6 * v;
        constexpr int u_lo = R - (MR - 1) > 0 ? R - (MR - 1) : 0, u_hi = R < 2 ? R : 2;
        constexpr int g = cc * 18 + s;
        srk_static_for<u_lo, u_hi + 1>([&](auto uc) {
          constexpr int u = decltype(uc)::value;
          mfma3(uc, sc);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (u == u_lo) {
            if constexpr (s == 16) {
              // The next stage's slot must be full before its first fragments are requested (right below).  Checked HERE, behind
              // the hand-back of the current slot (step 15): a consumer never holds a slot while it waits for another one --
              // with the fused producers (one 4-wave rendezvous per tile, two slots per tile) holding and waiting closes a
              // cycle: chunk-1 producers wait for group B's slot, group B for its next stage, that stage's producer for the
              // rendezvous with the chunk-1 producers.
              if (has_next && !dead) {
                unsigned seen = (unsigned)__builtin_amdgcn_readfirstlane((int)peek_v), spins = 0;
                while ((int)(seen - NPS * (nk + 1)) < 0) {
                  __builtin_amdgcn_s_sleep(1);
                  seen = bfr_peek(cnt + nb);
                  if (++spins > BFR_SPIN_CAP) {
                    dead = true;
                    break;
                  }
                }
              }
              asm volatile("" ::: "memory");
            }
            if constexpr (s + 2 < 18) ldB(hb, std::integral_constant<int, s + 2>{});
            else ldB(hbn, std::integral_constant<int, s + 2 - 18>{});
            if constexpr (g % SPER == 0 && g / SPER < NST) store_slot(std::integral_constant<int, g / SPER>{});
            if constexpr (s == 8) peek_v = __hip_atomic_load(cnt + nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if constexpr (s == 15) {  // the slot's last fragment (step 17) has just been requested: hand the slot back (the LDS
                                      // executes one wave's operations in order: the count lands behind the reads)
              bfr_signal(cnt + BFR_MAXBUF + b);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          // filter fragments: kernel row u's last use in a column is step R = u + 3, and requesting its successor right
          // behind that use costs WAR wait states -- so row 0 follows at R = 4 and row 1 at R = 5 (for the next column; from
          // v = 2 on: the next stage's first column), row 2 at R = 0 of the column that needs it at R = 2
          if constexpr (u == u_lo) {
            if constexpr (R == 0) {
              ldA(I2{}, std::integral_constant<int, v>{}, ccc);
              __builtin_amdgcn_sched_barrier(0);
            } else if constexpr (R >= 4) {
              if constexpr (v < 2) ldA(std::integral_constant<int, R - 4>{}, std::integral_constant<int, v + 1>{}, ccc);
              else ldA(std::integral_constant<int, R - 4>{}, I0{}, std::integral_constant<int, ccn>{});
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        });
      });
      ct[1] += BFR_CLK() - k1; ct[5] += 1;
      b = nb;
      k = nk;
      hb = hbn;
    });
    const long long k3 = BFR_CLK();
    park(n, r0, c0);
    ct[3] += BFR_CLK() - k3;
    adv2(o_n, o_y, o_x);
  }
  srk_static_for<0, NST>([&](auto qc) { store_slot(qc); });   // the last tile
#ifdef BFR_PROF
  if (B.prof && tid == 0) {
    long long* pr = B.prof + (size_t)blockIdx.x * 16 + 8;
    for (int i = 0; i < 4; ++i) pr[i] = ct[i];
    pr[4] = BFR_CLK_TOTAL() - ct_begin;
    pr[5] = ct[5];
    pr[6] = wall_clock64() - cw_begin;   // 100 MHz: the loop in wall time
    pr[-2] = cw_begin - cw_entry;        // ... from the block's first instruction to the loop (filter copy, set-up, barrier)
    B.prof[(size_t)(2048 + blockIdx.x) * 16 + 0] = cw_p1 - cw_entry;   // (the prologue split: rows 2048 .. of the buffer)
    B.prof[(size_t)(2048 + blockIdx.x) * 16 + 1] = cw_p2 - cw_p1;
    B.prof[(size_t)(2048 + blockIdx.x) * 16 + 2] = cw_p3 - cw_p2;
    pr[-1] = cw_entry;                   // ... when the block started (absolute)
  }
#endif
  (void)ct; (void)ct_begin;
  if (P.ep.y_amax) amax_commit(P.ep.y_amax, amax, blockIdx.x + wave, amax_peek(P.ep.y_amax, blockIdx.x + wave));
#ifdef BFR_PROF
  if (B.prof && tid == 0) B.prof[(size_t)blockIdx.x * 16 + 15] = wall_clock64();   // ... when consumer wave 0 left (absolute)
#endif
  if (dead) bfr_fail(lane);
}

// the canvas variant (64 input channels -> 32-channel slices): f16x3 forward, bf16x3 with or without the output mask
static int bfr_launch_cv(const BfwParams& B, size_t lds, int grid, hipStream_t s) {
  note_amax_written(B.P.ep.y_amax != nullptr);
  const dim3 blk(512);
  const bool omask = B.P.ep.out_relu != nullptr, res = B.P.ep.residual != nullptr;
  auto go = [&](auto f16c, auto omc, auto resc) {
    constexpr bool F = decltype(f16c)::value, O = decltype(omc)::value, R = decltype(resc)::value;
    static LdsLimit lim;
    lim.ensure(reinterpret_cast<const void*>(&k_conv_bfr<2, 2, F, true, O, R>), lds);
    note_kernel("k_conv_bfr<2,2%s,canvas%s%s%s>", F ? ",f16" : "", B.cv_sep ? "" : "0", O ? ",relu" : "", R ? ",res" : "");   // (canvas0: no separators)
    hipLaunchKernelGGL((k_conv_bfr<2, 2, F, true, O, R>), dim3(grid), blk, lds, s, B);
  };
  if (B.w_descale && res) go(std::true_type{}, std::false_type{}, std::true_type{});
  else if (B.w_descale) go(std::true_type{}, std::false_type{}, std::false_type{});
  else if (res) go(std::false_type{}, std::false_type{}, std::true_type{});
  else if (omask) go(std::false_type{}, std::true_type{}, std::false_type{});
  else go(std::false_type{}, std::false_type{}, std::false_type{});
  return check_launch("conv_bfr");
}

template <int NTW, int ICC>
static int bfr_launch_t(const BfwParams& B, size_t lds, int grid, hipStream_t s) {
  note_amax_written(B.P.ep.y_amax != nullptr);
  const dim3 blk(512);
  if (B.w_descale) {  // f16x3 arithmetic
    static LdsLimit limh;
    limh.ensure(reinterpret_cast<const void*>(&k_conv_bfr<NTW, ICC, true>), lds);
    note_kernel("k_conv_bfr<%d,%d,f16>", NTW, ICC);
    hipLaunchKernelGGL((k_conv_bfr<NTW, ICC, true>), dim3(grid), blk, lds, s, B);
    return check_launch("conv_bfr");
  }
  static LdsLimit lim;
  lim.ensure(reinterpret_cast<const void*>(&k_conv_bfr<NTW, ICC, false>), lds);
  note_kernel("k_conv_bfr<%d,%d>", NTW, ICC);
  hipLaunchKernelGGL((k_conv_bfr<NTW, ICC, false>), dim3(grid), blk, lds, s, B);
  return check_launch("conv_bfr");
}

// The canvas variant: 3x3 "same" gathers (stride 1, pad 1, output as large as the input), 64 input channels in two chunks,
// 32-channel output slices, plain NHWC stores, no input mask, f16x3 without / bf16x3 with or without the output mask;
// in- and output below 2 GiB.  SRK_BFR_CV: 0 never, 2 whenever applicable, unset = where the plain 8 x 16 tiling would be
// less than 90 % full (or the layer needs slices), the canvas is at least 88 % full and has four tiles per CU.
static int conv_bfr_canvas(const BfwParams& B0, hipStream_t s) {
  const int mode = env_int("SRK_BFR_CV", 1);
  if (mode == 0) return -1;
  BfwParams B = B0;
  MfmaConvParams& P = B.P;
  if (P.KHv != 3 || P.KWv != 3 || P.is != 1 || P.os != 1 || B.NB != 32 || B.ICc != 2 || P.IC != 64) return -1;
  if (P.IH != P.PH || P.IW != P.PW || P.OH != P.PH || P.OW != P.PW || P.iy0 != -1 || P.ix0 != -1 || P.oy0 != 0 || P.ox0 != 0)
    return -1;
  // patches of whole tiles: no separators, one patch per canvas row of cells -- every tile full (SRK_BFR_CV_EXACT=0: separators)
  const bool exact = P.PH % BFR_TH == 0 && P.PW % BFR_TW == 0 && env_int("SRK_BFR_CV_EXACT", 1) != 0;
  if (P.mask_y || P.ep.ps_r > 1 || P.PH < 9 + (exact ? 7 : 0) || P.PW < 17 + (exact ? 15 : 0)) return -1;
  if (P.ep.out_relu && (B.w_descale || P.ep.residual)) return -1;
  if (P.ep.residual && (uintptr_t)P.ep.residual % 16 != 0) return -1;
  const size_t in_bytes = (size_t)P.N * P.IH * P.IW * P.IC * 4, out_bytes = (size_t)P.N * P.OH * P.OW * P.OC * 4;
  if (in_bytes >= (1ull << 31) || out_bytes >= (1ull << 31)) return -1;
  const int cvH = P.PH + (exact ? 0 : 1), cvW = P.PW + (exact ? 0 : 1);
  long best_tiles = 0;
  int best_kx = 0;
  if (exact) {
    best_kx = 1;
    best_tiles = (long)P.N * (P.PH / BFR_TH) * (P.PW / BFR_TW);
  }
  for (int kx = 1; !exact && kx <= P.N && kx * cvW <= 4096; ++kx) {
    const long tx = (kx * cvW - 1 + BFR_TW - 1) / BFR_TW;
    const long ky = (P.N + kx - 1) / kx;
    const long ty = (ky * cvH - 1 + BFR_TH - 1) / BFR_TH;
    if (best_kx == 0 || tx * ty < best_tiles) {
      best_tiles = tx * ty;
      best_kx = kx;
    }
  }
  if (best_kx == 0 || best_tiles >= (1L << 29)) return -1;
  if (((double)P.N * cvH + 64.0) * cvH >= 4.0e9 || ((double)best_kx * cvW + 64.0) * cvW >= 4.0e9) return -1;  // (div_h / div_w exact)
  if (mode != 2) {
    const double px = (double)P.N * P.PH * P.PW;
    const long plain = (long)((P.PH + BFR_TH - 1) / BFR_TH) * ((P.PW + BFR_TW - 1) / BFR_TW) * P.N;
    const bool plain_ok = B.nsl == 1 && px >= 0.9 * (double)plain * (BFR_TH * BFR_TW);
    if (plain_ok || px < 0.88 * (double)best_tiles * (BFR_TH * BFR_TW) || best_tiles * B.nsl < 4L * kNumCU) return -1;
  }
  B.cv_kx = best_kx;
  B.cv_sep = exact ? 0 : 1;
  B.cv_mh = (unsigned)(((1ull << 32) + cvH - 1) / cvH);
  B.cv_mw = (unsigned)(((1ull << 32) + cvW - 1) / cvW);
  P.TH = BFR_TH; P.TW = BFR_TW; P.HH = BFR_HH; P.HW = BFR_HW;
  P.tiles_x = exact ? P.PW / BFR_TW : (best_kx * cvW - 1 + BFR_TW - 1) / BFR_TW;
  P.tiles_y = (int)(best_tiles / P.tiles_x);
  const size_t wbytes = (size_t)9 * B.ICc * 8 * B.NB * 16;
  const size_t slot_bytes = (size_t)8 * BFR_NPIXP * 16;
  const long lds_cap = 160L * 1024 - 512 - BFR_CNT_BYTES;
  long nbuf = (lds_cap - (long)wbytes) / (long)slot_bytes;
  const int want_buf = env_int("SRK_BFR_NBUF", 0);
  if (want_buf >= 3 && want_buf < nbuf) nbuf = want_buf;
  if (nbuf > BFR_MAXBUF) nbuf = BFR_MAXBUF;
  if (nbuf < 3) return -1;
  B.nbuf = (int)nbuf;
  B.perm = 1;
  B.NPIXp = BFR_NPIXP;
  B.late = env_int("SRK_BFR_PRIO", 2);
  const size_t lds = wbytes + (size_t)nbuf * slot_bytes + BFR_CNT_BYTES;
  B.ntiles = (int)best_tiles;
#ifdef BFR_PROF
  B.prof = g_bfr_prof;
#endif
  B.out_bytes = (unsigned)out_bytes;
  int grid = kNumCU - kNumCU % (8 * B.nsl);
  const long want = ((best_tiles + 7) / 8) * 8 * B.nsl;
  if (grid == 0) return -1;
  if (want < grid) grid = (int)want;
  if (B.dbg & 32)
    fprintf(stderr, "[srk] k_conv_bfr<2,2,canvas>: %d patches across, %d x %d tiles, %d slices, ring %d, grid %d\n", best_kx,
            P.tiles_y, P.tiles_x, B.nsl, B.nbuf, grid);
  return bfr_launch_cv(B, lds, grid, s);
}

// B: the launch as conv_bfw_gather prepared it up to the tile choice (P, wq, ICc, NB, nsl, ...).  -1 when the layer is not
// one of the ring kernel's: a stride-1 3x3 gather with 32 or 48 output channels in one slice and 32 or 64 input
// channels, no gradient masks, an output whose 8 x 16 tiles are mostly full, a ring of at least three slots beside the
// filter.  SRK_BFR: 0 never, 1 whenever applicable, unset = the efficiency rule.
int conv_bfr_launch(const BfwParams& B0, hipStream_t s) {
  const int mode = env_int("SRK_BFR", 2);
  if (mode == 0) return -1;
  BfwParams B = B0;
  MfmaConvParams& P = B.P;
  const int ntw = B.NB / 16;
  {
    const int rc = conv_bfr_canvas(B0, s);
    if (rc != -1) return rc;
  }
  if (P.ep.residual) return -1;   // (only the canvas variant above adds a residual)
  if (P.KHv != 3 || P.KWv != 3 || P.is != 1 || B.nsl != 1 || (ntw != 2 && ntw != 3) || B.NB != P.OC) return -1;
  if ((B.ICc != 1 && B.ICc != 2) || P.IC % 8 != 0) return -1;
  if (ntw == 3 && B.ICc != 1) return -1;  // (48 channels from 64: the instantiation spills; no net has that layer)
  if (P.mask_y || P.ep.out_relu) return -1;
  P.TH = BFR_TH; P.TW = BFR_TW; P.HH = BFR_HH; P.HW = BFR_HW;
  P.tiles_y = (P.PH + BFR_TH - 1) / BFR_TH;
  P.tiles_x = (P.PW + BFR_TW - 1) / BFR_TW;
  const long ntiles = (long)P.tiles_x * P.tiles_y * P.N;
  if (ntiles >= (1L << 29)) return -1;
  if (mode != 1) {  // worth it from ~90 % full tiles and four tiles per CU upwards (else k_conv_bfw's free tile shapes)
    if ((double)P.PH * P.PW < 0.9 * (double)P.tiles_y * P.tiles_x * (BFR_TH * BFR_TW)) return -1;
    if (ntiles < 4L * kNumCU) return -1;
  }
  const size_t wbytes = (size_t)9 * B.ICc * 8 * B.NB * 16;
  const size_t slot_bytes = (size_t)8 * BFR_NPIXP * 16;
  const long lds_cap = 160L * 1024 - 512 - BFR_CNT_BYTES;
  long nbuf = (lds_cap - (long)wbytes) / (long)slot_bytes;
  const int want = env_int("SRK_BFR_NBUF", 0);
  if (want >= 3 && want < nbuf) nbuf = want;
  if (nbuf > BFR_MAXBUF) nbuf = BFR_MAXBUF;
  if (nbuf < 3) return -1;
  B.nbuf = (int)nbuf;
  B.perm = 1;
  B.NPIXp = BFR_NPIXP;
  B.late = env_int("SRK_BFR_PRIO", 2);  // which role issues first on its SIMD (k_conv_bfr: prio)
  const size_t lds = wbytes + (size_t)nbuf * slot_bytes + BFR_CNT_BYTES;
  B.ntiles = (int)ntiles;
#ifdef BFR_PROF
  B.prof = g_bfr_prof;
#endif
  B.out_bytes = (unsigned)((size_t)P.N * P.OH * P.OW * P.OC * sizeof(float));
  int grid = kNumCU;
  if (grid > ntiles) grid = (int)ntiles;
  if (B.dbg & 32)
    fprintf(stderr, "[srk] k_conv_bfr<%d,%d>: lds %zu B (filter %zu), ring %d x %zu B, grid %d of %ld tiles\n", ntw, B.ICc, lds,
            wbytes, B.nbuf, slot_bytes, grid, ntiles);
  if (ntw == 2) return B.ICc == 1 ? bfr_launch_t<2, 1>(B, lds, grid, s) : bfr_launch_t<2, 2>(B, lds, grid, s);
  return bfr_launch_t<3, 1>(B, lds, grid, s);
}

}  // namespace srk

#ifdef BFR_PROF
extern "C" void srk_debug_ring_prof(void* p) { srk::g_bfr_prof = static_cast<long long*>(p); }
#endif

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
