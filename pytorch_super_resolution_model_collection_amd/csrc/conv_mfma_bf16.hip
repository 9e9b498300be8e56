// Implicit-GEMM convolution on the bf16 matrix cores with a 3-term split ("bf16x3"):
//
//     a = a_hi + a_lo,  b = b_hi + b_lo   (hi = bf16(x) round-to-nearest, lo = bf16(x - hi))
//     a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi          (dropped a_lo*b_lo <= 2^-16 |a*b|)
//
// accumulated in fp32 by v_mfma_f32_16x16x32_bf16.  Three bf16 MFMAs (each 16x the fp32 MFMA
// rate) replace one fp32 MFMA: ~5.3x the fp32 ceiling (157 TF -> ~830 TF effective) at an error of
// ~1e-5 relative — two orders inside the 1e-3 contract (tests hold this path to 1e-4).  This is
// what lifts the FLOP-bound nets (ESPCN forward: 75 FLOP/B) towards the HBM roofline.
//
// Structure (one 256-thread block = <=256 output pixels x <=64 output channels):
//   * the NHWC fp32 halo of the tile is read ONCE from HBM (coalesced 16-byte loads), split into
//     hi/lo bf16 on the fly and staged in LDS as [plane][8-channel group][pixel][8 x bf16]: the 16
//     lanes of an MFMA row group read 16 consecutive pixels x 16 B = 256 contiguous bytes
//     (bank-conflict-free ds_read_b128), 32 channels per pass;
//   * filters are pre-split at pack time into the same [plane][group][co][8] layout, so the
//     per-tap slice (8 KB) is a linear copy into a double-buffered LDS slot (next tap prefetched
//     into registers during the MFMAs) — one barrier per tap;
//   * each wave owns 64 pixels x 64 channels = 4 x 4 MFMA tiles (64 accumulator VGPRs); per tap
//     and 32-channel chunk it issues 16 ds_read_b128 for 48 MFMAs;
//   * epilogue identical to the fp32 kernel (bias / activation / residual / pixel-shuffle store).
#include "srk_common.h"
#include "conv_problem.h"
#include "conv_tile.h"
#include "pack_items.h"
#include "bf16_frag.h"
#include <type_traits>
#include <stdlib.h>

namespace srk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Bf3Params {
  MfmaConvParams P;
  const float* w_descale;  // k_conv_bf3_rows<.., f16>: trailer {2^-kw, 2^kw} of the fp16 filter section (wq points there)
  const uint4* wq;  // prepared filters
  int ICc;          // 32-channel chunks
  int OCb;          // 64-channel output blocks
  int NB;           // output channels per block in the prepared layout (NT*16)
  int NPIXp;        // halo pixels rounded up to 16
  int dbg;          // ablation switches (SRK_DBG env): 1 skip halo loads, 2 skip epilogue, 4 skip MFMAs, 8 skip weight copies
};

__device__ __forceinline__ void split8(const float (&f)[8], uint4& hi, uint4& lo) {
  bf16x8 h, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 hh = (__bf16)f[e];
    h[e] = hh;
    l[e] = (__bf16)(f[e] - (float)hh);
  }
  hi = __builtin_bit_cast(uint4, h);
  lo = __builtin_bit_cast(uint4, l);
}

// ---------------------------------------------------------------------------------------------
// Filter preparation: torch-layout fp32 weights -> [tap][chunk][ocb][plane][group][co][8] bf16
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bf3_pack(const float* __restrict__ w, uint4* __restrict__ dst, int Cout,
                                                  int Cin, int KH, int KW, int transposed, int ps_r, int bwd, int IC,
                                                  int OC, int ICc, int OCb, int NB, uint4* __restrict__ f16dst,
                                                  const float* __restrict__ f16trailer) {
  const long items = (long)KH * KW * ICc * OCb * 4 * NB;
  const long it = (long)blockIdx.x * 256 + threadIdx.x;
  if (it >= items) return;
  pack_bf3_item(it, w, dst, Cout, Cin, KH, KW, transposed, ps_r, bwd, IC, OC, ICc, OCb, NB, f16dst,
                f16dst ? f16trailer[1] : 1.f);
}

__global__ __launch_bounds__(256) void k_pack_f16_trailer(const float* __restrict__ w, long elems, float* __restrict__ trailer) {
  pack_f16_trailer(w, elems, trailer);
}

__global__ void k_bf3_pack_rows(const float* __restrict__ w, uint4* __restrict__ dst, int Cout, int Cin, int KH, int KW,
                                int transposed, int ps_r, int bwd, int IC, int OC, int KS, int OCb, int NB,
                                uint4* __restrict__ f16dst, const float* __restrict__ f16trailer);
__global__ void k_pack_f16_trailer(const float* __restrict__ w, long elems, float* __restrict__ trailer);

static inline int bf3_nb(int OC) { return pk_nb(OC); }

size_t bf3_prepared_offset(size_t elems) { return pk_prepared_offset(elems); }

// planes h, m of the main layout; the third plane (bf16x6 kernels, conv_bfd.hip) follows it
size_t bf3_main_bytes(int IC, int OC, int T) { return pk_main_bytes(IC, OC, T); }

size_t bf3_prepared_bytes(int IC, int OC, int T) { return bf3_main_bytes(IC, OC, T) / 2 * 3; }

// forward buffers: fp16 planes (h, m of w * 2^kw) + trailer behind the bf16 planes (pack_items.h)
size_t f16_section_offset(int IC, int OC, int T) { return pk_f16_offset(IC, OC, T); }
size_t f16_section_bytes(int IC, int OC, int T) { return pk_f16_bytes(IC, OC, T); }

int bf3_pack_prepared(const float* w, void* packed_base, int Cout, int Cin, int KH, int KW, int transposed, int ps_r,
                      int bwd, hipStream_t s) {
  const int IC = bwd ? Cout : Cin, OC = bwd ? Cin : Cout;
  const int ICc = (IC + 31) / 32, OCb = (OC + 63) / 64, NB = bf3_nb(OC);
  const size_t elems = (size_t)KH * KW * Cin * Cout;
  uint4* dst = reinterpret_cast<uint4*>(static_cast<char*>(packed_base) + bf3_prepared_offset(elems));
  const int gather_trans = bwd ? !transposed : transposed;
  if (IC <= 4 && !gather_trans) {  // row-packed layout of k_conv_bf3_rows
    const int KS = (KW + 7) / 8;
    const long items = (long)KH * KS * OCb * 4 * NB;
    uint4* f16dst = nullptr;
    float* trailer = nullptr;
    if (!bwd) {  // forward buffers: fp16 row-packed planes + trailer in the f16 section (k_conv_bf3_rows<.., f16>)
      char* f16base = reinterpret_cast<char*>(dst) + f16_section_offset(IC, OC, KH * KW);
      f16dst = reinterpret_cast<uint4*>(f16base);
      trailer = reinterpret_cast<float*>(f16base + bf3_main_bytes(IC, OC, KH * KW));
      hipLaunchKernelGGL(k_pack_f16_trailer, dim3(1), dim3(256), 0, s, w, (long)elems, trailer);
    }
    hipLaunchKernelGGL(k_bf3_pack_rows, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, w, dst, Cout, Cin, KH,
                       KW, transposed, ps_r, bwd, IC, OC, KS, OCb, NB, f16dst, (const float*)trailer);
    return check_launch("bf3_pack_prepared_rows");
  }
  const long items = (long)KH * KW * ICc * OCb * 4 * NB;
  uint4* f16dst = nullptr;
  float* trailer = nullptr;
  if (!bwd) {  // forward buffers carry the fp16 planes of SRK_ALGO_MFMA_F16X3: layer maximum first, then the planes
    char* f16base = reinterpret_cast<char*>(dst) + f16_section_offset(IC, OC, KH * KW);
    f16dst = reinterpret_cast<uint4*>(f16base);
    trailer = reinterpret_cast<float*>(f16base + bf3_main_bytes(IC, OC, KH * KW));
    hipLaunchKernelGGL(k_pack_f16_trailer, dim3(1), dim3(256), 0, s, w, (long)elems, trailer);
  }
  hipLaunchKernelGGL(k_bf3_pack, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, w, dst, Cout, Cin, KH, KW,
                     transposed, ps_r, bwd, IC, OC, ICc, OCb, NB, f16dst, (const float*)trailer);
  return check_launch("bf3_pack_prepared");
}

// ---------------------------------------------------------------------------------------------
// Kernel
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 mfma_bf16(const uint4& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0,
                                                 0);
}

// halo chunk: channels [cb, cb+32) of every halo pixel -> hi/lo planes.
// thread -> (8-channel group g = tid&3, pixel slot tid>>2); pixels advance by 64 per pass with an
// incremental (row, col) update instead of a division per item.  All global loads of a thread
// (<= BF3_STAGE_IT passes) are issued before the first conversion, so the HBM/L2 latency is paid
// once per chunk instead of once per pass.
constexpr int BF3_STAGE_IT = 6;  // covers halos of <= 384 pixels per batch; larger halos loop over batches

template <bool MASK, int NTHR>
__device__ __forceinline__ void bf3_stage_halo_t(const Bf3Params& B, uint4* hal, int n, int r0, int c0, int cb) {
  const MfmaConvParams& P = B.P;
  const int npix = P.HH * P.HW;
  // (4 adjacent lanes = the 4 channel groups of one pixel: 128 contiguous bytes per pixel for the global loads.
  //  The resulting ds_write_b128 pattern is 4-way bank-conflicted; remapping lanes to make the LDS writes
  //  conflict-free breaks the adjacent-lane coalescing of the loads and measured 0.64 -> 0.81 ms on the c2 layer.)
  const int g = threadIdx.x & 3;
  const int hp0 = threadIdx.x >> 2;
  int hy = hp0 / P.HW, hx = hp0 - hy * P.HW;
  constexpr int PPP = NTHR / 4;  // pixels per pass
  const int dy64 = PPP / P.HW, dx64 = PPP - dy64 * P.HW;
  const int iyb = r0 * P.is + P.iy0, ixb = c0 * P.is + P.ix0;
  const int ch = cb + g * 8;
  const bool ch_any = ch < P.IC, ch_vec = P.vec_in && ch + 7 < P.IC;
  const InAddr ia = conv_in_addr(P, ch);
  const size_t img_off = (size_t)n * P.IH * ia.sA;  // wave-uniform: scalar base pointers
  const float* __restrict__ inb = P.in + img_off;
  const float* __restrict__ mkb = MASK ? P.mask_y + img_off : nullptr;
  for (int base = hp0; base < npix; base += PPP * BF3_STAGE_IT) {
    f32x4 v0[BF3_STAGE_IT], v1[BF3_STAGE_IT], m0[BF3_STAGE_IT], m1[BF3_STAGE_IT];
    // pass 1: issue every load of this batch
#pragma unroll
    for (int k = 0; k < BF3_STAGE_IT; ++k) {
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
      hy += dy64;
      hx += dx64;
      if (hx >= P.HW) {
        hx -= P.HW;
        ++hy;
      }
    }
    // pass 2: mask, split, store
#pragma unroll
    for (int k = 0; k < BF3_STAGE_IT; ++k) {
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
        uint4 hi, lo;
        split8(f, hi, lo);
        hal[(0 * 4 + g) * B.NPIXp + hp] = hi;
        hal[(1 * 4 + g) * B.NPIXp + hp] = lo;
      }
    }
  }
}

template <int NTHR>
__device__ __forceinline__ void bf3_stage_halo(const Bf3Params& B, uint4* hal, int n, int r0, int c0, int cb) {
  if (B.P.mask_y)
    bf3_stage_halo_t<true, NTHR>(B, hal, n, r0, c0, cb);
  else
    bf3_stage_halo_t<false, NTHR>(B, hal, n, r0, c0, cb);
}

constexpr int BF3_MAXTAPS = 128;   // taps with precomputed tables (larger kernels: computed on the fly)
constexpr int BF3_EPI_STRIDE = 68; // floats per staged output row (64 + 4: conflict-free float4 rows)

// Epilogue shared by the bf16x3 kernels: accumulators -> LDS (wave-private 32 x 64 slab, two halves)
// -> 16-byte stores in which a wave writes 4 pixels x 256 contiguous bytes.  (Storing straight from
// the MFMA layout — 64-byte segments of 16 different pixels per instruction — measured 2x slower on
// the epilogue: partial-line HBM writes.)  C/D layout: col = lane&15 (channel), row = (lane>>4)*4+reg
// (pixel).  Everything that depends only on the lane's channel group is hoisted (EpiCol).
// VEC_ONLY: the host guarantees that every lane's 4-channel group takes the 16-byte store path (epi_all_vector), so
// the scalar fallback — and the integer divisions of its index math that the compiler hoists in front of the store
// loop — is not compiled into the kernel.
template <int NT, bool VEC_ONLY = false>
__device__ __forceinline__ void bf3_epilogue(const MfmaConvParams& P, float* smem_f, const f32x4 (&acc)[4][NT], int n,
                                             int r0, int c0, int ocb, int wave, int lane, int dbg = 0) {
  if (dbg & 2) return;
  float amax = 0.f;  // running maximum of what this lane stores (ep.y_amax; vector store path only)
  const float peeked = VEC_ONLY ? amax_peek(P.ep.y_amax, blockIdx.x + wave) : 0.f;
  const int j = lane & 15, kq = lane >> 4;
  const int npx = P.TH * P.TW;
  __syncthreads();
  float* st = smem_f + wave * (32 * BF3_EPI_STRIDE);
  constexpr int Q4 = NT * 4;        // float4 columns per row
  constexpr int RPI = 64 / Q4;      // rows per wave pass (NT = 3 leaves 4 lanes idle)
  const int row0 = lane / Q4, q4 = lane - row0 * Q4;
  const int oc4 = ocb + q4 * 4;
  const bool lane_on = row0 < RPI && oc4 < P.OC;
  const int tw_magic = div_small_magic(P.TW);
  EpiCol col{};
  if (lane_on) col = epi_col_setup(P.ep, P.OW, P.OC, oc4);
  const EpiTile et = epi_tile_setup(P, n, r0, c0);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
          st[(mh * 16 + kq * 4 + reg) * BF3_EPI_STRIDE + nt * 16 + j] = acc[2 * h + mh][nt][reg];
    __syncthreads();
    if (lane_on) {
#pragma unroll 2
      for (int row = row0; row < 32; row += RPI) {
        const int m = wave * 64 + h * 32 + row;
        if (m < npx) {
          const int r = div_small(m, tw_magic), c = m - r * P.TW;
          const int pr = r0 + r, pc = c0 + c;
          if (pr < P.PH && pc < P.PW && !((dbg & 64) && acc[0][0][0] != 123.456f)) {
            const epi_f4 v = *reinterpret_cast<const epi_f4*>(st + row * BF3_EPI_STRIDE + q4 * 4);
            if (VEC_ONLY || col.vec) {
              const epi_f4 o = epi_store4_tile(P.ep, col, et, r, c, v, P.out);
              if (VEC_ONLY && P.ep.y_amax) amax = abs_max4(amax, o);
            } else {
              epi_store4_col(P.ep, col, P.OH, P.OW, P.OC, n, P.oy0 + pr * P.os, P.ox0 + pc * P.os, v, P.out);
            }
          }
        }
      }
    }
    __syncthreads();
  }
  if (VEC_ONLY && P.ep.y_amax) amax_commit(P.ep.y_amax, amax, blockIdx.x + wave, peeked);  // per wave: no barrier at the tail
}

// NT <= 2 (the c2 benchmark's 64->32 layer) must stay within 168 VGPRs: three resident blocks per CU instead of two
// is worth 25 % on that layer (0.64 vs 0.89 ms) — the bound makes the compiler hold the line when code is added.
template <int NT, int NW, bool VEC_ONLY = false>
__global__ __launch_bounds__(64 * NW, (NT <= 2 && NW == 4) ? 3 : 2) void k_conv_bf3(Bf3Params B) {
  constexpr int NTHR = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
  const MfmaConvParams& P = B.P;
  uint4* hal = smem4;                       // [2][4][NPIXp]
  uint4* wl = smem4 + 8 * B.NPIXp;          // [2 bufs][2][4][NB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kq = lane >> 4;
  int b = blockIdx.x;
  const int txi = b % P.tiles_x;
  b /= P.tiles_x;
  const int tyi = b % P.tiles_y;
  const int n = b / P.tiles_y;
  const int r0 = tyi * P.TH, c0 = txi * P.TW;
  const int ocbi = blockIdx.y;
  const int ocb = ocbi * 64;
  const int npx = P.TH * P.TW;
  const int NB = B.NB;
  const int wslot = 8 * NB;  // uint4 per weight buffer
  const int T = P.KHv * P.KWv;

  int hp[4];
  {
    const int tw_magic = div_small_magic(P.TW);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      int m = wave * 64 + mt * 16 + j;
      if (m >= npx) m = 0;
      const int r = div_small(m, tw_magic), c = m - r * P.TW;
      hp[mt] = (r * P.is) * P.HW + c * P.is + kq * B.NPIXp;
    }
  }
  f32x4 acc[4][NT];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const bool wave_live = wave * 64 < npx;
  constexpr int WCP = 512 / NTHR;  // weight-slice uint4 per thread (wslot <= 512)
  const int lo_plane = 4 * B.NPIXp;
  const int wlane = kq * NB + j;
  // tap walk kept in scalar registers: (u, v) -> halo offset u*HW + v and weight tap (wh0 + wdh*u)*KW_full + ww0 + wdw*v
  const int wtap_row = P.wdh * P.KW_full - P.wdw * P.KWv;  // weight-tap step at a row wrap (added to the +wdw step)
  const size_t wtap_stride = (size_t)B.ICc * B.OCb * (size_t)wslot;

  if (T > 0) {
    for (int cc = 0; cc < B.ICc; ++cc) {
      __syncthreads();  // previous chunk fully consumed
      const uint4* wbase = B.wq + ((size_t)cc * B.OCb + ocbi) * (size_t)wslot;
      int wtap = P.wh0 * P.KW_full + P.ww0;  // weight tap of (u, v) = (0, 0)
      if (!(SRK_KDBG(B.dbg) & 1)) bf3_stage_halo<NTHR>(B, hal, n, r0, c0, cc * 32);
      {
        // (loads unconditional from a clamped index, only the LDS writes conditional: load + write under one branch
        //  compiled to load, s_waitcnt vmcnt(0), write -- per copy, one after the other)
        const uint4* src = wbase + (size_t)wtap * wtap_stride;
        uint4 w0[WCP];
#pragma unroll
        for (int c = 0; c < WCP; ++c) w0[c] = src[tid + c * NTHR < wslot ? tid + c * NTHR : 0];
#pragma unroll
        for (int c = 0; c < WCP; ++c)
          if (tid + c * NTHR < wslot) wl[tid + c * NTHR] = w0[c];
      }
      __syncthreads();
      int toff = 0, tv = 0;
      for (int t = 0; t < T; ++t) {
        // next tap's weight index (row wrap when v reaches KWv)
        int wnext = wtap + P.wdw;
        if (tv + 1 == P.KWv) wnext += wtap_row;
        uint4 wr[WCP];
#pragma unroll
        for (int c = 0; c < WCP; ++c) wr[c] = make_uint4(0, 0, 0, 0);
        if (t + 1 < T && !(SRK_KDBG(B.dbg) & 8)) {  // prefetch the next tap's slice; lands while the MFMAs below run
          const uint4* src = wbase + (size_t)wnext * wtap_stride;
#pragma unroll
          for (int c = 0; c < WCP; ++c)
            if (tid + c * NTHR < wslot) wr[c] = src[tid + c * NTHR];
        }
        if (wave_live && !(SRK_KDBG(B.dbg) & 4)) {
          const uint4* wb = wl + (t & 1) * wslot + wlane;
          const uint4* hb = hal + toff;
          uint4 ah[4], al[4];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            ah[mt] = hb[hp[mt]];
            al[mt] = hb[hp[mt] + lo_plane];
          }
          uint4 bh[NT], bl[NT];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            bh[nt] = wb[nt * 16];
            bl[nt] = wb[4 * NB + nt * 16];
          }
          // three passes, each over 4*NT independent accumulators (no back-to-back dependency)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma_bf16(al[mt], bh[nt], acc[mt][nt]);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma_bf16(ah[mt], bl[nt], acc[mt][nt]);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma_bf16(ah[mt], bh[nt], acc[mt][nt]);
        }
        if (t + 1 < T) {
          uint4* wn = wl + ((t + 1) & 1) * wslot;
#pragma unroll
          for (int c = 0; c < WCP; ++c)
            if (tid + c * NTHR < wslot) wn[tid + c * NTHR] = wr[c];
        }
        // advance the tap walk
        wtap = wnext;
        ++toff;
        if (++tv == P.KWv) {
          tv = 0;
          toff += P.HW - P.KWv;
        }
        __syncthreads();
      }
    }
  }
  if (SRK_KDBG(B.dbg) & 2) {
    if (acc[0][0][0] == 123.456f) P.out[0] = 1.f;  // keep the accumulators live
    return;
  }
  bf3_epilogue<NT, VEC_ONLY>(P, reinterpret_cast<float*>(smem4), acc, n, r0, c0, ocb, wave, lane);
}

// ---------------------------------------------------------------------------------------------
// Row-packed variant for IC <= 4 (CONV gathers: the 3->64 first layers).  A K step of 32 is one
// kernel row segment: 8 kw positions x 4 (zero-padded) channels; lane group kq supplies kw slots
// 2kq, 2kq+1, i.e. 16 contiguous bytes of the [pixel][4 x bf16] halo planes.  A 5x5x3 filter is 5
// K steps (47 % dense) instead of 25 mostly-empty channel-padded taps.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bf3_pack_rows(const float* __restrict__ w, uint4* __restrict__ dst, int Cout,
                                                       int Cin, int KH, int KW, int transposed, int ps_r, int bwd,
                                                       int IC, int OC, int KS, int OCb, int NB, uint4* __restrict__ f16dst,
                                                       const float* __restrict__ f16trailer) {
  const long items = (long)KH * KS * OCb * 4 * NB;
  const long it = (long)blockIdx.x * 256 + threadIdx.x;
  if (it >= items) return;
  pack_bf3_rows_item(it, w, dst, Cout, Cin, KH, KW, transposed, ps_r, bwd, IC, OC, KS, OCb, NB, f16dst,
                     f16dst ? f16trailer[1] : 1.f);
}

// ---------------------------------------------------------------------------------------------
// Whole-model packing in ONE launch: grid = (blocks, layers).  Table row (int64 x 12) per layer:
//   0 w_off (floats from params_base)   1 fwd_off (bytes from packed_base)   2 bwd_off (bytes)
//   3 Cout  4 Cin  5 KH  6 KW  7 transposed  8 ps_r  9 bias_off (floats, -1 none)
//   10 bias_ps_off (bytes, -1 none)  11 reserved
// Each packed buffer has the srk_pack_weight_fwd / _bwd format (fp32 layout + bf16x3 planes).
// ---------------------------------------------------------------------------------------------
// fp16 section of a forward buffer inside the batched pack: destination of the planes and the layer's scale -- from the
// scratch word of k_pack_batched_amax (`wamax`; the first thread of the layer records the trailer) or from the trailer
// k_pack_batched_trailers wrote in front of this kernel
__device__ __forceinline__ void pack_f16_setup(uint4* q, int IC, int OC, int T, const float* wamax, long tid0,
                                               uint4*& f16dst, float& f16scale) {
  char* f16base = reinterpret_cast<char*>(q) + pk_f16_offset(IC, OC, T);
  f16dst = reinterpret_cast<uint4*>(f16base);
  float* trailer = reinterpret_cast<float*>(f16base + pk_main_bytes(IC, OC, T));
  if (wamax) {
    const float a = *wamax;
    int k = 0;
    if (a != 0.f) k = 140 - (int)((__float_as_uint(a) >> 23) & 0xff);
    k = k < -126 ? -126 : (k > 126 ? 126 : k);
    f16scale = __uint_as_float((unsigned)(127 + k) << 23);
    if (tid0 == 0) {
      trailer[0] = __uint_as_float((unsigned)(127 - k) << 23);
      trailer[1] = f16scale;
    }
  } else {
    f16scale = trailer[1];
  }
}

__device__ __forceinline__ void pack_one_dir(const float* w, char* dst, int Cout, int Cin, int KH, int KW,
                                             int transposed, int ps_r, int bwd, long tid0, long stride,
                                             const float* wamax = nullptr) {
  const long elems = (long)KH * KW * Cin * Cout;
  float* wp = reinterpret_cast<float*>(dst);
  for (long e = tid0; e < elems; e += stride) pack_f32_item((int)e, w, wp, Cout, Cin, KH, KW, transposed, ps_r, bwd);
  const int IC = bwd ? Cout : Cin, OC = bwd ? Cin : Cout;
  const int OCb = (OC + 63) / 64, NB = OC >= 64 ? 64 : ((OC + 15) / 16) * 16;
  uint4* q = reinterpret_cast<uint4*>(dst + ((elems * 4 + 255) & ~255L));
  const int gather_trans = bwd ? !transposed : transposed;
  uint4* f16dst = nullptr;
  float f16scale = 1.f;
  if (IC <= 4 && !gather_trans) {
    const int KS = (KW + 7) / 8;
    const long items = (long)KH * KS * OCb * 4 * NB;
    if (!bwd) pack_f16_setup(q, IC, OC, KH * KW, wamax, tid0, f16dst, f16scale);
    for (long it = tid0; it < items; it += stride)
      pack_bf3_rows_item(it, w, q, Cout, Cin, KH, KW, transposed, ps_r, bwd, IC, OC, KS, OCb, NB, f16dst, f16scale);
  } else {
    const int ICc = (IC + 31) / 32;
    const long items = (long)KH * KW * ICc * OCb * 4 * NB;
    if (!bwd) pack_f16_setup(q, IC, OC, KH * KW, wamax, tid0, f16dst, f16scale);
    for (long it = tid0; it < items; it += stride)
      pack_bf3_item(it, w, q, Cout, Cin, KH, KW, transposed, ps_r, bwd, IC, OC, ICc, OCb, NB, f16dst, f16scale);
  }
}

// Per-layer weight maxima for the f16 trailers of the forward buffers.  Column 11 of a table row = byte offset (in
// `packed`) of the layer's 4-byte scratch word, zeroed by the caller before the launch: blocks of 4096 weights take the
// maximum of their slice and raise the word (one atomic per block); k_pack_batched then derives the scale from it and
// its first block writes the trailer.  Column 11 < 0: one block scans the whole layer (k_pack_batched_trailers).
__global__ __launch_bounds__(256) void k_pack_batched_amax(const float* __restrict__ params, char* __restrict__ packed,
                                                           const long long* __restrict__ table, int slices) {
  const long long* row = table + (size_t)blockIdx.y * kPackCols;
  if (row[1] < 0 || row[11] < 0) return;
  const int Cout = (int)row[3], Cin = (int)row[4], KH = (int)row[5], KW = (int)row[6];
  const long elems = (long)KH * KW * Cin * Cout;
  const float* w = params + row[0];
  __shared__ float sm[4];
  float a = 0.f;
  const bool vec = (reinterpret_cast<uintptr_t>(w) & 15) == 0;   // flat parameter buffers keep every tensor 16-byte aligned
  // (whole block: nothing of this layer left for the slice -- a block's first element is 1024 x its index on the float4
  // path, 256 x its index on the scalar path of an unaligned tensor)
  if ((long)blockIdx.x * (vec ? 1024 : 256) >= elems) return;
  if (vec) {
    typedef float pa_f4 __attribute__((ext_vector_type(4)));
    const long n4 = elems >> 2;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)slices * 256) {
      const pa_f4 v = reinterpret_cast<const pa_f4*>(w)[e];
      a = fmaxf(fmaxf(a, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    if (blockIdx.x == 0)
      for (long e = (n4 << 2) + threadIdx.x; e < elems; e += 256) a = fmaxf(a, fabsf(w[e]));
  } else {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < elems; e += (long)slices * 256) a = fmaxf(a, fabsf(w[e]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a = fmaxf(a, __shfl_xor(a, o, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    a = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    atomicMax(reinterpret_cast<unsigned*>(packed + row[11]), __float_as_uint(a));
  }
}

__global__ __launch_bounds__(256) void k_pack_batched_trailers(const float* __restrict__ params, char* __restrict__ packed,
                                                               const long long* __restrict__ table) {
  const long long* row = table + (size_t)blockIdx.x * kPackCols;
  if (row[1] < 0 || row[11] >= 0) return;
  const int Cout = (int)row[3], Cin = (int)row[4], KH = (int)row[5], KW = (int)row[6];
  const long elems = (long)KH * KW * Cin * Cout;
  char* f16base = packed + row[1] + pk_prepared_offset(elems) + pk_f16_offset(Cin, Cout, KH * KW);
  pack_f16_trailer(params + row[0], elems, reinterpret_cast<float*>(f16base + pk_main_bytes(Cin, Cout, KH * KW)));
}

__global__ __launch_bounds__(256) void k_pack_batched(const float* __restrict__ params, char* __restrict__ packed,
                                                      const long long* __restrict__ table) {
  const long long* row = table + (size_t)blockIdx.y * kPackCols;
  const float* w = params + row[0];
  const int Cout = (int)row[3], Cin = (int)row[4], KH = (int)row[5], KW = (int)row[6];
  const int transposed = (int)row[7], ps_r = (int)row[8];
  const long tid0 = (long)blockIdx.x * 256 + threadIdx.x, stride = (long)gridDim.x * 256;
  const float* wamax = row[11] >= 0 ? reinterpret_cast<const float*>(packed + row[11]) : nullptr;
  const bool fast = row[12] >= 0;   // filters packed by k_pack_fast (only the pixel-shuffle bias is left for this kernel)
  if (fast && !(row[9] >= 0 && row[10] >= 0 && ps_r > 1)) return;
  if (!fast && row[1] >= 0) pack_one_dir(w, packed + row[1], Cout, Cin, KH, KW, transposed, ps_r, 0, tid0, stride, wamax);
  if (!fast && row[2] >= 0) pack_one_dir(w, packed + row[2], Cout, Cin, KH, KW, transposed, ps_r, 1, tid0, stride);
  if (row[9] >= 0 && row[10] >= 0 && ps_r > 1) {
    const float* b = params + row[9];
    float* bp = reinterpret_cast<float*>(packed + row[10]);
    const int C = Cout / (ps_r * ps_r);
    for (long e = tid0; e < Cout; e += stride) {
      const int q = (int)e / C, c = (int)e % C;
      bp[e] = b[c * ps_r * ps_r + q];
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Fast path of the whole-model pack for plain Conv2d filters (not transposed, <= 25 taps, channel counts without padding in
// the prepared layouts -- Cout = 32 or a multiple of 64, Cin = 16 / 32 / 48 or a multiple of 64 --, both directions wanted): the generic kernel gathers every packed value with its own 4-byte load at a stride of KH*KW or
// Cin*KH*KW floats -- ~6 M scattered loads per EDSR pack, 38 us.  Here a block reads the filters of 8 packed-consecutive
// output channels x one 32-channel chunk ONCE, coalesced, into LDS ([8][32][taps], row stride + 1 float against bank
// conflicts) and writes all six layouts of that tile from there in contiguous runs: fp32 forward / backward, bf16 planes
// h, m, l of both directions, fp16 planes of the forward buffer.  `fast_blocks`: (layer, local block) per block; local block
// = octet * ICc + chunk.  Same bytes as pack_one_dir (tests/test_train_gpu.py::test_pack_plan_equals_per_layer_pack).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_fast(const float* __restrict__ params, char* __restrict__ packed,
                                                   const long long* __restrict__ table, const int* __restrict__ fast_blocks) {
  extern __shared__ float tile[];
  const int layer = fast_blocks[2 * blockIdx.x], lb = fast_blocks[2 * blockIdx.x + 1];
  const long long* row = table + (size_t)layer * kPackCols;
  const float* __restrict__ w = params + row[0];
  const int Cout = (int)row[3], Cin = (int)row[4], KH = (int)row[5], KW = (int)row[6], ps_r = (int)row[8];
  const int T = KH * KW;
  const int ICc = (Cin + 31) / 32;
  const int oct = lb / ICc, cc = lb - oct * ICc;
  const int cop0 = oct * 8;                       // first of 8 packed-consecutive output channels
  const int CW = Cin - cc * 32 < 32 ? Cin - cc * 32 : 32;
  const int RS = 32 * T + 1;                      // LDS row stride (floats)
  const int tid = threadIdx.x;
  const long elems = (long)T * Cin * Cout;
  // torch channel of packed channel cop (pixel-shuffle layers: packed order (i, j, c) -> c * r^2 + i * r + j)
  auto torch_co = [&](int cop) {
    if (ps_r <= 1) return cop;
    const int C = Cout / (ps_r * ps_r);
    const int q = cop / C, c = cop - q * C;
    return c * ps_r * ps_r + q;
  };
#pragma unroll 1
  for (int k = 0; k < 8; ++k) {
    const float* src = w + ((size_t)torch_co(cop0 + k) * Cin + cc * 32) * T;
    for (int r = tid; r < CW * T; r += 256) tile[k * RS + r] = src[r];
  }
  __syncthreads();
  char* dstf = packed + row[1];
  char* dstb = packed + row[2];
  // ---- fp32 layouts: forward wp[tap][ci][co'], backward wp[tap][co'][ci]
  {
    float* wf = reinterpret_cast<float*>(dstf);
    float* wb = reinterpret_cast<float*>(dstb);
    const int n = T * CW * 8;
    for (int i = tid; i < n; i += 256) {
      const int k = i & 7, rest = i >> 3;
      const int tap = rest / CW, c = rest - tap * CW;
      wf[((size_t)tap * Cin + cc * 32 + c) * Cout + cop0 + k] = tile[k * RS + c * T + tap];
    }
    for (int i = tid; i < n; i += 256) {
      const int c = i % CW, rest = i / CW;
      const int k = rest & 7, tap = rest >> 3;
      wb[((size_t)tap * Cout + cop0 + k) * Cin + cc * 32 + c] = tile[k * RS + c * T + tap];
    }
  }
  // ---- forward planes: item (tap, group g, k): 8 input channels g*8 .. g*8+7 of output channel cop0 + k
  {
    const int OCb = (Cout + 63) / 64, NB = pk_nb(Cout);
    uint4* q = reinterpret_cast<uint4*>(dstf + pk_prepared_offset(elems));
    uint4* third = q + (size_t)T * ICc * OCb * (size_t)(8 * NB);
    char* f16base = reinterpret_cast<char*>(q) + pk_f16_offset(Cin, Cout, T);
    uint4* f16dst = reinterpret_cast<uint4*>(f16base);
    float* trailer = reinterpret_cast<float*>(f16base + pk_main_bytes(Cin, Cout, T));
    float f16scale;
    if (row[11] >= 0) {
      const float a = *reinterpret_cast<const float*>(packed + row[11]);
      int kx = 0;
      if (a != 0.f) kx = 140 - (int)((__float_as_uint(a) >> 23) & 0xff);
      kx = kx < -126 ? -126 : (kx > 126 ? 126 : kx);
      f16scale = __uint_as_float((unsigned)(127 + kx) << 23);
      if (lb == 0 && tid == 0) {
        trailer[0] = __uint_as_float((unsigned)(127 - kx) << 23);
        trailer[1] = f16scale;
      }
    } else {
      f16scale = trailer[1];
    }
    const int n = T * 32;
    for (int it = tid; it < n; it += 256) {
      const int k = it & 7, g = (it >> 3) & 3, tap = it >> 5;
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = g * 8 + e < CW ? tile[k * RS + (g * 8 + e) * T + tap] : 0.f;
      const int cop = cop0 + k;
      const int ocb = cop >> 6, col = cop - ocb * 64;
      const size_t slot = (size_t)(tap * ICc + cc) * OCb + ocb;
      uint4 hi, mid, lo, fh, fm;
      pk_split8x3(f, hi, mid, lo);
      pk_split8h(f, f16scale, fh, fm);
      uint4* blk = q + slot * (size_t)(8 * NB);
      blk[(0 * 4 + g) * NB + col] = hi;
      blk[(1 * 4 + g) * NB + col] = mid;
      third[slot * (size_t)(4 * NB) + g * NB + col] = lo;
      uint4* fb = f16dst + slot * (size_t)(8 * NB);
      fb[(0 * 4 + g) * NB + col] = fh;
      fb[(1 * 4 + g) * NB + col] = fm;
    }
  }
  // ---- backward planes (roles swapped: K runs over the packed output channels): item (tap, c): the 8 channels of this
  // block are one 8-channel group of the contraction axis
  {
    const int ICcb = (Cout + 31) / 32, OCbb = (Cin + 63) / 64, NBb = pk_nb(Cin);
    uint4* q = reinterpret_cast<uint4*>(dstb + pk_prepared_offset(elems));
    uint4* third = q + (size_t)T * ICcb * OCbb * (size_t)(8 * NBb);
    const int ccb = cop0 >> 5, gb = (cop0 & 31) >> 3;
    const int n = T * CW;
    for (int it = tid; it < n; it += 256) {
      const int c = it % CW, tap = it / CW;
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = tile[e * RS + c * T + tap];
      const int ci = cc * 32 + c;
      const int ocb = ci >> 6, col = ci - ocb * 64;
      const size_t slot = (size_t)(tap * ICcb + ccb) * OCbb + ocb;
      uint4 hi, mid, lo;
      pk_split8x3(f, hi, mid, lo);
      uint4* blk = q + slot * (size_t)(8 * NBb);
      blk[(0 * 4 + gb) * NBb + col] = hi;
      blk[(1 * 4 + gb) * NBb + col] = mid;
      third[slot * (size_t)(4 * NBb) + gb * NBb + col] = lo;
    }
  }
}

int pack_weights_batched(const float* params, void* packed, const long long* table, int n_layers, int blocks,
                         hipStream_t s, bool has_unscratched, const int* fast_blocks, int n_fast_blocks) {
  // layer maxima for the fp16 planes: parallel slices into the caller-zeroed scratch words (table column 11 >= 0), or
  // one block per layer for rows without one
  // (64 slices: the SRGAN discriminator's 512 x 512 x 9 filters are 9.4 MB each -- with 16 blocks per layer the pass took
  //  39 us per step; blocks past a small layer's end exit at once)
  const int slices = 64;
  hipLaunchKernelGGL(k_pack_batched_amax, dim3((unsigned)slices, (unsigned)n_layers), dim3(256), 0, s, params,
                     static_cast<char*>(packed), table, slices);
  if (has_unscratched)
    hipLaunchKernelGGL(k_pack_batched_trailers, dim3((unsigned)n_layers), dim3(256), 0, s, params, static_cast<char*>(packed),
                       table);
  if (fast_blocks && n_fast_blocks > 0) {
    const size_t lds = (size_t)8 * (32 * 25 + 1) * sizeof(float);   // tile of the largest eligible filter (25 taps)
    hipLaunchKernelGGL(k_pack_fast, dim3((unsigned)n_fast_blocks), dim3(256), lds, s, params, static_cast<char*>(packed),
                       table, fast_blocks);
  }
  hipLaunchKernelGGL(k_pack_batched, dim3((unsigned)blocks, (unsigned)n_layers), dim3(256), 0, s, params,
                     static_cast<char*>(packed), table);
  return check_launch("pack_weights_batched");
}

template <int NT, bool VEC_ONLY = false, bool F16 = false>
__global__ __launch_bounds__(256, 2) void k_conv_bf3_rows(Bf3Params B) {
  extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
  const MfmaConvParams& P = B.P;
  // f16x3 (SRK_ALGO_MFMA_F16X3): the two planes are fp16(x 2^kx) and fp16 of its remainder
  float sx = 1.f, dsc = 1.f;
  if constexpr (F16) {
    const int kx = amax_scale_exp(amax_read(P.ep.x_amax));
    sx = exp2i(kx);
    dsc = exp2i(-kx) * B.w_descale[0];
  }
  uint2* hal = reinterpret_cast<uint2*>(smem4);   // [2 planes][NPIXp] x (4 bf16)
  uint4* wl = smem4 + B.NPIXp;                    // 2 planes * NPIXp * 8 B = NPIXp uint4
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kq = lane >> 4;
  int b = blockIdx.x;
  const int txi = b % P.tiles_x;
  b /= P.tiles_x;
  const int tyi = b % P.tiles_y;
  const int n = b / P.tiles_y;
  const int r0 = tyi * P.TH, c0 = txi * P.TW;
  const int ocbi = blockIdx.y;
  const int ocb = ocbi * 64;
  const int npx = P.TH * P.TW;
  const int NB = B.NB;
  const int wslot = 8 * NB;
  const int KS = B.ICc;           // K steps per kernel row
  const int Q = P.KHv * KS;       // total K steps
  const int npix = P.HH * P.HW;

  // stage the whole (<=4 channel) halo once: one thread per pixel; pad pixels are zeroed (they are
  // multiplied by zero filter taps and must be finite)
  {
    const int iyb = r0 * P.is + P.iy0, ixb = c0 * P.is + P.ix0;
    for (int hp = tid; hp < B.NPIXp; hp += 256) {
      float f[4] = {0.f, 0.f, 0.f, 0.f};
      if (hp < npix) {
        const int hy = hp / P.HW, hx = hp - hy * P.HW;
        const int iy = iyb + hy, ix = ixb + hx;
        if (iy >= 0 && iy < P.IH && ix >= 0 && ix < P.IW && !(SRK_KDBG(B.dbg) & 1)) {
          const size_t off = (((size_t)n * P.IH + iy) * P.IW + ix) * P.IC;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (e < P.IC) {
              float x = P.in_nchw ? P.in[(((size_t)n * P.IC + e) * P.IH + iy) * P.IW + ix] : P.in[off + e];
              if (P.mask_y) x = P.mask_y[off + e] > 0.f ? x : x * P.mask_slope;
              f[e] = x;
            }
        }
      }
      if constexpr (F16) {
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        f16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xs = f[e] * sx;
          const _Float16 hh = (_Float16)xs;
          h[e] = hh;
          l[e] = (_Float16)(xs - (float)hh);
        }
        hal[hp] = __builtin_bit_cast(uint2, h);
        hal[B.NPIXp + hp] = __builtin_bit_cast(uint2, l);
      } else {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        bf16x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const __bf16 hh = (__bf16)f[e];
          h[e] = hh;
          l[e] = (__bf16)(f[e] - (float)hh);
        }
        hal[hp] = __builtin_bit_cast(uint2, h);
        hal[B.NPIXp + hp] = __builtin_bit_cast(uint2, l);
      }
    }
  }
  int hp[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    int m = wave * 64 + mt * 16 + j;
    if (m >= npx) m = 0;
    const int r = m / P.TW, c = m - r * P.TW;
    hp[mt] = (r * P.is) * P.HW + c * P.is + 2 * kq;
  }
  f32x4 acc[4][NT];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool wave_live = wave * 64 < npx;
  const bool w0_ok = tid < wslot, w1_ok = tid + 256 < wslot;
  const int wlane = kq * NB + j;
  auto wsrc = [&](int q) -> const uint4* { return B.wq + ((size_t)q * B.OCb + ocbi) * (size_t)wslot; };
  if (Q > 0) {
    {
      const uint4* src = wsrc(0);
      const uint4 w0 = src[w0_ok ? tid : 0], w1 = src[w1_ok ? tid + 256 : 0];  // (unconditional loads: see k_conv_bf3)
      if (w0_ok) wl[tid] = w0;
      if (w1_ok) wl[tid + 256] = w1;
    }
    __syncthreads();
    int u = 0, ks = 0;
    for (int q = 0; q < Q; ++q) {
      uint4 wr0 = {0, 0, 0, 0}, wr1 = {0, 0, 0, 0};
      if (q + 1 < Q) {
        const uint4* src = wsrc(q + 1);
        if (w0_ok) wr0 = src[tid];
        if (w1_ok) wr1 = src[tid + 256];
      }
      if (wave_live && !(SRK_KDBG(B.dbg) & 4)) {
        const int toff = u * P.HW + ks * 8;
        const uint4* wb = wl + (q & 1) * wslot + wlane;
        uint4 ah[4], al[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const uint2* p = hal + hp[mt] + toff;
          const uint2 h0 = p[0], h1 = p[1];
          const uint2 l0 = p[B.NPIXp], l1 = p[B.NPIXp + 1];
          ah[mt] = make_uint4(h0.x, h0.y, h1.x, h1.y);
          al[mt] = make_uint4(l0.x, l0.y, l1.x, l1.y);
        }
        uint4 bh[NT], bl[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          bh[nt] = wb[nt * 16];
          bl[nt] = wb[4 * NB + nt * 16];
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma16x<F16>(al[mt], bh[nt], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma16x<F16>(ah[mt], bl[nt], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma16x<F16>(ah[mt], bh[nt], acc[mt][nt]);
      }
      if (q + 1 < Q) {
        uint4* wn = wl + ((q + 1) & 1) * wslot;
        if (w0_ok) wn[tid] = wr0;
        if (w1_ok) wn[tid + 256] = wr1;
      }
      __syncthreads();
      if (++ks == KS) {
        ks = 0;
        ++u;
      }
    }
  }
  if constexpr (F16) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] *= dsc;
  }
  bf3_epilogue<NT, VEC_ONLY>(P, reinterpret_cast<float*>(smem4), acc, n, r0, c0, ocb, wave, lane, SRK_KDBG(B.dbg));
}

// ---------------------------------------------------------------------------------------------
// Host
// ---------------------------------------------------------------------------------------------
template <int NT>
static void bf3_launch_vec(const Bf3Params& B, dim3 grid, size_t lds, hipStream_t s) {
  static LdsLimit lim;
  lim.ensure(reinterpret_cast<const void*>(&k_conv_bf3<NT, 4, true>), lds);
  if (SRK_KDBG(B.dbg) & 32) {
    int nb = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&k_conv_bf3<NT, 4, true>), 256,
                                                       lds);
    fprintf(stderr, "[srk] k_conv_bf3<%d,4,vec>: lds %zu B, grid %u x %u, occupancy %d blocks/CU, tile %dx%d halo %dx%d\n", NT,
            lds, grid.x, grid.y, nb, B.P.TH, B.P.TW, B.P.HH, B.P.HW);
  }
  note_kernel("k_conv_bf3<%d,4,vec>", NT);
  hipLaunchKernelGGL((k_conv_bf3<NT, 4, true>), grid, dim3(256), lds, s, B);
}

template <int NT, int NW>
static void bf3_launch(const Bf3Params& B, dim3 grid, size_t lds, hipStream_t s) {
  static LdsLimit lim;
  lim.ensure(reinterpret_cast<const void*>(&k_conv_bf3<NT, NW>), lds);
  if (SRK_KDBG(B.dbg) & 32) {
    int nb = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&k_conv_bf3<NT, NW>), 64 * NW,
                                                       lds);
    fprintf(stderr, "[srk] k_conv_bf3<%d,%d>: lds %zu B, grid %u x %u, occupancy %d blocks/CU, tile %dx%d halo %dx%d\n", NT,
            NW, lds, grid.x, grid.y, nb, B.P.TH, B.P.TW, B.P.HH, B.P.HW);
  }
  note_kernel("k_conv_bf3<%d,%d>", NT, NW);
  hipLaunchKernelGGL((k_conv_bf3<NT, NW>), grid, dim3(64 * NW), lds, s, B);
}

bool conv_bf3_gather_supported(const GatherConv& g, const Epi& ep) {
  (void)ep;
  if (g.OC < 8) return false;                        // <= 4: direct kernel; 5..7: fp32 kernels
  if (g.IC < 8 && !(g.IC <= 4 && !g.trans)) return false;  // small IC: row-packed variant, CONV gathers only
  if (g.KH * g.KW > 32 * 32) return false;
  if ((long)g.N * g.OH * g.OW > (1L << 30)) return false;
  if ((long)g.IH * g.IW * g.IC >= (1L << 30)) return false;  // 32-bit in-image offsets in the staging loops
  return true;
}

template <int NT, bool VEC_ONLY, bool F16 = false>
static void bf3_launch_rows_v(const Bf3Params& B, dim3 grid, size_t lds, hipStream_t s) {
  static LdsLimit lim;
  lim.ensure(reinterpret_cast<const void*>(&k_conv_bf3_rows<NT, VEC_ONLY, F16>), lds);
  note_kernel("k_conv_bf3_rows<%d%s%s>", NT, F16 ? ",f16" : "", VEC_ONLY ? "" : ",scalar");
  note_amax_written(VEC_ONLY && B.P.ep.y_amax != nullptr);   // (the scalar-store variant keeps no maximum)
  hipLaunchKernelGGL((k_conv_bf3_rows<NT, VEC_ONLY, F16>), grid, dim3(256), lds, s, B);
}
template <int NT>
static void bf3_launch_rows(const Bf3Params& B, dim3 grid, size_t lds, hipStream_t s) {
  if (B.w_descale)   // f16x3 (the host checked the vector-store condition: conv_bf3_rows_f16_supported)
    bf3_launch_rows_v<NT, true, true>(B, grid, lds, s);
  else if (B.P.OC % 16 == 0 && epi_all_vector(B.P))
    bf3_launch_rows_v<NT, true>(B, grid, lds, s);
  else
    bf3_launch_rows_v<NT, false>(B, grid, lds, s);
}

static int bf3_launch_rows_phase(MfmaConvParams P, const uint4* wq, hipStream_t s, const float* w_descale = nullptr) {
  Bf3Params B{};
  B.w_descale = w_descale;
  const int NT = P.OC >= 64 ? 4 : (P.OC + 15) / 16;
  B.NB = NT * 16;
  B.ICc = (P.KWv + 7) / 8;  // K steps per kernel row
  B.OCb = (P.OC + 63) / 64;
  B.wq = wq;
  const int wbytes = 2 * 8 * B.NB * 16;
  TilePick best{};
  if (!pick_tile(256, P.PH, P.PW, P.is, P.KHv > 0 ? P.KHv : 1, P.KWv > 0 ? P.KWv : 1, 4,
                 (kLdsBudgetBytes - wbytes) / 4 - 32 * 4, best)) {
    set_error("conv_bf3_rows: no tile fits LDS");
    return SRK_ERR_UNSUPPORTED;
  }
  P.TH = best.TH; P.TW = best.TW; P.tiles_y = best.tiles_y; P.tiles_x = best.tiles_x; P.HH = best.HH; P.HW = best.HW;
  B.NPIXp = (best.HH * best.HW + 16 + 15) & ~15;  // +16: the last row's padded kw slots read past the halo
  B.P = P;
  B.dbg = SRK_EXP_INT("SRK_ROWS_DBG", 0);   // ablation: 1 no halo loads, 2 no epilogue, 4 no MFMAs, 64 no global stores
  size_t lds = (size_t)B.NPIXp * 16 + wbytes;
  const size_t epi_bytes = (size_t)4 * 32 * BF3_EPI_STRIDE * sizeof(float);
  if (lds < epi_bytes) lds = epi_bytes;
  dim3 grid((unsigned)((size_t)P.tiles_x * P.tiles_y * P.N), B.OCb);
  switch (NT) {
    case 1: bf3_launch_rows<1>(B, grid, lds, s); break;
    case 2: bf3_launch_rows<2>(B, grid, lds, s); break;
    case 3: bf3_launch_rows<3>(B, grid, lds, s); break;
    default: bf3_launch_rows<4>(B, grid, lds, s); break;
  }
  return check_launch("conv_bf3_rows");
}

template <int NW>
static int bf3_launch_phase_nw(MfmaConvParams P, Bf3Params B, int NT, int dbg, hipStream_t s) {
  const int wbytes = 2 * 8 * B.NB * 16;
  const int maxpix = 64 * NW;
  // LDS share that lets `blocks` blocks of this size be co-resident on a CU (160 KiB)
  const int budget = NW == 4 ? kLdsBudgetBytes : (NW == 2 ? 39 * 1024 : 31 * 1024);
  // halo pixel = 128 B (2 planes x 4 groups x 16 B); fall back to one block per CU for huge halos
  TilePick best{};
  bool ok = pick_tile(maxpix, P.PH, P.PW, P.is, P.KHv > 0 ? P.KHv : 1, P.KWv > 0 ? P.KWv : 1, 32,
                      (budget - wbytes) / 4 - 16 * 32, best);
  if (!ok || best.eff < 0.6) {
    TilePick big{};
    if (pick_tile(maxpix, P.PH, P.PW, P.is, P.KHv > 0 ? P.KHv : 1, P.KWv > 0 ? P.KWv : 1, 32,
                  (156 * 1024 - wbytes) / 4 - 16 * 32, big) &&
        (!ok || big.eff > best.eff * 1.2)) {
      best = big;
      ok = true;
    }
  }
  if (!ok) {
    set_error("conv_bf3: no tile fits LDS");
    return SRK_ERR_UNSUPPORTED;
  }
  P.TH = best.TH; P.TW = best.TW; P.tiles_y = best.tiles_y; P.tiles_x = best.tiles_x; P.HH = best.HH; P.HW = best.HW;
  B.NPIXp = (best.HH * best.HW + 15) & ~15;  // multiple of 16: the kq lane groups of a ds_read_b128 interleave conflict-free
  B.P = P;
  B.dbg = dbg;
  size_t lds = (size_t)8 * B.NPIXp * 16 + wbytes;
  const size_t epi_bytes = (size_t)NW * 32 * BF3_EPI_STRIDE * sizeof(float);
  if (lds < epi_bytes) lds = epi_bytes;
  if (dbg & 64) lds = 100 * 1024;  // experiment: force 1 block per CU
  dim3 grid((unsigned)((size_t)P.tiles_x * P.tiles_y * P.N), B.OCb);
  if (NW == 4 && NT >= 2 && P.OC % 16 == 0 && epi_all_vector(P)) {  // full 16-channel tiles, vector stores everywhere
    switch (NT) {
      case 2: bf3_launch_vec<2>(B, grid, lds, s); break;
      case 3: bf3_launch_vec<3>(B, grid, lds, s); break;
      default: bf3_launch_vec<4>(B, grid, lds, s); break;
    }
    return check_launch("conv_bf3");
  }
  switch (NT) {
    case 1: bf3_launch<1, NW>(B, grid, lds, s); break;
    case 2: bf3_launch<2, NW>(B, grid, lds, s); break;
    case 3: bf3_launch<3, NW>(B, grid, lds, s); break;
    default: bf3_launch<4, NW>(B, grid, lds, s); break;
  }
  return check_launch("conv_bf3");
}

static int bf3_launch_phase(MfmaConvParams P, const uint4* wq, hipStream_t s) {
  if (P.IC <= 4) return bf3_launch_rows_phase(P, wq, s);
  Bf3Params B{};
  const int NT = P.OC >= 64 ? 4 : (P.OC + 15) / 16;
  B.NB = NT * 16;
  B.ICc = (P.IC + 31) / 32;
  B.OCb = (P.OC + 63) / 64;
  B.wq = wq;
  const int dbg = SRK_EXP_INT("SRK_DBG", 0);
  const int nw = SRK_EXP_INT("SRK_BF3_WAVES", 0);  // waves per block: 4 (256-px tiles), 2 or 1 -- more, smaller blocks per CU; 0 = automatic
  int use = nw;
  if (nw <= 0) {
    // small problems (strong-scaled shards): fewer pixels than 2 resident 256-pixel tiles per CU ->
    // smaller blocks so every CU gets work (measured: 256-px tiles win whenever the chip is full)
    const long px = (long)P.N * P.PH * P.PW * B.OCb;
    use = px >= 256L * 2 * kNumCU ? 4 : (px >= 128L * 2 * kNumCU ? 2 : 1);
  }
  if (use == 1) return bf3_launch_phase_nw<1>(P, B, NT, dbg, s);
  if (use == 2) return bf3_launch_phase_nw<2>(P, B, NT, dbg, s);
  return bf3_launch_phase_nw<4>(P, B, NT, dbg, s);
}

int conv_bf3_gather(const GatherConv& g, const float* in, const float* wp, float* out, const Epi& ep,
                    const float* mask_y, float mask_slope, hipStream_t s) {
  const size_t elems = (size_t)g.KH * g.KW * g.IC * g.OC;
  const uint4* wq = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(wp) + bf3_prepared_offset(elems));
  return for_each_phase(g, in, wp, out, ep, mask_y, mask_slope,
                        [&](const MfmaConvParams& P) { return bf3_launch_phase(P, wq, s); });
}

// f16x3 for the row-packed first layers (Cin <= 4 CONV gathers): the fp16 row planes + trailer of the forward buffer
bool conv_bf3_rows_f16_supported(const GatherConv& g, const Epi& ep, const float* out) {
  return g.IC <= 4 && !g.trans && g.OC >= 8 && g.OC % 16 == 0 && conv_bf3_gather_supported(g, ep) && !g.in_ps_r &&
         conv_epi_all_vector(g.OC, ep, out);
}
int conv_bf3_rows_f16_gather(const GatherConv& g, const float* in, const float* wp, float* out, const Epi& ep,
                             hipStream_t s) {
  const size_t elems = (size_t)g.KH * g.KW * g.IC * g.OC;
  const char* fsec = reinterpret_cast<const char*>(wp) + bf3_prepared_offset(elems) + f16_section_offset(g.IC, g.OC, g.KH * g.KW);
  const uint4* wq = reinterpret_cast<const uint4*>(fsec);
  const float* trailer = reinterpret_cast<const float*>(fsec + bf3_main_bytes(g.IC, g.OC, g.KH * g.KW));
  return for_each_phase(g, in, wp, out, ep, nullptr, 0.f,
                        [&](const MfmaConvParams& P) { return bf3_launch_rows_phase(P, wq, s, trailer); });
}

}  // namespace srk
