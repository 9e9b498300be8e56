// Weight gradient of stride-1 Conv2d on the bf16 matrix cores (3-term split, fp32 accumulate).
//
//   dW[tap][ci][co] = sum over output pixels q of  X[q - pad + tap][ci] * dY[q][co]
//
// GEMM view per tap: M = ci, N = co, K = pixels.  v_mfma_f32_16x16x32_bf16 wants, per lane, 8
// consecutive K values of one row (A: channel ci of X) / column (B: channel co of dY): the NHWC
// fp32 tiles are therefore staged TRANSPOSED in LDS as bf16 planes [hi|lo][channel][pixel], a K
// step being 4 "octets" (8 consecutive pixels of one tile row, 16-byte aligned in LDS).
//   * the tap's row shift u is a row offset in the X halo plane; the column shift v (0..2) would
//     misalign the 16-byte fragment by v*2 bytes, so each lane reads the aligned octet plus the
//     next two pixels (ds_read_b128 + ds_read_b32) ONCE per (u, plane) and derives the three
//     shifted fragments in registers (v_alignbit_b32 by 16 bits for v = 1, a register rename for
//     v = 2): 12 + 2*NTW LDS reads and 24 VALU ops feed 27*NTW MFMAs per K step;
//   * a block owns a chunk of CIT*16 input channels x COW*NTW*16 output channels and ALL <= 9
//     taps; wave (cit, cow) keeps its 9 x NTW accumulator tiles in registers across all pixel
//     tiles the persistent block walks (split-K over blocks), then writes one partial slab; the
//     deterministic slab reduction / bias-gradient finish are shared with the fp32 kernel;
//   * dY is masked with the ReLU / LeakyReLU gradient on the fly and its column sums (bias
//     gradient partials) are accumulated in fp32 registers while it is staged.
// Arithmetic: x = h + m (bf16 each), products m*h + h*m + h*h -> ~5e-6 relative on dW.
// Anything else (stride > 1, ConvTranspose2d, kernels larger than 3x3, Cin < 8) runs the exact fp32
// kernels of conv_wgrad_mfma.hip.
#include "srk_common.h"
#include "conv_problem.h"
#include <stdlib.h>

namespace srk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// conv_wgrad_mfma.hip
int conv_wgrad_reduce_launch(const float* ws, float* dw, int G, int Cout, int Cin, int KH, int KW, int transposed,
                             float beta, const float* bias_partial, float* db, int bias_cout, int out_ps_r,
                             hipStream_t s);
bool wgrad_reduce_deferring();
int wgrad_reduce_submit(bool wide, const float* ws, float* dw, int G, int Cout, int Cin, int KH, int KW, int transposed,
                        float beta, const float* bias_partial, float* db, int bias_cout, int out_ps_r, hipStream_t s);

constexpr int WB_MAXOCT = 64;  // octets per tile (<= 512 pixels)
constexpr int WB_SST = 512;    // SPEC: staging threads (8 waves next to the 4 working waves; 4 stager waves: 0.157 -> 0.20 ms on the VDSR layer)
constexpr int WB_PIT = 1024 / WB_SST;  // SPEC stagers: register batches per tensor and tile when prefetching one tile ahead
// Staging-thread -> (channel group q, first pixel pair) map.  A wave covers 4 channel groups x 16 consecutive pixel pairs:
// its transposing 4-byte LDS stores then fall on 4 x 16 distinct banks (channel-group stride = 16 * odd dwords, pixel pairs =
// consecutive dwords).  The round-1 map (q fastest: 16 groups x 4 pairs per wave for a 64-channel tensor) put 4 lanes on
// every bank it touched.  QN = channel groups of the staged tensor, NSTT = staging threads (a multiple of 16 * QN).
#ifndef WB_CF_MAP
#define WB_CF_MAP 1
#endif
template <int QN, int NSTT>
__device__ __forceinline__ void wb_item(int tid, int& q, int& pp0) {
  if (WB_CF_MAP) {
    constexpr int QB = QN / 4;  // 4-group blocks
    const int w = tid >> 6, l = tid & 63;
    q = (w % QB) * 4 + (l >> 4);
    pp0 = (w / QB) * 16 + (l & 15);
  } else {
    q = tid % QN;
    pp0 = tid / QN;
  }
}
// WB_PIPE: software-pipelined K loop of the 3x3 kernels (see kloop); 0 = the round-4 loop (tools/build_variant.sh A/B)
#ifndef WB_PIPE
#define WB_PIPE 0
#endif
#ifndef WB_PRIO
#define WB_PRIO 0
#endif
// ablation bits of k_wgrad_tr's inner loop (16 no fragment reads, 64 no column shifts): compile-time only (-DSRK_KDBG_CONST=..),
// also in the experiments build -- a run-time test inside the phases changes their schedule (K step 1450 -> 3200 clocks)
#ifdef SRK_KDBG_CONST
#define WT_ABL (SRK_KDBG_CONST)
#else
#define WT_ABL 0
#endif
#ifndef WT_SGB
#define WT_SGB 1
#endif
constexpr int WB_IT = 1;       // pixel pairs per thread loaded together while staging (4 measured: no gain plain, spills in the specialised variant)

struct WgBfParams {
  const float* x;
  const float* dy;
  const float* mask_y;
  float mask_slope;
  float* ws;            // [G][T][Cin][Cout]
  float* bias_partial;  // [G][Cout] or NULL
  int N, Cin, Cout;
  int XH, XW, YH, YW;
  int KH, KW, pad;
  int TH, TW, TWo, tiles_y, tiles_x, HH, HWp;
  int CS, DS;  // per-channel plane strides (bf16 elements) of the X halo / dY tile
  int ntiles, G, nks;
  int vec_x, vec_y;
  int dbg;  // ablation (SRK_DBG): 2 skip the staging, 4 skip the K loop
  int dy_ps_r, dy_ps_C;  // dY handed over pixel-shuffled [N, YH*r, YW*r, Cout/r^2]: un-shuffled while staging
  int prefetch;          // SPEC: a tile's loads fit the stagers' register batch (WB_PIT x 512 items): load one tile ahead
  long long* prof;       // experiments build only (srk_debug_wgrad_prof): per block 16 int64 -- clock64() sums of the first stager
                         // wave {commit, issue, barrier wait, tiles} and of worker wave 0 {K loop, barrier wait, tiles, K steps}
  int ring;              // SPEC + prefetch: X halo rows live in a ring of 2 * HH rows shared by vertically adjacent tiles
                         // (a block walks a CONTIGUOUS range of tiles, rows fastest): a tile below its predecessor loads
                         // only its TH new rows instead of all TH + KH - 1 (2-row tiles: the X read halves).  CS is then
                         // the plane stride of the ring, and the dY tiles keep their two buffer sets behind it.
  int XP, XPL, YPL;      // k_wgrad_tr: pixels per ring row, bytes per X plane / dY plane of its pixel-major LDS image
};

// Grouped launch: the weight gradients of up to WB_MAXGROUP convolutions that share ONE geometry (the 33 body convs
// of EDSR, edsr.py:37-45; the 18 of VDSR, vdsr.py:17-24) in one launch.  Only the tensors differ per layer; their
// pointers travel by value in the kernel arguments (no device-side table, so the call is hipGraph-capturable as is).
// Block x = layer * G + g: layer `layer` has G split-K partial slabs, slab / bias-partial index = blockIdx.x.
constexpr int WB_MAXGROUP = 40;
struct WgLayer {
  const float* x;
  const float* dy;
  const float* mask_y;
  float mask_slope;
  int pad_;
};
struct WgGroup {
  WgLayer L[WB_MAXGROUP];
};
struct WgNoGroup {
  int unused;
};
template <bool GRP>
struct WgGroupArg {
  typedef WgNoGroup type;
};
template <>
struct WgGroupArg<true> {
  typedef WgGroup type;
};
struct WgOut {
  float* dw;
  float* db;
};
struct WgGroupOut {
  WgOut L[WB_MAXGROUP];
};

#ifdef SRK_EXPERIMENTS
#define WB_CLK() clock64()
static long long* g_wb_prof = nullptr;
#else
#define WB_CLK() 0ll
#endif

__device__ __forceinline__ f32x4 wb_mfma(const uint4& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0,
                                                 0);
}

// ablation helpers: a value the compiler must materialise / may not assume anything about
__device__ __forceinline__ void wb_keep(const uint4& a) { asm volatile("" ::"v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w)); }
__device__ __forceinline__ void wb_touch(uint4& a, int dep) {
  asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w) : "v"(dep));
}

__device__ __forceinline__ unsigned short wb_bits(__bf16 v) { return __builtin_bit_cast(unsigned short, v); }

// two horizontally adjacent pixels x 4 channels (fp32) -> 4 channels x {hi, lo} dwords (pixel pair packed):
// one v_cvt_pk_bf16_f32 per (channel, plane) produces the packed pair directly
typedef __bf16 wb_bf16x2 __attribute__((ext_vector_type(2)));
typedef float wb_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void wb_split_pair(const f32x4& p0, const f32x4& p1, unsigned (&hi)[4], unsigned (&lo)[4]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const wb_f32x2 v = {p0[c], p1[c]};
    const wb_bf16x2 h = __builtin_convertvector(v, wb_bf16x2);
    const unsigned hb = __builtin_bit_cast(unsigned, h);
    const wb_f32x2 hf = {__builtin_bit_cast(float, hb << 16), __builtin_bit_cast(float, hb & 0xffff0000u)};
    const wb_bf16x2 l = __builtin_convertvector(v - hf, wb_bf16x2);
    hi[c] = hb;
    lo[c] = __builtin_bit_cast(unsigned, l);
  }
}

// Buffer descriptor over [base, base + bytes) built from wave-uniform values (the readfirstlane makes the uniformity
// provable: no waterfall loop around the loads), and a 16-byte load through it: an offset outside the range reads zero.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wb_rsrc(const float* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  void* p = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(p, (short)0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ f32x4 wb_bload(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  typedef unsigned wb_u32x4 __attribute__((ext_vector_type(4)));
  const wb_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
  return __builtin_bit_cast(f32x4, v);
}

// Exact floor(m / d) for 0 <= m < 2^20 / d with one multiply (magic = ceil(2^20 / d)): pixel-pair counters of a tile
__host__ __device__ __forceinline__ unsigned wb_magic20(int d) { return (unsigned)(((1u << 20) + d - 1) / d); }
__device__ __forceinline__ int wb_div20(int m, unsigned magic) { return (int)(((unsigned)m * magic) >> 20); }

// Load 4 channels at element offset `off` from the (wave-uniform) image base `src`; `ok` = inside the image and
// channel range.  `mask` (same offsets) applies the ReLU-family gradient.  vec = 16-byte path legal.
__device__ __forceinline__ f32x4 wb_load4(const float* __restrict__ src, const float* __restrict__ mask, float mslope,
                                          unsigned off, bool ok, int nch, int vec) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (ok) {
    if (vec && nch >= 4) {
      v = *reinterpret_cast<const f32x4*>(src + off);
      if (mask) {
        const f32x4 m = *reinterpret_cast<const f32x4*>(mask + off);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = m[e] > 0.f ? v[e] : v[e] * mslope;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (e < nch) {
          float t = src[off + e];
          if (mask) t = mask[off + e] > 0.f ? t : t * mslope;
          v[e] = t;
        }
    }
  }
  return v;
}

// CIT: 16-channel input tiles per block (one per wave row), COW: output-channel wave columns,
// NTW: 16-channel output tiles per wave.  CIT * COW = 4 waves.
// SPEC (wave-specialised, 768 threads, one block per CU): waves 4-11 stage tile t+1 into the second LDS buffer set
// (global loads, masking, bf16 split, transposed stores) WHILE waves 0-3 run the K loop of tile t; one barrier per tile.
// The per-tile kernel's two phases are about equally long and co-resident blocks run them in lockstep; here they
// overlap by construction, and one block per CU halves the number of partial slabs.
// K33: 3x3 kernels (every body layer) get a K loop without the runtime tap tests: the fragment reads of a whole K step are
// issued together and the 27 * NTW MFMAs follow without control flow in between (the generic loop reads each row shift's
// fragments right in front of its MFMAs, behind a branch: two exposed LDS round trips per 18 MFMAs).
template <int CIT, int COW, int NTW, bool SPEC, bool GRP, bool K33 = false>
__global__ __launch_bounds__(SPEC ? 64 * CIT * COW + WB_SST : 64 * CIT * COW,
                                 SPEC ? (64 * CIT * COW + WB_SST) / 256 : (CIT * COW == 4 ? 2 : 1)) void k_wgrad_bf(WgBfParams P,
                                                                             typename WgGroupArg<GRP>::type GR) {
  constexpr int CIB = CIT * 16, COB = COW * NTW * 16;
  // NWV working waves: 4.  (Round 4, tools/wgrad_prof.py: the worker wave's K loop is 97 % of its time and ~2080 clocks per K
  // step for ~920 clocks of MFMA issue, while the stagers wait 30 - 49 % of a tile at the barrier -- the workers pace the kernel.
  // EIGHT working waves (<2, 4, 1>: two per SIMD taking turns at the matrix pipe, 104 VGPRs, -DSRK_EXPERIMENTS + SRK_WG_W8=1)
  // were measured SLOWER: VDSR layer 0.154 -> 0.184 ms, step 6.00 -> 6.62 ms, EDSR 6.07 -> 6.42 ms.  They read 1.75x the LDS
  // bytes per K step (every X fragment once per output-channel column): the shared LDS pipe -- fragment reads against the
  // stagers' 4-byte transposing writes -- is what the K loop waits for, not latency a second wave could hide.)
  constexpr int NWV = CIT * COW, WTHR = 64 * NWV;
  constexpr int NTHR = SPEC ? WTHR + WB_SST : WTHR;
  constexpr int NST = SPEC ? WB_SST : WTHR;  // staging threads (SPEC: 8 stager waves keep the staging rate of two 256-thread blocks)
  extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
  __shared__ int oct_x[WB_MAXOCT], oct_y[WB_MAXOCT], oct_r[WB_MAXOCT], oct_c[WB_MAXOCT], oct_pk[WB_MAXOCT];
  __shared__ float bred[NST][4];
  const size_t buf_shorts = (size_t)2 * CIB * P.CS + (size_t)2 * COB * P.DS;  // one buffer set: [2][CIB][CS] + [2][COB][DS]
  // ring mode: [2][CIB][CS] (the ring, once) followed by two dY sets [2][COB][DS]
  const bool ring = SPEC && P.ring;
  const size_t x_shorts = (size_t)2 * CIB * P.CS, yset_shorts = (size_t)2 * COB * P.DS;
  const int R2 = 2 * P.HH;
  auto ys_of = [&](int bsel) -> unsigned short* {
    return ring ? smem16 + x_shorts + (size_t)bsel * yset_shorts : smem16 + (size_t)bsel * buf_shorts + x_shorts;
  };
  auto xs_of = [&](int bsel) -> unsigned short* { return ring ? smem16 : smem16 + (size_t)bsel * buf_shorts; };
  // ring base of the next tile: a tile that continues its predecessor's column moves on by TH rows, anything else
  // (first tile of the block, top of a column) takes the other half of the ring
  auto ring_next = [&](int prev, bool first) {
    const int b = prev + (first ? P.HH : P.TH);
    return b >= R2 ? b - R2 : b;
  };
  const int tid0 = threadIdx.x, lane = tid0 & 63, wave = tid0 >> 6;
  const bool stager = !SPEC || wave >= NWV;    // stages tiles
  const bool worker = !SPEC || wave < NWV;     // runs the K loop, owns accumulators
  const int tid = SPEC ? (wave >= NWV ? tid0 - WTHR : tid0) : tid0;  // index among the staging threads (stagers) / working threads
  const int i = lane & 15, kq = lane >> 4;
  const int cit = (wave % NWV) % CIT, cow = (wave % NWV) / CIT;
  // (slab index, input-channel chunk) of this block.  The gy = 2 blocks that share a slab index read the SAME dY / mask
  // tiles in the same order: place them on one XCD (workgroups are dealt round-robin over the 8 XCDs in dispatch order)
  // so that the second reader hits that XCD's L2 instead of HBM — the weight gradient moves ~3.5 TB/s, it is bound by
  // bytes, and dY + mask are two thirds of them.
  int bxl = blockIdx.x, byl = blockIdx.y;
  if (gridDim.y == 2 && (gridDim.x & 7) == 0) {
    const int lin = (int)(blockIdx.y * gridDim.x + blockIdx.x);
    bxl = (lin >> 4) * 8 + (lin & 7);
    byl = (lin >> 3) & 1;
  }
  const int cib = byl * CIB, cob = blockIdx.z * COB;
  const int noct = P.TH * P.TWo;
  const bool want_bias = P.bias_partial != nullptr && byl == 0;
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  // this block's layer (grouped launch) and its split-K index inside the layer; tensors of that layer
  int bx = bxl;
  const float* __restrict__ Lx = P.x;
  const float* __restrict__ Ldy = P.dy;
  const float* __restrict__ Lmask = P.mask_y;
  float Lslope = P.mask_slope;
  if constexpr (GRP) {
    const int layer = bxl / P.G;
    bx = bxl - layer * P.G;
    Lx = GR.L[layer].x;
    Ldy = GR.L[layer].dy;
    Lmask = GR.L[layer].mask_y;
    Lslope = GR.L[layer].mask_slope;
  }

  for (int o = tid0; o < P.nks * 4; o += NTHR) {
    if (o < noct) {
      const int orow = o / P.TWo, oc = o - orow * P.TWo;
      oct_x[o] = orow * P.HWp + oc * 8;
      oct_y[o] = o * 8;
      oct_r[o] = orow;
      oct_c[o] = oc * 8;
      oct_pk[o] = (o * 8) | ((oc * 8) << 12) | (orow << 20);
    } else {  // K padding: multiply by the zero octet appended to every dY plane
      oct_x[o] = 0;
      oct_y[o] = P.TH * P.TW;
      oct_r[o] = 0;
      oct_c[o] = 0;
      oct_pk[o] = P.TH * P.TW;
    }
  }
  for (int e = tid0; e < (SPEC ? 2 : 1) * 2 * COB * 4; e += NTHR) {  // zero octets (never overwritten by the staging)
    const int bsel = e / (2 * COB * 4), e2 = e - bsel * (2 * COB * 4);
    const int pc = e2 >> 2, w = e2 & 3;
    unsigned short* ysb = ys_of(bsel);
    reinterpret_cast<unsigned*>(ysb + (size_t)pc * P.DS + P.TH * P.TW)[w] = 0u;
  }

  f32x4 acc[3][3][NTW];
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int v = 0; v < 3; ++v)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) acc[u][v][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int tw2 = P.TW >> 1;

  // ---- stage one tile into buffer set `bsel` (256 staging threads)
  auto stage = [&](int tile, int bsel) {
    unsigned short* xs = smem16 + bsel * buf_shorts;          // [2][CIB][CS]
    unsigned short* ys = xs + (size_t)2 * CIB * P.CS;          // [2][COB][DS]
    int b = tile;
    const int txi = b % P.tiles_x;
    b /= P.tiles_x;
    const int tyi = b % P.tiles_y;
    const int n = b / P.tiles_y;
    const int r0 = tyi * P.TH, c0 = txi * P.TW;
    if (!(SRK_KDBG(P.dbg) & 2)) {  // X halo: rows [r0-pad, +HH), cols [c0-pad, +TW+KW-1), channels [cib, cib+CIB) -> planes [ci][hy][hx]
      // A thread's channel group q is fixed (256 % QN == 0): it walks pixel pairs pp, pp + 256/QN, ...
      constexpr int QN = CIB / 4, PSTEP = NST / QN;
      // only the TW + KW - 1 columns the fragments can touch are loaded (the plane row stride HWp = TW + 8 is for
      // the 16-byte alignment of the octets)
      const int need2 = (P.TW + P.KW) >> 1;
      const unsigned need2_magic = wb_magic20(need2);
      const int npairs = P.HH * need2;
      const int by0 = r0 - P.pad, bx0 = c0 - P.pad;
      int q, ppb;
      wb_item<QN, NST>(tid, q, ppb);
      const int ch = cib + q * 4;
      const int nch = P.Cin - ch;  // channels of this group that exist (<= 0: none)
      const float* __restrict__ xb = Lx + (size_t)n * P.XH * P.XW * P.Cin;  // wave-uniform image base
      unsigned short* xq = xs + (size_t)(q * 4) * P.CS;
      // batches of WB_IT pixel pairs per thread: every global load of a batch is issued before the first conversion
      // (one exposed load latency per batch instead of one per pair)
      for (int pp0 = ppb; pp0 < npairs; pp0 += PSTEP * WB_IT) {
        f32x4 p0[WB_IT], p1[WB_IT];
#pragma unroll
        for (int k = 0; k < WB_IT; ++k) {
          const int pp = pp0 + k * PSTEP;
          const int hy = wb_div20(pp, need2_magic), hx = (pp - hy * need2) * 2;
          const int iy = by0 + hy, ix = bx0 + hx;
          const bool rowok = pp < npairs && (unsigned)iy < (unsigned)P.XH && nch > 0;
          const unsigned off = (unsigned)(iy * P.XW + ix) * (unsigned)P.Cin + (unsigned)ch;
          p0[k] = wb_load4(xb, nullptr, 0.f, off, rowok && (unsigned)ix < (unsigned)P.XW, nch, P.vec_x);
          p1[k] = wb_load4(xb, nullptr, 0.f, off + (unsigned)P.Cin, rowok && (unsigned)(ix + 1) < (unsigned)P.XW, nch,
                           P.vec_x);
        }
#pragma unroll
        for (int k = 0; k < WB_IT; ++k) {
          const int pp = pp0 + k * PSTEP;
          if (pp < npairs) {
            const int hy = wb_div20(pp, need2_magic), hx = (pp - hy * need2) * 2;
            unsigned hi[4], lo[4];
            wb_split_pair(p0[k], p1[k], hi, lo);
            unsigned short* dst = xq + hy * P.HWp + hx;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              *reinterpret_cast<unsigned*>(dst + (size_t)c * P.CS) = hi[c];
              *reinterpret_cast<unsigned*>(dst + (size_t)(CIB + c) * P.CS) = lo[c];
            }
          }
        }
      }
    }
    if (!(SRK_KDBG(P.dbg) & 2)) {  // dY tile: rows [r0, +TH), cols [c0, +TW), channels [cob, cob+COB) -> planes [co][r][c] (masked)
      constexpr int QN = COB / 4, PSTEP = NST / QN;
      const unsigned tw2_magic = wb_magic20(tw2);
      const int npairs = P.TH * tw2;
      int q, ppb;
      wb_item<QN, NST>(tid, q, ppb);
      const int ch = cob + q * 4;
      const int nch = P.Cout - ch;
      // element strides of (row, col) and the channel-group offset inside a pixel; pixel-shuffled dY: packed channel
      // (i, j, c) of pixel (y, x) lives at (y*r + i, x*r + j, c) of the [N, YH*r, YW*r, C] tensor
      unsigned srow, scol, koff;
      if (P.dy_ps_r > 1) {
        const int r = P.dy_ps_r, C = P.dy_ps_C;
        const int chc = ch < P.Cout ? ch : 0;
        const int qq = chc / C, c = chc - qq * C;
        const int i = qq / r, j = qq - i * r;
        scol = (unsigned)(r * C);
        srow = (unsigned)(r * P.YW) * scol;
        koff = (unsigned)(i * P.YW) * scol + (unsigned)(j * C + c);
      } else {
        scol = (unsigned)P.Cout;
        srow = (unsigned)P.YW * scol;
        koff = (unsigned)ch;
      }
      const size_t img = (size_t)n * P.YH * srow;
      const float* __restrict__ yb = Ldy + img;
      const float* __restrict__ mb = Lmask ? Lmask + img : nullptr;
      unsigned short* yq = ys + (size_t)(q * 4) * P.DS;
      for (int pp0 = ppb; pp0 < npairs; pp0 += PSTEP * WB_IT) {
        f32x4 p0[WB_IT], p1[WB_IT];
#pragma unroll
        for (int k = 0; k < WB_IT; ++k) {
          const int pp = pp0 + k * PSTEP;
          const int r = wb_div20(pp, tw2_magic), c = (pp - r * tw2) * 2;
          const int iy = r0 + r, ix = c0 + c;
          const bool rowok = pp < npairs && iy < P.YH && nch > 0;
          const unsigned off = (unsigned)iy * srow + (unsigned)ix * scol + koff;
          p0[k] = wb_load4(yb, mb, Lslope, off, rowok && ix < P.YW, nch, P.vec_y);
          p1[k] = wb_load4(yb, mb, Lslope, off + scol, rowok && ix + 1 < P.YW, nch, P.vec_y);
        }
#pragma unroll
        for (int k = 0; k < WB_IT; ++k) {
          const int pp = pp0 + k * PSTEP;
          if (pp < npairs) {
            const int r = wb_div20(pp, tw2_magic), c = (pp - r * tw2) * 2;
            bsum += p0[k] + p1[k];  // this thread's channel group is fixed
            unsigned hi[4], lo[4];
            wb_split_pair(p0[k], p1[k], hi, lo);
            unsigned short* dst = yq + r * P.TW + c;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
              *reinterpret_cast<unsigned*>(dst + (size_t)cc * P.DS) = hi[cc];
              *reinterpret_cast<unsigned*>(dst + (size_t)(COB + cc) * P.DS) = lo[cc];
            }
          }
        }
      }
    }
  };
  // ---- SPEC stagers, one tile ahead.  The staging above is four dependent load rounds per tile (X and dY, two batches
  // each): ~1.5 us of exposed latency per round against 1.1 us of matrix work per tile.  Here every load of tile t+2
  // is issued (into registers, nothing depends on it yet) right after tile t+1 has been committed to LDS, and lands
  // while the workers run the K loop of tile t+1: the stagers never wait for memory.
  f32x4 pxa[SPEC ? WB_PIT : 1], pxb[SPEC ? WB_PIT : 1], pya[SPEC ? WB_PIT : 1], pyb[SPEC ? WB_PIT : 1];
  f32x4 pma[SPEC ? WB_PIT : 1], pmb[SPEC ? WB_PIT : 1];
  auto tile_origin = [&](int tile, int& n, int& r0, int& c0) {
    int b = tile, txi, tyi;
    if (ring) {  // rows fastest: consecutive tiles are vertically adjacent
      tyi = b % P.tiles_y;
      b /= P.tiles_y;
      txi = b % P.tiles_x;
      n = b / P.tiles_x;
    } else {
      txi = b % P.tiles_x;
      b /= P.tiles_x;
      tyi = b % P.tiles_y;
      n = b / P.tiles_y;
    }
    r0 = tyi * P.TH;
    c0 = txi * P.TW;
  };
  auto dy_strides = [&](int ch, unsigned& srow, unsigned& scol, unsigned& koff) {
    if (P.dy_ps_r > 1) {
      const int r = P.dy_ps_r, C = P.dy_ps_C;
      const int chc = ch < P.Cout ? ch : 0;
      const int qq = chc / C, c = chc - qq * C;
      const int i = qq / r, j = qq - i * r;
      scol = (unsigned)(r * C);
      srow = (unsigned)(r * P.YW) * scol;
      koff = (unsigned)(i * P.YW) * scol + (unsigned)(j * C + c);
    } else {
      scol = (unsigned)P.Cout;
      srow = (unsigned)P.YW * scol;
      koff = (unsigned)ch;
    }
  };
  // Everything about a stager's items that does not depend on the tile is computed once: byte offset of the item
  // relative to the tile origin, its LDS destination (-1: no item), its column inside the tile.  Per tile and item
  // what is left is one add, two column compares and two selects: loads go through a buffer descriptor of the image
  // (rows above / below the image fall outside it and read as zero, columns left / right get an out-of-range offset),
  // so there is no per-load branch either.  The staging waves are bound by their VALU issue slots (2.1 us per tile
  // with the offsets recomputed per tile: magic divisions, 32-bit multiplies and exec-mask branches around each load).
  int x_rel[SPEC ? WB_PIT : 1], x_dst[SPEC ? WB_PIT : 1], x_hx[SPEC ? WB_PIT : 1], x_hy[SPEC ? WB_PIT : 1];
  int x_q = 0;                         // ring: LDS offset of the thread's channel group
  int st_base[2] = {0, 0}, st_cont[2] = {0, 0}, st_prev = P.HH;   // ring: base / "continues its column" of the tiles in flight
  int y_rel[SPEC ? WB_PIT : 1], y_dst[SPEC ? WB_PIT : 1], y_c[SPEC ? WB_PIT : 1];
  unsigned y_srow = 0, y_scol = 0;
  if constexpr (SPEC) {
    if (stager && P.prefetch) {
      {
        constexpr int QN = CIB / 4, PSTEP = NST / QN;
        const int need2 = (P.TW + P.KW) >> 1;
        const int npairs = P.HH * need2;
        int q, ppb;
        wb_item<QN, NST>(tid, q, ppb);
        const int ch = cib + q * 4;
#pragma unroll
        for (int k = 0; k < WB_PIT; ++k) {
          const int pp = ppb + k * PSTEP;
          const int hy = pp / need2, hx = (pp - hy * need2) * 2;
          const bool act = pp < npairs && ch < P.Cin;
          x_rel[k] = ((hy * P.XW + hx) * P.Cin + ch) * 4;
          x_dst[k] = act ? (q * 4) * P.CS + hy * P.HWp + hx : -1;
          x_hx[k] = hx;
          x_hy[k] = hy;
        }
        x_q = (q * 4) * P.CS;
      }
      {
        constexpr int QN = COB / 4, PSTEP = NST / QN;
        const int npairs = P.TH * tw2;
        int q, ppb;
        wb_item<QN, NST>(tid, q, ppb);
        const int ch = cob + q * 4;
        unsigned koff;
        dy_strides(ch, y_srow, y_scol, koff);
#pragma unroll
        for (int k = 0; k < WB_PIT; ++k) {
          const int pp = ppb + k * PSTEP;
          const int r = pp / tw2, c = (pp - r * tw2) * 2;
          const bool act = pp < npairs && ch < P.Cout;
          y_rel[k] = (int)(((unsigned)r * y_srow + (unsigned)c * y_scol + koff) * 4u);
          y_dst[k] = act ? (q * 4) * P.DS + r * P.TW + c : -1;
          y_c[k] = c;
        }
      }
    }
  }
  auto issue = [&](int tile, int j) {
    if constexpr (SPEC) {
      int n, r0, c0;
      tile_origin(tile, n, r0, c0);
      constexpr unsigned OOB = 0x80000000u;
      int skip_rows = 0;   // ring: halo rows [0, skip_rows) are already in LDS (the predecessor's last rows)
      if (ring) {
        const bool first = j == 0 || (tile % P.tiles_y) == 0;
        st_prev = ring_next(st_prev, first);
        st_base[j & 1] = st_prev;
        st_cont[j & 1] = first ? 0 : 1;
        skip_rows = first ? 0 : P.HH - P.TH;
      }
      {
        const size_t img = (size_t)P.XH * P.XW * P.Cin;
        const __amdgpu_buffer_rsrc_t rx = wb_rsrc(Lx + (size_t)n * img, (unsigned)(img * 4));
        const int by0 = r0 - P.pad, bx0 = c0 - P.pad;
        const int obase = (by0 * P.XW + bx0) * P.Cin * 4;  // may be negative: rows above the image wrap out of range
        const unsigned pstride = (unsigned)P.Cin * 4u;
#pragma unroll
        for (int k = 0; k < WB_PIT; ++k) {
          const unsigned o = (unsigned)(obase + x_rel[k]);
          const int ix = bx0 + x_hx[k];
          const bool act = x_dst[k] >= 0 && x_hy[k] >= skip_rows;
          pxa[k] = wb_bload(rx, act && (unsigned)ix < (unsigned)P.XW ? o : OOB);
          pxb[k] = wb_bload(rx, act && (unsigned)(ix + 1) < (unsigned)P.XW ? o + pstride : OOB);
        }
      }
      {
        const size_t img = (size_t)P.YH * y_srow;
        const __amdgpu_buffer_rsrc_t ry = wb_rsrc(Ldy + (size_t)n * img, (unsigned)(img * 4));
        const __amdgpu_buffer_rsrc_t rm = wb_rsrc(Lmask ? Lmask + (size_t)n * img : Ldy, Lmask ? (unsigned)(img * 4) : 0u);
        const unsigned obase = ((unsigned)r0 * y_srow + (unsigned)c0 * y_scol) * 4u;
        const unsigned pstride = y_scol * 4u;
#pragma unroll
        for (int k = 0; k < WB_PIT; ++k) {
          const unsigned o = obase + (unsigned)y_rel[k];
          const int ix = c0 + y_c[k];
          const bool act = y_dst[k] >= 0;
          const unsigned o0 = act && ix < P.YW ? o : OOB, o1 = act && ix + 1 < P.YW ? o + pstride : OOB;
          pya[k] = wb_bload(ry, o0);
          pyb[k] = wb_bload(ry, o1);
          if (Lmask) {  // raw mask values: applied at commit time, so that nothing here waits for a load
            pma[k] = wb_bload(rm, o0);
            pmb[k] = wb_bload(rm, o1);
          }
        }
      }
    }
  };
  auto commit = [&](int bsel) {
    if constexpr (SPEC) {
      unsigned short* xs = xs_of(bsel);
      unsigned short* ys = ys_of(bsel);
      const int rbase = st_base[bsel], skip_rows = (ring && st_cont[bsel]) ? P.HH - P.TH : 0;
#pragma unroll
      for (int k = 0; k < WB_PIT; ++k) {
        if (x_dst[k] >= 0 && x_hy[k] >= skip_rows) {
          unsigned hi[4], lo[4];
          wb_split_pair(pxa[k], pxb[k], hi, lo);
          int slot = rbase + x_hy[k];
          slot = slot >= R2 ? slot - R2 : slot;
          unsigned short* dst = xs + (ring ? x_q + slot * P.HWp + x_hx[k] : x_dst[k]);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            *reinterpret_cast<unsigned*>(dst + (size_t)c * P.CS) = hi[c];
            *reinterpret_cast<unsigned*>(dst + (size_t)(CIB + c) * P.CS) = lo[c];
          }
        }
      }
#pragma unroll
      for (int k = 0; k < WB_PIT; ++k) {
        if (y_dst[k] >= 0) {
          f32x4 v0 = pya[k], v1 = pyb[k];
          if (Lmask) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v0[e] = pma[k][e] > 0.f ? v0[e] : v0[e] * Lslope;
              v1[e] = pmb[k][e] > 0.f ? v1[e] : v1[e] * Lslope;
            }
          }
          bsum += v0 + v1;
          unsigned hi[4], lo[4];
          wb_split_pair(v0, v1, hi, lo);
          unsigned short* dst = ys + y_dst[k];
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            *reinterpret_cast<unsigned*>(dst + (size_t)cc * P.DS) = hi[cc];
            *reinterpret_cast<unsigned*>(dst + (size_t)(COB + cc) * P.DS) = lo[cc];
          }
        }
      }
    }
  };
  // ---- K loop of the tile in buffer set `bsel` (4 working waves)
  auto kloop = [&](int bsel, int rbase) {
    const unsigned short* xs = xs_of(bsel);
    const unsigned short* ys = ys_of(bsel);
    const unsigned short* xa_h = xs + (size_t)(cit * 16 + i) * P.CS;
    const unsigned short* xa_l = xa_h + (size_t)CIB * P.CS;
    const unsigned short* yb_h = ys + (size_t)(cow * NTW * 16 + i) * P.DS;
    const unsigned short* yb_l = yb_h + (size_t)COB * P.DS;
    if constexpr (K33 && WB_PIPE) {
      // Round 5: the K loop as a software pipeline over (K step, kernel row) phases.  The loop below this one reads a K
      // step's fragments at the top of that step: its ISA has three exposed LDS round trips per K step (the first
      // row's reads, and two ds_read_b128 the compiler sinks into the MFMA stream with an s_waitcnt lgkmcnt(0) right
      // behind each) -- with ONE working wave per SIMD nothing else can use the matrix pipe meanwhile (~2080 clocks per K
      // step for 864 clocks of MFMA issue, tools/wgrad_prof.py).  Here the X fragments of phase p + 1 (the next kernel
      // row; at a step's last row the next step's first row and its dY fragments) are requested BEFORE the 9 * NTW MFMAs
      // of phase p, into a second register set; sched_barriers keep the requests where they are written, the waits the
      // compiler inserts are then counted ones (lgkmcnt(4..8)) behind a phase's worth of MFMAs.  Two K steps per loop
      // trip so that the two register sets alternate without copies.  Same products, same order per accumulator:
      // bit-equal to the loop below.
      struct XF {
        uint4 h, l;
        unsigned eh, el;
      };
      // one packed table word per octet (oct_pk: dY offset | halo column << 12 | tile row << 20); ring and plain tiles share
      // the address form  halo row slot = (base + tile row + u) mod ring rows  (plain: base 0, no wrap)
      struct TB {
        int oy, orr, oc;
      };
      const int nks_run = (SRK_KDBG(P.dbg) & 4) ? 0 : P.nks;
      if (nks_run <= 0) return;
      const int last = nks_run - 1;
      const int rb = ring ? rbase : 0, rwrap = ring ? R2 : (1 << 20);
      auto tab = [&](int ks) -> TB {
        const unsigned w = (unsigned)oct_pk[(ks < last ? ks : last) * 4 + kq];
        TB t;
        t.oy = (int)(w & 0xfffu);
        t.oc = (int)((w >> 12) & 0xffu);
        t.orr = (int)(w >> 20) + rb;
        return t;
      };
      auto xload = [&](XF& f, const TB& t, int u) {
        int slot = t.orr + u;
        slot = slot >= rwrap ? slot - rwrap : slot;
        const int xo = slot * P.HWp + t.oc;
        if (SRK_KDBG(P.dbg) & 16) {  // ablation: no fragment reads (the MFMAs run on whatever the registers hold)
          wb_touch(f.h, xo);
          wb_touch(f.l, xo);
          asm volatile("" : "+v"(f.eh), "+v"(f.el) : "v"(xo));
          return;
        }
        f.h = *reinterpret_cast<const uint4*>(xa_h + xo);
        f.l = *reinterpret_cast<const uint4*>(xa_l + xo);
        if (SRK_KDBG(P.dbg) & 32) return;   // ablation: no 4-byte tail reads
        f.eh = *reinterpret_cast<const unsigned*>(xa_h + xo + 8);
        f.el = *reinterpret_cast<const unsigned*>(xa_l + xo + 8);
      };
      auto yload = [&](uint4 (&bh)[NTW], uint4 (&bl)[NTW], int oy) {
        if (SRK_KDBG(P.dbg) & (16 | 128)) {
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) {
            wb_touch(bh[nt], oy);
            wb_touch(bl[nt], oy);
          }
          return;
        }
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          bh[nt] = *reinterpret_cast<const uint4*>(yb_h + (size_t)nt * 16 * P.DS + oy);
          bl[nt] = *reinterpret_cast<const uint4*>(yb_l + (size_t)nt * 16 * P.DS + oy);
        }
      };
      auto rowmm = [&](const XF& f, const uint4 (&bh)[NTW], const uint4 (&bl)[NTW], f32x4 (&a)[3][NTW]) {
        uint4 ah[3], al[3];
        ah[0] = f.h;
        al[0] = f.l;
        ah[1] = make_uint4(__builtin_amdgcn_alignbit(f.h.y, f.h.x, 16), __builtin_amdgcn_alignbit(f.h.z, f.h.y, 16),
                           __builtin_amdgcn_alignbit(f.h.w, f.h.z, 16), __builtin_amdgcn_alignbit(f.eh, f.h.w, 16));
        al[1] = make_uint4(__builtin_amdgcn_alignbit(f.l.y, f.l.x, 16), __builtin_amdgcn_alignbit(f.l.z, f.l.y, 16),
                           __builtin_amdgcn_alignbit(f.l.w, f.l.z, 16), __builtin_amdgcn_alignbit(f.el, f.l.w, 16));
        ah[2] = make_uint4(f.h.y, f.h.z, f.h.w, f.eh);
        al[2] = make_uint4(f.l.y, f.l.z, f.l.w, f.el);
        if (SRK_KDBG(P.dbg) & 64) {  // ablation: no column shifts (every tap column multiplies the unshifted fragment)
          ah[1] = ah[2] = ah[0];
          al[1] = al[2] = al[0];
        }
        if (SRK_KDBG(P.dbg) & 8) {  // ablation: fragment reads and shifts, no MFMAs
#pragma unroll
          for (int v = 0; v < 3; ++v) {
            wb_keep(ah[v]);
            wb_keep(al[v]);
          }
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) {
            wb_keep(bh[nt]);
            wb_keep(bl[nt]);
          }
          return;
        }
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) a[v][nt] = wb_mfma(al[v], bh[nt], a[v][nt]);
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) a[v][nt] = wb_mfma(ah[v], bl[nt], a[v][nt]);
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) a[v][nt] = wb_mfma(ah[v], bh[nt], a[v][nt]);
      };
      // one K step: rows 0..2 of step `tc` from (x0, yh, yl); leaves row 0 and the dY fragments of step `tn` in (xn, nh, nl)
      auto step = [&](const XF& x0, const uint4 (&yh)[NTW], const uint4 (&yl)[NTW], XF& xn, uint4 (&nh)[NTW],
                      uint4 (&nl)[NTW], const TB& tc, const TB& tn) {
        XF x1 = {}, x2 = {};
        xload(x1, tc, 1);
        __builtin_amdgcn_sched_barrier(0);
        rowmm(x0, yh, yl, acc[0]);
        __builtin_amdgcn_sched_barrier(0);
        xload(x2, tc, 2);
        __builtin_amdgcn_sched_barrier(0);
        rowmm(x1, yh, yl, acc[1]);
        __builtin_amdgcn_sched_barrier(0);
        xload(xn, tn, 0);
        yload(nh, nl, tn.oy);
        __builtin_amdgcn_sched_barrier(0);
        rowmm(x2, yh, yl, acc[2]);
        __builtin_amdgcn_sched_barrier(0);
      };
      TB t0 = tab(0), t1 = tab(1);
      XF xa = {}, xb = {};
      uint4 yah[NTW] = {}, yal[NTW] = {}, ybh[NTW] = {}, ybl[NTW] = {};
      xload(xa, t0, 0);
      yload(yah, yal, t0.oy);
      int ks = 0;
      for (; ks + 1 < nks_run; ks += 2) {
        const TB t2 = tab(ks + 2);
        step(xa, yah, yal, xb, ybh, ybl, t0, t1);
        const TB t3 = tab(ks + 3);
        step(xb, ybh, ybl, xa, yah, yal, t1, t2);
        t0 = t2;
        t1 = t3;
      }
      if (ks < nks_run) step(xa, yah, yal, xb, ybh, ybl, t0, t1);
      return;
    }
    if constexpr (K33) {
      // (round 4: the octet tables of K step ks + 1 are read while step ks multiplies -- the lookups were a dependent LDS
      //  round trip in front of every K step's fragment reads, with one working wave per SIMD and nothing to hide it)
      const int nks_run = (SRK_KDBG(P.dbg) & 4) ? 0 : P.nks;
      int ox_n = oct_x[kq], oy_n = oct_y[kq], or_n = ring ? oct_r[kq] : 0, oc_n = ring ? oct_c[kq] : 0;
      for (int ks = 0; ks < nks_run; ++ks) {
        const int ox = ox_n, oy = oy_n;
        const int orw = ring ? rbase + or_n : 0, ocl = oc_n;
        {
          const int nx = (ks + 1 < nks_run ? ks + 1 : ks) * 4 + kq;
          ox_n = oct_x[nx];
          oy_n = oct_y[nx];
          if (ring) {
            or_n = oct_r[nx];
            oc_n = oct_c[nx];
          }
        }
        uint4 bh[NTW], bl[NTW], qh[3], ql[3];
        unsigned eh[3], el[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          int xo = ox + u * P.HWp;
          if (ring) {
            int slot = orw + u;
            slot = slot >= R2 ? slot - R2 : slot;
            xo = slot * P.HWp + ocl;
          }
          qh[u] = *reinterpret_cast<const uint4*>(xa_h + xo);
          ql[u] = *reinterpret_cast<const uint4*>(xa_l + xo);
          eh[u] = *reinterpret_cast<const unsigned*>(xa_h + xo + 8);
          el[u] = *reinterpret_cast<const unsigned*>(xa_l + xo + 8);
        }
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          bh[nt] = *reinterpret_cast<const uint4*>(yb_h + (size_t)nt * 16 * P.DS + oy);
          bl[nt] = *reinterpret_cast<const uint4*>(yb_l + (size_t)nt * 16 * P.DS + oy);
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          // the three column shifts of this row, then pass-major over its 3 * NTW accumulators: two MFMAs on one
          // accumulator are 3 * NTW issues apart (back to back they wait for each other's result)
          // (Round 4, tools/wgrad_prof.py: the WORKER wave, not the stagers, paces this kernel -- K loop 97 % of its time,
          //  ~2080 clocks per K step for 54 MFMAs, while the first stager wave waits 30 - 49 % of a tile at the barrier.  The
          //  compiler sinks two fragment reads into the MFMA stream and threads the 24 v_perm of the column shifts between the
          //  matrix instructions; scheduling fences that force "all reads, 8 v_perm, 18 MFMAs" per row shift (168 VGPRs) were
          //  measured SLOWER: VDSR layer 0.157 -> 0.172 ms, step 6.30 -> 6.93 ms.  Not kept.)
          uint4 ah[3], al[3];
          ah[0] = qh[u];
          al[0] = ql[u];
          ah[1] = make_uint4(__builtin_amdgcn_alignbit(qh[u].y, qh[u].x, 16), __builtin_amdgcn_alignbit(qh[u].z, qh[u].y, 16),
                             __builtin_amdgcn_alignbit(qh[u].w, qh[u].z, 16), __builtin_amdgcn_alignbit(eh[u], qh[u].w, 16));
          al[1] = make_uint4(__builtin_amdgcn_alignbit(ql[u].y, ql[u].x, 16), __builtin_amdgcn_alignbit(ql[u].z, ql[u].y, 16),
                             __builtin_amdgcn_alignbit(ql[u].w, ql[u].z, 16), __builtin_amdgcn_alignbit(el[u], ql[u].w, 16));
          ah[2] = make_uint4(qh[u].y, qh[u].z, qh[u].w, eh[u]);
          al[2] = make_uint4(ql[u].y, ql[u].z, ql[u].w, el[u]);
#pragma unroll
          for (int v = 0; v < 3; ++v)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[u][v][nt] = wb_mfma(al[v], bh[nt], acc[u][v][nt]);
#pragma unroll
          for (int v = 0; v < 3; ++v)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[u][v][nt] = wb_mfma(ah[v], bl[nt], acc[u][v][nt]);
#pragma unroll
          for (int v = 0; v < 3; ++v)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[u][v][nt] = wb_mfma(ah[v], bh[nt], acc[u][v][nt]);
        }
      }
      return;
    }
    for (int ks = 0; ks < ((SRK_KDBG(P.dbg) & 4) ? 0 : P.nks); ++ks) {
      const int ox = oct_x[ks * 4 + kq], oy = oct_y[ks * 4 + kq];
      const int orw = ring ? rbase + oct_r[ks * 4 + kq] : 0, ocl = ring ? oct_c[ks * 4 + kq] : 0;
      uint4 bh[NTW], bl[NTW];
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        bh[nt] = *reinterpret_cast<const uint4*>(yb_h + (size_t)nt * 16 * P.DS + oy);
        bl[nt] = *reinterpret_cast<const uint4*>(yb_l + (size_t)nt * 16 * P.DS + oy);
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        if (u < P.KH) {
          int xo = ox + u * P.HWp;
          if (ring) {  // halo row orow + u of the tile sits in ring slot (base + orow + u) mod 2 HH
            int slot = orw + u;
            slot = slot >= R2 ? slot - R2 : slot;
            xo = slot * P.HWp + ocl;
          }
          const unsigned short* ph = xa_h + xo;
          const unsigned short* pl = xa_l + xo;
          const uint4 qh = *reinterpret_cast<const uint4*>(ph);
          const uint4 ql = *reinterpret_cast<const uint4*>(pl);
          const unsigned eh = *reinterpret_cast<const unsigned*>(ph + 8);
          const unsigned el = *reinterpret_cast<const unsigned*>(pl + 8);
#pragma unroll
          for (int v = 0; v < 3; ++v) {
            if (v < P.KW) {
              uint4 ah, al;
              if (v == 0) {
                ah = qh;
                al = ql;
              } else if (v == 1) {
                ah = make_uint4(__builtin_amdgcn_alignbit(qh.y, qh.x, 16), __builtin_amdgcn_alignbit(qh.z, qh.y, 16),
                                __builtin_amdgcn_alignbit(qh.w, qh.z, 16), __builtin_amdgcn_alignbit(eh, qh.w, 16));
                al = make_uint4(__builtin_amdgcn_alignbit(ql.y, ql.x, 16), __builtin_amdgcn_alignbit(ql.z, ql.y, 16),
                                __builtin_amdgcn_alignbit(ql.w, ql.z, 16), __builtin_amdgcn_alignbit(el, ql.w, 16));
              } else {
                ah = make_uint4(qh.y, qh.z, qh.w, eh);
                al = make_uint4(ql.y, ql.z, ql.w, el);
              }
#pragma unroll
              for (int nt = 0; nt < NTW; ++nt) acc[u][v][nt] = wb_mfma(al, bh[nt], acc[u][v][nt]);
#pragma unroll
              for (int nt = 0; nt < NTW; ++nt) acc[u][v][nt] = wb_mfma(ah, bl[nt], acc[u][v][nt]);
#pragma unroll
              for (int nt = 0; nt < NTW; ++nt) acc[u][v][nt] = wb_mfma(ah, bh[nt], acc[u][v][nt]);
            }
          }
        }
      }
    }
  };

  if (!SPEC) {
    for (int tile = bx; tile < P.ntiles; tile += P.G) {
      __syncthreads();  // previous tile fully consumed (tables / zero octets visible on the first pass)
      stage(tile, 0);
      __syncthreads();
      kloop(0, 0);
    }
  } else {
    // tiles of this block: every G-th one, or (ring) a contiguous range in rows-fastest order
    const int per = (P.ntiles + P.G - 1) / P.G;
    const int first_tile = ring ? bx * per : bx, tstep = ring ? 1 : P.G;
    int ntb = (bx < P.ntiles) ? (P.ntiles - bx + P.G - 1) / P.G : 0;
    if (ring) {
      ntb = P.ntiles - first_tile;
      ntb = ntb < 0 ? 0 : (ntb > per ? per : ntb);
    }
    __syncthreads();  // tables / zero octets visible
    // Two loops, one per role, with the same barriers (1 + ntb): the stagers' prefetch registers and the workers'
    // accumulators are never live in the same wave.
    if (stager) {
      if (P.prefetch && !(SRK_KDBG(P.dbg) & 2)) {
        if (ntb > 0) {
          issue(first_tile, 0);
          commit(0);
          if (ntb > 1) issue(first_tile + tstep, 1);
        }
        __syncthreads();
        long long st_commit = 0, st_issue = 0, st_wait = 0;
        for (int it = 0; it < ntb; ++it) {
          const long long c0 = WB_CLK();
          if (it + 1 < ntb) commit((it + 1) & 1);        // tile it+1: its loads were issued an iteration ago
          const long long c1 = WB_CLK();
          if (it + 2 < ntb) issue(first_tile + (it + 2) * tstep, it + 2);  // tile it+2: lands under the K loop of tile it+1
          const long long c2 = WB_CLK();
          __syncthreads();
          st_commit += c1 - c0;
          st_issue += c2 - c1;
          st_wait += WB_CLK() - c2;
        }
#ifdef SRK_EXPERIMENTS
        if (P.prof && tid0 == WTHR) {
          long long* pr = P.prof + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16;
          pr[0] = st_commit; pr[1] = st_issue; pr[2] = st_wait; pr[3] = ntb;
        }
#endif
        (void)st_commit; (void)st_issue; (void)st_wait;
      } else {
        if (ntb > 0) stage(bx, 0);     // (no ring without the prefetch path: the host never sets both)
        __syncthreads();
        for (int it = 0; it < ntb; ++it) {
          if (it + 1 < ntb) stage(bx + (it + 1) * P.G, (it + 1) & 1);
          __syncthreads();
        }
      }
    } else {
      __syncthreads();
      if (WB_PRIO) __builtin_amdgcn_s_setprio(WB_PRIO);  // (this branch is taken by whole waves: wave >= NWV is wave-uniform)
      int wprev = P.HH;
      long long wk_loop = 0, wk_wait = 0;
      for (int it = 0; it < ntb; ++it) {
        const long long k0 = WB_CLK();
        if (ring) wprev = ring_next(wprev, it == 0 || ((first_tile + it) % P.tiles_y) == 0);
        kloop(it & 1, wprev);
        const long long k1 = WB_CLK();
        __syncthreads();
        wk_loop += k1 - k0;
        wk_wait += WB_CLK() - k1;
      }
#ifdef SRK_EXPERIMENTS
      if (P.prof && tid0 == 0) {
        long long* pr = P.prof + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + 8;
        pr[0] = wk_loop; pr[1] = wk_wait; pr[2] = ntb; pr[3] = (long long)ntb * P.nks;
      }
#endif
      (void)wk_loop; (void)wk_wait;
    }
  }

  if (want_bias) {  // column sums of dY: combine the threads that staged the same channel group
    constexpr int QN = COB / 4;
    __syncthreads();
    if (stager) {
#pragma unroll
      for (int e = 0; e < 4; ++e) bred[tid][e] = bsum[e];
    }
    __syncthreads();
    if (stager && tid < COB && cob + tid < P.Cout) {
      const int q = tid >> 2, e = tid & 3;
      float s = 0.f;
      if (WB_CF_MAP) {   // the staging threads of channel group q: lanes 16 (q % 4) .. + 15 of waves q / 4, q / 4 + QN / 4, ...
        for (int w = q >> 2; w < NST / 64; w += QN / 4)
          for (int jj = 0; jj < 16; ++jj) s += bred[w * 64 + (q & 3) * 16 + jj][e];
      } else {
        for (int t = q; t < NST; t += QN) s += bred[t][e];
      }
      P.bias_partial[(size_t)bxl * P.Cout + cob + tid] = s;
    }
  }
  if (!worker) return;
  // partial slab ws[g][t][ci][co]; C/D layout: col = lane&15 (co), row = (lane>>4)*4 + reg (ci)
  float* slab = P.ws + (size_t)bxl * P.KH * P.KW * P.Cin * P.Cout;
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      if (u < P.KH && v < P.KW) {
        const int t = u * P.KW + v;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          const int co = cob + (cow * NTW + nt) * 16 + i;
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int ci = cib + cit * 16 + kq * 4 + reg;
            if (ci < P.Cin && co < P.Cout) slab[((size_t)t * P.Cin + ci) * P.Cout + co] = acc[u][v][nt][reg];
          }
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------
// k_wgrad_tr: the wave-specialised 3x3 weight gradient on a PIXEL-MAJOR LDS image read with the LDS transpose read
// (ds_read_b64_tr_b16, gfx950).  Same roles, tiles, ring of X halo rows, split-K slabs and summation order as
// k_wgrad_bf<2, 2, 2, SPEC, ., K33> -- results are bit-equal -- but the staged tensors stay in their NHWC order:
//
//   X ring  [hi | lo][ring row][halo column][32 ci] bf16      (64 B per pixel and plane)
//   dY set  [hi | lo][tile pixel (+ 8 zero pixels)][64 co]     (128 B per pixel and plane), two sets
//
// * a staging item is ONE pixel x 8 channels (two 16-byte loads): the bf16 split packs CHANNEL pairs and leaves as one
//   ds_write_b128 per plane -- 2 LDS stores per item instead of the 8 transposing ds_write_b32 of k_wgrad_bf (the eight
//   stager waves were bound by their instruction issue: tools/wgrad_variant.sh ablations, stagers alone 133 us of the
//   170 us VDSR layer);
// * a working wave's MFMA fragment (lane = channel, 8 consecutive pixels of a tile row) comes out of two transpose reads:
//   in a 16-lane group lane r addresses pixel (r >> 2), channels 4 (r & 3) .. + 3 and receives channel r of the
//   group's four pixels.  The three column shifts of a kernel row are derived in registers from pixels 0 .. 9 as before
//   (a third read fetches pixels 8 .. 11);
// * bank conflicts: a transpose read serves 32 lanes (two octets) per clock over 64 banks.  X: the two 16-channel
//   halves of a pixel swap places in odd halo octets (column >> 3 odd); dY: the four 16-channel tiles of a pixel are
//   XORed with  (pixel >> 1 & 1) | (pixel >> 3 & 1) << 1.  Both are lane constants on the reading side.
// Requirements (wt_eligible): 3x3, stride 1, Cin % 32 == 0, Cout % 64 == 0, 16-byte aligned tensors, pixel-shuffled dY
// only with >= 64 channels per sub-pixel; everything else stays on k_wgrad_bf.
// ---------------------------------------------------------------------------------------------
typedef short wt_s16x4 __attribute__((ext_vector_type(4)));
typedef short wt_s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) wt_s16x4* wt_ldsp;

// (a, b, c, d) x (e, f, g, h) fp32 -> 4 dwords of packed bf16 channel pairs, hi and lo parts
__device__ __forceinline__ void wt_split8(const f32x4& v0, const f32x4& v1, uint4& hi, uint4& lo) {
  const f32x4 ev = {v0[0], v0[2], v1[0], v1[2]}, od = {v0[1], v0[3], v1[1], v1[3]};
  unsigned h[4], l[4];
  wb_split_pair(ev, od, h, l);
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  lo = make_uint4(l[0], l[1], l[2], l[3]);
}

// NTW = 2: four working waves (2 ci tiles x 2 pairs of co tiles, one per SIMD beside two stager waves; 168 VGPRs).
// NTW = 1: EIGHT working waves (2 x 4 single co tiles, two per SIMD: one wave's transpose reads are issued while the other's
// MFMAs run -- a lone working wave pays ~16 clocks of matrix pipe per read, section 12.3 of DESIGN.md; 128 VGPRs).
template <bool GRP, int NTW>
__global__ __launch_bounds__(NTW == 2 ? 768 : 1024, NTW == 2 ? 3 : 4) void k_wgrad_tr(WgBfParams P,
                                                                                       typename WgGroupArg<GRP>::type GR) {
  constexpr int CIB = 32, COB = 64, NWV = 8 / NTW, WTHR = 64 * NWV, NST = WB_SST, NTHR = WTHR + NST, PIT = 1024 / NST;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
  __shared__ unsigned oct_tw[WB_MAXOCT];
  const long long clk_begin = WB_CLK();
  const int tid0 = threadIdx.x, lane = tid0 & 63, wave = tid0 >> 6;
  const bool worker = wave < NWV;
  const int tid = worker ? tid0 : tid0 - WTHR;
  const int R2 = 2 * P.HH;
  const unsigned XPL = (unsigned)P.XPL, YPL = (unsigned)P.YPL;   // plane sizes in bytes
  const unsigned XROW = (unsigned)P.XP * 64u;                   // bytes per ring row
  const unsigned YSET0 = 2u * XPL, YSET = 2u * YPL;
  const int noct = P.TH * P.TWo;
  int bxl = blockIdx.x, byl = blockIdx.y;
  if (gridDim.y == 2 && (gridDim.x & 7) == 0) {   // the two ci halves of a slab index on one XCD (see k_wgrad_bf)
    const int lin = (int)(blockIdx.y * gridDim.x + blockIdx.x);
    bxl = (lin >> 4) * 8 + (lin & 7);
    byl = (lin >> 3) & 1;
  }
  const int cib = byl * CIB, cob = blockIdx.z * COB;
  const bool want_bias = P.bias_partial != nullptr && byl == 0;
  int bx = bxl;
  const float* __restrict__ Lx = P.x;
  const float* __restrict__ Ldy = P.dy;
  const float* __restrict__ Lmask = P.mask_y;
  float Lslope = P.mask_slope;
  if constexpr (GRP) {
    const int layer = bxl / P.G;
    bx = bxl - layer * P.G;
    Lx = GR.L[layer].x;
    Ldy = GR.L[layer].dy;
    Lmask = GR.L[layer].mask_y;
    Lslope = GR.L[layer].mask_slope;
  }
  // octet table: tile row << 16 | dY octet << 8 | octet column; K padding: X octet (0, 0) times the zero octet behind a dY plane
  for (int o = tid0; o < P.nks * 4; o += NTHR) {
    if (o < noct) {
      const int orow = o / P.TWo;
      oct_tw[o] = (unsigned)(orow << 16) | (unsigned)(o << 8) | (unsigned)(o - orow * P.TWo);
    } else {
      oct_tw[o] = (unsigned)(noct << 8);
    }
  }
  for (int e = tid0; e < 2 * 2 * 64; e += NTHR) {   // 8 zero pixels x 128 B behind each of the four dY planes
    const int pl = e >> 6, w = e & 63;
    *reinterpret_cast<uint4*>(smem8 + YSET0 + (unsigned)pl * YPL + (unsigned)(P.TH * P.TW) * 128u + (unsigned)w * 16u) =
        make_uint4(0u, 0u, 0u, 0u);
  }
  // tiles of this block: a contiguous range in rows-fastest order (ring of halo rows)
  const int per = (P.ntiles + P.G - 1) / P.G;
  const int first_tile = bx * per;
  int ntb = P.ntiles - first_tile;
  ntb = ntb < 0 ? 0 : (ntb > per ? per : ntb);
  auto ring_next = [&](int prev, bool first) {
    const int b = prev + (first ? P.HH : P.TH);
    return b >= R2 ? b - R2 : b;
  };
  auto tile_origin = [&](int tile, int& n, int& r0, int& c0) {
    int b = tile;
    const int tyi = b % P.tiles_y;
    b /= P.tiles_y;
    const int txi = b % P.tiles_x;
    n = b / P.tiles_x;
    r0 = tyi * P.TH;
    c0 = txi * P.TW;
  };
  __syncthreads();  // table / zero pixels visible

  if (!worker) {
    // ------------------------------------------------------------------ stagers (8 waves)
    f32x4 bsum0 = {0.f, 0.f, 0.f, 0.f}, bsum1 = {0.f, 0.f, 0.f, 0.f};
    f32x4 pxa[PIT], pxb[PIT], pya[PIT], pyb[PIT], pma[PIT], pmb[PIT];
    int x_rel[PIT], x_lds[PIT], x_hx[PIT], x_hy[PIT];
    int y_rel[PIT], y_lds[PIT], y_c[PIT];
    unsigned y_srow = 0, y_scol = 0;
    {
      const int XN = P.TW + 2, npix = P.HH * XN;
      const int g = tid & 3;
#pragma unroll
      for (int k = 0; k < PIT; ++k) {
        const int pi = (tid + k * NST) >> 2;
        const int hy = pi / XN, hx = pi - hy * XN;
        x_rel[k] = ((hy * P.XW + hx) * P.Cin + cib + g * 8) * 4;
        x_lds[k] = pi < npix ? hx * 64 + ((((g >> 1) ^ (hx >> 3)) & 1) << 5) + ((g & 1) << 4) : -1;
        x_hx[k] = hx;
        x_hy[k] = hy;
      }
    }
    {
      const int npix = P.TH * P.TW;
      const int g = tid & 7;
      unsigned koff;
      if (P.dy_ps_r > 1) {   // packed channel (i, j, c) of pixel (y, x) lives at (y r + i, x r + j, c): this block = one (i, j)
        const int r = P.dy_ps_r, C = P.dy_ps_C;
        const int qq = cob / C, c = cob - qq * C;
        const int i = qq / r, j = qq - i * r;
        y_scol = (unsigned)(r * C);
        y_srow = (unsigned)(r * P.YW) * y_scol;
        koff = (unsigned)(i * P.YW) * y_scol + (unsigned)(j * C + c);
      } else {
        y_scol = (unsigned)P.Cout;
        y_srow = (unsigned)P.YW * y_scol;
        koff = (unsigned)cob;
      }
#pragma unroll
      for (int k = 0; k < PIT; ++k) {
        const int px = (tid + k * NST) >> 3;
        const int r = px / P.TW, c = px - r * P.TW;
        const int f = ((px >> 1) & 1) | (((px >> 3) & 1) << 1);
        y_rel[k] = (int)(((unsigned)r * y_srow + (unsigned)c * y_scol + koff + (unsigned)(g * 8)) * 4u);
        y_lds[k] = px < npix ? px * 128 + (((g >> 1) ^ f) << 5) + ((g & 1) << 4) : -1;
        y_c[k] = c;
      }
    }
    int st_base[2] = {0, 0}, st_skip[2] = {0, 0}, st_prev = P.HH;
    auto issue = [&](int tile, int j) {
      int n, r0, c0;
      tile_origin(tile, n, r0, c0);
      constexpr unsigned OOB = 0x80000000u;
      const bool first = j == 0 || (tile % P.tiles_y) == 0;
      st_prev = ring_next(st_prev, first);
      st_base[j & 1] = st_prev;
      const int skip_rows = first ? 0 : P.HH - P.TH;   // halo rows [0, skip) are the predecessor's last rows
      st_skip[j & 1] = skip_rows;
      {
        const size_t img = (size_t)P.XH * P.XW * P.Cin;
        const __amdgpu_buffer_rsrc_t rx = wb_rsrc(Lx + (size_t)n * img, (unsigned)(img * 4));
        const int by0 = r0 - P.pad, bx0 = c0 - P.pad;
        const int obase = (by0 * P.XW + bx0) * P.Cin * 4;  // may be negative: rows above the image wrap out of range
#pragma unroll
        for (int k = 0; k < PIT; ++k) {
          const unsigned o = (unsigned)(obase + x_rel[k]);
          const int ix = bx0 + x_hx[k];
          const bool act = x_lds[k] >= 0 && x_hy[k] >= skip_rows && (unsigned)ix < (unsigned)P.XW;
          pxa[k] = wb_bload(rx, act ? o : OOB);
          pxb[k] = wb_bload(rx, act ? o + 16u : OOB);
        }
      }
      {
        const size_t img = (size_t)P.YH * y_srow;
        const __amdgpu_buffer_rsrc_t ry = wb_rsrc(Ldy + (size_t)n * img, (unsigned)(img * 4));
        const __amdgpu_buffer_rsrc_t rm = wb_rsrc(Lmask ? Lmask + (size_t)n * img : Ldy, Lmask ? (unsigned)(img * 4) : 0u);
        const unsigned obase = ((unsigned)r0 * y_srow + (unsigned)c0 * y_scol) * 4u;
#pragma unroll
        for (int k = 0; k < PIT; ++k) {
          const unsigned o = obase + (unsigned)y_rel[k];
          const bool act = y_lds[k] >= 0 && c0 + y_c[k] < P.YW;
          const unsigned o0 = act ? o : OOB, o1 = act ? o + 16u : OOB;
          pya[k] = wb_bload(ry, o0);
          pyb[k] = wb_bload(ry, o1);
          if (Lmask) {  // raw mask values: applied at commit time, so that nothing here waits for a load
            pma[k] = wb_bload(rm, o0);
            pmb[k] = wb_bload(rm, o1);
          }
        }
      }
    };
    auto commit = [&](int bsel) {
      const int rbase = st_base[bsel], skip_rows = st_skip[bsel];
#pragma unroll
      for (int k = 0; k < PIT; ++k) {
        if (x_lds[k] >= 0 && x_hy[k] >= skip_rows) {
          uint4 hi, lo;
          wt_split8(pxa[k], pxb[k], hi, lo);
          int slot = rbase + x_hy[k];
          slot = slot >= R2 ? slot - R2 : slot;
          unsigned char* dst = smem8 + (unsigned)slot * XROW + (unsigned)x_lds[k];
          *reinterpret_cast<uint4*>(dst) = hi;
          *reinterpret_cast<uint4*>(dst + XPL) = lo;
        }
      }
      unsigned char* ys = smem8 + YSET0 + (unsigned)bsel * YSET;
#pragma unroll
      for (int k = 0; k < PIT; ++k) {
        if (y_lds[k] >= 0) {
          f32x4 v0 = pya[k], v1 = pyb[k];
          if (Lmask) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v0[e] = pma[k][e] > 0.f ? v0[e] : v0[e] * Lslope;
              v1[e] = pmb[k][e] > 0.f ? v1[e] : v1[e] * Lslope;
            }
          }
          if (want_bias) {   // (block-uniform: bias-free stacks -- VDSR -- and the second ci half skip the column sums)
            bsum0 += v0;
            bsum1 += v1;
          }
          uint4 hi, lo;
          wt_split8(v0, v1, hi, lo);
          unsigned char* dst = ys + (unsigned)y_lds[k];
          *reinterpret_cast<uint4*>(dst) = hi;
          *reinterpret_cast<uint4*>(dst + YPL) = lo;
        }
      }
    };
    if (!(SRK_KDBG(P.dbg) & 2)) {
      if (ntb > 0) {
        issue(first_tile, 0);
        commit(0);
        if (ntb > 1) issue(first_tile + 1, 1);
      }
      __syncthreads();
      long long st_commit = 0, st_issue = 0, st_wait = 0;
      for (int it = 0; it < ntb; ++it) {
        const long long c0 = WB_CLK();
        if (it + 1 < ntb) commit((it + 1) & 1);              // tile it + 1: its loads were issued an iteration ago
        const long long c1 = WB_CLK();
        if (it + 2 < ntb) issue(first_tile + it + 2, it + 2);  // tile it + 2: lands under the K loop of tile it + 1
        const long long c2 = WB_CLK();
        __syncthreads();
        st_commit += c1 - c0;
        st_issue += c2 - c1;
        st_wait += WB_CLK() - c2;
      }
#ifdef SRK_EXPERIMENTS
      if (P.prof && tid0 == WTHR) {
        long long* pr = P.prof + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16;
        pr[0] = st_commit; pr[1] = st_issue; pr[2] = st_wait; pr[3] = ntb;
      }
#endif
      (void)st_commit; (void)st_issue; (void)st_wait;
    } else {
      __syncthreads();
      for (int it = 0; it < ntb; ++it) __syncthreads();
    }
    if (want_bias) {   // column sums of dY: the 64 staging threads of one 8-channel group (tid & 7) add up in LDS
      float* bred = reinterpret_cast<float*>(smem8);   // [NST][8 + 1 pad], the X ring is no longer read (barrier above)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bred[tid * 9 + e] = bsum0[e];
        bred[tid * 9 + 4 + e] = bsum1[e];
      }
    }
    __syncthreads();
    if (want_bias && tid < COB) {
      const float* bred = reinterpret_cast<const float*>(smem8);
      const int g = tid >> 3, e = tid & 7;
      float s = 0.f;
      for (int t = g; t < NST; t += 8) s += bred[t * 9 + e];
      P.bias_partial[(size_t)bxl * P.Cout + cob + tid] = s;
    }
    return;
  }

  // -------------------------------------------------------------------- workers (cit x cow)
  const int r16 = lane & 15, kq = lane >> 4;
  const int cit = wave & 1, cow = wave >> 1;
  f32x4 acc[3][3][NTW];
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int v = 0; v < 3; ++v)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) acc[u][v][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  struct XF {
    uint4 h, l;
    unsigned eh, el;
    unsigned xh, xl;   // pixels 10, 11 of the third reads: never multiplied, kept live until the row has been (a register the
                       // allocator hands out again while its read is in flight puts an s_waitcnt into the MFMA stream)
  };
  typedef unsigned TB;   // an octet's table word
  const unsigned lane_x = (unsigned)((r16 >> 2) * 64 + (r16 & 3) * 8);
  const int fl = ((r16 >> 3) & 1) | ((kq & 1) << 1);
  const unsigned lane_y0 = (unsigned)((r16 >> 2) * 128 + (r16 & 3) * 8 + (((cow * NTW) ^ fl) << 5));
  const int nks_run = (SRK_KDBG(P.dbg) & 4) ? 0 : P.nks;
  const int last = nks_run - 1;
  auto tab = [&](int ks) -> TB { return oct_tw[(ks < last ? ks : last) * 4 + kq]; };
  auto tr = [&](unsigned off) -> uint2 {
    const wt_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wt_ldsp)(smem8 + off));
    return __builtin_bit_cast(uint2, v);
  };
  int rb = 0;   // ring row of the current tile's halo row 0
  unsigned ysel = YSET0;   // dY buffer set of the current tile
  // byte address of a lane's first transpose read of halo row (octet row + u)
  auto xaddr = [&](const TB& t, int u) -> unsigned {
    const unsigned oc = t & 0xffu;
    int slot = rb + (int)(t >> 16) + u;
    slot = slot >= R2 ? slot - R2 : slot;
    return __umul24((unsigned)slot, XROW) + (oc << 9) + (((oc ^ (unsigned)cit) & 1u) << 5) + lane_x;
  };
  auto yaddr = [&](const TB& t) -> unsigned { return ysel + ((t & 0xff00u) << 2) + lane_y0; };
  auto xissue = [&](XF& f, unsigned a) {
    if (WT_ABL & 16) return;   // ablation: no fragment reads (the MFMAs run on whatever the registers hold)
    const unsigned b = (a + 512u) ^ 32u;   // pixels 8 .. 11: the next halo octet (its channel halves are swapped)
    const uint2 h0 = tr(a), h1 = tr(a + 256u), l0 = tr(a + XPL), l1 = tr(a + XPL + 256u);
    const uint2 h2 = tr(b), l2 = tr(b + XPL);
    f.h = make_uint4(h0.x, h0.y, h1.x, h1.y);
    f.l = make_uint4(l0.x, l0.y, l1.x, l1.y);
    f.eh = h2.x;
    f.el = l2.x;
    f.xh = h2.y;
    f.xl = l2.y;
  };
  auto yissue = [&](uint4 (&bh)[NTW], uint4 (&bl)[NTW], unsigned a0) {
    if (WT_ABL & 16) return;
    const uint2 p0 = tr(a0), p1 = tr(a0 + 512u), q0 = tr(a0 + YPL), q1 = tr(a0 + YPL + 512u);
    bh[0] = make_uint4(p0.x, p0.y, p1.x, p1.y);
    bl[0] = make_uint4(q0.x, q0.y, q1.x, q1.y);
    if constexpr (NTW == 2) {
      const unsigned a1 = a0 ^ 32u;   // the wave's second 16-channel tile: lane_y1 = lane_y0 ^ 32
      const uint2 p2 = tr(a1), p3 = tr(a1 + 512u), q2 = tr(a1 + YPL), q3 = tr(a1 + YPL + 512u);
      bh[NTW - 1] = make_uint4(p2.x, p2.y, p3.x, p3.y);
      bl[NTW - 1] = make_uint4(q2.x, q2.y, q3.x, q3.y);
    }
  };
  auto rowmm = [&](const XF& f, const uint4 (&bh)[NTW], const uint4 (&bl)[NTW], f32x4 (&a)[3][NTW]) {
    uint4 ah[3], al[3];
    ah[0] = f.h;
    al[0] = f.l;
    ah[1] = make_uint4(__builtin_amdgcn_alignbit(f.h.y, f.h.x, 16), __builtin_amdgcn_alignbit(f.h.z, f.h.y, 16),
                       __builtin_amdgcn_alignbit(f.h.w, f.h.z, 16), __builtin_amdgcn_alignbit(f.eh, f.h.w, 16));
    al[1] = make_uint4(__builtin_amdgcn_alignbit(f.l.y, f.l.x, 16), __builtin_amdgcn_alignbit(f.l.z, f.l.y, 16),
                       __builtin_amdgcn_alignbit(f.l.w, f.l.z, 16), __builtin_amdgcn_alignbit(f.el, f.l.w, 16));
    ah[2] = make_uint4(f.h.y, f.h.z, f.h.w, f.eh);
    al[2] = make_uint4(f.l.y, f.l.z, f.l.w, f.el);
    if (WT_ABL & 64) {   // ablation: no column shifts
      ah[1] = ah[2] = ah[0];
      al[1] = al[2] = al[0];
    }
    // (dY is the A operand: the accumulator rows are output channels, a lane's four registers four consecutive co of
    //  one ci -- one 16-byte store per tile into the [tap][ci][co] slab; same products, same order as X-first)
#pragma unroll
    for (int v = 0; v < 3; ++v)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) a[v][nt] = wb_mfma(bh[nt], al[v], a[v][nt]);
#pragma unroll
    for (int v = 0; v < 3; ++v)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) a[v][nt] = wb_mfma(bl[nt], ah[v], a[v][nt]);
#pragma unroll
    for (int v = 0; v < 3; ++v)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) a[v][nt] = wb_mfma(bh[nt], ah[v], a[v][nt]);
    asm volatile("" ::"v"(f.xh), "v"(f.xl));
  };
  // A phase = the 18 MFMAs of one (K step, kernel row).  Its region starts with the transpose reads of the NEXT phase
  // (addresses computed one phase earlier) and then holds ONLY matrix instructions with the VALU work -- this row's column
  // shifts, the address of the phase after next -- threaded between them: the matrix pipe takes an MFMA every 16 clocks and
  // stands still while its wave issues anything else in front of one.
  // WT_SGB == 3: the same phase with the order of its matrix instructions and LDS reads written down -- MFMA q, then read q of
  // the next phase's fragments (sched_barrier(0x6): only VALU / SALU instructions may move across, so the column shifts and
  // the address arithmetic settle into the gaps the compiler finds for them).  The first reads are the ones the next phase's
  // first MFMAs need (lo halves, then dY).
  auto xread1 = [&](XF& f, unsigned a, int k) {
    if (WT_ABL & 16) return;
    const unsigned b = (a + 512u) ^ 32u;
    if (k == 0) { const uint2 t = tr(a + XPL); f.l.x = t.x; f.l.y = t.y; }
    if (k == 1) { const uint2 t = tr(a + XPL + 256u); f.l.z = t.x; f.l.w = t.y; }
    if (k == 2) { const uint2 t = tr(a); f.h.x = t.x; f.h.y = t.y; }
    if (k == 3) { const uint2 t = tr(a + 256u); f.h.z = t.x; f.h.w = t.y; }
    if (k == 4) { const uint2 t = tr(b + XPL); f.el = t.x; f.xl = t.y; }
    if (k == 5) { const uint2 t = tr(b); f.eh = t.x; f.xh = t.y; }
  };
  auto yread1 = [&](uint4 (&bh)[NTW], uint4 (&bl)[NTW], unsigned a0, int k) {
    if (WT_ABL & 16) return;
    const unsigned a = (k & 4) ? (a0 ^ 32u) : a0;
    const int nt = (k >> 2) < NTW ? (k >> 2) : NTW - 1;
    if ((k & 3) == 0) { const uint2 t = tr(a); bh[nt].x = t.x; bh[nt].y = t.y; }
    if ((k & 3) == 1) { const uint2 t = tr(a + 512u); bh[nt].z = t.x; bh[nt].w = t.y; }
    if ((k & 3) == 2) { const uint2 t = tr(a + YPL); bl[nt].x = t.x; bl[nt].y = t.y; }
    if ((k & 3) == 3) { const uint2 t = tr(a + YPL + 512u); bl[nt].z = t.x; bl[nt].w = t.y; }
  };
  // MFMA q of a row's 18 (pass-major: lo x hi, hi x lo, hi x hi; tap column, then output tile)
  auto mm1 = [&](const uint4 (&ah)[3], const uint4 (&al)[3], const uint4 (&bh)[NTW], const uint4 (&bl)[NTW], f32x4 (&a)[3][NTW],
                 int q) {
    const int pass = q / 6, v = (q % 6) / NTW, nt = q % NTW;
    a[v][nt] = wb_mfma(pass == 1 ? bl[nt] : bh[nt], pass == 0 ? al[v] : ah[v], a[v][nt]);
  };
  auto shifts = [&](const XF& f, uint4 (&ah)[3], uint4 (&al)[3]) {
    ah[0] = f.h;
    al[0] = f.l;
    ah[1] = make_uint4(__builtin_amdgcn_alignbit(f.h.y, f.h.x, 16), __builtin_amdgcn_alignbit(f.h.z, f.h.y, 16),
                       __builtin_amdgcn_alignbit(f.h.w, f.h.z, 16), __builtin_amdgcn_alignbit(f.eh, f.h.w, 16));
    al[1] = make_uint4(__builtin_amdgcn_alignbit(f.l.y, f.l.x, 16), __builtin_amdgcn_alignbit(f.l.z, f.l.y, 16),
                       __builtin_amdgcn_alignbit(f.l.w, f.l.z, 16), __builtin_amdgcn_alignbit(f.el, f.l.w, 16));
    ah[2] = make_uint4(f.h.y, f.h.z, f.h.w, f.eh);
    al[2] = make_uint4(f.l.y, f.l.z, f.l.w, f.el);
    if (WT_ABL & 64) {
      ah[1] = ah[2] = ah[0];
      al[1] = al[2] = al[0];
    }
  };
  // (tools/wgrad_variant.sh ablations, workers alone on the VDSR layer: all 92.6 us, no column shifts 87.6, no fragment reads
  //  65.4 -- a transpose read issued in front of the MFMAs costs the matrix pipe ~16 clocks: with WT_SGB = 2 the NR reads of a
  //  region are threaded one per MFMA gap as well, oldest first.)
  auto mm_region = [&](int nreads) {
    if (WT_SGB == 2) {
#pragma unroll
      for (int q = 0; q < 9 * NTW; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (q < nreads) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
        } else {
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
      }
    } else if (WT_SGB) {
#pragma unroll
      for (int q = 0; q < 9 * NTW; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  unsigned nxa = 0;   // address of the next phase's X reads
  // one K step: rows 0 .. 2 of step `tc` from (x0, yh, yl); row 0 and the dY fragments of step `tn` are requested with
  // the last row's MFMAs; on entry nxa = xaddr(tc, 1), on exit nxa = xaddr(tn, 1)
  auto step = [&](const XF& x0, const uint4 (&yh)[NTW], const uint4 (&yl)[NTW], XF& xn, uint4 (&nh)[NTW], uint4 (&nl)[NTW],
                  const TB& tc, const TB& tn) {
    XF x1 = {}, x2 = {};
    if (WT_SGB == 3 && NTW == 2) {
      uint4 ah[3], al[3];
      {
        const unsigned a = nxa;
        shifts(x0, ah, al);
#pragma unroll
        for (int q = 0; q < 18; ++q) {
          mm1(ah, al, yh, yl, acc[0], q);
          if (q < 6) xread1(x1, a, q);
          __builtin_amdgcn_sched_barrier(0x6);
        }
        asm volatile("" ::"v"(x0.xh), "v"(x0.xl));
        nxa = xaddr(tc, 2);
        __builtin_amdgcn_sched_barrier(0);
      }
      {
        const unsigned a = nxa;
        shifts(x1, ah, al);
#pragma unroll
        for (int q = 0; q < 18; ++q) {
          mm1(ah, al, yh, yl, acc[1], q);
          if (q < 6) xread1(x2, a, q);
          __builtin_amdgcn_sched_barrier(0x6);
        }
        asm volatile("" ::"v"(x1.xh), "v"(x1.xl));
        nxa = xaddr(tn, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      {
        const unsigned a = nxa, ya = yaddr(tn);
        shifts(x2, ah, al);
#pragma unroll
        for (int q = 0; q < 18; ++q) {
          mm1(ah, al, yh, yl, acc[2], q);
          if (q < 2) xread1(xn, a, q);            // lo halves of the next step's first row,
          else if (q < 10) yread1(nh, nl, ya, q - 2);   // its dY fragments,
          else if (q < 14) xread1(xn, a, q - 8);  // the rest of the row
          __builtin_amdgcn_sched_barrier(0x6);
        }
        asm volatile("" ::"v"(x2.xh), "v"(x2.xl));
        nxa = xaddr(tn, 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      return;
    }
    xissue(x1, nxa);
    if (WT_SGB != 2) __builtin_amdgcn_sched_barrier(0);
    rowmm(x0, yh, yl, acc[0]);
    nxa = xaddr(tc, 2);
    mm_region(6);
    xissue(x2, nxa);
    if (WT_SGB != 2) __builtin_amdgcn_sched_barrier(0);
    rowmm(x1, yh, yl, acc[1]);
    nxa = xaddr(tn, 0);
    const unsigned nya = yaddr(tn);
    mm_region(6);
    xissue(xn, nxa);
    yissue(nh, nl, nya);
    if (WT_SGB != 2) __builtin_amdgcn_sched_barrier(0);
    rowmm(x2, yh, yl, acc[2]);
    nxa = xaddr(tn, 1);
    mm_region(14);
  };
  if (WB_PRIO) __builtin_amdgcn_s_setprio(WB_PRIO);
  __syncthreads();   // tile 0 staged
  const long long clk_first = WB_CLK();   // tile 0 staged: everything before is launch + first-tile latency
  long long clk_loops = clk_first;
  if (nks_run > 0) {
    const TB tb0 = tab(0), tb1 = tab(1);   // the first two K steps' table words do not depend on the tile
    int wprev = P.HH;
    long long wk_loop = 0, wk_wait = 0, wk_pro = 0;
    for (int it = 0; it < ntb; ++it) {
      const long long k0 = WB_CLK();
      wprev = ring_next(wprev, it == 0 || ((first_tile + it) % P.tiles_y) == 0);
      rb = wprev;
      ysel = YSET0 + (unsigned)(it & 1) * YSET;
      TB t0 = tb0, t1 = tb1;
      XF xa = {}, xb = {};
      uint4 yah[NTW] = {}, yal[NTW] = {}, ybh[NTW] = {}, ybl[NTW] = {};
      xissue(xa, xaddr(t0, 0));
      yissue(yah, yal, yaddr(t0));
      nxa = xaddr(t0, 1);
      const long long k1 = WB_CLK();
      int ks = 0;
      for (; ks + 1 < nks_run; ks += 2) {
        const TB t2 = tab(ks + 2);
        step(xa, yah, yal, xb, ybh, ybl, t0, t1);
        const TB t3 = tab(ks + 3);
        step(xb, ybh, ybl, xa, yah, yal, t1, t2);
        t0 = t2;
        t1 = t3;
      }
      if (ks < nks_run) step(xa, yah, yal, xb, ybh, ybl, t0, t1);
      const long long k2 = WB_CLK();
      __syncthreads();
      wk_pro += k1 - k0;
      wk_loop += k2 - k0;
      wk_wait += WB_CLK() - k2;
    }
    clk_loops = WB_CLK();
#ifdef SRK_EXPERIMENTS
    if (P.prof && tid0 == 0) {
      long long* pr = P.prof + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + 8;
      pr[0] = wk_loop; pr[1] = wk_wait; pr[2] = ntb; pr[3] = (long long)ntb * P.nks; pr[4] = wk_pro;
      pr[5] = clk_first - clk_begin;   // launch of the block .. its first tile is staged
    }
#endif
    (void)wk_loop; (void)wk_wait; (void)wk_pro;
  } else {
    for (int it = 0; it < ntb; ++it) __syncthreads();
  }
  __syncthreads();   // (the stagers' bias reduction)
  // partial slab ws[g][t][ci][co]; C/D layout: col = lane & 15 (ci), row = (lane >> 4) * 4 + reg (co)
  float* slab = P.ws + (size_t)bxl * 9 * P.Cin * P.Cout + (size_t)(cib + cit * 16 + r16) * P.Cout + cob + cow * (NTW * 16) + kq * 4;
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      const int t = u * 3 + v;
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
        *reinterpret_cast<f32x4*>(slab + (size_t)t * P.Cin * P.Cout + nt * 16) = acc[u][v][nt];
    }
#ifdef SRK_EXPERIMENTS
  if (P.prof && tid0 == 0) {
    __builtin_amdgcn_s_waitcnt(0);   // (vmcnt(0): the slab stores have left)
    long long* pr = P.prof + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + 8;
    pr[6] = WB_CLK() - clk_loops;   // last tile done .. slab stored
    pr[7] = WB_CLK() - clk_begin;   // the block's life
  }
#endif
  (void)clk_begin; (void)clk_loops;
}

// ---------------------------------------------------------------------------------------------
// Stride-2 3x3 pad-1 weight gradient (SRGAN's discriminator, srgan.py:57,59,61,63: conv -> BatchNorm, no fused
// activation) on the bf16 matrix cores.  Until round 4 these four layers ran the exact-fp32 kernel of conv_wgrad_mfma.hip:
// 12 launches x 100 us = 8 % of the adversarial step, for 4.8 GFLOP each.
//
//   dW[u][v][ci][co] = sum over output pixels (r, c) of  X[2r + u - 1][2c + v - 1][ci] * dY[r][c][co]
//
// Same GEMM view as k_wgrad_bf (M = ci, N = co, K = output pixels, 8 consecutive pixels of a tile row per fragment), but
// the 8 input pixels a fragment needs are two columns apart.  The X halo of a tile is therefore staged DE-INTERLEAVED:
// each halo row holds its even halo columns (E) followed by its odd ones (O), halo column hx = 2 (c - c0) + v:
//   v = 0 -> E[c - c0 ..], v = 1 -> O[c - c0 ..], v = 2 -> E[c - c0 + 1 ..] (the v_alignbit shift of k_wgrad_bf, by one element);
// a staging thread loads four adjacent halo pixels x 4 channels and stores the E pair and the O pair as packed dwords.
// Per-tile kernel (stage, barrier, K loop), two blocks of 4 waves (cit, cow) = 32 ci x 64 co per CU; slabs and the bias
// partials go to the shared deterministic reduce.  Requires Cin % 32 == 0, Cout % 64 == 0, no gradient mask.
// ---------------------------------------------------------------------------------------------
struct WgS2Params {
  const float* x;
  const float* dy;
  float* ws;
  float* bias_partial;
  int N, Cin, Cout, XH, XW, YH, YW;
  int TH, TW, TWo, tiles_y, tiles_x, HH, TWp, CS, DS, ntiles, G, nks, nq;
};
constexpr int WS2_XB = 3;  // register batches of the X staging (4 pixels x 4 channels each) per thread and tile
constexpr int WS2_YB = 2;  // ... of the dY staging (pixel pairs)

__global__ __launch_bounds__(256, 2) void k_wgrad_s2(WgS2Params P) {
  constexpr int CIB = 32, COB = 64, NTW = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
  __shared__ int oct_x[WB_MAXOCT], oct_y[WB_MAXOCT];
  __shared__ float bred[256][4];
  unsigned short* xs = smem16;                              // [2 planes][CIB][CS]: rows of E | O halves
  unsigned short* ys = smem16 + (size_t)2 * CIB * P.CS;     // [2 planes][COB][DS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int cit = wave & 1, cow = wave >> 1;
  const int cib = blockIdx.y * CIB, cob = blockIdx.z * COB;
  const int ROW = 2 * P.TWp;
  const int noct = P.TH * P.TWo;
  for (int o = tid; o < P.nks * 4; o += 256) {
    if (o < noct) {
      const int orow = o / P.TWo, oc = o - orow * P.TWo;
      oct_x[o] = (2 * orow) * ROW + oc * 8;
      oct_y[o] = o * 8;
    } else {  // K padding: multiply by the zero octet appended to every dY plane
      oct_x[o] = 0;
      oct_y[o] = P.TH * P.TW;
    }
  }
  for (int e = tid; e < 2 * COB * 4; e += 256) {  // zero octets (never overwritten by the staging)
    const int pc = e >> 2, w = e & 3;
    reinterpret_cast<unsigned*>(ys + (size_t)pc * P.DS + P.TH * P.TW)[w] = 0u;
  }
  f32x4 acc[3][3][NTW];
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int v = 0; v < 3; ++v)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) acc[u][v][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  const bool want_bias = P.bias_partial != nullptr && blockIdx.y == 0;
  const int tw2 = P.TW >> 1;
  constexpr unsigned OOB = 0x80000000u;
  const size_t ximg = (size_t)P.XH * P.XW * P.Cin, yimg = (size_t)P.YH * P.YW * P.Cout;

  for (int tile = blockIdx.x; tile < P.ntiles; tile += P.G) {
    int b = tile;
    const int txi = b % P.tiles_x;
    b /= P.tiles_x;
    const int tyi = b % P.tiles_y;
    const int n = b / P.tiles_y;
    const int r0 = tyi * P.TH, c0 = txi * P.TW;
    __syncthreads();  // previous tile fully consumed (tables / zero octets visible on the first pass)
    {  // ---- X halo: rows 2 r0 - 1 + hy, columns 2 c0 - 1 + hx; item = (4 adjacent halo pixels, 4 channels)
      const __amdgpu_buffer_rsrc_t rx = wb_rsrc(P.x + (size_t)n * ximg, (unsigned)(ximg * 4));
      const int q = tid & 7, ch = cib + q * 4;
      const int nitems = P.HH * P.nq;
      const int by0 = 2 * r0 - 1, bx0 = 2 * c0 - 1;
      f32x4 px[WS2_XB][4];
#pragma unroll
      for (int k = 0; k < WS2_XB; ++k) {
        const int ip = (tid >> 3) + k * 32;
        const int hy = ip / P.nq, j = ip - hy * P.nq;
        const int iy = by0 + hy, ix = bx0 + 4 * j;
        const int rowoff = ((iy * P.XW + ix) * P.Cin + ch) * 4;   // (rows above / below the image fall outside the descriptor)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = ip < nitems && (unsigned)iy < (unsigned)P.XH && (unsigned)(ix + e) < (unsigned)P.XW;
          px[k][e] = wb_bload(rx, ok ? (unsigned)(rowoff + e * P.Cin * 4) : OOB);
        }
      }
      unsigned short* xq = xs + (size_t)(q * 4) * P.CS;
#pragma unroll
      for (int k = 0; k < WS2_XB; ++k) {
        const int ip = (tid >> 3) + k * 32;
        if (ip < nitems) {
          const int hy = ip / P.nq, j = ip - hy * P.nq;
          unsigned hi[4], lo[4];
          unsigned short* dst = xq + hy * ROW + 2 * j;
          wb_split_pair(px[k][0], px[k][2], hi, lo);   // even halo columns 4 j, 4 j + 2 -> E[2 j], E[2 j + 1]
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            *reinterpret_cast<unsigned*>(dst + (size_t)c * P.CS) = hi[c];
            *reinterpret_cast<unsigned*>(dst + (size_t)(CIB + c) * P.CS) = lo[c];
          }
          wb_split_pair(px[k][1], px[k][3], hi, lo);   // odd halo columns -> O[2 j], O[2 j + 1]
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            *reinterpret_cast<unsigned*>(dst + P.TWp + (size_t)c * P.CS) = hi[c];
            *reinterpret_cast<unsigned*>(dst + P.TWp + (size_t)(CIB + c) * P.CS) = lo[c];
          }
        }
      }
    }
    {  // ---- dY tile: rows [r0, +TH), cols [c0, +TW), channels [cob, +64) -> planes [co][r][c]
      const __amdgpu_buffer_rsrc_t ry = wb_rsrc(P.dy + (size_t)n * yimg, (unsigned)(yimg * 4));
      const int q = tid & 15, ch = cob + q * 4;
      const int npairs = P.TH * tw2;
      f32x4 p0[WS2_YB], p1[WS2_YB];
#pragma unroll
      for (int k = 0; k < WS2_YB; ++k) {
        const int pp = (tid >> 4) + k * 16;
        const int r = pp / tw2, c = (pp - r * tw2) * 2;
        const int iy = r0 + r, ix = c0 + c;
        const unsigned off = (unsigned)(((iy * P.YW + ix) * P.Cout + ch) * 4);
        const bool rowok = pp < npairs && iy < P.YH;
        p0[k] = wb_bload(ry, rowok && ix < P.YW ? off : OOB);
        p1[k] = wb_bload(ry, rowok && ix + 1 < P.YW ? off + (unsigned)P.Cout * 4u : OOB);
      }
      unsigned short* yq = ys + (size_t)(q * 4) * P.DS;
#pragma unroll
      for (int k = 0; k < WS2_YB; ++k) {
        const int pp = (tid >> 4) + k * 16;
        if (pp < npairs) {
          const int r = pp / tw2, c = (pp - r * tw2) * 2;
          bsum += p0[k] + p1[k];
          unsigned hi[4], lo[4];
          wb_split_pair(p0[k], p1[k], hi, lo);
          unsigned short* dst = yq + r * P.TW + c;
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            *reinterpret_cast<unsigned*>(dst + (size_t)cc * P.DS) = hi[cc];
            *reinterpret_cast<unsigned*>(dst + (size_t)(COB + cc) * P.DS) = lo[cc];
          }
        }
      }
    }
    __syncthreads();
    // ---- K loop: per K step the fragments of the three row shifts (E octet + next pair, O octet; both planes), then the
    // MFMAs pass-major over the 3 * NTW accumulators of a row shift
    const unsigned short* xa_h = xs + (size_t)(cit * 16 + i) * P.CS;
    const unsigned short* xa_l = xa_h + (size_t)CIB * P.CS;
    const unsigned short* yb_h = ys + (size_t)(cow * NTW * 16 + i) * P.DS;
    const unsigned short* yb_l = yb_h + (size_t)COB * P.DS;
    for (int ks = 0; ks < P.nks; ++ks) {
      const int ox = oct_x[ks * 4 + kq], oy = oct_y[ks * 4 + kq];
      uint4 bh[NTW], bl[NTW], eh4[3], el4[3], oh4[3], ol4[3];
      unsigned eh[3], el[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int xo = ox + u * ROW;
        eh4[u] = *reinterpret_cast<const uint4*>(xa_h + xo);
        el4[u] = *reinterpret_cast<const uint4*>(xa_l + xo);
        eh[u] = *reinterpret_cast<const unsigned*>(xa_h + xo + 8);
        el[u] = *reinterpret_cast<const unsigned*>(xa_l + xo + 8);
        oh4[u] = *reinterpret_cast<const uint4*>(xa_h + xo + P.TWp);
        ol4[u] = *reinterpret_cast<const uint4*>(xa_l + xo + P.TWp);
      }
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        bh[nt] = *reinterpret_cast<const uint4*>(yb_h + (size_t)nt * 16 * P.DS + oy);
        bl[nt] = *reinterpret_cast<const uint4*>(yb_l + (size_t)nt * 16 * P.DS + oy);
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        uint4 ah[3], al[3];
        ah[0] = eh4[u];
        al[0] = el4[u];
        ah[1] = oh4[u];
        al[1] = ol4[u];
        ah[2] = make_uint4(__builtin_amdgcn_alignbit(eh4[u].y, eh4[u].x, 16), __builtin_amdgcn_alignbit(eh4[u].z, eh4[u].y, 16),
                           __builtin_amdgcn_alignbit(eh4[u].w, eh4[u].z, 16), __builtin_amdgcn_alignbit(eh[u], eh4[u].w, 16));
        al[2] = make_uint4(__builtin_amdgcn_alignbit(el4[u].y, el4[u].x, 16), __builtin_amdgcn_alignbit(el4[u].z, el4[u].y, 16),
                           __builtin_amdgcn_alignbit(el4[u].w, el4[u].z, 16), __builtin_amdgcn_alignbit(el[u], el4[u].w, 16));
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) acc[u][v][nt] = wb_mfma(al[v], bh[nt], acc[u][v][nt]);
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) acc[u][v][nt] = wb_mfma(ah[v], bl[nt], acc[u][v][nt]);
#pragma unroll
        for (int v = 0; v < 3; ++v)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) acc[u][v][nt] = wb_mfma(ah[v], bh[nt], acc[u][v][nt]);
      }
    }
  }
  if (want_bias) {  // column sums of dY: combine the threads that staged the same channel group
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) bred[tid][e] = bsum[e];
    __syncthreads();
    if (tid < COB) {
      const int q = tid >> 2, e = tid & 3;
      float sacc = 0.f;
      for (int t = q; t < 256; t += 16) sacc += bred[t][e];
      P.bias_partial[(size_t)blockIdx.x * P.Cout + cob + tid] = sacc;
    }
  }
  // partial slab ws[g][t][ci][co]; C/D layout: col = lane&15 (co), row = (lane>>4)*4 + reg (ci)
  float* slab = P.ws + (size_t)blockIdx.x * 9 * P.Cin * P.Cout;
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      const int t = u * 3 + v;
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int co = cob + (cow * NTW + nt) * 16 + i;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int ci = cib + cit * 16 + kq * 4 + reg;
          slab[((size_t)t * P.Cin + ci) * P.Cout + co] = acc[u][v][nt][reg];
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------
// Host
// ---------------------------------------------------------------------------------------------
static constexpr int kWbLdsBudget = 74 * 1024;  // + ~4.6 KB static (tables, bias reduction): 2 blocks per CU

struct WbPlan {
  bool ok;
  int cfg;  // 0: 32 ci x 64 co (CIT 2, COW 2, NTW 2); 1: 64 ci x 32 co (4,1,2); 2: 64 ci x 16 co (4,1,1)
  int CIB, COB;
  int TH, TW, TWo, tiles_y, tiles_x, HH, HWp, CS, DS, nks;
  size_t lds;
  int G, ntiles, gy, gz;
};

static inline int round_8odd(int v) {  // smallest multiple of 8 >= v whose quotient by 8 is odd
  int q = (v + 7) / 8;
  if ((q & 1) == 0) ++q;
  return q * 8;
}

static WbPlan wb_plan(const srk_conv_desc& d) {
  WbPlan pl{};
  pl.ok = false;
  if (d.transposed || d.stride != 1 || d.KH > 3 || d.KW > 3 || d.Cin < 8 || d.Cout < 1) return pl;
  if ((long)d.H * d.W * d.Cin >= (1L << 30) || (long)d.OH * d.OW * d.Cout >= (1L << 30)) return pl;  // 32-bit in-image offsets
  if (d.dy_ps_r > 1 && (d.Cout % (d.dy_ps_r * d.dy_ps_r) != 0 || (d.Cout / (d.dy_ps_r * d.dy_ps_r)) % 4 != 0)) return pl;
  if (d.Cout > 32) {
    pl.cfg = 0; pl.CIB = 32; pl.COB = 64;
  } else if (d.Cout > 16) {
    pl.cfg = 1; pl.CIB = 64; pl.COB = 32;
  } else {
    pl.cfg = 2; pl.CIB = 64; pl.COB = 16;
  }
  // tile = TH rows x TWo octets (8 pixels each).  Search the shapes that fit LDS for the one with the most useful
  // pixels per padded K step (e.g. 41-wide VDSR patches: 2 x 48 -> 83 % instead of 4 x 32 -> 60 %); ties -> taller
  // tiles (less halo per pixel).
  double best_eff = -1.0;
  const char* tile_env = SRK_EXP_STR("SRK_WG_TILE");  // experiment: "TH,TWo" forces the tile shape
  int force_th = 0, force_two = 0;
  if (tile_env) sscanf(tile_env, "%d,%d", &force_th, &force_two);
  for (int TWo = 1; TWo <= 6 && (TWo - 1) * 8 < d.OW; ++TWo) {
    const int TW = TWo * 8;
    int TH = 16 / TWo;  // <= 128 pixels per tile
    if (TH > d.OH) TH = d.OH;
    if (force_two) {
      if (TWo != force_two) continue;
      TH = force_th < d.OH ? force_th : d.OH;
    }
    for (; TH >= 1; --TH) {
      const int HH = TH + d.KH - 1, HWp = TW + 8;
      const int CS = round_8odd(HH * HWp), DS = round_8odd(TH * TW + 8);
      const size_t lds = ((size_t)2 * pl.CIB * CS + (size_t)2 * pl.COB * DS) * 2;
      if (lds > (size_t)kWbLdsBudget || TH * TWo > WB_MAXOCT) continue;
      const int nks = cdiv(TH * TWo, 4);
      const double tiles = (double)cdiv(d.OH, TH) * cdiv(d.OW, TW);
      // cost per tile: nks K steps + staging, calibrated on the VDSR / EDSR body layers (ablation: staging one
      // channel-pixel costs 1/3136 of a K step; only TW + KW - 1 halo columns are loaded)
      const double cost = tiles * (nks + ((double)HH * (TW + d.KW - 1) * pl.CIB + (double)TH * TW * pl.COB) / 3136.0);
      const double eff = (double)d.OH * d.OW / cost;
      if (eff > best_eff * 1.02 || (eff > best_eff * 0.98 && eff > 0 && TH > pl.TH)) {
        if (eff > best_eff) best_eff = eff;
        pl.TWo = TWo; pl.TW = TW; pl.TH = TH; pl.HH = HH; pl.HWp = HWp; pl.CS = CS; pl.DS = DS; pl.lds = lds;
      }
      break;  // smaller TH only gets worse for this width
    }
  }
  if (best_eff < 0) return pl;
  pl.tiles_x = cdiv(d.OW, pl.TW);
  pl.tiles_y = cdiv(d.OH, pl.TH);
  pl.nks = cdiv(pl.TH * pl.TWo, 4);
  const long nt = (long)d.N * pl.tiles_y * pl.tiles_x;
  if (nt > (1L << 30)) return pl;
  pl.ntiles = (int)nt;
  pl.gy = cdiv(d.Cin, pl.CIB);
  pl.gz = cdiv(d.Cout, pl.COB);
  int blocks_per_cu = SRK_EXP_INT("SRK_WG_BLOCKS", 2);  // experiment: resident blocks per CU (slab count vs overlap)
  if (blocks_per_cu < 1) blocks_per_cu = 1;
  int g = (blocks_per_cu * kNumCU) / (pl.gy * pl.gz);
  if (g < 1) g = 1;
  pl.G = pl.ntiles < g ? pl.ntiles : g;
  pl.ok = true;
  return pl;
}

// SPEC stagers prefetch one tile ahead when a tile's pixel pairs fit their register batches (WB_PIT x 512 items per tensor)
static int wb_prefetch_ok(const WbPlan& pl, const srk_conv_desc& d) {
  const int env = env_int("SRK_WGRAD_PREFETCH", 1);
  if (!env) return 0;
  const long x_items = (long)pl.HH * ((pl.TW + d.KW) >> 1) * (pl.CIB / 4);
  const long y_items = (long)pl.TH * (pl.TW >> 1) * (pl.COB / 4);
  // the prefetching stagers load through per-image buffer descriptors with 32-bit byte offsets
  const long ximg = (long)d.H * d.W * d.Cin * 4, yimg = (long)d.OH * d.OW * d.Cout * 4;
  return x_items <= (long)WB_PIT * WB_SST && y_items <= (long)WB_PIT * WB_SST && ximg < (1L << 31) && yimg < (1L << 31);
}

// Ring mode of the wave-specialised kernel (WgBfParams.ring): plane stride of the 2 * HH-row ring and the LDS it needs
// (ring + two dY buffer sets); 0 when it does not apply.  SRK_WG_RING=0: off.
static size_t wb_ring_setup(const WbPlan& pl, bool spec, int prefetch, int& cs_ring) {
  const int ring_env = env_int("SRK_WG_RING", 1);
  if (!ring_env || !spec || !prefetch) return 0;
  cs_ring = round_8odd(2 * pl.HH * pl.HWp);
  const size_t bytes = ((size_t)2 * pl.CIB * cs_ring + (size_t)2 * 2 * pl.COB * pl.DS) * 2;
  return bytes + 8 * 1024 <= 160 * 1024 ? bytes : 0;
}

// ---- stride-2 plan (k_wgrad_s2) ----
struct Ws2Plan {
  bool ok;
  int TH, TW, TWo, tiles_y, tiles_x, HH, TWp, CS, DS, nks, nq, ntiles, G, gy, gz;
  size_t lds;
};

static Ws2Plan ws2_plan(const srk_conv_desc& d) {
  Ws2Plan pl{};
  pl.ok = false;
  if (env_int("SRK_WGRAD_S2", 1) == 0) return pl;
  if (d.transposed || d.stride != 2 || d.KH != 3 || d.KW != 3 || d.pad != 1 || d.dy_ps_r > 1) return pl;
  if (d.Cin % 32 != 0 || d.Cout % 64 != 0) return pl;
  if ((long)d.H * d.W * d.Cin * 4 >= (1L << 31) || (long)d.OH * d.OW * d.Cout * 4 >= (1L << 31)) return pl;  // buffer descriptors
  int best_px = 0;
  for (int TWo = 1; TWo <= 4 && (TWo - 1) * 8 < d.OW; TWo *= 2) {
    const int TW = TWo * 8;
    int TH = 16 / TWo;
    if (TH > d.OH) TH = d.OH;
    for (; TH >= 1; --TH) {
      const int HH = 2 * TH + 1, TWp = TW + 8;
      const int CS = round_8odd(HH * 2 * TWp), DS = round_8odd(TH * TW + 8);
      const size_t lds = ((size_t)2 * 32 * CS + (size_t)2 * 64 * DS) * 2;
      const int nq = (2 * TW + 1 + 3) / 4;
      if (lds > (size_t)kWbLdsBudget || HH * nq > 32 * WS2_XB || TH * (TW / 2) > 16 * WS2_YB || TH * TWo > WB_MAXOCT) continue;
      // useful pixels per padded tile, then the larger tile
      const double eff = (double)d.OH * d.OW / ((double)cdiv(d.OH, TH) * cdiv(d.OW, TW) * cdiv(TH * TWo, 4) * 32);
      const int score = (int)(eff * 1000.0) * 1000 + TH * TW;
      if (score > best_px) {
        best_px = score;
        pl.TH = TH; pl.TW = TW; pl.TWo = TWo; pl.HH = HH; pl.TWp = TWp; pl.CS = CS; pl.DS = DS; pl.lds = lds; pl.nq = nq;
      }
      break;
    }
  }
  if (!best_px) return pl;
  pl.tiles_x = cdiv(d.OW, pl.TW);
  pl.tiles_y = cdiv(d.OH, pl.TH);
  pl.nks = cdiv(pl.TH * pl.TWo, 4);
  const long nt = (long)d.N * pl.tiles_y * pl.tiles_x;
  if (nt > (1L << 30)) return pl;
  pl.ntiles = (int)nt;
  pl.gy = d.Cin / 32;
  pl.gz = d.Cout / 64;
  int g = (2 * kNumCU) / (pl.gy * pl.gz);
  if (g < 1) g = 1;
  pl.G = pl.ntiles < g ? pl.ntiles : g;
  pl.ok = true;
  return pl;
}

bool conv_wgrad_s2_supported(const srk_conv_desc& d, const float* x, const float* dy, const srk_bwd_mask* mask) {
  if (mask && mask->y) return false;
  if ((((uintptr_t)x | (uintptr_t)dy) & 15) != 0) return false;
  return ws2_plan(d).ok;
}

size_t conv_wgrad_s2_ws(const srk_conv_desc& d) {
  const Ws2Plan pl = ws2_plan(d);
  if (!pl.ok) return 0;
  return (size_t)pl.G * 9 * d.Cin * d.Cout * sizeof(float) + (size_t)pl.G * d.Cout * sizeof(float) + 256;
}

int conv_wgrad_s2(const srk_conv_desc& d, const float* x, const float* dy, float* dw, float* db, float beta, void* ws,
                  size_t ws_bytes, hipStream_t s) {
  const Ws2Plan pl = ws2_plan(d);
  if (!pl.ok) {
    set_error("conv_wgrad_s2: shape not covered");
    return SRK_ERR_UNSUPPORTED;
  }
  if (ws_bytes < conv_wgrad_s2_ws(d)) {
    set_error("conv_wgrad_s2: workspace too small (%zu < %zu)", ws_bytes, conv_wgrad_s2_ws(d));
    return SRK_ERR_WORKSPACE;
  }
  const size_t slab_bytes = ((size_t)pl.G * 9 * d.Cin * d.Cout * sizeof(float) + 255) & ~(size_t)255;
  WgS2Params P{};
  P.x = x; P.dy = dy;
  P.ws = static_cast<float*>(ws);
  float* bias_ws = reinterpret_cast<float*>(static_cast<char*>(ws) + slab_bytes);
  P.bias_partial = db ? bias_ws : nullptr;
  P.N = d.N; P.Cin = d.Cin; P.Cout = d.Cout; P.XH = d.H; P.XW = d.W; P.YH = d.OH; P.YW = d.OW;
  P.TH = pl.TH; P.TW = pl.TW; P.TWo = pl.TWo; P.tiles_y = pl.tiles_y; P.tiles_x = pl.tiles_x; P.HH = pl.HH; P.TWp = pl.TWp;
  P.CS = pl.CS; P.DS = pl.DS; P.ntiles = pl.ntiles; P.G = pl.G; P.nks = pl.nks; P.nq = pl.nq;
  static LdsLimit lim;
  lim.ensure(reinterpret_cast<const void*>(&k_wgrad_s2), pl.lds);
  hipLaunchKernelGGL(k_wgrad_s2, dim3(pl.G, pl.gy, pl.gz), dim3(256), pl.lds, s, P);
  const int rc = check_launch("conv_wgrad_s2");
  if (rc) return rc;
  return conv_wgrad_reduce_launch((const float*)ws, dw, pl.G, d.Cout, d.Cin, 3, 3, 0, beta, db ? bias_ws : nullptr, db, d.Cout,
                                  0, s);
}

// ---- k_wgrad_tr (pixel-major LDS image + transpose reads): eligibility, LDS layout, launch.  SRK_WGRAD_TR=0: never.
static bool wt_setup(WgBfParams& P, const srk_conv_desc& d, const WbPlan& pl, bool spec, size_t& lds_bytes) {
  if (!spec || pl.cfg != 0 || !env_int("SRK_WGRAD_TR", 1)) return false;
  if (d.KH != 3 || d.KW != 3 || d.Cin % 32 != 0 || d.Cout % 64 != 0 || !P.vec_x || !P.vec_y) return false;
  if (d.dy_ps_r > 1 && (d.Cout / (d.dy_ps_r * d.dy_ps_r)) % 64 != 0) return false;
  const long ximg = (long)d.H * d.W * d.Cin * 4, yimg = (long)d.OH * d.OW * d.Cout * 4;
  if (ximg >= (1L << 31) || yimg >= (1L << 31)) return false;   // per-image buffer descriptors, 32-bit byte offsets
  if ((long)pl.HH * (pl.TW + 2) * 4 > 1024 || (long)pl.TH * pl.TW * 8 > 1024) return false;  // two items per stager and tensor
  const int XP = pl.TW + 4;   // halo columns 0 .. TW + 1 are staged; the third transpose read of a row touches TW + 3
  const size_t XPL = (size_t)2 * pl.HH * XP * 64, YPL = (size_t)(pl.TH * pl.TW + 8) * 128;
  const size_t bytes = 2 * XPL + 4 * YPL;
  if (bytes + 1024 > 160 * 1024 || 2 * XPL < (size_t)WB_SST * 9 * 4 || XPL + 512 >= 65536 || YPL + 1024 >= 65536) return false;
  P.XP = XP;
  P.XPL = (int)XPL;
  P.YPL = (int)YPL;
  lds_bytes = bytes;
  return true;
}
template <bool GRP>
static void wt_launch(const WgBfParams& P, const typename WgGroupArg<GRP>::type& GR, dim3 grid, size_t lds, hipStream_t s) {
  if (env_int("SRK_WGRAD_TR_W8", 0)) {   // eight working waves (two per SIMD)
    static LdsLimit lim8;
    lim8.ensure(reinterpret_cast<const void*>(&k_wgrad_tr<GRP, 1>), lds);
    note_kernel("k_wgrad_tr<%s,w8>", GRP ? "grouped" : "single");
    hipLaunchKernelGGL((k_wgrad_tr<GRP, 1>), grid, dim3(1024), lds, s, P, GR);
    return;
  }
  static LdsLimit lim;
  lim.ensure(reinterpret_cast<const void*>(&k_wgrad_tr<GRP, 2>), lds);
  note_kernel("k_wgrad_tr<%s>", GRP ? "grouped" : "single");
  hipLaunchKernelGGL((k_wgrad_tr<GRP, 2>), grid, dim3(768), lds, s, P, GR);
}

bool conv_wgrad_bf_supported(const srk_conv_desc& d) { return wb_plan(d).ok; }

size_t conv_wgrad_bf_ws(const srk_conv_desc& d) {
  WbPlan pl = wb_plan(d);
  if (!pl.ok) return 0;
  return (size_t)pl.G * d.KH * d.KW * d.Cin * d.Cout * sizeof(float) + conv_bias_grad_ws(d);
}

static bool wb_k33(const WgBfParams& P) {
  const int env = env_int("SRK_WG_K33", 1);
  return env && P.KH == 3 && P.KW == 3;
}

template <int CIT, int COW, int NTW>
static void wb_launch(const WgBfParams& P, dim3 grid, size_t lds, bool spec, hipStream_t s) {
  note_kernel("k_wgrad_bf<%d,%d,%d,%s>", CIT, COW, NTW, spec ? "spec" : "tile");
  if (spec && wb_k33(P)) {
    static LdsLimit lim3;
    lim3.ensure(reinterpret_cast<const void*>(&k_wgrad_bf<CIT, COW, NTW, true, false, true>), 2 * lds);
    hipLaunchKernelGGL((k_wgrad_bf<CIT, COW, NTW, true, false, true>), grid, dim3(64 * CIT * COW + WB_SST), 2 * lds, s, P, WgNoGroup{0});
    return;
  }
  if (spec) {
    static LdsLimit lim2;
    lim2.ensure(reinterpret_cast<const void*>(&k_wgrad_bf<CIT, COW, NTW, true, false>), 2 * lds);
    hipLaunchKernelGGL((k_wgrad_bf<CIT, COW, NTW, true, false>), grid, dim3(64 * CIT * COW + WB_SST), 2 * lds, s, P, WgNoGroup{0});
    return;
  }
  static LdsLimit lim;
  lim.ensure(reinterpret_cast<const void*>(&k_wgrad_bf<CIT, COW, NTW, false, false>), lds);
  hipLaunchKernelGGL((k_wgrad_bf<CIT, COW, NTW, false, false>), grid, dim3(64 * CIT * COW), lds, s, P, WgNoGroup{0});
}

template <int CIT, int COW, int NTW>
static void wb_launch_grouped(const WgBfParams& P, const WgGroup& GR, dim3 grid, size_t lds, bool spec, hipStream_t s) {
  note_kernel("k_wgrad_bf<%d,%d,%d,%s,grouped>", CIT, COW, NTW, spec ? "spec" : "tile");
  if (spec && wb_k33(P)) {
    static LdsLimit lim3;
    lim3.ensure(reinterpret_cast<const void*>(&k_wgrad_bf<CIT, COW, NTW, true, true, true>), 2 * lds);
    hipLaunchKernelGGL((k_wgrad_bf<CIT, COW, NTW, true, true, true>), grid, dim3(64 * CIT * COW + WB_SST), 2 * lds, s, P, GR);
    return;
  }
  if (spec) {
    static LdsLimit lim2;
    lim2.ensure(reinterpret_cast<const void*>(&k_wgrad_bf<CIT, COW, NTW, true, true>), 2 * lds);
    hipLaunchKernelGGL((k_wgrad_bf<CIT, COW, NTW, true, true>), grid, dim3(64 * CIT * COW + WB_SST), 2 * lds, s, P, GR);
    return;
  }
  static LdsLimit lim;
  lim.ensure(reinterpret_cast<const void*>(&k_wgrad_bf<CIT, COW, NTW, false, true>), lds);
  hipLaunchKernelGGL((k_wgrad_bf<CIT, COW, NTW, false, true>), grid, dim3(64 * CIT * COW), lds, s, P, GR);
}

// dw_l (torch layout [co][ci][kh][kw]) = beta*dw_l + sum_g ws[l][g][t][ci][co] and db_l likewise, for every layer of
// a grouped launch (blockIdx.y = layer): the single-layer k_wgrad_reduce of conv_wgrad_mfma.hip with a layer axis.
__global__ __launch_bounds__(256) void k_wgrad_reduce_grouped(const float* __restrict__ ws, WgGroupOut O, int G, int Cout,
                                                              int Cin, int KH, int KW, float beta,
                                                              const float* __restrict__ bias_partial) {
  // (256 slab elements per block, four per lane as 16-byte loads; per element the summation order of the 64-element form)
  typedef float r4 __attribute__((ext_vector_type(4)));
  constexpr int EPB = 256;
  __shared__ __attribute__((aligned(16))) float sm[4][EPB];
  const int elems = KH * KW * Cin * Cout;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nwb = (elems + EPB - 1) / EPB;
  const int layer = blockIdx.y;
  float* __restrict__ dw = O.L[layer].dw;
  float* __restrict__ db = O.L[layer].db;
  if ((int)blockIdx.x >= nwb) {
    if (!db || !bias_partial) return;
    const float* bp = bias_partial + (size_t)layer * G * Cout;
    const int co = ((int)blockIdx.x - nwb) * 64 + lane;
    float t0 = 0.f, t1 = 0.f;
    if (co < Cout) {
      int g = w;
      for (; g + 4 < G; g += 8) {
        t0 += bp[(size_t)g * Cout + co];
        t1 += bp[(size_t)(g + 4) * Cout + co];
      }
      for (; g < G; g += 4) t0 += bp[(size_t)g * Cout + co];
    }
    sm[w][lane] = t0 + t1;
    __syncthreads();
    if (w == 0 && co < Cout) {
      const float t = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
      db[co] = beta != 0.f ? beta * db[co] + t : t;
    }
    return;
  }
  const float* wl = ws + (size_t)layer * G * elems;
  const int e0 = blockIdx.x * EPB + lane * 4;
  const bool vec = (elems & 3) == 0 && (reinterpret_cast<uintptr_t>(wl) & 15) == 0;
  auto load4 = [&](int g) -> r4 {
    const float* p = wl + (size_t)g * elems + e0;
    if (vec) return *reinterpret_cast<const r4*>(p);
    r4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (e0 + k < elems) v[k] = p[k];
    return v;
  };
  r4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
  if (e0 < elems) {
    int g = w;
    for (; g + 4 < G; g += 8) {
      s0 += load4(g);
      s1 += load4(g + 4);
    }
    for (; g < G; g += 4) s0 += load4(g);
  }
  *reinterpret_cast<r4*>(&sm[w][lane * 4]) = s0 + s1;
  __syncthreads();
  const int t = threadIdx.x, e = blockIdx.x * EPB + t;
  if (e >= elems) return;
  const float v = (sm[0][t] + sm[1][t]) + (sm[2][t] + sm[3][t]);
  const int co = e % Cout;
  const int ci = (e / Cout) % Cin;
  const int tap = e / (Cout * Cin);
  const int kh = tap / KW, kw = tap - kh * KW;
  const size_t o = (((size_t)co * Cin + ci) * KH + kh) * KW + kw;
  dw[o] = beta != 0.f ? beta * dw[o] + v : v;
}

int conv_wgrad_bf(const srk_conv_desc& d, const float* x, const float* dy, const srk_bwd_mask* mask, float* dw,
                  float* db, float beta, void* ws, size_t ws_bytes, hipStream_t s) {
  WbPlan pl = wb_plan(d);
  if (!pl.ok) {
    set_error("conv_wgrad_bf: shape not covered");
    return SRK_ERR_UNSUPPORTED;
  }
  const size_t slab_bytes = (size_t)pl.G * d.KH * d.KW * d.Cin * d.Cout * sizeof(float);
  const size_t need = slab_bytes + conv_bias_grad_ws(d);
  if (!ws || ws_bytes < need) {
    set_error("conv_wgrad_bf: workspace %zu < %zu", ws_bytes, need);
    return SRK_ERR_WORKSPACE;
  }
  WgBfParams P{};
  P.x = x; P.dy = dy; P.mask_y = mask ? mask->y : nullptr; P.mask_slope = mask ? mask->slope : 0.f;
  P.ws = (float*)ws;
  float* bias_ws = reinterpret_cast<float*>(static_cast<char*>(ws) + slab_bytes);
  P.bias_partial = db ? bias_ws : nullptr;
  P.N = d.N; P.Cin = d.Cin; P.Cout = d.Cout;
  P.XH = d.H; P.XW = d.W; P.YH = d.OH; P.YW = d.OW;
  P.KH = d.KH; P.KW = d.KW; P.pad = d.pad;
  P.TH = pl.TH; P.TW = pl.TW; P.TWo = pl.TWo; P.tiles_y = pl.tiles_y; P.tiles_x = pl.tiles_x;
  P.HH = pl.HH; P.HWp = pl.HWp; P.CS = pl.CS; P.DS = pl.DS;
  P.ntiles = pl.ntiles; P.G = pl.G; P.nks = pl.nks;
  P.vec_x = (d.Cin % 4 == 0) && ((uintptr_t)x % 16 == 0);
  P.vec_y = (d.Cout % 4 == 0) && ((uintptr_t)dy % 16 == 0) && (!P.mask_y || (uintptr_t)P.mask_y % 16 == 0);
  P.dy_ps_r = d.dy_ps_r > 1 ? d.dy_ps_r : 0;
  P.dy_ps_C = d.dy_ps_r > 1 ? d.Cout / (d.dy_ps_r * d.dy_ps_r) : d.Cout;
  P.prefetch = wb_prefetch_ok(pl, d) && P.vec_x && P.vec_y;  // 16-byte channel groups only
  // wave-specialised variant: one 512-thread block per CU with two LDS buffer sets, when every block has >= 2 tiles
  // to pipeline (SRK_WGRAD_SPEC=0: never)
  const int spec_env = env_int("SRK_WGRAD_SPEC", 1);
  int G = pl.G;
  bool spec = false;
  if (spec_env && 2 * pl.lds + 8 * 1024 <= 160 * 1024) {
    int g1 = kNumCU / (pl.gy * pl.gz);
    if (g1 < 1) g1 = 1;
    if (pl.ntiles >= 2 * g1) {
      spec = true;
      G = g1;
    }
  }
  P.G = G;
  dim3 grid(G, pl.gy, pl.gz);
  size_t lds_half = pl.lds;   // (the specialised launch asks for 2 x this)
  {
    int cs_ring = 0;
    const size_t ring_bytes = wb_ring_setup(pl, spec, P.prefetch, cs_ring);
    if (ring_bytes) {
      P.ring = 1;
      P.CS = cs_ring;
      lds_half = (ring_bytes + 1) / 2;
    }
  }
  {
    const int dbg = SRK_EXP_INT("SRK_DBG", 0);
    P.dbg = dbg;
#ifdef SRK_EXPERIMENTS
    P.prof = g_wb_prof;
#endif
    if (dbg & 32)
      fprintf(stderr, "[srk] k_wgrad_bf cfg %d%s%s: tile %d x %d (%d K steps), %d tiles over %d x %d x %d blocks, lds %zu B\n",
              pl.cfg, spec ? " (wave-specialised)" : "", P.ring ? " (X ring)" : "", pl.TH, pl.TW, pl.nks, pl.ntiles, G, pl.gy,
              pl.gz, spec ? 2 * lds_half : pl.lds);
  }
  size_t tr_lds = 0;
  switch (pl.cfg) {
    case 0:
#ifdef SRK_EXPERIMENTS
      // experiment (SRK_WG_W8=1): eight working waves (2 ci tiles x 4 single-tile co columns) -- measured SLOWER: see k_wgrad_bf
      if (spec && wb_k33(P) && SRK_EXP_INT("SRK_WG_W8", 0)) { wb_launch<2, 4, 1>(P, grid, lds_half, spec, s); break; }
#endif
      if (wt_setup(P, d, pl, spec, tr_lds)) { wt_launch<false>(P, WgNoGroup{0}, grid, tr_lds, s); break; }
      wb_launch<2, 2, 2>(P, grid, lds_half, spec, s);
      break;
    case 1: wb_launch<4, 1, 2>(P, grid, lds_half, spec, s); break;
    default: wb_launch<4, 1, 1>(P, grid, lds_half, spec, s); break;
  }
  int rc = check_launch("conv_wgrad_bf");
  if (rc) return rc;
  return conv_wgrad_reduce_launch((const float*)ws, dw, G, d.Cout, d.Cin, d.KH, d.KW, 0, beta, db ? bias_ws : nullptr,
                                  db, d.Cout, d.dy_ps_r > 1 ? d.dy_ps_r : 0, s);
}

// ---------------------------------------------------------------------------------------------
// Grouped weight gradient: n layers of one geometry in ONE launch (+ one reduce launch).
// Why: a strong-scaled data-parallel shard (EDSR x4, 16 patches per GPU) gives each layer's weight gradient only
// 128 pixel tiles — one tile per block, a 73 KB partial slab written for 64 KB read, 26 us per layer for 3 us of matrix
// work.  With the layer axis in the grid every block walks ~30-40 tiles of ONE layer (double-buffered by the
// wave-specialised variant), writes one slab, and 33 layers cost two launches instead of 66.
// ---------------------------------------------------------------------------------------------
static int wb_group_G(const WbPlan& pl, int n, bool& spec) {
  const int spec_env = env_int("SRK_WGRAD_SPEC", 1);
  const int g_env = SRK_EXP_INT("SRK_WG_GROUP_G", 0);  // experiment: slabs per layer
  const int per = n * pl.gy * pl.gz;  // (layer, channel-chunk) pairs
  // one block per CU (wave-specialised, two LDS buffer sets) when every block gets >= 2 tiles; else two per CU
  int g1 = kNumCU / per;
  if (g1 < 1) g1 = 1;
  spec = spec_env && 2 * pl.lds + 8 * 1024 <= 160 * 1024 && pl.ntiles >= 2 * g1;
  int G = spec ? g1 : (2 * kNumCU) / per;
  if (G < 1) G = 1;
  if (g_env > 0) G = g_env;
  if (G > pl.ntiles) G = pl.ntiles;
  if (spec && pl.ntiles < 2 * G) spec = false;
  return G;
}

size_t conv_wgrad_bf_grouped_ws(const srk_conv_desc& d, int n) {
  WbPlan pl = wb_plan(d);
  if (!pl.ok || n < 1) return 0;
  bool spec;
  // workspace for the largest G either variant may pick (2 blocks per CU)
  int G = (2 * kNumCU) / (n * pl.gy * pl.gz);
  const int G2 = wb_group_G(pl, n, spec);
  if (G2 > G) G = G2;
  if (G < 1) G = 1;
  if (G > pl.ntiles) G = pl.ntiles;
  return (size_t)n * G * ((size_t)d.KH * d.KW * d.Cin * d.Cout + d.Cout) * sizeof(float);
}

int conv_wgrad_bf_grouped(const srk_conv_desc& d, int n, const float* const* xs, const float* const* dys,
                          const srk_bwd_mask* masks, float* const* dws, float* const* dbs, float beta, void* ws,
                          size_t ws_bytes, hipStream_t s) {
  WbPlan pl = wb_plan(d);
  if (!pl.ok || n < 1 || n > WB_MAXGROUP || d.dy_ps_r > 1) {
    set_error("conv_wgrad_bf_grouped: shape / group size not covered");
    return SRK_ERR_UNSUPPORTED;
  }
  bool spec = false;
  const int G = wb_group_G(pl, n, spec);
  const size_t elems = (size_t)d.KH * d.KW * d.Cin * d.Cout;
  const size_t slab_bytes = (size_t)n * G * elems * sizeof(float);
  const size_t need = slab_bytes + (size_t)n * G * d.Cout * sizeof(float);
  if (!ws || ws_bytes < need) {
    set_error("conv_wgrad_bf_grouped: workspace %zu < %zu", ws_bytes, need);
    return SRK_ERR_WORKSPACE;
  }
  const bool has_bias = dbs && dbs[0];
  WgGroup GR{};
  WgGroupOut GO{};
  bool vec_x = d.Cin % 4 == 0, vec_y = d.Cout % 4 == 0;
  for (int l = 0; l < n; ++l) {
    if (!xs[l] || !dys[l] || !dws[l] || (has_bias != (dbs && dbs[l] != nullptr))) {
      set_error("conv_wgrad_bf_grouped: null tensor or mixed bias / bias-free layers in one group (layer %d)", l);
      return SRK_ERR_BAD_ARG;
    }
    GR.L[l].x = xs[l];
    GR.L[l].dy = dys[l];
    GR.L[l].mask_y = masks ? masks[l].y : nullptr;
    GR.L[l].mask_slope = masks ? masks[l].slope : 0.f;
    GO.L[l].dw = dws[l];
    GO.L[l].db = has_bias ? dbs[l] : nullptr;
    vec_x = vec_x && ((uintptr_t)xs[l] % 16 == 0);
    vec_y = vec_y && ((uintptr_t)dys[l] % 16 == 0) && (!GR.L[l].mask_y || (uintptr_t)GR.L[l].mask_y % 16 == 0);
  }
  WgBfParams P{};
  P.ws = (float*)ws;
  float* bias_ws = reinterpret_cast<float*>(static_cast<char*>(ws) + slab_bytes);
  P.bias_partial = has_bias ? bias_ws : nullptr;
  P.N = d.N; P.Cin = d.Cin; P.Cout = d.Cout;
  P.XH = d.H; P.XW = d.W; P.YH = d.OH; P.YW = d.OW;
  P.KH = d.KH; P.KW = d.KW; P.pad = d.pad;
  P.TH = pl.TH; P.TW = pl.TW; P.TWo = pl.TWo; P.tiles_y = pl.tiles_y; P.tiles_x = pl.tiles_x;
  P.HH = pl.HH; P.HWp = pl.HWp; P.CS = pl.CS; P.DS = pl.DS;
  P.ntiles = pl.ntiles; P.G = G; P.nks = pl.nks;
  P.vec_x = vec_x; P.vec_y = vec_y;
  P.dy_ps_r = 0; P.dy_ps_C = d.Cout;
  P.prefetch = wb_prefetch_ok(pl, d) && P.vec_x && P.vec_y;  // 16-byte channel groups only
  {
    const int dbg = SRK_EXP_INT("SRK_DBG", 0);
    P.dbg = dbg;
#ifdef SRK_EXPERIMENTS
    P.prof = g_wb_prof;
#endif
    if (dbg & 32)
      fprintf(stderr, "[srk] k_wgrad_bf grouped cfg %d%s: %d layers x %d slabs, tile %d x %d, %d tiles per layer, grid %d x %d x %d\n",
              pl.cfg, spec ? " (wave-specialised)" : "", n, G, pl.TH, pl.TW, pl.ntiles, n * G, pl.gy, pl.gz);
  }
  dim3 grid(n * G, pl.gy, pl.gz);
  size_t lds_half = pl.lds;
  {
    int cs_ring = 0;
    const size_t ring_bytes = wb_ring_setup(pl, spec, P.prefetch, cs_ring);
    if (ring_bytes) {
      P.ring = 1;
      P.CS = cs_ring;
      lds_half = (ring_bytes + 1) / 2;
    }
  }
  size_t tr_lds = 0;
  switch (pl.cfg) {
    case 0:
#ifdef SRK_EXPERIMENTS
      if (spec && wb_k33(P) && SRK_EXP_INT("SRK_WG_W8", 0)) { wb_launch_grouped<2, 4, 1>(P, GR, grid, lds_half, spec, s); break; }
#endif
      if (wt_setup(P, d, pl, spec, tr_lds)) { wt_launch<true>(P, GR, grid, tr_lds, s); break; }
      wb_launch_grouped<2, 2, 2>(P, GR, grid, lds_half, spec, s);
      break;
    case 1: wb_launch_grouped<4, 1, 2>(P, GR, grid, lds_half, spec, s); break;
    default: wb_launch_grouped<4, 1, 1>(P, GR, grid, lds_half, spec, s); break;
  }
  int rc = check_launch("conv_wgrad_bf_grouped");
  if (rc) return rc;
  if (wgrad_reduce_deferring()) {   // queued: one job per layer, summed in this kernel's order by the merged launch
    for (int l = 0; l < n; ++l) {
      rc = wgrad_reduce_submit(false, (const float*)ws + (size_t)l * G * elems, GO.L[l].dw, G, d.Cout, d.Cin, d.KH, d.KW, 0,
                               beta, has_bias ? (const float*)bias_ws + (size_t)l * G * d.Cout : nullptr, GO.L[l].db, d.Cout,
                               0, s);
      if (rc == -100) break;        // (does not fit a job record: the launch below does the whole group)
      if (rc) return rc;
    }
    if (rc == SRK_OK) return SRK_OK;
  }
  const int nwb = cdiv(elems, 256), bias_blocks = has_bias ? cdiv(d.Cout, 64) : 0;
  hipLaunchKernelGGL(k_wgrad_reduce_grouped, dim3(nwb + bias_blocks, n), dim3(256), 0, s, (const float*)ws, GO, G, d.Cout,
                     d.Cin, d.KH, d.KW, beta, has_bias ? (const float*)bias_ws : nullptr);
  return check_launch("conv_wgrad_reduce_grouped");
}

}  // namespace srk

#ifdef SRK_EXPERIMENTS
// experiments build only: device buffer of 16 int64 per block for the role-time sums of the next k_wgrad_bf launches
extern "C" void srk_debug_wgrad_prof(void* p) { srk::g_wb_prof = static_cast<long long*>(p); }
#endif

