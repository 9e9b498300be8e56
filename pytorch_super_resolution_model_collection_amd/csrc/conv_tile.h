// Shared host/device definitions of the halo-tiled convolution kernels (fp32 MFMA, bf16x3 MFMA,
// direct): launch parameters, the tile picker and the phase decomposition of strided gathers.
#pragma once
#include "srk_common.h"
#include "conv_problem.h"

namespace srk {

struct MfmaConvParams {
  const float* in;
  const float* wp;
  float* out;
  const float* mask_y;
  float mask_slope;
  Epi ep;
  int N, IH, IW, IC;
  int OH, OW, OC;
  // phase space: this launch produces outputs (oy0 + r*os, ox0 + c*os), r < PH, c < PW;
  // virtual tap (u,v) reads input (r*is + iy0 + u, c*is + ix0 + v) and weight tap
  // (wh0 + wdh*u, ww0 + wdw*v).
  int PH, PW, oy0, ox0, os;
  int iy0, ix0, is;
  int KHv, KWv, wh0, wdh, ww0, wdw, KW_full;
  int TH, TW, tiles_y, tiles_x, HH, HW;
  int CK;   // input-channel chunk staged per pass (multiple of 16, <= 64)
  int PSA;  // halo pixel stride in floats (CK + 4, or 4 for the tap-group variant)
  int BNp;  // LDS filter row stride in floats (NT*16 + 4)
  int halo_floats;
  int vec_in, vec_w;
  int in_nchw;  // input tensor is NCHW (row-packed Cin <= 4 kernel only)
  int in_ps_r;  // input tensor is pixel-shuffled: packed channel (i, j, c) of pixel (y, x) lives at (y*r+i, x*r+j, c)
  int in_ps_C;  // IC / r^2
};

// Input addressing of the halo staging loops as an affine map with loop-invariant terms:
//   offset(n, iy, ix, ch) = (n*IH + iy) * sA + ix * sB + K(ch)
// plain NHWC: sA = IW*IC, sB = IC, K = ch;  pixel-shuffled [N, IH*r, IW*r, C] (packed channel ch = (i*r + j)*C + c
// lives at (iy*r + i, ix*r + j, c)): sA = r*IW*r*C, sB = r*C, K = i*IW*r*C + j*C + c.  sA / sB are wave-uniform,
// K is one per-thread value, so the pixel-shuffled case costs no registers in the loops.
struct InAddr {
  unsigned sA, sB, K;  // element units; one image holds < 2^30 elements (checked by the *_supported functions)
};
__device__ __forceinline__ InAddr conv_in_addr(const MfmaConvParams& P, int ch) {
  InAddr a;
  if (P.in_ps_r <= 1) {
    a.sA = (unsigned)(P.IW * P.IC);
    a.sB = (unsigned)P.IC;
    a.K = (unsigned)ch;
  } else {
    const int r = P.in_ps_r, C = P.in_ps_C;
    const int q = ch / C, c = ch - q * C;
    const int i = q / r, j = q - i * r;
    a.sB = (unsigned)(r * C);
    a.sA = (unsigned)(r * P.IW) * a.sB;
    a.K = (unsigned)(i * P.IW) * a.sB + (unsigned)(j * C + c);
  }
  return a;
}

static constexpr int kLdsBudgetBytes = 78 * 1024;  // 2 blocks per CU out of 160 KiB

struct TilePick {
  int TH, TW, tiles_y, tiles_x, HH, HW;
  double eff;
};

// Choose the <=128-pixel tile (any aspect, any width) that covers PH x PW with the fewest tiles
// under the LDS budget; ties -> smaller halo.
static inline bool pick_tile(int maxpix, int PH, int PW, int is, int KHv, int KWv, int psa, int budget_floats,
                      TilePick& best) {
  bool found = false;
  long best_tiles = 0, best_halo = 0;
  const int maxTW = PW < maxpix ? PW : maxpix;
  for (int TW = 1; TW <= maxTW; ++TW) {
    int TH = maxpix / TW;
    if (TH > PH) TH = PH;
    for (; TH >= 1; --TH) {
      const int HH = (TH - 1) * is + KHv, HWd = (TW - 1) * is + KWv;
      if ((long)HH * HWd * psa <= budget_floats) {
        const long tiles = (long)cdiv(PH, TH) * cdiv(PW, TW);
        const long halo = (long)HH * HWd * tiles;
        // fewest tiles; then (within 6 %) the WIDEST tile: wave lanes map to consecutive pixels of a
        // row, so wide rows keep the 16-lane fragment reads contiguous in LDS (conflict-free) and the
        // global halo loads / output stores long and coalesced; then the smallest halo.
        const bool fewer = tiles < best_tiles;
        const bool same = tiles == best_tiles;
        const bool wider = same && halo * 100 <= best_halo * 106 && TW > best.TW;
        const bool smaller = same && halo < best_halo && TW >= best.TW;
        if (!found || fewer || wider || smaller) {
          found = true;
          best_tiles = tiles;
          best_halo = halo;
          best = TilePick{TH, TW, (int)cdiv(PH, TH), (int)cdiv(PW, TW), HH, HWd,
                          (double)PH * PW / ((double)tiles * (double)maxpix)};
        }
        break;  // smaller TH only gets worse for this TW
      }
    }
  }
  return found;
}


// Fills the geometry of `P` from the gather problem and calls launch(P) once (CONV gather, TRANS
// gather with stride 1) or once per output phase (TRANS gather with stride s: s*s launches).
template <typename Launch>
static inline int for_each_phase(const GatherConv& g, const float* in, const float* wp, float* out, const Epi& ep,
                                 const float* mask_y, float mask_slope, Launch launch) {
  MfmaConvParams P{};
  P.in = in; P.wp = wp; P.out = out; P.mask_y = mask_y; P.mask_slope = mask_slope; P.ep = ep;
  P.N = g.N; P.IH = g.IH; P.IW = g.IW; P.IC = g.IC; P.OH = g.OH; P.OW = g.OW; P.OC = g.OC;
  P.KW_full = g.KW;
  P.in_nchw = g.in_nchw;
  P.in_ps_r = g.in_ps_r;
  P.in_ps_C = g.in_ps_r > 1 ? g.IC / (g.in_ps_r * g.in_ps_r) : g.IC;
  P.vec_in = (g.IC % 4 == 0) && ((uintptr_t)in % 16 == 0) && (!mask_y || (uintptr_t)mask_y % 16 == 0);
  P.vec_w = (g.OC % 4 == 0) && ((uintptr_t)wp % 16 == 0);
  if (!g.trans) {
    P.PH = g.OH; P.PW = g.OW; P.oy0 = 0; P.ox0 = 0; P.os = 1;
    P.iy0 = -g.pad; P.ix0 = -g.pad; P.is = g.stride;
    P.KHv = g.KH; P.KWv = g.KW; P.wh0 = 0; P.wdh = 1; P.ww0 = 0; P.wdw = 1;
    return launch(P);
  }
  // TRANS gather: iy = (oy + p - kh)/s.  One launch per output phase (py,px) = ((oy+p)%s, (ox+p)%s).
  const int st = g.stride;
  for (int py = 0; py < st; ++py) {
    const int oy0 = (((py - g.pad) % st) + st) % st;
    if (oy0 >= g.OH) continue;
    const int KHv = py < g.KH ? (g.KH - py + st - 1) / st : 0;
    const int by = (oy0 + g.pad - py) / st;
    for (int px = 0; px < st; ++px) {
      const int ox0 = (((px - g.pad) % st) + st) % st;
      if (ox0 >= g.OW) continue;
      const int KWv = px < g.KW ? (g.KW - px + st - 1) / st : 0;
      const int bx = (ox0 + g.pad - px) / st;
      MfmaConvParams Q = P;
      Q.PH = (g.OH - oy0 + st - 1) / st;
      Q.PW = (g.OW - ox0 + st - 1) / st;
      Q.oy0 = oy0; Q.ox0 = ox0; Q.os = st; Q.is = 1;
      if (KHv == 0 || KWv == 0) {
        Q.KHv = 0; Q.KWv = 0; Q.iy0 = 0; Q.ix0 = 0; Q.wh0 = 0; Q.wdh = 0; Q.ww0 = 0; Q.wdw = 0;
      } else {
        Q.KHv = KHv; Q.KWv = KWv;
        Q.iy0 = by - (KHv - 1); Q.ix0 = bx - (KWv - 1);
        Q.wh0 = py + st * (KHv - 1); Q.wdh = -st;
        Q.ww0 = px + st * (KWv - 1); Q.wdw = -st;
      }
      const int rc = launch(Q);
      if (rc) return rc;
    }
  }
  return SRK_OK;
}


// ---------------------------------------------------------------------------------------------
// Vector epilogue: 4 consecutive packed output channels [oc, oc+4) of output pixel (n, oy, ox).
// Used by the LDS-staged epilogues (each lane owns 16 bytes of one pixel -> coalesced stores).
//   out = PS_r(act(v + bias)) + residual, see struct Epi.
// ---------------------------------------------------------------------------------------------
typedef float epi_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void epi_store4(const Epi& ep, int OH, int OW, int OC, int n, int oy, int ox, int oc,
                                           epi_f4 v, float* __restrict__ out) {
  size_t off;
  int c_first;
  bool run_ok;
  if (ep.ps_r > 1) {
    const int r = ep.ps_r;
    const int C = OC / (r * r);
    const int RL = r * C;  // (j, c) run: contiguous floats of one output row segment
    const int i = oc / RL, rem = oc - i * RL;
    off = (((size_t)n * OH * r + (size_t)oy * r + i) * ((size_t)OW * r) + (size_t)ox * r) * C + rem;
    c_first = rem % C;
    run_ok = rem + 3 < RL;
  } else {
    off = (((size_t)n * OH + oy) * OW + ox) * OC + oc;
    c_first = oc;
    run_ok = true;
  }
  const bool full = oc + 3 < OC && run_ok;
  const bool simple_act = !(ep.act == SRK_ACT_PRELU && ep.prelu_n > 1);
  if (full && simple_act && (off & 3) == 0 && (!ep.bias || (oc & 3) == 0)) {
    if (ep.bias) {
      const epi_f4 b = *reinterpret_cast<const epi_f4*>(ep.bias + oc);
      v += b;
    }
    if (ep.act != SRK_ACT_NONE) {
      const float a = ep.act == SRK_ACT_PRELU ? ep.prelu_w[0] : ep.slope;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], ep.act, a);
    }
    if (ep.residual) v += *reinterpret_cast<const epi_f4*>(ep.residual + off);
    *reinterpret_cast<epi_f4*>(out + off) = v;
    return;
  }
  // ragged / unaligned / per-channel-PReLU tail: element-wise through the scalar path
  GatherConv g{};
  g.OH = OH; g.OW = OW; g.OC = OC;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (oc + e < OC) epi_store(ep, g, v[e], n, oy, ox, oc + e, out);
  (void)c_first;
}

// ---------------------------------------------------------------------------------------------
// Column-hoisted form of epi_store4 for the LDS-staged epilogues: everything that depends only on
// the lane's 4 output channels is computed once (EpiCol), the per-pixel part is a handful of
// multiplies (no divisions in the store loop).
// ---------------------------------------------------------------------------------------------
struct EpiCol {
  size_t off_oc;   // offset of channel group inside the output pixel block
  int oc;
  int vec;         // 16-byte path legal for this column group
  int i_row;       // pixel-shuffle sub-row (0 without PS)
  epi_f4 bias;
  float slope;
};

__device__ __forceinline__ EpiCol epi_col_setup(const Epi& ep, int OW, int OC, int oc) {
  EpiCol c;
  c.oc = oc;
  c.bias = (epi_f4){0.f, 0.f, 0.f, 0.f};
  c.slope = ep.act == SRK_ACT_PRELU ? ep.prelu_w[0] : ep.slope;
  const bool simple_act = !(ep.act == SRK_ACT_PRELU && ep.prelu_n > 1);
  bool ok = oc + 3 < OC && (oc & 3) == 0 && simple_act;
  if (ep.ps_r > 1) {
    const int r = ep.ps_r;
    const int C = OC / (r * r);
    const int RL = r * C;
    const int i = oc / RL, rem = oc - i * RL;
    c.i_row = i;
    c.off_oc = (size_t)i * ((size_t)OW * r) * C + rem;
    ok = ok && rem + 3 < RL && (RL & 3) == 0 && (rem & 3) == 0;
  } else {
    c.i_row = 0;
    c.off_oc = oc;
    ok = ok && (OC & 3) == 0;
  }
  c.vec = ok;
  if (ok && ep.bias) c.bias = *reinterpret_cast<const epi_f4*>(ep.bias + oc);
  return c;
}

// (returns what the vector path stored; the scalar fallback returns its input)
__device__ __forceinline__ epi_f4 epi_store4_col(const Epi& ep, const EpiCol& col, int OH, int OW, int OC, int n, int oy,
                                                 int ox, epi_f4 v, float* __restrict__ out) {
  if (col.vec) {
    size_t base;
    if (ep.ps_r > 1) {
      const int r = ep.ps_r;
      const int C = OC / (r * r);
      base = (((size_t)n * OH * r + (size_t)oy * r) * ((size_t)OW * r) + (size_t)ox * r) * C;
    } else {
      base = (((size_t)n * OH + oy) * OW + ox) * OC;
    }
    const size_t off = base + col.off_oc;
    v += col.bias;
    if (ep.act != SRK_ACT_NONE) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], ep.act, col.slope);
    }
    if (ep.residual) v += *reinterpret_cast<const epi_f4*>(ep.residual + off);
    *reinterpret_cast<epi_f4*>(out + off) = v;
  } else {
    epi_store4(ep, OH, OW, OC, n, oy, ox, col.oc, v, out);
  }
  return v;
}

// Host-side mirror of epi_col_setup's `vec` test for EVERY 4-channel group of the layer: true when no lane can need
// the scalar fallback (then kernels compiled without it may be used).
static inline bool epi_all_vector(const MfmaConvParams& P) { return conv_epi_all_vector(P.OC, P.ep, P.out); }

// Tile-affine output addressing for the LDS-staged epilogues: the element offset of output pixel (r, c) of the
// block's tile is off0 + r*RS + c*CS (+ the lane's channel offset), with off0 computed once per tile from
// wave-uniform values and RS / CS 32-bit — no 64-bit multiplies per store.
struct EpiTile {
  size_t off0;
  int RS, CS;
};

__device__ __forceinline__ EpiTile epi_tile_setup(const MfmaConvParams& P, int n, int r0, int c0) {
  EpiTile t;
  const int oy = P.oy0 + r0 * P.os, ox = P.ox0 + c0 * P.os;
  if (P.ep.ps_r > 1) {
    const int r = P.ep.ps_r;
    const int C = P.OC / (r * r);
    t.off0 = ((((size_t)n * P.OH + oy) * r) * ((size_t)P.OW * r) + (size_t)ox * r) * C;
    t.CS = P.os * r * C;
    t.RS = P.os * r * P.OW * r * C;
  } else {
    t.off0 = (((size_t)n * P.OH + oy) * P.OW + ox) * P.OC;
    t.CS = P.os * P.OC;
    t.RS = P.os * P.OW * P.OC;
  }
  return t;
}

// vector path of epi_store4_col (col.vec must hold) for tile pixel (r, c); returns the stored values
__device__ __forceinline__ epi_f4 epi_store4_tile(const Epi& ep, const EpiCol& col, const EpiTile& t, int r, int c,
                                                  epi_f4 v, float* __restrict__ out) {
  const size_t off = t.off0 + col.off_oc + (size_t)(unsigned)(r * t.RS + c * t.CS);
  v += col.bias;
  // one (wave-uniform) dispatch per 16-byte store instead of one switch per element
  if (ep.act == SRK_ACT_RELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
  } else if (ep.act == SRK_ACT_PRELU || ep.act == SRK_ACT_LRELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : col.slope * v[e];
  } else if (ep.act != SRK_ACT_NONE) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], ep.act, col.slope);
  }
  if (ep.residual) v += *reinterpret_cast<const epi_f4*>(ep.residual + off);
  *reinterpret_cast<epi_f4*>(out + off) = v;
  return v;
}

// Exact floor(m / d) for 0 <= m < 256, 1 <= d <= 256 with one multiply (magic = ceil(65536 / d)).
__host__ __device__ __forceinline__ int div_small_magic(int d) { return (65536 + d - 1) / d; }
__device__ __forceinline__ int div_small(int m, int magic) { return (m * magic) >> 16; }

}  // namespace srk
