// Generic gather convolution: the plain, shape-agnostic implementation of Conv2d /
// ConvTranspose2d forward, data gradient and weight gradient.  It covers every geometry the
// reference can express (any kernel size, stride, padding, output_padding, channel count) and
// is the cross-check for the MFMA / direct fast paths; SRK_ALGO_AUTO only lands here for
// shapes no fast kernel covers.
//
//   out[n,oy,ox,oc] = sum_{kh,kw,ic} in[n,iy,ix,ic] * W[kh][kw][ic][oc]
//     CONV  gather: iy = oy*s - p + kh
//     TRANS gather: iy = (oy + p - kh) / s   (only when divisible)
//
// Forward of Conv2d is a CONV gather, forward of ConvTranspose2d a TRANS gather; the data
// gradient of each is the other gather with the channel roles swapped (see api.hip).
#include "srk_common.h"
#include "conv_problem.h"
#include "pack_items.h"

namespace srk {

__global__ __launch_bounds__(256) void k_gather_conv(GatherConv g, const float* __restrict__ in,
                                                     const float* __restrict__ Wp, float* __restrict__ out, Epi ep,
                                                     const float* __restrict__ mask_y, float mask_slope) {
  const size_t total = (size_t)g.N * g.OH * g.OW * g.OC;
  for (size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (size_t)gridDim.x * 256) {
    const int oc = (int)(gid % g.OC);
    size_t pix = gid / g.OC;
    const int ox = (int)(pix % g.OW);
    pix /= g.OW;
    const int oy = (int)(pix % g.OH);
    const int n = (int)(pix / g.OH);
    float acc = 0.f;
    for (int kh = 0; kh < g.KH; ++kh) {
      int iy;
      if (!g.trans) {
        iy = oy * g.stride - g.pad + kh;
      } else {
        const int t = oy + g.pad - kh;
        if (t < 0 || (t % g.stride) != 0) continue;
        iy = t / g.stride;
      }
      if (iy < 0 || iy >= g.IH) continue;
      for (int kw = 0; kw < g.KW; ++kw) {
        int ix;
        if (!g.trans) {
          ix = ox * g.stride - g.pad + kw;
        } else {
          const int t = ox + g.pad - kw;
          if (t < 0 || (t % g.stride) != 0) continue;
          ix = t / g.stride;
        }
        if (ix < 0 || ix >= g.IW) continue;
        const size_t ioff = (((size_t)n * g.IH + iy) * g.IW + ix) * g.IC;
        const float* ip = in + ioff;
        const float* wp = Wp + ((size_t)(kh * g.KW + kw) * g.IC) * g.OC + oc;
        if (mask_y) {
          const float* mp = mask_y + ioff;
          for (int ic = 0; ic < g.IC; ++ic) {
            float v = ip[ic];
            v = mp[ic] > 0.f ? v : v * mask_slope;
            acc = fmaf(v, wp[(size_t)ic * g.OC], acc);
          }
        } else {
          for (int ic = 0; ic < g.IC; ++ic) acc = fmaf(ip[ic], wp[(size_t)ic * g.OC], acc);
        }
      }
    }
    epi_store(ep, g, acc, n, oy, ox, oc, out);
  }
}

int conv_generic_gather(const GatherConv& g, const float* in, const float* wp, float* out, const Epi& ep,
                        const float* mask_y, float mask_slope, hipStream_t s) {
  const size_t total = (size_t)g.N * g.OH * g.OW * g.OC;
  size_t nb = (total + 255) / 256;
  if (nb > 256 * 32) nb = 256 * 32;
  note_kernel("k_gather_conv");
  hipLaunchKernelGGL(k_gather_conv, dim3((unsigned)nb), dim3(256), 0, s, g, in, wp, out, ep, mask_y, mask_slope);
  return check_launch("conv_generic_gather");
}

// ---------------------------------------------------------------------------------------------
// Weight gradient.  "small" tensor S at positions q, "big" tensor B at positions q*s - p + k:
//   Conv2d:          S = dy (Cout), B = x (Cin)     dw[co][ci][kh][kw]
//   ConvTranspose2d: S = x  (Cin),  B = dy (Cout)   dw[ci][co][kh][kw]
// Accumulated with fp32 atomics into a zeroed [KH][KW][Cin][Cout] workspace (split over the
// pixel axis), then written to the torch layout by k_wgrad_finalize.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_wgrad_generic(srk_conv_desc d, const float* __restrict__ x,
                                                       const float* __restrict__ dy,
                                                       const float* __restrict__ mask_y, float mask_slope,
                                                       float* __restrict__ ws, int pix_per_split) {
  const int elems = d.KH * d.KW * d.Cin * d.Cout;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= elems) return;
  const int co = e % d.Cout;
  const int ci = (e / d.Cout) % d.Cin;
  const int tap = e / (d.Cout * d.Cin);
  const int kh = tap / d.KW, kw = tap % d.KW;
  // small-tensor spatial extent
  const int SH = d.transposed ? d.H : d.OH, SW = d.transposed ? d.W : d.OW;
  const int BH = d.transposed ? d.OH : d.H, BW = d.transposed ? d.OW : d.W;
  const size_t npix = (size_t)d.N * SH * SW;
  size_t p0 = (size_t)blockIdx.y * pix_per_split;
  size_t p1 = p0 + pix_per_split;
  if (p1 > npix) p1 = npix;
  float acc = 0.f;
  for (size_t p = p0; p < p1; ++p) {
    const int sx = (int)(p % SW);
    const int sy = (int)((p / SW) % SH);
    const size_t n = p / ((size_t)SW * SH);
    const int by = sy * d.stride - d.pad + kh, bx = sx * d.stride - d.pad + kw;
    if (by < 0 || by >= BH || bx < 0 || bx >= BW) continue;
    const size_t spix = (n * SH + sy) * SW + sx;
    const size_t bpix = (n * BH + by) * BW + bx;
    float xv, gv;
    size_t goff;
    if (!d.transposed) {
      xv = x[bpix * d.Cin + ci];
      goff = spix * d.Cout + co;
    } else {
      xv = x[spix * d.Cin + ci];
      goff = bpix * d.Cout + co;
    }
    gv = dy[goff];
    if (mask_y) gv = mask_y[goff] > 0.f ? gv : gv * mask_slope;
    acc = fmaf(xv, gv, acc);
  }
  if (acc != 0.f) atomicAdd(&ws[e], acc);
}

// ws[kh][kw][ci][co] -> dw torch layout, dw = beta*dw + ws
__global__ __launch_bounds__(256) void k_wgrad_finalize(const float* __restrict__ ws, float* __restrict__ dw, int Cout,
                                                        int Cin, int KH, int KW, int transposed, float beta) {
  const int elems = KH * KW * Cin * Cout;
  const int e = blockIdx.x * 256 + threadIdx.x;  // index in torch layout
  if (e >= elems) return;
  const int kw = e % KW;
  const int kh = (e / KW) % KH;
  int ci, co;
  if (!transposed) {
    ci = (e / (KW * KH)) % Cin;
    co = e / (KW * KH * Cin);
  } else {
    co = (e / (KW * KH)) % Cout;
    ci = e / (KW * KH * Cout);
  }
  const float v = ws[((size_t)(kh * KW + kw) * Cin + ci) * Cout + co];
  dw[e] = beta != 0.f ? beta * dw[e] + v : v;
}

// db[co] = beta*db + sum over pixels of (masked) dy[pix][co].  Two deterministic stages:
// grid (channel groups of 64, pixel splits): lanes = channels (coalesced 256-byte rows), the 4
// waves stride the split's pixels; partial[split][co] then a fixed-order reduction.
__global__ __launch_bounds__(256) void k_bias_grad_partial(const float* __restrict__ dy,
                                                           const float* __restrict__ mask_y, float mask_slope,
                                                           float* __restrict__ partial, size_t npix, int Cout,
                                                           size_t pix_per_split) {
  __shared__ float sm[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int co = blockIdx.x * 64 + lane;
  size_t p0 = (size_t)blockIdx.y * pix_per_split, p1 = p0 + pix_per_split;
  if (p1 > npix) p1 = npix;
  float acc = 0.f;
  if (co < Cout) {
    for (size_t p = p0 + w; p < p1; p += 4) {
      float g = dy[p * Cout + co];
      if (mask_y) g = mask_y[p * Cout + co] > 0.f ? g : g * mask_slope;
      acc += g;
    }
  }
  sm[w][lane] = acc;
  __syncthreads();
  if (w == 0 && co < Cout)
    partial[(size_t)blockIdx.y * Cout + co] = sm[0][lane] + sm[1][lane] + sm[2][lane] + sm[3][lane];
}

// one block per 64 channels: lanes = channels, the 4 waves stride the splits with 4 independent
// accumulators each (loads stay in flight), fixed combination order => deterministic.
__global__ __launch_bounds__(256) void k_bias_grad_final(const float* __restrict__ partial, float* __restrict__ db,
                                                         int splits, int Cout, float beta) {
  __shared__ float sm[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int co = blockIdx.x * 64 + lane;
  float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
  if (co < Cout) {
    int s = w;
    for (; s + 12 < splits; s += 16) {
      t0 += partial[(size_t)s * Cout + co];
      t1 += partial[(size_t)(s + 4) * Cout + co];
      t2 += partial[(size_t)(s + 8) * Cout + co];
      t3 += partial[(size_t)(s + 12) * Cout + co];
    }
    for (; s < splits; s += 4) t0 += partial[(size_t)s * Cout + co];
  }
  sm[w][lane] = (t0 + t1) + (t2 + t3);
  __syncthreads();
  if (w == 0 && co < Cout) {
    const float t = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
    db[co] = beta != 0.f ? beta * db[co] + t : t;
  }
}

constexpr int kBiasSplits = 1024;

size_t conv_bias_grad_ws(const srk_conv_desc& d) { return (size_t)kBiasSplits * d.Cout * sizeof(float); }

size_t conv_generic_wgrad_ws(const srk_conv_desc& d) {
  return (size_t)d.KH * d.KW * d.Cin * d.Cout * sizeof(float) + conv_bias_grad_ws(d);
}

int conv_bias_grad(const srk_conv_desc& d, const float* dy, const srk_bwd_mask* mask, float* db, float beta, void* ws,
                   hipStream_t s) {
  const size_t npix = (size_t)d.N * d.OH * d.OW;
  int splits = (int)((npix + 255) / 256);
  if (splits > kBiasSplits) splits = kBiasSplits;
  if (splits < 1) splits = 1;
  const size_t pps = (npix + splits - 1) / splits;
  hipLaunchKernelGGL(k_bias_grad_partial, dim3(cdiv(d.Cout, 64), splits), dim3(256), 0, s, dy,
                     mask ? mask->y : nullptr, mask ? mask->slope : 0.f, (float*)ws, npix, d.Cout, pps);
  hipLaunchKernelGGL(k_bias_grad_final, dim3(cdiv(d.Cout, 64)), dim3(256), 0, s, (const float*)ws, db, splits,
                     d.Cout, beta);
  return check_launch("conv_bias_grad");
}

// Final stage only: partial[splits][Cout] was produced by another kernel (the MFMA weight-gradient
// kernel sums its LDS-resident dY tiles on the side).
int conv_bias_grad_finish(const float* partial, int splits, float* db, int Cout, float beta, hipStream_t s) {
  hipLaunchKernelGGL(k_bias_grad_final, dim3(cdiv(Cout, 64)), dim3(256), 0, s, partial, db, splits, Cout, beta);
  return check_launch("conv_bias_grad_finish");
}

int conv_wgrad_finalize(const srk_conv_desc& d, const float* ws, float* dw, float beta, hipStream_t s) {
  const int elems = d.KH * d.KW * d.Cin * d.Cout;
  hipLaunchKernelGGL(k_wgrad_finalize, dim3(cdiv(elems, 256)), dim3(256), 0, s, ws, dw, d.Cout, d.Cin, d.KH, d.KW,
                     d.transposed, beta);
  return check_launch("conv_wgrad_finalize");
}

int conv_generic_wgrad(const srk_conv_desc& d, const float* x, const float* dy, const srk_bwd_mask* mask, float* dw,
                       float* db, float beta, void* ws, size_t ws_bytes, hipStream_t s) {
  const size_t need = conv_generic_wgrad_ws(d);
  if (ws_bytes < need || !ws) {
    set_error("conv_generic_wgrad: workspace %zu < %zu", ws_bytes, need);
    return SRK_ERR_WORKSPACE;
  }
  const size_t wbytes = (size_t)d.KH * d.KW * d.Cin * d.Cout * sizeof(float);
  hipError_t me = hipMemsetAsync(ws, 0, wbytes, s);
  if (me != hipSuccess) {
    set_error("conv_generic_wgrad: memset failed: %s", hipGetErrorString(me));
    return SRK_ERR_LAUNCH;
  }
  const int elems = d.KH * d.KW * d.Cin * d.Cout;
  const int SH = d.transposed ? d.H : d.OH, SW = d.transposed ? d.W : d.OW;
  const size_t npix = (size_t)d.N * SH * SW;
  int splits = (int)((npix + 511) / 512);
  if (splits > 256) splits = 256;
  if (splits < 1) splits = 1;
  const int pps = (int)((npix + splits - 1) / splits);
  const float* my = mask ? mask->y : nullptr;
  const float ms = mask ? mask->slope : 0.f;
  hipLaunchKernelGGL(k_wgrad_generic, dim3(cdiv(elems, 256), splits), dim3(256), 0, s, d, x, dy, my, ms, (float*)ws,
                     pps);
  int rc = check_launch("conv_generic_wgrad");
  if (rc) return rc;
  rc = conv_wgrad_finalize(d, (const float*)ws, dw, beta, s);
  if (rc) return rc;
  if (db) rc = conv_bias_grad(d, dy, mask, db, beta, (char*)ws + wbytes, s);
  return rc;
}

// ---------------------------------------------------------------------------------------------
// Weight / bias packing (state_dict layout -> kernel layout)
// ---------------------------------------------------------------------------------------------
// fwd:  wp[kh][kw][ci][co'] ; co' = ps-permuted output channel
// bwd:  wp[kh][kw][co][ci]
__global__ __launch_bounds__(256) void k_pack_weight(const float* __restrict__ w, float* __restrict__ wp, int Cout,
                                                     int Cin, int KH, int KW, int transposed, int ps_r, int bwd) {
  const int elems = KH * KW * Cin * Cout;
  const int e = blockIdx.x * 256 + threadIdx.x;  // index in the PACKED layout (coalesced stores)
  if (e >= elems) return;
  pack_f32_item(e, w, wp, Cout, Cin, KH, KW, transposed, ps_r, bwd);
}

__global__ void k_pack_bias_ps(const float* __restrict__ b, float* __restrict__ bp, int Cout, int ps_r) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= Cout) return;
  const int C = Cout / (ps_r * ps_r);
  const int q = e / C, c = e % C;
  bp[e] = b[c * ps_r * ps_r + q];
}

}  // namespace srk

using namespace srk;

extern "C" int srk_pack_weight_fwd(const float* w, float* wp, int Cout, int Cin, int KH, int KW, int transposed,
                                   int ps_r, void* stream) {
  SRK_REQUIRE(w && wp, "pack_weight_fwd: null pointer");
  SRK_REQUIRE(Cout > 0 && Cin > 0 && KH > 0 && KW > 0, "pack_weight_fwd: bad dims");
  SRK_REQUIRE(ps_r <= 1 || Cout % (ps_r * ps_r) == 0, "pack_weight_fwd: Cout %d not divisible by r^2", Cout);
  const int elems = KH * KW * Cin * Cout;
  hipLaunchKernelGGL(k_pack_weight, dim3(cdiv(elems, 256)), dim3(256), 0, (hipStream_t)stream, w, wp, Cout, Cin, KH,
                     KW, transposed, ps_r, 0);
  int rc = check_launch("pack_weight_fwd");
  if (rc) return rc;
  return bf3_pack_prepared(w, wp, Cout, Cin, KH, KW, transposed, ps_r, 0, (hipStream_t)stream);
}
extern "C" size_t srk_packed_weight_bytes(int Cout, int Cin, int KH, int KW, int bwd) {
  if (Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return 0;
  const size_t elems = (size_t)KH * KW * Cin * Cout;
  if (bwd) return bf3_prepared_offset(elems) + bf3_prepared_bytes(Cout, Cin, KH * KW);
  // forward buffers: + the fp16 planes and trailer of SRK_ALGO_MFMA_F16X3
  return bf3_prepared_offset(elems) + f16_section_offset(Cin, Cout, KH * KW) + f16_section_bytes(Cin, Cout, KH * KW);
}
extern "C" int srk_pack_weight_bwd(const float* w, float* wp, int Cout, int Cin, int KH, int KW, int transposed,
                                   int ps_r, void* stream) {
  SRK_REQUIRE(w && wp, "pack_weight_bwd: null pointer");
  SRK_REQUIRE(Cout > 0 && Cin > 0 && KH > 0 && KW > 0, "pack_weight_bwd: bad dims");
  SRK_REQUIRE(ps_r <= 1 || Cout % (ps_r * ps_r) == 0, "pack_weight_bwd: Cout %d not divisible by r^2", Cout);
  const int elems = KH * KW * Cin * Cout;
  hipLaunchKernelGGL(k_pack_weight, dim3(cdiv(elems, 256)), dim3(256), 0, (hipStream_t)stream, w, wp, Cout, Cin, KH,
                     KW, transposed, ps_r, 1);
  int rc = check_launch("pack_weight_bwd");
  if (rc) return rc;
  return bf3_pack_prepared(w, wp, Cout, Cin, KH, KW, transposed, ps_r, 1, (hipStream_t)stream);
}
extern "C" int srk_pack_bias_ps(const float* b, float* bp, int Cout, int ps_r, void* stream) {
  SRK_REQUIRE(b && bp && Cout > 0 && ps_r >= 1, "pack_bias_ps: bad args");
  SRK_REQUIRE(Cout % (ps_r * ps_r) == 0, "pack_bias_ps: Cout %d not divisible by r^2", Cout);
  hipLaunchKernelGGL(k_pack_bias_ps, dim3(cdiv(Cout, 256)), dim3(256), 0, (hipStream_t)stream, b, bp, Cout, ps_r);
  return check_launch("pack_bias_ps");
}
