// Pointwise / permutation kernels: layout copies, pixel shuffle, activations, axpby.
// All HBM-bound: 16-byte accesses where the layout allows, grid capped at ~2048 blocks with a
// grid-stride loop (cdna_hip_programming.md Guideline 11/13).
#include "srk_common.h"

namespace srk {

static inline unsigned ew_grid(size_t work_items, int per_block) {
  size_t b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > 256 * 16) b = 256 * 16;
  return (unsigned)b;
}

// ---------------------------------------------------------------------------------------------
// NCHW <-> NHWC: per image a [C][HW] <-> [HW][C] transpose through a padded LDS tile.
// ---------------------------------------------------------------------------------------------
template <bool TO_NHWC>
__global__ __launch_bounds__(256) void k_layout(const float* __restrict__ x, float* __restrict__ y, int C, int HW) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* xi = x + (size_t)n * C * HW;
  float* yo = y + (size_t)n * C * HW;
  if (TO_NHWC) {
    // read x[c][p] with p fastest; write y[p][c] with c fastest
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
      int c = c0 + ty + k, p = p0 + tx;
      if (c < C && p < HW) tile[ty + k][tx] = xi[(size_t)c * HW + p];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
      int p = p0 + ty + k, c = c0 + tx;
      if (c < C && p < HW) yo[(size_t)p * C + c] = tile[tx][ty + k];
    }
  } else {
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
      int p = p0 + ty + k, c = c0 + tx;
      if (c < C && p < HW) tile[ty + k][tx] = xi[(size_t)p * C + c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
      int c = c0 + ty + k, p = p0 + tx;
      if (c < C && p < HW) yo[(size_t)c * HW + p] = tile[tx][ty + k];
    }
  }
}

static int layout_launch(bool to_nhwc, const float* x, float* y, int N, int C, int H, int W, hipStream_t s) {
  SRK_REQUIRE(x && y, "layout: null pointer");
  SRK_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, "layout: bad dims %d %d %d %d", N, C, H, W);
  SRK_REQUIRE(N <= 65535 && cdiv(C, 32) <= 65535, "layout: N or C too large for the grid");
  const int HW = H * W;
  dim3 grid(cdiv(HW, 32), cdiv(C, 32), N);
  if (to_nhwc)
    hipLaunchKernelGGL(k_layout<true>, grid, dim3(256), 0, s, x, y, C, HW);
  else
    hipLaunchKernelGGL(k_layout<false>, grid, dim3(256), 0, s, x, y, C, HW);
  return check_launch("layout");
}

// ---------------------------------------------------------------------------------------------
// Pixel shuffle (NHWC).  x[n,h,w, c*r*r + i*r + j]  <->  y[n, h*r+i, w*r+j, c]
// One thread per OUTPUT element of the direction being computed so that stores are coalesced;
// the gather side stays inside one 4*C*r*r-byte input pixel (L1/L2-resident).
// ---------------------------------------------------------------------------------------------
template <bool FWD>
__global__ __launch_bounds__(256) void k_pixel_shuffle(const float* __restrict__ src, float* __restrict__ dst, int H,
                                                       int W, int C, int r, size_t total) {
  const int rr = r * r;
  const int Cin = C * rr;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    if (FWD) {
      // e indexes y[n][oy][ox][c]
      const int c = (int)(e % C);
      size_t t = e / C;
      const int ox = (int)(t % ((size_t)W * r));
      t /= (size_t)W * r;
      const int oy = (int)(t % ((size_t)H * r));
      const size_t n = t / ((size_t)H * r);
      const int h = oy / r, i = oy - h * r, w = ox / r, j = ox - w * r;
      dst[e] = src[((n * H + h) * W + w) * Cin + c * rr + i * r + j];
    } else {
      // e indexes dx[n][h][w][c*rr + i*r + j]
      const int ch = (int)(e % Cin);
      size_t t = e / Cin;
      const int w = (int)(t % W);
      t /= W;
      const int h = (int)(t % H);
      const size_t n = t / H;
      const int c = ch / rr, q = ch - c * rr, i = q / r, j = q - i * r;
      dst[e] = src[((n * (size_t)H * r + (size_t)h * r + i) * ((size_t)W * r) + (size_t)w * r + j) * C + c];
    }
  }
}

// Tiled form for the (r, C) pairs the reference's nets use (espcn.py:26 / base_networks.py:157,181: r = scale factor with
// C = 3 or 1 image channels, r = 2 with C = 64 feature channels).  One block moves a run of WT input pixels of one image
// row: WT * C * r * r consecutive floats on the x side, r runs of WT * r * C consecutive floats (output rows h r .. h r + r - 1)
// on the y side -- both sides are read / written as whole 16-byte vectors of consecutive addresses, the permutation
// happens in LDS (x-side layout, word = w * Cin + c r^2 + i r + j) with every division by a compile-time constant.
// HBM traffic = the 8 bytes per element the permutation needs; the round-1 kernel above (one element per thread behind
// 64-bit divisions, 4-byte gathers) stays as the fallback for other shapes and unaligned tensors.
template <int R, int C, bool FWD>
__global__ __launch_bounds__(256) void k_pixel_shuffle_tile(const float* __restrict__ src, float* __restrict__ dst, int H,
                                                            int W, int WT, int nwt) {
  constexpr int RR = R * R, CIN = C * RR, RC = R * C;
  constexpr bool VX = CIN % 4 == 0;   // x-side runs start on 16-byte boundaries
  constexpr bool VY = RC % 4 == 0;    // y-side runs do
  typedef float ps_f4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float ps_sm[];
  int b = blockIdx.x;
  const int wt = b % nwt;
  b /= nwt;                            // b = n * H + h
  const int w0 = wt * WT;
  const int cnt = min(WT, W - w0);
  const int tid = threadIdx.x;
  const float* xs = FWD ? src : dst;   // (only for the address arithmetic below)
  (void)xs;
  const size_t x_off = ((size_t)b * W + w0) * CIN;             // x-side run of this block
  const size_t y_row0 = ((size_t)b * R * W * R + (size_t)w0 * R) * C;   // y-side run of output row i = 0; + i * W * RC per row
  const int nx = cnt * CIN, ny = cnt * RC;
  auto word = [&](int i, int o) {      // LDS word of y-side element o of output row i
    const int pix = o / C, c = o - pix * C;
    const int w = pix / R, j = pix - w * R;
    return w * CIN + c * RR + i * R + j;
  };
  if (FWD) {
    if (VX) {
      for (int q = tid; q < nx / 4; q += 256)
        reinterpret_cast<ps_f4*>(ps_sm)[q] = reinterpret_cast<const ps_f4*>(src + x_off)[q];
    } else {
      for (int q = tid; q < nx; q += 256) ps_sm[q] = src[x_off + q];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < R; ++i) {
      float* row = dst + y_row0 + (size_t)i * W * RC;
      if (VY) {
        for (int q = tid; q < ny / 4; q += 256) {
          ps_f4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ps_sm[word(i, 4 * q + e)];
          reinterpret_cast<ps_f4*>(row)[q] = v;
        }
      } else {
        for (int q = tid; q < ny; q += 256) row[q] = ps_sm[word(i, q)];
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const float* row = src + y_row0 + (size_t)i * W * RC;
      if (VY) {
        for (int q = tid; q < ny / 4; q += 256) {
          const ps_f4 v = reinterpret_cast<const ps_f4*>(row)[q];
#pragma unroll
          for (int e = 0; e < 4; ++e) ps_sm[word(i, 4 * q + e)] = v[e];
        }
      } else {
        for (int q = tid; q < ny; q += 256) ps_sm[word(i, q)] = row[q];
      }
    }
    __syncthreads();
    if (VX) {
      for (int q = tid; q < nx / 4; q += 256)
        reinterpret_cast<ps_f4*>(dst + x_off)[q] = reinterpret_cast<const ps_f4*>(ps_sm)[q];
    } else {
      for (int q = tid; q < nx; q += 256) dst[x_off + q] = ps_sm[q];
    }
  }
}

template <int R, int C>
static bool ps_tile_launch(const float* src, float* dst, int N, int H, int W, bool fwd, hipStream_t s) {
  constexpr int CIN = C * R * R;
  // ~3 K floats per block (12 KB of LDS: a dozen resident blocks per CU cover each other's load latency), whole pixels,
  // a multiple of 4 of them so that every run keeps its 16-byte alignment
  int WT = (3072 / CIN) & ~3;
  if (WT < 4) WT = 4;
  if (WT > W) WT = W;
  const int nwt = (W + WT - 1) / WT;
  const size_t blocks = (size_t)N * H * nwt;
  if (blocks > 0x7fffffffu) return false;
  const size_t lds = (size_t)WT * CIN * sizeof(float);
  if (fwd)
    hipLaunchKernelGGL((k_pixel_shuffle_tile<R, C, true>), dim3((unsigned)blocks), dim3(256), lds, s, src, dst, H, W, WT, nwt);
  else
    hipLaunchKernelGGL((k_pixel_shuffle_tile<R, C, false>), dim3((unsigned)blocks), dim3(256), lds, s, src, dst, H, W, WT, nwt);
  return true;
}

// true: the tiled kernel took the call
static bool ps_tile(const float* src, float* dst, int N, int H, int W, int C, int r, bool fwd, hipStream_t s) {
  if ((((uintptr_t)src | (uintptr_t)dst) & 15) != 0 || env_int("SRK_PS_TILE", 1) == 0) return false;
  // (a side is vectorised only when its pixel size is a multiple of 16 bytes -- then every run start is aligned)
  if ((long)W * C * r * r >= (1L << 30)) return false;
#define SRK_PS_CASE(RV, CV) if (r == RV && C == CV) return ps_tile_launch<RV, CV>(src, dst, N, H, W, fwd, s);
  SRK_PS_CASE(2, 1) SRK_PS_CASE(3, 1) SRK_PS_CASE(4, 1) SRK_PS_CASE(8, 1)
  SRK_PS_CASE(2, 3) SRK_PS_CASE(3, 3) SRK_PS_CASE(4, 3) SRK_PS_CASE(8, 3)
  SRK_PS_CASE(2, 64) SRK_PS_CASE(2, 32) SRK_PS_CASE(2, 16) SRK_PS_CASE(4, 16) SRK_PS_CASE(3, 64)
#undef SRK_PS_CASE
  return false;
}

// ---------------------------------------------------------------------------------------------
// Activations
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_act_fwd(const float* __restrict__ x, float* __restrict__ y, size_t n,
                                                 int channels, int act, float slope, const float* __restrict__ pw,
                                                 int pn) {
  const bool per_ch = (act == SRK_ACT_PRELU && pn > 1);
  float a = slope;
  if (act == SRK_ACT_PRELU && !per_ch) a = pw[0];
  const size_t n4 = n / 4;
  const bool vec_ok = (channels % 4 == 0) || !per_ch;
  if (vec_ok) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
      float4 v = reinterpret_cast<const float4*>(x)[i];
      float a0 = a, a1 = a, a2 = a, a3 = a;
      if (per_ch) {
        const int c = (int)((i * 4) % channels);
        a0 = pw[c]; a1 = pw[c + 1]; a2 = pw[c + 2]; a3 = pw[c + 3];
      }
      v.x = act_apply(v.x, act, a0);
      v.y = act_apply(v.y, act, a1);
      v.z = act_apply(v.z, act, a2);
      v.w = act_apply(v.w, act, a3);
      reinterpret_cast<float4*>(y)[i] = v;
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
      y[i] = act_apply(x[i], act, per_ch ? pw[i % channels] : a);
  } else {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
      y[i] = act_apply(x[i], act, pw[i % channels]);
  }
}

// dx = dy * act'(.) ; PReLU slope gradient reduced per block then atomically added.
__global__ __launch_bounds__(256) void k_act_bwd(const float* __restrict__ dy, const float* __restrict__ saved,
                                                 float* __restrict__ dx, size_t n, int channels, int act, float slope,
                                                 const float* __restrict__ pw, int pn, float* __restrict__ dpw) {
  __shared__ float sm[4];
  const bool per_ch = (act == SRK_ACT_PRELU && pn > 1);
  float a = slope;
  if (act == SRK_ACT_PRELU && !per_ch) a = pw[0];
  float dslope = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float g = dy[i], s = saved[i];
    float d;
    switch (act) {
      case SRK_ACT_RELU: d = s > 0.f ? g : 0.f; break;
      case SRK_ACT_LRELU: d = s > 0.f ? g : g * slope; break;
      case SRK_ACT_PRELU: {
        const float ai = per_ch ? pw[i % channels] : a;
        d = s > 0.f ? g : g * ai;
        const float ds = s > 0.f ? 0.f : g * s;
        if (per_ch) {
          if (ds != 0.f) atomicAdd(&dpw[i % channels], ds);
        } else {
          dslope += ds;
        }
        break;
      }
      case SRK_ACT_TANH: d = g * (1.f - s * s); break;
      case SRK_ACT_SIGMOID: d = g * s * (1.f - s); break;
      default: d = g;
    }
    dx[i] = d;
  }
  if (act == SRK_ACT_PRELU && !per_ch && dpw) {
    const float tot = block_sum_256(dslope, sm);
    if (threadIdx.x == 0 && tot != 0.f) atomicAdd(dpw, tot);
  }
}

// 16-byte version of k_act_bwd for the activations the nets use in training (ReLU / LeakyReLU / single-slope PReLU):
// one float4 per thread and pass.  The scalar kernel walks 8 elements per thread one dependent 4-byte load pair at a
// time (13 us for the 1 M-element tensors of the SRGAN generator; 65 calls per adversarial step).
__global__ __launch_bounds__(256) void k_act_bwd4(const float* __restrict__ dy, const float* __restrict__ saved,
                                                  float* __restrict__ dx, size_t n4, int act, float slope,
                                                  const float* __restrict__ pw, float* __restrict__ dpw) {
  typedef float af4 __attribute__((ext_vector_type(4)));
  __shared__ float sm[4];
  float a = act == SRK_ACT_RELU ? 0.f : slope;
  if (act == SRK_ACT_PRELU) a = pw[0];
  float dslope = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const af4 g = reinterpret_cast<const af4*>(dy)[i], s = reinterpret_cast<const af4*>(saved)[i];
    af4 d;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      d[e] = s[e] > 0.f ? g[e] : g[e] * a;
      if (act == SRK_ACT_PRELU) dslope += s[e] > 0.f ? 0.f : g[e] * s[e];
    }
    reinterpret_cast<af4*>(dx)[i] = d;
  }
  if (act == SRK_ACT_PRELU && dpw) {
    const float tot = block_sum_256(dslope, sm);
    if (threadIdx.x == 0 && tot != 0.f) atomicAdd(dpw, tot);
  }
}

__global__ __launch_bounds__(256) void k_axpby(const float* __restrict__ a, const float* __restrict__ b,
                                               float* __restrict__ out, size_t n, float alpha, float beta) {
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 u = reinterpret_cast<const float4*>(a)[i];
    const float4 v = reinterpret_cast<const float4*>(b)[i];
    float4 o;
    o.x = alpha * u.x + beta * v.x;
    o.y = alpha * u.y + beta * v.y;
    o.z = alpha * u.z + beta * v.z;
    o.w = alpha * u.w + beta * v.w;
    reinterpret_cast<float4*>(out)[i] = o;
  }
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    out[i] = alpha * a[i] + beta * b[i];
}

__global__ __launch_bounds__(256) void k_scale_dev(const float* __restrict__ x, const float* __restrict__ alpha,
                                                   float* __restrict__ out, size_t n) {
  const float a = *alpha;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = a * x[i];
}

}  // namespace srk


namespace srk {
// max|x| -> the SRK_AMAX_FLOATS buffer (atomic max of non-negative floats as unsigned), one atomic per block
__global__ __launch_bounds__(256) void k_absmax(const float* __restrict__ x, size_t n, float* __restrict__ slots) {
  float a = 0.f;
  const size_t n4 = n / 4;
  typedef float am_f4 __attribute__((ext_vector_type(4)));
  if (((uintptr_t)x & 15) == 0) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
      const am_f4 v = reinterpret_cast<const am_f4*>(x)[i];
      a = fmaxf(fmaxf(a, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a = fmaxf(a, fabsf(x[i]));
  } else {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a = fmaxf(a, fabsf(x[i]));
  }
  __shared__ float sm[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a = fmaxf(a, __shfl_xor(a, o, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    a = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    float* p = slots + (blockIdx.x & 15) * 16;   // one slot per 64-byte line (SRK_AMAX_FLOATS)
    if (a > *p) atomicMax(reinterpret_cast<unsigned*>(p), __float_as_uint(a));
  }
}
}  // namespace srk

using namespace srk;

extern "C" int srk_scale_dev(const float* x, const float* alpha_dev, float* out, size_t n, void* stream) {
  SRK_REQUIRE(x && alpha_dev && out && n > 0, "scale_dev: null pointer or empty");
  hipLaunchKernelGGL(k_scale_dev, dim3(ew_grid(n, 256 * 4)), dim3(256), 0, (hipStream_t)stream, x, alpha_dev, out, n);
  return check_launch("scale_dev");
}

extern "C" int srk_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, void* stream) {
  return layout_launch(true, x, y, N, C, H, W, (hipStream_t)stream);
}
extern "C" int srk_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, void* stream) {
  return layout_launch(false, x, y, N, C, H, W, (hipStream_t)stream);
}

extern "C" int srk_pixel_shuffle_forward(const float* x, float* y, int N, int H, int W, int C, int r, void* stream) {
  SRK_REQUIRE(x && y, "pixel_shuffle_forward: null pointer");
  SRK_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && r >= 1, "pixel_shuffle_forward: bad dims");
  const size_t total = (size_t)N * H * W * C * r * r;
  if (ps_tile(x, y, N, H, W, C, r, true, (hipStream_t)stream)) return check_launch("pixel_shuffle_forward");
  hipLaunchKernelGGL(k_pixel_shuffle<true>, dim3(ew_grid(total, 256 * 4)), dim3(256), 0, (hipStream_t)stream, x, y, H,
                     W, C, r, total);
  return check_launch("pixel_shuffle_forward");
}
extern "C" int srk_pixel_shuffle_backward(const float* dy, float* dx, int N, int H, int W, int C, int r,
                                          void* stream) {
  SRK_REQUIRE(dy && dx, "pixel_shuffle_backward: null pointer");
  SRK_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && r >= 1, "pixel_shuffle_backward: bad dims");
  const size_t total = (size_t)N * H * W * C * r * r;
  if (ps_tile(dy, dx, N, H, W, C, r, false, (hipStream_t)stream)) return check_launch("pixel_shuffle_backward");
  hipLaunchKernelGGL(k_pixel_shuffle<false>, dim3(ew_grid(total, 256 * 4)), dim3(256), 0, (hipStream_t)stream, dy, dx,
                     H, W, C, r, total);
  return check_launch("pixel_shuffle_backward");
}

extern "C" int srk_act_forward(const float* x, float* y, size_t n, int channels, int act, float slope,
                               const float* prelu_weight, int prelu_n, void* stream) {
  SRK_REQUIRE(x && y && n > 0, "act_forward: null pointer or empty");
  SRK_REQUIRE(act >= SRK_ACT_NONE && act <= SRK_ACT_SIGMOID, "act_forward: unknown act %d", act);
  if (act == SRK_ACT_PRELU) {
    SRK_REQUIRE(prelu_weight && prelu_n >= 1, "act_forward: PReLU needs its weight");
    SRK_REQUIRE(prelu_n == 1 || prelu_n == channels, "act_forward: prelu_n %d != channels %d", prelu_n, channels);
  }
  SRK_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0), "act_forward: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(k_act_fwd, dim3(ew_grid(n, 256 * 8)), dim3(256), 0, (hipStream_t)stream, x, y, n, channels, act,
                     slope, prelu_weight, prelu_n);
  return check_launch("act_forward");
}

extern "C" int srk_act_backward(const float* dy, const float* saved, float* dx, size_t n, int channels, int act,
                                float slope, const float* prelu_weight, int prelu_n, float* dprelu, void* stream) {
  SRK_REQUIRE(dy && saved && dx && n > 0, "act_backward: null pointer or empty");
  SRK_REQUIRE(act >= SRK_ACT_NONE && act <= SRK_ACT_SIGMOID, "act_backward: unknown act %d", act);
  if (act == SRK_ACT_PRELU) {
    SRK_REQUIRE(prelu_weight && prelu_n >= 1, "act_backward: PReLU needs its weight");
    SRK_REQUIRE(prelu_n == 1 || prelu_n == channels, "act_backward: prelu_n %d != channels %d", prelu_n, channels);
  }
  const bool relu_family = act == SRK_ACT_RELU || act == SRK_ACT_LRELU || (act == SRK_ACT_PRELU && prelu_n == 1);
  if (relu_family && (n & 3) == 0 && (((uintptr_t)dy | (uintptr_t)saved | (uintptr_t)dx) & 15) == 0) {
    hipLaunchKernelGGL(k_act_bwd4, dim3(ew_grid(n / 4, 256 * 2)), dim3(256), 0, (hipStream_t)stream, dy, saved, dx, n / 4,
                       act, slope, prelu_weight, dprelu);
    return check_launch("act_backward");
  }
  hipLaunchKernelGGL(k_act_bwd, dim3(ew_grid(n, 256 * 8)), dim3(256), 0, (hipStream_t)stream, dy, saved, dx, n,
                     channels, act, slope, prelu_weight, prelu_n, dprelu);
  return check_launch("act_backward");
}

extern "C" int srk_axpby(const float* a, const float* b, float* out, size_t n, float alpha, float beta, void* stream) {
  SRK_REQUIRE(a && b && out && n > 0, "axpby: null pointer or empty");
  SRK_REQUIRE(((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0) && ((uintptr_t)out % 16 == 0),
              "axpby: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(k_axpby, dim3(ew_grid(n, 256 * 8)), dim3(256), 0, (hipStream_t)stream, a, b, out, n, alpha, beta);
  return check_launch("axpby");
}

extern "C" int srk_absmax(const float* x, size_t n, float* amax_slots, void* stream) {
  SRK_REQUIRE(x && amax_slots && n > 0, "absmax: null pointer or empty");
  size_t b = (n + 256 * 16 - 1) / (256 * 16);
  if (b > 1024) b = 1024;
  hipLaunchKernelGGL(k_absmax, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, x, n, amax_slots);
  return check_launch("absmax");
}
