// utils.img_interp (utils.py:242-269) on the GPU, bit-exact with the reference's CPU path.
//
// The reference resizes every image of a batch in a Python loop through
//     ToPILImage()            float [0,1] -> uint8 by  pic.mul(255).byte()        (truncation)
//     Image.resize(BICUBIC)   Pillow's two-pass separable resampler on 8-bit pixels
//     ToTensor()              uint8 -> float by  /255
// and SRCNN / VDSR / LapSRN call it on every iteration (srcnn.py:119, vdsr.py:137, lapsrn.py:183).
// Pillow's resampler (src/libImaging/Resample.c; the algorithm is unchanged since Pillow 4 and is the
// one in the Pillow 12.2 of this image) is integer arithmetic on uint8 data:
//   * per output coordinate xx: center = (xx + 0.5) * scale, support = filter_support * max(scale, 1),
//     xmin = (int)(center - support + 0.5) clamped to 0, xmax = (int)(center + support + 0.5) clamped to
//     the input size; weights w = filter((x + xmin - center + 0.5) / max(scale, 1)) in double,
//     normalised by their sum, then quantised to 22-bit fixed point with round-half-away;
//   * a pass accumulates  ss = 2^21 + sum pixel * k  in int32 and stores clip8(ss >> 22);
//   * horizontal pass first (uint8 intermediate), then vertical.
// The coefficient tables are produced on the device by k_resize_tables in IEEE double with
// contraction off — the same operations in the same order as the C source — so they are identical
// to Pillow's; everything after that is integer.  Result: bit-equal to the reference (tested against
// Pillow itself).  Filters: bicubic (a = -0.5), bilinear; nearest uses Pillow's affine/nearest rule.
#include "srk_common.h"
#include <math.h>

namespace srk {

constexpr int kPrecisionBits = 32 - 8 - 2;

__device__ __forceinline__ double pil_filter(int filter, double x) {
#pragma clang fp contract(off)
  if (x < 0.0) x = -x;
  if (filter == SRK_INTERP_BILINEAR) return x < 1.0 ? 1.0 - x : 0.0;
  const double a = -0.5;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

// precompute_coeffs + normalize_coeffs_8bpc of Resample.c for one axis; one thread per output index
__global__ void k_resize_tables(int inSize, int outSize, int filter, int ksize, int* __restrict__ kk,
                                int* __restrict__ bounds) {
#pragma clang fp contract(off)
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  if (xx >= outSize) return;
  const double fsupport = filter == SRK_INTERP_BILINEAR ? 1.0 : 2.0;
  const double scale = (double)((float)inSize - 0.0f) / outSize;
  double filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = fsupport * filterscale;
  const double center = 0.0 + (xx + 0.5) * scale;
  const double ss = 1.0 / filterscale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > inSize) xmax = inSize;
  xmax -= xmin;
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) ww += pil_filter(filter, (x + xmin - center + 0.5) * ss);
  int* k = kk + (size_t)xx * ksize;
  for (int x = 0; x < ksize; ++x) {
    double w = 0.0;
    if (x < xmax) {
      w = pil_filter(filter, (x + xmin - center + 0.5) * ss);
      if (ww != 0.0) w /= ww;
    }
    k[x] = w < 0 ? (int)(-0.5 + w * (1 << kPrecisionBits)) : (int)(0.5 + w * (1 << kPrecisionBits));
  }
  bounds[2 * xx] = xmin;
  bounds[2 * xx + 1] = xmax;
}

__device__ __forceinline__ unsigned char to_u8(float v) {  // pic.mul(255).byte(): fp32 multiply, truncation
  const float s = v * 255.f;
  return (unsigned char)(s <= 0.f ? 0 : (s >= 255.f ? 255 : (int)s));
}

__device__ __forceinline__ unsigned char clip8(int ss) {
  const int v = ss >> kPrecisionBits;  // arithmetic shift, as Pillow's lookup index
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: float NCHW rows -> uint8 rows of the output width
__global__ __launch_bounds__(256) void k_resize_h(const float* __restrict__ x, unsigned char* __restrict__ tmp,
                                                  size_t rows, int W, int OW, int ksize, const int* __restrict__ kk,
                                                  const int* __restrict__ bounds) {
  const size_t total = rows * (size_t)OW;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t row = e / OW;
    const int ox = (int)(e - row * OW);
    const int xmin = bounds[2 * ox], xmax = bounds[2 * ox + 1];
    const int* k = kk + (size_t)ox * ksize;
    const float* src = x + row * W + xmin;
    int ss = 1 << (kPrecisionBits - 1);
    for (int t = 0; t < xmax; ++t) ss += (int)to_u8(src[t]) * k[t];
    tmp[e] = clip8(ss);
  }
}

// horizontal pass on 8-bit pixels addressed through element strides (plane, row, pixel): a decoded interleaved HWC
// image is read in place (plane stride 1, pixel stride 3), planar images with (H*W, W, 1)
__global__ __launch_bounds__(256) void k_resize_h_u8(const unsigned char* __restrict__ x, long long sp, long long sr,
                                                     long long sx, unsigned char* __restrict__ tmp, int planes, int H,
                                                     int OW, int ksize, const int* __restrict__ kk,
                                                     const int* __restrict__ bounds) {
  const size_t total = (size_t)planes * H * OW;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int ox = (int)(e % OW);
    const size_t r = e / OW;
    const int row = (int)(r % H);
    const int pl = (int)(r / H);
    const int xmin = bounds[2 * ox], xmax = bounds[2 * ox + 1];
    const int* k = kk + (size_t)ox * ksize;
    const unsigned char* src = x + pl * sp + row * sr + xmin * sx;
    int ss = 1 << (kPrecisionBits - 1);
    for (int t = 0; t < xmax; ++t) ss += (int)src[t * sx] * k[t];
    tmp[e] = clip8(ss);
  }
}

// vertical pass with an 8-bit result (the image stays a PIL-style uint8 image for the next transform)
__global__ __launch_bounds__(256) void k_resize_v_u8(const unsigned char* __restrict__ tmp, unsigned char* __restrict__ y,
                                                     size_t planes, int H, int OH, int OW, int ksize,
                                                     const int* __restrict__ kk, const int* __restrict__ bounds) {
  const size_t total = planes * (size_t)OH * OW;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int ox = (int)(e % OW);
    const size_t r = e / OW;
    const int oy = (int)(r % OH);
    const size_t pl = r / OH;
    const int ymin = bounds[2 * oy], ymax = bounds[2 * oy + 1];
    const int* k = kk + (size_t)oy * ksize;
    const unsigned char* src = tmp + (pl * H + ymin) * (size_t)OW + ox;
    int ss = 1 << (kPrecisionBits - 1);
    for (int t = 0; t < ymax; ++t) ss += (int)src[(size_t)t * OW] * k[t];
    y[e] = clip8(ss);
  }
}

// gather copy of 8-bit planes through element strides (a pass Pillow skips because the size does not change)
__global__ __launch_bounds__(256) void k_copy_u8_strided(const unsigned char* __restrict__ x, long long sp, long long sr,
                                                         long long sx, unsigned char* __restrict__ y, int planes, int H,
                                                         int W) {
  const size_t total = (size_t)planes * H * W;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int xx = (int)(e % W);
    const size_t r = e / W;
    y[e] = x[(long long)(r / H) * sp + (long long)(r % H) * sr + xx * sx];
  }
}

// planar uint8 -> float / 255 (ToTensor)
__global__ __launch_bounds__(256) void k_u8_to_float(const unsigned char* __restrict__ x, float* __restrict__ y, size_t n) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) y[e] = (float)x[e] / 255.f;
}

// crop -> rotate by k x 90 degrees counter-clockwise (PIL Image.rotate(90*k, expand=True) = transpose(ROTATE_*)) ->
// horizontal flip -> vertical flip (dataset.py:65-84), composed into one gather: out [C][OH][OW] planar.
__global__ __launch_bounds__(256) void k_patch_augment(const unsigned char* __restrict__ x, long long sp, long long sr,
                                                       long long sx, unsigned char* __restrict__ y, int C, int cx0, int cy0,
                                                       int cw, int ch, int rot, int fliplr, int fliptb) {
  const int OW = (rot & 1) ? ch : cw, OH = (rot & 1) ? cw : ch;
  const size_t total = (size_t)C * OH * OW;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    int ox = (int)(e % OW);
    const size_t r = e / OW;
    int oy = (int)(r % OH);
    const int c = (int)(r / OH);
    if (fliptb) oy = OH - 1 - oy;   // undo the flips (the last transform applied comes off first)
    if (fliplr) ox = OW - 1 - ox;
    int iy, ix;                     // position in the cropped image A [ch][cw]: R = rot90^k(A)
    switch (rot & 3) {
      case 1: iy = ox; ix = cw - 1 - oy; break;            // R[i][j] = A[j][cw-1-i]
      case 2: iy = ch - 1 - oy; ix = cw - 1 - ox; break;
      case 3: iy = ch - 1 - ox; ix = oy; break;            // R[i][j] = A[ch-1-j][i]
      default: iy = oy; ix = ox;
    }
    y[e] = x[c * sp + (long long)(cy0 + iy) * sr + (long long)(cx0 + ix) * sx];
  }
}

// vertical pass: uint8 [planes][H][OW] -> float [planes][OH][OW] (ToTensor: /255)
__global__ __launch_bounds__(256) void k_resize_v(const unsigned char* __restrict__ tmp, float* __restrict__ y,
                                                  size_t planes, int H, int OH, int OW, int ksize,
                                                  const int* __restrict__ kk, const int* __restrict__ bounds) {
  const size_t total = planes * (size_t)OH * OW;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int ox = (int)(e % OW);
    const size_t r = e / OW;
    const int oy = (int)(r % OH);
    const size_t pl = r / OH;
    const int ymin = bounds[2 * oy], ymax = bounds[2 * oy + 1];
    const int* k = kk + (size_t)oy * ksize;
    const unsigned char* src = tmp + (pl * H + ymin) * (size_t)OW + ox;
    int ss = 1 << (kPrecisionBits - 1);
    for (int t = 0; t < ymax; ++t) ss += (int)src[(size_t)t * OW] * k[t];
    y[e] = (float)clip8(ss) / 255.f;
  }
}

// Image.NEAREST: Pillow's ImagingScaleAffine — xo = scale * 0.5, then xo += scale per output column (a running
// double sum, reproduced sequentially so that inexact scales round identically), xin = (int)xo.
__global__ void k_nearest_tables(int W, int OW, int H, int OH, int* __restrict__ xtab, int* __restrict__ ytab) {
#pragma clang fp contract(off)
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const double ax = (double)((float)W - 0.0f) / OW, ay = (double)((float)H - 0.0f) / OH;
  double xo = 0.0 + ax * 0.5;
  for (int x = 0; x < OW; ++x) {
    int xin = xo < 0.0 ? -1 : (int)xo;
    if (xin > W - 1) xin = W - 1;
    xtab[x] = xin < 0 ? 0 : xin;
    xo += ax;
  }
  double yo = 0.0 + ay * 0.5;
  for (int y = 0; y < OH; ++y) {
    int yin = yo < 0.0 ? -1 : (int)yo;
    if (yin > H - 1) yin = H - 1;
    ytab[y] = yin < 0 ? 0 : yin;
    yo += ay;
  }
}

__global__ __launch_bounds__(256) void k_resize_nearest(const float* __restrict__ x, float* __restrict__ y,
                                                        size_t planes, int H, int W, int OH, int OW,
                                                        const int* __restrict__ xtab, const int* __restrict__ ytab) {
  const size_t total = planes * (size_t)OH * OW;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int ox = (int)(e % OW);
    const size_t r = e / OW;
    const int oy = (int)(r % OH);
    const size_t pl = r / OH;
    y[e] = (float)to_u8(x[(pl * H + ytab[oy]) * (size_t)W + xtab[ox]]) / 255.f;
  }
}

static int resize_ksize(int inSize, int outSize, int filter) {
  const double fsupport = filter == SRK_INTERP_BILINEAR ? 1.0 : 2.0;
  double filterscale = (double)((float)inSize) / outSize;
  if (filterscale < 1.0) filterscale = 1.0;
  return (int)ceil(fsupport * filterscale) * 2 + 1;
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace srk

using namespace srk;

extern "C" size_t srk_img_interp_workspace_bytes(int N, int C, int H, int W, int OH, int OW, int filter) {
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return 0;
  if (filter == SRK_INTERP_NEAREST) return align256((size_t)OW * 4) + align256((size_t)OH * 4);
  const size_t kh = (size_t)resize_ksize(W, OW, filter), kv = (size_t)resize_ksize(H, OH, filter);
  return align256((size_t)N * C * H * OW) + align256((size_t)OW * kh * 4) + align256((size_t)OW * 8) +
         align256((size_t)OH * kv * 4) + align256((size_t)OH * 8);
}

extern "C" int srk_img_interp(const float* x_nchw, float* y_nchw, int N, int C, int H, int W, int OH, int OW, int filter,
                              void* workspace, size_t workspace_bytes, void* stream) {
  SRK_REQUIRE(x_nchw && y_nchw, "img_interp: null tensor pointer");
  SRK_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "img_interp: non-positive dims");
  SRK_REQUIRE(filter == SRK_INTERP_NEAREST || filter == SRK_INTERP_BILINEAR || filter == SRK_INTERP_BICUBIC,
              "img_interp: unknown filter %d", filter);
  hipStream_t s = (hipStream_t)stream;
  const size_t planes = (size_t)N * C;
  const size_t need = srk_img_interp_workspace_bytes(N, C, H, W, OH, OW, filter);
  SRK_REQUIRE(workspace && workspace_bytes >= need, "img_interp: workspace %zu < %zu", workspace_bytes, need);
  if (filter == SRK_INTERP_NEAREST) {
    int* xtab = static_cast<int*>(workspace);
    int* ytab = reinterpret_cast<int*>(static_cast<char*>(workspace) + align256((size_t)OW * 4));
    hipLaunchKernelGGL(k_nearest_tables, dim3(1), dim3(1), 0, s, W, OW, H, OH, xtab, ytab);
    size_t nb = (planes * OH * OW + 255) / 256;
    if (nb > 65535) nb = 65535;
    hipLaunchKernelGGL(k_resize_nearest, dim3((unsigned)nb), dim3(256), 0, s, x_nchw, y_nchw, planes, H, W, OH, OW, xtab,
                       ytab);
    return check_launch("img_interp(nearest)");
  }
  const int kh = resize_ksize(W, OW, filter), kv = resize_ksize(H, OH, filter);
  char* p = static_cast<char*>(workspace);
  unsigned char* tmp = reinterpret_cast<unsigned char*>(p);
  p += align256(planes * H * OW);
  int* kkh = reinterpret_cast<int*>(p);
  p += align256((size_t)OW * kh * 4);
  int* bh = reinterpret_cast<int*>(p);
  p += align256((size_t)OW * 8);
  int* kkv = reinterpret_cast<int*>(p);
  p += align256((size_t)OH * kv * 4);
  int* bv = reinterpret_cast<int*>(p);
  hipLaunchKernelGGL(k_resize_tables, dim3(cdiv(OW, 128)), dim3(128), 0, s, W, OW, filter, kh, kkh, bh);
  hipLaunchKernelGGL(k_resize_tables, dim3(cdiv(OH, 128)), dim3(128), 0, s, H, OH, filter, kv, kkv, bv);
  size_t nb = (planes * H * OW + 255) / 256;
  if (nb > 65535) nb = 65535;
  hipLaunchKernelGGL(k_resize_h, dim3((unsigned)nb), dim3(256), 0, s, x_nchw, tmp, planes * H, W, OW, kh, kkh, bh);
  nb = (planes * OH * OW + 255) / 256;
  if (nb > 65535) nb = 65535;
  hipLaunchKernelGGL(k_resize_v, dim3((unsigned)nb), dim3(256), 0, s, tmp, y_nchw, planes, H, OH, OW, kv, kkv, bv);
  return check_launch("img_interp");
}

// ---- 8-bit image transforms of the training-set pipeline (dataset.py:51-99) ------------------------------------------
extern "C" size_t srk_img_resize_u8_workspace_bytes(int planes, int H, int W, int OH, int OW, int filter) {
  if (planes <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return 0;
  const size_t kh = (size_t)resize_ksize(W, OW, filter), kv = (size_t)resize_ksize(H, OH, filter);
  return align256((size_t)planes * H * OW) + align256((size_t)planes * OH * OW) + align256((size_t)OW * kh * 4) +
         align256((size_t)OW * 8) + align256((size_t)OH * kv * 4) + align256((size_t)OH * 8);
}

extern "C" int srk_img_resize_u8(const uint8_t* x, int64_t plane_stride, int64_t row_stride, int64_t px_stride, void* y,
                                 int out_float, int planes, int H, int W, int OH, int OW, int filter, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  SRK_REQUIRE(x && y, "img_resize_u8: null pointer");
  SRK_REQUIRE(planes > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "img_resize_u8: non-positive dims");
  SRK_REQUIRE(filter == SRK_INTERP_BILINEAR || filter == SRK_INTERP_BICUBIC, "img_resize_u8: filter must be bilinear or bicubic");
  const size_t need = srk_img_resize_u8_workspace_bytes(planes, H, W, OH, OW, filter);
  SRK_REQUIRE(workspace && workspace_bytes >= need, "img_resize_u8: workspace %zu < %zu", workspace_bytes, need);
  hipStream_t s = (hipStream_t)stream;
  const int kh = resize_ksize(W, OW, filter), kv = resize_ksize(H, OH, filter);
  char* p = static_cast<char*>(workspace);
  unsigned char* tmp = reinterpret_cast<unsigned char*>(p);
  p += align256((size_t)planes * H * OW);
  unsigned char* out8 = reinterpret_cast<unsigned char*>(p);
  p += align256((size_t)planes * OH * OW);
  int* kkh = reinterpret_cast<int*>(p);
  p += align256((size_t)OW * kh * 4);
  int* bh = reinterpret_cast<int*>(p);
  p += align256((size_t)OW * 8);
  int* kkv = reinterpret_cast<int*>(p);
  p += align256((size_t)OH * kv * 4);
  int* bv = reinterpret_cast<int*>(p);
  auto grid = [](size_t n) {
    size_t nb = (n + 255) / 256;
    return dim3((unsigned)(nb > 65535 ? 65535 : (nb < 1 ? 1 : nb)));
  };
  // Pillow (ImagingResample) runs a pass only when that axis changes size
  const bool need_h = OW != W, need_v = OH != H;
  if (need_h) {
    hipLaunchKernelGGL(k_resize_tables, dim3(cdiv(OW, 128)), dim3(128), 0, s, W, OW, filter, kh, kkh, bh);
    hipLaunchKernelGGL(k_resize_h_u8, grid((size_t)planes * H * OW), dim3(256), 0, s, x, (long long)plane_stride,
                       (long long)row_stride, (long long)px_stride, tmp, planes, H, OW, kh, kkh, bh);
  } else {
    hipLaunchKernelGGL(k_copy_u8_strided, grid((size_t)planes * H * W), dim3(256), 0, s, x, (long long)plane_stride,
                       (long long)row_stride, (long long)px_stride, tmp, planes, H, W);
  }
  unsigned char* v_out = out_float ? out8 : static_cast<unsigned char*>(y);
  const unsigned char* last = tmp;
  if (need_v) {
    hipLaunchKernelGGL(k_resize_tables, dim3(cdiv(OH, 128)), dim3(128), 0, s, H, OH, filter, kv, kkv, bv);
    hipLaunchKernelGGL(k_resize_v_u8, grid((size_t)planes * OH * OW), dim3(256), 0, s, (const unsigned char*)tmp, v_out,
                       (size_t)planes, H, OH, OW, kv, kkv, bv);
    last = v_out;
  }
  const size_t n = (size_t)planes * OH * OW;
  if (out_float)
    hipLaunchKernelGGL(k_u8_to_float, grid(n), dim3(256), 0, s, last, static_cast<float*>(y), n);
  else if (last != v_out)
    hipLaunchKernelGGL(k_copy_u8_strided, grid(n), dim3(256), 0, s, last, (long long)OH * OW, (long long)OW, 1LL, v_out,
                       planes, OH, OW);
  return check_launch("img_resize_u8");
}

extern "C" int srk_patch_augment_u8(const uint8_t* x, int64_t plane_stride, int64_t row_stride, int64_t px_stride,
                                    uint8_t* y, int C, int H, int W, int crop_x, int crop_y, int crop_w, int crop_h,
                                    int rot_k, int fliplr, int fliptb, void* stream) {
  SRK_REQUIRE(x && y, "patch_augment_u8: null pointer");
  SRK_REQUIRE(C > 0 && H > 0 && W > 0 && crop_w > 0 && crop_h > 0, "patch_augment_u8: non-positive dims");
  SRK_REQUIRE(crop_x >= 0 && crop_y >= 0 && crop_x + crop_w <= W && crop_y + crop_h <= H,
              "patch_augment_u8: crop [%d,%d)+(%d,%d) outside the %dx%d image", crop_x, crop_y, crop_w, crop_h, W, H);
  SRK_REQUIRE(rot_k >= 0 && rot_k <= 3, "patch_augment_u8: rot_k must be 0..3 (quarter turns, counter-clockwise)");
  size_t nb = ((size_t)C * crop_w * crop_h + 255) / 256;
  if (nb > 65535) nb = 65535;
  hipLaunchKernelGGL(k_patch_augment, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, (long long)plane_stride,
                     (long long)row_stride, (long long)px_stride, y, C, crop_x, crop_y, crop_w, crop_h, rot_k, fliplr ? 1 : 0,
                     fliptb ? 1 : 0);
  return check_launch("patch_augment_u8");
}

// One call per patch for the training loader (dataset.py:51-82): optional whole-image rescale (Image.resize, BICUBIC)
// and the crop / quarter-turn / flips, from the interleaved 8-bit image straight into the batch's planar 8-bit patch.
// The host side of the loader is bound by Python calls per patch; this is three of them folded into one.
extern "C" size_t srk_patch_from_image_u8_workspace_bytes(int C, int H, int W, int scale_h, int scale_w) {
  if (C <= 0 || H <= 0 || W <= 0 || scale_h <= 0 || scale_w <= 0) return 256;
  return align256((size_t)C * scale_h * scale_w) + srk_img_resize_u8_workspace_bytes(C, H, W, scale_h, scale_w, SRK_INTERP_BICUBIC);
}

extern "C" int srk_patch_from_image_u8(const uint8_t* img_hwc, int C, int H, int W, int scale_h, int scale_w, int crop_x,
                                       int crop_y, int crop_w, int crop_h, int rot_k, int fliplr, int fliptb,
                                       uint8_t* out_planar, void* workspace, size_t workspace_bytes, void* stream) {
  SRK_REQUIRE(img_hwc && out_planar, "patch_from_image_u8: null pointer");
  SRK_REQUIRE(C > 0 && H > 0 && W > 0, "patch_from_image_u8: non-positive dims");
  if (scale_h <= 0 || scale_w <= 0)   // no rescale: crop straight from the interleaved image
    return srk_patch_augment_u8(img_hwc, 1, (int64_t)W * C, C, out_planar, C, H, W, crop_x, crop_y, crop_w, crop_h, rot_k,
                                fliplr, fliptb, stream);
  const size_t need = srk_patch_from_image_u8_workspace_bytes(C, H, W, scale_h, scale_w);
  SRK_REQUIRE(workspace && workspace_bytes >= need, "patch_from_image_u8: workspace %zu < %zu", workspace_bytes, need);
  uint8_t* scaled = static_cast<uint8_t*>(workspace);   // planar [C][scale_h][scale_w]
  const size_t off = align256((size_t)C * scale_h * scale_w);
  int rc = srk_img_resize_u8(img_hwc, 1, (int64_t)W * C, C, scaled, 0, C, H, W, scale_h, scale_w, SRK_INTERP_BICUBIC,
                             static_cast<char*>(workspace) + off, workspace_bytes - off, stream);
  if (rc) return rc;
  return srk_patch_augment_u8(scaled, (int64_t)scale_h * scale_w, scale_w, 1, out_planar, C, scale_h, scale_w, crop_x, crop_y,
                              crop_w, crop_h, rot_k, fliplr, fliptb, stream);
}
