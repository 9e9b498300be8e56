// Steps either side of the nets that the reference runs on the host every iteration / every test image
// (SURVEY.md §8 f2, a5): PSNR (utils.py:208-216), norm / denorm (utils.py:219-239), the nearest x2 resize of
// Upsample2xBlock('rnc') (base_networks.py:204-210), the 2x2 max-pool of the VGG19 feature extractor
// (srgan.py:84-90) and per-sample statistics for norm='instance' (base_networks.py:48,83,119,163).
// All HBM-bound streaming kernels; grid-stride loops, 16-byte accesses where the layout allows.
#include "srk_common.h"

namespace srk {

constexpr int kPsnrPartials = 1024;

static inline unsigned pp_grid(size_t items, int per_block) {
  size_t b = (items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > 4096) b = 4096;
  return (unsigned)b;
}

// ---------------------------------------------------------------------------------------------
// PSNR: mse = mean((clamp(pred,0,1) - gt)^2); psnr = mse == 0 ? 100 : 10*log10(1/mse).  pred is addressed
// NHWC-dense or through element strides like the target (a net output is channels_last, a loader's target NCHW).
// ---------------------------------------------------------------------------------------------
struct Strides4 {
  int64_t n, c, h, w;
};

__global__ __launch_bounds__(256) void k_psnr_partial(const float* __restrict__ pred, Strides4 ps,
                                                      const float* __restrict__ gt, Strides4 gs, int C, int H, int W,
                                                      size_t total, double* __restrict__ partials) {
  __shared__ double sm[4];
  double acc = 0.0;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    // e enumerates (n, h, w, c) with c fastest
    const int c = (int)(e % C);
    size_t t = e / C;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int64_t n = (int64_t)(t / H);
    float p = pred[n * ps.n + c * ps.c + h * ps.h + w * ps.w];
    p = fminf(fmaxf(p, 0.f), 1.f);
    const float d = p - gt[n * gs.n + c * gs.c + h * gs.h + w * gs.w];  // fp32 difference, as the reference
    acc += (double)d * (double)d;
  }
  const double tot = block_sum_256_d(acc, sm);
  if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void k_psnr_final(const double* __restrict__ partials, int nparts, double inv_count,
                                                    float* __restrict__ psnr, float* __restrict__ mse_out) {
  __shared__ double sm[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += partials[i];
  const double tot = block_sum_256_d(acc, sm);
  if (threadIdx.x == 0) {
    const double mse = tot * inv_count;
    if (mse_out) *mse_out = (float)mse;
    *psnr = mse == 0.0 ? 100.f : (float)(10.0 * log10(1.0 / mse));
  }
}

// ---------------------------------------------------------------------------------------------
// y = (x - sub[c]) / div[c], optionally clamped to [0,1].  torchvision.transforms.Normalize is
// tensor.sub_(mean).div_(std) in fp32: the same two IEEE operations here, so results are bit-equal.
// channel of element e = (e / inner) % C  (inner = H*W for NCHW storage, 1 for NHWC storage).
// ---------------------------------------------------------------------------------------------
struct AffineConsts {
  float sub[8], div[8];
};

__global__ __launch_bounds__(256) void k_channel_affine(const float* __restrict__ x, float* __restrict__ y, size_t total,
                                                        int C, size_t inner, AffineConsts k, int clamp01) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int c = (int)((e / inner) % C);
    float v = (x[e] - k.sub[c]) / k.div[c];
    if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
    y[e] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// Nearest-neighbour integer up-sampling, NHWC: y[n, oy, ox, c] = x[n, oy / r, ox / r, c]
// (torch.nn.Upsample(scale_factor=r, mode='nearest')); backward sums each r x r block.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_upsample_nearest_fwd(const float* __restrict__ x, float* __restrict__ y, int H,
                                                              int W, int C4, int r, size_t total4) {
  typedef float lf4 __attribute__((ext_vector_type(4)));
  const int OW = W * r, OH = H * r;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total4; e += (size_t)gridDim.x * 256) {
    const int c = (int)(e % C4);
    size_t t = e / C4;
    const int ox = (int)(t % OW);
    t /= OW;
    const int oy = (int)(t % OH);
    const size_t n = t / OH;
    const size_t src = ((n * H + oy / r) * W + ox / r) * C4 + c;
    reinterpret_cast<lf4*>(y)[e] = reinterpret_cast<const lf4*>(x)[src];
  }
}

__global__ __launch_bounds__(256) void k_upsample_nearest_fwd1(const float* __restrict__ x, float* __restrict__ y, int H,
                                                               int W, int C, int r, size_t total) {
  const int OW = W * r, OH = H * r;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int c = (int)(e % C);
    size_t t = e / C;
    const int ox = (int)(t % OW);
    t /= OW;
    const int oy = (int)(t % OH);
    const size_t n = t / OH;
    y[e] = x[((n * H + oy / r) * W + ox / r) * C + c];
  }
}

__global__ __launch_bounds__(256) void k_upsample_nearest_bwd(const float* __restrict__ dy, float* __restrict__ dx,
                                                              int H, int W, int C, int r, size_t total) {
  const int OW = W * r;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int c = (int)(e % C);
    size_t t = e / C;
    const int xx = (int)(t % W);
    t /= W;
    const int yy = (int)(t % H);
    const size_t n = t / H;
    float acc = 0.f;
    for (int i = 0; i < r; ++i)
      for (int j = 0; j < r; ++j) acc += dy[((n * H * r + (size_t)yy * r + i) * OW + (size_t)xx * r + j) * C + c];
    dx[e] = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// MaxPool2d(kernel 2, stride 2), NHWC, floor mode (odd trailing row / column dropped) — vgg19.features[4].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_maxpool2(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C,
                                                  size_t total) {
  const int OH = H / 2, OW = W / 2;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int c = (int)(e % C);
    size_t t = e / C;
    const int ox = (int)(t % OW);
    t /= OW;
    const int oy = (int)(t % OH);
    const size_t n = t / OH;
    const float* p = x + ((n * H + (size_t)oy * 2) * W + (size_t)ox * 2) * C + c;
    const float a = p[0], b = p[C], d = p[(size_t)W * C], f = p[(size_t)W * C + C];
    // NaN-propagating like ATen's max_pool2d: a NaN anywhere in the window wins
    float m = a;
    if (b > m || b != b) m = b;
    if (d > m || d != d) m = d;
    if (f > m || f != f) m = f;
    y[e] = m;
  }
}

}  // namespace srk

using namespace srk;

extern "C" size_t srk_psnr_workspace_bytes(void) { return kPsnrPartials * sizeof(double); }

extern "C" int srk_psnr(const float* pred, const int64_t* pred_strides, const float* gt, const int64_t* gt_strides, int N,
                        int C, int H, int W, float* psnr_out, float* mse_out, void* workspace, void* stream) {
  SRK_REQUIRE(pred && gt && psnr_out && workspace, "psnr: null pointer");
  SRK_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, "psnr: bad dims");
  const Strides4 dense = {(int64_t)H * W * C, 1, (int64_t)W * C, C};
  Strides4 ps = dense, gs = dense;
  if (pred_strides) ps = {pred_strides[0], pred_strides[1], pred_strides[2], pred_strides[3]};
  if (gt_strides) gs = {gt_strides[0], gt_strides[1], gt_strides[2], gt_strides[3]};
  const size_t total = (size_t)N * C * H * W;
  unsigned nb = pp_grid(total, 256 * 8);
  if (nb > kPsnrPartials) nb = kPsnrPartials;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_psnr_partial, dim3(nb), dim3(256), 0, s, pred, ps, gt, gs, C, H, W, total, (double*)workspace);
  hipLaunchKernelGGL(k_psnr_final, dim3(1), dim3(256), 0, s, (const double*)workspace, (int)nb, 1.0 / (double)total,
                     psnr_out, mse_out);
  return check_launch("psnr");
}

extern "C" int srk_channel_affine(const float* x, float* y, size_t n, int C, size_t inner, const float* sub_host,
                                  const float* div_host, int clamp01, void* stream) {
  SRK_REQUIRE(x && y && sub_host && div_host && n > 0, "channel_affine: null pointer or empty");
  SRK_REQUIRE(C >= 1 && C <= 8 && inner >= 1, "channel_affine: 1..8 channels supported (got %d)", C);
  AffineConsts k;
  for (int c = 0; c < 8; ++c) {
    k.sub[c] = c < C ? sub_host[c] : 0.f;
    k.div[c] = c < C ? div_host[c] : 1.f;
    SRK_REQUIRE(k.div[c] != 0.f, "channel_affine: zero divisor for channel %d", c);
  }
  hipLaunchKernelGGL(k_channel_affine, dim3(pp_grid(n, 256 * 4)), dim3(256), 0, (hipStream_t)stream, x, y, n, C, inner, k,
                     clamp01);
  return check_launch("channel_affine");
}

extern "C" int srk_upsample_nearest_forward(const float* x, float* y, int N, int H, int W, int C, int r, void* stream) {
  SRK_REQUIRE(x && y, "upsample_nearest: null pointer");
  SRK_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && r >= 1, "upsample_nearest: bad dims");
  const size_t total = (size_t)N * H * r * W * r * C;
  hipStream_t s = (hipStream_t)stream;
  if (C % 4 == 0)
    hipLaunchKernelGGL(k_upsample_nearest_fwd, dim3(pp_grid(total / 4, 256 * 2)), dim3(256), 0, s, x, y, H, W, C / 4, r,
                       total / 4);
  else
    hipLaunchKernelGGL(k_upsample_nearest_fwd1, dim3(pp_grid(total, 256 * 4)), dim3(256), 0, s, x, y, H, W, C, r, total);
  return check_launch("upsample_nearest_forward");
}

extern "C" int srk_upsample_nearest_backward(const float* dy, float* dx, int N, int H, int W, int C, int r,
                                             void* stream) {
  SRK_REQUIRE(dy && dx, "upsample_nearest_backward: null pointer");
  SRK_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && r >= 1, "upsample_nearest_backward: bad dims");
  const size_t total = (size_t)N * H * W * C;
  hipLaunchKernelGGL(k_upsample_nearest_bwd, dim3(pp_grid(total, 256 * 2)), dim3(256), 0, (hipStream_t)stream, dy, dx, H,
                     W, C, r, total);
  return check_launch("upsample_nearest_backward");
}

extern "C" int srk_maxpool2x2_forward(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  SRK_REQUIRE(x && y, "maxpool2x2: null pointer");
  SRK_REQUIRE(N > 0 && H >= 2 && W >= 2 && C > 0, "maxpool2x2: bad dims");
  const size_t total = (size_t)N * (H / 2) * (W / 2) * C;
  hipLaunchKernelGGL(k_maxpool2, dim3(pp_grid(total, 256 * 4)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, C, total);
  return check_launch("maxpool2x2_forward");
}
