// Scalar losses (mean reduction) fused with their gradient, flat-buffer optimizers and the
// global gradient-norm clip.  All HBM-bound streaming kernels.
#include "srk_common.h"

namespace srk {

constexpr int kMaxPartials = 1024;  // doubles

// ---------------------------------------------------------------------------------------------
// Loss forward + backward in one pass.
//   pred/dpred : NHWC dense, element e = ((n*H+h)*W+w)*C + c
//   target     : element strides (sn, sc, sh, sw); `contig` = target is NHWC dense too.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void loss_term(int kind, float p, float t, float eps, float& val, float& grad) {
  const float d = p - t;
  switch (kind) {
    case SRK_LOSS_MSE:
      val = d * d;
      grad = 2.f * d;
      break;
    case SRK_LOSS_L1:
      val = fabsf(d);
      grad = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      break;
    case SRK_LOSS_CHARBONNIER: {
      const float e = sqrtf(d * d + eps);  // lapsrn.py:81-85
      val = e;
      grad = d / e;
      break;
    }
    default: {  // BCE, torch semantics: logs clamped at -100, grad denominator clamped at 1e-12
      const float lp = fmaxf(logf(p), -100.f);
      const float l1p = fmaxf(logf(1.f - p), -100.f);
      val = -(t * lp + (1.f - t) * l1p);
      grad = (p - t) / fmaxf((1.f - p) * p, 1e-12f);
    }
  }
}

__global__ __launch_bounds__(256) void k_loss_partial(int kind, const float* __restrict__ pred,
                                                      const float* __restrict__ target, int64_t sn, int64_t sc,
                                                      int64_t sh, int64_t sw, int contig, int C, int H, int W,
                                                      size_t total, float eps, float gscale,
                                                      float* __restrict__ dpred, double* __restrict__ partials) {
  __shared__ double sm[4];
  float acc = 0.f;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    size_t ti = e;
    if (!contig) {
      const int c = (int)(e % C);
      size_t t = e / C;
      const int w = (int)(t % W);
      t /= W;
      const int h = (int)(t % H);
      const size_t n = t / H;
      ti = n * sn + c * sc + h * sh + w * sw;
    }
    float v, g;
    loss_term(kind, pred[e], target[ti], eps, v, g);
    acc += v;
    if (dpred) dpred[e] = g * gscale;
  }
  const double tot = block_sum_256_d((double)acc, sm);
  if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// 16-byte form for dense, aligned tensors (pred, target and dpred in one layout): four elements per thread and pass
typedef float loss_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_loss_partial4(int kind, const float* __restrict__ pred,
                                                       const float* __restrict__ target, size_t total4, float eps,
                                                       float gscale, float* __restrict__ dpred,
                                                       double* __restrict__ partials) {
  __shared__ double sm[4];
  float acc = 0.f;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total4; e += (size_t)gridDim.x * 256) {
    const loss_f4 p = reinterpret_cast<const loss_f4*>(pred)[e], t = reinterpret_cast<const loss_f4*>(target)[e];
    loss_f4 g;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float v, gg;
      loss_term(kind, p[k], t[k], eps, v, gg);
      acc += v;
      g[k] = gg * gscale;
    }
    if (dpred) reinterpret_cast<loss_f4*>(dpred)[e] = g;
  }
  const double tot = block_sum_256_d((double)acc, sm);
  if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void k_loss_final(const double* __restrict__ partials, int nparts, double inv_count,
                                                    float* __restrict__ loss) {
  __shared__ double sm[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += partials[i];
  const double tot = block_sum_256_d(acc, sm);
  if (threadIdx.x == 0) *loss = (float)(tot * inv_count);
}

// ---------------------------------------------------------------------------------------------
// Optimizers
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sgd(float* __restrict__ p, const float* __restrict__ g,
                                             float* __restrict__ buf, size_t n, float lr, float mom, float wd,
                                             int nesterov, int first, const float* __restrict__ lr_dev,
                                             const float* __restrict__ gs_dev) {
  if (lr_dev) lr = *lr_dev;
  const float gs = gs_dev ? *gs_dev : 1.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float pi = p[i];
    float d = g[i] * gs;
    if (wd != 0.f) d += wd * pi;
    if (mom != 0.f) {
      float b = first ? d : buf[i] * mom + d;
      buf[i] = b;
      d = nesterov ? d + mom * b : b;
    }
    p[i] = pi - lr * d;
  }
}

// 16-byte form of k_sgd (n % 4 == 0, aligned buffers): the same operations per element, four elements per thread and pass
// (SRGAN-D's 23 M parameters: 184 -> ~100 us; the scalar loop moved 2.5 TB/s)
typedef float sgd_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_sgd4(float* __restrict__ p, const float* __restrict__ g,
                                              float* __restrict__ buf, size_t n4, float lr, float mom, float wd,
                                              int nesterov, int first, const float* __restrict__ lr_dev,
                                              const float* __restrict__ gs_dev) {
  if (lr_dev) lr = *lr_dev;
  const float gs = gs_dev ? *gs_dev : 1.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    sgd_f4 pi = reinterpret_cast<const sgd_f4*>(p)[i];
    const sgd_f4 gi = reinterpret_cast<const sgd_f4*>(g)[i];
    sgd_f4 bi = {0.f, 0.f, 0.f, 0.f};
    if (mom != 0.f && !first) bi = reinterpret_cast<const sgd_f4*>(buf)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float d = gi[e] * gs;
      if (wd != 0.f) d += wd * pi[e];
      if (mom != 0.f) {
        const float b = first ? d : bi[e] * mom + d;
        bi[e] = b;
        d = nesterov ? d + mom * b : b;
      }
      pi[e] = pi[e] - lr * d;
    }
    if (mom != 0.f) reinterpret_cast<sgd_f4*>(buf)[i] = bi;
    reinterpret_cast<sgd_f4*>(p)[i] = pi;
  }
}

// The step counter is advanced by the kernel itself (it used to be a one-thread launch of its own in front, ~5 us of a
// 1.2 ms train step): every block reads the count of the PREVIOUS steps before anything else and works with count + 1;
// the block that finishes last -- a ticket in step_dev[1]: it has seen every other block's arrival, so every block has
// read the old count -- stores count + 1 and clears the ticket.  The count is consumed by the next launch only.
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g,
                                              float* __restrict__ m, float* __restrict__ v, size_t n, float lr,
                                              float b1, float b2, float eps, float wd,
                                              int32_t* __restrict__ step_dev, const float* __restrict__ lr_dev,
                                              const float* __restrict__ gs_dev) {
  if (lr_dev) lr = *lr_dev;
  const float gs = gs_dev ? *gs_dev : 1.f;
  const int32_t t_i = __hip_atomic_load(step_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
  const float t = (float)t_i;
  // torch/optim/adam.py (_single_tensor_adam): bias corrections, step_size, denom
  const float bc1 = 1.f - powf(b1, t);
  const float bc2 = 1.f - powf(b2, t);
  const float step_size = lr / bc1;
  const float bc2_sqrt = sqrtf(bc2);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float pi = p[i];
    float gi = g[i] * gs;
    if (wd != 0.f) gi += wd * pi;
    float mi = m[i];
    mi = mi + (1.f - b1) * (gi - mi);  // exp_avg.lerp_(grad, 1 - beta1)
    float vi = v[i] * b2 + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
  }
  __syncthreads();   // (the whole block has read the old count)
  if (threadIdx.x == 0) {
    const int32_t arrived = atomicAdd(step_dev + 1, 1);
    if (arrived == (int32_t)gridDim.x - 1) {
      __hip_atomic_store(step_dev, t_i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(step_dev + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// 16-byte form of k_adam (n % 4 == 0, aligned buffers): the same operations per element, four elements per thread and
// pass, and at most two blocks per CU -- the arrival ticket is one atomic per block on one word, and ~740 of them in the
// scalar kernel's grid made the folded step counter cost what the separate launch had (EDSR: 19.4 vs 14.5 + 2.8 us).
typedef float adam_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_adam4(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                               float* __restrict__ v, size_t n4, float lr, float b1, float b2, float eps,
                                               float wd, int32_t* __restrict__ step_dev, const float* __restrict__ lr_dev,
                                               const float* __restrict__ gs_dev) {
  if (lr_dev) lr = *lr_dev;
  const float gs = gs_dev ? *gs_dev : 1.f;
  const int32_t t_i = __hip_atomic_load(step_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
  const float t = (float)t_i;
  const float bc1 = 1.f - powf(b1, t);
  const float bc2 = 1.f - powf(b2, t);
  const float step_size = lr / bc1;
  const float bc2_sqrt = sqrtf(bc2);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    adam_f4 pi = reinterpret_cast<const adam_f4*>(p)[i];
    const adam_f4 gi4 = reinterpret_cast<const adam_f4*>(g)[i];
    adam_f4 mi = reinterpret_cast<const adam_f4*>(m)[i];
    adam_f4 vi = reinterpret_cast<const adam_f4*>(v)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float gi = gi4[e] * gs;
      if (wd != 0.f) gi += wd * pi[e];
      mi[e] = mi[e] + (1.f - b1) * (gi - mi[e]);
      vi[e] = vi[e] * b2 + (1.f - b2) * gi * gi;
      const float denom = sqrtf(vi[e]) / bc2_sqrt + eps;
      pi[e] = pi[e] - step_size * (mi[e] / denom);
    }
    reinterpret_cast<adam_f4*>(m)[i] = mi;
    reinterpret_cast<adam_f4*>(v)[i] = vi;
    reinterpret_cast<adam_f4*>(p)[i] = pi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int32_t arrived = atomicAdd(step_dev + 1, 1);
    if (arrived == (int32_t)gridDim.x - 1) {
      __hip_atomic_store(step_dev, t_i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(step_dev + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ __launch_bounds__(256) void k_sqsum_partial(const float* __restrict__ g, size_t n,
                                                       double* __restrict__ partials) {
  __shared__ double sm[4];
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float x = g[i];
    acc += x * x;
  }
  const double tot = block_sum_256_d((double)acc, sm);
  if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void k_norm_final(const double* __restrict__ partials, int nparts, float max_norm,
                                                    float* __restrict__ norm_out, float* __restrict__ scale_out) {
  __shared__ double sm[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += partials[i];
  const double tot = block_sum_256_d(acc, sm);
  if (threadIdx.x == 0) {
    const float nrm = (float)sqrt(tot);
    if (norm_out) *norm_out = nrm;
    if (scale_out) {
      const float c = max_norm / (nrm + 1e-6f);  // torch.nn.utils.clip_grad_norm_
      *scale_out = c < 1.f ? c : 1.f;
    }
  }
}

static inline unsigned red_grid(size_t n) {
  size_t b = (n + 256 * 8 - 1) / (256 * 8);
  if (b < 1) b = 1;
  if (b > kMaxPartials) b = kMaxPartials;
  return (unsigned)b;
}

}  // namespace srk

using namespace srk;

extern "C" size_t srk_loss_workspace_bytes(void) { return kMaxPartials * sizeof(double); }
extern "C" size_t srk_grad_norm_workspace_bytes(void) { return kMaxPartials * sizeof(double); }

extern "C" int srk_loss_forward_backward(int kind, const float* pred, const float* target,
                                         const int64_t* target_strides, int N, int C, int H, int W, float eps,
                                         float grad_scale, float* loss, float* dpred, void* workspace, void* stream) {
  SRK_REQUIRE(pred && target && loss && workspace, "loss: null pointer");
  SRK_REQUIRE(kind >= SRK_LOSS_MSE && kind <= SRK_LOSS_BCE, "loss: unknown kind %d", kind);
  SRK_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0, "loss: bad dims");
  const size_t total = (size_t)N * C * H * W;
  int64_t sn = (int64_t)H * W * C, sc = 1, sh = (int64_t)W * C, sw = C;
  int contig = 1;
  if (target_strides) {
    contig = (target_strides[0] == sn || N == 1) && (target_strides[1] == sc || C == 1) &&
             (target_strides[2] == sh || H == 1) && (target_strides[3] == sw || W == 1);
    sn = target_strides[0]; sc = target_strides[1]; sh = target_strides[2]; sw = target_strides[3];
  }
  unsigned nb = red_grid(total);
  hipStream_t s = (hipStream_t)stream;
  const bool vec = contig && (total & 3) == 0 && (((uintptr_t)pred | (uintptr_t)target | (uintptr_t)dpred) & 15) == 0;
  if (vec) {   // (the summation order differs from the scalar kernel's: partials of 4-element groups)
    size_t b = (total / 4 + 255) / 256;
    nb = (unsigned)(b > kMaxPartials ? kMaxPartials : (b < 1 ? 1 : b));
    hipLaunchKernelGGL(k_loss_partial4, dim3(nb), dim3(256), 0, s, kind, pred, target, total / 4, eps,
                       grad_scale / (float)total, dpred, (double*)workspace);
  } else
  hipLaunchKernelGGL(k_loss_partial, dim3(nb), dim3(256), 0, s, kind, pred, target, sn, sc, sh, sw, contig, C, H, W,
                     total, eps, grad_scale / (float)total, dpred, (double*)workspace);
  hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(256), 0, s, (const double*)workspace, (int)nb, 1.0 / (double)total,
                     loss);
  return check_launch("loss_forward_backward");
}

extern "C" int srk_sgd_step(float* p, const float* g, float* momentum_buf, size_t n, float lr, float momentum,
                            float weight_decay, int nesterov, int first_step, const float* lr_dev,
                            const float* grad_scale_dev, void* stream) {
  SRK_REQUIRE(p && g && n > 0, "sgd_step: null pointer or empty");
  SRK_REQUIRE(momentum == 0.f || momentum_buf, "sgd_step: momentum needs a buffer");
  const bool vec = (n & 3) == 0 && (((uintptr_t)p | (uintptr_t)g | (uintptr_t)momentum_buf) & 15) == 0;
  if (vec) {
    size_t b = (n / 4 + 256 * 2 - 1) / (256 * 2);   // two float4 per thread
    if (b > 65535) b = 65535;
    hipLaunchKernelGGL(k_sgd4, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, p, g, momentum_buf, n / 4, lr, momentum,
                       weight_decay, nesterov, first_step, lr_dev, grad_scale_dev);
    return check_launch("sgd_step");
  }
  hipLaunchKernelGGL(k_sgd, dim3(red_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, momentum_buf, n, lr, momentum,
                     weight_decay, nesterov, first_step, lr_dev, grad_scale_dev);
  return check_launch("sgd_step");
}

extern "C" int srk_adam_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, size_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int32_t* step_dev,
                             const float* lr_dev, const float* grad_scale_dev, void* stream) {
  SRK_REQUIRE(p && g && exp_avg && exp_avg_sq && step_dev && n > 0, "adam_step: null pointer or empty");
  hipStream_t s = (hipStream_t)stream;
  if (n % 4 == 0 && (((uintptr_t)p | (uintptr_t)g | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0) {
    size_t nb = (n / 4 + 256 * 2 - 1) / (256 * 2);
    const size_t cap = (size_t)2 * kNumCU;
    if (nb > cap) nb = cap;
    hipLaunchKernelGGL(k_adam4, dim3((unsigned)nb), dim3(256), 0, s, p, g, exp_avg, exp_avg_sq, n / 4, lr, beta1, beta2, eps,
                       weight_decay, step_dev, lr_dev, grad_scale_dev);
    return check_launch("adam_step");
  }
  hipLaunchKernelGGL(k_adam, dim3(red_grid(n)), dim3(256), 0, s, p, g, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps,
                     weight_decay, step_dev, lr_dev, grad_scale_dev);
  return check_launch("adam_step");
}

extern "C" int srk_grad_norm_clip(const float* g, size_t n, float max_norm, float* norm_out, float* scale_out,
                                  void* workspace, void* stream) {
  SRK_REQUIRE(g && workspace && n > 0, "grad_norm_clip: null pointer or empty");
  const unsigned nb = red_grid(n);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_sqsum_partial, dim3(nb), dim3(256), 0, s, g, n, (double*)workspace);
  hipLaunchKernelGGL(k_norm_final, dim3(1), dim3(256), 0, s, (const double*)workspace, (int)nb, max_norm, norm_out,
                     scale_out);
  return check_launch("grad_norm_clip");
}
