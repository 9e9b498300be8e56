// BatchNorm2d (train / eval) and Linear — the SRGAN-only pieces of the hot path
// (base_networks.py:7,46,117,161; srgan.py:49-81).  NHWC: a tensor is [rows][C] with
// rows = N*H*W, so per-channel statistics are column sums with lanes = channels (coalesced rows).
#include "srk_common.h"
#include "bf16_frag.h"
#include <stdlib.h>

namespace srk {

constexpr int kBnRowSplits = 512;

// partial[split][2][C] (double): sum(a), sum(a*b') where the second operand depends on MODE:
//   MODE 0: a = x,  second = x*x                      (forward statistics)
//   MODE 1: a = dy, second = dy * (x-mean)*rstd       (backward statistics)
template <int MODE, bool DBL>
__global__ __launch_bounds__(256) void k_bn_colsum(const float* __restrict__ a, const float* __restrict__ x,
                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                   double* __restrict__ partial, size_t rows, int C,
                                                   size_t rows_per_split) {
  // lane = channel of a 64-channel slab (coalesced 256-byte rows), the 4 waves interleave rows; 8 rows per
  // wave are loaded before the first add (8 independent fp32 accumulator pairs), flushed to double per batch
  __shared__ double sm[2][4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  size_t r0 = (size_t)blockIdx.y * rows_per_split, r1 = r0 + rows_per_split;
  if (r1 > rows) r1 = rows;
  double s0 = 0.0, s1 = 0.0;
  if (c < C) {
    float mu = 0.f, rs = 1.f;
    if (MODE == 1) {
      mu = mean[c];
      rs = rstd[c];
    }
    // 16 rows per wave are loaded before the first add: with 64 rows per block (bn_colsum) a block is ONE load round
    // (these kernels are 5-8 us latency chains on the SRGAN tensors, not bandwidth problems)
    constexpr int U = 16;
    for (size_t rb = r0 + w; rb < r1; rb += 4 * U) {
      float va[U], vx[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {   // (unconditional loads of a clamped row + select below: a load under a branch the
        const size_t r = rb + 4 * (size_t)u;            //  compiler cannot prove uniform is followed by s_waitcnt vmcnt(0),
        const size_t rc = r < r1 ? r : r1 - 1;          //  and the "one load round" became 16 serialised round trips)
        va[u] = a[rc * C + c];
        if (MODE == 1) vx[u] = x[rc * C + c];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool in = rb + 4 * (size_t)u < r1;
        va[u] = in ? va[u] : 0.f;
        if (MODE == 1) vx[u] = in ? vx[u] : mu;
      }
      if (DBL || MODE == 0) {
        // backward statistics feed a difference that cancels to ~1e-4 of its mass when a BatchNorm follows another
        // (SRGAN-D at 128x128): products and sums in double, only the inputs are fp32.  Forward: x*x is exact in
        // double, so mean / var carry no rounding but that of the inputs.
        double d0 = 0.0, d1 = 0.0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          d0 += (double)va[u];
          d1 += MODE == 0 ? (double)va[u] * (double)va[u] : (double)va[u] * (((double)vx[u] - (double)mu) * (double)rs);
        }
        s0 += d0;
        s1 += d1;
      } else {
        float f0 = 0.f, f1 = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          f0 += va[u];
          f1 += va[u] * ((vx[u] - mu) * rs);
        }
        s0 += f0;
        s1 += f1;
      }
    }
  }
  sm[0][w][lane] = s0;
  sm[1][w][lane] = s1;
  __syncthreads();
  if (w == 0 && c < C) {
    double* p = partial + (size_t)blockIdx.y * 2 * C;
    p[c] = sm[0][0][lane] + sm[0][1][lane] + sm[0][2][lane] + sm[0][3][lane];
    p[C + c] = sm[1][0][lane] + sm[1][1][lane] + sm[1][2][lane] + sm[1][3][lane];
  }
}

// stats[i] = sum_k partial[k][i]: 64 columns per block, the 4 waves each take every 4th split with 4 independent
// accumulators, combined through LDS in a fixed order (deterministic)
__global__ __launch_bounds__(256) void k_bn_reduce(const double* __restrict__ partial, double* __restrict__ stats,
                                                   int nsplit, int C2) {
  __shared__ double sm[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (i < C2) {
    int k = w;
    for (; k + 12 < nsplit; k += 16) {
      s0 += partial[(size_t)k * C2 + i];
      s1 += partial[(size_t)(k + 4) * C2 + i];
      s2 += partial[(size_t)(k + 8) * C2 + i];
      s3 += partial[(size_t)(k + 12) * C2 + i];
    }
    for (; k < nsplit; k += 4) s0 += partial[(size_t)k * C2 + i];
  }
  sm[w][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (w == 0 && i < C2) stats[i] = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
}

// k_bn_reduce fused with what consumes the reduced sums (one launch less per BatchNorm call and direction):
//   MODE 0 (forward):  + k_bn_finalize  (mean / rstd / running statistics / num_batches_tracked)
//   MODE 1 (backward): + k_bn_param_grads (dbeta += sum dy, dgamma += sum dy * xhat)
// 64 channels per block, lane = channel; the NW waves take every NW-th split, combined through LDS in a fixed order.
// (NW = 16 for more than 64 splits: with 4 waves the 512 splits of a large activation were 8 dependent L2 round trips
// of ONE block -- the kernel is a latency chain, 8.8 us average on the SRGAN step.)
template <int MODE, int NW = 4>
__global__ __launch_bounds__(64 * NW) void k_bn_reduce_fused(const double* __restrict__ partial, double* __restrict__ stats,
                                                         int nsplit, int C, double count, float* __restrict__ o0,
                                                         float* __restrict__ o1, float* __restrict__ rm,
                                                         float* __restrict__ rv, float momentum, float eps,
                                                         long long* __restrict__ nbt) {
  __shared__ double sm[2][NW][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
  if (c < C) {
    // 16 splits per wave in flight (every load issued before the first add): one L2 round trip per 16 NW splits
    constexpr int U = 16;
    for (int kb = w; kb < nsplit; kb += NW * U) {
      double va[U], vb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = kb + NW * u, kc = k < nsplit ? k : nsplit - 1;   // (clamped unconditional loads + select: see k_bn_colsum)
        va[u] = partial[(size_t)kc * 2 * C + c];
        vb[u] = partial[(size_t)kc * 2 * C + C + c];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (kb + NW * u >= nsplit) va[u] = vb[u] = 0.0;
#pragma unroll
      for (int u = 0; u < U; u += 2) {
        a0 += va[u];
        b0 += vb[u];
        a1 += va[u + 1];
        b1 += vb[u + 1];
      }
    }
  }
  sm[0][w][lane] = a0 + a1;
  sm[1][w][lane] = b0 + b1;
  __syncthreads();
  if (w != 0 || c >= C) return;
  double s0 = 0.0, s1 = 0.0;
#pragma unroll
  for (int q = 0; q < NW; q += 4) {
    s0 += (sm[0][q][lane] + sm[0][q + 1][lane]) + (sm[0][q + 2][lane] + sm[0][q + 3][lane]);
    s1 += (sm[1][q][lane] + sm[1][q + 1][lane]) + (sm[1][q + 2][lane] + sm[1][q + 3][lane]);
  }
  stats[c] = s0;
  stats[C + c] = s1;
  if (MODE == 0) {
    if (c == 0 && nbt) *nbt += 1;
    const double mean = s0 / count;
    double var = s1 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    o0[c] = (float)mean;
    o1[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rm) rm[c] = (1.f - momentum) * rm[c] + momentum * (float)mean;
    if (rv) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      rv[c] = (1.f - momentum) * rv[c] + momentum * (float)unbiased;
    }
  } else {
    if (o1) o1[c] += (float)s0;  // dbeta
    if (o0) o0[c] += (float)s1;  // dgamma
  }
}

// The same reduction with 16 channels per block and 64 split phases of 16 lanes each (1024 threads): a 64-channel BatchNorm
// of the SRGAN generator is FOUR blocks reading 128 KB of partial sums each -- every split of a thread in flight at once for
// up to 512 splits -- instead of ONE block reading 512 KB (k_bn_reduce_fused<., 16>: 7.3 us average, ~110 launches per
// adversarial step; the kernel is the L2 read time of a single CU).  Double sums: the order of the adds is immaterial at
// the fp32 precision of everything downstream.
template <int MODE>
__global__ __launch_bounds__(1024) void k_bn_reduce16(const double* __restrict__ partial, double* __restrict__ stats, int nsplit,
                                                      int C, double count, float* __restrict__ o0, float* __restrict__ o1,
                                                      float* __restrict__ rm, float* __restrict__ rv, float momentum, float eps,
                                                      long long* __restrict__ nbt) {
  __shared__ double sm[2][64][16];
  const int t = threadIdx.x, cl = t & 15, ph = t >> 4;
  const int c = blockIdx.x * 16 + cl;
  const int cc = c < C ? c : C - 1;
  double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
  constexpr int U = 8;
  for (int kb = ph; kb < nsplit; kb += 64 * U) {
    double va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = kb + 64 * u, kc = k < nsplit ? k : nsplit - 1;   // (clamped unconditional loads + select: see k_bn_colsum)
      va[u] = partial[(size_t)kc * 2 * C + cc];
      vb[u] = partial[(size_t)kc * 2 * C + C + cc];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (kb + 64 * u >= nsplit) va[u] = vb[u] = 0.0;
#pragma unroll
    for (int u = 0; u < U; u += 2) {
      a0 += va[u];
      b0 += vb[u];
      a1 += va[u + 1];
      b1 += vb[u + 1];
    }
  }
  sm[0][ph][cl] = a0 + a1;
  sm[1][ph][cl] = b0 + b1;
  __syncthreads();
  double s0 = 0.0, s1 = 0.0;
  if (ph < 8) {   // phases 8 ph .. 8 ph + 7
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      s0 += sm[0][ph * 8 + j][cl];
      s1 += sm[1][ph * 8 + j][cl];
    }
  }
  __syncthreads();
  if (ph < 8) {
    sm[0][ph][cl] = s0;
    sm[1][ph][cl] = s1;
  }
  __syncthreads();
  if (ph != 0 || c >= C) return;
  s0 = s1 = 0.0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    s0 += sm[0][j][cl];
    s1 += sm[1][j][cl];
  }
  stats[c] = s0;
  stats[C + c] = s1;
  if (MODE == 0) {
    if (c == 0 && nbt) *nbt += 1;
    const double mean = s0 / count;
    double var = s1 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    o0[c] = (float)mean;
    o1[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rm) rm[c] = (1.f - momentum) * rm[c] + momentum * (float)mean;
    if (rv) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      rv[c] = (1.f - momentum) * rv[c] + momentum * (float)unbiased;
    }
  } else {
    if (o1) o1[c] += (float)s0;  // dbeta
    if (o0) o0[c] += (float)s1;  // dgamma
  }
}

__global__ __launch_bounds__(256) void k_bn_finalize(const double* __restrict__ stats, double count,
                                                     float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                                     float* __restrict__ rm, float* __restrict__ rv, float momentum,
                                                     float eps, int C, long long* __restrict__ nbt) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c == 0 && nbt) *nbt += 1;  // nn.BatchNorm's num_batches_tracked bookkeeping, without a launch of its own
  if (c >= C) return;
  const double mean = stats[c] / count;
  double var = stats[C + c] / count - mean * mean;
  if (var < 0.0) var = 0.0;
  save_mean[c] = (float)mean;
  save_rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (rm) rm[c] = (1.f - momentum) * rm[c] + momentum * (float)mean;
  if (rv) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    rv[c] = (1.f - momentum) * rv[c] + momentum * (float)unbiased;
  }
}

__global__ __launch_bounds__(256) void k_bn_eval_params(const float* __restrict__ rm, const float* __restrict__ rv,
                                                        float eps, float* __restrict__ mean, float* __restrict__ rstd,
                                                        int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  mean[c] = rm[c];
  rstd[c] = 1.f / sqrtf(rv[c] + eps);
}

__global__ __launch_bounds__(256) void k_bn_apply(const float* __restrict__ x, float* __restrict__ y,
                                                  const float* __restrict__ mean, const float* __restrict__ rstd,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  size_t total, int C, int act, float slope) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    float v = (x[i] - mean[c]) * rstd[c];
    v = v * (gamma ? gamma[c] : 1.f) + (beta ? beta[c] : 0.f);
    y[i] = act_apply(v, act, slope);
  }
}

template <bool DBL>
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float* __restrict__ dy, const float* __restrict__ x,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ gamma,
                                                      const double* __restrict__ dstats, double count,
                                                      float* __restrict__ dx, size_t total, int C) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const float rs = rstd[c];
    const float g = gamma ? gamma[c] : 1.f;
    if (DBL) {
      // dy - mean(dy) - xhat*mean(dy*xhat) cancels heavily when this BN's output feeds another BN: evaluate the
      // difference in double from the fp32 inputs and the double sums, round once
      const double xhat = ((double)x[i] - (double)mean[c]) * (double)rs;
      const double m1 = dstats[c] / count;
      const double m2 = dstats[C + c] / count;
      dx[i] = (float)((double)g * (double)rs * ((double)dy[i] - m1 - xhat * m2));
    } else {
      const float xhat = (x[i] - mean[c]) * rs;
      const float m1 = (float)(dstats[c] / count);
      const float m2 = (float)(dstats[C + c] / count);
      dx[i] = g * rs * (dy[i] - m1 - xhat * m2);
    }
  }
}

// 16-byte versions (C % 4 == 0, aligned tensors): one float4 per thread and pass, no per-element modulo
typedef float bn_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_bn_apply4(const float* __restrict__ x, float* __restrict__ y,
                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   size_t total4, int C, int act, float slope, float* __restrict__ y_amax) {
  __shared__ float sm_amax[4];
  float amax = 0.f;
  const float peeked = amax_peek(y_amax, blockIdx.x);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
    const int c = (int)((i * 4) % (size_t)C);
    const bn_f4 xv = *reinterpret_cast<const bn_f4*>(x + i * 4);
    const bn_f4 mu = *reinterpret_cast<const bn_f4*>(mean + c), rs = *reinterpret_cast<const bn_f4*>(rstd + c);
    bn_f4 v = (xv - mu) * rs;
    if (gamma) v *= *reinterpret_cast<const bn_f4*>(gamma + c);
    if (beta) v += *reinterpret_cast<const bn_f4*>(beta + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = act_apply(v[e], act, slope);
    *reinterpret_cast<bn_f4*>(y + i * 4) = v;
    if (y_amax) amax = abs_max4(amax, v);
  }
  if (y_amax) amax_commit_block(y_amax, amax, blockIdx.x, sm_amax, 4, peeked);   // the running maximum the next conv scales by
}

template <bool DBL>
__global__ __launch_bounds__(256) void k_bn_bwd_apply4(const float* __restrict__ dy, const float* __restrict__ x,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma,
                                                       const double* __restrict__ dstats, double count,
                                                       float* __restrict__ dx, size_t total4, int C) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
    const int c = (int)((i * 4) % (size_t)C);
    const bn_f4 dv = *reinterpret_cast<const bn_f4*>(dy + i * 4), xv = *reinterpret_cast<const bn_f4*>(x + i * 4);
    const bn_f4 mu = *reinterpret_cast<const bn_f4*>(mean + c), rs = *reinterpret_cast<const bn_f4*>(rstd + c);
    bn_f4 g = {1.f, 1.f, 1.f, 1.f};
    if (gamma) g = *reinterpret_cast<const bn_f4*>(gamma + c);
    bn_f4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (DBL) {  // (same evaluation as k_bn_bwd_apply<true>)
        const double xhat = ((double)xv[e] - (double)mu[e]) * (double)rs[e];
        const double m1 = dstats[c + e] / count;
        const double m2 = dstats[C + c + e] / count;
        o[e] = (float)((double)g[e] * (double)rs[e] * ((double)dv[e] - m1 - xhat * m2));
      } else {
        const float xhat = (xv[e] - mu[e]) * rs[e];
        const float m1 = (float)(dstats[c + e] / count);
        const float m2 = (float)(dstats[C + c + e] / count);
        o[e] = g[e] * rs[e] * (dv[e] - m1 - xhat * m2);
      }
    }
    *reinterpret_cast<bn_f4*>(dx + i * 4) = o;
  }
}

// ---- BatchNorm with the activation that follows it (and a residual add) folded in ------------------------------
// The reference's blocks run act(bn(conv(x))) (base_networks.py:58-71) and, in the BN ResnetBlock, bn(conv2(.)) + x
// (base_networks.py:141-150): the activation / add were passes of their own over the tensor (forward + backward, 190
// launches per SRGAN step).  z = gamma * xhat + beta is recomputed from x in the backward kernels, so no activation
// input / output is saved at all.
struct BnAct {
  const float* gamma;
  const float* beta;
  const float* prelu_w;  // SRK_ACT_PRELU: device slope(s)
  int act;               // SRK_ACT_RELU / LRELU / PRELU (backward: NONE = plain)
  int prelu_n;           // 1 or C
  float slope;           // SRK_ACT_LRELU
};

// z = gamma * xhat + beta, evaluated the SAME way (one subtraction, one product, one fused multiply-add) in the forward
// and in both backward kernels, so that they agree bit for bit on the sign of every z
__device__ __forceinline__ float bn_z(float x, float mu, float rs, float g, float b) {
  return __builtin_fmaf((x - mu) * rs, g, b);
}

__device__ __forceinline__ float bn_act_slope(const BnAct& A, int c) {
  if (A.act == SRK_ACT_RELU) return 0.f;
  if (A.act == SRK_ACT_PRELU) return A.prelu_n > 1 ? A.prelu_w[c] : A.prelu_w[0];
  return A.slope;
}

// y = act(gamma * (x - mean) * rstd + beta) [+ residual], one float4 per thread and pass
__global__ __launch_bounds__(256) void k_bn_apply_act4(const float* __restrict__ x, float* __restrict__ y,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       BnAct A, const float* __restrict__ residual, size_t total4, int C,
                                                       float* __restrict__ y_amax) {
  __shared__ float sm_amax[4];
  float amax = 0.f;
  const float peeked = amax_peek(y_amax, blockIdx.x);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
    const int c = (int)((i * 4) % (size_t)C);
    const bn_f4 xv = *reinterpret_cast<const bn_f4*>(x + i * 4);
    const bn_f4 mu = *reinterpret_cast<const bn_f4*>(mean + c), rs = *reinterpret_cast<const bn_f4*>(rstd + c);
    bn_f4 g = {1.f, 1.f, 1.f, 1.f}, b = {0.f, 0.f, 0.f, 0.f}, v;
    if (A.gamma) g = *reinterpret_cast<const bn_f4*>(A.gamma + c);
    if (A.beta) b = *reinterpret_cast<const bn_f4*>(A.beta + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float z = bn_z(xv[e], mu[e], rs[e], g[e], b[e]);
      v[e] = (A.act == SRK_ACT_NONE || z > 0.f) ? z : bn_act_slope(A, c + e) * z;
    }
    if (residual) v += *reinterpret_cast<const bn_f4*>(residual + i * 4);
    *reinterpret_cast<bn_f4*>(y + i * 4) = v;
    if (y_amax) amax = abs_max4(amax, v);
  }
  if (y_amax) amax_commit_block(y_amax, amax, blockIdx.x, sm_amax, 4, peeked);   // the running maximum the next conv scales by
}

// backward statistics with dz = dy * act'(z):  partial[split][3][C] = sum dz, sum dz * xhat, sum_{z <= 0} dy * z (PReLU)
__global__ __launch_bounds__(256) void k_bn_colsum_act(const float* __restrict__ dy, const float* __restrict__ x,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       BnAct A, double* __restrict__ partial, size_t rows, int C,
                                                       size_t rows_per_split) {
  __shared__ double sm[3][4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  size_t r0 = (size_t)blockIdx.y * rows_per_split, r1 = r0 + rows_per_split;
  if (r1 > rows) r1 = rows;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  if (c < C) {
    const float mu = mean[c], rs = rstd[c];
    const float g = A.gamma ? A.gamma[c] : 1.f, b = A.beta ? A.beta[c] : 0.f;
    const float a = bn_act_slope(A, c);
    constexpr int U = 32;
    for (size_t rb = r0 + w; rb < r1; rb += 4 * U) {
      float va[U], vx[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t r = rb + 4 * (size_t)u, rc = r < r1 ? r : r1 - 1;   // (clamped unconditional loads + select)
        va[u] = dy[rc * C + c];
        vx[u] = x[rc * C + c];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (rb + 4 * (size_t)u >= r1) {
          va[u] = 0.f;
          vx[u] = mu;
        }
      double d0 = 0.0, d1 = 0.0, d2 = 0.0;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        // z in fp32 exactly as the forward kernel evaluated it (same operations, same order): the mask must be the
        // forward's mask
        const float z = bn_z(vx[u], mu, rs, g, b);
        const double xhat = ((double)vx[u] - (double)mu) * (double)rs;
        const double dz = z > 0.f ? (double)va[u] : (double)va[u] * (double)a;
        d0 += dz;
        d1 += dz * xhat;
        if (A.act == SRK_ACT_PRELU) d2 += z > 0.f ? 0.0 : (double)va[u] * (double)z;
      }
      s0 += d0;
      s1 += d1;
      s2 += d2;
    }
  }
  sm[0][w][lane] = s0;
  sm[1][w][lane] = s1;
  sm[2][w][lane] = s2;
  __syncthreads();
  if (w == 0 && c < C) {
    double* p = partial + (size_t)blockIdx.y * 3 * C;
#pragma unroll
    for (int q = 0; q < 3; ++q) p[q * C + c] = sm[q][0][lane] + sm[q][1][lane] + sm[q][2][lane] + sm[q][3][lane];
  }
}

// reduce of the above + dbeta += sum dz, dgamma += sum dz * xhat, dprelu += sum_{z<=0} dy * z; stats[2C] as usual
template <int NW>   // waves per block (16 for many splits: see k_bn_reduce_fused)
__global__ __launch_bounds__(64 * NW) void k_bn_reduce_act(const double* __restrict__ partial, double* __restrict__ stats,
                                                          int nsplit, int C, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta, float* __restrict__ dprelu, int prelu_n) {
  __shared__ double sm[3][NW][64];
  __shared__ float psum[64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  double a[3] = {0.0, 0.0, 0.0};
  if (c < C) {
    constexpr int U = 8;
    for (int kb = w; kb < nsplit; kb += NW * U) {
      double v[U][3];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = kb + NW * u;
#pragma unroll
        for (int q = 0; q < 3; ++q) v[u][q] = partial[(size_t)(k < nsplit ? k : nsplit - 1) * 3 * C + q * C + c];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (kb + NW * u >= nsplit) v[u][0] = v[u][1] = v[u][2] = 0.0;
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int q = 0; q < 3; ++q) a[q] += v[u][q];
    }
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) sm[q][w][lane] = a[q];
  __syncthreads();
  if (w != 0) return;
  double s[3] = {0.0, 0.0, 0.0};
  if (c < C) {
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int g = 0; g < NW; g += 4) s[q] += (sm[q][g][lane] + sm[q][g + 1][lane]) + (sm[q][g + 2][lane] + sm[q][g + 3][lane]);
    stats[c] = s[0];
    stats[C + c] = s[1];
    if (dbeta) dbeta[c] += (float)s[0];
    if (dgamma) dgamma[c] += (float)s[1];
    if (dprelu && prelu_n > 1) dprelu[c] += (float)s[2];
  }
  if (dprelu && prelu_n == 1) {  // one slope: the block's 64 channels summed in lane order, one atomic per block
    psum[lane] = c < C ? (float)s[2] : 0.f;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      double t = 0.0;
      for (int q = 0; q < 64; ++q) t += (double)psum[q];
      atomicAdd(dprelu, (float)t);
    }
  }
}

template <bool DBL>
__global__ __launch_bounds__(256) void k_bn_bwd_apply_act4(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           BnAct A, const double* __restrict__ dstats, double count,
                                                           float* __restrict__ dx, size_t total4, int C) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
    const int c = (int)((i * 4) % (size_t)C);
    const bn_f4 dv = *reinterpret_cast<const bn_f4*>(dy + i * 4), xv = *reinterpret_cast<const bn_f4*>(x + i * 4);
    const bn_f4 mu = *reinterpret_cast<const bn_f4*>(mean + c), rs = *reinterpret_cast<const bn_f4*>(rstd + c);
    bn_f4 g = {1.f, 1.f, 1.f, 1.f}, b = {0.f, 0.f, 0.f, 0.f};
    if (A.gamma) g = *reinterpret_cast<const bn_f4*>(A.gamma + c);
    if (A.beta) b = *reinterpret_cast<const bn_f4*>(A.beta + c);
    bn_f4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float z = bn_z(xv[e], mu[e], rs[e], g[e], b[e]);
      const float dz = (A.act == SRK_ACT_NONE || z > 0.f) ? dv[e] : dv[e] * bn_act_slope(A, c + e);
      if (DBL) {
        const double xhat = ((double)xv[e] - (double)mu[e]) * (double)rs[e];
        const double m1 = dstats[c + e] / count;
        const double m2 = dstats[C + c + e] / count;
        o[e] = (float)((double)g[e] * (double)rs[e] * ((double)dz - m1 - xhat * m2));
      } else {
        const float xhat = (xv[e] - mu[e]) * rs[e];
        const float m1 = (float)(dstats[c + e] / count);
        const float m2 = (float)(dstats[C + c + e] / count);
        o[e] = g[e] * rs[e] * (dz - m1 - xhat * m2);
      }
    }
    *reinterpret_cast<bn_f4*>(dx + i * 4) = o;
  }
}

static bool bn_vec4(int C, const void* a, const void* b, const void* c, const void* d, const void* e, const void* f) {
  if (C & 3) return false;
  return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d | (uintptr_t)e | (uintptr_t)f) & 15) == 0;
}

__global__ __launch_bounds__(256) void k_bn_param_grads(const double* __restrict__ dstats, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  if (dbeta) dbeta[c] += (float)dstats[c];
  if (dgamma) dgamma[c] += (float)dstats[C + c];
}

// SRK_BN_F32=1 (debugging / A-B only): the backward arithmetic of round 1 (fp32 per-element terms)
static bool bn_fp32_backward() {
  return env_int("SRK_BN_F32", 0) == 1;
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the split reduction FINISHED INSIDE the apply kernel ("finalize-in-apply").
// A BatchNorm call was column sums -> reduce -> apply (forward) and column sums -> reduce -> apply (backward): six launches,
// each at the ~5 us per-node floor of a replayed hipGraph on the 4 MB tensors of the SRGAN step (458 of its 846 nodes).  An
// in-kernel hand-off does not beat that floor -- but the reduce needs no hand-off at all if every block of the
// apply kernel sums the partials of ITS OWN channels again: blocks own a slab of 16 channels x a range of rows, start with
// k_bn_reduce16's summation (64 phases x 16 channels, the same order: forward statistics bit-equal to the two-launch path)
// over the slab's [nsplit][16] partials -- 32 - 96 KB from L2 per block, no cross-block dependency -- and go straight on to
// their rows.  The first row-range block of a slab also writes what the reduce kernel wrote (mean / rstd / running
// statistics / the [2C] sums; dgamma / dbeta / dprelu).  One launch less per BatchNorm and direction.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int BNF_CS = 16;      // channels per slab
constexpr int BNF_THR = 1024;   // 64 phases x 16 channels in the reduction, 256 rows x 4 float4 in the apply loop

// totals of NQ sums for channel c (the thread's t & 15) over partial[k * kstride + q * qstride + c], k < nsplit; every thread
// of the block returns the totals of ITS channel.  Order of k_bn_reduce16.
template <int NQ>
__device__ __forceinline__ void bn_slab_totals(const double* __restrict__ partial, int nsplit, size_t kstride, size_t qstride,
                                               int c, double (*sm)[64][BNF_CS], double (&tot)[NQ]) {
  const int t = threadIdx.x, cl = t & 15, ph = t >> 4;
  double a0[NQ], a1[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) a0[q] = a1[q] = 0.0;
  constexpr int U = 8;
  for (int kb = ph; kb < nsplit; kb += 64 * U) {
    double v[U][NQ];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = kb + 64 * u, kc = k < nsplit ? k : nsplit - 1;   // (clamped unconditional loads + select: see k_bn_colsum)
#pragma unroll
      for (int q = 0; q < NQ; ++q) v[u][q] = partial[(size_t)kc * kstride + q * qstride + c];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (kb + 64 * u >= nsplit) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) v[u][q] = 0.0;
      }
#pragma unroll
    for (int u = 0; u < U; u += 2)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        a0[q] += v[u][q];
        a1[q] += v[u + 1][q];
      }
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) sm[q][ph][cl] = a0[q] + a1[q];
  __syncthreads();
  double s[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) s[q] = 0.0;
  if (ph < 8) {   // phases 8 ph .. 8 ph + 7
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int q = 0; q < NQ; ++q) s[q] += sm[q][ph * 8 + j][cl];
  }
  __syncthreads();
  if (ph < 8) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) sm[q][ph][cl] = s[q];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    double r = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) r += sm[q][j][cl];
    tot[q] = r;
  }
}

// block -> (slab, row range): the slabs of ONE row range sit on consecutive blocks of one XCD (hardware deals block ids
// round-robin over the 8 XCDs): a 128-byte line of the tensor holds two slabs' 64 bytes, and with slab = blockIdx.x the two
// readers of a line were on different XCDs -- every line fetched into two L2s
__device__ __forceinline__ void bnf_block(int slabs, int& slab, int& range) {
  const int nb = gridDim.x, per = nb >> 3, rem = nb & 7;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int b = xcd < rem ? xcd * (per + 1) + idx : rem * (per + 1) + (xcd - rem) * per + idx;
  range = b / slabs;
  slab = b - range * slabs;
}

// forward: statistics from [nsplit][2][C] partials + y = act(gamma * (x - mean) * rstd + beta) [+ residual]
__global__ __launch_bounds__(BNF_THR) void k_bn_fin_apply_act(const double* __restrict__ partial, int nsplit, double count,
                                                              const float* __restrict__ x, float* __restrict__ y, BnAct A,
                                                              const float* __restrict__ residual, size_t rows, int C,
                                                              size_t rows_per_block, float* __restrict__ save_mean,
                                                              float* __restrict__ save_rstd, double* __restrict__ stats,
                                                              float* __restrict__ rm, float* __restrict__ rv, float momentum,
                                                              float eps, long long* __restrict__ nbt,
                                                              float* __restrict__ y_amax) {
  __shared__ double sm[2][64][BNF_CS];
  __shared__ float prm[5][BNF_CS];   // mean, rstd, gamma, beta, negative-side slope of the slab's channels
  __shared__ float sm_amax[BNF_THR / 64];
  const int t = threadIdx.x, cl = t & 15, ph = t >> 4;
  int bslab, brange;
  bnf_block(C / BNF_CS, bslab, brange);
  const int c0 = bslab * BNF_CS, c = c0 + cl;
  // the thread's first row is requested BEFORE the reduction (it does not depend on the statistics): its latency hides
  // behind the partial loads and the three barriers of the prologue -- most blocks have exactly one row per thread
  const int q4 = t & 3, rr = t >> 2;
  const size_t r_begin = (size_t)brange * rows_per_block;
  size_t r_end = r_begin + rows_per_block;
  if (r_end > rows) r_end = rows;
  const size_t r_first = r_begin + rr < r_end ? r_begin + rr : (r_end - 1);   // (clamped: unconditional loads)
  const size_t off_first = r_first * (size_t)C + c0 + q4 * 4;
  bn_f4 x_first = *reinterpret_cast<const bn_f4*>(x + off_first);
  bn_f4 res_first = {0.f, 0.f, 0.f, 0.f};
  if (residual) res_first = *reinterpret_cast<const bn_f4*>(residual + off_first);
  double tot[2];
  bn_slab_totals<2>(partial, nsplit, (size_t)2 * C, (size_t)C, c, sm, tot);
  if (ph == 0) {   // (the arithmetic of k_bn_reduce16<0>)
    const double mean = tot[0] / count;
    double var = tot[1] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float mf = (float)mean, rf = (float)(1.0 / sqrt(var + (double)eps));
    prm[0][cl] = mf;
    prm[1][cl] = rf;
    prm[2][cl] = A.gamma ? A.gamma[c] : 1.f;
    prm[3][cl] = A.beta ? A.beta[c] : 0.f;
    prm[4][cl] = A.act == SRK_ACT_NONE ? 1.f : bn_act_slope(A, c);
    if (brange == 0) {
      stats[c] = tot[0];
      stats[C + c] = tot[1];
      if (c == 0 && nbt) *nbt += 1;
      save_mean[c] = mf;
      save_rstd[c] = rf;
      if (rm) rm[c] = (1.f - momentum) * rm[c] + momentum * mf;
      if (rv) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        rv[c] = (1.f - momentum) * rv[c] + momentum * (float)unbiased;
      }
    }
  }
  __syncthreads();
  bn_f4 mu, rs, g, b, sl;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    mu[e] = prm[0][q4 * 4 + e];
    rs[e] = prm[1][q4 * 4 + e];
    g[e] = prm[2][q4 * 4 + e];
    b[e] = prm[3][q4 * 4 + e];
    sl[e] = prm[4][q4 * 4 + e];
  }
  float amax = 0.f;
  const float peeked = amax_peek(y_amax, blockIdx.x);
  bn_f4 xv = x_first, rv4 = res_first;
  for (size_t r = r_begin + rr; r < r_end; r += BNF_THR / 4) {
    const size_t off = r * (size_t)C + c0 + q4 * 4;
    bn_f4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float z = bn_z(xv[e], mu[e], rs[e], g[e], b[e]);
      v[e] = z > 0.f ? z : sl[e] * z;   // (no activation: slope 1)
    }
    if (residual) v += rv4;
    *reinterpret_cast<bn_f4*>(y + off) = v;
    if (y_amax) amax = abs_max4(amax, v);
    const size_t rn = r + BNF_THR / 4;
    if (rn < r_end) {
      const size_t offn = rn * (size_t)C + c0 + q4 * 4;
      xv = *reinterpret_cast<const bn_f4*>(x + offn);
      if (residual) rv4 = *reinterpret_cast<const bn_f4*>(residual + offn);
    }
  }
  if (y_amax) amax_commit_block(y_amax, amax, blockIdx.x, sm_amax, BNF_THR / 64, peeked);
}

// backward: (sum dz, sum dz * xhat[, sum_{z<=0} dy * z]) from [nsplit][NQ][C] partials, parameter gradients, and
// dx = gamma * rstd * (dz - m1 - xhat * m2) with dz = dy * act'(z), z recomputed from x as in the forward
// NQ sums are reduced per channel; `pq` = planes per split row of `partial` (3 behind k_bn_colsum_act even where the third --
// PReLU's slope gradient -- is not wanted: NQ = 2 then reads two of the three)
template <int NQ, bool DBL>
__global__ __launch_bounds__(BNF_THR) void k_bn_fin_bwd_apply_act(const double* __restrict__ partial, int nsplit, int pq, double count,
                                                                  const float* __restrict__ dy, const float* __restrict__ x,
                                                                  const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                  BnAct A, float* __restrict__ dx, size_t rows, int C,
                                                                  size_t rows_per_block, double* __restrict__ dstats,
                                                                  float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                  float* __restrict__ dprelu) {
  __shared__ double sm[NQ][64][BNF_CS];
  __shared__ double mm[2][BNF_CS];
  __shared__ float prm[5][BNF_CS];
  const int t = threadIdx.x, cl = t & 15, ph = t >> 4;
  int bslab, brange;
  bnf_block(C / BNF_CS, bslab, brange);
  const int c0 = bslab * BNF_CS, c = c0 + cl;
  // (first row requested before the reduction: see k_bn_fin_apply_act)
  const int q4 = t & 3, rr = t >> 2;
  const size_t r_begin = (size_t)brange * rows_per_block;
  size_t r_end = r_begin + rows_per_block;
  if (r_end > rows) r_end = rows;
  const size_t r_first = r_begin + rr < r_end ? r_begin + rr : (r_end - 1);
  const size_t off_first = r_first * (size_t)C + c0 + q4 * 4;
  bn_f4 dv = *reinterpret_cast<const bn_f4*>(dy + off_first), xv = *reinterpret_cast<const bn_f4*>(x + off_first);
  double tot[NQ];
  bn_slab_totals<NQ>(partial, nsplit, (size_t)pq * C, (size_t)C, c, sm, tot);
  if (ph == 0) {
    mm[0][cl] = tot[0] / count;
    mm[1][cl] = tot[1] / count;
    prm[0][cl] = mean[c];
    prm[1][cl] = rstd[c];
    prm[2][cl] = A.gamma ? A.gamma[c] : 1.f;
    prm[3][cl] = A.beta ? A.beta[c] : 0.f;
    prm[4][cl] = A.act == SRK_ACT_NONE ? 1.f : bn_act_slope(A, c);
    if (brange == 0) {   // (what k_bn_reduce_act / k_bn_reduce16<1> wrote)
      dstats[c] = tot[0];
      dstats[C + c] = tot[1];
      if (dbeta) dbeta[c] += (float)tot[0];
      if (dgamma) dgamma[c] += (float)tot[1];
      if (NQ == 3 && dprelu && A.prelu_n > 1) dprelu[c] += (float)tot[NQ - 1];
    }
  }
  __syncthreads();
  bn_f4 mu, rs, g, b, sl;
  double m1[4], m2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    mu[e] = prm[0][q4 * 4 + e];
    rs[e] = prm[1][q4 * 4 + e];
    g[e] = prm[2][q4 * 4 + e];
    b[e] = prm[3][q4 * 4 + e];
    sl[e] = prm[4][q4 * 4 + e];
    m1[e] = mm[0][q4 * 4 + e];
    m2[e] = mm[1][q4 * 4 + e];
  }
  const bool has_act = A.act != SRK_ACT_NONE;
  for (size_t r = r_begin + rr; r < r_end; r += BNF_THR / 4) {
    const size_t off = r * (size_t)C + c0 + q4 * 4;
    bn_f4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float dz = dv[e];
      if (has_act) {
        const float z = bn_z(xv[e], mu[e], rs[e], g[e], b[e]);
        dz = z > 0.f ? dv[e] : dv[e] * sl[e];
      }
      if (DBL) {
        const double xhat = ((double)xv[e] - (double)mu[e]) * (double)rs[e];
        o[e] = (float)((double)g[e] * (double)rs[e] * ((double)dz - m1[e] - xhat * m2[e]));
      } else {
        const float xhat = (xv[e] - mu[e]) * rs[e];
        o[e] = g[e] * rs[e] * (dz - (float)m1[e] - xhat * (float)m2[e]);
      }
    }
    *reinterpret_cast<bn_f4*>(dx + off) = o;
    const size_t rn = r + BNF_THR / 4;
    if (rn < r_end) {
      const size_t offn = rn * (size_t)C + c0 + q4 * 4;
      dv = *reinterpret_cast<const bn_f4*>(dy + offn);
      xv = *reinterpret_cast<const bn_f4*>(x + offn);
    }
  }
  // one PReLU slope for all channels: block (0, 0) sums the whole third plane (sum_{z<=0} dy * z: [nsplit][C] doubles) in a
  // fixed order -- a strided pass per thread, then a sequential tree through LDS -- and adds it once.  (The first version
  // re-ran the slab reduction for every slab here: C / 16 x four barriers behind the block's own rows, 17 us per launch.)
  if (NQ == 3 && dprelu && A.prelu_n == 1 && bslab == 0 && brange == 0) {
    double acc = 0.0;
    const int total = nsplit * C;
    for (int idx = t; idx < total; idx += BNF_THR) {
      const int k = idx / C, cc = idx - k * C;
      acc += partial[(size_t)k * pq * C + (size_t)2 * C + cc];
    }
    __syncthreads();
    double* red = &sm[0][0][0];   // 1024 doubles of the 3072
    red[t] = acc;
    __syncthreads();
    for (int w = BNF_THR / 2; w >= 64; w >>= 1) {
      if (t < w) red[t] += red[t + w];
      __syncthreads();
    }
    if (t == 0) {
      double r = 0.0;
      for (int q = 0; q < 64; ++q) r += red[q];
      *dprelu += (float)r;
    }
  }
}

static void bnf_grid(size_t rows, int C, dim3& grid, size_t& rows_per_block) {
  const int slabs = C / BNF_CS;
  size_t g = (rows + BNF_THR / 4 - 1) / (BNF_THR / 4);
  size_t cap = (size_t)(2 * kNumCU / slabs);
  if (cap < 1) cap = 1;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  size_t rpb = (rows + g - 1) / g;
  rpb = (rpb + BNF_THR / 4 - 1) / (BNF_THR / 4) * (BNF_THR / 4);
  g = (rows + rpb - 1) / rpb;
  rows_per_block = rpb;
  grid = dim3((unsigned)(slabs * g));
}

struct BnFused {   // fused tail of bn_colsum: what the reduce kernel also computes
  int mode = -1;   // -1: plain reduce
  double count = 0.0;
  float *o0 = nullptr, *o1 = nullptr, *rm = nullptr, *rv = nullptr;
  float momentum = 0.f, eps = 0.f;
  long long* nbt = nullptr;
};

static int bn_colsum(int mode, const float* a, const float* x, const float* mean, const float* rstd, double* out,
                     size_t rows, int C, void* ws, hipStream_t s, const BnFused& fu = BnFused()) {
  int splits = (int)((rows + 63) / 64);
  if (splits > kBnRowSplits) splits = kBnRowSplits;
  if (splits < 1) splits = 1;
  const size_t rps = (rows + splits - 1) / splits;
  dim3 grid(cdiv(C, 64), splits);
  if (mode == 0)
    hipLaunchKernelGGL((k_bn_colsum<0, false>), grid, dim3(256), 0, s, a, x, mean, rstd, (double*)ws, rows, C, rps);
  else if (bn_fp32_backward())
    hipLaunchKernelGGL((k_bn_colsum<1, false>), grid, dim3(256), 0, s, a, x, mean, rstd, (double*)ws, rows, C, rps);
  else
    hipLaunchKernelGGL((k_bn_colsum<1, true>), grid, dim3(256), 0, s, a, x, mean, rstd, (double*)ws, rows, C, rps);
  const bool wide = splits > 64;
  const bool r16 = wide && env_int("SRK_BN_RED16", 1) != 0;
  if (fu.mode == 0 && r16)
    hipLaunchKernelGGL((k_bn_reduce16<0>), dim3(cdiv(C, 16)), dim3(1024), 0, s, (const double*)ws, out, splits, C, fu.count,
                       fu.o0, fu.o1, fu.rm, fu.rv, fu.momentum, fu.eps, fu.nbt);
  else if (fu.mode == 1 && r16)
    hipLaunchKernelGGL((k_bn_reduce16<1>), dim3(cdiv(C, 16)), dim3(1024), 0, s, (const double*)ws, out, splits, C, 0.0, fu.o0,
                       fu.o1, nullptr, nullptr, 0.f, 0.f, nullptr);
  else if (fu.mode == 0 && wide)
    hipLaunchKernelGGL((k_bn_reduce_fused<0, 16>), dim3(cdiv(C, 64)), dim3(1024), 0, s, (const double*)ws, out, splits, C,
                       fu.count, fu.o0, fu.o1, fu.rm, fu.rv, fu.momentum, fu.eps, fu.nbt);
  else if (fu.mode == 0)
    hipLaunchKernelGGL((k_bn_reduce_fused<0, 4>), dim3(cdiv(C, 64)), dim3(256), 0, s, (const double*)ws, out, splits, C,
                       fu.count, fu.o0, fu.o1, fu.rm, fu.rv, fu.momentum, fu.eps, fu.nbt);
  else if (fu.mode == 1 && wide)
    hipLaunchKernelGGL((k_bn_reduce_fused<1, 16>), dim3(cdiv(C, 64)), dim3(1024), 0, s, (const double*)ws, out, splits, C, 0.0,
                       fu.o0, fu.o1, nullptr, nullptr, 0.f, 0.f, nullptr);
  else if (fu.mode == 1)
    hipLaunchKernelGGL((k_bn_reduce_fused<1, 4>), dim3(cdiv(C, 64)), dim3(256), 0, s, (const double*)ws, out, splits, C, 0.0,
                       fu.o0, fu.o1, nullptr, nullptr, 0.f, 0.f, nullptr);
  else
    hipLaunchKernelGGL(k_bn_reduce, dim3(cdiv(2 * C, 64)), dim3(256), 0, s, (const double*)ws, out, splits, 2 * C);
  return check_launch("bn_colsum");
}

// ---------------------------------------------------------------------------------------------
// Linear.  y[B,Out] = act(x[B,In] @ w[Out,In]^T + b).  Weight-streaming (B is small): a block
// owns 4 output neurons and up to 8 batch rows; lanes stride the In axis with float4 loads.
// ---------------------------------------------------------------------------------------------
constexpr int LIN_BT = 8;

__global__ __launch_bounds__(256) void k_linear_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ b, float* __restrict__ y, int B, int In,
                                                    int Out, int act, float slope) {
  __shared__ float sm[4];
  const int o0 = blockIdx.x * 4;
  const int b0 = blockIdx.y * LIN_BT;
  float acc[4][LIN_BT];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < LIN_BT; ++r) acc[k][r] = 0.f;
  for (int i = threadIdx.x; i < In; i += 256) {
    float wv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) wv[k] = (o0 + k < Out) ? w[(size_t)(o0 + k) * In + i] : 0.f;
#pragma unroll
    for (int r = 0; r < LIN_BT; ++r) {
      const float xv = (b0 + r < B) ? x[(size_t)(b0 + r) * In + i] : 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k][r] = fmaf(wv[k], xv, acc[k][r]);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int r = 0; r < LIN_BT; ++r) {
      const float t = block_sum_256(acc[k][r], sm);
      if (threadIdx.x == 0 && o0 + k < Out && b0 + r < B) {
        float v = t + (b ? b[o0 + k] : 0.f);
        y[(size_t)(b0 + r) * Out + o0 + k] = act_apply(v, act, slope);
      }
    }
  }
}

// dx[b][i] = sum_o dy[b][o] * w[o][i]   (thread per i, up to 8 batch rows per pass; 8 weight rows in flight)
__global__ __launch_bounds__(256) void k_linear_dx(const float* __restrict__ dy, const float* __restrict__ w,
                                                   float* __restrict__ dx, int B, int In, int Out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int b0 = blockIdx.y * LIN_BT;
  if (i >= In) return;
  float acc[LIN_BT];
#pragma unroll
  for (int r = 0; r < LIN_BT; ++r) acc[r] = 0.f;
  constexpr int U = 8;
  int o = 0;
  for (; o + U <= Out; o += U) {
    float wv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) wv[u] = w[(size_t)(o + u) * In + i];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int r = 0; r < LIN_BT; ++r) {
        const float g = (b0 + r < B) ? dy[(size_t)(b0 + r) * Out + o + u] : 0.f;
        acc[r] = fmaf(g, wv[u], acc[r]);
      }
  }
  for (; o < Out; ++o) {
    const float wv = w[(size_t)o * In + i];
#pragma unroll
    for (int r = 0; r < LIN_BT; ++r) {
      const float g = (b0 + r < B) ? dy[(size_t)(b0 + r) * Out + o] : 0.f;
      acc[r] = fmaf(g, wv, acc[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < LIN_BT; ++r)
    if (b0 + r < B) dx[(size_t)(b0 + r) * In + i] = acc[r];
}

// dw[o][i] = beta*dw + sum_b dy[b][o]*x[b][i]: a streaming write of Out*In floats; a thread owns 4 consecutive i
// of one o (16-byte loads/stores when In % 4 == 0), batch loop unrolled by 4
__global__ __launch_bounds__(256) void k_linear_dw(const float* __restrict__ dy, const float* __restrict__ x,
                                                   float* __restrict__ dw, int B, int In, int Out, float beta) {
  typedef float lf4 __attribute__((ext_vector_type(4)));
  const int in4 = (In + 3) >> 2;
  const size_t total = (size_t)Out * in4;
  const bool vec = (In & 3) == 0;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int i = (int)(e % in4) * 4;
    const int o = (int)(e / in4);
    lf4 acc = {0.f, 0.f, 0.f, 0.f};
    if (vec) {
      int b = 0;
      for (; b + 4 <= B; b += 4) {
        lf4 xv[4];
        float g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          xv[u] = *reinterpret_cast<const lf4*>(x + (size_t)(b + u) * In + i);
          g[u] = dy[(size_t)(b + u) * Out + o];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += g[u] * xv[u];
      }
      for (; b < B; ++b) acc += dy[(size_t)b * Out + o] * *reinterpret_cast<const lf4*>(x + (size_t)b * In + i);
      lf4* dst = reinterpret_cast<lf4*>(dw + (size_t)o * In + i);
      *dst = beta != 0.f ? beta * *dst + acc : acc;
    } else {
      for (int k = 0; k < 4 && i + k < In; ++k) {
        float a1 = 0.f;
        for (int b = 0; b < B; ++b) a1 = fmaf(dy[(size_t)b * Out + o], x[(size_t)b * In + i + k], a1);
        float* dst = dw + (size_t)o * In + i + k;
        *dst = beta != 0.f ? beta * *dst + a1 : a1;
      }
    }
  }
}

// ---- wide layers (the SRGAN discriminator's 18432 -> 1024, srgan.py:66-70; 3 forward + 3 backward passes per
// adversarial step).  A pass is bound by the 75 MB of weights: the kernels above re-read them once per 8 batch rows and
// re-read x (or dy) from L2 once per weight element (forward 110 us, dx 183 us, dw 142 us at batch 16).  These stream
// every weight ONCE with 16-byte loads and keep the batch side in registers / LDS.
typedef float lf4 __attribute__((ext_vector_type(4)));
constexpr int LW_BT = 16;  // batch rows per pass

// forward: block = 4 output rows x 16 batch rows, its 8 waves split the In axis; wave sums, then 8 partials per value
__global__ __launch_bounds__(512) void k_linear_fwd_wide(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ y, int B,
                                                         int In, int Out, int act, float slope) {
  __shared__ float part[8][4 * LW_BT];
  const int o0 = blockIdx.x * 4, b0 = blockIdx.y * LW_BT;
  const int tid = threadIdx.x;
  float acc[4][LW_BT];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < LW_BT; ++r) acc[k][r] = 0.f;
  const float* wr[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) wr[k] = w + (size_t)(o0 + k < Out ? o0 + k : Out - 1) * In;
  // The blocks walk their rows in ROTATED order (block b starts at step b mod nsteps): rows are In * 4 bytes apart -- a
  // multiple of 8 KB for the 18432-wide layer -- so with every block at the same column the 1024 row streams sat on
  // the same few HBM channels at any moment (0.8 TB/s for a plain weight stream).
  const int nsteps = (In + 512 * 4 - 1) / (512 * 4);
  int step = blockIdx.x % nsteps;
  // (round 4: the weights of step it + 1 are requested before step it multiplies -- one block per CU with 64 bytes per
  //  thread in flight and the load latency of every step exposed streamed the 134 MB weight of SRGAN's 32768 -> 1024
  //  layer at 1.8 TB/s.  Unconditional loads from a clamped column + a select: a load under a branch is followed by
  //  s_waitcnt vmcnt(0).)
  lf4 wn[4];
  auto wload = [&](int st) {
    const int i = st * 512 * 4 + tid * 4;
    const int ic = i < In ? i : 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) wn[k] = *reinterpret_cast<const lf4*>(wr[k] + ic);
  };
  wload(step);
  for (int it = 0; it < nsteps; ++it) {
    const int i = step * 512 * 4 + tid * 4;
    const bool live = i < In;
    lf4 wv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) wv[k] = live ? wn[k] : (lf4){0.f, 0.f, 0.f, 0.f};
    step = step + 1 == nsteps ? 0 : step + 1;
    if (it + 1 < nsteps) wload(step);
    const int ix = live ? i : 0;
#pragma unroll
    for (int r = 0; r < LW_BT; ++r) {
      const int br = b0 + r < B ? b0 + r : B - 1;
      const lf4 xv = *reinterpret_cast<const lf4*>(x + (size_t)br * In + ix);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        acc[k][r] = fmaf(wv[k][0], xv[0], fmaf(wv[k][1], xv[1], fmaf(wv[k][2], xv[2], fmaf(wv[k][3], xv[3], acc[k][r]))));
    }
  }
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < LW_BT; ++r) {
      const float t = wave_sum(acc[k][r]);
      if (lane == 0) part[wave][k * LW_BT + r] = t;
    }
  __syncthreads();
  if (tid < 4 * LW_BT) {
    const int k = tid / LW_BT, r = tid - k * LW_BT;
    if (o0 + k < Out && b0 + r < B) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) v += part[q][tid];
      v += b ? b[o0 + k] : 0.f;
      y[(size_t)(b0 + r) * Out + o0 + k] = act_apply(v, act, slope);
    }
  }
}

// dx: block = 64 inputs i x 16 batch rows; its 16 waves split the Out axis (64 rows each per 1024-row pass); dy of the
// pass sits transposed in LDS ([o][16 b]: four broadcast 16-byte reads per weight), the partials meet in LDS
__global__ __launch_bounds__(1024) void k_linear_dx_wide(const float* __restrict__ dy, const float* __restrict__ w,
                                                         float* __restrict__ dx, int B, int In, int Out) {
  __shared__ __attribute__((aligned(16))) float sm[1024 * LW_BT];  // 64 KB: dyT of a pass, then the partials
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = blockIdx.x * 64 + lane, b0 = blockIdx.y * LW_BT;
  const bool iok = i < In;
  float acc[LW_BT];
#pragma unroll
  for (int r = 0; r < LW_BT; ++r) acc[r] = 0.f;
  for (int ob = 0; ob < Out; ob += 1024) {
    if (ob) __syncthreads();
    {
      const int o = ob + tid;
#pragma unroll
      for (int r = 0; r < LW_BT; ++r)
        sm[tid * LW_BT + r] = (o < Out && b0 + r < B) ? dy[(size_t)(b0 + r) * Out + o] : 0.f;
    }
    __syncthreads();
    const int oend = Out - ob < 1024 ? Out - ob : 1024;
    constexpr int U = 8;
    for (int ol = wave * 64; ol < wave * 64 + 64 && ol < oend; ol += U) {
      float wv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) wv[u] = (iok && ol + u < oend) ? w[(size_t)(ob + ol + u) * In + i] : 0.f;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const lf4* g = reinterpret_cast<const lf4*>(sm + (size_t)(ol + u < 1024 ? ol + u : 1023) * LW_BT);
#pragma unroll
        for (int q = 0; q < LW_BT / 4; ++q) {
          const lf4 gv = g[q];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[q * 4 + e] = fmaf(gv[e], wv[u], acc[q * 4 + e]);
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < LW_BT; ++r) sm[(wave * LW_BT + r) * 64 + lane] = acc[r];  // [16 waves][16 b][64 i]
  __syncthreads();
  {
    const int r = wave;  // 16 waves = 16 batch rows
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) v += sm[(q * LW_BT + r) * 64 + lane];
    if (iok && b0 + r < B) dx[(size_t)(b0 + r) * In + i] = v;
  }
}

// dw: thread = 4 consecutive i x 16 output rows: x of the batch stays in registers, dy comes through scalar loads
// (its address is uniform in the block), every dw element is read (beta) and written once
__global__ __launch_bounds__(256) void k_linear_dw_wide(const float* __restrict__ dy, const float* __restrict__ x,
                                                        float* __restrict__ dw, int B, int In, int Out, float beta) {
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int o0 = blockIdx.y * 16;
  if (i >= In) return;
  lf4 acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = (lf4){0.f, 0.f, 0.f, 0.f};
  for (int b0 = 0; b0 < B; b0 += LW_BT) {
    // (every load unconditional, from a clamped row; rows past the batch get a zero dy: a conditional load is followed by
    //  s_waitcnt vmcnt(0) and this kernel is nothing but loads)
    lf4 xv[LW_BT];
#pragma unroll
    for (int r = 0; r < LW_BT; ++r)
      xv[r] = *reinterpret_cast<const lf4*>(x + (size_t)(b0 + r < B ? b0 + r : B - 1) * In + i);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int o = o0 + k < Out ? o0 + k : Out - 1;
#pragma unroll
      for (int r = 0; r < LW_BT; ++r) {
        float g = dy[(size_t)(b0 + r < B ? b0 + r : B - 1) * Out + o];
        g = b0 + r < B ? g : 0.f;
        acc[k] += g * xv[r];
      }
    }
  }
  // read-modify-write of 16 rows: all reads first (a row past Out re-reads the last one and is not written)
  if (beta != 0.f) {
    lf4 old[16];
#pragma unroll
    for (int k = 0; k < 16; ++k)
      old[k] = *reinterpret_cast<const lf4*>(dw + (size_t)(o0 + k < Out ? o0 + k : Out - 1) * In + i);
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] += beta * old[k];
  }
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (o0 + k < Out) *reinterpret_cast<lf4*>(dw + (size_t)(o0 + k) * In + i) = acc[k];
}

static bool linear_wide(const void* a, const void* b, const void* c, int In, int Out) {
  const bool off = env_int("SRK_LINEAR_WIDE", 1) == 0;
  if (off || (In & 3) || In < 2048 || Out < 64) return false;
  return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0;
}

__global__ __launch_bounds__(256) void k_linear_db(const float* __restrict__ dy, float* __restrict__ db, int B,
                                                   int Out, float beta) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= Out) return;
  float acc = 0.f;
  for (int b = 0; b < B; ++b) acc += dy[(size_t)b * Out + o];
  db[o] = beta != 0.f ? beta * db[o] + acc : acc;
}

}  // namespace srk

using namespace srk;

extern "C" size_t srk_bn_workspace_bytes(int C) { return (size_t)kBnRowSplits * 2 * (size_t)(C > 0 ? C : 0) * sizeof(double); }

extern "C" int srk_bn_stats(const float* x, double* stats, size_t rows, int C, void* workspace, void* stream) {
  SRK_REQUIRE(x && stats && workspace && rows > 0 && C > 0, "bn_stats: bad args");
  return bn_colsum(0, x, nullptr, nullptr, nullptr, stats, rows, C, workspace, (hipStream_t)stream);
}

extern "C" int srk_bn_stats_finalize(const float* x, double* stats, size_t rows, int C, float* save_mean, float* save_rstd,
                                     float* running_mean, float* running_var, float momentum, float eps,
                                     int64_t* num_batches_tracked, void* workspace, void* stream) {
  SRK_REQUIRE(x && stats && workspace && save_mean && save_rstd && rows > 0 && C > 0, "bn_stats_finalize: bad args");
  BnFused fu;
  fu.mode = 0; fu.count = (double)rows; fu.o0 = save_mean; fu.o1 = save_rstd; fu.rm = running_mean; fu.rv = running_var;
  fu.momentum = momentum; fu.eps = eps; fu.nbt = (long long*)num_batches_tracked;
  return bn_colsum(0, x, nullptr, nullptr, nullptr, stats, rows, C, workspace, (hipStream_t)stream, fu);
}

extern "C" int srk_bn_finalize_partials(const double* partials, int splits, double* stats, size_t rows, int C,
                                        float* save_mean, float* save_rstd, float* running_mean, float* running_var,
                                        float momentum, float eps, int64_t* num_batches_tracked, void* stream) {
  SRK_REQUIRE(partials && stats && save_mean && save_rstd && splits > 0 && rows > 0 && C > 0, "bn_finalize_partials: bad args");
  hipStream_t s = (hipStream_t)stream;
  if (splits > 64 && env_int("SRK_BN_RED16", 1) != 0)
    hipLaunchKernelGGL((k_bn_reduce16<0>), dim3(cdiv(C, 16)), dim3(1024), 0, s, partials, stats, splits, C, (double)rows,
                       save_mean, save_rstd, running_mean, running_var, momentum, eps, (long long*)num_batches_tracked);
  else if (splits > 64)
    hipLaunchKernelGGL((k_bn_reduce_fused<0, 16>), dim3(cdiv(C, 64)), dim3(1024), 0, s, partials, stats, splits, C,
                       (double)rows, save_mean, save_rstd, running_mean, running_var, momentum, eps,
                       (long long*)num_batches_tracked);
  else
    hipLaunchKernelGGL((k_bn_reduce_fused<0, 4>), dim3(cdiv(C, 64)), dim3(256), 0, s, partials, stats, splits, C,
                       (double)rows, save_mean, save_rstd, running_mean, running_var, momentum, eps,
                       (long long*)num_batches_tracked);
  return check_launch("bn_finalize_partials");
}

extern "C" int srk_bn_backward_stats_grads(const float* dy, const float* x, const float* mean, const float* rstd,
                                           double* dstats, size_t rows, int C, float* dgamma, float* dbeta, void* workspace,
                                           void* stream) {
  SRK_REQUIRE(dy && x && mean && rstd && dstats && workspace && rows > 0 && C > 0, "bn_backward_stats_grads: bad args");
  BnFused fu;
  fu.mode = 1; fu.o0 = dgamma; fu.o1 = dbeta;
  return bn_colsum(1, dy, x, mean, rstd, dstats, rows, C, workspace, (hipStream_t)stream, fu);
}

extern "C" int srk_bn_finalize(const double* stats, double count, float* save_mean, float* save_rstd,
                               float* running_mean, float* running_var, float momentum, float eps, int C,
                               int64_t* num_batches_tracked, void* stream) {
  SRK_REQUIRE(stats && save_mean && save_rstd && C > 0 && count > 0, "bn_finalize: bad args");
  hipLaunchKernelGGL(k_bn_finalize, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, stats, count, save_mean,
                     save_rstd, running_mean, running_var, momentum, eps, C, (long long*)num_batches_tracked);
  return check_launch("bn_finalize");
}

extern "C" int srk_bn_eval_params(const float* running_mean, const float* running_var, float eps, float* mean,
                                  float* rstd, int C, void* stream) {
  SRK_REQUIRE(running_mean && running_var && mean && rstd && C > 0, "bn_eval_params: bad args");
  hipLaunchKernelGGL(k_bn_eval_params, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, running_mean,
                     running_var, eps, mean, rstd, C);
  return check_launch("bn_eval_params");
}

// ---------------------------------------------------------------------------------------------
// Row normalisation: nn.InstanceNorm1d(F) applied to a [B, F] activation (DenseBlock(norm='instance'),
// base_networks.py:12-13).  torch reads the 2-D input as ONE unbatched sample of B channels x F positions, so every ROW
// is normalised with its own biased statistics over the F features (no affine parameters, no running statistics).
// One 256-thread block per row; sums in double.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rownorm_fwd(const float* __restrict__ x, float* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int cols,
                                                     float eps) {
  __shared__ double sm[4];
  __shared__ float stat[2];
  const float* xr = x + (size_t)blockIdx.x * cols;
  float* yr = y + (size_t)blockIdx.x * cols;
  double s = 0.0;
  for (int i = threadIdx.x; i < cols; i += 256) s += (double)xr[i];
  const double m = block_sum_256_d(s, sm) / cols;
  double q = 0.0;
  for (int i = threadIdx.x; i < cols; i += 256) {
    const double d = (double)xr[i] - m;
    q += d * d;
  }
  const double var = block_sum_256_d(q, sm) / cols;
  if (threadIdx.x == 0) {
    stat[0] = (float)m;
    stat[1] = (float)(1.0 / sqrt(var + (double)eps));
    mean[blockIdx.x] = stat[0];
    rstd[blockIdx.x] = stat[1];
  }
  __syncthreads();
  const float mf = stat[0], rf = stat[1];
  for (int i = threadIdx.x; i < cols; i += 256) yr[i] = (xr[i] - mf) * rf;
}

// dx = rstd * (dy - mean(dy) - xhat * mean(dy * xhat)) per row
__global__ __launch_bounds__(256) void k_rownorm_bwd(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     float* __restrict__ dx, int cols) {
  __shared__ double sm[4];
  const size_t base = (size_t)blockIdx.x * cols;
  const float mf = mean[blockIdx.x], rf = rstd[blockIdx.x];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < cols; i += 256) {
    const float g = dy[base + i], xh = (x[base + i] - mf) * rf;
    a += (double)g;
    b += (double)g * (double)xh;
  }
  const double sa = block_sum_256_d(a, sm) / cols;
  __syncthreads();
  const double sb = block_sum_256_d(b, sm) / cols;
  const float fa = (float)sa, fb = (float)sb;
  for (int i = threadIdx.x; i < cols; i += 256) {
    const float xh = (x[base + i] - mf) * rf;
    dx[base + i] = rf * (dy[base + i] - fa - xh * fb);
  }
}

extern "C" int srk_rownorm_forward(const float* x, float* y, float* mean, float* rstd, int rows, int cols, float eps,
                                   void* stream) {
  SRK_REQUIRE(x && y && mean && rstd && rows > 0 && cols > 0, "rownorm_forward: bad args");
  hipLaunchKernelGGL(k_rownorm_fwd, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, y, mean, rstd, cols, eps);
  return check_launch("rownorm_forward");
}

extern "C" int srk_rownorm_backward(const float* dy, const float* x, const float* mean, const float* rstd, float* dx,
                                    int rows, int cols, void* stream) {
  SRK_REQUIRE(dy && x && mean && rstd && dx && rows > 0 && cols > 0, "rownorm_backward: bad args");
  hipLaunchKernelGGL(k_rownorm_bwd, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, dy, x, mean, rstd, dx, cols);
  return check_launch("rownorm_backward");
}

extern "C" int srk_bn_apply(const float* x, float* y, const float* mean, const float* rstd, const float* gamma,
                            const float* beta, size_t rows, int C, int act, float slope, float* y_amax, void* stream) {
  SRK_REQUIRE(x && y && mean && rstd && rows > 0 && C > 0, "bn_apply: bad args");
  SRK_REQUIRE(act != SRK_ACT_PRELU, "bn_apply: PReLU is not fused here (use srk_act_forward)");
  const size_t total = rows * (size_t)C;
  size_t nb = (total + 256 * 4 - 1) / (256 * 4);
  if (nb > 4096) nb = 4096;
  if (bn_vec4(C, x, y, mean, rstd, gamma, beta))
    hipLaunchKernelGGL(k_bn_apply4, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, y, mean, rstd, gamma, beta,
                       total / 4, C, act, slope, y_amax);
  else if (y_amax) {
    set_error("bn_apply: y_amax needs the 16-byte path (C %% 4 == 0, aligned tensors)");
    return SRK_ERR_UNSUPPORTED;
  } else
    hipLaunchKernelGGL(k_bn_apply, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, y, mean, rstd, gamma, beta,
                       total, C, act, slope);
  return check_launch("bn_apply");
}

extern "C" int srk_bn_backward_stats(const float* dy, const float* x, const float* mean, const float* rstd,
                                     double* dstats, size_t rows, int C, void* workspace, void* stream) {
  SRK_REQUIRE(dy && x && mean && rstd && dstats && workspace && rows > 0 && C > 0, "bn_backward_stats: bad args");
  return bn_colsum(1, dy, x, mean, rstd, dstats, rows, C, workspace, (hipStream_t)stream);
}

extern "C" int srk_bn_backward_apply(const float* dy, const float* x, const float* mean, const float* rstd,
                                     const float* gamma, const double* dstats, double count, float* dx, size_t rows,
                                     int C, void* stream) {
  SRK_REQUIRE(dy && x && mean && rstd && dstats && dx && rows > 0 && C > 0 && count > 0, "bn_backward_apply: bad args");
  const size_t total = rows * (size_t)C;
  size_t nb = (total + 256 * 4 - 1) / (256 * 4);
  if (nb > 4096) nb = 4096;
  const bool v4 = bn_vec4(C, dy, x, mean, rstd, gamma, dx);
  if (bn_fp32_backward()) {
    if (v4)
      hipLaunchKernelGGL(k_bn_bwd_apply4<false>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, dy, x, mean, rstd,
                         gamma, dstats, count, dx, total / 4, C);
    else
      hipLaunchKernelGGL(k_bn_bwd_apply<false>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, dy, x, mean, rstd,
                         gamma, dstats, count, dx, total, C);
  } else if (v4) {
    hipLaunchKernelGGL(k_bn_bwd_apply4<true>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, dy, x, mean, rstd,
                       gamma, dstats, count, dx, total / 4, C);
  } else {
    hipLaunchKernelGGL(k_bn_bwd_apply<true>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, dy, x, mean, rstd,
                       gamma, dstats, count, dx, total, C);
  }
  return check_launch("bn_backward_apply");
}

static int bn_act_check(int act, const float* prelu_w, int prelu_n, int C, const char* who) {
  SRK_REQUIRE(act == SRK_ACT_NONE || act == SRK_ACT_RELU || act == SRK_ACT_LRELU || act == SRK_ACT_PRELU,
              "%s: only ReLU / LeakyReLU / PReLU fold into BatchNorm", who);
  if (act == SRK_ACT_PRELU) SRK_REQUIRE(prelu_w && (prelu_n == 1 || prelu_n == C), "%s: PReLU needs 1 or C slopes", who);
  return SRK_OK;
}

extern "C" int srk_bn_apply_act(const float* x, float* y, const float* mean, const float* rstd, const float* gamma,
                                const float* beta, size_t rows, int C, int act, float slope, const float* prelu_weight,
                                int prelu_n, const float* residual, float* y_amax, void* stream) {
  SRK_REQUIRE(x && y && mean && rstd && rows > 0 && C > 0, "bn_apply_act: bad args");
  int rc = bn_act_check(act, prelu_weight, prelu_n, C, "bn_apply_act");
  if (rc) return rc;
  SRK_REQUIRE(bn_vec4(C, x, y, mean, rstd, gamma, beta) && ((uintptr_t)residual & 15) == 0,
              "bn_apply_act: C must be a multiple of 4 and the tensors 16-byte aligned");
  const size_t total = rows * (size_t)C;
  size_t nb = (total + 256 * 4 - 1) / (256 * 4);
  if (nb > 4096) nb = 4096;
  BnAct A{gamma, beta, prelu_weight, act, prelu_n, slope};
  hipLaunchKernelGGL(k_bn_apply_act4, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, y, mean, rstd, A, residual,
                     total / 4, C, y_amax);
  return check_launch("bn_apply_act");
}

extern "C" int srk_bn_backward_stats_grads_act(const float* dy, const float* x, const float* mean, const float* rstd,
                                               const float* gamma, const float* beta, double* dstats, size_t rows, int C,
                                               float* dgamma, float* dbeta, int act, float slope,
                                               const float* prelu_weight, int prelu_n, float* dprelu, void* workspace,
                                               void* stream) {
  SRK_REQUIRE(dy && x && mean && rstd && dstats && workspace && rows > 0 && C > 0, "bn_backward_stats_grads_act: bad args");
  int rc = bn_act_check(act, prelu_weight, prelu_n, C, "bn_backward_stats_grads_act");
  if (rc) return rc;
  int splits = (int)((rows + 127) / 128);
  if (splits > kBnRowSplits * 2 / 3) splits = kBnRowSplits * 2 / 3;  // three sums per split in the same workspace
  if (splits < 1) splits = 1;
  const size_t rps = (rows + splits - 1) / splits;
  hipStream_t s = (hipStream_t)stream;
  BnAct A{gamma, beta, prelu_weight, act, prelu_n, slope};
  hipLaunchKernelGGL(k_bn_colsum_act, dim3(cdiv(C, 64), splits), dim3(256), 0, s, dy, x, mean, rstd, A, (double*)workspace,
                     rows, C, rps);
  if (splits > 64)
    hipLaunchKernelGGL(k_bn_reduce_act<16>, dim3(cdiv(C, 64)), dim3(1024), 0, s, (const double*)workspace, dstats, splits, C,
                       dgamma, dbeta, act == SRK_ACT_PRELU ? dprelu : nullptr, prelu_n);
  else
    hipLaunchKernelGGL(k_bn_reduce_act<4>, dim3(cdiv(C, 64)), dim3(256), 0, s, (const double*)workspace, dstats, splits, C,
                       dgamma, dbeta, act == SRK_ACT_PRELU ? dprelu : nullptr, prelu_n);
  return check_launch("bn_backward_stats_grads_act");
}

extern "C" int srk_bn_backward_apply_act(const float* dy, const float* x, const float* mean, const float* rstd,
                                         const float* gamma, const float* beta, const double* dstats, double count,
                                         float* dx, size_t rows, int C, int act, float slope, const float* prelu_weight,
                                         int prelu_n, void* stream) {
  SRK_REQUIRE(dy && x && mean && rstd && dstats && dx && rows > 0 && C > 0 && count > 0, "bn_backward_apply_act: bad args");
  int rc = bn_act_check(act, prelu_weight, prelu_n, C, "bn_backward_apply_act");
  if (rc) return rc;
  SRK_REQUIRE(bn_vec4(C, dy, x, mean, rstd, gamma, dx) && ((uintptr_t)beta & 15) == 0,
              "bn_backward_apply_act: C must be a multiple of 4 and the tensors 16-byte aligned");
  const size_t total = rows * (size_t)C;
  size_t nb = (total + 256 * 4 - 1) / (256 * 4);
  if (nb > 4096) nb = 4096;
  BnAct A{gamma, beta, prelu_weight, act, prelu_n, slope};
  if (bn_fp32_backward())
    hipLaunchKernelGGL(k_bn_bwd_apply_act4<false>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, dy, x, mean, rstd, A,
                       dstats, count, dx, total / 4, C);
  else
    hipLaunchKernelGGL(k_bn_bwd_apply_act4<true>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, dy, x, mean, rstd, A,
                       dstats, count, dx, total / 4, C);
  return check_launch("bn_backward_apply_act");
}

// ---- finalize-in-apply entry points (round 6) ---------------------------------------------------------------------------
extern "C" int srk_bn_fused_supported(int C) { return C > 0 && C % BNF_CS == 0 && C <= 512 && env_int("SRK_BN_FIN_APPLY", 1) != 0; }

extern "C" int srk_bn_stats_partials(const float* x, size_t rows, int C, void* workspace, int* splits_out, void* stream) {
  SRK_REQUIRE(x && workspace && splits_out && rows > 0 && C > 0, "bn_stats_partials: bad args");
  int splits = (int)((rows + 63) / 64);
  if (splits > kBnRowSplits) splits = kBnRowSplits;
  if (splits < 1) splits = 1;
  const size_t rps = (rows + splits - 1) / splits;
  hipLaunchKernelGGL((k_bn_colsum<0, false>), dim3(cdiv(C, 64), splits), dim3(256), 0, (hipStream_t)stream, x, nullptr, nullptr,
                     nullptr, (double*)workspace, rows, C, rps);
  *splits_out = splits;
  return check_launch("bn_stats_partials");
}

extern "C" int srk_bn_finalize_apply_act(const double* partials, int splits, double* stats, size_t rows, int C, float* save_mean,
                                         float* save_rstd, float* running_mean, float* running_var, float momentum, float eps,
                                         int64_t* num_batches_tracked, const float* x, float* y, const float* gamma,
                                         const float* beta, int act, float slope, const float* prelu_weight, int prelu_n,
                                         const float* residual, float* y_amax, void* stream) {
  SRK_REQUIRE(partials && stats && save_mean && save_rstd && x && y && splits > 0 && rows > 0, "bn_finalize_apply_act: bad args");
  SRK_REQUIRE(srk_bn_fused_supported(C), "bn_finalize_apply_act: C must be a multiple of 16, <= 512");
  int rc = bn_act_check(act, prelu_weight, prelu_n, C, "bn_finalize_apply_act");
  if (rc) return rc;
  SRK_REQUIRE(bn_vec4(C, x, y, save_mean, save_rstd, gamma, beta) && ((uintptr_t)residual & 15) == 0,
              "bn_finalize_apply_act: tensors must be 16-byte aligned");
  dim3 grid;
  size_t rpb;
  bnf_grid(rows, C, grid, rpb);
  BnAct A{gamma, beta, prelu_weight, act, prelu_n, slope};
  hipLaunchKernelGGL(k_bn_fin_apply_act, grid, dim3(BNF_THR), 0, (hipStream_t)stream, partials, splits, (double)rows, x, y, A,
                     residual, rows, C, rpb, save_mean, save_rstd, stats, running_mean, running_var, momentum, eps,
                     (long long*)num_batches_tracked, y_amax);
  return check_launch("bn_finalize_apply_act");
}

// column sums of the backward only: [splits][3][C] (act != NONE: sum dz, sum dz * xhat, sum_{z<=0} dy * z) or [splits][2][C]
extern "C" int srk_bn_backward_partials_act(const float* dy, const float* x, const float* mean, const float* rstd,
                                            const float* gamma, const float* beta, size_t rows, int C, int act, float slope,
                                            const float* prelu_weight, int prelu_n, void* workspace, int* splits_out,
                                            void* stream) {
  SRK_REQUIRE(dy && x && mean && rstd && workspace && splits_out && rows > 0 && C > 0, "bn_backward_partials_act: bad args");
  int rc = bn_act_check(act, prelu_weight, prelu_n, C, "bn_backward_partials_act");
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (act == SRK_ACT_NONE) {
    int splits = (int)((rows + 63) / 64);
    if (splits > kBnRowSplits) splits = kBnRowSplits;
    if (splits < 1) splits = 1;
    const size_t rps = (rows + splits - 1) / splits;
    dim3 grid(cdiv(C, 64), splits);
    if (bn_fp32_backward())
      hipLaunchKernelGGL((k_bn_colsum<1, false>), grid, dim3(256), 0, s, dy, x, mean, rstd, (double*)workspace, rows, C, rps);
    else
      hipLaunchKernelGGL((k_bn_colsum<1, true>), grid, dim3(256), 0, s, dy, x, mean, rstd, (double*)workspace, rows, C, rps);
    *splits_out = splits;
    return check_launch("bn_backward_partials_act");
  }
  int splits = (int)((rows + 127) / 128);
  if (splits > kBnRowSplits * 2 / 3) splits = kBnRowSplits * 2 / 3;  // three sums per split in the same workspace
  if (splits < 1) splits = 1;
  const size_t rps = (rows + splits - 1) / splits;
  BnAct A{gamma, beta, prelu_weight, act, prelu_n, slope};
  hipLaunchKernelGGL(k_bn_colsum_act, dim3(cdiv(C, 64), splits), dim3(256), 0, s, dy, x, mean, rstd, A, (double*)workspace, rows,
                     C, rps);
  *splits_out = splits;
  return check_launch("bn_backward_partials_act");
}

extern "C" int srk_bn_backward_finalize_apply_act(const double* partials, int splits, double* dstats, double count,
                                                  const float* dy, const float* x, const float* mean, const float* rstd,
                                                  const float* gamma, const float* beta, float* dx, size_t rows, int C,
                                                  float* dgamma, float* dbeta, int act, float slope,
                                                  const float* prelu_weight, int prelu_n, float* dprelu, void* stream) {
  SRK_REQUIRE(partials && dstats && dy && x && mean && rstd && dx && splits > 0 && rows > 0 && count > 0,
              "bn_backward_finalize_apply_act: bad args");
  SRK_REQUIRE(srk_bn_fused_supported(C), "bn_backward_finalize_apply_act: C must be a multiple of 16, <= 512");
  int rc = bn_act_check(act, prelu_weight, prelu_n, C, "bn_backward_finalize_apply_act");
  if (rc) return rc;
  SRK_REQUIRE(bn_vec4(C, dy, x, mean, rstd, gamma, dx) && ((uintptr_t)beta & 15) == 0,
              "bn_backward_finalize_apply_act: tensors must be 16-byte aligned");
  dim3 grid;
  size_t rpb;
  bnf_grid(rows, C, grid, rpb);
  BnAct A{gamma, beta, prelu_weight, act, prelu_n, slope};
  hipStream_t s = (hipStream_t)stream;
  const bool f32 = bn_fp32_backward();
  float* dp = act == SRK_ACT_PRELU ? dprelu : nullptr;
  const int pq = act == SRK_ACT_NONE ? 2 : 3;   // planes per split row (srk_bn_backward_partials_act)
#define SRK_BNF_LAUNCH(NQ, DBL)                                                                                            \
  hipLaunchKernelGGL((k_bn_fin_bwd_apply_act<NQ, DBL>), grid, dim3(BNF_THR), 0, s, partials, splits, pq, count, dy, x, mean, rstd, A, \
                     dx, rows, C, rpb, dstats, dgamma, dbeta, dp)
  if (!dp) {    // no PReLU slope gradient wanted: two sums
    if (f32) SRK_BNF_LAUNCH(2, false); else SRK_BNF_LAUNCH(2, true);
  } else {
    if (f32) SRK_BNF_LAUNCH(3, false); else SRK_BNF_LAUNCH(3, true);
  }
#undef SRK_BNF_LAUNCH
  return check_launch("bn_backward_finalize_apply_act");
}

extern "C" int srk_bn_param_grads(const double* dstats, float* dgamma, float* dbeta, int C, void* stream) {
  SRK_REQUIRE(dstats && C > 0, "bn_param_grads: bad args");
  hipLaunchKernelGGL(k_bn_param_grads, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, dstats, dgamma, dbeta, C);
  return check_launch("bn_param_grads");
}

extern "C" int srk_linear_forward(const float* x, const float* w, const float* b, float* y, int B, int In, int Out,
                                  int act, float slope, void* stream) {
  SRK_REQUIRE(x && w && y && B > 0 && In > 0 && Out > 0, "linear_forward: bad args");
  SRK_REQUIRE(act != SRK_ACT_PRELU, "linear_forward: PReLU is not fused here");
  if (linear_wide(x, w, nullptr, In, Out)) {
    hipLaunchKernelGGL(k_linear_fwd_wide, dim3(cdiv(Out, 4), cdiv(B, LW_BT)), dim3(512), 0, (hipStream_t)stream, x, w, b, y,
                       B, In, Out, act, slope);
    return check_launch("linear_forward");
  }
  dim3 grid(cdiv(Out, 4), cdiv(B, LIN_BT));
  hipLaunchKernelGGL(k_linear_fwd, grid, dim3(256), 0, (hipStream_t)stream, x, w, b, y, B, In, Out, act, slope);
  return check_launch("linear_forward");
}

extern "C" int srk_linear_backward(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db,
                                   int B, int In, int Out, float beta, void* stream) {
  SRK_REQUIRE(x && w && dy && B > 0 && In > 0 && Out > 0, "linear_backward: bad args");
  hipStream_t s = (hipStream_t)stream;
  const bool wide = linear_wide(x, w, dw ? (const void*)dw : (const void*)dx, In, Out) && ((uintptr_t)dx & 15) == 0;
  if (dx && wide)
    hipLaunchKernelGGL(k_linear_dx_wide, dim3(cdiv(In, 64), cdiv(B, LW_BT)), dim3(1024), 0, s, dy, w, dx, B, In, Out);
  else if (dx)
    hipLaunchKernelGGL(k_linear_dx, dim3(cdiv(In, 256), cdiv(B, LIN_BT)), dim3(256), 0, s, dy, w, dx, B, In, Out);
  if (dw && wide) {
    hipLaunchKernelGGL(k_linear_dw_wide, dim3(cdiv(In / 4, 256), cdiv(Out, 16)), dim3(256), 0, s, dy, x, dw, B, In, Out,
                       beta);
  } else if (dw) {
    size_t nb = ((size_t)Out * ((In + 3) / 4) + 255) / 256;
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(k_linear_dw, dim3((unsigned)nb), dim3(256), 0, s, dy, x, dw, B, In, Out, beta);
  }
  if (db) hipLaunchKernelGGL(k_linear_db, dim3(cdiv(Out, 256)), dim3(256), 0, s, dy, db, B, Out, beta);
  return check_launch("linear_backward");
}
