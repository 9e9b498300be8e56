// The consumer stage loop of k_conv_bfr in isolation (one wave per SIMD, fragments from an LDS full of noise, no producers,
// no ring): ticks per 18-step stage against the 216 x 16 = 3456 cycles of its MFMAs, features switched on one by one.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tap_loop.hip -o tap_loop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <type_traits>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int I0, int I1, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I0 < I1) {
    f(std::integral_constant<int, I0>{});
    static_for<I0 + 1, I1>(f);
  }
}
__device__ __forceinline__ void touch(uint4& u) { asm volatile("" : "+v"(u.x), "+v"(u.y), "+v"(u.z), "+v"(u.w)); }
__device__ __forceinline__ f32x4 mfma(const uint4& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// FEAT bits: 1 fragment loads, 2 stores (one per 4 steps), 4 peek + signal, 8 park arithmetic per 2 stages, 16 sched_barriers
template <int NTW, int FEAT>
__global__ __launch_bounds__(256) void k(float* out, long long* clk, int stages, unsigned out_bytes) {
  constexpr int NB = 16 * NTW, WSLOT = 8 * NB, HW = 18, NPIXP = 190, PLANE_B = 4 * NPIXP, PLANE_A = 4 * NB, MR = 4, ICC = 2;
  extern __shared__ uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kq = lane >> 4;
  for (int e = tid; e < 9000; e += 256) lds[e] = uint4{0x3c003c00u + e, 0x38003800u, 0x34003400u + tid, 0x30003000u};
  __shared__ unsigned cnt[8];
  if (tid < 8) cnt[tid] = 1000000u;
  __syncthreads();
  const int pj = j < 4 ? 2 * j : (j < 12 ? 2 * j - 7 : 2 * j - 16);
  const uint4* wl = lds;
  const uint4* hb = lds + 9 * ICC * WSLOT + (wave & 1) * MR * HW + pj + kq * NPIXP;
  const int lane_a = kq * NB + j;
  f32x4 acc[NTW][MR], pend[NTW][MR];
  for (int nt = 0; nt < NTW; ++nt)
    for (int r = 0; r < MR; ++r) { acc[nt][r] = (f32x4){0.f, 0.f, 0.f, 0.f}; pend[nt][r] = (f32x4){1.f, 2.f, 3.f, 4.f}; }
  uint4 fa[3][2][NTW], fb[3][2];
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, out_bytes, 0x00020000);
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  unsigned voff = (blockIdx.x * 256 + tid) * 64u;
  auto ldA = [&](auto uc, auto vc, auto ccc) {
    constexpr int u = decltype(uc)::value, v = decltype(vc)::value, cc = decltype(ccc)::value;
    const uint4* wb = wl + lane_a;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      if constexpr (FEAT & 1) {
        fa[u][0][nt] = wb[((u * 3 + v) * ICC + cc) * WSLOT + nt * 16];
        fa[u][1][nt] = wb[((u * 3 + v) * ICC + cc) * WSLOT + PLANE_A + nt * 16];
      } else {
        touch(fa[u][0][nt]);
        touch(fa[u][1][nt]);
      }
    }
  };
  auto ldB = [&](auto sc) {
    constexpr int s = decltype(sc)::value, v = s / 6, R = s - 6 * v;
    if constexpr (FEAT & 1) {
      fb[s % 3][0] = hb[R * HW + v];
      fb[s % 3][1] = hb[R * HW + v + PLANE_B];
    } else {
      touch(fb[s % 3][0]);
      touch(fb[s % 3][1]);
    }
  };
  auto mfma3 = [&](auto uc, auto sc) {
    constexpr int u = decltype(uc)::value, s = decltype(sc)::value, R = s % 6, r = R - u;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[nt][r] = mfma(fa[u][0][nt], fb[s % 3][1], acc[nt][r]);
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[nt][r] = mfma(fa[u][1][nt], fb[s % 3][0], acc[nt][r]);
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[nt][r] = mfma(fa[u][0][nt], fb[s % 3][0], acc[nt][r]);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  for (int u = 0; u < 3; ++u)
    for (int p = 0; p < 2; ++p) {
      for (int nt = 0; nt < NTW; ++nt) fa[u][p][nt] = lds[lane + 64 * (u * 2 + p) + nt];
      fb[u][p] = lds[lane + 777 + 64 * (u * 2 + p)];
    }
  float amax = 0.f;
  const long long t0 = clock64();
  for (int st = 0; st < stages; st += 2) {
    static_for<0, ICC>([&](auto ccc) {
      constexpr int cc = decltype(ccc)::value, ccn = cc + 1 < ICC ? cc + 1 : 0;
      unsigned peek_v = 0;
      static_for<0, 18>([&](auto sc) {
        constexpr int s = decltype(sc)::value, v = s / 6, R = s - 6 * v;
        constexpr int u_lo = R - (MR - 1) > 0 ? R - (MR - 1) : 0, u_hi = R < 2 ? R : 2;
        constexpr int g = cc * 18 + s;
        static_for<u_lo, u_hi + 1>([&](auto uc) {
          constexpr int u = decltype(uc)::value;
          mfma3(uc, sc);
          if constexpr (FEAT & 16) __builtin_amdgcn_sched_barrier(0);
          if constexpr (u == u_lo) {
            ldB(std::integral_constant<int, (s + 2) % 18>{});
            if constexpr ((FEAT & 2) != 0 && g % 4 == 0 && g / 4 < NTW * MR) {
              constexpr int q = g / 4;
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, pend[q % NTW][q / NTW]), orsrc, (int)(voff + 16u * (q & 3)), 0, 0);
            }
            if constexpr (FEAT & 4) {
              if constexpr (s == 8) peek_v = __hip_atomic_load(cnt + cc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              if constexpr (s == 14) {
                unsigned seen = (unsigned)__builtin_amdgcn_readfirstlane((int)peek_v);
                while ((int)(seen - 4u) < 0) {
                  __builtin_amdgcn_s_sleep(1);
                  seen = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(cnt + cc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                }
                asm volatile("" ::: "memory");
              }
              if constexpr (s == 15) {
                asm volatile("" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(cnt + 4 + cc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                asm volatile("" ::: "memory");
              }
            }
            if constexpr (FEAT & 16) __builtin_amdgcn_sched_barrier(0);
            if constexpr (R == 0) {
              ldA(I2{}, std::integral_constant<int, v>{}, ccc);
            } else if constexpr (R >= 4) {
              if constexpr (v < 2) ldA(std::integral_constant<int, R - 4>{}, std::integral_constant<int, v + 1>{}, ccc);
              else ldA(std::integral_constant<int, R - 4>{}, I0{}, std::integral_constant<int, ccn>{});
            }
            if constexpr (FEAT & 16) __builtin_amdgcn_sched_barrier(0);
          }
        });
      });
    });
    if constexpr (FEAT & 8) {
      voff += 16u;
#pragma unroll
      for (int r = 0; r < MR; ++r) {
        float rmax = 0.f;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          f32x4 v = __builtin_elementwise_fma(acc[nt][r], (f32x4){0.5f, 0.5f, 0.5f, 0.5f}, (f32x4){0.1f, 0.2f, 0.3f, 0.4f});
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          rmax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), rmax);
          rmax = fmaxf(fmaxf(fabsf(v[2]), fabsf(v[3])), rmax);
          pend[nt][r] = v;
          acc[nt][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        amax = fmaxf(amax, rmax);
      }
    }
  }
  const long long t1 = clock64();
  f32x4 s4 = (f32x4){amax, 0.f, 0.f, 0.f};
  for (int nt = 0; nt < NTW; ++nt)
    for (int r = 0; r < MR; ++r) s4 += acc[nt][r] + pend[nt][r];
  out[(blockIdx.x * 256 + tid) * 16 + 15] = s4[0] + s4[1] + s4[2] + s4[3];
  if (lane == 0) clk[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int NTW, int FEAT>
void run(const char* what, float* out, long long* clk) {
  const int stages = 400;
  const size_t lds = 150 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<NTW, FEAT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  k<NTW, FEAT><<<256, 256, lds>>>(out, clk, 4, 256u * 256 * 64);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<NTW, FEAT><<<256, 256, lds>>>(out, clk, stages, 256u * 256 * 64);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(1024);
  hipMemcpy(h.data(), clk, sizeof(long long) * 1024, hipMemcpyDeviceToHost);
  double sum = 0; for (auto v : h) sum += v;
  const double mf = 72.0 * NTW * 1.5;  // MFMAs per stage: 4 rows x 9 taps x NTW x 3
  printf("NTW %d %-44s: %7.0f ticks per stage (%5.2f per MFMA), %6.1f TF of products (x3 MFMAs: %6.1f)\n", NTW, what, sum / 1024 / stages,
         sum / 1024 / stages / mf, 2.0 * 16 * 16 * 32 * mf * stages * 1024 / (ms * 1e-3) / 1e12 / 3, 2.0 * 16 * 16 * 32 * mf * stages * 1024 / (ms * 1e-3) / 1e12);
}

int main() {
  float* out; long long* clk;
  hipMalloc(&out, 256u * 256 * 64);
  hipMalloc(&clk, sizeof(long long) * 1024);
  run<2, 0>("MFMAs only", out, clk);
  run<2, 16>("MFMAs + sched_barriers", out, clk);
  run<2, 1>("+ fragment loads (no barriers)", out, clk);
  run<2, 17>("+ fragment loads, barriers", out, clk);
  run<2, 19>("+ loads, barriers, stores", out, clk);
  run<2, 21>("+ loads, barriers, peek/signal", out, clk);
  run<2, 25>("+ loads, barriers, park", out, clk);
  run<2, 31>("everything", out, clk);
  run<2, 15>("everything, no barriers", out, clk);
  run<3, 0>("MFMAs only", out, clk);
  run<3, 17>("+ fragment loads, barriers", out, clk);
  run<3, 31>("everything", out, clk);
  run<3, 15>("everything, no barriers", out, clk);
  return 0;
}
