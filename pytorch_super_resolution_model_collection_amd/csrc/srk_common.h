// Shared helpers for the libsrk kernels (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>
#include "../../include/srk.h"

namespace srk {

void set_error(const char* fmt, ...);
// Records the kernel a conv entry point just launched (thread-local; read back through srk_last_kernel_name()) so that
// measurements can name the kernel that actually ran instead of guessing it from the shape.
void note_kernel(const char* fmt, ...);
// A conv launcher whose kernel keeps the running maximum of what it stores (srk_epilogue.y_amax) says so here; read back
// through srk_last_conv_wrote_amax().  Reset by every srk_conv2d_forward call.
void note_amax_written(bool written);
// ... and a launcher that filled srk_epilogue.bn_partial says how many rows (srk_last_conv_bn_partial_rows()).
void note_bn_partial_rows(int rows);

// Environment switches.  env_int / env_str read a variable ONCE per process and answer from a table
// afterwards -- a dispatch must not pay getenv's environment scan -- unless SRK_ENV_LIVE is set when the library is first
// used: the test-suite flips switches between calls of one process (tests/conftest.py sets it).
const char* env_str(const char* name);
int env_int(const char* name, int dflt);
// Experiment switches (ablation bits, alternative block shapes, forced tiles): compiled into the library only with
// -DSRK_EXPERIMENTS (SRK_BUILD_EXPERIMENTS=1 python -m ..._build --force, which tools/ab.sh and the ablation tools need);
// the release library carries the defaults as constants.
// SRK_KDBG(x): a kernel's ablation word (a launch parameter) -- a COMPILE-TIME 0 in the release library.  Not only dead
// code: a run-time `if (dbg & 4)` around a kernel's MFMA loop and its stores gives the compiler's s_waitcnt pass a path on
// which the stores were never issued, and the waits for loads issued before them then assume the worst (k_conv_rowsw:
// s_waitcnt vmcnt(7..0) instead of vmcnt(15..8) in front of the staging commit = every stage waited for the write
// acknowledgements of its own eight stores).
#ifdef SRK_EXPERIMENTS
#define SRK_EXP_INT(name, dflt) (::srk::env_int(name, dflt))
#define SRK_EXP_STR(name) (::srk::env_str(name))
#define SRK_KDBG(x) (x)
#elif defined(SRK_KDBG_CONST)   // ablation builds that keep the release code shape: tools/build_variant.sh ... -DSRK_KDBG_CONST=4
#define SRK_EXP_INT(name, dflt) (dflt)
#define SRK_EXP_STR(name) (static_cast<const char*>(nullptr))
#define SRK_KDBG(x) (SRK_KDBG_CONST)
#else
#define SRK_EXP_INT(name, dflt) (dflt)
#define SRK_EXP_STR(name) (static_cast<const char*>(nullptr))
#define SRK_KDBG(x) 0
#endif

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return SRK_ERR_LAUNCH;
  }
  return SRK_OK;
}

#define SRK_REQUIRE(cond, ...)          \
  do {                                  \
    if (!(cond)) {                      \
      ::srk::set_error(__VA_ARGS__);    \
      return SRK_ERR_BAD_ARG;           \
    }                                   \
  } while (0)

constexpr int kWave = 64;
// Compute units of the current device (MI355X: 256), asked once per device: grids of the persistent kernels and the
// small- / large-problem thresholds follow the part the library runs on instead of a compile-time constant.
inline int num_cu() {
  static std::atomic<int> cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int n = cached[dev].load(std::memory_order_relaxed);
  if (n > 0) return n;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    n = 256;
  }
  cached[dev].store(n, std::memory_order_relaxed);
  return n;
}
#define kNumCU (::srk::num_cu())
constexpr int kMaxDynLds = 160 * 1024;  // gfx950: 160 KB of LDS per CU, all of it usable by one workgroup

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE property of a kernel.  Every launcher that needs more than
// the 64 KB default owns one `static LdsLimit` and calls ensure() before launching: the limit is raised to the hardware
// maximum once per (kernel, device).  Thread-safe (relaxed atomics; a race only repeats an idempotent host call), so the
// main and the autograd threads — or several devices driven from one process — can launch concurrently.
struct LdsLimit {
  static constexpr int kMaxDevices = 64;
  std::atomic<unsigned char> done[kMaxDevices];
  LdsLimit() {
    for (int i = 0; i < kMaxDevices; ++i) done[i].store(0, std::memory_order_relaxed);
  }
  void ensure(const void* fn, size_t lds) {
    if (lds <= 48 * 1024) return;  // under the default limit: nothing to raise
    int dev = -1;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < kMaxDevices && done[dev].load(std::memory_order_relaxed)) return;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynLds) != hipSuccess) {
      // (a kernel that also has static LDS cannot take the full 160 KB as dynamic: ask for what this launch needs)
      (void)hipGetLastError();
      (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      return;
    }
    if (dev >= 0 && dev < kMaxDevices) done[dev].store(1, std::memory_order_relaxed);
  }
};

inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// XCD-aware tile index: hardware deals consecutive block ids round-robin over the 8 XCDs (each with its own L2); this maps
// block `bx` of `nb` to a tile index such that every XCD walks a CONTIGUOUS range of tiles -- neighbouring tiles share
// their halo in ONE L2 instead of fetching it once per L2.  A bijection on [0, nb) for any nb.
__device__ __forceinline__ int xcd_tile_index(int bx, int nb) {
  const int per = nb >> 3, rem = nb & 7;
  const int xcd = bx & 7, idx = bx >> 3;
  return xcd < rem ? xcd * (per + 1) + idx : rem * (per + 1) + (xcd - rem) * per + idx;
}

// Activation forward on a scalar. `a` is the negative-side slope for PReLU / LeakyReLU.
__device__ __forceinline__ float act_apply(float v, int act, float a) {
  switch (act) {
    case SRK_ACT_RELU: return v > 0.f ? v : 0.f;
    case SRK_ACT_PRELU:
    case SRK_ACT_LRELU: return v > 0.f ? v : a * v;
    case SRK_ACT_TANH: return tanhf(v);
    case SRK_ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
    default: return v;
  }
}

// Sum over the 64 lanes of a wave; every lane gets the total.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum for blockDim.x == 256 (4 waves). `sm` needs 4 floats. All threads get the total.
__device__ __forceinline__ float block_sum_256(float v, float* sm) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}
__device__ __forceinline__ double block_sum_256_d(double v, double* sm) {
  v = wave_sum_d(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}

}  // namespace srk

