// extern "C" entry points of the convolution family + library queries.  Validates arguments,
// maps Conv2d / ConvTranspose2d forward / data-gradient onto the gather problem, dispatches to
// the MFMA, direct or generic kernels.
#include "srk_common.h"
#include "conv_problem.h"
#include <stddef.h>
#include <string.h>
#include <stdlib.h>

namespace srk {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static thread_local char g_kernel[128] = "";

void note_kernel(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
  va_end(ap);
}

// ---- environment switches: one getenv per variable and process (see srk_common.h) ----
namespace {
struct EnvSlot { const char* name; const char* val; };
constexpr int kEnvSlots = 64;
EnvSlot g_env[kEnvSlots];
std::atomic<int> g_env_n{0};
std::atomic<int> g_env_lock{0};
}  // namespace

const char* env_str(const char* name) {
  static const bool live = getenv("SRK_ENV_LIVE") != nullptr;
  if (live) return getenv(name);
  const int n = g_env_n.load(std::memory_order_acquire);
  for (int i = 0; i < n; ++i)
    if (g_env[i].name == name || !strcmp(g_env[i].name, name)) return g_env[i].val;
  while (g_env_lock.exchange(1, std::memory_order_acquire)) {}
  const int m = g_env_n.load(std::memory_order_relaxed);
  const char* v = nullptr;
  bool found = false;
  for (int i = 0; i < m && !found; ++i)
    if (!strcmp(g_env[i].name, name)) { v = g_env[i].val; found = true; }
  if (!found) {
    const char* e = getenv(name);
    if (m < kEnvSlots) {  // (both strings are copied: the caller's `name` need not be a literal)
      v = e ? strdup(e) : nullptr;
      g_env[m] = EnvSlot{strdup(name), v};
      g_env_n.store(m + 1, std::memory_order_release);
    } else {
      v = e;  // table full: answer from the environment itself, nothing is allocated
    }
  }
  g_env_lock.store(0, std::memory_order_release);
  return v;
}

int env_int(const char* name, int dflt) {
  const char* e = env_str(name);
  return e ? atoi(e) : dflt;
}

static thread_local int g_amax_written = 0;

void note_amax_written(bool written) { g_amax_written = written ? 1 : 0; }
static thread_local int g_bn_partial_rows = 0;
void note_bn_partial_rows(int rows) { g_bn_partial_rows = rows; }

// conv_rown.hip
bool conv_rown_gather_supported(const GatherConv& g, const float* in, const float* mask_y);
int conv_rown_gather(const GatherConv& g, const float* in, const float* wp, float* out, const Epi& ep, bool x6,
                     hipStream_t s);
// conv_tapn.hip
bool conv_tapn_gather_supported(const GatherConv& g, const float* in, const float* mask_y);
int conv_tapn_gather(const GatherConv& g, const float* in, const float* wp, float* out, const Epi& ep, bool x6,
                     hipStream_t s, const float* mask_y = nullptr, float mask_slope = 0.f);
// conv_c64.hip
bool conv_c64_applicable(const GatherConv& g, const Epi& ep, const float* in, const float* out, const float* mask_y);
int conv_c64_gather(const GatherConv& g, const float* in, const float* wp, float* out, const Epi& ep, int planes,
                    hipStream_t s);
// conv_bfw.hip
bool conv_bfw_applicable(const GatherConv& g, const Epi& ep, const float* in, const float* out, const float* mask_y);
int conv_bfw_gather(const GatherConv& g, const float* in, const float* wp, float* out, const Epi& ep, const float* mask_y,
                    float mask_slope, hipStream_t s, bool f16);
bool conv_rowsw_applicable(const GatherConv& g, const Epi& ep, const float* in, const float* out, const float* mask_y);
int conv_rowsw_gather(const GatherConv& g, const float* in, const float* wp, float* out, const Epi& ep, hipStream_t s, bool f16);
bool conv_bf3_rows_f16_supported(const GatherConv& g, const Epi& ep, const float* out);
int conv_bf3_rows_f16_gather(const GatherConv& g, const float* in, const float* wp, float* out, const Epi& ep, hipStream_t s);
bool conv_tapk_gather_supported(const GatherConv& g, const Epi& ep, const float* out, const float* mask_y);
int conv_tapk_gather(const GatherConv& g, const float* in, const float* wp, float* out, const Epi& ep, hipStream_t s);
bool conv_tapkm_gather_supported(const GatherConv& g, const Epi& ep, const float* out, const float* mask_y);
int conv_tapkm_gather(const GatherConv& g, const float* in, const float* wp, float* out, const Epi& ep, hipStream_t s);
bool conv_wgrad_tapn_supported(const srk_conv_desc& d, const float* x, const srk_bwd_mask* mask);
size_t conv_wgrad_tapn_ws(const srk_conv_desc& d);
int conv_wgrad_tapn(const srk_conv_desc& d, const float* x, const float* dy, float* dw, float* db, float beta, void* ws,
                    size_t ws_bytes, hipStream_t s);
// conv_direct.hip
bool conv_direct_gather_supported(const GatherConv& g, const Epi& ep);
int conv_direct_gather(const GatherConv& g, const float* in, const float* wp, float* out, const Epi& ep,
                       const float* mask_y, float mask_slope, hipStream_t s);
// conv_generic.hip
size_t conv_generic_wgrad_ws(const srk_conv_desc& d);
int conv_generic_wgrad(const srk_conv_desc& d, const float* x, const float* dy, const srk_bwd_mask* mask, float* dw,
                       float* db, float beta, void* ws, size_t ws_bytes, hipStream_t s);
// conv_wgrad_mfma.hip
bool wgrad_reduce_deferring();
int wgrad_reduce_flush(hipStream_t s);
int wgrad_reduce_before_update(const float* dw, const float* db, hipStream_t s);
bool conv_wgrad_mfma_supported(const srk_conv_desc& d);
size_t conv_wgrad_mfma_ws(const srk_conv_desc& d);
int conv_wgrad_mfma(const srk_conv_desc& d, const float* x, const float* dy, const srk_bwd_mask* mask, float* dw,
                    float* db, float beta, void* ws, size_t ws_bytes, hipStream_t s);

// conv_wgrad_bf16.hip
bool conv_wgrad_s2_supported(const srk_conv_desc& d, const float* x, const float* dy, const srk_bwd_mask* mask);
size_t conv_wgrad_s2_ws(const srk_conv_desc& d);
int conv_wgrad_s2(const srk_conv_desc& d, const float* x, const float* dy, float* dw, float* db, float beta, void* ws,
                  size_t ws_bytes, hipStream_t s);
bool conv_wgrad_bf_supported(const srk_conv_desc& d);
size_t conv_wgrad_bf_ws(const srk_conv_desc& d);
int conv_wgrad_bf(const srk_conv_desc& d, const float* x, const float* dy, const srk_bwd_mask* mask, float* dw,
                  float* db, float beta, void* ws, size_t ws_bytes, hipStream_t s);

size_t conv_wgrad_bf_grouped_ws(const srk_conv_desc& d, int n);
int conv_wgrad_bf_grouped(const srk_conv_desc& d, int n, const float* const* xs, const float* const* dys,
                          const srk_bwd_mask* masks, float* const* dws, float* const* dbs, float beta, void* ws,
                          size_t ws_bytes, hipStream_t s);
constexpr int kMaxWgradGroup = 40;  // = WB_MAXGROUP of conv_wgrad_bf16.hip

// conv_mfma_bf16.hip
int pack_weights_batched(const float* params, void* packed, const long long* table, int n_layers, int blocks,
                         hipStream_t s, bool has_unscratched, const int* fast_blocks, int n_fast_blocks);

static int validate_desc(const srk_conv_desc* d, const char* who) {
  SRK_REQUIRE(d, "%s: null descriptor", who);
  SRK_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "%s: non-positive tensor dims", who);
  SRK_REQUIRE(d->KH > 0 && d->KW > 0 && d->stride > 0 && d->pad >= 0, "%s: bad kernel/stride/pad", who);
  SRK_REQUIRE(d->transposed == 0 || d->transposed == 1, "%s: transposed must be 0/1", who);
  SRK_REQUIRE(d->out_pad >= 0 && (d->out_pad == 0 || (d->transposed && d->out_pad < d->stride)),
              "%s: bad output_padding %d", who, d->out_pad);
  const int oh = srk_conv_out_dim(d->H, d->KH, d->stride, d->pad, d->transposed, d->out_pad);
  const int ow = srk_conv_out_dim(d->W, d->KW, d->stride, d->pad, d->transposed, d->out_pad);
  SRK_REQUIRE(oh > 0 && ow > 0, "%s: empty output (%d x %d)", who, oh, ow);
  SRK_REQUIRE(d->OH == oh && d->OW == ow, "%s: OH/OW (%d,%d) != expected (%d,%d)", who, d->OH, d->OW, oh, ow);
  SRK_REQUIRE(d->algo >= SRK_ALGO_AUTO && d->algo <= SRK_ALGO_MFMA_F16X3, "%s: unknown algo %d", who, d->algo);
  return SRK_OK;
}

static int forced_algo(int algo) {
  if (algo != SRK_ALGO_AUTO) return algo;
  const char* e = env_str("SRK_FORCE_ALGO");  // debugging aid: generic|mfma|direct
  if (!e) return SRK_ALGO_AUTO;
  if (!strcmp(e, "generic")) return SRK_ALGO_GENERIC;
  if (!strcmp(e, "mfma")) return SRK_ALGO_MFMA;
  if (!strcmp(e, "direct")) return SRK_ALGO_DIRECT;
  if (!strcmp(e, "bf16x3")) return SRK_ALGO_MFMA_BF16X3;
  if (!strcmp(e, "bf16x6")) return SRK_ALGO_MFMA_BF16X6;
  return SRK_ALGO_AUTO;
}

// SRK_ALGO_MFMA_F16X3 (forward gathers only): what the fp16 kernels cover -- the shapes of conv_bfd.hip, minus the
// layers the few-output-channel kernels take and inputs in a foreign layout
static bool f16x3_gather_ok(const GatherConv& g, const Epi& ep, const float* in, const float* out) {
  if (!ep.x_amax || g.in_ps_r > 1) return false;
  if (conv_bf3_rows_f16_supported(g, ep, out)) return true;   // first layers (Cin <= 4), also on an NCHW input in place
  if (g.in_nchw) return false;
  if (!conv_bfd_gather_supported(g, ep) || !conv_epi_all_vector(g.OC, ep, out)) return false;
  if (conv_direct_gather_supported(g, ep) || conv_tapn_gather_supported(g, in, nullptr)) return false;
  return true;
}

static int run_gather(const GatherConv& g, int algo, const float* in, const float* wp, float* out, const Epi& ep,
                      const float* mask_y, float mask_slope, hipStream_t s, const char* who) {
  algo = forced_algo(algo);
  if (g.in_ps_r > 1) {  // pixel-shuffled input: only the bf16x3 kernels un-shuffle while staging
    if ((algo != SRK_ALGO_AUTO && algo != SRK_ALGO_MFMA_BF16X3) || !conv_bf3_gather_supported(g, ep) || g.IC <= 4 ||
        g.OC <= 4) {
      set_error("%s: pixel-shuffled dy is only supported by the bf16x3 kernels (un-shuffle with srk_pixel_shuffle_backward)", who);
      return SRK_ERR_UNSUPPORTED;
    }
    if (conv_bfd_gather_supported(g, ep) && conv_epi_all_vector(g.OC, ep, out) && conv_bfd_small_problem(g))
      return conv_bfd_gather(g, in, wp, out, ep, mask_y, mask_slope, 2, s);
    return conv_bf3_gather(g, in, wp, out, ep, mask_y, mask_slope, s);
  }
  if (algo == SRK_ALGO_MFMA_F16X3) {
    if (mask_y || !f16x3_gather_ok(g, ep, in, out)) {
      set_error("%s: SRK_ALGO_MFMA_F16X3 covers forward convs with Cin, Cout >= 8 on the 16-byte store path and needs "
                "srk_epilogue.x_amax (srk_conv2d_f16x3_supported)", who);
      return SRK_ERR_UNSUPPORTED;
    }
    if (conv_bf3_rows_f16_supported(g, ep, out)) {
      if (conv_rowsw_applicable(g, ep, in, out, nullptr)) {  // persistent wave-specialised form (benchmark-size first layers)
        const int rc = conv_rowsw_gather(g, in, wp, out, ep, s, true);
        if (rc >= 0) return rc;
      }
      return conv_bf3_rows_f16_gather(g, in, wp, out, ep, s);
    }
    if (conv_c64_applicable(g, ep, in, out, nullptr))   // small 64 -> 64 3x3 problems: one 8x8 tile per block
      return conv_c64_gather(g, in, wp, out, ep, 4, s);
    if (conv_bfw_applicable(g, ep, in, out, nullptr)) {  // wave-specialised persistent kernel (ESPCN-size layers)
      const int rc = conv_bfw_gather(g, in, wp, out, ep, nullptr, 0.f, s, true);
      if (rc >= 0) return rc;
    }
    return conv_bfd_gather(g, in, wp, out, ep, nullptr, 0.f, 4, s);
  }
  const bool direct_ok = conv_direct_gather_supported(g, ep);
  const bool mfma_ok = conv_mfma_gather_supported(g, ep);
  if (algo == SRK_ALGO_DIRECT && !direct_ok) {
    set_error("%s: direct kernel does not cover this shape", who);
    return SRK_ERR_UNSUPPORTED;
  }
  if (algo == SRK_ALGO_MFMA && !mfma_ok && !direct_ok) {
    set_error("%s: MFMA kernel does not cover this shape", who);
    return SRK_ERR_UNSUPPORTED;
  }
  // few-output-channel convs with more taps than one 32-column group holds (9x9 / 5x5 x 3 channels): kernel rows on N,
  // kernel columns in K, one GEMM per input row
  if ((algo == SRK_ALGO_AUTO || algo == SRK_ALGO_MFMA_BF16X3 || algo == SRK_ALGO_MFMA_BF16X6) &&
      conv_rown_gather_supported(g, in, mask_y))
    return conv_rown_gather(g, in, wp, out, ep, algo == SRK_ALGO_MFMA_BF16X6, s);
  // few-output-channel 3x3 convs (the 64 -> 3 reconstruction layers): taps-as-N bf16x6 kernel for every bf16 class
  if ((algo == SRK_ALGO_AUTO || algo == SRK_ALGO_MFMA_BF16X3 || algo == SRK_ALGO_MFMA_BF16X6) &&
      conv_tapn_gather_supported(g, in, mask_y))
    return conv_tapn_gather(g, in, wp, out, ep, algo == SRK_ALGO_MFMA_BF16X6, s, mask_y, mask_slope);
  // ... and their data gradients (<= 3 input channels, TRANS gather): taps-in-K bf16x3 kernel
  if ((algo == SRK_ALGO_AUTO || algo == SRK_ALGO_MFMA_BF16X3) && conv_tapk_gather_supported(g, ep, out, mask_y))
    return conv_tapk_gather(g, in, wp, out, ep, s);
  // ... with more taps than one 32-slot K step holds (9x9 x 3 channels): one kernel row per step
  if ((algo == SRK_ALGO_AUTO || algo == SRK_ALGO_MFMA_BF16X3) && conv_tapkm_gather_supported(g, ep, out, mask_y))
    return conv_tapkm_gather(g, in, wp, out, ep, s);
  // the bfd kernels are compiled without the scalar store fallback
  const bool bfd_ok = conv_bfd_gather_supported(g, ep) && conv_epi_all_vector(g.OC, ep, out);
  if (algo == SRK_ALGO_MFMA_BF16X6) {  // fp32-faithful class: bf16x6 where it applies, else the exact fp32 kernels
    if (conv_c64_applicable(g, ep, in, out, mask_y)) return conv_c64_gather(g, in, wp, out, ep, 3, s);
    if (bfd_ok && !direct_ok) return conv_bfd_gather(g, in, wp, out, ep, mask_y, mask_slope, 3, s);
    algo = SRK_ALGO_MFMA;
  }
  const bool bf3_ok = conv_bf3_gather_supported(g, ep);
  if (algo == SRK_ALGO_MFMA_BF16X3 && !bf3_ok) {
    set_error("%s: bf16x3 MFMA kernel does not cover this shape", who);
    return SRK_ERR_UNSUPPORTED;
  }
  if (algo == SRK_ALGO_DIRECT || ((algo == SRK_ALGO_AUTO || (algo == SRK_ALGO_MFMA && !mfma_ok)) && direct_ok))
    return conv_direct_gather(g, in, wp, out, ep, mask_y, mask_slope, s);
  if (algo == SRK_ALGO_MFMA_BF16X3 || (algo == SRK_ALGO_AUTO && bf3_ok)) {
    // filters-from-global variant: small problems (channel-split 64-pixel blocks), or on request
    const char* e = env_str("SRK_BF3_DIRECT");  // 0 = never, 1 = always, unset = small problems only
    const int direct_w = e ? (atoi(e) ? 1 : 0) : 2;
    // small 64 -> 64 3x3 problems without activation / mask (SRGAN's BatchNorm-separated convs and their data gradients)
    if (direct_w != 0 && conv_c64_applicable(g, ep, in, out, mask_y)) return conv_c64_gather(g, in, wp, out, ep, 2, s);
    if (bfd_ok && (direct_w == 1 || (direct_w == 2 && conv_bfd_small_problem(g))))
      return conv_bfd_gather(g, in, wp, out, ep, mask_y, mask_slope, 2, s);
    if (conv_bfw_applicable(g, ep, in, out, mask_y)) {  // wave-specialised persistent kernel (ESPCN-size layers)
      const int rc = conv_bfw_gather(g, in, wp, out, ep, mask_y, mask_slope, s, false);
      if (rc >= 0) return rc;
    }
    if (conv_rowsw_applicable(g, ep, in, out, mask_y)) {  // ... and its row-packed form for first layers (Cin <= 4)
      const int rc = conv_rowsw_gather(g, in, wp, out, ep, s, false);
      if (rc >= 0) return rc;
    }
    return conv_bf3_gather(g, in, wp, out, ep, mask_y, mask_slope, s);
  }
  if (algo == SRK_ALGO_MFMA || (algo == SRK_ALGO_AUTO && mfma_ok))
    return conv_mfma_gather(g, in, wp, out, ep, mask_y, mask_slope, s);
  return conv_generic_gather(g, in, wp, out, ep, mask_y, mask_slope, s);
}

}  // namespace srk

using namespace srk;

extern "C" int srk_version(void) { return SRK_VERSION; }

extern "C" const char* srk_status_string(int status) {
  switch (status) {
    case SRK_OK: return "ok";
    case SRK_ERR_BAD_ARG: return "bad argument";
    case SRK_ERR_UNSUPPORTED: return "unsupported configuration";
    case SRK_ERR_LAUNCH: return "kernel launch failure";
    case SRK_ERR_WORKSPACE: return "workspace too small";
    default: return "unknown status";
  }
}

extern "C" const char* srk_last_error_string(void) { return g_err; }

extern "C" const char* srk_last_kernel_name(void) { return g_kernel; }

extern "C" int srk_last_conv_wrote_amax(void) { return g_amax_written; }
extern "C" int srk_last_conv_bn_partial_rows(void) { return g_bn_partial_rows; }

extern "C" int srk_conv_out_dim(int in, int k, int stride, int pad, int transposed, int out_pad) {
  if (in <= 0 || k <= 0 || stride <= 0 || pad < 0) return -1;
  if (!transposed) return (in + 2 * pad - k) / stride + 1;
  return (in - 1) * stride - 2 * pad + k + out_pad;
}

extern "C" int srk_conv2d_forward(const srk_conv_desc* d, const float* x, const float* w_packed_fwd, float* y,
                                  const srk_epilogue* ep_in, void* stream) {
  note_amax_written(false);
  note_bn_partial_rows(0);
  int rc = validate_desc(d, "conv2d_forward");
  if (rc) return rc;
  SRK_REQUIRE(x && w_packed_fwd && y, "conv2d_forward: null tensor pointer");
  Epi ep = make_epi(ep_in);
  SRK_REQUIRE(ep.act >= SRK_ACT_NONE && ep.act <= SRK_ACT_SIGMOID, "conv2d_forward: unknown act %d", ep.act);
  SRK_REQUIRE(ep.act != SRK_ACT_PRELU || (ep.prelu_w && ep.prelu_n >= 1), "conv2d_forward: PReLU needs its weight");
  if (ep.ps_r > 1) {
    SRK_REQUIRE(d->Cout % (ep.ps_r * ep.ps_r) == 0, "conv2d_forward: Cout %d not divisible by r^2", d->Cout);
    SRK_REQUIRE(ep.act != SRK_ACT_PRELU || ep.prelu_n == 1 || ep.prelu_n == d->Cout / (ep.ps_r * ep.ps_r),
                "conv2d_forward: PReLU after pixel-shuffle must have 1 or Cout/r^2 slopes");
  } else {
    SRK_REQUIRE(ep.act != SRK_ACT_PRELU || ep.prelu_n == 1 || ep.prelu_n == d->Cout,
                "conv2d_forward: PReLU must have 1 or Cout slopes");
  }
  GatherConv g{d->N, d->H, d->W, d->Cin, d->OH, d->OW, d->Cout, d->KH, d->KW, d->stride, d->pad, d->transposed, 0};
  SRK_REQUIRE(d->dy_ps_r == 0, "conv2d_forward: dy_ps_r is a backward-only field");
  if (d->x_nchw) {
    // only the row-packed bf16x3 first-layer kernel reads an NCHW input in place
    const int algo = forced_algo(d->algo);
    if (!(d->Cin <= 4 && !d->transposed && d->Cout >= 8 &&
          (algo == SRK_ALGO_AUTO || algo == SRK_ALGO_MFMA_BF16X3 || algo == SRK_ALGO_MFMA_F16X3) &&
          conv_bf3_gather_supported(g, ep))) {
      set_error("conv2d_forward: x_nchw is only supported by the Cin <= 4 bf16x3 kernel");
      return SRK_ERR_UNSUPPORTED;
    }
    g.in_nchw = 1;
  }
  rc = run_gather(g, d->algo, x, w_packed_fwd, y, ep, nullptr, 0.f, (hipStream_t)stream, "conv2d_forward");
  return rc;
}

extern "C" int srk_conv2d_forward_ex(const srk_conv_desc* d, const float* x, const float* w_packed_fwd, float* y,
                                     const srk_epilogue* ep_in, srk_conv_result* res, void* stream) {
  const int rc = srk_conv2d_forward(d, x, w_packed_fwd, y, ep_in, stream);
  if (res) {
    // (the caller says how much of the struct it knows: only those fields are written)
    if (res->struct_size >= offsetof(srk_conv_result, wrote_amax) + sizeof(int32_t)) res->wrote_amax = rc == SRK_OK ? g_amax_written : 0;
    if (res->struct_size >= offsetof(srk_conv_result, bn_partial_rows) + sizeof(int32_t))
      res->bn_partial_rows = rc == SRK_OK ? g_bn_partial_rows : 0;
  }
  return rc;
}

extern "C" int srk_conv2d_f16x3_supported(const srk_conv_desc* d, const srk_epilogue* ep_in, const float* y) {
  if (!d || validate_desc(d, "conv2d_f16x3_supported") || d->dy_ps_r) return 0;
  Epi ep = make_epi(ep_in);
  static const float dummy = 0.f;
  if (!ep.x_amax) ep.x_amax = &dummy;   // (the question is about the shape; the call itself needs the real slots)
  GatherConv g{d->N, d->H, d->W, d->Cin, d->OH, d->OW, d->Cout, d->KH, d->KW, d->stride, d->pad, d->transposed, 0};
  g.in_nchw = d->x_nchw;
  return f16x3_gather_ok(g, ep, reinterpret_cast<const float*>(16), y ? y : reinterpret_cast<const float*>(16)) ? 1 : 0;
}

extern "C" int srk_conv2d_backward_data(const srk_conv_desc* d, const float* dy, const float* w_packed_bwd, float* dx,
                                        const srk_bwd_mask* mask, const float* add_to, void* stream) {
  int rc = validate_desc(d, "conv2d_backward_data");
  if (rc) return rc;
  SRK_REQUIRE(d->algo != SRK_ALGO_MFMA_F16X3, "conv2d_backward_data: SRK_ALGO_MFMA_F16X3 is a forward-only arithmetic "
              "(the backward kernels run bf16x3)");
  SRK_REQUIRE(dy && w_packed_bwd && dx, "conv2d_backward_data: null tensor pointer");
  // dx is a gather over dy with the channel roles swapped and the opposite gather kind.
  GatherConv g{d->N, d->OH, d->OW, d->Cout, d->H, d->W, d->Cin, d->KH, d->KW, d->stride, d->pad, !d->transposed, 0, 0};
  if (d->dy_ps_r > 1) {
    const int r2 = d->dy_ps_r * d->dy_ps_r;
    SRK_REQUIRE(d->Cout % r2 == 0 && (d->Cout / r2) % 8 == 0, "conv2d_backward_data: dy_ps_r needs Cout/r^2 to be a multiple of 8");
    g.in_ps_r = d->dy_ps_r;
  }
  Epi ep{};
  ep.residual = add_to;
  return run_gather(g, d->algo, dy, w_packed_bwd, dx, ep, mask ? mask->y : nullptr, mask ? mask->slope : 0.f,
                    (hipStream_t)stream, "conv2d_backward_data");
}

static GatherConv bwd_data_gather(const srk_conv_desc* d) {
  return GatherConv{d->N, d->OH, d->OW, d->Cout, d->H, d->W, d->Cin, d->KH, d->KW, d->stride, d->pad, !d->transposed, 0, 0};
}

extern "C" int srk_conv2d_backward_data_relu_supported(const srk_conv_desc* d, const float* dy, const float* dx,
                                                       const srk_bwd_mask* mask) {
  if (!d || validate_desc(d, "conv2d_backward_data_relu_supported") || d->dy_ps_r > 1) return 0;
  const int algo = forced_algo(d->algo);
  if (algo != SRK_ALGO_AUTO && algo != SRK_ALGO_MFMA_BF16X3) return 0;
  const GatherConv g = bwd_data_gather(d);
  Epi ep{};
  ep.out_relu = dx ? dx : reinterpret_cast<const float*>(16);   // (any aligned non-null pointer: only its presence matters)
  const float* a16 = reinterpret_cast<const float*>(16);
  return conv_bfw_applicable(g, ep, dy ? dy : a16, dx ? dx : a16, mask ? mask->y : nullptr) ? 1 : 0;
}

extern "C" int srk_conv2d_backward_data_relu(const srk_conv_desc* d, const float* dy, const float* w_packed_bwd, float* dx,
                                             const srk_bwd_mask* mask, const float* x_relu, void* stream) {
  int rc = validate_desc(d, "conv2d_backward_data_relu");
  if (rc) return rc;
  SRK_REQUIRE(dy && w_packed_bwd && dx && x_relu, "conv2d_backward_data_relu: null tensor pointer");
  SRK_REQUIRE((uintptr_t)x_relu % 16 == 0, "conv2d_backward_data_relu: x_relu must be 16-byte aligned");
  if (!srk_conv2d_backward_data_relu_supported(d, dy, dx, mask)) {
    set_error("conv2d_backward_data_relu: only where the wave-specialised kernel applies (srk_conv2d_backward_data_relu_supported)");
    return SRK_ERR_UNSUPPORTED;
  }
  const GatherConv g = bwd_data_gather(d);
  Epi ep{};
  ep.out_relu = x_relu;
  rc = conv_bfw_gather(g, dy, w_packed_bwd, dx, ep, mask ? mask->y : nullptr, mask ? mask->slope : 0.f, (hipStream_t)stream, false);
  if (rc < 0) {
    set_error("conv2d_backward_data_relu: no tile of the wave-specialised kernel fits this problem");
    return SRK_ERR_UNSUPPORTED;
  }
  return rc;
}

extern "C" int srk_resblock2_supported(int N, int H, int W, int C) { return conv_res2_supported(N, H, W, C) ? 1 : 0; }

static int resblock2_planes(int algo, const char* who) {
  algo = forced_algo(algo);
  if (algo == SRK_ALGO_MFMA_BF16X6) return 3;
  if (algo == SRK_ALGO_MFMA_F16X3) return 4;
  if (algo == SRK_ALGO_AUTO || algo == SRK_ALGO_MFMA_BF16X3) return 2;
  set_error("%s: only the bf16x3 (SRK_ALGO_AUTO) and bf16x6 arithmetic exist for the fused block", who);
  return 0;
}

extern "C" int srk_resblock2_forward(int N, int H, int W, int C, const float* x, const float* w1_packed_fwd,
                                     const float* b1, const float* w2_packed_fwd, const float* b2, float* y_mid, float* y,
                                     int algo, const float* x_amax, float* y_amax, void* stream) {
  SRK_REQUIRE(x && w1_packed_fwd && w2_packed_fwd && y_mid && y, "resblock2_forward: null tensor pointer");
  SRK_REQUIRE(conv_res2_supported(N, H, W, C), "resblock2_forward: unsupported problem (see srk_resblock2_supported)");
  SRK_REQUIRE(((uintptr_t)x | (uintptr_t)y_mid | (uintptr_t)y | (uintptr_t)b1 | (uintptr_t)b2) % 16 == 0,
              "resblock2_forward: tensors must be 16-byte aligned");
  const int planes = resblock2_planes(algo, "resblock2_forward");
  if (!planes) return SRK_ERR_UNSUPPORTED;
  SRK_REQUIRE(planes != 4 || x_amax, "resblock2_forward: SRK_ALGO_MFMA_F16X3 needs x_amax");
  return conv_res2(x, w1_packed_fwd, w2_packed_fwd, b1, b2, nullptr, y_mid, y, N, H, W, planes, false,
                   (hipStream_t)stream, x_amax, y_amax);
}

extern "C" int srk_resblock2_backward_data(int N, int H, int W, int C, const float* dy, const float* w2_packed_bwd,
                                           const float* w1_packed_bwd, const float* y_mid, float* d_mid, float* dx,
                                           int algo, void* stream) {
  SRK_REQUIRE(dy && w2_packed_bwd && w1_packed_bwd && y_mid && d_mid && dx, "resblock2_backward_data: null tensor pointer");
  SRK_REQUIRE(conv_res2_supported(N, H, W, C), "resblock2_backward_data: unsupported problem (see srk_resblock2_supported)");
  SRK_REQUIRE(((uintptr_t)dy | (uintptr_t)y_mid | (uintptr_t)d_mid | (uintptr_t)dx) % 16 == 0,
              "resblock2_backward_data: tensors must be 16-byte aligned");
  const int planes = resblock2_planes(algo, "resblock2_backward_data");
  if (!planes) return SRK_ERR_UNSUPPORTED;
  SRK_REQUIRE(planes != 4, "resblock2_backward_data: SRK_ALGO_MFMA_F16X3 is forward-only");
  // in this direction conv2's transposed filter runs first, conv1's second
  return conv_res2(dy, w2_packed_bwd, w1_packed_bwd, nullptr, nullptr, y_mid, d_mid, dx, N, H, W, planes, true,
                   (hipStream_t)stream);
}

extern "C" size_t srk_conv2d_backward_weight_workspace_bytes(const srk_conv_desc* d) {
  if (!d) return 0;
  size_t a = conv_generic_wgrad_ws(*d);
  size_t b = conv_wgrad_mfma_supported(*d) ? conv_wgrad_mfma_ws(*d) : 0;
  size_t c = conv_wgrad_bf_supported(*d) ? conv_wgrad_bf_ws(*d) : 0;
  if (b > a) a = b;
  if (c > a) a = c;
  const size_t t = conv_wgrad_tapn_supported(*d, nullptr, nullptr) ? conv_wgrad_tapn_ws(*d) : 0;
  const size_t s2 = conv_wgrad_s2_ws(*d);   // (0 when the stride-2 bf16x3 kernel does not cover the layer)
  if (s2 > a) a = s2;
  return a > t ? a : t;
}

extern "C" int srk_conv2d_backward_weight(const srk_conv_desc* d, const float* x, const float* dy,
                                          const srk_bwd_mask* mask, float* dw, float* db, float beta, void* workspace,
                                          size_t workspace_bytes, void* stream) {
  int rc = validate_desc(d, "conv2d_backward_weight");
  if (rc) return rc;
  SRK_REQUIRE(x && dy && dw, "conv2d_backward_weight: null tensor pointer");
  SRK_REQUIRE(beta == 0.f || beta == 1.f, "conv2d_backward_weight: beta must be 0 or 1");
  if ((rc = wgrad_reduce_before_update(dw, db, (hipStream_t)stream))) return rc;   // (deferred reductions: srk_wgrad_reduce_defer)
  // AUTO / BF16X3: bf16x3 MFMA kernel where it applies (stride-1 convs up to 3x3); MFMA / BF16X6 / DIRECT: the
  // exact fp32 MFMA kernel; GENERIC: the plain kernel
  const int algo = forced_algo(d->algo);
  const char* wb = env_str("SRK_WGRAD_BF16");  // 0 disables the bf16x3 weight-gradient kernels
  if ((algo == SRK_ALGO_AUTO || algo == SRK_ALGO_MFMA_BF16X3) && !(wb && atoi(wb) == 0) &&
      conv_wgrad_tapn_supported(*d, x, mask))  // few-output-channel reconstruction convs
    return conv_wgrad_tapn(*d, x, dy, dw, db, beta, workspace, workspace_bytes, (hipStream_t)stream);
  if ((algo == SRK_ALGO_AUTO || algo == SRK_ALGO_MFMA_BF16X3) && !(wb && atoi(wb) == 0) && conv_wgrad_bf_supported(*d))
    return conv_wgrad_bf(*d, x, dy, mask, dw, db, beta, workspace, workspace_bytes, (hipStream_t)stream);
  // stride-2 3x3 convs (SRGAN's discriminator): bf16x3 on de-interleaved halo columns
  if ((algo == SRK_ALGO_AUTO || algo == SRK_ALGO_MFMA_BF16X3) && !(wb && atoi(wb) == 0) && conv_wgrad_s2_supported(*d, x, dy, mask))
    return conv_wgrad_s2(*d, x, dy, dw, db, beta, workspace, workspace_bytes, (hipStream_t)stream);
  if (d->dy_ps_r > 1) {
    set_error("conv2d_backward_weight: pixel-shuffled dy is only supported by the bf16x3 kernel (un-shuffle with "
              "srk_pixel_shuffle_backward)");
    return SRK_ERR_UNSUPPORTED;
  }
  if (algo != SRK_ALGO_GENERIC && conv_wgrad_mfma_supported(*d))
    return conv_wgrad_mfma(*d, x, dy, mask, dw, db, beta, workspace, workspace_bytes, (hipStream_t)stream);
  return conv_generic_wgrad(*d, x, dy, mask, dw, db, beta, workspace, workspace_bytes, (hipStream_t)stream);
}

static bool wgrad_group_uses_bf(const srk_conv_desc& d) {
  const int algo = forced_algo(d.algo);
  const char* wb = env_str("SRK_WGRAD_BF16");
  const char* gg = env_str("SRK_WGRAD_GROUPED");  // 0: grouped calls run layer by layer (A/B against the per-layer kernels)
  return (algo == SRK_ALGO_AUTO || algo == SRK_ALGO_MFMA_BF16X3) && !(wb && atoi(wb) == 0) && !(gg && atoi(gg) == 0) &&
         d.dy_ps_r == 0 && !conv_wgrad_tapn_supported(d, nullptr, nullptr) && conv_wgrad_bf_supported(d);
}

extern "C" size_t srk_conv2d_backward_weight_grouped_workspace_bytes(const srk_conv_desc* d, int n) {
  if (!d || n < 1) return 0;
  size_t a = srk_conv2d_backward_weight_workspace_bytes(d);
  if (wgrad_group_uses_bf(*d)) {
    // the call runs chunks of kMaxWgradGroup layers and one of n % kMaxWgradGroup; the slab count of a chunk is not
    // monotonic in its layer count (G = blocks / layers is rounded), so size for every chunk length the call will use
    const int full = n < kMaxWgradGroup ? n : kMaxWgradGroup;
    const int rest = n > kMaxWgradGroup ? n % kMaxWgradGroup : 0;
    size_t b = conv_wgrad_bf_grouped_ws(*d, full);
    if (rest) {
      const size_t c = conv_wgrad_bf_grouped_ws(*d, rest);
      if (c > b) b = c;
    }
    if (b > a) a = b;
  }
  return a;
}

extern "C" int srk_conv2d_backward_weight_grouped(const srk_conv_desc* d, int n, const float* const* x,
                                                  const float* const* dy, const srk_bwd_mask* masks, float* const* dw,
                                                  float* const* db, float beta, void* workspace, size_t workspace_bytes,
                                                  void* stream) {
  int rc = validate_desc(d, "conv2d_backward_weight_grouped");
  if (rc) return rc;
  SRK_REQUIRE(n >= 1 && x && dy && dw, "conv2d_backward_weight_grouped: empty group or null pointer array");
  SRK_REQUIRE(beta == 0.f || beta == 1.f, "conv2d_backward_weight_grouped: beta must be 0 or 1");
  for (int l = 0; l < n; ++l) {
    SRK_REQUIRE(x[l] && dy[l] && dw[l], "conv2d_backward_weight_grouped: null tensor pointer in layer %d", l);
    for (int k = 0; k < l; ++k)
      SRK_REQUIRE(dw[k] != dw[l], "conv2d_backward_weight_grouped: layers %d and %d write the same dw (shared weights must "
                  "go into separate calls)", k, l);
    if ((rc = wgrad_reduce_before_update(dw[l], db ? db[l] : nullptr, (hipStream_t)stream))) return rc;
  }
  if (n >= 2 && wgrad_group_uses_bf(*d)) {
    for (int l0 = 0; l0 < n; l0 += kMaxWgradGroup) {
      const int m = n - l0 < kMaxWgradGroup ? n - l0 : kMaxWgradGroup;
      rc = conv_wgrad_bf_grouped(*d, m, x + l0, dy + l0, masks ? masks + l0 : nullptr, dw + l0, db ? db + l0 : nullptr, beta,
                                 workspace, workspace_bytes, (hipStream_t)stream);
      if (rc) return rc;
      // (deferred reductions read the slabs at the flush: the next chunk writes the same workspace)
      if (l0 + m < n && wgrad_reduce_deferring() && (rc = wgrad_reduce_flush((hipStream_t)stream))) return rc;
    }
    return SRK_OK;
  }
  for (int l = 0; l < n; ++l) {  // geometry without a grouped kernel (or a single layer): the per-layer path, same results
    rc = srk_conv2d_backward_weight(d, x[l], dy[l], (masks && masks[l].y) ? &masks[l] : nullptr, dw[l], db ? db[l] : nullptr,
                                    beta, workspace, workspace_bytes, stream);
    if (rc) return rc;
    if (l + 1 < n && wgrad_reduce_deferring() && (rc = wgrad_reduce_flush((hipStream_t)stream))) return rc;
  }
  return SRK_OK;
}

extern "C" int srk_pack_weights_batched(const float* params_base, void* packed_base, const int64_t* table, int n_layers,
                                        int blocks_per_layer, const int32_t* fast_blocks, int n_fast_blocks, void* stream) {
  SRK_REQUIRE(params_base && packed_base && table, "pack_weights_batched: null pointer");
  SRK_REQUIRE(n_layers > 0 && n_layers <= 65535 && blocks_per_layer != 0, "pack_weights_batched: bad sizes");
  // blocks_per_layer < 0: EVERY row carries a scratch word (column 11 >= 0), no single-block scan needed
  const bool all_scratch = blocks_per_layer < 0;
  return pack_weights_batched(params_base, packed_base, reinterpret_cast<const long long*>(table), n_layers,
                              all_scratch ? -blocks_per_layer : blocks_per_layer, (hipStream_t)stream, !all_scratch,
                              fast_blocks, n_fast_blocks);
}
