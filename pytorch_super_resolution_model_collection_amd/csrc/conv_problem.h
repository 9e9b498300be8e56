// Internal description of one "gather convolution" launch plus the fused epilogue, shared by the
// generic, direct and MFMA kernels.
#pragma once
#include "srk_common.h"

namespace srk {

// out[n,oy,ox,oc] = sum in[n,iy,ix,ic] * W[kh][kw][ic][oc]; see conv_generic.hip for iy(oy,kh).
struct GatherConv {
  int N, IH, IW, IC;
  int OH, OW, OC;
  int KH, KW, stride, pad, trans;
  int in_nchw;  // input is NCHW (only the bf16x3 row-packed kernel reads it in place)
  int in_ps_r;  // input is stored pixel-shuffled [N, IH*r, IW*r, IC/r^2] (data gradient of a fused conv + PS forward)
};

// Device-side epilogue: out = PS_r(act(acc + bias)) + residual
struct Epi {
  const float* bias;
  const float* prelu_w;
  const float* residual;
  float slope;
  int act;
  int prelu_n;
  int ps_r;
  const float* x_amax;  // SRK_AMAX_SLOTS floats (f16x3 kernels)
  float* y_amax;        // optional running max of |out| (kernels that support it)
  const float* out_relu;  // optional: out <- out * (out_relu > 0), a tensor of out's shape (k_conv_bfw<2,9,2> only)
  double* bn_partial;     // optional: per-tile column sums of the output (srk_epilogue.bn_partial; k_c64 forward only)
};

inline Epi make_epi(const srk_epilogue* e) {
  Epi r{};
  if (e) {
    r.bias = e->bias;
    r.prelu_w = e->prelu_weight;
    r.residual = e->residual;
    r.slope = e->slope;
    r.act = e->act;
    r.prelu_n = e->prelu_n;
    r.ps_r = e->ps_r > 1 ? e->ps_r : 0;
    r.x_amax = e->x_amax;
    r.y_amax = e->y_amax;
    r.bn_partial = e->bn_partial;
  }
  return r;
}

// Address of output element (n, oy, ox, oc) after the optional fused pixel shuffle.
// With ps_r > 1 the packed channel order is (i, j, c): oc = (i*r + j)*C + c, and `c_out`
// receives the post-shuffle channel c (the index a per-channel PReLU of PSBlock uses).
__device__ __forceinline__ size_t epi_out_index(int ps_r, int OH, int OW, int OC, int n, int oy, int ox, int oc,
                                                int& c_out) {
  if (ps_r > 1) {
    const int C = OC / (ps_r * ps_r);
    const int q = oc / C, c = oc - q * C;
    const int i = q / ps_r, j = q - i * ps_r;
    c_out = c;
    return (((size_t)n * OH * ps_r + (size_t)oy * ps_r + i) * ((size_t)OW * ps_r) + (size_t)ox * ps_r + j) * C + c;
  }
  c_out = oc;
  return (((size_t)n * OH + oy) * OW + ox) * OC + oc;
}

__device__ __forceinline__ float epi_value(const Epi& ep, float acc, int oc, int c_act) {
  if (ep.bias) acc += ep.bias[oc];
  if (ep.act != SRK_ACT_NONE) {
    float a = ep.slope;
    if (ep.act == SRK_ACT_PRELU) a = ep.prelu_n > 1 ? ep.prelu_w[c_act] : ep.prelu_w[0];
    acc = act_apply(acc, ep.act, a);
  }
  return acc;
}

__device__ __forceinline__ void epi_store(const Epi& ep, const GatherConv& g, float acc, int n, int oy, int ox, int oc,
                                          float* __restrict__ out) {
  int c_act;
  const size_t o = epi_out_index(ep.ps_r, g.OH, g.OW, g.OC, n, oy, ox, oc, c_act);
  float v = epi_value(ep, acc, oc, c_act);
  if (ep.residual) v += ep.residual[o];
  out[o] = v;
}

// Implemented in conv_generic.hip
int conv_generic_gather(const GatherConv& g, const float* in, const float* wp, float* out, const Epi& ep,
                        const float* mask_y, float mask_slope, hipStream_t s);
size_t conv_bias_grad_ws(const srk_conv_desc& d);
int conv_bias_grad(const srk_conv_desc& d, const float* dy, const srk_bwd_mask* mask, float* db, float beta, void* ws,
                   hipStream_t s);
int conv_bias_grad_finish(const float* partial, int splits, float* db, int Cout, float beta, hipStream_t s);
int conv_wgrad_finalize(const srk_conv_desc& d, const float* ws, float* dw, float beta, hipStream_t s);
// Implemented in conv_mfma.hip
bool conv_mfma_gather_supported(const GatherConv& g, const Epi& ep);
int conv_mfma_gather(const GatherConv& g, const float* in, const float* wp, float* out, const Epi& ep,
                     const float* mask_y, float mask_slope, hipStream_t s);

// Implemented in conv_mfma_bf16.hip
size_t bf3_prepared_offset(size_t elems);
size_t bf3_prepared_bytes(int IC, int OC, int T);
size_t bf3_main_bytes(int IC, int OC, int T);
// True when every 4-channel group of the layer's output takes the 16-byte store path of the LDS-staged epilogues
// (mirror of epi_col_setup's test): kernels compiled without the scalar store fallback require it.
static inline bool conv_epi_all_vector(int OC, const Epi& ep, const float* out) {
  if (OC % 4 != 0) return false;
  if (ep.act == SRK_ACT_PRELU && ep.prelu_n > 1) return false;
  if (((uintptr_t)out % 16) != 0 || (ep.bias && ((uintptr_t)ep.bias % 16) != 0) ||
      (ep.residual && ((uintptr_t)ep.residual % 16) != 0))
    return false;
  if (ep.ps_r > 1) {
    const int r = ep.ps_r;
    if (OC % (r * r) != 0) return false;
    if ((r * (OC / (r * r))) % 4 != 0) return false;
  }
  return true;
}

// f16x3 section of a forward packed buffer (conv_mfma_bf16.hip): fp16 planes h, m in the bf16 main layout, then a
// 256-byte trailer {float descale = 2^-kw, float scale = 2^kw}
size_t f16_section_offset(int IC, int OC, int T);  // from the start of the prepared (bf16) section
size_t f16_section_bytes(int IC, int OC, int T);
// conv_bfd.hip: filters straight from global memory; planes 2 = bf16x3, 3 = bf16x6 (fp32-faithful), 4 = f16x3
bool conv_bfd_gather_supported(const GatherConv& g, const Epi& ep);
bool conv_bfd_small_problem(const GatherConv& g);
// conv_res2.hip: both 3x3 64 -> 64 convs of a residual block in one launch per 8x8 tile (small problems)
bool conv_res2_supported(int N, int H, int W, int C);
int conv_res2(const float* in, const float* wp1, const float* wp2, const float* bias1, const float* bias2,
              const float* gate, float* mid, float* out, int N, int H, int W, int planes, bool bwd, hipStream_t s,
              const float* x_amax = nullptr, float* y_amax = nullptr);
int conv_bfd_gather(const GatherConv& g, const float* in, const float* wp, float* out, const Epi& ep,
                    const float* mask_y, float mask_slope, int planes, hipStream_t s);
int bf3_pack_prepared(const float* w, void* packed_base, int Cout, int Cin, int KH, int KW, int transposed, int ps_r,
                      int bwd, hipStream_t s);
bool conv_bf3_gather_supported(const GatherConv& g, const Epi& ep);
int conv_bf3_gather(const GatherConv& g, const float* in, const float* wp, float* out, const Epi& ep,
                    const float* mask_y, float mask_slope, hipStream_t s);

}  // namespace srk
