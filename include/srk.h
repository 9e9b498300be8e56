/*
 * srk.h — C ABI of libsrk.so, the MI355X (gfx950 / CDNA4) kernel library behind the
 * convolutional super-resolution hot path.
 *
 * The reference (togheppi/pytorch-super-resolution-model-collection) has no FFI layer: its
 * hot path is the set of torch.nn calls made by base_networks.py and the per-model Net
 * classes.  Each entry point below names the reference call site (file:line under
 * /root/reference) whose arithmetic it replaces.  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 data owned by the caller; the library never
 *     allocates or frees, and does not retain a pointer past the call
 *   - activations are dense NHWC ("channels_last"): x[n][h][w][c];
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*) of the CURRENT
 *     device (hipSetDevice is the caller's job); no call synchronises the device;
 *     -- with ONE sanctioned exception: between srk_wgrad_reduce_defer(1) and the next
 *     srk_wgrad_reduce_flush() the calling thread's weight-gradient calls queue their slab
 *     reductions, and the queue holds their workspace / dw / db pointers until that flush (the
 *     contract is spelled out at srk_wgrad_reduce_defer);
 *   - calls may be made concurrently from several host threads and for several devices of one
 *     process.  The library's mutable state is: thread-local -- the last-error string, the name
 *     of the last dispatched kernel (srk_last_kernel_name, a diagnostic), the deferred-reduction
 *     queue above, and the two deprecated srk_last_conv_* values (results now come back through
 *     the out-fields of srk_epilogue); process-wide -- one atomic "dynamic-LDS limit raised"
 *     flag per (kernel, device) (hipFuncSetAttribute is a per-device property, set on a kernel's
 *     first large-LDS launch on that device), a mutex-guarded cache of LDS layouts per tile
 *     geometry (conv_bfd), the SRK_* debugging environment switches, each read once, and the
 *     device-side timeout counter behind srk_ring_timeouts();
 *   - return value: SRK_OK (0) or a negative srk_status; no exception crosses the ABI.
 */
#ifndef SRK_H_
#define SRK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRK_VERSION 600 /* major*10000 + minor*100 + patch; 0.6.0: srk_conv2d_forward_ex + srk_conv_result, the pair entry points retired */

typedef enum srk_status {
  SRK_OK = 0,
  SRK_ERR_BAD_ARG = -1,     /* null pointer, non-positive dim, inconsistent shapes        */
  SRK_ERR_UNSUPPORTED = -2, /* legal request that no kernel in this build covers          */
  SRK_ERR_LAUNCH = -3,      /* hipGetLastError() != hipSuccess after a launch             */
  SRK_ERR_WORKSPACE = -4    /* caller-provided workspace smaller than srk_*_workspace_bytes */
} srk_status;

/* Activation kinds: base_networks.py:50-60 (ReLU / PReLU / LeakyReLU(0.2) / Tanh / Sigmoid). */
typedef enum srk_act {
  SRK_ACT_NONE = 0,
  SRK_ACT_RELU = 1,
  SRK_ACT_PRELU = 2, /* slope(s) read from device memory (learnable, base_networks.py:54) */
  SRK_ACT_LRELU = 3, /* slope passed by value (0.2 in the reference, base_networks.py:56)  */
  SRK_ACT_TANH = 4,
  SRK_ACT_SIGMOID = 5
} srk_act;

/* Which implementation a conv call may use.  AUTO picks the fastest kernel that covers the
 * shape; GENERIC forces the plain gather kernel (used by the tests to cross-check).  The environment
 * variable SRK_FORCE_ALGO=generic|mfma|direct|bf16x3|bf16x6 overrides AUTO (SRK_FORCE_ALGO=mfma gives the
 * exact-fp32 MFMA path everywhere). */
typedef enum srk_algo {
  SRK_ALGO_AUTO = 0,        /* bf16x3 MFMA where it applies, else fp32 MFMA / direct / generic            */
  SRK_ALGO_GENERIC = 1,     /* plain fp32 gather kernel (any shape)                                       */
  SRK_ALGO_MFMA = 2,        /* exact fp32 MFMA (v_mfma_f32_16x16x4_f32)                                    */
  SRK_ALGO_DIRECT = 3,      /* fp32 VALU kernel for Cout <= 4                                              */
  SRK_ALGO_MFMA_BF16X3 = 4, /* 3-term bf16 split on v_mfma_f32_16x16x32_bf16, fp32 accumulate (~1e-5 rel) */
  SRK_ALGO_MFMA_BF16X6 = 5, /* exact 3-way operand split, 6 bf16 MFMAs per product: fp32-faithful (~1e-7
                               rel) at 2.7x the fp32-MFMA rate; shapes it does not cover run SRK_ALGO_MFMA */
  SRK_ALGO_MFMA_F16X3 = 6   /* fp32-faithful with THREE MFMAs per product: both operands scaled by a power of two (exact)
                               so that their largest magnitude sits at 2^13..2^14, split x = h + m into two fp16 planes
                               (2 x 11 significant bits), products m*h + h*m + h*h on v_mfma_f32_16x16x32_f16, fp32
                               accumulate, exact power-of-two descale (~1.2e-7 rms vs 0.8e-7 for plain fp32).  Needs
                               an upper bound of max|x| on the device (srk_epilogue.x_amax); forward convs the fp16
                               kernels cover (srk_conv2d_f16x3_supported), SRK_ERR_UNSUPPORTED elsewhere */
} srk_algo;

/* A device-side running maximum of |values|: SRK_AMAX_FLOATS floats = SRK_AMAX_SLOTS slots, one per 64-byte line (slot
 * i is float 16 i: atomics on one line serialise in one L2 channel, so the kernels spread theirs over the slots); the
 * value is the maximum over the slots.  Zero the buffer before the first producer; any upper bound is a valid content. */
#define SRK_AMAX_SLOTS 16
#define SRK_AMAX_FLOATS 256

/* Geometry of one torch.nn.Conv2d / ConvTranspose2d call.
 *   conv       (base_networks.py:42,112-113,156): y[n,oy,ox,co] = sum x[n,oy*s-p+kh,ox*s-p+kw,ci] * w
 *   transposed (base_networks.py:77, fsrcnn.py:33):  y[n,oy,ox,co] = sum x[n,(oy+p-kh)/s,(ox+p-kw)/s,ci] * w
 * (H,W,Cin) describe x and (OH,OW,Cout) describe y in both cases; OH/OW must equal
 * srk_conv_out_dim(). */
typedef struct srk_conv_desc {
  int32_t N, H, W, Cin;
  int32_t OH, OW, Cout;
  int32_t KH, KW;
  int32_t stride, pad;
  int32_t transposed; /* 0: Conv2d, 1: ConvTranspose2d */
  int32_t out_pad;    /* ConvTranspose2d output_padding (fsrcnn.py:33 uses 1) */
  int32_t algo;       /* srk_algo */
  int32_t x_nchw;     /* forward only: x is the caller's NCHW tensor (read in place by the Cin <= 4 bf16x3
                         first-layer kernel; SRK_ERR_UNSUPPORTED elsewhere) */
  int32_t dy_ps_r;    /* backward only: dy is handed over in the pixel-shuffled layout [N, OH*r, OW*r, Cout/r^2] that the
                         fused conv + PixelShuffle forward produced (PSBlock, base_networks.py:179-181); the bf16x3
                         data- and weight-gradient kernels un-shuffle while staging (SRK_ERR_UNSUPPORTED elsewhere:
                         call srk_pixel_shuffle_backward first) */
} srk_conv_desc;

/* Fused epilogue of a forward conv:  y = PS_r( act(conv + bias) ) + residual
 *   bias     : Conv2d bias (may be NULL — vdsr.py:17-24 and lapsrn.py are bias-free)
 *   act      : ConvBlock activation (base_networks.py:67-70)
 *   residual : torch.add(out, residual) of ResnetBlock / VDSR / EDSR / SRGAN
 *              (base_networks.py:149, vdsr.py:31, edsr.py:42, srgan.py:39); indexed like y
 *   ps_r     : PSBlock's PixelShuffle(r) fused into the store (base_networks.py:157,179-181);
 *              0 or 1 = none.  With ps_r>1, y is [N, OH*r, OW*r, Cout/r^2]. */
typedef struct srk_epilogue {
  const float* bias;
  const float* prelu_weight; /* SRK_ACT_PRELU: device slopes */
  const float* residual;
  float slope;               /* SRK_ACT_LRELU */
  int32_t act;               /* srk_act */
  int32_t prelu_n;           /* 1 (nn.PReLU() default) or number of output channels */
  int32_t ps_r;
  const float* x_amax;       /* SRK_ALGO_MFMA_F16X3: SRK_AMAX_FLOATS floats, max over the slots >= max|x| (NULL otherwise) */
  float* y_amax;             /* optional: SRK_AMAX_FLOATS floats that receive (atomic max) max|y| of this call's output
                                -- the next layer's x_amax.  Honoured by the kernels listed at srk_conv2d_f16x3_supported;
                                srk_conv2d_forward_ex reports whether the dispatched kernel did (srk_conv_result) */
  double* bn_partial;        /* optional (forward): the conv leaves the per-channel column sums of its OUTPUT for the
                                BatchNorm behind it (base_networks.py:46,117: conv -> bn) -- row t of
                                [tiles][2 * Cout] doubles = {sum y, sum y*y} over the pixels of tile t, summed in a fixed
                                order; room for N * ceil(OH / 8) * ceil(OW / 8) rows.  Honoured by the per-tile 64 -> 64
                                3x3 kernel (k_c64) only: srk_conv_result.bn_partial_rows says how many rows the call wrote
                                (0: none -- run srk_bn_stats_finalize as usual), srk_bn_finalize_partials() consumes them */
} srk_epilogue;  /* (layout frozen at library version 600 = the round-4 layout: results of a call travel in srk_conv_result, never here) */

/* What the kernel a forward call dispatched to did -- a HOST struct the caller owns, filled by srk_conv2d_forward_ex.
 * `struct_size` is set by the CALLER to sizeof(srk_conv_result) of the header it was compiled against: the library writes
 * only the fields that fit, so the struct can grow without breaking older callers. */
typedef struct srk_conv_result {
  uint32_t struct_size;
  int32_t wrote_amax;       /* 1: the kernel keeps the running maximum of |y| in ep->y_amax (and y_amax was given) -- only
                               then may the slots be handed to the next layer as x_amax (base_networks.py:101-104) */
  int32_t bn_partial_rows;  /* rows of ep->bn_partial the call filled (0: none -- run srk_bn_stats_finalize as usual) */
} srk_conv_result;

/* Activation-gradient prologue of the backward kernels: the incoming gradient dy is
 * multiplied by act'(.) while it is loaded, using the SAVED FORWARD OUTPUT `y`
 * (post-activation; valid for ReLU and positive-slope LeakyReLU, which is what
 * ReLU(True)/LeakyReLU(0.2, True) in base_networks.py:52,56 keep alive). */
typedef struct srk_bwd_mask {
  const float* y; /* NULL = no mask */
  float slope;    /* 0 for ReLU */
} srk_bwd_mask;

/* ---- queries --------------------------------------------------------------------------- */
int srk_version(void);
const char* srk_status_string(int status);
const char* srk_last_error_string(void); /* thread-local, valid until the next failing call */
/* Name (with template arguments) of the kernel the calling thread's last srk_conv2d_forward / _backward_data call
 * dispatched to, e.g. "k_conv_bfw<2,9,2>" — what a measurement should quote (thread-local, never NULL). */
const char* srk_last_kernel_name(void);
/* DEPRECATED aliases of srk_conv_result.wrote_amax / .bn_partial_rows (thread-local side channels that have to be queried
 * before the thread's next conv call; kept for callers written against the round-4 header). */
int srk_last_conv_wrote_amax(void);
int srk_last_conv_bn_partial_rows(void);
/* Diagnostic of the ring kernels (k_conv_bfr: producer and consumer waves of a persistent block hand halo buffers over
 * through counters in LDS, every poll has an iteration cap): number of polls that ran into the cap since the last
 * reset -- 0 in a correct library.  Synchronises the device; tests and fuzzers call it, the product never does. */
int srk_ring_timeouts(int reset);
/* Output spatial size of a conv / transposed conv along one axis (torch semantics). */
int srk_conv_out_dim(int in, int k, int stride, int pad, int transposed, int out_pad);

/* ---- layout ---------------------------------------------------------------------------- */
/* NCHW <-> NHWC copies at the module boundary (the reference feeds NCHW tensors:
 * edsr.py:146-152). */
int srk_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, void* stream);
int srk_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, void* stream);

/* Weight repacks.  `w` is the state_dict tensor exactly as torch stores it:
 *   Conv2d          [Cout][Cin][KH][KW]   (transposed = 0)
 *   ConvTranspose2d [Cin][Cout][KH][KW]   (transposed = 1)
 * Packed layout consumed by srk_conv2d_forward:        wp[kh][kw][ci][co]
 * Packed layout consumed by srk_conv2d_backward_data:  wp[kh][kw][co][ci]
 *   (for the data gradient the roles of the channel axes swap; the spatial flip is handled by
 *   the kernels' index arithmetic, not by the packing).
 * ps_r > 1 (forward only) additionally permutes the output channels to (i, j, c) order so a
 * fused pixel-shuffle store is contiguous; bias must then be packed with srk_pack_bias_ps.
 * A packed buffer is srk_packed_weight_bytes() long: the fp32 layout above followed (256-byte
 * aligned) by the same filter pre-split into bf16 hi/lo planes for the bf16x3 MFMA kernel. */
size_t srk_packed_weight_bytes(int Cout, int Cin, int KH, int KW, int bwd);
int srk_pack_weight_fwd(const float* w, float* wp, int Cout, int Cin, int KH, int KW, int transposed, int ps_r,
                        void* stream);
/* bwd, ps_r > 1: the contraction axis (Cout) is ordered (i, j, c) — the channel order of a pixel-shuffled dy, for
 * srk_conv2d_backward_data with srk_conv_desc.dy_ps_r set */
int srk_pack_weight_bwd(const float* w, float* wp, int Cout, int Cin, int KH, int KW, int transposed, int ps_r,
                        void* stream);
int srk_pack_bias_ps(const float* b, float* bp, int Cout, int ps_r, void* stream);
/* Whole-model packing (training: the weights change every step).  `params_base` is the flat fp32 parameter buffer,
 * `packed_base` a caller-owned byte buffer, `table` a DEVICE array of n_layers rows x 14 int64:
 *   0 w_off (floats)  1 fwd_off (bytes, -1 skip)  2 bwd_off (bytes, -1 skip)  3 Cout  4 Cin  5 KH  6 KW  7 transposed
 *   8 ps_r  9 bias_off (floats, -1)  10 bias_ps_off (bytes, -1)
 *   11 amax_off (bytes into packed_base, -1 none): a 4-byte scratch word per layer, ZEROED BY THE CALLER before the call,
 *      that lets the layer's max|w| (scale of the fp16 planes of SRK_ALGO_MFMA_F16X3) be taken by parallel slices
 *      instead of one block per layer; pass -blocks_per_layer when every row has one (skips the single-block pass)
 *   12 first block of the layer in `fast_blocks` (-1: generic path)  13 reserved (0)
 * Every fwd_off / bwd_off region receives exactly what srk_pack_weight_fwd / _bwd would write there.
 * fast_blocks (DEVICE, may be NULL): n_fast_blocks pairs {layer, local block} for the layers with column 12 >= 0 -- plain
 * Conv2d filters (not transposed, KH*KW <= 25, Cout = 32 or a multiple of 64, Cin = 16 / 32 / 48 or a multiple of 64 --
 * no channel padding in any prepared layout --, both directions wanted) take
 * (Cout / 8) * ceil(Cin / 32) blocks each, local block = octet * ceil(Cin / 32) + chunk; a block reads its 8 x 32-channel
 * filter tile once, coalesced, and writes all layouts from LDS (the generic path gathers every value separately). */
int srk_pack_weights_batched(const float* params_base, void* packed_base, const int64_t* table, int n_layers,
                             int blocks_per_layer, const int32_t* fast_blocks, int n_fast_blocks, void* stream);
/* max|x| over n floats into an SRK_AMAX_FLOATS buffer (atomic max: zero it first) -- x_amax of a tensor no kernel
 * of this library produced (the network input). */
int srk_absmax(const float* x, size_t n, float* amax_slots, void* stream);
/* 1 when srk_conv2d_forward(d, ..., ep) with d->algo = SRK_ALGO_MFMA_F16X3 has a kernel: Conv2d / ConvTranspose2d with
 * Cin >= 8, Cout >= 8, and first layers (Cin <= 4 Conv2d, Cout a multiple of 16, also with x_nchw), every output group
 * on the 16-byte store path.  Those kernels (k_conv_bfd, k_conv_bfw, k_conv_bf3_rows) also fill ep->y_amax -- in every
 * arithmetic they run, not only f16x3. */
int srk_conv2d_f16x3_supported(const srk_conv_desc* d, const srk_epilogue* ep, const float* y);

/* ---- convolution (Conv2d / ConvTranspose2d: base_networks.py:42,77,112-113,156; fsrcnn.py:33) */
int srk_conv2d_forward(const srk_conv_desc* d, const float* x, const float* w_packed_fwd, float* y,
                       const srk_epilogue* ep, void* stream);
/* The same call, reporting what the dispatched kernel did in *res (may be NULL = srk_conv2d_forward).  Replaces the
 * thread-local srk_last_conv_* side channels below. */
int srk_conv2d_forward_ex(const srk_conv_desc* d, const float* x, const float* w_packed_fwd, float* y,
                          const srk_epilogue* ep, srk_conv_result* res, void* stream);
/* dx = d(loss)/dx given dy; optional act-grad prologue on dy; optional fused "+ add_to"
 * (gradient fan-in of a residual connection).  Replaces aten::convolution_backward (input
 * gradient) as dispatched from loss.backward() — edsr.py:154, vdsr.py:146, srgan.py:286,309. */
int srk_conv2d_backward_data(const srk_conv_desc* d, const float* dy, const float* w_packed_bwd, float* dx,
                             const srk_bwd_mask* mask, const float* add_to, void* stream);
/* The same gradient, already multiplied by the ReLU gradient of the layer that PRODUCED x (`x_relu` = this conv's
 * input x, the output of an upstream conv + ReLU: dx <- dx * (x_relu > 0)).  The upstream layer's backward
 * (aten::threshold_backward inside loss.backward(), vdsr.py:146) would apply exactly this mask to its dy -- reading x
 * once more in its data-gradient AND its weight-gradient kernel; applied here it costs one read at the output tile,
 * and that layer's srk_conv2d_backward_* calls take mask = NULL.  Only where the wave-specialised kernel runs the
 * gradient (srk_conv2d_backward_data_relu_supported: stride-1 3x3 Conv2d, Cout <= 64 ... of benchmark size);
 * SRK_ERR_UNSUPPORTED otherwise. */
int srk_conv2d_backward_data_relu_supported(const srk_conv_desc* d, const float* dy, const float* dx, const srk_bwd_mask* mask);
int srk_conv2d_backward_data_relu(const srk_conv_desc* d, const float* dy, const float* w_packed_bwd, float* dx,
                                  const srk_bwd_mask* mask, const float* x_relu, void* stream);
/* dw (torch layout, see srk_pack_weight_*) and db (may be NULL).  beta = 0 overwrites,
 * beta = 1 accumulates into dw/db (shared weights: lapsrn.py:40,44).  `workspace` holds the
 * split-K partial sums; size from srk_conv2d_backward_weight_workspace_bytes. */
size_t srk_conv2d_backward_weight_workspace_bytes(const srk_conv_desc* d);
int srk_conv2d_backward_weight(const srk_conv_desc* d, const float* x, const float* dy, const srk_bwd_mask* mask,
                               float* dw, float* db, float beta, void* workspace, size_t workspace_bytes,
                               void* stream);

/* The weight gradients of n convolutions that share ONE geometry `d` — the 32 + 1 body convs of EDSR (edsr.py:37-45),
 * the 18 of VDSR (vdsr.py:17-24), the 32 of SRResNet — in one launch plus one reduce launch.  x / dy / dw / db are HOST
 * arrays of n device pointers (db may be NULL, or hold n non-NULL pointers: bias and bias-free layers cannot share a
 * group), masks a host array of n srk_bwd_mask (NULL = none; an entry's y may be NULL).  No two layers may share a dw
 * (LapSRN's aliased branches go into separate calls).  Same results as n srk_conv2d_backward_weight calls; geometries
 * without a grouped kernel run exactly those.  Nothing but the pointer VALUES is read from the host arrays, at call
 * time (the call is hipGraph-capturable). */
size_t srk_conv2d_backward_weight_grouped_workspace_bytes(const srk_conv_desc* d, int n);
int srk_conv2d_backward_weight_grouped(const srk_conv_desc* d, int n, const float* const* x, const float* const* dy,
                                       const srk_bwd_mask* masks, float* const* dw, float* const* db, float beta,
                                       void* workspace, size_t workspace_bytes, void* stream);
/* Deferred slab reductions.  Every weight-gradient call above ends in a short launch that sums its split-K partial slabs
 * into dw / db.  After srk_wgrad_reduce_defer(1) the calls of this thread queue that reduction instead, and
 * srk_wgrad_reduce_flush(stream) runs everything queued as ONE launch (each reduction in its own summation order: same
 * results bit for bit) -- the end of loss.backward() (edsr.py:154, srgan.py:286,309) is a handful to dozens of such calls.
 * Contract while deferring: every call gets its OWN workspace, alive and untouched until the flush, all calls and the
 * flush use one stream; a second update of a queued dw / db flushes first.  srk_wgrad_reduce_defer returns the previous
 * setting; switching it off does not flush. */
int srk_wgrad_reduce_defer(int on);
int srk_wgrad_reduce_flush(void* stream);

/* ---- residual block, both convolutions in one launch (base_networks.py:109-150 with norm=None, activation='relu':
 * edsr.py:37-45 builds its body from it) ------------------------------------------------------------------------------
 * Forward:   y_mid = relu(conv1(x) + b1),  y = conv2(y_mid) + b2 + x      (3x3, stride 1, pad 1, C -> C -> C)
 * Backward:  d_mid = conv2^T(dy) * (y_mid > 0),  dx = conv1^T(d_mid) + dy
 * Same numbers as the two srk_conv2d_forward / srk_conv2d_backward_data calls they replace (same operand splits,
 * same products; the K-split partial sums meet in a different order).  Only for small problems (strong-scaled shards:
 * srk_resblock2_supported says when): one 8x8 tile per workgroup, the intermediate stays in LDS, and nothing waits for
 * the store -> load round trip between the two convs.  y_mid / d_mid are still written: the weight gradients
 * (srk_conv2d_backward_weight with x = y_mid, dy = dy for conv2 and x = x, dy = d_mid, NO mask, for conv1) need them.
 * algo: SRK_ALGO_MFMA_BF16X6 / SRK_ALGO_MFMA_F16X3 (fp32-faithful; the latter forward only, with x_amax = the
 * SRK_AMAX_FLOATS running-maximum buffer of |x|) or SRK_ALGO_AUTO / SRK_ALGO_MFMA_BF16X3.  y_amax (optional, any algo) receives
 * max|y|.  b1 / b2 may be NULL.
 * Filters: the packed buffers of srk_pack_weight_fwd / srk_pack_weight_bwd (ps_r = 0). */
int srk_resblock2_supported(int N, int H, int W, int C);
int srk_resblock2_forward(int N, int H, int W, int C, const float* x, const float* w1_packed_fwd, const float* b1,
                          const float* w2_packed_fwd, const float* b2, float* y_mid, float* y, int algo,
                          const float* x_amax, float* y_amax, void* stream);
int srk_resblock2_backward_data(int N, int H, int W, int C, const float* dy, const float* w2_packed_bwd,
                                const float* w1_packed_bwd, const float* y_mid, float* d_mid, float* dx, int algo,
                                void* stream);

/* ---- pixel shuffle (torch.nn.PixelShuffle: base_networks.py:157,179-181) ----------------- */
/* x [N,H,W,C*r*r] -> y [N,H*r,W*r,C];  channel c*r*r + i*r + j -> (c, h*r+i, w*r+j). */
int srk_pixel_shuffle_forward(const float* x, float* y, int N, int H, int W, int C, int r, void* stream);
/* dy [N,H*r,W*r,C] -> dx [N,H,W,C*r*r]  (aten::pixel_unshuffle in backward). */
int srk_pixel_shuffle_backward(const float* dy, float* dx, int N, int H, int W, int C, int r, void* stream);

/* ---- pointwise (base_networks.py:50-60,149; vdsr.py:31; edsr.py:42) ------------------------ */
/* y = act(x); channels = innermost (NHWC) extent, used only by per-channel PReLU. */
int srk_act_forward(const float* x, float* y, size_t n, int channels, int act, float slope,
                    const float* prelu_weight, int prelu_n, void* stream);
/* dx = dy * act'(.).  `saved` is the forward INPUT x for PReLU and the forward OUTPUT y for
 * every other kind.  PReLU also accumulates d(slope) into dprelu (+=, caller zeroes). */
int srk_act_backward(const float* dy, const float* saved, float* dx, size_t n, int channels, int act, float slope,
                     const float* prelu_weight, int prelu_n, float* dprelu, void* stream);
/* out = alpha*a + beta*b  (torch.add and autograd's gradient fan-in) */
int srk_axpby(const float* a, const float* b, float* out, size_t n, float alpha, float beta, void* stream);
/* out = x * (*alpha_dev): chain-rule scaling by an upstream scalar gradient that lives on the device
 * (e.g. the 1e-3 weight of the adversarial term, srgan.py:308). */
int srk_scale_dev(const float* x, const float* alpha_dev, float* out, size_t n, void* stream);

/* ---- losses, reduction='mean' (srcnn.py:84-86,129; edsr.py:98-100,153; lapsrn.py:75-85;
 *      srgan.py:157,276-297).  One pass: *loss = mean(...), dpred = d(loss)/d(pred)*grad_scale.
 * pred/dpred are NHWC-dense; target is addressed through explicit element strides
 * (n,c,h,w) so an NCHW target batch needs no copy.  dpred may be NULL (evaluation).
 * `partials` is a caller-provided scratch of srk_loss_workspace_bytes(). */
typedef enum srk_loss { SRK_LOSS_MSE = 0, SRK_LOSS_L1 = 1, SRK_LOSS_CHARBONNIER = 2, SRK_LOSS_BCE = 3 } srk_loss;
size_t srk_loss_workspace_bytes(void);
int srk_loss_forward_backward(int kind, const float* pred, const float* target, const int64_t* target_strides,
                              int N, int C, int H, int W, float eps, float grad_scale, float* loss, float* dpred,
                              void* workspace, void* stream);

/* ---- optimizers over flat fp32 buffers (srcnn.py:79; fsrcnn.py:105-106; vdsr.py:86-90,149;
 *      espcn.py:79; edsr.py:93; srgan.py:147-149) --------------------------------------------- */
/* torch.optim.SGD: g += wd*p; buf = mom*buf + g (first step: buf = g); p -= lr*(nesterov ? g+mom*buf : buf).
 * `first_step` selects the buf = g initialisation. lr/scale are read from device memory when the
 * *_dev pointers are non-NULL (keeps a captured hipGraph valid across LR decay / grad clipping). */
int srk_sgd_step(float* p, const float* g, float* momentum_buf, size_t n, float lr, float momentum,
                 float weight_decay, int nesterov, int first_step, const float* lr_dev, const float* grad_scale_dev,
                 void* stream);
/* torch.optim.Adam (betas, eps, no amsgrad): `step_dev` points at TWO device int32 {step count, 0}: the kernel works
 * with count + 1 (bias correction) and stores it when its last block finishes; the second word is the kernel's
 * arrival ticket and is 0 between launches. */
int srk_adam_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int32_t* step_dev, const float* lr_dev,
                  const float* grad_scale_dev, void* stream);
/* torch.nn.utils.clip_grad_norm (vdsr.py:149): *norm_out = ||g||_2; *scale_out = min(1, max_norm/(norm+1e-6)). */
size_t srk_grad_norm_workspace_bytes(void);
int srk_grad_norm_clip(const float* g, size_t n, float max_norm, float* norm_out, float* scale_out, void* workspace,
                       void* stream);

/* ---- BatchNorm2d, train/eval (base_networks.py:46,117,161; live in srgan.py only) ---------- */
/* Training forward: batch statistics over (N,H,W) per channel, y = (x-mean)*rstd*gamma+beta,
 * running stats updated with `momentum` (unbiased variance), save_mean/save_rstd kept for
 * backward.  sum/sumsq partials are exposed through `stats` ([2*C] doubles: sum, sumsq) so a
 * data-parallel caller can all-reduce them (SyncBN) between the two phases. */
int srk_bn_stats(const float* x, double* stats, size_t rows, int C, void* workspace, void* stream);
size_t srk_bn_workspace_bytes(int C);
/* num_batches_tracked: the module's int64 counter buffer (device), incremented by one; may be NULL */
int srk_bn_finalize(const double* stats, double count, float* save_mean, float* save_rstd, float* running_mean,
                    float* running_var, float momentum, float eps, int C, int64_t* num_batches_tracked, void* stream);
/* srk_bn_stats + srk_bn_finalize in two launches instead of three (single-GPU / per-shard statistics: no all-reduce
 * between the phases); `stats` still receives the [2*C] sums. */
int srk_bn_stats_finalize(const float* x, double* stats, size_t rows, int C, float* save_mean, float* save_rstd,
                          float* running_mean, float* running_var, float momentum, float eps,
                          int64_t* num_batches_tracked, void* workspace, void* stream);
/* The second launch of srk_bn_stats_finalize alone, on column sums a convolution's epilogue left
 * (srk_epilogue.bn_partial: `splits` rows of [2*C] doubles): mean / rstd / running statistics of base_networks.py:46,117's
 * BatchNorm without the pass over the activation.  `rows` = N*H*W of the activation. */
int srk_bn_finalize_partials(const double* partials, int splits, double* stats, size_t rows, int C, float* save_mean,
                             float* save_rstd, float* running_mean, float* running_var, float momentum, float eps,
                             int64_t* num_batches_tracked, void* stream);
/* y_amax (optional, here and in srk_bn_apply_act): SRK_AMAX_FLOATS floats that receive max|y| -- the x_amax of a
 * following SRK_ALGO_MFMA_F16X3 convolution (16-byte path only: C % 4 == 0, aligned tensors). */
int srk_bn_apply(const float* x, float* y, const float* mean, const float* rstd, const float* gamma,
                 const float* beta, size_t rows, int C, int act, float slope, float* y_amax, void* stream);
int srk_bn_eval_params(const float* running_mean, const float* running_var, float eps, float* mean, float* rstd,
                       int C, void* stream);
/* nn.InstanceNorm1d on the [B, F] output of a Linear (DenseBlock(norm='instance'), base_networks.py:12-13): torch reads the
 * 2-D tensor as one unbatched sample of B channels x F positions, i.e. every ROW is normalised with its own biased
 * statistics (no affine parameters, no running statistics).  mean / rstd: [rows] outputs the backward needs. */
int srk_rownorm_forward(const float* x, float* y, float* mean, float* rstd, int rows, int cols, float eps, void* stream);
int srk_rownorm_backward(const float* dy, const float* x, const float* mean, const float* rstd, float* dx, int rows,
                         int cols, void* stream);
/* Backward: `dstats` [2*C] doubles = (sum dy, sum dy*xhat) (all-reducible for SyncBN);
 * srk_bn_backward_apply writes dx = gamma*rstd*(dy - dstats[c]/count - xhat*dstats[C+c]/count)
 * (pass zeros for eval-mode BN); srk_bn_param_grads accumulates dbeta += dstats[c],
 * dgamma += dstats[C+c]. */
int srk_bn_backward_stats(const float* dy, const float* x, const float* mean, const float* rstd, double* dstats,
                          size_t rows, int C, void* workspace, void* stream);
/* srk_bn_backward_stats + srk_bn_param_grads (from the LOCAL sums, as the data-parallel gradient exchange expects) in
 * two launches instead of three; dgamma / dbeta may be NULL. */
int srk_bn_backward_stats_grads(const float* dy, const float* x, const float* mean, const float* rstd, double* dstats,
                                size_t rows, int C, float* dgamma, float* dbeta, void* workspace, void* stream);
int srk_bn_backward_apply(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                          const double* dstats, double count, float* dx, size_t rows, int C, void* stream);
int srk_bn_param_grads(const double* dstats, float* dgamma, float* dbeta, int C, void* stream);
/* BatchNorm with the activation that follows it in the reference's blocks, and the residual add of the BatchNorm
 * ResnetBlock, folded in: act(bn(.)) of ConvBlock / DeconvBlock / PSBlock / DenseBlock (base_networks.py:58-71) and
 * bn(conv2(.)) + x (base_networks.py:141-150).
 *   srk_bn_apply_act:   y = act(gamma * (x - mean) * rstd + beta) [+ residual]   act: NONE / RELU / LRELU / PRELU
 *   srk_bn_backward_stats_grads_act:  dz = dy * act'(z) with z RECOMPUTED from x (nothing of the activation is saved);
 *       dstats = (sum dz, sum dz * xhat), dbeta += / dgamma += those, dprelu += sum_{z <= 0} dy * z (PRELU; NULL ok)
 *   srk_bn_backward_apply_act:        dx = gamma * rstd * (dz - dstats[c]/count - xhat * dstats[C+c]/count)
 * The gradient of `residual` is dy itself.  C must be a multiple of 4, tensors 16-byte aligned.  prelu_n: 1 or C. */
int srk_bn_apply_act(const float* x, float* y, const float* mean, const float* rstd, const float* gamma,
                     const float* beta, size_t rows, int C, int act, float slope, const float* prelu_weight, int prelu_n,
                     const float* residual, float* y_amax, void* stream);
int srk_bn_backward_stats_grads_act(const float* dy, const float* x, const float* mean, const float* rstd,
                                    const float* gamma, const float* beta, double* dstats, size_t rows, int C,
                                    float* dgamma, float* dbeta, int act, float slope, const float* prelu_weight,
                                    int prelu_n, float* dprelu, void* workspace, void* stream);
int srk_bn_backward_apply_act(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                              const float* beta, const double* dstats, double count, float* dx, size_t rows, int C,
                              int act, float slope, const float* prelu_weight, int prelu_n, void* stream);

/* Round 6, "finalize-in-apply": the same BatchNorm in ONE launch less per direction.  The apply kernels finish the split
 * reduction themselves (every block re-sums the partials of its own 16-channel slab in the reduce kernel's order: forward
 * statistics are bit-equal to the calls above), so a training BatchNorm is
 *   forward : [srk_bn_stats_partials |  a conv that left srk_epilogue.bn_partial]  ->  srk_bn_finalize_apply_act
 *   backward:  srk_bn_backward_partials_act                                         ->  srk_bn_backward_finalize_apply_act
 * Per-shard statistics only (SyncBN all-reduces the [2C] sums between the phases: use the calls above).
 * srk_bn_fused_supported(C): C % 16 == 0 and C <= 512.  `partials`: [splits][2][C] doubles (backward with an activation:
 * [splits][3][C]) in a workspace of srk_bn_workspace_bytes(C); *splits_out is a HOST int the partials call fills.
 * What the separate calls return is still returned: stats / dstats [2C], save_mean / save_rstd, running statistics,
 * num_batches_tracked, dgamma += / dbeta += / dprelu +=.  Tensors 16-byte aligned. */
int srk_bn_fused_supported(int C);
int srk_bn_stats_partials(const float* x, size_t rows, int C, void* workspace, int* splits_out, void* stream);
int srk_bn_finalize_apply_act(const double* partials, int splits, double* stats, size_t rows, int C, float* save_mean,
                              float* save_rstd, float* running_mean, float* running_var, float momentum, float eps,
                              int64_t* num_batches_tracked, const float* x, float* y, const float* gamma, const float* beta,
                              int act, float slope, const float* prelu_weight, int prelu_n, const float* residual,
                              float* y_amax, void* stream);
int srk_bn_backward_partials_act(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                 const float* beta, size_t rows, int C, int act, float slope, const float* prelu_weight,
                                 int prelu_n, void* workspace, int* splits_out, void* stream);
int srk_bn_backward_finalize_apply_act(const double* partials, int splits, double* dstats, double count, const float* dy,
                                       const float* x, const float* mean, const float* rstd, const float* gamma,
                                       const float* beta, float* dx, size_t rows, int C, float* dgamma, float* dbeta,
                                       int act, float slope, const float* prelu_weight, int prelu_n, float* dprelu,
                                       void* stream);

/* ---- Linear (DenseBlock: base_networks.py:7; srgan.py:66-70) -------------------------------- */
/* y[B,Out] = act(x[B,In] @ w[Out,In]^T + b) */
int srk_linear_forward(const float* x, const float* w, const float* b, float* y, int B, int In, int Out, int act,
                       float slope, void* stream);
int srk_linear_backward(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int B,
                        int In, int Out, float beta, void* stream);

/* ---- utils.img_interp (utils.py:242-269): ToPILImage -> Image.resize -> ToTensor, bit-exact ---------------------
 * x, y: dense NCHW fp32 (the reference's FloatTensor layout); values are quantised to uint8 by truncation of x*255
 * (saturating outside [0,1]), resampled with Pillow's 8-bit two-pass algorithm and returned as uint8/255.
 * SRCNN / VDSR / LapSRN call it every iteration (srcnn.py:119, vdsr.py:137, lapsrn.py:183). */
typedef enum srk_interp { SRK_INTERP_NEAREST = 0, SRK_INTERP_BILINEAR = 2, SRK_INTERP_BICUBIC = 3 } srk_interp; /* PIL codes */
size_t srk_img_interp_workspace_bytes(int N, int C, int H, int W, int OH, int OW, int filter);
int srk_img_interp(const float* x_nchw, float* y_nchw, int N, int C, int H, int W, int OH, int OW, int filter,
                   void* workspace, size_t workspace_bytes, void* stream);

/* ---- training-set pipeline on 8-bit images (dataset.py:51-99; torchvision's Scale / RandomCrop / flips around Pillow) ----
 * Image.resize((OW, OH), BICUBIC | BILINEAR) on 8-bit planes, bit-exact with Pillow (same two-pass resampler as
 * srk_img_interp; a pass is skipped when that axis keeps its size, as Pillow does).  x is addressed through element
 * strides (plane, row, pixel): an interleaved HWC decode buffer is read in place with (1, W*C, C).  y: planar
 * [planes][OH][OW], uint8 (out_float = 0) or float = value / 255 (out_float = 1: ToTensor). */
size_t srk_img_resize_u8_workspace_bytes(int planes, int H, int W, int OH, int OW, int filter);
int srk_img_resize_u8(const uint8_t* x, int64_t plane_stride, int64_t row_stride, int64_t px_stride, void* y,
                      int out_float, int planes, int H, int W, int OH, int OW, int filter, void* workspace,
                      size_t workspace_bytes, void* stream);
/* RandomCrop -> Image.rotate(90*rot_k, expand=True) (counter-clockwise quarter turns) -> horizontal flip -> vertical
 * flip (dataset.py:65-84) as one gather: y planar [C][crop_h or crop_w][...] uint8. */
int srk_patch_augment_u8(const uint8_t* x, int64_t plane_stride, int64_t row_stride, int64_t px_stride, uint8_t* y, int C,
                         int H, int W, int crop_x, int crop_y, int crop_w, int crop_h, int rot_k, int fliplr, int fliptb,
                         void* stream);
/* The two calls above folded into one per training patch (dataset.py:51-82): img_hwc is the decoded interleaved 8-bit
 * image [H][W][C]; scale_h / scale_w > 0 rescale the whole image first (Image.resize, BICUBIC), 0 = no rescale; then
 * crop -> rot_k quarter turns ccw -> flips into the planar patch out_planar [C][oh][ow]. */
size_t srk_patch_from_image_u8_workspace_bytes(int C, int H, int W, int scale_h, int scale_w);
int srk_patch_from_image_u8(const uint8_t* img_hwc, int C, int H, int W, int scale_h, int scale_w, int crop_x, int crop_y,
                            int crop_w, int crop_h, int rot_k, int fliplr, int fliptb, uint8_t* out_planar,
                            void* workspace, size_t workspace_bytes, void* stream);

/* ---- steps either side of the nets ------------------------------------------------
 * utils.PSNR (utils.py:208-216): mse = mean((clamp(pred,0,1) - gt)^2) over all elements, *psnr_out = mse == 0 ? 100 :
 * 10*log10(1/mse), on the device (the reference copies both images to the host per test image).  pred / gt are
 * addressed through element strides (n,c,h,w); NULL = NHWC-dense.  mse_out may be NULL. */
size_t srk_psnr_workspace_bytes(void);
int srk_psnr(const float* pred, const int64_t* pred_strides, const float* gt, const int64_t* gt_strides, int N, int C,
             int H, int W, float* psnr_out, float* mse_out, void* workspace, void* stream);
/* utils.norm / utils.denorm (utils.py:219-239; torchvision Normalize = sub_(mean).div_(std)):
 * y[e] = (x[e] - sub[c]) / div[c], c = (e / inner) % C (inner = H*W for NCHW storage, 1 for NHWC), optionally clamped
 * to [0,1] (denorm's non-VGG branch).  sub_host / div_host are HOST arrays of C <= 8 floats. Bit-equal to torch. */
int srk_channel_affine(const float* x, float* y, size_t n, int C, size_t inner, const float* sub_host,
                       const float* div_host, int clamp01, void* stream);
/* torch.nn.Upsample(scale_factor=r, mode='nearest') of Upsample2xBlock('rnc') (base_networks.py:204-210), NHWC:
 * y[n,oy,ox,c] = x[n,oy/r,ox/r,c]; backward: dx = sum of each r x r block of dy. */
int srk_upsample_nearest_forward(const float* x, float* y, int N, int H, int W, int C, int r, void* stream);
int srk_upsample_nearest_backward(const float* dy, float* dx, int N, int H, int W, int C, int r, void* stream);
/* nn.MaxPool2d(2, 2) of the VGG19 feature extractor (srgan.py:84-90: vgg19.features[:9] holds one), NHWC, floor mode:
 * y [N, H/2, W/2, C].  Forward only — the reference evaluates the VGG loss on detached tensors (srgan.py:302-305). */
int srk_maxpool2x2_forward(const float* x, float* y, int N, int H, int W, int C, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SRK_H_ */
