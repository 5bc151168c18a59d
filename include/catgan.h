/*
 * catgan.h — C ABI of the MI355X-native DCGAN step engine (libcatgan_hip.so).
 *
 * This is the drop-in boundary for the one hot path of aleju/cat-generator:
 * the alternating D/G update of adversarial.lua:51-275 on G32up-c / G32up /
 * D32_st3 (models.lua:196-228, :138-160, :640-711, :814-906).  The reference
 * reaches its arithmetic through un-vendored Torch7 rocks (nn / cunn /
 * cudnn.torch / stn / optim); every entry point below replaces the native
 * call one nn.Module method makes, and cites the reference call site that
 * constructs / invokes that module.  A LuaJIT `ffi.cdef` of this header is
 * the binding a maintainer adds.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes only.  All tensors fp32.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  All
 *    compute entry points are asynchronous on that stream.
 *  - every function returns 0 on success, non-zero on failure;
 *    cg_last_error() returns a thread-local message.  No C++ exception
 *    crosses the ABI (LuaJIT FFI cannot unwind them).
 *  - feature maps are NHWC in device memory ("BHWD" in stn's vocabulary):
 *    x[((n*H + y)*W + x)*C + c].  2-D tensors are row-major [rows][cols].
 *  - parameters and their gradients stay in the Torch7 *canonical* layout
 *    (conv weight [Cout][Cin][kH][kW], linear weight [out][in], see
 *    Module:getParameters(), train.lua:184-185); the engine keeps packed
 *    copies (cg_pack_*) that the GEMM kernels read.
 *  - "accumulate" outputs (gradWeight/gradBias/gradAlpha) follow Torch7's
 *    accGradParameters contract: result is ADDED (scaled) to the buffer.
 */
#ifndef CATGAN_H
#define CATGAN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CG_ABI_VERSION 1

/* ---- runtime: device memory / streams for hosts without their own (LuaJIT).
 * Replaces cutorch.setDevice (train.lua:109), CudaTensor storage, and the
 * nn.Copy('torch.FloatTensor','torch.CudaTensor') layers (models.lua:643,704;
 * utils/nn_utils.lua:638-643). */
int         cg_abi_version(void);
const char* cg_last_error(void);
/* Tunables of the kernel dispatch, named like the environment variables that set their defaults (13 names; value == -1 restores the
 * default).  Results never depend on them beyond fp32 re-association; the parity tests use them to run every compiled kernel variant
 * against the oracle.
 *   CG_NN_TILE / CG_TN_TILE (bm*1000+bn), CG_NN_SPLITS / CG_TN_SPLITS : block tile and split count of the forward (data-gradient) /
 *       weight-gradient GEMMs, 0 = the plan's choice;
 *   CG_GEMM_BK32, CG_WINO_BK : K step of the LDS-direct kernels;
 *   CG_NN_GLDS (0..3) / CG_TN_GLDS / CG_WINO_GLDS : how a K tile reaches the MFMAs - LDS-direct loads (default) or register-staged (0);
 *   CG_XCD_SWIZZLE : bits 1 row ranges per XCD, 2 pixel chunks per XCD in the weight gradients, 4 weights-stationary XCDs (default 7);
 *   CG_SKINNY : 1 = the MFMA scatter-form kernels of csrc/skinny.hip for 3x3 layers with <= 3 planes on one side, 2 = round 1's VALU
 *       kernels, 0 = the generic GEMM;
 *   CG_WINO3 : 1 = fused-transform Winograd F(2x2,3x3) for plain 64 -> 64 plane 3x3 layers with >= 2 workgroups per CU, 2 = wherever the
 *       geometry fits, 0 = never;
 *   CG_PAD_SKIP : least share (per cent) of zero-padding MACs from which a plain convolution's forward / data gradient runs on
 *       POSITION-MAJOR row tiles and does not issue them (default 20: D32_st3's 7x7 layer at 8x8 with 38 %, models.lua:685, not its 5x5
 *       layer at 16x16 with 14 %).  The weight gradient (round 6: position-major K tiles, igemm_tng_kernel mode 2) takes HALF that share
 *       (7x7, 5x5, the 3x3 layers at 8x8), and with any value > 0 the weight gradient of View -> Linear (an H x W kernel on the H x W map)
 *       is one kernel straight into the canonical gradWeight (csrc/headwg.hip).  0 = image-major tiles everywhere.
 * (CG_SPLIT_TARGET / _MINK, CG_TN_SMAX / _TARGET, CG_EPILOGUE_STATS, CG_COLREDUCE_WGS_PER_CU, CG_EW_WGS_PER_CU became constants in round 6.) */
int cg_set_option(const char* name, long value);
int cg_get_option(const char* name, long* value);
int cg_device_count(int* count);
int cg_set_device(int device);
int cg_malloc(void** dptr, size_t bytes);
int cg_free(void* dptr);
int cg_memcpy_h2d(void* stream, void* dst, const void* src, size_t bytes);
int cg_memcpy_d2h(void* stream, void* dst, const void* src, size_t bytes);
int cg_memcpy_d2d(void* stream, void* dst, const void* src, size_t bytes);
int cg_memset_zero(void* stream, void* dst, size_t bytes);
int cg_stream_create(void** stream);
/* A side stream on a chosen HARDWARE QUEUE.  The HIP runtime serves every stream of a process from four hardware queues, assigned
 * in creation order; two streams of one queue run back to back whatever the events between them say, so which queue a side stream
 * lands on decides whether its work overlaps with the caller's at all (csrc/common.h has the measurements).  queue_class 0 = the
 * queue of ref_stream itself, 1..3 = the other three (numbered by a one-off timing probe); slot picks among the pool's streams of
 * that class.  The stream belongs to the library: do NOT pass it to cg_stream_destroy.  Not callable for the first time inside a
 * graph capture. */
int cg_stream_on_queue(void* ref_stream, int queue_class, int slot, void** stream);
int cg_stream_destroy(void* stream);
int cg_stream_sync(void* stream);
/* Input pipeline (dataset.lua:123-170, adversarial.lua:225-230): page-locked host buffers, so that cg_memcpy_h2d on a copy
 * stream really is asynchronous, and events to hand a finished upload over to the compute stream without a host sync. */
int cg_host_alloc(void** hptr, size_t bytes);
int cg_host_free(void* hptr);
int cg_event_create(void** event);
int cg_event_destroy(void* event);
int cg_event_record(void* event, void* stream);
int cg_event_sync(void* event);
int cg_stream_wait_event(void* stream, void* event);
/* Decoded images as the loader produces them (8-bit RGB, [N][H][W][3]) -> the engine's fp32 NHWC pool in [0,1]:
 * colorspace 0 'rgb' (3 planes, v / 255), 1 'y' (1 plane, 0.21 R + 0.72 G + 0.07 B of the scaled values,
 * nn_utils.lua:253-277).  Every operation is a single correctly rounded fp32 one, in the host loader's order. */
int cg_images_u8_to_f32(void* stream, const unsigned char* src, float* dst, long npixels, int colorspace);
/* The loader's whole per-image arithmetic (dataset.lua:123-131,166) on the device: image.load's floats (byte / 255) -> image.scale to
 * Hd x Wd -> colour space, from the decoded 8-bit image at its own size [N][Hs][Ws][3] to the fp32 NHWC pool [N][Hd][Wd][C].
 * image.scale = the `image` rock's default separable 'bilinear' [upstream, recalled]: rows first, then columns, each pass in fp32;
 * shrinking an axis averages the source samples a target sample covers (fractional ends weighted), enlarging interpolates
 * linearly between the two neighbours (corner aligned), equal sizes copy.  Shrink factors up to 6 per axis. */
int cg_images_u8_scale_to_f32(void* stream, const unsigned char* src, float* dst, int N, int Hs, int Ws, int Hd, int Wd, int colorspace);

/* ---- convolution / linear (implicit GEMM on fp32 MFMA) -------------------
 * Replaces cudnn.SpatialConvolution (models.lua:206,212,218,222),
 * nn.SpatialConvolution (models.lua:646-685, 844-846) and nn.Linear
 * (models.lua:199,697,700,850,853); stride 1, zero padding.
 *
 * cg_conv2d_forward, ups == 0: for packed weights wpk[(ky*kW+kx)*Cin+ci][Cout] (cg_pack_conv_weight's wf),
 *   y[n,oy,ox,co] = bias[co] + sum_{ky,kx,ci} x[n, oy+ky-padH, ox+kx-padW, ci] * wpk[..][co]
 * with Ho = Hp + 2*padH - kH + 1.  The same entry point is updateGradInput when called with the gradOutput as x
 * and the backward-packed weights (wb) with pad' = k-1-pad, and nn.Linear when Hp=Wp=kH=kW=1.
 *
 * ups == 1 folds nn.SpatialUpSamplingNearest(2) (models.lua:205,211,217) into the convolution: x is the LOW-RES
 * tensor [N,Hp,Wp,Cin], y is [N,2Hp,2Wp,Cout] = conv(upsample2(x)).  It is evaluated as four phase convolutions
 * with pre-summed k' x k' weights, so wpk must come from cg_pack_conv_weight_ups2 (wf_ph); requires an odd
 * square kernel with pad = (k-1)/2.  bias may be NULL.  ws/ws_bytes: scratch for split-K partials,
 * cg_conv2d_workspace_bytes() tells how much is needed (may be 0). */
size_t cg_conv2d_workspace_bytes(int N, int Hp, int Wp, int Cin, int Cout,
                                 int kH, int kW, int padH, int padW, int ups);
int cg_conv2d_forward(void* stream, const float* x, const float* wpk, const float* bias,
                      float* y, int N, int Hp, int Wp, int Cin, int Cout,
                      int kH, int kW, int padH, int padW, int ups,
                      void* ws, size_t ws_bytes);

/* Grouped form: `ngroups` (<= 4) independent convolutions of IDENTICAL geometry in one launch (blockIdx.z = group);
 * x/wpk/bias/y are arrays of ngroups device pointers (bias may be NULL, or hold NULL entries).  Used for the
 * structurally identical branches of D32_st3 (models.lua:653-678): same shapes, separate tensors and parameters. */
size_t cg_conv2d_workspace_bytes_grouped(int ngroups, int N, int Hp, int Wp, int Cin, int Cout,
                                         int kH, int kW, int padH, int padW, int ups);
int cg_conv2d_forward_grouped(void* stream, int ngroups, const float* const* x, const float* const* wpk,
                              const float* const* bias, float* const* y, int N, int Hp, int Wp, int Cin, int Cout,
                              int kH, int kW, int padH, int padW, int ups, void* ws, size_t ws_bytes);

/* The same launch with a fused epilogue (what Torch7 runs as separate modules right behind the convolution):
 *  - act = 1: nn.PReLU (models.lua:647,649,...; alpha[g] = device pointer to branch g's shared slope),
 *    act = 2: nn.LeakyReLU(slope) (models.lua:845,847,851).  y keeps the pre-activation (the activation's backward
 *    needs it), y_act[g] receives act(y).  act = 0: alpha / y_act ignored.
 *  - stats != NULL (ngroups == 1): per-tile column sums of y and y^2, [cg_conv2d_stats_rows()][2][Cout] floats, from
 *    which cg_bn_stats_finalize builds the batch statistics of the nn.SpatialBatchNormalization behind the layer
 *    (models.lua:206-207,212-213) without another pass over y.  cg_conv2d_stats_rows() == 0 means this geometry is
 *    run split-K / skinny and cannot produce them (use cg_bn_stats). */
size_t cg_conv2d_stats_rows(int N, int Hp, int Wp, int Cin, int Cout, int kH, int kW, int padH, int padW, int ups);
int cg_conv2d_forward_ex(void* stream, int ngroups, const float* const* x, const float* const* wpk,
                         const float* const* bias, float* const* y, int N, int Hp, int Wp, int Cin, int Cout,
                         int kH, int kW, int padH, int padW, int ups,
                         int act, float slope, const float* const* alpha, float* const* y_act, float* stats,
                         void* ws, size_t ws_bytes);

/* updateGradInput of upsample2 -> conv as ONE GEMM: dy [N,2Hp,2Wp,Cout] -> dx_lo [N,Hp,Wp,Cin], i.e. the
 * gradient w.r.t. the low-res input with SpatialUpSamplingNearest's 2x2 block sum folded in.  wb_ph from
 * cg_pack_conv_weight_ups2.  (Cin, Cout are the FORWARD layer's plane counts.) */
size_t cg_conv2d_dgrad_ups2_workspace_bytes(int N, int Hp, int Wp, int Cin, int Cout, int k, int pad);
int cg_conv2d_dgrad_ups2(void* stream, const float* dy, const float* wb_ph, float* dx_lo,
                         int N, int Hp, int Wp, int Cin, int Cout, int k, int pad,
                         void* ws, size_t ws_bytes);

/* accGradParameters for the weight: gw_canonical += scale * dW, where
 *   dW[co][ci][ky][kx] = sum_{n,oy,ox} X(n,oy+ky-padH,ox+kx-padW,ci) * dy[n,oy,ox,co]
 * (X = x, or its virtual 2x upsampling when ups == 1; then evaluated per phase on the low-res grid).  gw layout is canonical [Cout][Cin][kH][kW]
 * (== [out][in] for Linear).  If gb != NULL, gb[co] += scale * sum dy[..,co] (gradBias) rides along in the same
 * pass over dy.  Deterministic two-stage split-K reduction. */
size_t cg_conv2d_wgrad_workspace_bytes(int N, int Hp, int Wp, int Cin, int Cout,
                                       int kH, int kW, int padH, int padW, int ups);
int cg_conv2d_wgrad(void* stream, const float* x, const float* dy, float* gw_canonical, float* gb,
                    int N, int Hp, int Wp, int Cin, int Cout,
                    int kH, int kW, int padH, int padW, int ups, float scale,
                    void* ws, size_t ws_bytes);

size_t cg_conv2d_wgrad_workspace_bytes_grouped(int ngroups, int N, int Hp, int Wp, int Cin, int Cout,
                                               int kH, int kW, int padH, int padW, int ups);
int cg_conv2d_wgrad_grouped(void* stream, int ngroups, const float* const* x, const float* const* dy,
                            float* const* gw_canonical, float* const* gb, int N, int Hp, int Wp, int Cin, int Cout,
                            int kH, int kW, int padH, int padW, int ups, float scale, void* ws, size_t ws_bytes);
/* Deferred form: runs the GEMM into `ws` and, for layers whose split reduction is a small kernel, QUEUES that reduction on the
 * stream instead of launching it; cg_conv2d_wgrad_flush(stream) then reduces every queued layer in one launch (a dozen ~10 us
 * kernels overlap instead of running back to back).  `ws` must not be reused before the flush - one workspace per layer - and
 * gw / gb are incomplete until then.  Results are bit-identical to the immediate form.  cg_conv2d_wgrad_pending: queue length. */
int cg_conv2d_wgrad_grouped_deferred(void* stream, int ngroups, const float* const* x, const float* const* dy,
                            float* const* gw_canonical, float* const* gb, int N, int Hp, int Wp, int Cin, int Cout,
                            int kH, int kW, int padH, int padW, int ups, float scale, void* ws, size_t ws_bytes);
int cg_conv2d_wgrad_flush(void* stream);
/* The weight-gradient GEMM of cg_conv2d_wgrad alone (split partial sums left in ws, cg_conv2d_wgrad_workspace_bytes; no gradient is
 * written): exported for timing / profiling the kernel in isolation (bench.py's roofline entry), not used by a training step. */
int cg_conv2d_wgrad_gemm(void* stream, const float* x, const float* dy, int N, int Hp, int Wp, int Cin, int Cout,
                         int kH, int kW, int padH, int padW, int ups, void* ws, size_t ws_bytes);
int cg_conv2d_wgrad_pending(void* stream, int* njobs);
/* Up to 16 weight gradients of ONE geometry whose tensors are equally spaced in memory (group g: x + g*x_stride, dy + g*dy_stride,
 * gw + g*gw_stride, strides in floats) as one GEMM launch + one reduction: the 16 Winograd-domain products of the upsample2 -> 5x5
 * layer's accGradParameters (models.lua:217-218; cg_conv2d_ups2_wino_wgrad below), which ran as four 4-group launches before.  No
 * bias gradient.  Up to 36 groups (the 4 x 9 planes of F(2x2,2x2)).  Workspace: cg_conv2d_wgrad_workspace_bytes_strided(ngroups, ...);
 * each size query is the capability check of ITS entry point: the _grouped one returns 0 above 4 groups (separate tensors), this one
 * above 36. */
size_t cg_conv2d_wgrad_workspace_bytes_strided(int ngroups, int N, int Hp, int Wp, int Cin, int Cout,
                                               int kH, int kW, int padH, int padW, int ups);
int cg_conv2d_wgrad_strided(void* stream, int ngroups, const float* x, long x_stride, const float* dy, long dy_stride,
                            float* gw_canonical, long gw_stride, int N, int Hp, int Wp, int Cin, int Cout,
                            int kH, int kW, int padH, int padW, int ups, float scale, void* ws, size_t ws_bytes);

/* gb[c] += scale * sum_m dy[m][c]   (gradBias of conv / linear).
 * ws: scratch of at least 8*C bytes (fp64 column sums). */
int cg_bias_grad(void* stream, const float* dy, float* gb, long M, int C, float scale,
                 void* ws, size_t ws_bytes);

/* canonical [Cout][Cin][kH][kW] -> wf[(ky*kW+kx)*Cin+ci][Cout] (forward) and
 * wb[((kH-1-ky)*kW+(kW-1-kx))*Cout+co][Cin] (updateGradInput; flipped taps).
 * Either output may be NULL. */
int cg_pack_conv_weight(void* stream, const float* w_canonical, float* wf, float* wb,
                        int Cout, int Cin, int kH, int kW);
/* The same for n layers in one launch (host arrays of n entries; wb[i] may be NULL for 1x1 / linear layers): every
 * parameter of a net changes in the same cg_adam_step, so all its layers re-pack together. */
int cg_pack_conv_weight_batch(void* stream, int n, const float* const* w_canonical, float* const* wf, float* const* wb,
                              const int* Cout, const int* Cin, const int* kH, const int* kW, const int* wb_map);
/* nn.View(C*H*W) -> nn.Linear(C*H*W -> Cout) consuming the NHWC map directly (models.lua:696-697, 849-850): the canonical
 * [Cout][C*H*W] weight is the canonical weight of a convolution C -> Cout with an H x W kernel and no padding on the H x W map.
 * wf as cg_pack_conv_weight (forward: cg_conv2d_forward(x_nhwc, wf, ..., N, H, W, C, Cout, H, W, 0, 0, 0); weight gradient:
 * cg_conv2d_wgrad with the same geometry, canonical gw); wbT[co][(h*W+w)*C + c] is the data-gradient operand of the linear
 * form: cg_conv2d_forward(dy, wbT, NULL, dx_nhwc, N, 1, 1, Cout, H*W*C, 1, 1, 0, 0, 0).  wb_map[i] != 0 in the batch form
 * selects this layout for layer i (wb_map may be NULL). */
int cg_pack_conv_weight_map(void* stream, const float* w_canonical, float* wf, float* wbT, int Cout, int Cin, int kH, int kW);
/* phase-summed weights for upsample2 -> conv k x k (pad (k-1)/2), k' = (k+1)/2 rounded up (2 for 3, 3 for 5):
 * wf_ph[p][(t'*Cin+ci)][Cout], wb_ph[((p*k'*k' + t')*Cout+co)][Cin]; each holds
 * cg_pack_conv_weight_ups2_floats() floats.  Either output may be NULL. */
size_t cg_pack_conv_weight_ups2_floats(int Cout, int Cin, int k, int pad);
int cg_pack_conv_weight_ups2(void* stream, const float* w_canonical, float* wf_ph, float* wb_ph,
                             int Cout, int Cin, int k, int pad);

/* ---- Winograd F(2x2,3x3) path for upsample2 -> conv 5x5 pad 2 (models.lua:217-218) ------------------------
 * Each output phase of that layer is a 3x3 convolution of the low-res input (see cg_pack_conv_weight_ups2);
 * the four phases share their input tiles.  u_fwd / u_bwd: Winograd-domain phase kernels (from wf_ph / wb_ph,
 * cg_conv2d_ups2_wino_u_floats() floats each); v: transformed input, cg_conv2d_ups2_wino_v_floats(N,Hp,Wp,C)
 * floats with C = Cin (forward; kept for the weight gradient) or 4*Cout (data gradient scratch).
 * Same results as cg_conv2d_forward(ups=1) / cg_conv2d_dgrad_ups2 up to fp32 re-association. */
size_t cg_conv2d_ups2_wino_supported(int N, int Hp, int Wp, int Cin, int Cout, int k, int pad);
size_t cg_conv2d_ups2_wino_v_floats(int N, int Hp, int Wp, int C);
size_t cg_conv2d_ups2_wino_u_floats(int Cin, int Cout);
int cg_conv2d_ups2_wino_pack(void* stream, const float* wf_ph, const float* wb_ph, float* u_fwd, float* u_bwd,
                             int Cout, int Cin);
/* the 16 GEMMs + in-register output transform alone, on an already transformed input (dgrad: 0 forward, 1 data gradient) */
int cg_conv2d_ups2_wino_gemm(void* stream, const float* v, const float* u, const float* bias, float* y,
                             int N, int Hp, int Wp, int Cin, int Cout, int dgrad);
int cg_conv2d_ups2_wino_forward(void* stream, const float* x_lo, const float* u_fwd, const float* bias, float* y,
                                float* v, int N, int Hp, int Wp, int Cin, int Cout);
/* forward + batch-norm statistics partials in the GEMM epilogue (see cg_conv2d_forward_ex): stats is
 * [cg_conv2d_ups2_wino_stats_rows()][2][Cout] floats, or NULL. */
size_t cg_conv2d_ups2_wino_stats_rows(int N, int Hp, int Wp, int Cin, int Cout);
int cg_conv2d_ups2_wino_forward_stats(void* stream, const float* x_lo, const float* u_fwd, const float* bias, float* y,
                                      float* v, int N, int Hp, int Wp, int Cin, int Cout, float* stats);
/* F(2x2,2x2) for nn.SpatialUpSamplingNearest(2) -> 3x3 convolution (models.lua:211-212 at the full batch; round 4), FORWARD: the four
 * phases are 2x2-tap convolutions with windows one pixel apart - 9 multiplies per 2x2 low-res tile and phase instead of 16.  u22:
 * cg_conv2d_ups2_wino22_u_floats() floats from the phase-summed kernels (cg_pack_conv_weight_ups2's wf_ph); v: scratch of
 * cg_conv2d_ups2_wino22_v_floats() floats (4 phases x 9 planes); stats as
 * cg_conv2d_ups2_wino_forward_stats. */
size_t cg_conv2d_ups2_wino22_supported(int N, int Hp, int Wp, int Cin, int Cout);
/* ... and whether the data-gradient launch below takes the same geometry (its scratch adds limits of its own): ask before choosing
 * the path, fall back to cg_conv2d_dgrad_ups2 otherwise.  (The weight gradient of these layers stays phase-folded: its F(2x2,2x2)-domain
 * form of round 4 measured 6.04 against 6.06 ms per step for 430 MB of workspace and was removed in round 6.) */
size_t cg_conv2d_ups2_wino22_dgrad_supported(int N, int Hp, int Wp, int Cin, int Cout);
size_t cg_conv2d_ups2_wino22_v_floats(int N, int Hp, int Wp, int Cin);
size_t cg_conv2d_ups2_wino22_u_floats(int Cin, int Cout);
int cg_conv2d_ups2_wino22_pack(void* stream, const float* wf_ph, const float* wb_ph, float* u_fwd, float* u_bwd, int Cout, int Cin);
int cg_conv2d_ups2_wino22_forward_stats(void* stream, const float* x_lo, const float* u22, const float* bias, float* y,
                                        float* v, int N, int Hp, int Wp, int Cin, int Cout, float* stats);
/* updateGradInput of the same layers: dx_lo[N][Hp][Wp][Cin] (the upsampling's 2x2 block sum folded in); v_dy: scratch of
 * cg_conv2d_ups2_wino22_dgrad_v_floats() floats; u_bwd from cg_conv2d_ups2_wino22_pack (wb_ph: cg_pack_conv_weight_ups2). */
size_t cg_conv2d_ups2_wino22_dgrad_v_floats(int N, int Hp, int Wp, int Cin, int Cout);
int cg_conv2d_ups2_wino22_dgrad(void* stream, const float* dy, const float* u_bwd, float* dx_lo, float* v_dy,
                                int N, int Hp, int Wp, int Cin, int Cout);
/* updateGradInput of the 5x5 layers in F(2x2,3x3) (cg_conv2d_ups2_wino_forward's layers). */
int cg_conv2d_ups2_wino_dgrad(void* stream, const float* dy, const float* u_bwd, float* dx_lo, float* v_dy,
                              int N, int Hp, int Wp, int Cin, int Cout);
/* The same data gradient with its K rows (the four phases' channels) in 2 or 4 slices over the launch's z dimension and a fixed-order sum
 * of the partial results (round 4: the unsplit launch of G's 5x5 layer is one workgroup per CU).  part: scratch of
 * cg_conv2d_ups2_wino_dgrad_part_floats() floats; 0 floats / part == NULL = the unsplit launch above. */
size_t cg_conv2d_ups2_wino_dgrad_part_floats(int N, int Hp, int Wp, int Cin, int Cout);
int cg_conv2d_ups2_wino_dgrad_split(void* stream, const float* dy, const float* u_bwd, float* dx_lo, float* v_dy, float* part,
                                    int N, int Hp, int Wp, int Cin, int Cout);
/* accGradParameters in the Winograd domain, from the v the forward wrote: gw_canonical[Cout][Cin][5][5] += scale*dW,
 * gb (may be NULL) += scale * sum dy.  16 plain weight-gradient GEMMs (cg_conv2d_wgrad_grouped) + G^T . G. */
size_t cg_conv2d_ups2_wino_wgrad_workspace_bytes(int N, int Hp, int Wp, int Cin, int Cout);
int cg_conv2d_ups2_wino_wgrad(void* stream, const float* v, const float* dy, float* gw_canonical, float* gb,
                              int N, int Hp, int Wp, int Cin, int Cout, float scale, void* ws, size_t ws_bytes);

/* ---- activations --------------------------------------------------------
 * nn.PReLU(nil,nil,true): one shared slope (models.lua:201,208,214,220,647..698).
 * y = x>0 ? x : a*x ; dx = x>0 ? dy : a*dy ; *galpha += scale*sum_{x<=0} x*dy. */
int cg_prelu_forward(void* stream, const float* x, const float* alpha, float* y, long n);
/* galpha may be NULL (updateGradInput only); otherwise ws holds one double per workgroup (deterministic 2-stage sum). */
size_t cg_prelu_backward_workspace_bytes(long n);
int cg_prelu_backward(void* stream, const float* x, const float* dy, const float* alpha,
                      float* dx, float* galpha, float scale, long n, void* ws, size_t ws_bytes);
/* The same over a stacked batch of ngroups (<= 4) equal groups of n_per_group elements, group g with its own slope
 * alpha[g] and gradient accumulator galpha[g] (D32_st3's identical branches, models.lua:653-678): one launch for
 * ngroups PReLU modules. */
size_t cg_prelu_backward_grouped_workspace_bytes(int ngroups, long n_per_group);
int cg_prelu_backward_grouped(void* stream, const float* x, const float* dy, const float* const* alpha, float* dx,
                              float* const* galpha, float scale, int ngroups, long n_per_group, void* ws, size_t ws_bytes);
/* nn.LeakyReLU (LeakyReLU.lua:13-31): y = x>=0 ? x : s*x ; dx = x>=0 ? dy : s*dy. */
int cg_leakyrelu_forward(void* stream, const float* x, float* y, float slope, long n);
int cg_leakyrelu_backward(void* stream, const float* x, const float* dy, float* dx, float slope, long n);
/* nn.Sigmoid (models.lua:223,701): y = 1/(1+exp(-x)); dx = dy*y*(1-y). */
int cg_sigmoid_forward(void* stream, const float* x, float* y, long n);
int cg_sigmoid_backward(void* stream, const float* y, const float* dy, float* dx, long n);

/* nn.BCECriterion (train.lua:181), sizeAverage, eps = 1e-12:
 * *loss = -(1/n) sum t*log(p+eps) + (1-t)*log(1-p+eps) ;
 * dp = -(t-p)/((1-p+eps)*(p+eps))/n. */
int cg_bce_forward(void* stream, const float* p, const float* t, float* loss, long n);
int cg_bce_backward(void* stream, const float* p, const float* t, float* dp, long n);

/* ---- nn.SpatialBatchNormalization, training mode (models.lua:207,213,219) --
 * x,y: [M][C] (M = N*H*W).  Statistics are exchanged as fp64 sums so that a
 * data-parallel host can all-reduce them between the two calls (sync-BN).
 *   cg_bn_stats:   sums[0..C) = sum_m x, sums[C..2C) = sum_m x^2 (overwrites)
 *   cg_bn_forward: mean = s1/count, var = s2/count - mean^2 (biased), eps,
 *                  y = (x-mean)*invstd*gamma + beta; writes save_mean /
 *                  save_invstd; running stats updated with `momentum`
 *                  (running_var with the unbiased variance), NULL to skip. */
int cg_bn_stats(void* stream, const float* x, long M, int C, double* sums);
int cg_bn_forward(void* stream, const float* x, float* y, const float* gamma, const float* beta,
                  const double* sums, double count, long M, int C, float eps, float momentum,
                  float* running_mean, float* running_var, float* save_mean, float* save_invstd);
/*   cg_bn_backward_stats: sums[0..C) = sum dy, sums[C..2C) = sum dy*xhat
 *   cg_bn_backward: dx = gamma*invstd*(dy - s1/count - xhat*s2/count);
 *                   ggamma += scale*s2_local... (uses local_sums for the
 *                   parameter gradients, global `sums` for dx). */
int cg_bn_backward_stats(void* stream, const float* x, const float* dy, const float* save_mean,
                         const float* save_invstd, long M, int C, double* sums);
int cg_bn_backward(void* stream, const float* x, const float* dy, const float* gamma,
                   const float* save_mean, const float* save_invstd,
                   const double* sums, double count, const double* local_sums,
                   long M, int C, float* dx, float* ggamma, float* gbeta, float scale);
/* evaluate() mode: y = (x-running_mean)/sqrt(running_var+eps)*gamma+beta. */
int cg_bn_forward_eval(void* stream, const float* x, float* y, const float* gamma, const float* beta,
                       const float* running_mean, const float* running_var, long M, int C, float eps);

/* ---- fused memory-bound chains (csrc/fused.hip) ---------------------------
 * Same arithmetic per element as the separate entry points above / below, one pass over HBM per chain.
 *
 * Batch statistics from the producing convolution's epilogue (cg_conv2d_forward_ex / cg_conv2d_ups2_wino_forward_stats):
 * sums[0..C) = sum_rows partials[r][0][c], sums[C..2C) = sum_rows partials[r][1][c] (fp64, fixed order) - the same
 * quantities cg_bn_stats produces from a pass over x; a data-parallel host all-reduces `sums` next (sync-BN). */
int cg_bn_stats_finalize(void* stream, const float* partials, long rows, int C, double* sums);
/* The running-statistics update of nn.SpatialBatchNormalization:updateOutput (models.lua:207; momentum 0.1, unbiased variance) ALONE,
 * from the fp64 sums of a pass that was run with running_mean == running_var == NULL: same arithmetic as cg_bn_forward /
 * cg_bn_act_forward, so a deferred update is bit-identical to an in-line one (cg_net_apply_running uses it). */
int cg_bn_running_update(void* stream, const double* sums, double count, int C, float momentum, float* running_mean, float* running_var);
/* cg_bn_forward followed by cg_prelu_forward (models.lua:207-208) in one pass: y = prelu(bn(x)); alpha == NULL: no
 * activation.  The normalised tensor is not materialised: the backward below recomputes it from x. */
int cg_bn_act_forward(void* stream, const float* x, float* y, const float* gamma, const float* beta,
                      const double* sums, double count, long M, int C, float eps, float momentum,
                      float* running_mean, float* running_var, float* save_mean, float* save_invstd, const float* alpha);
/* Backward of y = prelu(bn(x)) given dy = dL/dy, in two passes over (x, dy):
 *   cg_bn_act_backward_stats: with d = prelu'(bn(x)) * dy:  sums[0..C) = sum d, sums[C..2C) = sum d*xhat,
 *                             sums[2C] = sum_{bn(x) <= 0} bn(x) * dy   (cg_bn_act_backward_sums(C) doubles, overwritten)
 *   cg_bn_act_backward:       dx = gamma*invstd*(d - s1/count - xhat*s2/count) from the (all-reduced) `sums`;
 *                             ggamma += scale*local s2, gbeta += scale*local s1, galpha += scale*local sums[2C]. */
size_t cg_bn_act_backward_sums(int C);
int cg_bn_act_backward_stats(void* stream, const float* x, const float* dy, const float* save_mean,
                             const float* save_invstd, const float* gamma, const float* beta, const float* alpha,
                             long M, int C, double* sums);
int cg_bn_act_backward(void* stream, const float* x, const float* dy, const float* gamma, const float* beta,
                       const float* save_mean, const float* save_invstd, const float* alpha,
                       const double* sums, double count, const double* local_sums, long M, int C,
                       float* dx, float* ggamma, float* gbeta, float* galpha, float scale);
/* activation -> 2x2 pooling -> spatial dropout in one pass (models.lua:649-651, 656-658, 682-684, 847-848):
 *   y[n,oy,ox,c] = mask[n,c] * pool( act(x[n, 2oy+{0,1}, 2ox+{0,1}, c]) ),   x: [ngroups*n_per_group, H, W, C].
 * act: 0 none, 1 PReLU (alpha[g] = slope pointer of group g; samples [g*n_per_group, (g+1)*n_per_group) use it),
 * 2 LeakyReLU(slope).  pool_max: 0 average, 1 maximum (first maximum in scan order takes the gradient).
 * mask [ngroups*n_per_group][C] or NULL.  C % 4 == 0, even H and W.
 * backward: dx = act'(x) * pool^T(mask * gy); galpha[g] += scale * sum_{x<=0} x * d (PReLU; galpha or entries may be
 * NULL); ws: cg_act_pool2_mask_backward_workspace_bytes(). */
int cg_act_pool2_mask_forward(void* stream, const float* x, float* y, const float* mask, int ngroups, int n_per_group,
                              int H, int W, int C, int act, float slope, const float* const* alpha, int pool_max);
size_t cg_act_pool2_mask_backward_workspace_bytes(int ngroups, int n_per_group, int H, int W, int C);
int cg_act_pool2_mask_backward(void* stream, const float* x, const float* gy, const float* mask, float* dx,
                               int ngroups, int n_per_group, int H, int W, int C, int act, float slope,
                               const float* const* alpha, float* const* galpha, float scale, int pool_max,
                               void* ws, size_t ws_bytes);

/* ---- resampling / pooling (NHWC) ---------------------------------------- */
/* nn.SpatialUpSamplingNearest(2) materialised (only when not folded into a conv);
 * H,W are the low-res dims.  backward sums each 2x2 block. */
int cg_upsample2x_forward(void* stream, const float* x, float* y, int N, int H, int W, int C);
int cg_upsample2x_backward(void* stream, const float* dy, float* dx, int N, int H, int W, int C);
/* nn.SpatialAveragePooling(2,2,2,2) / nn.SpatialMaxPooling(2,2) (models.lua:650,657,843,848);
 * H,W are input dims (even). */
int cg_avgpool2_forward(void* stream, const float* x, float* y, int N, int H, int W, int C);
int cg_avgpool2_backward(void* stream, const float* dy, float* dx, int N, int H, int W, int C);
int cg_maxpool2_forward(void* stream, const float* x, float* y, int N, int H, int W, int C);
int cg_maxpool2_backward(void* stream, const float* x, const float* dy, float* dx, int N, int H, int W, int C);

/* ---- dropout (models.lua:651,658,...,695,699) ---------------------------
 * y = x * mask.  spatial!=0: mask is [N][C], broadcast over HW (nn.SpatialDropout);
 * else mask has x's shape (nn.Dropout).  Used for forward and backward. */
int cg_mask_mul(void* stream, const float* x, const float* mask, float* y,
                int N, long HW, int C, int spatial);
/* counter-based generator (own design; TH's MT19937 cannot be reproduced):
 * out[i] = (u(seed,offset+i) < keep_prob) ? value : 0 ; uniform in [lo,hi). */
int cg_rng_bernoulli(void* stream, float* out, long n, float keep_prob, float value,
                     uint64_t seed, uint64_t offset);
int cg_rng_uniform(void* stream, float* out, long n, float lo, float hi,
                   uint64_t seed, uint64_t offset);
/* out[i] = floor(u * range) as int32: math.random(trainData:size()) - 1 (adversarial.lua:226). */
int cg_rng_randint(void* stream, int32_t* out, long n, int32_t range, uint64_t seed, uint64_t offset);
/* Replay-safe variants for hipGraph capture: the stream position is offset + *base, where *base is a device-side
 * counter the host advances once per step with cg_counter_add (kernel arguments are frozen under graph replay). */
int cg_rng_bernoulli_dev(void* stream, float* out, long n, float keep_prob, float value,
                         uint64_t seed, uint64_t offset, const uint64_t* base);
int cg_rng_uniform_dev(void* stream, float* out, long n, float lo, float hi,
                       uint64_t seed, uint64_t offset, const uint64_t* base);
int cg_rng_randint_dev(void* stream, int32_t* out, long n, int32_t range, uint64_t seed, uint64_t offset,
                       const uint64_t* base);
/* ngroups (<= 4) consecutive blocks of n_per_group floats, block g drawn at stream position off<g> (+ *base): the
 * spatial-dropout masks of D32_st3's identical branches (models.lua:658) in one launch. */
int cg_rng_bernoulli_dev_grouped(void* stream, float* out, long n_per_group, int ngroups, float keep_prob, float value,
                                 uint64_t seed, uint64_t off0, uint64_t off1, uint64_t off2, uint64_t off3,
                                 const uint64_t* base);
int cg_counter_add(void* stream, uint64_t* counter, uint64_t delta);

/* ---- layout / data movement --------------------------------------------- */
int cg_nchw_to_nhwc(void* stream, const float* in, float* out, int N, int C, int H, int W);
int cg_nhwc_to_nchw(void* stream, const float* in, float* out, int N, int C, int H, int W);
/* dst[m][dst_off + c] = src[m][src_off + c], c in [0,Ccopy): nn.Concat(2) fwd/bwd
 * (models.lua:688-692). */
int cg_copy_channels(void* stream, const float* src, float* dst, long M,
                     int Csrc, int src_off, int Cdst, int dst_off, int Ccopy);
/* nn.Concat(2) over n <= 4 branches in one launch: dst[m][off_b + c] = src[b][m][c] (C[b] channels each, host array,
 * all % 4); cg_split_channels is the reverse (the gradient slices handed to the branches); cg_sum_n: out = ((s0+s1)+s2)+s3,
 * the accumulation order of nn.Concat's gradInput. */
int cg_concat_channels(void* stream, int n, const float* const* src, const int* C, float* dst, long M);
int cg_split_channels(void* stream, int n, const float* src, float* const* dst, const int* C, long M);
int cg_sum_n(void* stream, int n, const float* const* src, float* out, long count);
/* nn.Concat(2) followed by nn.SpatialDropout in training mode (models.lua:688-693) in one launch: the mask element of sample
 * s, channel c is drawn inside the kernel from draw offset + s*Ct + c of the counter stream (the value cg_rng_bernoulli_dev
 * puts there) and stored in noise[N][Ct]; dst = concat * mask.  cg_split_channels_masked is the backward: the branches' gradient
 * slices from the gradient w.r.t. the dropped tensor, dst[b][m][c] = src[m][off_b + c] * noise[s][off_b + c]. */
int cg_concat_channels_dropout(void* stream, int n, const float* const* src, const int* C, float* dst, float* noise, int N, long HW,
                               float keep_prob, float value, uint64_t seed, uint64_t offset, const uint64_t* base);
int cg_split_channels_masked(void* stream, int n, const float* src, const float* noise, float* const* dst, const int* C, int N, long HW);
/* The discriminator's head nn.Dropout -> nn.Linear(F, O) -> nn.Sigmoid (models.lua:699-701) for O <= 4, one launch each way.
 * forward: noise[N][F] drawn in place (element i = draw offset + i), xd = x * noise, z = xd w^T + b (w canonical [O][F]), p = sigmoid(z).
 * backward from dp = dLoss/dp: gx = noise * ((dp p (1-p)) w); with gw != NULL also gw += scale * gz^T xd, gb += scale * colsum(gz)
 * (accGradParameters; NULL = updateGradInput only, the G-step of adversarial.lua:155-167). */
int cg_drop_linear_sigmoid_supported(int N, int F, int O);
int cg_drop_linear_sigmoid_forward(void* stream, const float* x, const float* w, const float* b, float* noise, float* xd, float* z, float* p,
                                   int N, int F, int O, float keep_prob, float value, uint64_t seed, uint64_t offset, const uint64_t* base);
int cg_drop_linear_sigmoid_backward(void* stream, const float* dp, const float* p, const float* xd, const float* noise, const float* w,
                                    float* gx, float* gw, float* gb, int N, int F, int O, float scale);
/* dst[i] = src[idx[i]] for rows of rowlen floats: D-batch assembly from the
 * real-image pool (adversarial.lua:225-230). */
int cg_gather_rows(void* stream, const float* src, const int32_t* idx, float* dst,
                   long nrows, long rowlen);
int cg_fill(void* stream, float* x, float value, long n);
int cg_add(void* stream, const float* a, const float* b, float* out, long n); /* out = a+b */
int cg_axpy(void* stream, float alpha, const float* x, float* y, long n);    /* y += alpha*x */
int cg_axpy_sign(void* stream, float alpha, const float* x, float* y, long n); /* y += alpha*sign(x) (adversarial.lua:97) */
int cg_scale(void* stream, float* x, float alpha, long n);
int cg_clamp(void* stream, float* x, float lo, float hi, long n);
/* *out (double) = sum x^2 / sum |x| : torch.norm(p,2)^2, torch.norm(p,1)
 * (adversarial.lua:94-95). */
int cg_sumsq(void* stream, const float* x, long n, double* out);
int cg_sumabs(void* stream, const float* x, long n, double* out);

/* ---- spatial transformer (stn rock; models.lua:877-878,888) --------------
 * AffineTransformMatrixGenerator(rot,scale,trans): params[N][P] consumed in the
 * order [theta][s][tx,ty]; T = R(theta)*S(s)*Tr(tx,ty), top two rows -> T[N][2][3],
 * acting on (y,x,1).  R = [[c,-s],[s,c]]. */
int cg_affine_matrix_forward(void* stream, const float* params, float* T, int N,
                             int use_rot, int use_scale, int use_trans);
int cg_affine_matrix_backward(void* stream, const float* params, const float* gT, float* gparams,
                              int N, int use_rot, int use_scale, int use_trans);
/* AffineGridGeneratorBHWD(H,W): grid[n,i,j,:] = T_n * (y_i, x_j, 1), y_i = -1+2i/(H-1). */
int cg_affine_grid_forward(void* stream, const float* T, float* grid, int N, int H, int W);
int cg_affine_grid_backward(void* stream, const float* ggrid, float* gT, int N, int H, int W);
/* BilinearSamplerBHWD: img[N,Hi,Wi,C], grid[N,Ho,Wo,2] (y,x) in [-1,1], corner aligned,
 * out-of-range taps contribute 0.  backward overwrites gimg and ggrid.  The backward is deterministic: gimg is GATHERED per
 * source pixel over the output pixels whose footprint covers it, in a fixed order (the reference keeps this module on the
 * CPU because stn's GPU scatter was non-reproducible, models.lua:889-899); CG_SAMPLER_ATOMICS=1 selects the atomic scatter
 * (also used beyond 8192 output pixels per image or 256 channels). */
int cg_bilinear_sampler_forward(void* stream, const float* img, const float* grid, float* out,
                                int N, int Hi, int Wi, int C, int Ho, int Wo);
int cg_bilinear_sampler_backward(void* stream, const float* img, const float* grid, const float* gout,
                                 float* gimg, float* ggrid,
                                 int N, int Hi, int Wi, int C, int Ho, int Wo);
/* `ngroups` sibling transformers sampling the SAME images [N,Hi,Wi,C] with their own grids (D32_st3's three branches,
 * models.lua:655-686): grid / out / gout / gimg / ggrid are the branch-major stacks of ngroups * N samples, one launch. */
int cg_bilinear_sampler_forward_shared(void* stream, int ngroups, const float* img, const float* grid, float* out,
                                       int N, int Hi, int Wi, int C, int Ho, int Wo);
int cg_bilinear_sampler_backward_shared(void* stream, int ngroups, const float* img, const float* grid, const float* gout,
                                        float* gimg, float* ggrid, int N, int Hi, int Wi, int C, int Ho, int Wo);

/* ---- the localisation network of a spatial transformer in one launch each way (csrc/locnet.hip) ----------------------
 * models.lua:842-860: AvgPool(2) -> conv3x3 Cin->16 -> LeakyReLU -> conv3x3 16->16 -> LeakyReLU -> AvgPool(2) -> View ->
 * Linear(16 (S/2)^2 -> 64) -> LeakyReLU -> Linear(64 -> P), then models.lua:877-878: AffineTransformMatrixGenerator ->
 * AffineGridGeneratorBHWD.  One workgroup per sample keeps the chain's activations in LDS; per element the arithmetic is that of
 * the separate entry points up to fp32 re-association of the convolution / linear sums.  S = spatial size after the first
 * pooling (the transformer's input is [.., 2S, 2S, Cin] NHWC), a power of two >= 8; cg_locnet_supported() == 0: use the modules.
 * `ngroups` (<= 4) networks of identical shape run in one launch, group g on samples [g*n, (g+1)*n) (x_shared != 0: every group
 * reads the SAME n input samples - D32_st3's three branches all look at the trunk's output, models.lua:653-678).
 * weights: host array of 12 device pointers per group: canonical Torch7 tensors conv1 weight [16][Cin][3][3], conv1 bias,
 * conv2 weight [16][16][3][3], conv2 bias, linear1 weight [64][16 (S/2)^2] (input index in (c, y, x) order, as nn.View leaves it),
 * linear1 bias, linear2 weight [P][64], linear2 bias, then cg_pack_conv_weight's copies wf, wb of conv1 and wf, wb of conv2.
 * forward outputs: grid [G n][Hg][Wg][2] (what cg_affine_grid_forward writes) and the activations the backward and the weight
 * gradients need: pooled [G n][S][S][Cin], h1 = LeakyReLU(conv1) [G n][S][S][16], m2 = LeakyReLU(conv2) [G n][S][S][16],
 * h2 = its 2x2 average in (c, y, x) order [G n][16 (S/2)^2], h3 = LeakyReLU(linear1) [G n][64], params [G n][P].
 * backward: ggrid -> gx [G n][2S][2S][Cin] (gradient w.r.t. the transformer's input through the localisation branch) and the
 * gradients w.r.t. the pre-activations: g4 [G n][P], g3 [G n][64], ga2 / ga1 [G n][S][S][16] - the weight gradients are
 * cg_conv2d_wgrad(pooled, ga1), (h1, ga2), (h2, g3) and (h3, g4) with the layers' geometries. */
int cg_locnet_supported(int S, int Cin, int P);
int cg_locnet_forward(void* stream, int ngroups, int n_per_group, const float* x, int x_shared, const float* const* weights, int S, int Cin,
                      int P, int use_rot, int use_scale, int use_trans, float slope, int Hg, int Wg, float* pooled, float* h1, float* m2,
                      float* h2, float* h3, float* params, float* grid);
int cg_locnet_backward(void* stream, int ngroups, int n_per_group, const float* const* weights, int S, int Cin, int P, int use_rot,
                       int use_scale, int use_trans, float slope, int Hg, int Wg, const float* h1, const float* m2, const float* h3,
                       const float* params, const float* ggrid, float* ga1, float* ga2, float* g3, float* g4, float* gx);

/* ---- collectives of the data-parallel step (csrc/comm.hip; RCCL over xGMI) ---------------------------------
 * The reference is single-GPU (train.lua:108-112).  The step shards on the batch: one process per GPU,
 * parameters replicated, and two exchanges per update:
 *   - all-reduce(average) of the flat GRAD_PARAMETERS vector - or a contiguous bucket of it - right after backward and
 *     BEFORE penalty / clamp / Adam (adversarial.lua:89-112), started asynchronously so that it travels under the next
 *     kernels; cg_comm_wait joins it into the compute stream just before cg_adam_step;
 *   - sync-BN: all-reduce(sum) of the fp64 sums between cg_bn_stats / cg_bn_stats_finalize and cg_bn_forward /
 *     cg_bn_act_forward, and of the backward sums.  Use a second communicator for these (tiny, latency-bound) so they
 *     never queue behind a gradient bucket.
 * A communicator owns one side HIP stream and two events: cg_comm_allreduce forks from `compute_stream` (everything
 * enqueued there so far is visible to the collective), runs the collective on the side stream and returns at once;
 * cg_comm_wait makes `compute_stream` wait (device-side) for everything enqueued on the communicator.  An exchange of at most
 * 16 KB (the sync-BN sums) runs on `compute_stream` itself - it is joined at once, the side stream would only add two event hops -
 * and cg_comm_wait then has nothing to do.  Collectives of different communicators are ordered on the device.
 * Bootstrapping: rank 0 calls cg_comm_unique_id and ships the CG_COMM_ID_BYTES bytes to the other ranks by any host
 * channel (file, socket, MPI ...); every rank then calls cg_comm_init with its own GPU current (cg_set_device).
 * dtype: 0 fp32, 1 fp64.  op: 0 sum, 1 average.  *available == 0: librccl.so.1 could not be loaded. */
#define CG_COMM_ID_BYTES 128
int cg_comm_available(int* available);
int cg_comm_unique_id(void* id_out, size_t id_bytes);
int cg_comm_init(void** comm, int nranks, int rank, const void* unique_id, size_t id_bytes);
int cg_comm_destroy(void* comm);
int cg_comm_size(void* comm, int* nranks, int* rank);
int cg_comm_version(int* version);   /* RCCL's ncclGetVersion code (e.g. 22606), 0 when RCCL is not loadable */
int cg_comm_allreduce(void* comm, void* compute_stream, void* buf, size_t count, int dtype, int op);
int cg_comm_broadcast(void* comm, void* compute_stream, void* buf, size_t count, int dtype, int root);
int cg_comm_wait(void* comm, void* compute_stream);
int cg_comm_sync(void* comm);

/* ---- planned executor (csrc/net.hip) ----------------------------------------------------------------------------
 * Where the reference calls MODEL_X:forward / :backward / :updateGradInput on a whole network (adversarial.lua:84-89,
 * 182-197; utils/nn_utils.lua:52,95) the host calls cg_net_forward / cg_net_backward on a `net` it described once, module by
 * module, with the constructor calls of models.lua (:138-160, 196-228, 640-711, 814-906).  The library plans the pass:
 * fused segments of nn.Sequential (conv|linear -> PReLU|LeakyReLU in the GEMM epilogue; activation -> 2x2 pooling ->
 * SpatialDropout in one pass; conv -> SpatialBatchNormalization (training) -> PReLU with the statistics in the GEMM epilogue;
 * nn.View -> nn.Linear on the NHWC map; nn.Concat -> nn.SpatialDropout; nn.Dropout -> nn.Linear -> nn.Sigmoid; a spatial transformer's
 * whole localisation branch), lockstep execution of identical nn.Concat branches (grouped GEMM launches, stacked
 * parameter-free layers, shared pooling / sampling), the other branch group on a side stream, deferred + batched weight-
 * gradient reductions, one batched weight re-pack per parameter update, sync-BN and gradient-bucket collectives in place.
 * Per-element arithmetic is that of the per-module entry points above; every buffer is allocated when a (net, input shape)
 * pair is first compiled, so later passes only launch (and can be captured, cg_graph_*).
 *
 * cg_net_add: `kind` and its arguments (iargs / fargs), mirroring the reference constructors:
 *   0 nn.Sequential()                      1 nn.Concat(dim): iargs {2}               2 nn.ConcatTable()
 *   3 nn.Linear(in, out): {in, out}        4 nn|cudnn.SpatialConvolution(nIn,nOut,kW,kH,dW,dH,padW,padH): {nIn,nOut,kW,kH,padW,padH,dW,dH}
 *   5 nn.PReLU()                           6 nn.LeakyReLU(s): fargs {s}              7 nn.Sigmoid()
 *   8 nn.SpatialBatchNormalization(n): {n}, fargs {eps, momentum}                    9 nn.View(sizes...): {sizes} (1 or 3)
 *  10 nn.Copy (identity on the device)    11 nn.Transpose: {0} NCHW->BHWD (models.lua:870), {1} BHWD->NCHW (:903)
 *  12 nn.SpatialUpSamplingNearest(2)      13 nn.SpatialAveragePooling(2,2,2,2)      14 nn.SpatialMaxPooling(2,2)
 *  15 nn.SpatialDropout(p): fargs {p}     16 nn.Dropout(p): fargs {p}
 *  17 nn.AffineTransformMatrixGenerator(rot,scale,trans): {rot,scale,trans}          18 nn.AffineGridGeneratorBHWD(H,W): {H,W}
 *  19 nn.BilinearSamplerBHWD()
 * parent: id of the container the module is :add()-ed to; the first module (parent -1) is the root.  *id: the new module.
 * cg_net_bind: the device tensors Module:getParameters() (train.lua:184-185) left in the module - slot 0 weight / gradWeight,
 *   1 bias / gradBias, 2 running_mean / running_var of a batch-norm.  Call again whenever the host re-points them.
 * cg_net_params_changed: the parameters moved (optimiser step, checkpoint load): re-pack before the next pass.
 * cg_net_set_training: module:training() / :evaluate() (utils/nn_utils.lua:334-349); id -1 = every module.
 * cg_net_set_option: "overlap_groups", "defer_wgrad", "winograd", "winograd_min_tiles", "share_pool", "sampler_shared",
 *   "view_fuse", "cat_fuse", "stacking", "grouped", "fusion", "fuse_locnet", "head_fuse", "pack_overlap" (0/1: ablation switches,
 *   results unchanged up to fp32 re-association; fuse_locnet: a spatial transformer's localisation branch as one launch each way,
 *   cg_locnet_*; head_fuse: nn.Concat -> nn.SpatialDropout and nn.Dropout -> nn.Linear(., <= 4) -> nn.Sigmoid as one launch each,
 *   masks drawn inside; pack_overlap: after a parameter update the weights of all but the first layer behind a folded upsampling
 *   are re-packed on a side stream beside the head of the forward pass); "trace" 1 (before the first pass): launches go to recording stubs instead of the GPU, cg_net_trace_take
 *   returns the text (one `call|<entry point>|<args>` line per launch; pointers as r<region>+<offset>, regions = the plan's
 *   allocations in order plus what cg_net_trace_region registered) - how the planner is tested without a GPU.
 * cg_net_set_allocator: device memory for the plan's buffers from the host's allocator (must return zeroed memory, owned by
 *   the host); default cg_malloc-style memory owned by the net.
 * cg_net_set_dp: data parallelism: world size, sync-BN on/off, the two cg_comm_* communicators (sync-BN sums /
 *   gradient buckets; NULL: the host hook carries the exchange), bucket_overlap != 0: cg_net_backward starts the all-reduce
 *   (average) of each gradient bucket of the root nn.Sequential (a convolution / linear layer with the parameters up to the next one;
 *   buckets below 256 KB ride with the next) once its backward is complete and the next sync-BN exchange in front of it has been
 *   issued (join: cg_comm_wait).
 *   bucket_overlap & 2: BOTH transports - the communicator's collective, then the host hook on the same buffer (functional tests
 *   with single-rank communicators on one GPU: the cg_comm_* fork / join path runs, the cross-rank sum travels over the hook).
 * cg_net_set_hook: host transport for those exchanges when no communicator is set: hook(user, what, buf, count, dtype, stream),
 *   what 0 = in-place SUM of `count` elements (dtype as cg_comm_allreduce) ordered on `stream`, 1 = start the averaging
 *   all-reduce of a gradient bucket (the host finishes them after cg_net_backward).
 *
 * cg_net_forward: x = the input ([dims], fmt 0 row-major / 1 NHWC); rng_seed / rng_offset / rng_base: position of the counter
 *   stream (cg_rng_*_dev semantics) the pass's dropout masks are drawn from, in module order; *draws = how many it consumed.
 *   Returns the output tensor (owned by the net, valid until the next pass at this shape).
 * cg_net_backward: continues the most recent forward.  gy = gradOutput in layout gy_fmt; acc != 0: Module:backward (gradInput +
 *   accGradParameters with `scale`), acc == 0: Module:updateGradInput only (fevalG_on_D's pass through D, adversarial.lua:192-193).
 *   gfmt: 0 row-major, 1 NHWC.
 * cg_net_module_state: .output (which 0) / .gradInput (1) / dropout mask (2) of one module after a pass (NULL when fused away).
 * cg_net_set_option "defer_running" 1 (a RUN-time switch, no re-planning): training-mode batch-norm layers of the following forward
 *   passes leave running_mean / running_var alone; cg_net_apply_running(net, stream) then applies the update of the most recent
 *   such pass (cg_bn_running_update per layer, bit-identical to the in-line update).  For a host that runs two forward passes of one
 *   net side by side on two streams - the generator's fake-image pass and the G-step's pass, adversarial.lua:232-233 and :185, which
 *   read the same parameters - and still wants the running statistics moved in the reference's order.  A pass that did not re-pack
 *   the weights itself waits (device-side) for a re-packing another pass of the net has in flight.
 * cg_net_forward_pair / cg_net_pair_join (round 6): that schedule as ONE call, below the ABI.  Pass 1 (x) runs on `stream` exactly as
 *   cg_net_forward would and its output is returned; pass 2 (x2: already enqueued on `stream` when the call is made) runs on a library
 *   stream of another hardware queue, leaves the running statistics to a deferred update the library applies behind pass 1's, and
 *   becomes the plan cg_net_backward continues.  Results, running statistics and counter-stream draws (*draws = both passes') are those
 *   of cg_net_forward(x) followed by cg_net_forward(x2) - adversarial.lua:232-233 then :185; MODEL_G moves only at :262.
 *   cg_net_pair_join makes `stream` wait for pass 2 and returns ITS output; exactly one join per pair, before the net's next pass. */
int cg_net_create(void** net);
int cg_net_destroy(void* net);
int cg_net_set_option(void* net, const char* name, long value);
typedef void* (*cg_alloc_fn)(void* user, size_t bytes);
typedef int (*cg_hook_fn)(void* user, int what, void* buf, size_t count, int dtype, void* stream);
int cg_net_set_allocator(void* net, cg_alloc_fn alloc, void* user);
int cg_net_set_hook(void* net, cg_hook_fn hook, void* user);
int cg_net_set_dp(void* net, int world, int sync_bn, void* comm_bn, void* comm_grad, int bucket_overlap);
int cg_net_add(void* net, int parent, int kind, const long* iargs, int niargs, const float* fargs, int nfargs, int* id);
int cg_net_bind(void* net, int id, int slot, float* param, float* grad);
int cg_net_set_training(void* net, int id, int train);
int cg_net_params_changed(void* net);
int cg_net_forward(void* net, void* stream, const float* x, int nd, const long* dims, int fmt, uint64_t rng_seed, uint64_t rng_offset,
                   const uint64_t* rng_base, uint64_t* draws, float** y, int* ynd, long* ydims, int* yfmt);
int cg_net_backward(void* net, void* stream, const float* x, const float* gy, int gy_fmt, int acc, float scale, float** gx, int* gnd,
                    long* gdims, int* gfmt);
int cg_net_apply_running(void* net, void* stream);
int cg_net_forward_pair(void* net, void* stream, const float* x, int nd, const long* dims, int fmt, const float* x2, int nd2, const long* dims2,
                        int fmt2, uint64_t rng_seed, uint64_t rng_offset, const uint64_t* rng_base, uint64_t* draws, float** y, int* ynd,
                        long* ydims, int* yfmt);
int cg_net_pair_join(void* net, void* stream, float** y, int* ynd, long* ydims, int* yfmt);
int cg_net_buckets(void* net, int* nbuckets);
int cg_net_module_state(void* net, int id, int which, float** ptr, int* nd, long* dims, int* fmt);
int cg_net_stats(void* net, long* nprograms, long* nlaunch_fwd, long* nlaunch_bwd, size_t* bytes);
int cg_net_trace_region(void* net, const void* base, size_t bytes);
int cg_net_trace_take(void* net, char* out, size_t cap, size_t* len);

/* Whole-iteration replay (the reference pays ~10^3 Lua -> C dispatches per iteration; adversarial.lua:51-275): between
 * cg_graph_begin and cg_graph_end every launch the host makes on `stream` (and on streams forked from it through events:
 * cg_net_* side streams, cg_stream_wait_event) is recorded into a hipGraph instead of executed; cg_graph_launch replays it.
 * Everything that varies per step must live in device memory (cg_rng_*_dev's base counter, cg_adam_step_dev's step count). */
int cg_graph_begin(void* stream);
int cg_graph_end(void* stream, void** graph_exec);
int cg_graph_launch(void* graph_exec, void* stream);
int cg_graph_destroy(void* graph_exec);

/* ---- optimiser ----------------------------------------------------------
 * Fuses adversarial.lua:92-98 (L1/L2 penalty on the gradient), :110-112
 * (clamp) and optim.adam (adversarial.lua:245,262; Torch7 form: eps is added
 * to sqrt(v) without bias-correcting v) into one pass over the flat vectors:
 *   g' = clamp(g + l1*sign(p) + l2*p, -clamp, +clamp)   (clamp<=0: no clamp)
 *   m = b1*m + (1-b1)*g' ; v = b2*v + (1-b2)*g'^2
 *   p -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps)
 * t is the 1-based step count.  If write_back_grad, g is overwritten with g'
 * (what GRAD_PARAMETERS holds after fevalD returns). */
int cg_adam_step(void* stream, float* p, float* g, float* m, float* v, long n,
                 float lr, float beta1, float beta2, float eps, int t,
                 float l1, float l2, float clamp, int write_back_grad);
/* Same update with the step count read from device memory (t = (int)*t_dev, a counter the host advances with
 * cg_counter_add before the call), so the launch can be replayed from a hipGraph. */
int cg_adam_step_dev(void* stream, float* p, float* g, float* m, float* v, long n,
                     float lr, float beta1, float beta2, float eps, const uint64_t* t_dev,
                     float l1, float l2, float clamp, int write_back_grad);

/* optim.sgd (adversarial.lua:240,257; train.lua:201-204: learningRate, momentum; Torch7 form, no Nesterov, no
 * dampening change: v = mom*v + (1-damp)*g with damp = mom on the first call semantics folded by the host):
 *   g' = penalty/clamp as in cg_adam_step;  v = momentum*v + (1-dampening)*g' (momentum != 0);  p -= lr * (v or g').
 * optim.adagrad (train.lua:193-196):  var += g'^2 ;  p -= lr * g' / (sqrt(var) + 1e-10). */
int cg_sgd_step(void* stream, float* p, float* g, float* v, long n, float lr, float momentum, float dampening,
                float l1, float l2, float clamp, int write_back_grad);
int cg_adagrad_step(void* stream, float* p, float* g, float* var, long n, float lr,
                    float l1, float l2, float clamp, int write_back_grad);

/* counts[pred*2 + target] += 1 with pred = out>0.5 (adversarial.lua:101-106). */
int cg_confusion_update(void* stream, const float* outputs, const float* targets,
                        int32_t* counts, long n);

#ifdef __cplusplus
}
#endif
#endif /* CATGAN_H */
