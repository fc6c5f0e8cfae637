/* nind_hip.h -- C ABI of libnind_hip.so: the MI355X (gfx950) native tiled-denoise hot path.
 *
 * The reference (esq4/nind-denoise) is pure Python on torch and has NO native ABI; every entry point
 * below therefore cites the reference Python interface it stands in for (paths relative to
 * /root/reference/src/nind_denoise/).  The host side (the nind_denoise_amd Python package) binds these with ctypes;
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success and a negative nd_status on failure; nd_last_error() returns a
 *     thread-local human-readable message for the last failure on the calling thread.
 *   - device pointers are plain `void*`/`float*` into HBM owned by the caller (torch's caching allocator);
 *     the library borrows them for the duration of the call, allocates nothing and frees nothing.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, no hidden synchronisation.
 *   - images are float32 CHW (RGB), tiles are float32 NCHW, exactly as in the reference.
 */
#ifndef NIND_HIP_H
#define NIND_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum nd_status {
    ND_OK = 0,
    ND_EINVAL = -1,   /* bad argument (e.g. cs not of the form 16k+56 for UtNet) */
    ND_ENOMEM = -2,   /* workspace / output buffer too small                     */
    ND_EHIP = -3      /* a HIP runtime call or kernel launch failed              */
} nd_status;

typedef enum nd_act {     /* UtNet.py:16-26 */
    ND_ACT_NONE = 0,
    ND_ACT_PRELU = 1,     /* one learned scalar slope per activation layer */
    ND_ACT_ELU = 2,
    ND_ACT_HARDSWISH = 3
} nd_act;

typedef enum nd_layer_kind {
    ND_CONV3 = 0,         /* nn.Conv2d(k=3, valid)            UtNet.py:29..54          */
    ND_CONVT3 = 1,        /* nn.ConvTranspose2d(k=3, s=1)     UtNet.py:56,61,63,...    */
    ND_CONVT2S2 = 2,      /* nn.ConvTranspose2d(k=2, s=2)     UtNet.py:59,66,73,80     */
    ND_CONV1 = 3,         /* nn.Conv2d(k=1)                   UtNet.py:86              */
    ND_CONV2S2 = 4        /* nn.Conv2d(k=2, stride=2): the data gradient of ConvTranspose2d(2, s=2) (training step) */
} nd_layer_kind;

typedef enum nd_dtype {
    ND_F32 = 0,           /* fp32 storage, fp32 MFMA (v_mfma_f32_32x32x2_f32), exact-fp32 products            */
    ND_BF16 = 1,          /* bf16 storage of activations + weights, fp32 accumulate (v_mfma_f32_32x32x16_bf16) */
    ND_F16 = 2            /* fp16 storage of activations + weights, fp32 accumulate (v_mfma_f32_32x32x16_f16)  */
} nd_dtype;

/* Per-call arithmetic flags (bit set; 0 = the default path).  There is no process-wide switch: two calls with different
 * flags may run concurrently on different streams, also from different host threads (the launchers' per-device caches -- CU
 * count, raised dynamic-LDS limits -- are atomics; nd_last_error is thread-local). */
typedef enum nd_flags {
    ND_FLAG_NO_SPLITK = 1,    /* keep every output tile whole: no split-K tail, so a tile's bits do not depend on which other
                                 tiles share its launch (the default splits the K loop of a launch's last, partial round of
                                 workgroups over the idle CUs: deterministic, but fp32 sums re-associate by <= 1e-5)        */
    ND_FLAG_DIRECT_CONV = 2,  /* direct convolution on every 3x3 layer (default on the fp32 path: Winograd F(6x6,3x3) from
                                 128 channels up, 1-D F(4,3) inside the implicit-GEMM kernel below; ~1e-5 re-association)   */
    ND_FLAG_W1D_REGS = 4,     /* A/B switch: the 1-D F(4,3) layers through the kernel that transforms in registers (conv_w1d)
                                 instead of the one that shares the transform through LDS (conv_w2d, the default)            */
    ND_FLAG_UNFUSED_POOL = 16, /* A/B switch: every MaxPool2d(2) as its own kernel (default: written from the producing layer's
                                 epilogue wherever its kernel can -- identical values)                                        */
    ND_FLAG_FULL_TILES = 8    /* nd_utnet_denoise_tiles / nd_utnet_profile_stack: compute every layer on the whole tile, as
                                 UtNet.forward does.  Default there: the last decoder levels compute only the pixels that the
                                 useful crop [pad, cs - pad) of a tile can reach (denoise_image.py:249-258 discards the rest of
                                 the network output before the canvas +=) -- same canvas, 19 % less work at cs 264 / ucs 200    */
} nd_flags;

int nd_version(void);   /* 103 = this header */
const char *nd_last_error(void);

/* ---------------------------------------------------------------- tile geometry (host, pure integer)
 * OneImageDS.__init__   denoise_image.py:100-104  -> nd_tile_grid
 * OneImageDS.__getitem__ denoise_image.py:131-143,172-173 -> nd_tile_geom
 */
int nd_tile_grid(int width, int height, int cs, int ucs, int ol, int *cols, int *rows, int *pad);
int nd_tile_geom(int i, int width, int height, int cs, int ucs, int ol,
                 int *x0, int *y0, int usefuldim[4], int usefulstart[2]);

/* ---------------------------------------------------------------- device tiler
 * nd_tile_gather: OneImageDS.__getitem__ (denoise_image.py:138-170) for tiles [tile_begin, tile_begin+tile_count):
 *   crop + symmetric mirror padding, img_chw [3,H,W] -> tiles_nchw [tile_count,3,cs,cs].  Bit-exact copies.
 * nd_stitch_add: main-loop body (denoise_image.py:249-267) + make_seamless_edges (:204-213):
 *   canvas_chw [3,H,W] += halved-overlap useful crops of tiles_nchw, tiles taken in ascending index order
 *   (same fp32 summation order as the reference).
 */
int nd_tile_gather(const float *img_chw, int width, int height, int cs, int ucs, int ol,
                   int tile_begin, int tile_count, float *tiles_nchw, void *stream);
int nd_stitch_add(float *canvas_chw, int width, int height, int cs, int ucs, int ol,
                  const float *tiles_nchw, int tile_begin, int tile_count, void *stream);

/* ---------------------------------------------------------------- UtNet (UtNet.py:13-109)
 * Weight contract: the reference state-dict.  nd_utnet_num_tensors / nd_utnet_tensor_name
 * enumerate the keys in the order nd_utnet_pack_weights expects host pointers (float32, contiguous, torch layout:
 * Conv2d [Cout,Cin,k,k], ConvTranspose2d [Cin,Cout,k,k], bias [Cout], PReLU weight [1]).  For ELU/Hardswish the
 * PReLU entries are absent from the state-dict; pass NULL for them.
 * nn_common.Model.instantiate_model (nn_common.py:116-138) -> pack once at load time, upload, keep resident.
 */
int nd_utnet_num_tensors(void);
const char *nd_utnet_tensor_name(int idx);
size_t nd_utnet_packed_bytes(int funit, int dtype);
int nd_utnet_pack_weights(int funit, int dtype, const float *const *tensors, int n_tensors,
                          void *packed_host, size_t packed_bytes);

/* Workspace for one (cs, batch) geometry: activations in the quad-planar layout, zero borders included.
 * nd_utnet_workspace_init must run once on a workspace before its first forward with that geometry. */
/* The same blob built in HBM from tensors that already live there (fp32 storage only): device-side packers; the Winograd
 * weight transforms are evaluated in fp32 instead of double (packed values agree to ~1e-7 relative).  Stream-ordered. */
int nd_utnet_pack_weights_device(int funit, int dtype, const float *const *dev_tensors, int n_tensors, void *packed_dev,
                                 size_t packed_bytes, void *stream);
size_t nd_utnet_workspace_bytes(int funit, int cs, int batch, int dtype);
int nd_utnet_workspace_init(void *workspace, size_t workspace_bytes, int funit, int cs, int batch, int dtype,
                            void *stream);

/* Non-square inputs (the --whole_image branch, denoise_image.py:110-128): h and w must each be of the form 16k+56. */
size_t nd_utnet_workspace_bytes_hw(int funit, int h, int w, int batch, int dtype);
int nd_utnet_workspace_init_hw(void *workspace, size_t workspace_bytes, int funit, int h, int w, int batch, int dtype,
                               void *stream);
int nd_utnet_forward_hw(int funit, int act, int dtype, int flags, const void *packed_dev,
                        const float *x_nchw, float *y_nchw, int batch, int h, int w,
                        void *workspace, size_t workspace_bytes, void *stream);

/* UtNet.forward (UtNet.py:97-109): x_nchw [batch,3,cs,cs] -> y_nchw [batch,3,cs,cs], both float32 in HBM. */
int nd_utnet_forward(int funit, int act, int dtype, int flags, const void *packed_dev,
                     const float *x_nchw, float *y_nchw, int batch, int cs,
                     void *workspace, size_t workspace_bytes, void *stream);

/* The whole hot loop of denoise_image.py:240-267 for tiles [tile_begin, tile_begin+tile_count) of one frame,
 * device resident: gather(+mirror) -> UtNet -> useful crop -> seamless edges -> canvas +=.
 * tile_count <= batch of the workspace.  Equivalent to nd_tile_gather + nd_utnet_forward + nd_stitch_add
 * without materialising the NCHW tile batch. */
int nd_utnet_denoise_tiles(int funit, int act, int dtype, int flags, const void *packed_dev,
                           const float *img_chw, float *canvas_chw, int width, int height,
                           int cs, int ucs, int ol, int tile_begin, int tile_count, int batch,
                           void *workspace, size_t workspace_bytes, void *stream);

/* Profiling entry point for the roofline report: one pass of the conv stack (22 MFMA conv layers + 4 pools, the launches
 * between the input pack and the final 1x1) with a HIP event recorded on `stream` between launches.  Synchronises the stream.
 * nd_utnet_step_name(i): reference layer key of step i ("maxpool" for pools). */
typedef struct nd_step_profile {
    float ms;               /* duration of the step (all its launches)                                                      */
    float ms_xform_in;      /* three-pass Winograd layers: input transform pass, GEMM launch, output transform pass (0 for   */
    float ms_gemm;          /*   the other forms, and for batches above one Winograd chunk)                                  */
    float ms_xform_out;
    int form;               /* -1 pool; 0 direct implicit GEMM (conv_qp); 1 fused 1-D Winograd F(4,3) (conv_w1d); 2 F(2,3);  */
                            /*   3 three-pass Winograd F(6x6,3x3): k_wino_in2 -> 64 GEMMs in one conv_qp launch -> k_wino_out2      */
    int kind;               /* nd_layer_kind, -1 for pools                                                                  */
    double flops;           /* algorithmic FLOP for `batch` tiles (SURVEY.md 2a convention; 0 for pools)                    */
    double mfma_flops;      /* FLOP the matrix cores execute in that form (MFMA instructions x 4096)                        */
    double bytes;           /* algorithmic HBM bytes: input + output activations + weights                                   */
    double xform_bytes_in;  /* three-pass layers: algorithmic HBM bytes of the two transform passes                          */
    double xform_bytes_out;
} nd_step_profile;
/* crop: margin of the useful tile centre the caller will keep ((cs - ucs) / 2 of the denoise loop; 0: the whole output, as
 * nd_utnet_forward computes it) -- the stack then runs exactly as inside nd_utnet_denoise_tiles with the same flags */
int nd_utnet_profile_stack(int funit, int act, int dtype, int flags, const void *packed_dev, int batch, int cs, int crop,
                           void *workspace, size_t workspace_bytes, void *stream, nd_step_profile *steps, int max_steps);
const char *nd_utnet_step_name(int i);
/* host-only: region {r0, c0, rows, cols} of stack step `step` that nd_utnet_denoise_tiles computes when the centre [crop, cs - crop)
 * of a tile is kept (output grid of a 3x3 layer, input grid of a 2x2 stride-2 transpose; zeros: the whole tensor); returns the number
 * of restricted steps (>= 0) or a negative nd_status */
int nd_utnet_useful_region(int funit, int cs, int crop, int step, int *rect);

/* ---------------------------------------------------------------- UNet (ThirdPartyNets.py:62-169, eval mode)
 * Same contract as the UtNet entry points; BatchNorm2d running statistics are folded into the convolutions when the
 * weights are packed (nd_unet_tensor_name enumerates conv weight/bias and BN weight/bias/running_mean/running_var keys).
 * Any h, w >= 16 (odd sizes go through the reference's F.pad fix-up).  y = sigmoid(outc(...)); `find_noise` is the
 * caller's x - y. */
int nd_unet_num_tensors(void);
const char *nd_unet_tensor_name(int idx);
size_t nd_unet_packed_bytes(int dtype);
int nd_unet_pack_weights(int dtype, const float *const *tensors, int n_tensors, void *packed_host, size_t packed_bytes);
size_t nd_unet_workspace_bytes(int h, int w, int batch, int dtype);
int nd_unet_workspace_init(void *workspace, size_t workspace_bytes, int h, int w, int batch, int dtype, void *stream);
int nd_unet_forward(int dtype, const void *packed_dev, const float *x_nchw, float *y_nchw, int batch, int h, int w,
                    void *workspace, size_t workspace_bytes, void *stream);

/* FLOP per tile by the reference's own accounting, for roofline reports. */
double nd_utnet_flops(int funit, int cs);

/* ---------------------------------------------------------------- single-layer entry points (parity tests)
 * One weighted layer on NCHW float32 tensors: pack -> quad-planar conv kernel -> unpack.
 * y = act(layer(x) + bias); output shape: CONV3 (H-2,W-2); CONVT3 (H+2,W+2); CONVT2S2 (2H,2W); CONV1 (H,W). */
size_t nd_layer_packed_bytes(int kind, int cin, int cout, int dtype);
int nd_layer_pack(int kind, int cin, int cout, int dtype, const float *weight, const float *bias,
                  void *packed_host, size_t packed_bytes);
size_t nd_layer_workspace_bytes(int kind, int batch, int cin, int cout, int h, int w, int dtype);
int nd_layer_forward(int kind, int act, float slope, int dtype, const void *packed_dev,
                     const float *x_nchw, int batch, int cin, int h, int w, int cout, float *y_nchw,
                     void *workspace, size_t workspace_bytes, int variant, int flags, void *stream);
/* nn.MaxPool2d(2) (UtNet.py:34) on NCHW float32 through the quad-planar pool kernel. */
int nd_maxpool2_forward(const float *x_nchw, int batch, int c, int h, int w, float *y_nchw,
                        void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------- training-step building blocks (next row f3)
 * Weight and bias gradient of one layer (autograd's conv backward-weight; nn_common.py:201-218 loss.backward()):
 * x [B,cin,h,w] = the layer's input, dy = gradient w.r.t. the layer's (pre-activation) output, NCHW fp32 in HBM;
 * dw in the torch weight layout of the layer kind, db [cout].  The data gradient of a layer is a FORWARD launch of the
 * transposed kind (nd_layer_forward: CONV3 <-> CONVT3, CONVT2S2 -> CONV2S2) on the same weight tensor. */
size_t nd_layer_wgrad_workspace_bytes(int kind, int batch, int cin, int cout, int h, int w);
int nd_layer_wgrad(int kind, const float *x_nchw, const float *dy_nchw, int batch, int cin, int h, int w, int cout,
                   float *dw, float *db, void *workspace, size_t workspace_bytes, void *stream);

/* UtNet training step (BASELINE config 5; nn_train.py:308-380, nn_common.py:198-255), PReLU networks, fp32.
 * Parameters and gradients are flat float buffers in state-dict order (nd_utnet_tensor_name / nd_utnet_param_range):
 * one buffer for the data-parallel all-reduce and for the optimizer.  nd_utnet_train_step = device-side weight
 * packing + forward + loss + backward:
 *     loss = w_l1 * mean|g - t| + w_mse * mean (g - t)^2 + w_ssim * mean_n(1 - SSIM_n(g, t))
 *            + w_msssim * mean_n(1 - MS-SSIM_n(g, t)),        g = clip(net(x), 0, 1), t = target   (nn_common.py:198-241;
 *     the SSIM terms as in nd_ssim_loss_grad; MS-SSIM needs cs >= 161, so it cannot be used on 136-pixel crops)
 * x, target, y_out: [batch,3,cs,cs] NCHW fp32 in HBM (cs = 16k+56, e.g. 136 / 184); loss_out: one float in HBM.
 * loss_cs: the criteria see the centre crop of loss_cs x loss_cs pixels of output and target (pt_ops.pt_crop_batch,
 * nn_train.py:319-323; --loss_cs); the gradient is zero outside it.  0 or cs: the whole output.
 * nd_adam_step = torch.optim.Adam(lr, betas, eps, amsgrad) on the flat buffers (nn_common.py:185). */
size_t nd_utnet_param_count(int funit);
int nd_utnet_param_range(int funit, int tensor_idx, size_t *offset, size_t *count);
size_t nd_utnet_train_blob_bytes(int funit);
size_t nd_utnet_train_workspace_bytes(int funit, int cs, int batch);
int nd_utnet_train_workspace_init(void *workspace, size_t workspace_bytes, int funit, int cs, int batch, void *stream);
int nd_utnet_train_step(int funit, int flags, const float *params, float *grads, void *blobs, const float *x_nchw,
                        const float *target_nchw, float *y_out_nchw, float w_l1, float w_mse, float w_ssim, float w_msssim,
                        float *loss_out, int batch, int cs, int loss_cs, void *workspace, size_t workspace_bytes, void *stream);
/* The two halves of the step for torch.autograd, so that the reference's own training statements
 * (nn_common.py:198-218: `self.model(noisy_batch).clip(0,1)`, `loss.backward()`) run unchanged on the module:
 * nd_utnet_train_forward = device-side weight packing + forward with the pre-activations kept in `workspace`;
 * nd_utnet_train_backward = the whole backward from gy = d loss / d output [batch,3,cs,cs] into the flat gradient buffer.
 * act: ND_ACT_PRELU | ND_ACT_ELU | ND_ACT_HARDSWISH (networks/UtNet.py:17-26; tensors of the parameter layout that the
 * activation does not have -- PReLU slopes -- are ignored and their gradients left untouched).  `workspace` and `blobs`
 * carry the forward's state to the backward call: nothing else may use them in between.  The gradient of the input image
 * is not produced. */
int nd_utnet_train_forward(int funit, int act, int flags, const float *params, void *blobs, const float *x_nchw, float *y_out_nchw,
                           int batch, int cs, void *workspace, size_t workspace_bytes, void *stream);
int nd_utnet_train_backward(int funit, int act, int flags, const float *params, float *grads, void *blobs, const float *gy_nchw,
                            int batch, int cs, void *workspace, size_t workspace_bytes, void *stream, void *const *bucket_events,
                            int n_events);
/* Data-parallel training that overlaps the gradient reduction with the backward pass (BASELINE configs[4]): the flat gradient
 * buffer is cut into nd_utnet_grad_buckets = 9 contiguous ranges, one per decoder / encoder level, numbered in the order the
 * backward pass completes them; bucket_events[k] (hipEvent_t, nullable array) is recorded on `stream` when bucket k is final,
 * so a reducer on another stream can all-reduce it under the backward of the shallower levels.  nd_utnet_train_step_ev = the
 * fused step with those events. */
int nd_utnet_grad_buckets(int funit, size_t *offsets, size_t *counts, int max);
int nd_utnet_train_step_ev(int funit, int flags, const float *params, float *grads, void *blobs, const float *x_nchw,
                           const float *target_nchw, float *y_out_nchw, float w_l1, float w_mse, float w_ssim, float w_msssim,
                           float *loss_out, int batch, int cs, int loss_cs, void *workspace, size_t workspace_bytes, void *stream,
                           void *const *bucket_events, int n_events);
int nd_adam_step(float *params, const float *grads, float *m, float *v, float *vmax, size_t n, float lr, float beta1,
                 float beta2, float eps, int step, int amsgrad, void *stream);

/* ---- image-quality scores of the eval harness ------------------------------------------
 * Replace pt_helpers.get_losses (common/libs/pt_helpers.py:40-48): F.mse_loss, piqa.SSIM and piqa.MS_SSIM with piqa's
 * defaults, reduction=None (common/libs/pt_losses.py:6-18 subtracts them from 1; so do the Python classes here).
 * x, y: float32 [n, c, h, w] in HBM, values in [0, 1].  out: float32 [n] in HBM (nd_mse: one float).
 * nd_ssim needs h, w >= 11; nd_ms_ssim needs h, w >= 161 (five scales of an 11-tap window) -- ND_EINVAL otherwise, where
 * piqa raises from the convolution.  The workspace of nd_ssim_workspace_bytes serves all three. */
size_t nd_ssim_workspace_bytes(int n, int c, int h, int w);
int nd_ssim(const float *x, const float *y, int n, int c, int h, int w, float *out, void *workspace, size_t workspace_bytes,
            void *stream);
int nd_ms_ssim(const float *x, const float *y, int n, int c, int h, int w, float *out, void *workspace,
               size_t workspace_bytes, void *stream);
int nd_mse(const float *x, const float *y, size_t count, float *out, void *workspace, size_t workspace_bytes, void *stream);

/* The same two scores as differentiable training losses (nn_common.py:170-177, 226-241: criterions['SSIM'|'MSSSIM'], weighted
 * sum, loss.backward()):   loss_acc[0] += weight * mean_n (1 - score_n(x, y));   gx = (accumulate ? gx : 0) + d(that)/dx.
 * x = generated batch, y = target, gx: float32 [n, c, h, w]; multiscale 0 = SSIM, 1 = MS-SSIM (h, w >= 161). */
size_t nd_ssim_loss_workspace_bytes(int n, int c, int h, int w);
int nd_ssim_loss_grad(const float *x, const float *y, int n, int c, int h, int w, int multiscale, float weight,
                      float *loss_acc, float *gx, int accumulate, void *workspace, size_t workspace_bytes, void *stream);

/* ---- Winograd forms of a 3x3 layer, fp32 inference (same math as nd_layer_forward on a CONV3 / CONVT3 layer, re-associated).
 * tile = 2 | 4 | 6: three-pass F(tile x tile, 3 x 3) (input transform, one launch of (tile+2)^2 GEMMs, output transform;
 *               Cin % 16 == 0; agrees with the direct kernel to ~1e-6 / ~1e-5 / ~2e-5 relative);
 * tile = 1 | 3: 1-D F(2,3) | F(4,3) along x inside the implicit-GEMM kernel, transform in registers (~1e-6 / ~5e-6);
 * tile = 5    : the same F(4,3) form with the input transform shared by the workgroup through LDS (conv_w2d). */
size_t nd_winograd_packed_bytes(int tile, int cin, int cout);
int nd_winograd_pack(int tile, int kind, int cin, int cout, const float *w, const float *bias, void *packed,
                     size_t packed_bytes);
size_t nd_layer_winograd_workspace_bytes(int tile, int kind, int batch, int cin, int cout, int h, int w);
int nd_layer_forward_winograd(int tile, int kind, int act, float slope, const void *packed, const float *x, int batch,
                              int cin, int h, int w, int cout, float *y, void *workspace, size_t workspace_bytes, int flags,
                              void *stream);

/* Kernel micro-benchmark: `iters` launches of one conv layer (variant -1 = automatic choice) on pseudo-random
 * quad-planar data carved from `workspace` (nd_layer_workspace_bytes + nd_layer_packed_bytes + 256 B); mean launch
 * duration from HIP events on `stream`.  Synchronises the stream. */
int nd_conv_bench(int kind, int dtype, int batch, int cin, int cout, int h, int w, int variant, int iters,
                  void *workspace, size_t workspace_bytes, void *stream, float *mean_ms);

int nd_winograd_bench(int tile, int kind, int batch, int cin, int cout, int h, int w, int iters, void *workspace,
                      size_t workspace_bytes, void *stream, float *mean_ms);

/* Name and average duration bookkeeping for bench.py: number of conv-kernel variants compiled in. */
int nd_num_conv_variants(void);
const char *nd_conv_variant_name(int variant);

#ifdef __cplusplus
}
#endif
#endif /* NIND_HIP_H */
