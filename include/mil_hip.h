/* mil_hip.h — C ABI of libmil_hip.so, the MI355X (gfx950) kernels under the ResNet-26 +
 * attention-MIL hot path.
 *
 * The reference (frankenz/Deep-convolutional-neural-network-ResNet-26-and-Attention-network) is
 * pure Python on stock PyTorch and has no FFI of its own; the "plugin interface" of this path is
 * the `nn.Module` surface of `Attention` (gbm/model.py:114-264).  This header is what a binding
 * for that path binds instead of the cuDNN/ATen library ops the reference dispatches to.  Each
 * entry point names the reference op(s) it replaces (file:line relative to the upstream repo).
 *
 * Conventions
 *   - plain C: raw device pointers, ints, floats; `stream` is a hipStream_t passed as void*.
 *   - every function returns MIL_OK (0) or an error code; nothing allocates, nothing synchronises,
 *     no global state: work is enqueued on `stream`, buffers are caller-owned.
 *   - activations are NHWC with channels padded to a multiple of 8 (20->24, 40, 60->64, 80; padded
 *     channels hold zeros); `dtype` selects the activation/weight-operand type:
 *         MIL_DT_F32  (0): exact-fp32 MFMA (v_mfma_f32_16x16x4_f32)  — the parity gate, bit-level
 *         MIL_DT_BF16 (1): bf16 operands, fp32 accumulate (v_mfma_f32_16x16x32_bf16) — the fast path
 *         MIL_DT_F32S (3): fp32 tensors exactly as MIL_DT_F32 (same layouts, same pointwise entry points), convolutions
 *                          and weight gradients as bf16x3 split products: each operand v = hi + lo (two bf16), product =
 *                          lo*hi + hi*lo + hi*hi with fp32 accumulation.  16 significant bits per operand: meets the 1e-3
 *                          gate on logits / attention weights (2.5e-4 measured) at 3/16 of the exact path's matrix time.
 *                          Accepted by mil_pack_conv_weights / mil_pack_job_fill (fragments [hi | lo], the byte count of
 *                          MIL_DT_F32), mil_conv_igemm, mil_conv_wgrad(_workspace), and — fused forms, where the shape has
 *                          one (MIL_ERR_UNSUPPORTED otherwise: the caller falls back to the un-fused calls) —
 *                          mil_conv_bwd_fused(_workspace) (20-channel layers), mil_conv_block_fwd (20-channel identity
 *                          blocks, maps of at least 8x16), mil_conv_chain (64 / 80 channels on 16x16 / 8x8 maps, and on the
 *                          19x19 / 10x10 maps of 300x300 tiles), mil_conv_s2_entry (all three stage entries: 20 -> 40,
 *                          40 -> 60 and 60 -> 80 channels, i.e. padded 24 -> 40, 40 -> 64, 64 -> 80), mil_conv_wgrad_pair
 *                          (20 -> 40 channels), mil_conv_dgrad_s2 (all three entries: 40 -> 20, 60 -> 40 and 80 -> 60
 *                          channels; the dense-output code MIL_DT_F32S_DGRAD only for the 40 -> 20 entry, cz_p == 40),
 *                          mil_stem_fwd_fused (xs must be null: no space-to-depth copy) and
 *                          mil_stem_bwd_fused_nchw(_workspace); every pointwise entry point takes MIL_DT_F32 for the same tensors.
 *   - master weights, biases, all gradients of parameters, and the whole MIL head are fp32.
 */
#ifndef MIL_HIP_H
#define MIL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIL_OK 0
#define MIL_ERR_ARG 1
#define MIL_ERR_UNSUPPORTED 2
#define MIL_ERR_LAUNCH 3

#define MIL_DT_F32 0
#define MIL_DT_BF16 1
/* bf16 with the GRADIENT tensors of the 20-channel stage stored dense: dz / addend / dx of mil_conv_bwd_fused, the output of
 * mil_conv_dgrad_s2 (40 -> 20 channels) and g_pool of mil_stem_bwd_fused(_nchw) are [n,H,W,20] (40 bytes per pixel)
 * instead of [n,H,W,24]; every activation (x, act, widx) keeps its padded layout.  These tensors are produced and consumed
 * only by those entry points (autograd's intermediate gradients of gbm/model.py:24-29), so the layout is theirs to choose:
 * 17 % fewer bytes on three of the four tensor passes of the stage's fused backward.  Entry points (or shapes) without a
 * kernel for it return MIL_ERR_UNSUPPORTED; query with the *_workspace functions before choosing it. */
#define MIL_DT_BF16_DGRAD 2
#define MIL_DT_F32S 3
/* MIL_DT_F32S with the same GRADIENT tensors stored dense as fp32: [n,H,W,20] = 80 bytes = five 16-byte pieces per pixel instead
 * of 96 (no padding traffic on three of the four tensor passes of the stage's fused backward).  Accepted by mil_conv_bwd_fused
 * (16x16 tiles: maps of at least 16x16), mil_conv_dgrad_s2 (40 -> 20 channels) and mil_stem_bwd_fused_nchw; query with the
 * *_workspace functions before choosing it. */
#define MIL_DT_F32S_DGRAD 4

#define MIL_PACK_FWD 0   /* B[(tap,ci)][co] = W[co][ci][ky][kx]                     */
#define MIL_PACK_DGRAD 1 /* B[(tap,co)][ci] = W[co][ci][k-1-ky][k-1-kx]             */
#define MIL_PACK_STEM 2  /* 7x7/s2 filter as a 4x4/s1 filter over space-to-depth x  */
#define MIL_PACK_DGRAD_S2 3 /* parity-class order for mil_conv_dgrad_s2; `bias` = 1x1 projection weight */

/* ABI version of this header (bumped on any signature change). */
int mil_abi_version(void);

/* Streaming device copy dst[0..bytes) = src[0..bytes) in 16-byte pieces (both 16-byte aligned, bytes a multiple of 16).  Not a
 * reference op: the calibration kernel bench.py times to state what a plain read+write stream reaches on the box, next to the
 * 8 TB/s specification its roofline fractions are priced against. */
int mil_stream_copy(void* dst, const void* src, size_t bytes, void* stream);
/* Diagnostic: the hi / lo bf16 split of MIL_DT_F32S as the split-precision kernels apply it to operands on their way into LDS
 * (hi = bf16(v), lo = bf16(v - hi); element pairs (2i, 2i+1) share one v_dot2c_f32_bf16 pair).  v [n] fp32 -> hi, lo [n] bf16 bit
 * patterns; n even.  Finite pairs: bit-identical to the plain subtraction.  A non-finite element makes its pair partner's lo a
 * NaN (Inf * 0 in the dot product). */
int mil_split_probe(const float* v, uint16_t* hi, uint16_t* lo, int n, void* stream);

/* ---- input packing ------------------------------------------------------------------------
 * fp32 NCHW tiles [n,3,H,W] (what `Attention.forward` receives, gbm/model.py:189-196) ->
 * NHWC space-to-depth [n, ceil(H/2), ceil(W/2), 16] of `dtype` (channel = c*4+dy*2+dx, 12 real).
 * Lets the 7x7 stride-2 stem conv (gbm/model.py:24) run as a 4x4 stride-1 implicit GEMM. */
int mil_stem_s2d(const float* x_nchw, void* out, int n, int H, int W, int dtype, void* stream);

/* ---- weight packing -----------------------------------------------------------------------
 * fp32 master weights in the reference state-dict layout [Cout][Cin][k][k] ->
 * MFMA B-fragment order [kstep][ntile][64 lanes][8].  `bias_pad` (optional) receives the bias
 * zero-padded to ntile*16 floats.  Element count of the packed buffer: mil_packed_weight_elems. */
int mil_packed_weight_elems(size_t* elems, int cout, int cin, int ks, int mode);
int mil_pack_conv_weights(const float* w, const float* bias, void* wpack, float* bias_pad, int cout, int cin,
                          int ks, int mode, int dtype, void* stream);

/* ---- convolution forward / data-gradient ---------------------------------------------------
 * Replaces nn.Conv2d forward (+bias) fused with LeakyReLU(0.1) and the residual add of
 * BasicResBlock.forward (nnBlocks.py:175-189), the stem conv+LeakyReLU (gbm/model.py:51-52), the
 * 1x1 stride-2 projection (gbm/model.py:38-40), and — run over dz with MIL_PACK_DGRAD weights — the
 * autograd data-gradient of those convs fused with the LeakyReLU backward mask.
 *     y = mask( lrelu?( conv(x) + bias? + res? ) ),   mask(v) = v * (act > 0 ? 1 : slope) if act
 *   x   [n,H,W,cin_p]; y/res/act [n,Ho,Wo,cout_p]; ks in {1,3,4}; stride in {1,2}; pad = leading pad.
 *   zero_insert=1 reads x as if zeros sat between its pixels (transposed stride-2 conv: dgrad of a
 *   stride-2 conv); H,W are then the dims of the small tensor and Ho,Wo the full-resolution dims. */
int mil_conv_igemm(const void* x, const void* wpack, const float* bias_pad, const void* res, const void* act,
                   void* y, int n_img, int H, int W, int cin_p, int Ho, int Wo, int cout_p, int ks, int stride,
                   int pad, int zero_insert, int apply_lrelu, float slope, int dtype, void* stream);

/* ---- convolution weight / bias gradient ----------------------------------------------------
 * Replaces the autograd weight- and bias-gradient of the same convs.  dw is written (not
 * accumulated) in the reference layout [cout][cin][k][k] fp32 (k = 7 when stem_mode), db [cout]
 * (may be null).  x [n,H,W,cpad(cin)] (or the 16-channel s2d tensor when stem_mode), dz
 * [n,Ho,Wo,cpad(cout)].  Split-K over workgroups with a fixed-order second pass: bitwise
 * reproducible.  `workspace` must hold mil_conv_wgrad_workspace() bytes.  accumulate=1 adds into dw/db
 * (gradient accumulation straight into the flat gradient bucket) instead of overwriting them. */
int mil_conv_wgrad_workspace(size_t* bytes, int n_img, int H, int W, int cin, int Ho, int Wo, int cout, int ks,
                             int stride, int pad, int stem_mode, int dtype);
int mil_conv_wgrad(const void* x, const void* dz, float* dw, float* db, void* workspace, size_t workspace_bytes,
                   int n_img, int H, int W, int cin, int Ho, int Wo, int cout, int ks, int stride, int pad,
                   int stem_mode, int accumulate, int dtype, void* stream);

/* ---- batched slab reductions -------------------------------------------------------------------------
 * Every weight-gradient producer below (mil_conv_wgrad, mil_conv_wgrad_pair, mil_conv_bwd_fused, mil_stem_bwd_fused) ends in
 * a fixed-order reduction of per-workgroup partial sums ("slabs" in its workspace) into dW/db: 28 launches of ~10 us in one
 * backward pass of the encoder (autograd of nnBlocks.py:169-171 / gbm/model.py:24).  Between mil_reduce_defer_begin and
 * mil_reduce_defer_end the producers called on THIS thread record that reduction as a job in the caller's host table
 * (max_jobs records of mil_reduce_job_bytes() bytes) instead of launching it; mil_wgrad_reduce_all then runs every recorded
 * job in ONE launch (same summation trees: bit-identical to the per-call reductions).  Contract: each producer's workspace
 * must stay untouched until mil_wgrad_reduce_all has run on the same stream; `jobs_dev` is a device copy of the table,
 * `jobs_host` the table itself; at most 64 jobs per table.  This is the library's only state: thread-local, and empty
 * outside a begin/end pair. */
int mil_reduce_job_bytes(void);
int mil_reduce_defer_begin(void* jobs_host, int max_jobs);
int mil_reduce_defer_end(int* njobs);
int mil_wgrad_reduce_all(const void* jobs_dev, const void* jobs_host, int njobs, void* stream);

/* ---- fused backward of a 3x3 stride-1 conv (bf16 path, narrow layers) --------------------------
 * One pass over dz and x yields BOTH autograd results of nn.Conv2d (nnBlocks.py:169-171):
 *     dx = (conv^T(dz, W) + addend?) * (apply_mask ? lrelu'(x) : 1)      and      dW, db
 * (replaces one mil_conv_igemm dgrad launch + one mil_conv_wgrad launch and their second read of x and
 * dz).  dz [n,H,W,cpad(cout)], x/addend/dx [n,H,W,cpad(cin)], wpack_dgrad = MIL_PACK_DGRAD weights.
 * Returns MIL_ERR_UNSUPPORTED for shapes without a fused kernel (caller then uses the two calls). */
int mil_conv_bwd_fused_workspace(size_t* bytes, int n_img, int H, int W, int cout, int cin, int ks, int pad, int dtype);
int mil_conv_bwd_fused(const void* dz, const void* wpack_dgrad, const void* x, const void* addend, void* dx, float* dw,
                       float* db, void* workspace, size_t workspace_bytes, int n_img, int H, int W, int cout, int cin,
                       int ks, int pad, int apply_mask, int accumulate, float slope, int dtype, void* stream);

/* ---- pooling -------------------------------------------------------------------------------
 * MaxPool2d(3, stride 2, pad 1) (gbm/model.py:26,53): y [n,Ho,Wo,cp], Ho=(H-1)/2+1; `widx` records per
 * output element (uint8, same shape as y) the winning tap (bits 0-3) and whether the winner is <= 0
 * (bit 4).  The backward gathers gy through widx into gx [n,H,W,cp]; with apply_lrelu_mask it also
 * applies the LeakyReLU backward of the stem activation (gbm/model.py:25,52) from bit 4, so the stem
 * output itself is never re-read. */
int mil_maxpool_fwd(const void* x, void* y, uint8_t* widx, int n, int H, int W, int cp, int dtype, void* stream);
int mil_maxpool_bwd(const void* gy, const uint8_t* widx, void* gx, int n, int H, int W, int cp,
                    int apply_lrelu_mask, float slope, int dtype, void* stream);

/* Tile pre-processing on the device (RoiBuilder.py:193-210: Pad(100) -> RandomCrop(roi) -> Resize(res) ->
 * RandomHorizontalFlip -> RandomVerticalFlip -> ToTensor -> Normalize(.5,.5); `img_finalize_flat` without pad/crop/flips):
 * uint8 ROIs [T,S,S,3] -> fp32 [T,3,R,R] in [-1,1], the tensor gbm/classify_combined.py:432 hands to the model.  The
 * resampling is Pillow's two-pass fixed-point bilinear filter, bit-exact.  mil_resize_plan/_coeffs run on the HOST and fill
 * host arrays (bounds [R*2], kk [R*ksize]) that the caller uploads once per (S,R); params [T,4] = (top, left, hflip,
 * vflip) drawn by the caller (top/left in [0, 2*pad]) or null for the flat chain. */
int mil_resize_plan(int in_size, int out_size, int* ksize);
int mil_resize_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* kk);
int mil_tile_preprocess(const uint8_t* rois, const int32_t* params, const int32_t* bounds_host, const int32_t* bounds_dev,
                        const int32_t* kk_dev, float* out, int T, int S, int pad, int R, void* stream);
/* Same chain, output as the bf16 space-to-depth NHWC tensor xs [T,R/2,R/2,16] (channel = c*4 + dy*2 + dx, 12 real, 4 zero)
 * that mil_stem_fwd_fused_xs / mil_stem_bwd_fused read: values = the bf16 roundings of mil_tile_preprocess's.  R even. */
int mil_tile_preprocess_s2d(const uint8_t* rois, const int32_t* params, const int32_t* bounds_host, const int32_t* bounds_dev,
                            const int32_t* kk_dev, void* xs, int T, int S, int pad, int R, void* stream);

/* Forward of a whole identity-shortcut residual block in one pass (bf16 path; nnBlocks.py:175-189 with
 * downsample=None): o1 = lrelu(conv3x3(x)+b1) — written because the backward needs it — and
 * y = lrelu(conv3x3(o1)+b2+x).  x is read once (operand and residual), the mid activation feeds conv2 from LDS.
 * cp in {24, 40}, H and W >= 16; otherwise MIL_ERR_UNSUPPORTED (caller: two mil_conv_igemm calls). */
/* Two 3x3 stride-1 convs back to back on the small maps of the last two stages — 80 channels on 8x8 maps, 64 channels on
 * 16x16 maps (layers 4 and 3 at 256x256 tiles) —, one launch: a workgroup keeps whole images in LDS (no halo exchange
 * between workgroups on whole images) and streams both filters:
 *     outA = epiA( convA(x) ),   outB = epiB( convB(outA) ),   epi(v) = mask( lrelu?( v + bias? + res? ) )
 * i.e. mil_conv_igemm's epilogue contract for each.  That is BasicResBlock.forward with an identity shortcut
 * (nnBlocks.py:175-189: A = conv1+bias+lrelu -> o1, B = conv2+bias+x+lrelu -> out, resB = x) and the block's data-gradient
 * chain (A = conv2^T with MIL_PACK_DGRAD weights masked by o1 -> dmid, B = conv1^T + dz masked by x -> dx).  bf16,
 * (cp, H, W) in {(80, 8, 8), (64, 16, 16)}; otherwise MIL_ERR_UNSUPPORTED (the caller runs two mil_conv_igemm calls —
 * which take the same pixel-resident kernel one conv at a time on these shapes). */
int mil_conv_pair(const void* x, const void* wpackA, const float* biasA, const void* resA, const void* actA, int lreluA,
                  void* outA, const void* wpackB, const float* biasB, const void* resB, const void* actB, int lreluB,
                  void* outB, int n_img, int H, int W, int cp, float slope, int dtype, void* stream);

/* The general form: a chain of 1..6 such convs on the same resident images, one launch —
 *     out_0 = epi_0( conv_0(x) ),   out_k = epi_k( conv_k(out_{k-1}) )
 * where the res / act operand of a conv may be the output of an EARLIER conv of the chain (every lane re-reads exactly
 * the bytes it stored itself).  Five convs are everything of the last stage behind its stride-2 entry convs (conv2 of the
 * entry block + two identity blocks, gbm/model.py:29), or the stage's whole data-gradient chain.  `convs` is a HOST array.
 * Same shapes and error behaviour as mil_conv_pair. */
typedef struct MilChainConv {
    const void* wpack;      /* MIL_PACK_FWD / MIL_PACK_DGRAD fragments */
    const float* bias;      /* padded bias or NULL */
    const void* res;        /* [n,H,W,cp] or NULL */
    const void* act;        /* [n,H,W,cp] or NULL */
    void* out;              /* [n,H,W,cp] */
    int lrelu;
    int pad_;
} MilChainConv;
int mil_conv_chain(const void* x, const MilChainConv* convs, int nconv, int n_img, int H, int W, int cp, float slope,
                   int dtype, void* stream);

/* Forward of a whole identity-shortcut block in one pass (nnBlocks.py:175-189):
 *     o1 = lrelu(conv3x3(x) + bias1),   y = lrelu(conv3x3(o1) + bias2 + x)
 * x is read once (operand and residual), o1 and y are written (what the backward needs), the mid activation goes from the
 * accumulators to LDS.  bf16: cp in {24, 40}, H and W >= 16;  MIL_DT_F32S: cp = 24, H >= 8, W >= 16;  otherwise
 * MIL_ERR_UNSUPPORTED (caller: two mil_conv_igemm calls — same results bit for bit in bf16, to fp32 rounding in split precision).
 * Two kernels per precision, same arithmetic per output element (bit-identical, tested): tiles of one image (16x16 / 16x8
 * pixels, input halo staged per tile), or — maps 64 pixels wide (bf16: also 128) and enough images to fill the resident
 * workgroups evenly, about 512 — a row walk: one workgroup per image, input and mid activation in LDS rings, every input
 * pixel fetched once.  The environment variable MIL_BLOCK_STRIP=0/1 (a test knob, read per call) forces either form. */
int mil_conv_block_fwd(const void* x, const void* wpack1, const float* bias1, const void* wpack2, const float* bias2,
                       void* o1, void* y, int n_img, int H, int W, int cp, float slope, int dtype, void* stream);

/* Forward of a stage-entry block's two stride-2 convs in one pass over the block input (bf16 path):
 *   y1 = lrelu(conv3x3_s2(x) + bias)   (nnBlocks.py:176-177)      y2 = conv1x1_s2(x)   (gbm/model.py:38-40)
 * wpack3 / wpack1: MIL_PACK_FWD fragments of the two filters.  (cin_p,cout_p) in {(24,40),(40,64),(64,80)}; otherwise
 * MIL_ERR_UNSUPPORTED (caller: two mil_conv_igemm calls). */
int mil_conv_s2_entry(const void* x, const void* wpack3, const float* bias_pad, const void* wpack1, void* y1, void* y2,
                      int n_img, int H, int W, int cin_p, int cout_p, float slope, int dtype, void* stream);

/* Weight gradients of a stage-entry block's two stride-2 convs from one pass over the block input (bf16 path; autograd
 * of the convs built at gbm/model.py:37-41): dw3 [cout,cin,3,3] and db3 [cout] from dz1 (gradient of the 3x3/s2 conv's
 * output), dw1 [cout,cin,1,1] from dz2 (gradient of the projection's output); x [n,H,W,cpad(cin)], dz1/dz2
 * [n,Ho,Wo,cpad(cout)].  The projection's A operand is the centre-tap rows the 3x3 weight gradient stages anyway.
 * (cin,cout) padded in {(24,40),(40,64),(64,80)}; otherwise MIL_ERR_UNSUPPORTED (caller: two mil_conv_wgrad calls). */
int mil_conv_wgrad_pair_workspace(size_t* bytes, int n_img, int H, int W, int cin, int Ho, int Wo, int cout, int dtype);
int mil_conv_wgrad_pair(const void* x, const void* dz1, const void* dz2, float* dw3, float* db3, float* dw1,
                        void* workspace, size_t workspace_bytes, int n_img, int H, int W, int cin, int Ho, int Wo,
                        int cout, int accumulate, int dtype, void* stream);

/* Data gradient of a stage-entry block's input in one pass (bf16 path; autograd of nnBlocks.py:175-189 for the
 * blocks built with stride 2 + projection at gbm/model.py:37-41):
 *   y = lrelu'(act) * ( conv3x3_s2^T(dz1) + conv1x1_s2^T(dz2) )
 * computed per output parity class over the compact dz maps (no zero-insert halo, no full-resolution temporary for
 * the projection term).  wpack: mil_pack_conv_weights(..., mode MIL_PACK_DGRAD_S2) of the 3x3 weight with the 1x1
 * projection weight passed in the `bias` argument (or null).  dz1/dz2 [n,h,w,cz_p], act/y [n,H,W,cx_p],
 * h = (H-1)/2+1; (cz_p,cx_p) in {(40,24),(64,40),(80,64)}; otherwise MIL_ERR_UNSUPPORTED (caller: two
 * mil_conv_igemm calls with zero_insert). */
int mil_conv_dgrad_s2(const void* dz1, const void* dz2, const void* wpack, const void* act, void* y, int n_img,
                      int h, int w, int cz_p, int H, int W, int cx_p, float slope, int dtype, void* stream);

/* Fused forward of the whole stem (bf16 path): space-to-depth + Conv2d(3,C,7,2,3) + bias + LeakyReLU +
 * MaxPool2d(3,2,1) in one pass over the fp32 NCHW tiles (gbm/model.py:24-26,51-53; alt_resnet.py:81-84,128-131
 * with slope 0).  Writes xs [n,H/2,W/2,16] (kept for the stem weight gradient; NULL = keep none, the backward is
 * then mil_stem_bwd_fused_nchw), pool [n,Hp,Wp,cout_p] and the
 * winner records widx (format of mil_maxpool_fwd); the intermediate tensors of mil_stem_s2d -> mil_conv_igemm(ks=4) ->
 * mil_maxpool_fwd are never materialised.  cout_p 64 (alt_resnet): bit-identical to that chain.  cout_p 24 (the 20-channel
 * stem, round 5): the maximum is taken over the fp32 accumulators in registers, as the reference pools fp32 activations
 * (gbm/model.py:51-53) — the chain pools the bf16-rounded stem tensor — so pool agrees with the chain's except where the
 * position code kept in the low four mantissa bits moves a bf16 rounding (< 1e-3 of the elements, one bf16 step), and
 * the recorded winner is the fp32 maximum (values that agree to 2^-19 compare by position; exact ties arise only between
 * identical input patches, whose weight-gradient contributions are identical whichever of them the gradient is routed to).
 * The packed filter (MIL_PACK_STEM) of a 20-channel stem carries six K-packed k-steps behind the eight standard ones
 * (mil_packed_weight_elems says so); bias_pad needs 32 floats.  H even, W % 4 == 0, x 16-byte aligned, otherwise
 * MIL_ERR_UNSUPPORTED (the caller then uses the three calls).
 * Two kernels for the 20-channel stem, bit-identical pool / widx: 8x16-pooled-pixel tiles, or — W == 256, xs == NULL and enough
 * images to fill the resident workgroups evenly, about 512 — a row walk (one workgroup per image, the space-to-depth rows in an
 * LDS ring, every input byte fetched once).  MIL_STEM_WALK=0/1 (a test knob, read per call) forces either form. */
int mil_stem_fwd_fused(const float* x_nchw, const void* wpack, const float* bias_pad, void* xs, void* pool,
                       uint8_t* widx, int n_img, int H, int W, int cout_p, float slope, int dtype, void* stream);
/* The same pass fed by the bf16 space-to-depth tensor xs [n,H2,W2,16] itself (mil_tile_preprocess_s2d's output): a
 * pre-processed tile then never exists as fp32 (RoiBuilder.py:193-210 -> gbm/model.py:24,51 without the fp32 stack in
 * between).  pool / widx are bit-identical to mil_stem_fwd_fused on the fp32 tiles xs is the bf16 rounding of; the
 * backward is mil_stem_bwd_fused(xs, ...).  bf16 only.  (The kernel keeps the next pixel's channels 8-11 in the padding
 * channels 12-15 of its LDS copy of a record; the tensor's own padding channels are not read and need not be zero.) */
int mil_stem_fwd_fused_xs(const void* xs, const void* wpack, const float* bias_pad, void* pool, uint8_t* widx, int n_img,
                          int H2, int W2, int cout_p, float slope, int dtype, void* stream);

/* Fused backward of the whole stem (bf16 path): max-pool backward + LeakyReLU backward + the 7x7 conv's weight
 * and bias gradient in one pass over xs [n,H2,W2,16] (mil_stem_s2d output), g_pool [n,Hp,Wp,24] (gradient of
 * the pooled output) and widx (mil_maxpool_fwd's winner records).  Replaces mil_maxpool_bwd + mil_conv_wgrad
 * (stem_mode) and the 4x-larger d(stem output) tensor between them.  dw [20,3,7,7], db [20] fp32. */
int mil_stem_bwd_fused_workspace(size_t* bytes, int n, int H2, int W2, int dtype);
int mil_stem_bwd_fused(const void* xs, const void* g_pool, const uint8_t* widx, float* dw, float* db, void* workspace,
                       size_t workspace_bytes, int n, int H2, int W2, float slope, int accumulate, int dtype,
                       void* stream);
/* Same backward when no space-to-depth copy was kept (mil_stem_fwd_fused with xs = NULL): reads the fp32 tiles
 * x [n,3,H,W] themselves and rebuilds the bf16 s2d tile in LDS (H even, W % 4 == 0, x 16-byte aligned, else
 * MIL_ERR_UNSUPPORTED).  The stem then moves 1.07 GB less in the forward and 0.5 GB more in the (compute-bound)
 * backward per 2048 tiles of 256x256.
 * bf16, W == 256 and enough images to fill the resident workgroups evenly (about 512): a row walk (one workgroup per image,
 * s2d rows and pooling windows in LDS rings) instead of 16x16 tiles; dW / db agree between the two forms to fp32 summation
 * order (a workgroup's partial sums cover other pixels).  MIL_STEM_WALK=0/1 (a test knob, read per call) forces either form. */
int mil_stem_bwd_fused_nchw_workspace(size_t* bytes, int n, int H, int W, int dtype);
int mil_stem_bwd_fused_nchw(const float* x_nchw, const void* g_pool, const uint8_t* widx, float* dw, float* db,
                            void* workspace, size_t workspace_bytes, int n, int H, int W, float slope, int accumulate,
                            int dtype, void* stream);

/* AdaptiveAvgPool2d((1,1)) + flatten + Linear(80,L,bias=False) (gbm/model.py:31-32,58-60).
 * x [n,hw,cp] -> pooled [n,c] fp32 (kept for backward), feats [n,nf] fp32 = pooled @ wfc^T (+ bias when given:
 * alt_resnet.py:93's Linear(512,num_classes) has one).  c, nf, cp <= 512.
 * Backward: dz [n,hw,cp] = lrelu'(act) * (dfeats @ wfc)/hw, dwfc [nf,c] = dfeats^T @ pooled, dbias [nf] (nullable). */
int mil_avgpool_fc_fwd(const void* x, const float* wfc, const float* bias, float* pooled, float* feats, int n, int hw,
                       int cp, int c, int nf, int dtype, void* stream);
int mil_avgpool_fc_bwd(const float* dfeats, const float* wfc, const float* pooled, const void* act, void* dz,
                       float* dwfc, float* dbias, int n, int hw, int cp, int c, int nf, int accumulate, float slope,
                       int dtype, void* stream);
/* The parameter gradients of the same layer on their own (two-stage, coalesced): dwfc [nf][c] (+)= dfeats^T pooled,
 * dbias [nf] (+)= column sums of dfeats (may be null).  mil_avgpool_fc_bwd skips them when its dwfc is null.
 * Workspace from mil_fc_wgrad_workspace. */
int mil_fc_wgrad_workspace(size_t* bytes, int n, int c, int nf);
int mil_fc_wgrad(const float* dfeats, const float* pooled, float* dwfc, float* dbias, void* workspace,
                 size_t workspace_bytes, int n, int c, int nf, int accumulate, void* stream);


/* ---- attention-MIL head --------------------------------------------------------------------
 * Replaces everything after the backbone in Attention.forward (gbm/model.py:198-246): ContextLayer
 * (batch-statistics BatchNorm1d + LeakyReLU + Dropout, :108-111), attention MLP + softplus/mask/L1
 * normalisation (:209-214), diagnostics (:201,216-219), buffer MLP (:223), pooling (:227-229),
 * softmax / argmax / label-smoothed weighted CE (:233-242, nnBlocks.py:71-85,121-134), l2 (:246).
 * Segmented over `nbags` bags: instances of bag b are rows [bag_offsets[b], bag_offsets[b+1]) of
 * H [ntot,80]; inst_bag[n] is the bag of row n.  keep_mask [ntot,80] uint8 (1 = kept) enables
 * Dropout(drop_p) as in training; null = eval.  class_weights [3] or null.
 * `weights` = 11 device pointers: context.bn.weight, context.bn.bias, attention.lin1.weight,
 * attention.lin1.bias, attention.lin2.weight, attention.lin2.bias, buffer.lin1.weight,
 * buffer.lin1.bias, buffer.classifier.weight, buffer.classifier.bias, weight_mask.
 * Outputs: a1 [ntot,3] (Aterm^T), wrois (bag b's [3,N_b] block at 3*bag_offsets[b]), bterm [ntot],
 * kld [nbags], rec [nbags, mil_head_rec_floats()] =
 *   {Mterm[3], y_pred[3], loss, error, Aterm_mu, Aterm_var, D[3], dloss/dM[3], y_hat, l2(bag 0)}.
 * Backward: dH [ntot,80] and `grads` (mil_head_grad_floats() floats: the 11 tensors' gradients back
 * to back, summed over bags), given grad_loss [nbags] and optional grad_l2 [1]. */
int mil_head_workspace_floats(size_t* floats, int ntot, int nbags);
int mil_head_grad_floats(void);
int mil_head_rec_floats(void);
int mil_head_fwd(const float* H, const int* bag_offsets, const int* inst_bag, const int64_t* labels,
                 const uint8_t* keep_mask, const float* class_weights, const float* const* weights,
                 float* workspace, float* a1, float* wrois, float* bterm, float* kld, float* rec, int ntot,
                 int nbags, float slope, float drop_p, float smoothing, float bn_eps, void* stream);
int mil_head_bwd(const float* H, const int* bag_offsets, const int* inst_bag, const uint8_t* keep_mask,
                 const float* const* weights, float* workspace, const float* bterm, const float* rec,
                 const float* grad_loss, const float* grad_l2, float* dH, float* grads, int ntot, int nbags,
                 float slope, float drop_p, void* stream);

/* ---- wide (64..512-channel) layers: the alt_resnet.py configuration -----------------------------
 * Channel-blocked implicit GEMM (64-wide output blocks x 32-wide input chunks staged through LDS) for channel
 * counts that do not fit the resident-filter kernels above.  Replaces conv3x3 / conv1x1 of alt_resnet.py:24-33
 * (bias-free, ReLU = slope 0) forward, data-gradient (MIL_PACK_DGRAD packing + zero_insert for stride 2) and
 * weight-gradient.  cin % 32 == 0, cout % 64 == 0; same tensor layout and epilogue semantics as mil_conv_igemm. */
int mil_wide_packed_elems(size_t* elems, int cout, int cin, int ks, int mode);
int mil_wide_pack_weights(const float* w, void* wpack, int cout, int cin, int ks, int mode, int dtype, void* stream);
int mil_wide_conv(const void* x, const void* wpack, const float* bias, const void* res, const void* act, void* y, int n_img,
                  int H, int W, int cin, int Ho, int Wo, int cout, int ks, int stride, int pad, int zero_insert,
                  int apply_relu, float slope, int dtype, void* stream);
int mil_wide_wgrad_workspace(size_t* bytes, int n_img, int H, int W, int cin, int Ho, int Wo, int cout, int ks, int stride,
                             int pad, int dtype);
int mil_wide_wgrad(const void* x, const void* dz, float* dw, void* workspace, size_t workspace_bytes, int n_img, int H, int W,
                   int cin, int Ho, int Wo, int cout, int ks, int stride, int pad, int accumulate, int dtype, void* stream);

/* ---- wide layers, gather-GEMM form (bf16): one deep-pipelined GEMM per 256-pixel x 128-channel tile whose K-steps
 * (one filter tap x 64 input channels) are copied to LDS by LDS-DMA, the next two in flight across the single barrier per
 * K-step.  Same role as mil_wide_conv (conv3x3 / conv1x1 of alt_resnet.py:24-33 forward and data gradient; ReLU = slope 0)
 * for channel counts with cin_x % 64 == 0 and cout_x % 128 == 0 (as executed: a data gradient contracts over the conv's
 * Cout and produces its Cin).  transposed == 0: y = conv(x); transposed == 1: x is dz on the conv's output grid, y is the
 * gradient on its input grid — a stride-2 gradient runs as four output-parity classes with only the taps that reach each.
 * mil_gconv_pack_weights: fp32 [Cout][Cin][k][k] -> swizzled 128-row x 64-channel filter images, mode 0 forward /
 * 1 data gradient. */
int mil_gconv_supported(int cin_x, int cout_x, int ks, int stride);
int mil_gconv_packed_elems(size_t* elems, int cout, int cin, int ks, int mode);
int mil_gconv_pack_weights(const float* w, void* wpack, int cout, int cin, int ks, int mode, void* stream);
int mil_gconv(const void* x, const void* wpack, const void* res, const void* act, void* y, int n_img, int H, int W, int cin_x,
              int Ho, int Wo, int cout_x, int ks, int stride, int pad, int transposed, int apply_relu, float slope, void* stream);

/* ---- training-step closure ---------------------------------------------------
 * mil_adam_step: torch.optim.Adam (gbm/classify_combined.py:519, stepped at :450-454) over the flat fp32
 * parameter / gradient buckets in ONE launch; `step` is the 1-based step count (bias correction),
 * `grad_scale` multiplies the gradient first (1/accumulated-bags, or 1).
 * mil_pack_all: re-pack every conv filter into MFMA fragment order in ONE launch from a device-resident
 * table of mil_pack_job_bytes()-sized records filled on the host by mil_pack_job_fill (same index maps as
 * mil_pack_conv_weights). */
int mil_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);
int mil_pack_job_bytes(void);
int mil_pack_job_fill(void* job_host, const float* w, const float* bias, void* out, float* bias_pad, int cout, int cin,
                      int ks, int mode, int dtype);
int mil_pack_all(const void* jobs_device, int njobs, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MIL_HIP_H */
