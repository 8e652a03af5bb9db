/* gridnext_hip.h - C ABI of libgridnext_hip.so: the MI355X (gfx950) kernels of GridNext's f∘g training path.
 *
 * The reference (adaly/gridnext) is pure Python on stock PyTorch and has NO FFI of its own; each entry point
 * below replaces the torch operator sequence at the cited reference location (paths relative to the reference
 * root).  Conventions:
 *   - plain C: device pointers + sizes + a stream handle (hipStream_t passed as void*); no torch types;
 *   - the caller owns every buffer (outputs and workspaces included); nothing is allocated, freed or cached
 *     across calls inside the library, kernels are enqueued on `stream` and the call returns immediately;
 *   - return value 0 = enqueued, <0 = error (GNX_ERR_*), never throws; re-entrant per stream;
 *   - activations are channels-last fp32 matrices X[rows][channels] with a leading dimension `ld*` in
 *     elements (rows = spots, or spots x H x W); labels are int64.
 */
#ifndef GRIDNEXT_HIP_H
#define GRIDNEXT_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define GNX_OK 0
#define GNX_ERR_BAD_ARG (-1)
#define GNX_ERR_LAUNCH (-2)
#define GNX_ERR_UNSUPPORTED (-3)

typedef void* gnx_stream_t; /* hipStream_t */

/* ---- corrector g: hexagonal convolution --------------------------------------------------------------------
 * Replaces hexagdly.Conv2d(kernel_size=1, stride=1, bias=True) as instantiated in
 * gridnext/gridnet_models.py:130-147.  x/y/dx/dy: [B][H][W][C] channels-last.  kernel0 [O][I][3][1],
 * kernel1 [O][I][2][2], bias [O] (hexagdly's parameter shapes).  mode 0: hexagdly addressing (odd columns
 * shifted down, gridnext/hexagdly_tools.py:68); mode 1: Visium odd-right grid, i.e. the
 * rot90/flip -> conv -> flip/rot90 sandwich of gridnet_models.py:178-185 without moving data.
 * Any I, O >= 1: up to 64 x 64 channels (weight gradient: 32 x 32) are one launch, wider layers are tiled over
 * channel chunks inside the call. */
int gnx_hexconv_fwd(const float* x, const float* kernel0, const float* kernel1, const float* bias, float* y,
                    int B, int H, int W, int I, int O, int mode, gnx_stream_t stream);
int gnx_hexconv_bwd_data(const float* dy, const float* kernel0, const float* kernel1, float* dx,
                         int B, int H, int W, int I, int O, int mode, gnx_stream_t stream);
long gnx_hexconv_bwd_weight_workspace(int B, int H, int W, int I, int O); /* floats */
int gnx_hexconv_bwd_weight(const float* x, const float* dy, float* dkernel0, float* dkernel1, float* dbias,
                           float* workspace, int B, int H, int W, int I, int O, int mode, int accumulate,
                           gnx_stream_t stream);
/* n hex layers' weight gradients as ONE launch plus one batched reduce - exactly gnx_hexconv_bwd_weight for every item, bit
 * for bit (same kernel body, slab layout, fixed-order reduce): the five hexagdly.Conv2d layers of the corrector
 * (gridnet_models.py:128-148) are independent once their output gradients exist and cost a launch apiece.  Layers wider than
 * 32 channels on either side, or more than 8 items: GNX_ERR_UNSUPPORTED, nothing launched (make the single calls).  `items`:
 * HOST array; each item's workspace: gnx_hexconv_bwd_weight_workspace(B, H, W, I, O) floats. */
typedef struct {
    const float* x; const float* dy; float* dkernel0; float* dkernel1; float* dbias; float* workspace;
    int B, H, W, I, O, mode, accumulate, pad;
} gnx_hexconv_wgrad_item;
int gnx_hexconv_bwd_weight_batch(const void* items, int n, gnx_stream_t stream);

/* ---- batch normalisation (+ReLU) over matrix rows -----------------------------------------------------------
 * nn.BatchNorm2d(32) of the corrector (gridnet_models.py:134-146) and nn.BatchNorm1d of the count MLP
 * (notebooks/Tutorial_visium_count.ipynb cell 12), torch semantics (biased var to normalise, unbiased for
 * running_var, momentum, num_batches_tracked). */
long gnx_bn_workspace(long M, int C); /* floats, for train_stats / relu_bwd / colsum */
int gnx_bn_train_stats(const float* x, long ld, long M, int C, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, long long* num_batches_tracked, float momentum,
                       float eps, float* scale, float* shift, float* save_mean, float* save_invstd,
                       float* workspace, gnx_stream_t stream);
/* The same with the layer's persistent sync words (gnx_bn_sync_words; see gnx_bn_train_stats_apply_sync): the form that puts
 * several workgroups on a channel block then needs no memset node in front of it.  sync == NULL: exactly gnx_bn_train_stats. */
int gnx_bn_train_stats_sync(const float* x, long ld, long M, int C, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, long long* num_batches_tracked, float momentum,
                       float eps, float* scale, float* shift, float* save_mean, float* save_invstd,
                       float* workspace, void* sync, gnx_stream_t stream);
/* gnx_bn_train_stats + gnx_scale_shift_relu in one call (one launch for matrices of up to 4992 rows: g's BatchNorm2d(32)
 * over one Visium grid, gridnet_models.py:134-146; the count MLP's BatchNorm1d). */
int gnx_bn_train_stats_apply(const float* x, long ld, long M, int C, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, long long* num_batches_tracked, float momentum,
                             float eps, float* scale, float* shift, float* save_mean, float* save_invstd, float* y, long ldy,
                             int relu, float* workspace, gnx_stream_t stream);
int gnx_bn_fold_eval(int C, const float* gamma, const float* beta, const float* running_mean,
                     const float* running_var, float eps, float* scale, float* shift, float* save_mean,
                     float* save_invstd, gnx_stream_t stream);
int gnx_scale_shift_relu(const float* x, long ldx, float* y, long ldy, long M, int C, const float* scale,
                         const float* shift, int relu, gnx_stream_t stream);
/* relu: 0 = plain BN, 1 = BN -> ReLU with x the BN input, 2 = BN -> ReLU with x holding the ACTIVATED output
 * relu(scale x + shift) (eval statistics only, scale != 0: the forward stored the bottleneck that way). */
int gnx_bn_relu_bwd(const float* dy, long lddy, const float* x, long ldx, float* dx, long lddx, long M, int C,
                    const float* scale, const float* shift, const float* save_mean, const float* save_invstd,
                    float* dgamma, float* dbeta, int relu, int training, int accumulate, int dx_accumulate,
                    float* workspace, gnx_stream_t stream);
/* Round 5 - matrices of 2049 ... 8192 rows (one 78 x 64 Visium grid: the corrector's two BatchNorm2d(32), gridnet_models.py:131-140,
 * and the count MLP's BatchNorm1d; dense block 2 of a batch of 32 patches) run gnx_bn_train_stats[_apply] / gnx_bn_relu_bwd with
 * FOUR workgroups per 16-channel block: a row lane keeps its <= 8 rows in registers (one round of loads for all walks) and the
 * four partial sums are exchanged behind a bounded barrier of those workgroups (added in workgroup order: deterministic; two-pass
 * variance as before).  The barrier's counters live in the workspace and are zeroed by a memset node per call - or, with the
 * `_sync` entry points, in gnx_bn_sync_words(C) 32-bit words the caller zeroed ONCE and keeps for this layer (never shared by
 * launches that may overlap): the kernels leave them zero, no memset node is needed.  sync == NULL: the plain entry points. */
long gnx_bn_sync_words(int C);
int gnx_bn_train_stats_apply_sync(const float* x, long ld, long M, int C, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, long long* num_batches_tracked, float momentum,
                                  float eps, float* scale, float* shift, float* save_mean, float* save_invstd, float* y, long ldy,
                                  int relu, float* workspace, void* sync, gnx_stream_t stream);
int gnx_bn_relu_bwd_sync(const float* dy, long lddy, const float* x, long ldx, float* dx, long lddx, long M, int C,
                         const float* scale, const float* shift, const float* save_mean, const float* save_invstd, float* dgamma,
                         float* dbeta, int relu, int training, int accumulate, int dx_accumulate, float* workspace, void* sync,
                         gnx_stream_t stream);
/* Transitions (norm -> relu -> conv 1x1 -> avgpool 2x2, densenet.py:47-54, run pool-first): gnx_bnrelu_avgpool2 = the pooled,
 * activated input [imgs*(S/2)^2][C] of the 1x1 conv (operand of its weight gradient); gnx_bn_relu_bwd_pooled = the adjoint
 * of norm -> relu given the gradient of the POOLED map (== gnx_avgpool2_bwd + gnx_bn_relu_bwd(relu = 1, training = 0)
 * without the full-size intermediate).  Eval statistics; 4 | C. */
int gnx_bnrelu_avgpool2(const float* in, long ldi, float* out, long ldo, long imgs, int C, int S, const float* scale,
                        const float* shift, gnx_stream_t stream);
int gnx_bn_relu_bwd_pooled(const float* dYp, long lddy, const float* x, long ldx, float* dx, long lddx, long imgs, int S,
                           int C, const float* scale, const float* shift, const float* save_mean,
                           const float* save_invstd, float* dgamma, float* dbeta, int accumulate, float* workspace,
                           gnx_stream_t stream);
int gnx_colsum(const float* x, long ld, long M, int C, float* out, int accumulate, float* workspace,
               gnx_stream_t stream);

/* ---- foreground-masked softmax cross-entropy + argmax/accuracy ------------------------------------------------
 * The permute/reshape/boolean-gather/labels-1/CrossEntropyLoss/torch.max chain of gridnext/training.py:152-160,
 * :176-177 (label_base 1) and the plain CE + argmax of :61-62 (label_base 0).  stats = {n_counted, n_correct}. */
long gnx_masked_ce_workspace(long M); /* doubles */
int gnx_masked_ce_fwd(const float* logits, long ld, const long long* labels, long M, int C, int label_base,
                      float accum_iters, float* loss, long long* stats, long long* preds, double* workspace,
                      gnx_stream_t stream);
int gnx_masked_ce_bwd(const float* logits, long ld, const long long* labels, long M, int C, int label_base,
                      const long long* stats, const float* dloss, float accum_iters, float* dlogits, long lddz,
                      gnx_stream_t stream);
/* The loops' per-batch bookkeeping (training.py:73-75, :176-178: running_loss += loss.item() * batch_size; running_corrects +=
 * correct) on device-resident sums, one launch: acc[0] += (double)*loss * weight, acc[1] += *correct, acc[2] += counted ?
 * *counted : counted_const (acc: 3 doubles on the device; loss / correct / counted: device scalars as gnx_masked_ce_fwd
 * leaves them). */
int gnx_meter_add(double* acc, const float* loss, double weight, const long long* correct, const long long* counted,
                  double counted_const, gnx_stream_t stream);

/* Row softmax + first-argmax of channels-last logits: torch.argmax / F.softmax of gridnext/utils.py:43-47. */
int gnx_softmax_rows(const float* logits, long ld, long M, int C, float* probs, long ldp, long long* preds,
                     gnx_stream_t stream);

/* ---- count-MLP spot head: fp32 MFMA GEMM -------------------------------------------------------------------------
 * F.linear forward / input-gradient / weight-gradient of the nn.Sequential in Tutorial_visium_count.ipynb cell 12.
 * C[M][N] = opA(A) opB(B) (+bias) (+C).  a_kmajor: A[k*lda+m] (a (genes, H*W) count grid read in place, replacing
 * the permute+copy of gridnet_models.py:167-169); b_kmajor: B[k*ldb+n]. */
int gnx_gemm_f32(const float* A, long lda, int a_kmajor, const float* B, long ldb, int b_kmajor, const float* bias,
                 float* C, long ldc, long M, long N, long K, int accumulate, gnx_stream_t stream);
/* The same product given `workspace` (gnx_gemm_f32_workspace(M, N, K) floats; NULL allowed when that is 0): the
 * 2000 -> 500 layer over a whole grid (gridnet_models.py:83-86 runs f on all 4992 B positions) then splits K over
 * workgroups and sums the partial tiles in a fixed order. */
long gnx_gemm_f32_workspace(long M, long N, long K);
int gnx_gemm_f32_ws(const float* A, long lda, int a_kmajor, const float* B, long ldb, int b_kmajor, const float* bias,
                    float* C, long ldc, long M, long N, long K, int accumulate, float* workspace, gnx_stream_t stream);

/* ---- DenseNet-BC image spot classifier, forward (gridnext/densenet.py) ------------------------------------------------
 * conv1x1_bnrelu : _DenseLayer norm1->relu1->conv1 (:35-40) and _Transition norm->relu->conv->pool (:47-54, pool=1)
 * conv3x3_bnrelu : _DenseLayer norm2->relu2->conv2 (:41); weights repacked by gnx_repack_conv3x3 to [tap][N][K]
 * conv_stem      : features.conv0 (:98-105), NCHW patches in, channels-last out
 * bnrelu_maxpool : norm0->relu0->pool0 (:106-110);  bnrelu_avgpool : norm_final->relu->adaptive_avg_pool (:153-156) */
int gnx_conv1x1_bnrelu(const float* A, long lda, const float* W, float* out, long ldc, long M, int N, int K,
                       const float* scale, const float* shift, int pool, int S_in, gnx_stream_t stream);
/* The same (pool = 0) given `workspace` (gnx_conv1x1_workspace(M, N, K) floats; 0 = the shape does not split, NULL allowed):
 * on matrices of few 128-row tiles - a batch of 32 patches, train_spotwise at the tutorial's batch size (training.py:11-98) -
 * the K range is split over workgroups and the partial tiles are summed in a fixed order. */
long gnx_conv1x1_workspace(long M, int N, int K);
int gnx_conv1x1_bnrelu_ws(const float* A, long lda, const float* W, float* out, long ldc, long M, int N, int K,
                          const float* scale, const float* shift, float* workspace, gnx_stream_t stream);
/* conv1x1_bnrelu_act: conv1x1_bnrelu (pool = 0) storing relu(out_scale[n] * y + out_shift[n]) -- _DenseLayer's norm2->relu2
 * (:38-39) folded into conv1's store in eval mode, so conv2 reads a ready operand (scale = shift = NULL below). */
int gnx_conv1x1_bnrelu_act(const float* A, long lda, const float* W, float* out, long ldc, long M, int N, int K,
                           const float* scale, const float* shift, const float* out_scale, const float* out_shift,
                           gnx_stream_t stream);
/* Opt-in form of conv1x1_bnrelu_act / conv1x1_bnrelu (pool = 0) for N = 128 on SPLIT bf16 operands (csrc/conv1x1_split.hip): every
 * fp32 operand = hi + lo in bf16, a product = three v_mfma_f32_32x32x16_bf16 with fp32 accumulation (a_lo b_hi + a_hi b_lo +
 * a_hi b_hi); tensors stay fp32 in HBM.  A whole DenseNet-121 in this arithmetic: logits 7e-6 of their range, CE 1e-6 from float64
 * (tools/diag/split_operand_feasibility.py).  Wp = gnx_conv1x1_split_pack of conv1.weight [128][K]
 * (gnx_conv1x1_split_pack_halves(K) 16-bit elements).  out_scale = NULL: no activation on the store.  4 | K, 4 | lda, 16-B aligned
 * A / scale / shift, else GNX_ERR_UNSUPPORTED. */
long gnx_conv1x1_split_pack_halves(int K);
int gnx_conv1x1_split_pack(const float* W, void* Wp, int K, gnx_stream_t stream);
int gnx_conv1x1_bnrelu_act_split(const float* A, long lda, const void* Wp, float* out, long ldc, long M, int K,
                                 const float* scale, const float* shift, const float* out_scale, const float* out_shift,
                                 gnx_stream_t stream);
/* The 3x3 companion (csrc/conv3x3_split.hip): conv2 (:41; N = 32, K = 128, padding 1) on an fp32 bottleneck that conv1 stored
 * activated, as nine shifted products of split bf16 operands - gnx_conv3x3_bnrelu with scale = shift = NULL in fp32-grade
 * arithmetic, bound by the bottleneck's bytes instead of the fp32 matrix instruction.  Wp = gnx_conv3x3_split_pack of
 * conv2.weight [32][128][3][3] (gnx_conv3x3_split_pack_halves() 16-bit elements).  S in {4, 8, 16, 32, 64}, S * S | M, 4 | lda,
 * 16-B aligned A, else GNX_ERR_UNSUPPORTED. */
long gnx_conv3x3_split_pack_halves(void);
int gnx_conv3x3_split_pack(const float* W, void* Wp, gnx_stream_t stream);
int gnx_conv3x3_split(const float* A, long lda, const void* Wp, float* out, long ldc, long M, int S, gnx_stream_t stream);
/* conv1's weight gradient on split bf16 operands (csrc/wgrad_split.hip; opt-in): gnx_wgrad_bnrelu with taps = 1, pool = 0 -
 * dW [N][K] (+)= dY^T relu(scale . X + shift) as torch.autograd derives it for densenet.py:35-37 - with fp32 operands in HBM, three
 * 16-bit matrix instructions per product and fp32 accumulation; slabs over pixels summed in a fixed order.  scale = shift = NULL: no
 * activation.  4 | N, K, lddy, ldx and 16-B aligned operands, else GNX_ERR_UNSUPPORTED. */
long gnx_wgrad1x1_split_workspace(long M, int N, int K);
int gnx_wgrad1x1_split(const float* dY, long lddy, const float* X, long ldx, const float* scale, const float* shift, float* dW,
                       float* workspace, long M, int N, int K, int accumulate, gnx_stream_t stream);
/* ... and conv2's (taps = 9, N = 32, K = 128, no prologue: A is the activated bottleneck, dY the layer's 32 gradient columns; dW in
 * conv2.weight's layout [32][128][3][3]).  S in {4, 8, 16, 32, 64}, S * S | M, 4 | lddy, lda, 16-B aligned operands. */
long gnx_wgrad3x3_split_workspace(long M);
int gnx_wgrad3x3_split(const float* dY, long lddy, const float* A, long lda, float* dW, float* workspace, long M, int S,
                       int accumulate, gnx_stream_t stream);
/* Training backward of norm1 -> relu1 -> conv1 (:35-37) w.r.t. the layer input, eval statistics: conv1's data gradient
 * dY . Wt^T (Wt = gnx_transpose_weight of conv1.weight) fused with the BN + ReLU backward and accumulated into the block
 * gradient dX[:, :N]; dbeta / dgamma from per-tile column sums (fixed order).  Same result as gnx_conv1x1_bnrelu followed by
 * gnx_bn_relu_bwd(relu = 1, training = 0, dx_accumulate = 1).  Whole tiles only, else GNX_ERR_UNSUPPORTED. */
long gnx_conv1x1_dgrad_bn_workspace(long M, int N);
int gnx_conv1x1_dgrad_bnrelu_bwd(const float* dY, long lddy, const float* Wt, const float* X, long ldx, float* dX, long lddx,
                                 long M, int N, int K, const float* scale, const float* shift, const float* mean,
                                 const float* invstd, float* dgamma, float* dbeta, int accumulate, float* workspace,
                                 gnx_stream_t stream);
/* The same pass ALSO producing conv1's weight gradient dW[K][N] (= gnx_wgrad_bnrelu(taps = 1) with the forward's BN + ReLU
 * prologue) from the tiles it stages: one pass over dY, X and dX instead of two (round 4).  Argument names as above, but with
 * N = 128 bottleneck channels fixed: dB [M][128] (lddb), W1t = conv1.weight transposed [K][128], G = the block gradient
 * (+= in columns [0, K)), 32 | K.  workspace: gnx_conv1x1_dgrad_wgrad_workspace(M, K) floats. */
long gnx_conv1x1_dgrad_wgrad_workspace(long M, int K);
int gnx_conv1x1_dgrad_wgrad_bnrelu_bwd(const float* dB, long lddb, const float* W1t, const float* X, long ldx, float* G, long ldg,
                                       long M, int K, const float* scale, const float* shift, const float* mean,
                                       const float* invstd, float* dgamma, float* dbeta, float* dW, float* workspace,
                                       int accumulate, gnx_stream_t stream);
/* conv2's data gradient fused with norm2 -> relu2's adjoint (eval statistics; A_act = the ACTIVATED bottleneck the training
 * forward stored): dX = scale * g * [A_act > 0] with g = conv3x3(dY, Wb), dbeta / dgamma from the same pass.  Replaces
 * gnx_conv3x3_bnrelu(dY, Wb) + gnx_bn_relu_bwd(relu = 2) (torch.autograd through densenet.py:41).  N == 128, K == 32. */
long gnx_conv3x3_dgrad_bn_workspace(long M, int N); /* floats */
int gnx_conv3x3_dgrad_bnrelu_bwd(const float* dY, long lddy, const float* Wb, const float* A_act, long lda, float* dX,
                                 long lddx, long M, int N, int K, int S, const float* scale, const float* shift,
                                 const float* mean, const float* invstd, float* dgamma, float* dbeta, int accumulate,
                                 float* workspace, gnx_stream_t stream);
/* The same operation with norm1 folded into the operands (eval mode, weights frozen): relu(sc x + sh) = sc clamp(x) + sh,
 * clamp = max(., -sh/sc) for sc > 0, min for sc < 0.  gnx_conv1x1_fold_clamp makes Wf[N][K] = W sc, bounds[2][K] (private
 * order) and out_shift_f[N] = out_scale (W . sh) + out_shift once per weight / BN update; gnx_conv1x1_clamped_act then
 * streams the raw activations global -> LDS by DMA and clamps them there with LDS float atomics (no vector ALU work beside
 * the matrix waves).  Equal to gnx_conv1x1_bnrelu_act up to rounding.  Whole tiles only (128 | M, 128 | N, 32 | K,
 * 16-B aligned): GNX_ERR_UNSUPPORTED otherwise -- call gnx_conv1x1_bnrelu_act then. */
int gnx_conv1x1_fold_clamp(const float* W, const float* scale, const float* shift, const float* out_scale,
                           const float* out_shift, float* Wf, float* bounds, float* out_shift_f, int N, int K,
                           gnx_stream_t stream);
int gnx_conv1x1_clamped_act(const float* A, long lda, const float* Wf, const float* bounds, float* out, long ldc, long M,
                            int N, int K, const float* out_scale, const float* out_shift_f, gnx_stream_t stream);
int gnx_repack_conv3x3(const float* w, float* wr, int N, int K, gnx_stream_t stream);
int gnx_conv3x3_bnrelu(const float* A, long lda, const float* Wr, float* out, long ldc, long M, int N, int K, int S,
                       const float* scale, const float* shift, gnx_stream_t stream);
/* conv2 (:41) of a pre-activated bottleneck with Winograd F(2,3) along x: 12 matrix "taps" per output pair instead of 18.
 * Same result as gnx_conv3x3_bnrelu(scale = shift = NULL) up to rounding; weights from gnx_winograd_conv3x3_weights
 * ([N][K][3][3] -> [3][4][N][K]).  GNX_ERR_UNSUPPORTED outside N == 32, 32 | K, S in {4, 8, 16, 32, 64}. */
int gnx_winograd_conv3x3_weights(const float* w, float* wu, int N, int K, gnx_stream_t stream);
int gnx_conv3x3_winograd(const float* A, long lda, const float* Wu, float* out, long ldc, long M, int N, int K, int S,
                         gnx_stream_t stream);
int gnx_conv_stem(const float* x, const float* w, float* out, long ldc, long imgs, int Cin, int H, int W, int O,
                  int KH, int KW, int stride, int pad, gnx_stream_t stream);
/* conv0 -> norm0 -> relu0 -> pool0 (:105-110) fused for the 128- and 256-px geometries (conv map 64 or 128 wide, even
 * height, W = 2 Wo): the conv0 map never goes to HBM.  Returns GNX_ERR_UNSUPPORTED for any other geometry: run gnx_conv_stem + gnx_bnrelu_maxpool then. */
int gnx_conv_stem_bnrelu_maxpool(const float* x, const float* w, float* out, long ldo, long imgs, int Cin, int H, int W,
                                 int O, int KH, int KW, int stride, int pad, const float* scale, const float* shift,
                                 gnx_stream_t stream);
/* The same kernel also recording each pooled element's window index (0..8, first maximum of the scan as torch's
 * max_pool2d, densenet.py:110) in argmax [imgs*(Ho/2)*(Wo/2)][O] bytes - the forward of the f-trained step under running
 * statistics (training.py:126); gnx_maxpool_bwd_argmax_bnrelu is its adjoint. */
int gnx_conv_stem_bnrelu_maxpool_argmax(const float* x, const float* w, float* out, long ldo, unsigned char* argmax,
                                        long imgs, int Cin, int H, int W, int O, int KH, int KW, int stride, int pad,
                                        const float* scale, const float* shift, gnx_stream_t stream);
int gnx_bnrelu_maxpool(const float* in, long ldi, float* out, long ldo, long imgs, int C, int Hi, int Wi,
                       const float* scale, const float* shift, gnx_stream_t stream);
int gnx_bnrelu_avgpool(const float* in, long ldi, float* out, long ldo, long imgs, int C, int S2,
                       const float* scale, const float* shift, gnx_stream_t stream);

/* fp16-MFMA variants (BASELINE config 5, "fp16 MFMA conv path"): same contract, operands rounded to fp16 in LDS,
 * v_mfma_f32_32x32x16_f16, fp32 accumulate/outputs.  Return GNX_ERR_UNSUPPORTED for unaligned pointers or K % 4 != 0. */
int gnx_conv1x1_bnrelu_f16(const float* A, long lda, const float* W, float* out, long ldc, long M, int N, int K,
                           const float* scale, const float* shift, int pool, int S_in, gnx_stream_t stream);
int gnx_conv3x3_bnrelu_f16(const float* A, long lda, const float* Wr, float* out, long ldc, long M, int N, int K, int S,
                           const float* scale, const float* shift, gnx_stream_t stream);
/* Config 5 with the bottleneck kept in fp16: conv1 stores it activated (the consumer's norm2 -> relu2, :38-39) and rounded to
 * fp16 [M][N] (ldc16 in halves); conv2 then streams it and the fp16 tap-major weights (gnx_repack_conv3x3's layout rounded
 * once) global -> LDS by DMA and multiplies with v_mfma_f32_32x32x16_f16, fp32 accumulation, fp32 output.
 * gnx_conv3x3_f16_dma: N == 32, 128 | K, 128 | M, S in {4, 8, 16, 32, 64}, 16-B aligned; else GNX_ERR_UNSUPPORTED. */
int gnx_conv1x1_bnrelu_f16_act16(const float* A, long lda, const float* W, void* out16, long ldc16, long M, int N, int K,
                                 const float* scale, const float* shift, const float* out_scale, const float* out_shift,
                                 gnx_stream_t stream);
int gnx_conv3x3_f16_dma(const void* A16, long lda16, const void* Wr16, float* out, long ldc, long M, int N, int K, int S,
                        gnx_stream_t stream);
/* Config 5 with fp16 BLOCK BUFFERS (the concatenated features themselves live in HBM as fp16, as they do under the
 * reference's autocast): the stem, conv1 (and the transitions: pool = 1, out_scale = out_shift = NULL), conv2 and the final
 * pool reading / writing [rows][ld halves]; arithmetic as the entry points above. */
int gnx_conv_stem_bnrelu_maxpool_h16(const float* x, const float* w, void* out16, long ldo, long imgs, int Cin, int H, int W,
                                     int O, int KH, int KW, int stride, int pad, const float* scale, const float* shift,
                                     gnx_stream_t stream);
/* ---- uint8 patches (SURVEY 8f-2 input pipeline) -------------------------------------------------------------------
 * The reference converts 8-bit patch files to floats on the host (torchvision ToTensor, gridnext/image_datasets.py:102-105,
 * :185; optionally Normalize) and moves 4 bytes per pixel host -> device (training.py:135-139).  Here patches stay uint8
 * [imgs][3][H][W] until the stem kernel's operand load: u8 / 255 and, with norm != NULL, (v - mean[c]) / std[c] - bit for bit
 * the floats torch produces.  norm: device floats {mean[3], std[3], 1/std[3]}.  gnx_conv_stem_bnrelu_maxpool_u8 = the fused
 * stem (same geometries; out_f16: pooled map stored as fp16); gnx_u8_to_f32 = the conversion alone (H*W % 4 == 0), for the
 * training forward (conv0's weight gradient re-reads float patches) and other geometries. */
int gnx_conv_stem_bnrelu_maxpool_u8(const unsigned char* x8, const float* w, void* out, long ldo, long imgs, int Cin, int H,
                                    int W, int O, int KH, int KW, int stride, int pad, const float* scale,
                                    const float* shift, const float* norm, int out_f16, gnx_stream_t stream);
int gnx_u8_to_f32(const unsigned char* x8, float* out, long imgs, int C, int H, int W, const float* norm,
                  gnx_stream_t stream);
/* The fused stem with fp16 matrix operands (config 5: patch and weights rounded to fp16 at the LDS stash,
 * v_mfma_f32_32x32x16_f16, fp32 accumulate), pooled map stored as fp16; x float patches, or uint8 when x_is_u8 (norm as
 * above, else NULL). */
int gnx_conv_stem_bnrelu_maxpool_f16mul(const void* x, int x_is_u8, const float* w, void* out16, long ldo, long imgs,
                                        int Cin, int H, int W, int O, int KH, int KW, int stride, int pad,
                                        const float* scale, const float* shift, const float* norm, gnx_stream_t stream);
int gnx_conv1x1_bnrelu_f16_h(const void* A16, long lda16, const float* W, void* out16, long ldc16, long M, int N, int K,
                             const float* scale, const float* shift, const float* out_scale, const float* out_shift,
                             int pool, int S_in, gnx_stream_t stream);
/* gnx_conv1x1_bnrelu_h16: the dense-layer case of gnx_conv1x1_bnrelu_f16_h (pool = 0, consumer activation) with the weight
 * rounded to fp16 once (W16 [N][K] halves): chunks of 64 channels, 16-B loads. */
int gnx_conv1x1_bnrelu_h16(const void* A16, long lda16, const void* W16, void* out16, long ldc16, long M, int N, int K,
                           const float* scale, const float* shift, const float* out_scale, const float* out_shift,
                           gnx_stream_t stream);
/* Transitions of config 5 in two steps (pool-first: norm -> relu -> 2x2 mean, then the 1x1 conv on a quarter of the rows,
 * densenet.py:47-54): gnx_bnrelu_avgpool2_h16 writes the pooled, activated block buffer [imgs*(S/2)^2][C] halves;
 * gnx_conv1x1_bnrelu_h16 with scale = shift = NULL (no prologue) and out_scale = out_shift = NULL (no consumer activation)
 * multiplies it. */
int gnx_bnrelu_avgpool2_h16(const void* in16, long ldi, void* out16, long ldo, long imgs, int C, int S, const float* scale,
                            const float* shift, gnx_stream_t stream);
int gnx_conv3x3_f16_dma_h(const void* A16, long lda16, const void* Wr16, void* out16, long ldc16, long M, int N, int K, int S,
                          gnx_stream_t stream);
int gnx_bnrelu_avgpool_h16(const void* in16, long ldi, float* out, long ldo, long imgs, int C, int S2, const float* scale,
                           const float* shift, gnx_stream_t stream);
/* A DenseNet transition of the fp16 path as ONE kernel on channel-blocked fp16 buffers (transition_f16.hip):
 * norm -> relu -> conv 1x1 (K -> N) -> avgpool 2x2 (/root/reference/gridnext/densenet.py:47-53), evaluated pool-first; the
 * pooled activated operand exists in the LDS only (bit-identical to gnx_bnrelu_avgpool2_h16_cb's output).
 * gnx_transition_f16_pack: conv.weight [N][K] fp32 -> wp (N * K halves, MFMA fragment order).
 * gnx_transition_f16: X16 [K / 32][rows_in][32] -> channel blocks [0, N / 32) of Y16 [..][rows_out][32]; S in {8, 16, 32, 64};
 * 32 | K, 64 <= K <= 1024; 128 | N <= 512; 128 | n_img * (S / 2)^2; anything else: GNX_ERR_UNSUPPORTED (callers take the
 * two-kernel form). */
int gnx_transition_f16_pack(const float* w, void* wp, int N, int K, hipStream_t stream);
int gnx_transition_f16(const void* X16, long rows_in, long n_img, int S, int K, int N, const void* wp, const float* scale,
                       const float* shift, void* Y16, long rows_out, hipStream_t stream);
/* The same transition as the TAPED forward of the fp16 gradient path (training.py:164-171 stepping f_opt): additionally stores
 * the pooled activated operand avgpool2(relu(norm(x))) - which the kernel holds in the LDS anyway; bit for bit
 * gnx_bnrelu_avgpool2_h16_cb's output - as a row-major fp16 matrix P16 [n_img * (S / 2)^2][ldp] (columns [0, K); 8 | ldp), the
 * operand of the transition's 1x1 weight gradient.  Everything else is gnx_transition_f16 bit for bit. */
int gnx_transition_f16_tape(const void* X16, long rows_in, long n_img, int S, int K, int N, const void* wp, const float* scale,
                            const float* shift, void* Y16, long rows_out, void* P16, long ldp, gnx_stream_t stream);
/* One whole dense layer of config 5 in ONE kernel (gridnext/densenet.py:35-44: cat -> norm1 -> relu1 -> conv1 -> norm2 ->
 * relu2 -> conv2) on a CHANNEL-BLOCKED fp16 block buffer X16 [channels / 32][rows_total][32] - element (row, c) at
 * (c >> 5) * rows_total * 32 + row * 32 + (c & 31), so the 32 channels a K-loop stage needs of consecutive pixels are
 * contiguous memory: reads channel blocks [0, K / 32), writes block K / 32; the 128-channel bottleneck lives in LDS only.
 * gnx_dense_layer_f16_pack rounds conv1.weight [128][K] and conv2.weight [32][128][3][3] (fp32, torch layouts) to fp16 ONCE
 * into the MFMA-fragment order the kernel streams (w1p: 128 * K halves, w2p: 36 864 halves).  bn_size * growth_rate = 128,
 * growth_rate = 32; S in {4, 8, 16, 32, 64}; 32 | K, 64 <= K <= 1024; 128 | n_img * S * S; anything else: GNX_ERR_UNSUPPORTED (callers
 * fall back to the two-kernel pair on row-major buffers).  The `_cb` entry points are the neighbours of that layout: the
 * fp16 stem and the transition's 1x1 conv STORE channel-blocked, the transition's pooling pass and the final pool READ it. */
int gnx_dense_layer_f16_pack(const float* w1, const float* w2, void* w1p, void* w2p, int K, gnx_stream_t stream);
int gnx_dense_layer_f16(void* X16, long rows_total, long n_img, int S, int K, const void* w1p, const void* w2p,
                        const float* scale1, const float* shift1, const float* scale2, const float* shift2, gnx_stream_t stream);
/* The same layer as the TAPED forward of the fp16 gradient path (training.py:164-171 stepping f_opt): additionally stores the
 * activated bottleneck relu2(norm2(conv1(.))) - the tile the kernel holds in the LDS anyway - as A16 [4][a_rows_total][32]
 * halves (channel-blocked like X16), the operand of conv2's weight gradient and of norm2's adjoint.  Everything else is
 * gnx_dense_layer_f16 bit for bit. */
int gnx_dense_layer_f16_tape(void* X16, long rows_total, long n_img, int S, int K, const void* w1p, const void* w2p,
                             const float* scale1, const float* shift1, const float* scale2, const float* shift2, void* A16,
                             long a_rows_total, gnx_stream_t stream);
/* Which kernel gnx_dense_layer_f16 / _tape run on 64 x 64 and 32 x 32 maps with K <= 512 (densenet.py:35-44 either way; a
 * process-wide tuning switch): 0 (default) keeps W2 and the bottleneck tile in LDS; 1 = the K-SPLIT form
 * (csrc/dense_layer_f16_ks.hip): the four waves of a workgroup own 32 bottleneck channels each, conv1's accumulators become
 * conv2's operand in registers, W2's slice lives in registers, dx taps are DPP lane shifts, the four partial sums are added
 * through LDS in a fixed order.  Same rounding points; conv2's fp32 sum runs in another order (last-bit differences before the
 * fp16 rounding of the output).  Any other value: GNX_ERR_BAD_ARG. */
int gnx_dense_layer_f16_set_form(int form);
int gnx_conv_stem_bnrelu_maxpool_f16mul_cb(const void* x, int x_is_u8, const float* w, void* out16, long rows_total, long imgs,
                                           int Cin, int H, int W, int O, int KH, int KW, int stride, int pad, const float* scale,
                                           const float* shift, const float* norm, gnx_stream_t stream);
int gnx_conv1x1_bnrelu_h16_cb(const void* A16, long lda16, const void* W16, void* out16, long rows_total, long M, int N, int K,
                              const float* scale, const float* shift, const float* out_scale, const float* out_shift,
                              gnx_stream_t stream);
int gnx_bnrelu_avgpool2_h16_cb(const void* in16, long rows_total, void* out16, long ldo, long imgs, int C, int S,
                               const float* scale, const float* shift, gnx_stream_t stream);
int gnx_bnrelu_avgpool_h16_cb(const void* in16, long rows_total, float* out, long ldo, long imgs, int C, int S2,
                              const float* scale, const float* shift, gnx_stream_t stream);

/* ---- DenseNet-BC backward (the gradients torch.autograd derives for gridnext/densenet.py) -----------------------------
 * Data gradients reuse gnx_conv1x1_bnrelu / gnx_conv3x3_bnrelu with weights transformed by gnx_transpose_weight
 * ([N][K] -> [K][N]) and gnx_repack_conv3x3_bwd ([N][K][3][3] -> [8-tap][K][N]).
 * gnx_wgrad_bnrelu: dW of conv1 / conv2 / transition conv (taps 1|9, pool), BN+ReLU prologue recomputed while staging.
 * gnx_conv0_wgrad : dW of features.conv0 from NCHW patches.  Pool adjoints: maxpool (pool0), avgpool2 (transition),
 * rows_broadcast (adaptive_avg_pool). */
long gnx_wgrad_workspace(long M, int N, int K, int taps); /* floats */
int gnx_wgrad_bnrelu(const float* dY, long lddy, const float* X, long ldx, const float* scale, const float* shift,
                     float* dW, float* workspace, long M, int N, int K, int S, int taps, int pool, int accumulate,
                     gnx_stream_t stream);
/* n independent weight gradients of ONE kind as one launch per 24 items plus one batched reduce: exactly
 * gnx_wgrad_bnrelu(item..., taps, pool = 0) for every item, bit for bit (same kernel bodies, slab layout and fixed-order
 * reduce) - the dense layers of a block at a batch of 32 patches are 16-64 workgroups each and cost their launch whatever they
 * compute (densenet.py:35-44 backwards, training.py:45-71).  taps = 9: the same S for all items.  Shapes the transposed-image
 * kernels do not take: GNX_ERR_UNSUPPORTED, nothing launched (make the single calls).  `items`: HOST array. */
typedef struct {
    const float* dY; long lddy; const float* X; long ldx; const float* scale; const float* shift; float* dW; float* workspace;
    long M; int N, K, S, accumulate;
} gnx_wgrad_item;
int gnx_wgrad_bnrelu_batch(const void* items, int n, int taps, gnx_stream_t stream);
int gnx_transpose_weight(const float* w, float* wt, int N, int K, gnx_stream_t stream);
/* Every weight re-layout of one kind for a whole network in one launch (a training step re-lays all conv weights out after
 * each optimizer step: 3 x 58 launches for DenseNet-121).  `table`: n entries {const float* src; float* dst; int N; int K;}
 * (24 bytes each, DEVICE memory); kind 0 = gnx_repack_conv3x3, 1 = gnx_repack_conv3x3_bwd, 2 = gnx_transpose_weight. */
int gnx_relayout_weights_batch(const void* table, int n, int kind, gnx_stream_t stream);
int gnx_repack_conv3x3_bwd(const float* w, float* wb, int N, int K, gnx_stream_t stream);
int gnx_rows_broadcast(const float* in, long ldi, float* out, long ldo, long imgs, int C, int S2, float alpha,
                       gnx_stream_t stream);
int gnx_avgpool2_bwd(const float* dP, long ldp, float* dA, long lda, long imgs, int C, int S, gnx_stream_t stream);
int gnx_maxpool_bwd(const float* in, long ldi, const float* pooled, long ldp, const float* dOut, long lddo, float* dAct,
                    long lda, long imgs, int C, int Hi, int Wi, const float* scale, const float* shift,
                    gnx_stream_t stream);
/* Pooling by recorded index (the training forward): gnx_bnrelu_maxpool_argmax = gnx_bnrelu_maxpool that also stores, per
 * pooled element, which of its 3x3 window elements (0..8, row-major) is the maximum - the FIRST maximal one of the scan, as
 * torch's max_pool2d (densenet.py:110) picks it; gnx_maxpool_bwd_argmax routes d(pooled) back by that index: no value
 * comparison, no re-read of the conv0 map, ties (constant regions: white background, empty spots) go where torch sends
 * them.  argmax: [imgs*Ho*Wo][C] bytes; 4 | C for the backward. */
int gnx_bnrelu_maxpool_argmax(const float* in, long ldi, float* out, long ldo, unsigned char* argmax, long imgs, int C,
                              int Hi, int Wi, const float* scale, const float* shift, gnx_stream_t stream);
int gnx_maxpool_bwd_argmax(const unsigned char* argmax, const float* dOut, long lddo, float* dAct, long lda, long imgs,
                           int C, int Hi, int Wi, gnx_stream_t stream);
/* The same adjoint carried through norm0 -> relu0 (densenet.py:107-109) with running statistics: dPre = gradient of the
 * conv0 map; the ReLU mask is read off the pooled activated output (`pooled`, ldp), `scale` = gamma / sqrt(var + eps). */
int gnx_maxpool_bwd_argmax_bnrelu(const unsigned char* argmax, const float* dOut, long lddo, const float* pooled, long ldp,
                                  const float* scale, float* dPre, long lda, long imgs, int C, int Hi, int Wi,
                                  gnx_stream_t stream);
long gnx_conv0_wgrad_workspace(long imgs, int H, int W, int O, int KH, int KW, int stride, int pad); /* floats */
int gnx_conv0_wgrad(const float* x, const float* dS, long ldd, float* dW, float* workspace, long imgs, int H, int W,
                    int O, int KH, int KW, int stride, int pad, int accumulate, gnx_stream_t stream);

/* ---- DenseNet-BC backward on the fp16-MFMA path (BASELINE config 5 with f trained: what torch.autograd derives for
 * gridnext/densenet.py:35-54 when training.py:164-171 steps f_opt; BatchNorm on running statistics, training.py:126) --------
 * Tape and gradients are fp16, row-major: block buffer X16 [M][c_total], block gradient G16 (same shape), activated bottleneck
 * A16 [M][128] (conv1's output after norm2 -> relu2, as gnx_conv1x1_bnrelu_h16 stores it), its gradient dB16 [M][128].
 * v_mfma_f32_32x32x16_f16, fp32 accumulation, fp32 parameter gradients.  `ls`: device floats {s, 1/s}, a power-of-two loss
 * scale - every fp16 gradient tensor holds s x the true gradient, every fp32 result is multiplied by 1/s; `flag` (device int,
 * may be NULL) is OR-ed with 1 when a result is not finite.  Workspaces in floats; `accumulate` != 0 adds to the outputs.
 *   gnx_wgrad1x1_f16: dW[N][K] = sum_m dY16[m][n] act(X16[m][k]), act = relu(scale x + shift) or identity (scale NULL)
 *                     - conv1 (:37) and the transition conv (:51, X16 = the pooled activated map of gnx_bnrelu_avgpool2_h16)
 *   gnx_wgrad3x3_f16: dW[32][128][3][3] of conv2 (:40) from dY16 (the layer's 32 gradient columns) and A16, S x S maps
 *   gnx_conv3x3_dgrad_bnrelu_bwd_f16: dB16 = scale2 * conv3x3^T(dY16, W2) * [A16 > 0] (+ dgamma2, dbeta2; gamma2 != 0);
 *                     W2b16 = conv2.weight as [tap][128][32] halves
 *   gnx_conv1x1_dgrad_bnrelu_bwd_f16: G16[:, :K] += scale1 * (dB16 . W1) * [bn1(X16) > 0] (+ dgamma1, dbeta1); W1t16 =
 *                     conv1.weight transposed to [K][128] halves; 32 | K
 *   gnx_tail_bwd_f16: norm_final -> relu -> adaptive_avg_pool (:153-156): G16 [imgs S2][C] = s * scale [bn(X16) > 0] dfeats / S2
 *   gnx_trans_bwd_f16: transition norm -> relu -> avgpool 2x2 (:48-53, pool-first) from the pooled gradient dP16: writes G16
 *   gnx_h16_cols_to_f32: out[M][C] = G16[:, :C] / s (the gradient of the pooled stem map, handed to the fp32 stem adjoints)
 *   gnx_stem_bwd_f16: conv0 -> norm0 -> relu0 -> pool0 (:105-110) differentiated in one pass over the PATCHES (P = 128 / 256,
 *                     64 channels): the conv0 rows are recomputed exactly as gnx_conv_stem_bnrelu_maxpool_f16mul computes them,
 *                     pool0's winners found by torch's first-maximum rule, G16[:, :64] (s x the pooled map's gradient) routed to
 *                     them and contracted with the im2col of the staged rows: dW [64][3][7][7], dgamma / dbeta [64] (gamma != 0) */
long gnx_wgrad1x1_f16_workspace(long M, int N, int K);
int gnx_wgrad1x1_f16(const void* dY16, long lddy, const void* X16, long ldx, const float* scale, const float* shift, float* dW,
                     float* workspace, long M, int N, int K, const float* ls, int accumulate, int* flag, gnx_stream_t stream);
long gnx_wgrad3x3_f16_workspace(long M);
int gnx_wgrad3x3_f16(const void* dY16, long lddy, const void* A16, float* dW, float* workspace, long M, int S, const float* ls,
                     int accumulate, int* flag, gnx_stream_t stream);
long gnx_conv3x3_dgrad_bnrelu_bwd_f16_workspace(long M);
int gnx_conv3x3_dgrad_bnrelu_bwd_f16(const void* dY16, long lddy, const void* W2b16, const void* A16, void* dB16, long M, int S,
                                     const float* scale2, const float* gamma2, const float* beta2, float* dgamma, float* dbeta,
                                     float* workspace, const float* ls, int accumulate, int* flag, gnx_stream_t stream);
long gnx_conv1x1_dgrad_bnrelu_bwd_f16_workspace(long M, int K);
int gnx_conv1x1_dgrad_bnrelu_bwd_f16(const void* dB16, const void* W1t16, const void* X16, long ldx, void* G16, long ldg, long M,
                                     int K, const float* scale, const float* shift, const float* mean, const float* invstd,
                                     float* dgamma, float* dbeta, float* workspace, const float* ls, int accumulate, int* flag,
                                     gnx_stream_t stream);
/* The same pass also producing conv1's weight gradient dW[128][K] (fp32, (+)=; = gnx_wgrad1x1_f16(dB16, X16, scale, shift)) from
 * the tiles it stages: the separate pass over dB16 and X16 is not made.  workspace: gnx_conv1x1_dgrad_wgrad_f16_workspace floats. */
long gnx_conv1x1_dgrad_wgrad_f16_workspace(long M, int K);
int gnx_conv1x1_dgrad_wgrad_bnrelu_bwd_f16(const void* dB16, const void* W1t16, const void* X16, long ldx, void* G16, long ldg,
                                           long M, int K, const float* scale, const float* shift, const float* mean,
                                           const float* invstd, float* dgamma, float* dbeta, float* dW, float* workspace,
                                           const float* ls, int accumulate, int* flag, gnx_stream_t stream);
long gnx_tail_bwd_f16_workspace(long imgs, int C);
int gnx_tail_bwd_f16(const float* dfeats, long ldf, const void* X16, long ldx, void* G16, long ldg, long imgs, int C, int S2,
                     const float* scale, const float* shift, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                     float* workspace, const float* ls, int accumulate, int* flag, gnx_stream_t stream);
long gnx_trans_bwd_f16_workspace(long imgs, int C, int S);
int gnx_trans_bwd_f16(const void* dP16, long ldp, const void* X16, long ldx, void* G16, long ldg, long imgs, int C, int S,
                      const float* scale, const float* shift, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                      float* workspace, const float* ls, int accumulate, int* flag, gnx_stream_t stream);
int gnx_h16_cols_to_f32(const void* G16, long ldg, float* out, long ldo, long M, int C, const float* ls, int* flag,
                        gnx_stream_t stream);
/* `_lb` forms (round 5): the same kernels with the block buffer X16, its gradient G16, the pooled gradient dP16 and the
 * activated bottleneck A16 each addressed through (ld, bs) - element (row, c) at row * ld + (c >> 5) * bs + (c & 31).
 * (ld, 32) is the row-major [rows][ld] matrix of the entry points above; (32, rows_total * 32) is the CHANNEL-BLOCKED form
 * [C / 32][rows_total][32] of gnx_dense_layer_f16_tape, so the gradient path of gridnext/densenet.py:35-54 runs forward and
 * backward on the same buffers (a layer's dY = its own 32-channel block: pass that block's address with lddy = 32).
 * Same workspaces; gnx_conv1x1_dgrad_wgrad_bnrelu_bwd_f16_lb with dW == NULL is the pass without the weight gradient
 * (workspace gnx_conv1x1_dgrad_bnrelu_bwd_f16_workspace). */
/* The fp16 operands of one dense layer's backward from its fp32 weights in one launch (densenet.py:35-44 backwards; bn_size *
 * growth = 128, growth = 32): W1t16 [K][128] = conv1.weight [128][K] transposed, W2b16 [9][128][32] = conv2.weight
 * [32][128][3][3] as [tap][m][n]; each value rounded once. */
int gnx_dense_bwd_f16_pack(const float* w1, const float* w2, void* W1t16, void* W2b16, int K, gnx_stream_t stream);
int gnx_wgrad3x3_f16_lb(const void* dY16, long lddy, const void* A16, long lda, long bsa, float* dW, float* workspace, long M, int S,
                        const float* ls, int accumulate, int* flag, gnx_stream_t stream);
int gnx_conv3x3_dgrad_bnrelu_bwd_f16_lb(const void* dY16, long lddy, const void* W2b16, const void* A16, long lda, long bsa,
                                        void* dB16, long M, int S, const float* scale2, const float* gamma2, const float* beta2,
                                        float* dgamma, float* dbeta, float* workspace, const float* ls, int accumulate, int* flag,
                                        gnx_stream_t stream);
/* conv2's WHOLE backward in one pass over dY16 and A16 (round 5): dB16, dgamma2 / dbeta2 (= gnx_conv3x3_dgrad_bnrelu_bwd_f16_lb)
 * and dW2 [32][128][3][3] (= gnx_wgrad3x3_f16_lb) - a workgroup of eight waves stages each 128-pixel tile once, four waves take
 * the data gradient, four the weight gradient: 576 instead of 896 bytes per pixel.  S in {4, 8, 16, 32, 64}, 128 | M; otherwise
 * GNX_ERR_UNSUPPORTED (make the two calls).  workspace: gnx_conv3x3_bwd_f16_workspace(M) floats. */
long gnx_conv3x3_bwd_f16_workspace(long M);
int gnx_conv3x3_bwd_f16_lb(const void* dY16, long lddy, const void* W2b16, const void* A16, long lda, long bsa, void* dB16, float* dW,
                           long M, int S, const float* scale2, const float* gamma2, const float* beta2, float* dgamma,
                           float* dbeta, float* workspace, const float* ls, int accumulate, int* flag, gnx_stream_t stream);
int gnx_conv1x1_dgrad_wgrad_bnrelu_bwd_f16_lb(const void* dB16, const void* W1t16, const void* X16, long ldx, long bsx, void* G16,
                                              long ldg, long bsg, long M, int K, const float* scale, const float* shift,
                                              const float* mean, const float* invstd, float* dgamma, float* dbeta, float* dW,
                                              float* workspace, const float* ls, int accumulate, int* flag, gnx_stream_t stream);
int gnx_tail_bwd_f16_lb(const float* dfeats, long ldf, const void* X16, long ldx, long bsx, void* G16, long ldg, long bsg, long imgs,
                        int C, int S2, const float* scale, const float* shift, const float* mean, const float* invstd,
                        float* dgamma, float* dbeta, float* workspace, const float* ls, int accumulate, int* flag,
                        gnx_stream_t stream);
int gnx_trans_bwd_f16_lb(const void* dP16, long ldp, long bsp, const void* X16, long ldx, long bsx, void* G16, long ldg, long bsg,
                         long imgs, int C, int S, const float* scale, const float* shift, const float* mean, const float* invstd,
                         float* dgamma, float* dbeta, float* workspace, const float* ls, int accumulate, int* flag,
                         gnx_stream_t stream);
long gnx_stem_bwd_f16_workspace(long imgs, int P);
int gnx_stem_bwd_f16(const float* x, const float* w, const float* scale, const float* shift, const float* gamma, const float* beta,
                     const void* G16, long ldg, float* dW, float* dgamma, float* dbeta, float* workspace, long imgs, int P, int O,
                     const float* ls, int accumulate, int* flag, gnx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GRIDNEXT_HIP_H */
