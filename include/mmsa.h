/*
 * mmsa.h -- C ABI of libmmsa_hip.so: hand-written gfx950 (MI355X) kernels for the CNN multimodal sentiment model
 * and its log-mel front end.
 *
 * The reference (suedaakkoc/multimodal-sentiment-analysis) is pure Python on torch.nn and has no FFI; the entry
 * points below are what a binding for its hot path needs, one per fused operation.  Each comment cites the
 * reference lines (relative to "CNN implementation/") whose arithmetic the entry point performs.
 *
 * Conventions
 *   - every pointer is DEVICE memory unless the name ends in _host or the comment says "host";
 *   - fp32 everywhere, token ids / labels / num_batches_tracked are int64 (long long);
 *   - activations are channels-last [B][H][W][C]; weights arrive in the reference's layouts;
 *   - `stream` is a hipStream_t (0 = default stream); kernels are only enqueued, never synchronised;
 *   - return value: 0 on success, MMSA_ERR_ARG (-1) for rejected arguments (nothing was launched),
 *     MMSA_ERR_LAUNCH (-2) if HIP reported a launch error;
 *   - the library allocates no device memory: scratch comes from the caller, sized by the *_floats helpers
 *     (one entry point, mmsa_ln_scale_bwd_static, keeps its scratch in a device array inside the library instead);
 *   - entry points are stateless and re-entrant; two calls that share output or scratch buffers must be ordered
 *     by the caller (same stream).
 */
#ifndef MMSA_H
#define MMSA_H

#ifdef __cplusplus
extern "C" {
#endif

#define MMSA_OK 0
#define MMSA_ERR_ARG (-1)
#define MMSA_ERR_LAUNCH (-2)
#define MMSA_ERR_LIMIT (-3)

#define MMSA_API __attribute__((visibility("default")))

/* Dense-layer epilogues (mmsa_dense_fwd / mmsa_dense_bwd). */
#define MMSA_EPI_NONE 0     /* y = x W^T + b                                              */
#define MMSA_EPI_LN_RELU 1  /* y = relu(LayerNorm(x W^T + b))   multimodal_model.py:237-249,395-401 */
#define MMSA_EPI_RELU_LN 2  /* y = LayerNorm(relu(x W^T + b))   multimodal_model.py:382-388          */
#define MMSA_EPI_RELU 3     /* y = relu(x W^T + b)                                        */
#define MMSA_EPI_LN 4       /* y = LayerNorm(x W^T + b)         Pretrained model implementations/early_fusion.py:137-138 */
#define MMSA_EPI_TANH 5     /* y = tanh(x W^T + b)              early_fusion.py:103-104   */
#define MMSA_EPI_SIGMOID 6  /* y = sigmoid(x W^T + b)           early_fusion.py:110-113   */
#define MMSA_EPI_LN_GELU 7  /* y = gelu(LayerNorm(x W^T + b))   early_fusion.py:213-219   */

/* ---- log-mel front end: preprocess.py:151-168 (pad/trim, MelSpectrogram, AmplitudeToDB, normalize, NaN scrub)
 *      and :111-123 (z-score, unbiased std, clamp +-6). ------------------------------------------------------- */
/* host helper: periodic Hann window [1024], FFT twiddles tw512 [256][2], tw1024 [513][2] (cos, -sin), in double */
MMSA_API int mmsa_logmel_tables_host(float* window_host, float* tw512_host, float* tw1024_host);
/* wave: [batch] rows of n_valid samples (row stride wave_stride floats) -> out [batch][1][64][1 + n_target/512].
 * melfb [513][64] and melrange [64][2] (first bin, one past last bin with weight) describe the filterbank.
 * specmask: null, or [batch][4] ints {first mel row, rows, first frame, frames}: the training-time frequency / time
 * masks of preprocess.py:92-109, zero-filled after the dB floor and before normalisation (:162-165). */
MMSA_API int mmsa_logmel_fwd(const float* wave, long wave_stride, int n_valid, int n_target, int batch,
                             const float* window, const float* tw512, const float* tw1024, const float* melfb,
                             const int* melrange, float top_db, const int* specmask, float* out, void* stream);
/* Training-time waveform augmentations of preprocess.py:56-90,137-140 (time shift, Gaussian noise, resample-by-index
 * "pitch shift") for a batch, fused with the pad / trim to n_target (:151-156).  params: [batch][4] doubles drawn on
 * the host {shift mode 0 none / 1 right / 2 left, shift in samples, noise on, index step or 0}.  The noise comes from
 * a counter-based generator keyed by (seed, utterance, sample), not from torch's stream.  lens: null, or [batch] ints:
 * row b holds an utterance of lens[b] <= n_in samples and is augmented as a signal of that length. */
MMSA_API int mmsa_augment_wave(const float* in, long in_stride, int n_in, const int* lens, const double* params,
                               float noise_factor, unsigned long long seed, float* out, int n_target, int batch,
                               void* stream);
/* Resampling ahead of the mel stage: torchaudio.transforms.Resample's sinc interpolation (Hann window,
 * lowpass_filter_width 6, rolloff 0.99) as applied at preprocess.py:131-134, fused with the mono mix (:127-129) and
 * the pad / trim to n_out samples (:151-156).  mmsa_resample_filters_host builds the filter bank kernT[K][ld] on the
 * host (ld = new rounded up to a multiple of 4, padding zero, 16-byte aligned on the device; null kernT: size query); mmsa_resample_fwd maps pcm [batch][channels][T] to out [batch][n_out]. */
MMSA_API int mmsa_resample_filters_host(int orig_freq, int new_freq, int lowpass_filter_width, double rolloff,
                                        float* kernT, int* orig_out, int* new_out, int* width_out, int* K_out);
MMSA_API int mmsa_resample_fwd(const float* pcm, int channels, long T, long stride_b, const float* kernT, int orig,
                               int nw, int width, int K, float* out, int n_out, int batch, void* stream);
/* Noise reduction ahead of the pad / trim: replaces the per-utterance `nr.reduce_noise(y=audio_np, sr=SAMPLE_RATE)` of
 * preprocess.py:142-148 (package defaults: non-stationary spectral gating, n_fft 1024, hop 256, chunk padding 30 000)
 * for a batch of equally long rows.  mmsa_denoise_plan (host only) returns, for one chunk of L output samples with
 * `pad` samples of context per side, the STFT frame count, the first frame and the number of frames that reach the
 * output, and the scratch need in floats per utterance.  mmsa_denoise_fwd denoises samples [c0, c0+L) of every row of
 * wave [batch][T] (row stride `stride`) into out[b][c0 .. c0+L) (row stride out_stride); the samples of the row up to
 * pad before and after the chunk are its context, zeros beyond the row.  window / tw512 / tw1024: the tables of
 * mmsa_logmel_tables_host.  vt [nt], vf [nf]: normalised time / frequency taps of the mask-smoothing filter (odd
 * counts, at most 33 / 65).  b_iir: coefficient of the one-pole time-smoothing filter; thresh, slope: sigmoid gate
 * parameters; prop: share of the mask applied.  ws: batch * ws_floats floats, 16-byte aligned.  lens: null, or [batch]
 * valid lengths (1..T) when the rows hold utterances of different lengths (then c0 = 0, L = T): each row is gated as
 * a signal of its own length and its output is zero behind it. */
MMSA_API int mmsa_denoise_plan(long L, int pad, int* nseg, int* f_lo, int* nfr, long* ws_floats);
MMSA_API int mmsa_denoise_fwd(const float* wave, long stride, long T, long c0, long L, int pad, int batch,
                              const int* lens, const float* window, const float* tw512, const float* tw1024,
                              const float* vt, int nt, const float* vf, int nf, double b_iir, float thresh,
                              float slope, float prop, float* ws, float* out, long out_stride, void* stream);


/* ---- convolutions as implicit GEMM on fp32 MFMA: multimodal_model.py:48-59,64-68 (ResidualBlock convs and 1x1
 *      shortcut), :152-155,188 (text Conv1d); backward = autograd of the same (multimodal_train.py:192). ------ */
/* w [Cout][Cin][ntaps] -> fwd [ntaps][Cout][Cin] and/or dgrad [ntaps][Cin][Cout] (either may be null) */
MMSA_API int mmsa_conv_pack_weights(const float* w, float* fwd, float* dgrad, int Cout, int Cin, int ntaps,
                                    void* stream);
/* the same for n tensors in one launch; descs is a DEVICE array of records
 * {const float* w; float* fwd; float* dgrad; int Cout, Cin, ntaps, pad;} of mmsa_conv_pack_desc_bytes() bytes each */
MMSA_API int mmsa_conv_pack_desc_bytes(void);
MMSA_API int mmsa_conv_pack_weights_multi(const void* descs, int n, void* stream);
/* LDS ring depth the forward / data-gradient launch picks for this geometry, number of problems per launch and K loop
 * length (kchunks = taps x Cin / 32, one tap more with a second operand pair): the last template argument of the
 * kernel name a profiler prints */
MMSA_API int mmsa_conv_ring_buffers(int B, int H, int W, int Cout, int nproblems, int kchunks);
/* 1 when the launchers below hand this geometry and offset table to the kernel that stages an input strip once per
 * row of three taps (conv_igemm3_kernel: the 3x3 / pad 1 convolutions of multimodal_model.py:48-59 and their data
 * gradients), 0 when the general kernel runs (conv_igemm_kernel: any offset table, e.g. the text Conv1d) */
MMSA_API int mmsa_conv_shares_tap_rows(int B, int H, int W, int Cout, int ntaps, const int* dh_host,
                                       const int* dw_host);
/* tile height the forward kernel uses for this geometry; statistics rows = B * ceil(H*W / tile height) */
MMSA_API int mmsa_conv_tile_rows(int B, int H, int W, int Cout);
/* Y[B,H,W,Cout] (= or +=) sum_t X[b, h+dh[t], w+dw[t], :] . Wp[t][:, :]; reads outside the image are zero.
 * dh/dw are host arrays.  stats (optional): [2][Cout][B*tiles] per-tile column sums and sums of squares. */
MMSA_API int mmsa_conv_fwd(const float* X, const float* Wp, float* Y, float* stats, int B, int H, int W, int Cin,
                           int Cout, int ntaps, const int* dh_host, const int* dw_host, int accumulate, void* stream);
/* mmsa_conv_fwd for the data gradient of a block's second convolution (autograd of multimodal_model.py:64-65): Y is
 * the gradient with respect to relu(bn1(x)); the launch also leaves, in stats ([2][Cout][B*tiles], the layout above),
 * the per-tile column sums of g = Y [bn1(x) > 0] and of g * xhat -- the two reductions of that BatchNorm's backward --
 * so mmsa_bn_relu_bwd_finish needs no pass of its own over Y and x for them.  x [B,H,W,Cout]; coeff: bn1's [4][Cout]
 * table.  3x3 offset tables with Cout % 64 == 0 only (mmsa_conv_shares_tap_rows tells): MMSA_ERR_ARG otherwise. */
MMSA_API int mmsa_conv_fwd_bnrelu_bwd_stats(const float* X, const float* Wp, float* Y, float* stats, const float* x,
                                            const float* coeff, int B, int H, int W, int Cin, int Cout, int ntaps,
                                            const int* dh_host, const int* dw_host, void* stream);
/* Y = conv(X, Wp) + X2 . Wp2 in one launch (X2 [B,H,W,Cin], Wp2 [Cout][Cin]): the data gradient of a residual block's
 * input, 3x3 branch plus 1x1 shortcut (autograd of multimodal_model.py:64-70) */
MMSA_API int mmsa_conv_fwd_dual(const float* X, const float* Wp, const float* X2, const float* Wp2, float* Y, int B,
                                int H, int W, int Cin, int Cout, int ntaps, const int* dh_host, const int* dw_host,
                                void* stream);
/* mmsa_conv_fwd_dual with the K range split over two workgroups per output tile (lower | upper half of the channels,
 * every tap and the second pair in both; the general kernel splits taps 0..4 | taps 5..8 plus the second pair), for
 * maps with at most one tile per CU of the chip (mmsa_conv_fwd_dual_prefers_split: 1 / 0).
 * Y MUST BE ZERO on entry: both halves add their part with hardware float atomics -- two addends per element, so the
 * result does not depend on their order.  3x3 offsets only, Cout a multiple of 64. */
MMSA_API int mmsa_conv_fwd_dual_prefers_split(int B, int H, int W, int Cout);
MMSA_API int mmsa_conv_fwd_dual_split(const float* X, const float* Wp, const float* X2, const float* Wp2,
                                      float* Y_zeroed, int B, int H, int W, int Cin, int Cout, int ntaps,
                                      const int* dh_host, const int* dw_host, void* stream);
/* n (<= 4) convolutions of one input in one launch -- the text branch's filter widths (multimodal_model.py:186-188),
 * or a residual block's first 3x3 convolution together with its 1x1 shortcut (:48-49,64-68): problem i uses
 * Wp_host[i], entries tap0_host[i] .. tap0_host[i] + ntaps_host[i] - 1 of the offset table (tap0_host null: 0) and
 * writes Y_host[i]; stats_host (null, or n pointers) receives each problem's statistics as in mmsa_conv_fwd.
 * bnrelu_host (null, or n pointers, each null or a [4][Cout] BatchNorm coefficient table as mmsa_bn_finalize writes
 * it): problem i stores max(scale * conv + shift, 0) -- eval-mode BatchNorm + ReLU (multimodal_model.py:64) folded
 * into the convolution, for inference, where the table comes from the running statistics (stats_host must be null
 * with it). */
MMSA_API int mmsa_conv_fwd_multi(const float* X, const float* const* Wp_host, float* const* Y_host,
                                 const int* ntaps_host, const int* tap0_host, float* const* stats_host,
                                 const float* const* bnrelu_host, int n, int B, int H, int W, int Cin, int Cout,
                                 int ntable, const int* dh_host, const int* dw_host, void* stream);
MMSA_API long mmsa_conv_wgrad_workspace_floats(int B, int H, int W, int Cin, int Cout, int ntaps, int* nsplit_host);
/* dW[Cout][Cin][ntaps] (= or +=) sum_{b,h,w} X[b,h+dh,w+dw,cin] * dY[b,h,w,cout], in two launches on one stream:
 * mmsa_conv_wgrad writes partial slabs into workspace, mmsa_conv_wgrad_reduce sums them into dW in fixed order */
MMSA_API int mmsa_conv_wgrad(const float* X, const float* dY, float* workspace, int B, int H, int W, int w_valid,
                             int Cin, int Cout, int ntaps, const int* dh_host, const int* dw_host, void* stream);
MMSA_API int mmsa_conv_wgrad_reduce(const float* workspace, float* dW, int B, int H, int W, int Cin, int Cout,
                                    int ntaps, int accumulate, void* stream);
/* both weight gradients that read a residual block's input X (3x3 convolution from dY9, 1x1 shortcut from dY1;
 * multimodal_model.py:48-49,64-68 under autograd) in one launch plus one reduction launch; ws9 / ws1 as sized by
 * mmsa_conv_wgrad_workspace_floats for 9 and 1 taps */
MMSA_API int mmsa_conv_wgrad_pair(const float* X, const float* dY9, const float* dY1, float* ws9, float* ws1,
                                  float* dW9, float* dW1, int B, int H, int W, int Cin, int Cout, const int* dh_host,
                                  const int* dw_host, void* stream);
/* The same gradients by the second-generation kernel (one workgroup per CU, operands streamed into LDS by DMA under
 * the matrix work): mmsa_conv_wgrad_slabs writes *nslabs_host partial slabs [9][Cin][Cout] (and, when dY_sc is given,
 * as many [Cin][Cout] slabs of the 1x1 shortcut that reads the same X; multimodal_model.py:56-59) as sized by
 * mmsa_conv_wgrad_slab_floats (returns the floats of the 3x3 slabs; MMSA_ERR_LIMIT if a strip of the map does not fit
 * in LDS); mmsa_conv_wgrad_reduce_multi sums the slabs of any number of layers into dW[Cout][Cin][ntaps] in one launch
 * (fixed order).  descs_device: n records {const float* slabs; float* dW; int nslabs, ntaps, Cin, Cout, block0,
 * nblocks} of mmsa_conv_wgrad_reduce_desc_bytes() bytes, nblocks = ceil(ntaps*Cin*Cout/256), block0 = sum of the
 * earlier nblocks, total_blocks = sum of all. */
MMSA_API long mmsa_conv_wgrad_slab_floats(int B, int H, int W, int Cin, int Cout, int with_shortcut, int* nslabs_host);
MMSA_API int mmsa_conv_wgrad_slabs(const float* X, const float* dY, const float* dY_sc, float* slabs, float* slabs_sc,
                                   int B, int H, int W, int Cin, int Cout, const int* dh_host, const int* dw_host,
                                   void* stream);
MMSA_API int mmsa_conv_wgrad_reduce_desc_bytes(void);
MMSA_API int mmsa_conv_wgrad_reduce_multi(const void* descs_device, int n, int total_blocks, void* stream);

/* ---- audio branch, bandwidth-bound parts: multimodal_model.py:78-83 (stem), :27-42 (SEBlock), :64-70
 *      (ResidualBlock tail), :87-96,118-127 (pools, global average). ----------------------------------------- */
/* stem Conv2d(1,32,3,pad 1,bias): partial sums / sums of squares of the conv output per group of image rows,
 * partial [2][32][rows] with rows = mmsa_stem_stats_rows(B, H) (the row count mmsa_bn_finalize then takes) */
MMSA_API int mmsa_stem_stats_rows(int B, int H);
MMSA_API int mmsa_stem_stats(const float* x, const float* w, const float* bias, float* partial, int B, int H, int W,
                             void* stream);
/* stem conv + BN (coeff [4][32]) + ReLU + MaxPool2d(2) -> out [B][H/2][W/2][32] */
MMSA_API int mmsa_stem_apply_pool(const float* x, const float* w, const float* bias, const float* coeff, float* out,
                                  int B, int H, int W, void* stream);
/* stem backward from the pooled gradient; scratch >= B*(H/2)*384 + 64 floats */
MMSA_API int mmsa_stem_bwd(const float* x, const float* w, const float* bias, const float* coeff, const float* dout,
                           float* dW, float* dbias, float* dgamma, float* dbeta, float* scratch, int B, int H, int W,
                           int training, void* stream);
/* BatchNorm coefficients [4][C] = mean, invstd, scale, shift from partial sums [rows][2][C] (training; also updates
 * running_mean/var with momentum and increments *nbt) or from the running statistics (eval).
 * bsum (optional) [B][C]: per-utterance sums of the raw input (rows = B * tiles_per_img). */
MMSA_API int mmsa_bn_finalize(const float* partial, int rows, int tiles_per_img, double count, const float* gamma,
                              const float* beta, float* running_mean, float* running_var, long long* nbt,
                              float momentum, float eps, int training, float* coeff, float* bsum, int C, void* stream);
/* the same for two BatchNorm layers over maps of one geometry (host arrays of two pointers each; no bsum) */
MMSA_API int mmsa_bn_finalize_pair(const float* const* partial_host, int rows, int tiles_per_img, double count,
                                   const float* const* gamma_host, const float* const* beta_host,
                                   float* const* running_mean_host, float* const* running_var_host,
                                   long long* const* nbt_host, float momentum, float eps, int training,
                                   float* const* coeff_host, int C, void* stream);
MMSA_API int mmsa_bn_relu_apply(const float* x, const float* coeff, float* y, long npos, int C, void* stream);
/* squeeze-excite gate: gap = BN(mean_hw raw), hid = relu(W1 gap), gate = sigmoid(W2 hid) */
MMSA_API int mmsa_se_gate(const float* bsum, const float* coeff, const float* W1, const float* W2, int HW, float* gap,
                          float* hid, float* gate, int B, int C, int Ch, void* stream);
/* out[B,H/2,W/2,C] = maxpool2( relu( BN2(c2) * gate + BN_s(sc) ) ) */
MMSA_API int mmsa_res_gate_relu_pool(const float* c2, const float* sc, const float* coeff2, const float* coeffs,
                                     const float* gate, float* out, int B, int H, int W, int C, void* stream);
/* the same for the last block of the audio branch, whose pooled output only feeds AdaptiveAvgPool2d(1)
 * (multimodal_model.py:123-125): feat[B][C] = mean over the pooled positions, no pooled map is stored */
MMSA_API int mmsa_res_gate_relu_pool_mean(const float* c2, const float* sc, const float* coeff2, const float* coeffs,
                                          const float* gate, float* feat, int B, int H, int W, int C, void* stream);
/* mmsa_se_gate followed by mmsa_res_gate_relu_pool (feat null) or mmsa_res_gate_relu_pool_mean (out null) in one
 * launch, the same values bit for bit: every workgroup works its utterance's gate out for itself.
 * multimodal_model.py:38-42,64-70,123-125. */
MMSA_API int mmsa_res_se_gate_relu_pool(const float* c2, const float* sc, const float* coeff2, const float* coeffs,
                                        const float* bsum, const float* W1, const float* W2, float* gap, float* hid,
                                        float* gate, float* out, float* feat, int B, int H, int W, int C, int Ch,
                                        void* stream);
/* the block tail without the pooling, every part optional: out = [relu]((a*scale2+shift2)*gate + (sc*scale_s+shift_s));
 * null coefficient table = identity, null gate = 1, null sc = 0: SEBlock.forward alone (multimodal_model.py:38-42:
 * x * gate) and ResidualBlock.forward alone (:64-70, identity or projection shortcut) */
MMSA_API int mmsa_gate_affine_add(const float* a, const float* coeff2, const float* gate, const float* sc,
                                  const float* coeffs, float* out, int B, int P, int C, int relu, void* stream);
MMSA_API int mmsa_gap_fwd(const float* x, float* out, int B, int P, int C, void* stream);
MMSA_API int mmsa_gap_bwd(const float* dout, float* dx, int B, int P, int C, void* stream);
MMSA_API long mmsa_resblock_bwd_scratch_floats(int B, int H, int W, int C);
/* backward of pool + ReLU + residual add + SE + bn2 / shortcut BN: dout -> d_c2, d_sc and parameter gradients */
MMSA_API int mmsa_resblock_bwd_ew(const float* dout, const float* c2, const float* sc, const float* coeff2,
                                  const float* coeffs, const float* gate, const float* hid, const float* gap,
                                  const float* bsum, const float* W1, const float* W2, float* d_c2, float* d_sc,
                                  float* dgamma2, float* dbeta2, float* dgammas, float* dbetas, float* dW1, float* dW2,
                                  float* scratch, int B, int H, int W, int C, int training, void* stream);
/* mmsa_resblock_bwd_ew for that last block: dfeat [B][C] is the gradient with respect to the pooled mean (autograd of
 * multimodal_model.py:123-125), no gradient map of the pooled output is formed */
MMSA_API int mmsa_resblock_bwd_ew_from_mean(const float* dfeat, const float* c2, const float* sc, const float* coeff2,
                                            const float* coeffs, const float* gate, const float* hid,
                                            const float* gap, const float* bsum, const float* W1, const float* W2,
                                            float* d_c2, float* d_sc, float* dgamma2, float* dbeta2, float* dgammas,
                                            float* dbetas, float* dW1, float* dW2, float* scratch, int B, int H, int W,
                                            int C, int training, void* stream);
/* mmsa_resblock_bwd_ew for a block with no pooling behind it -- ResidualBlock.forward used on its own, autograd of
 * multimodal_model.py:64-70: dout is [B][H][W][C] (any H, W >= 2).  training: bit 0 batch statistics; bit 1 the shortcut
 * is a fixed affine map whatever bit 0 says (the identity shortcut of :56-61 when the channel count stays: pass
 * coeffs = [0, 1, 1, 0] per channel, dgammas / dbetas are then scratch) */
MMSA_API long mmsa_resblock_bwd_unpooled_scratch_floats(int B, int H, int W, int C);
MMSA_API int mmsa_resblock_bwd_ew_unpooled(const float* dout, const float* c2, const float* sc, const float* coeff2,
                                           const float* coeffs, const float* gate, const float* hid, const float* gap,
                                           const float* bsum, const float* W1, const float* W2, float* d_c2,
                                           float* d_sc, float* dgamma2, float* dbeta2, float* dgammas, float* dbetas,
                                           float* dW1, float* dW2, float* scratch, int B, int H, int W, int C,
                                           int training, void* stream);
/* dW1, dW2 may both be null in mmsa_resblock_bwd_ew: the squeeze-excite weight gradients are then left to this call
 * (same scratch and geometry), which can run on another stream since nothing but the optimiser reads them */
MMSA_API int mmsa_resblock_bwd_se_wgrad(const float* scratch, const float* hid, const float* gap, float* dW1,
                                        float* dW2, int B, int H, int W, int C, void* stream);
MMSA_API long mmsa_bn_relu_bwd_scratch_floats(long npos, int C);
/* in place: gradient w.r.t. relu(BN(c1)) -> gradient w.r.t. c1; also dgamma, dbeta */
MMSA_API int mmsa_bn_relu_bwd(float* dh, const float* c1, const float* coeff, float* dgamma, float* dbeta,
                              float* scratch, long npos, int C, int training, void* stream);
/* the same with its two reductions already done (autograd of multimodal_model.py:64): partial [2][C][rows] holds the
 * per-tile sums mmsa_conv_fwd_bnrelu_bwd_stats left behind; cb: 3 C floats of scratch */
MMSA_API int mmsa_bn_relu_bwd_finish(float* dh, const float* c1, const float* coeff, const float* partial, int rows,
                                     float* dgamma, float* dbeta, float* cb, long npos, int C, int training,
                                     void* stream);

/* ---- text branch: multimodal_model.py:149,179 (embedding gather, bit-exact), :186-197 (ReLU, max over time). */
MMSA_API int mmsa_embedding_gather(const long long* ids, const float* table, float* out, long ntok, int E, long vocab,
                                   int* err, void* stream);
MMSA_API int mmsa_bias_relu_maxtime(const float* y, const float* bias, float* feat, int* arg, int B, int L, int Lout,
                                    int F, int ldf, void* stream);
MMSA_API int mmsa_bias_relu_maxtime_multi(const float* const* y_host, const float* const* bias_host,
                                          const int* Lout_host, int n, float* feat, int* arg, int B, int L, int F,
                                          int ldf, void* stream);
/* sparse backward of conv + ReLU + max over time + gather; ksize/feat_off/wp/dW/dbias are host arrays
 * (wp/dW/dbias: host arrays of device pointers); dtable [vocab][E]: the rows of the batch's tokens are overwritten
 * (the caller clears the table beforehand); rows == padding_idx get nothing; tokens outside [0, vocab) write nothing
 * and set *err (optional device int) to 1, where the reference's nn.Embedding raises IndexError.  A token that
 * occurs several times in the batch has its rows summed in ascending position, as the reference's CPU
 * embedding backward does: the result is the same bits on every run.  groups: what mmsa_token_groups computed for
 * these ids; drow: B*L*E floats of scratch.
 * mmsa_token_groups sorts the token positions into groups of equal id (it depends on the ids only, so it can run
 * beside the forward pass): groups 5*ntok ints, id_table 2*vocab ints (zero before the first call; every call leaves
 * them zero again) */
MMSA_API int mmsa_token_groups(const long long* ids, long ntok, long vocab, long padding_idx, int* groups,
                               int* id_table, void* stream);
MMSA_API int mmsa_text_bwd(const float* dfeat, const float* feat, const int* arg, const float* emb,
                           const long long* ids, int nconv, const int* ksize_host, const int* feat_off_host,
                           const float* const* wp_host, float* const* dW_host, float* const* dbias_host, float* dtable,
                           float* drow, const int* groups, int B, int L, int E, int F, int ldf,
                           long padding_idx, long vocab, int* err, void* stream);

/* ---- fusion head and classifiers: multimodal_model.py:202-225, :228-285, :365-404, :448-454. ---------------- */
/* Dropout in this family (mmsa_dense_fwd/bwd, mmsa_bn1d_relu_fwd/bwd, mmsa_dropout): the mask of element i is a hash of
 * (seed, i).  step: null, or a device float holding a step counter (the optimiser's) that the kernels mix into the
 * seed, so that a launch sequence replayed from a HIP graph -- whose scalar arguments are frozen -- draws new masks
 * every step; forward and backward of one step read the same counter value. */
MMSA_API int mmsa_dense_fwd(const float* X, int ldx, const float* W, const float* bias, const float* gamma,
                            const float* beta, float* Y, int ldy, float* pre, float* stats, int B, int K, int N,
                            int epilogue, float drop_p, unsigned long long seed, const float* step, void* stream);
MMSA_API int mmsa_dense_bwd(const float* dY, int ldy, const float* X, int ldx, const float* pre, const float* stats,
                            const float* W, const float* gamma, const float* beta, float* dPre, float* dLn, float* dX,
                            int lddx, float* dW, float* db, float* dgamma, float* dbeta, int B, int K, int N,
                            int epilogue, float drop_p, unsigned long long seed, int accumulate_dx, const float* step,
                            void* stream);
/* the parameter gradients alone (after mmsa_dense_bwd was called with dW = null) */
MMSA_API int mmsa_dense_bwd_w(const float* dPre, const float* dLn, const float* pre, const float* stats, const float* X,
                              int ldx, float* dW, float* db, float* dgamma, float* dbeta, int B, int K, int N,
                              int epilogue, void* stream);
/* y = LayerNorm(x) * scale[0] * keep[0]   (keep optional: the modality-dropout switch, :264-268) */
MMSA_API int mmsa_ln_scale_fwd(const float* x, const float* gamma, const float* beta, const float* scale,
                               const float* keep, float* y, int ldy, float* stats, int B, int N, void* stream);
MMSA_API int mmsa_ln_scale_bwd(const float* dy, int ldy, const float* x, const float* gamma, const float* beta,
                               const float* scale, const float* keep, const float* stats, float* dx, float* dgamma,
                               float* dbeta, float* dscale, float* scratch, int B, int N, void* stream);
/* mmsa_ln_scale_bwd without the scratch argument.  No counterpart in the reference: it exists for callers that can
 * neither allocate device memory nor carry a workspace (code under HIP graph capture, tight serving / training
 * loops, harnesses that forbid hipMalloc).  The scratch is a 4 MiB device array inside the library, one copy per GPU;
 * the call allocates nothing.  B*N + B must not exceed 1 048 576 floats (MMSA_ERR_LIMIT otherwise, nothing launched).
 * The device that owns `stream` must be the current device.  All calls on one GPU share that GPU's array: calls that
 * overlap on one GPU (different streams without ordering) race.  Results are bit-identical to mmsa_ln_scale_bwd.
 * The first call on a device looks that device's copy of the array up (a mutex and one runtime query), which must not
 * happen while a stream capture is in progress: capturing callers call mmsa_static_scratch_init() once per device,
 * with that device current and outside any capture, before the first captured call (it is idempotent and cheap). */
MMSA_API int mmsa_static_scratch_init(void);
MMSA_API int mmsa_ln_scale_bwd_static(const float* dy, int ldy, const float* x, const float* gamma, const float* beta,
                                      const float* scale, const float* keep, const float* stats, float* dx,
                                      float* dgamma, float* dbeta, float* dscale, int B, int N, void* stream);
/* ---- fused fusion head of the 'both' model: multimodal_model.py:216-225 (normaliser), :263-285 (fusion),
 *      :393-404 (classifier).  One workgroup per utterance carries the row through every layer.
 * ptrs_host: host array of mmsa_head_both_slots() device pointers (float unless noted), in this order:
 *   inputs and parameters: a_feat[B,A] t_feat[B,T] an_g an_b a_scale[1] tn_g tn_b t_scale[1] keep_a[1]|null
 *     keep_t[1]|null wpa[Hd,A] bpa gpa bepa wpt[Hd,T] bpt gpt bept watt[2,2Hd] batt w0[N0,Hd] b0 g0 be0
 *     w1[N1,N0] b1 g1 be1 w2[C,N1] b2
 *   saved by the forward: an[B,A] tn[B,T] an_stats[2B] tn_stats[2B] pa_pre[B,Hd] pa_stats[2B] pt_pre pt_stats
 *     P[B,2Hd] att[B,2] wts[B,2] fused[B,Hd] c0_pre[B,N0] c0_stats h1[B,N0] c1_pre[B,N1] c1_stats h2[B,N1]
 *     logits[B,C]
 *   backward: dlogits[B,C] (in) c1_dpre c1_dln c0_dpre c0_dln att_dpre[B,2] pa_dpre pa_dln pt_dpre pt_dln
 *     an_dln[B,A] tn_dln[B,T] a_dspart[B] t_dspart[B] d_afeat[B,A] d_tfeat[B,T]
 *   parameter gradients (shapes of the parameters): dwpa dbpa dgpa dbepa dwpt dbpt dgpt dbept dwatt dbatt dw0 db0
 *     dg0 dbe0 dw1 db1 dg1 dbe1 dw2 db2 dan_g dan_b da_scale dtn_g dtn_b dt_scale
 *   optional: labels[B] (int64) -- with it and dlogits set, the forward also writes
 *     dlogits = grad_scale * (softmax(logits) - onehot(labels)) / B, the gradient of the mean cross-entropy of
 *     multimodal_model.py:452-456, so the backward call can follow directly;
 *     step[1] (float) -- a device-resident step counter: forward and backward add step[0] * 0x9E3779B97F4A7C15 to
 *     the seed, so a replayed launch sequence draws new dropout masks every step
 * Slots a call does not use may be null.  A, T, Hd, N0, N1 <= 512 and multiples of 4; weights 16-byte aligned.
 * Dropout (p after each projector and the first classifier layer, p/2 after the second) uses seeds seed+1..seed+4. */
MMSA_API int mmsa_head_both_slots(void);
MMSA_API int mmsa_head_both_fwd(const void* const* ptrs_host, int nptrs, int B, int A, int T, int Hd, int N0, int N1,
                                int C, float drop_p, unsigned long long seed, float grad_scale, void* stream);
MMSA_API int mmsa_head_both_bwd(const void* const* ptrs_host, int nptrs, int B, int A, int T, int Hd, int N0, int N1,
                                int C, float drop_p, unsigned long long seed, void* stream);
MMSA_API int mmsa_head_both_bwd_params(const void* const* ptrs_host, int nptrs, int B, int A, int T, int Hd, int N0,
                                       int N1, int C, void* stream);
MMSA_API int mmsa_fusion_mix_fwd(const float* P, const float* att, float* wts, float* fused, int B, int Hd,
                                 void* stream);
MMSA_API int mmsa_fusion_mix_bwd(const float* dfused, const float* P, const float* wts, float* dP, float* datt, int B,
                                 int Hd, void* stream);
/* attention-fusion head of `Pretrained model implementations/early_fusion.py:127-158`:
 * (w0,w1) = softmax(scores[b]); Wm = [M[:, :Fd]*w0 | M[:, Fd:]*w1]  (:140-152) */
MMSA_API int mmsa_pair_attn_mix_fwd(const float* M, const float* scores, float* wts, float* Wm, int B, int Fd,
                                    void* stream);
MMSA_API int mmsa_pair_attn_mix_bwd(const float* dWm, const float* M, const float* wts, float* dM, float* dscores, int B,
                                    int Fd, int accumulate, void* stream);
/* fused = Wm[:, :Fd]*gate + Wm[:, Fd:]*(1-gate)  (:156) */
MMSA_API int mmsa_gate_mix_fwd(const float* Wm, const float* gate, float* fused, int B, int Fd, void* stream);
MMSA_API int mmsa_gate_mix_bwd(const float* dfused, const float* Wm, const float* gate, float* dWm, float* dgate, int B,
                               int Fd, int accumulate, void* stream);
/* what nn.CrossEntropyLoss checks before it computes (:452-454; "Target N is out of bounds"): raises bit 1 of the
   sticky flag word *err when a label lies outside [0, C) (bit 0 is the embedding gather's: a token id outside the
   table); the host turns the flag into the reference's IndexError */
MMSA_API int mmsa_check_labels(const long long* labels, int B, int C, int* err, void* stream);
/* probabilities, mean cross-entropy and d(loss)/d(logits) = (p - onehot)/B   (:449-454) */
MMSA_API int mmsa_softmax_ce(const float* logits, const long long* labels, float* probs, float* loss, float* dlogits,
                             int B, int C, void* stream);
/* BatchNorm1d + ReLU (+dropout) of the audio-only classifier (:369-371,373-375) */
MMSA_API int mmsa_bn1d_relu_fwd(const float* x, const float* gamma, const float* beta, float* running_mean,
                                float* running_var, long long* nbt, float momentum, float eps, int training, float* y,
                                float* coeff, int B, int N, float drop_p, unsigned long long seed, const float* step,
                                void* stream);
MMSA_API int mmsa_bn1d_relu_bwd(const float* dy, const float* x, const float* coeff, float* dx, float* dgamma,
                                float* dbeta, int B, int N, int training, float drop_p, unsigned long long seed,
                                const float* step, void* stream);
MMSA_API int mmsa_dropout(const float* x, float* y, long n, float drop_p, unsigned long long seed, const float* step,
                          void* stream);

/* ---- loss regulariser, clipping, optimiser: multimodal_model.py:456-462; multimodal_train.py:194-195,481-485. */
MMSA_API int mmsa_param_norms(const float* p, const long long* chunk_off, const int* chunk_len, const int* chunk_tensor,
                              int nchunks, const int* tensor_first_chunk, int ntensors, double* chunk_ss, float* tnorm,
                              float* sum_out, int chunk_sums_ready, void* stream);
/* g += (lambda * *scale) * p / ||p|| per tensor; scale: one float in device memory, or NULL for 1 (the gradient
   arriving at the loss, which autograd hands over as a device scalar: no host read of it is needed) */
MMSA_API int mmsa_l2_grad_add(float* g, const float* p, const float* tnorm, float lambda, const float* scale,
                              const long long* chunk_off, const int* chunk_len, const int* chunk_tensor, int nchunks,
                              void* stream);
/* state[4] = global norm of g (+ lambda * p / ||p|| per tensor when p and tnorm are given) */
MMSA_API int mmsa_grad_norm(const float* g, const float* p, const float* tnorm, float lambda,
                            const long long* chunk_off, const int* chunk_len, const int* chunk_tensor, int nchunks,
                            const int* tensor_first_chunk, int ntensors, double* chunk_ss, float* state, void* stream);
/* mmsa_grad_norm that also advances the step scalars in its second launch (step count + 1, bias corrections, clip
 * coefficient): follow it with mmsa_adamw_step(..., scalars_ready = 1).  multimodal_train.py:194-195.
 * sparse_tensor >= 0: that tensor (chunks sparse_chunk0 .. + sparse_nchunks of the chunk table) is a [rows][sparse_width]
 * table -- the token embedding, multimodal_model.py:149 -- whose data gradient is zero outside the rows sparse_ids[q] of
 * the sparse_n positions q with sparse_first[q] > 0 (the seg_cnt array mmsa_token_groups leaves); its share of the norm
 * is then read from those rows only (the regulariser's share lambda^2 of it needs no pass at all).  -1: dense. */
MMSA_API int mmsa_grad_norm_finish(const float* g, const float* p, const float* tnorm, float lambda,
                                   const long long* chunk_off, const int* chunk_len, const int* chunk_tensor, int nchunks,
                                   const int* tensor_first_chunk, int ntensors, double* chunk_ss, float* state,
                                   double beta1, double beta2, float max_norm, int do_clip, int sparse_tensor,
                                   int sparse_chunk0, int sparse_nchunks, int sparse_width, long sparse_n,
                                   const long long* sparse_ids, const int* sparse_first, void* stream);
/* state (8 floats): 0 step, 1 1-beta1^t, 2 sqrt(1-beta2^t), 3 clip coefficient, 4 total grad norm, 5 sum of norms.
 * The betas are doubles: torch.optim.AdamW forms 1 - beta and the bias corrections in double before anything is rounded
 * to fp32 (1 - 0.999 -> 0.001f, where 1.f - 0.999f = 0.00099998713f). */
MMSA_API int mmsa_adamw_step(float* p, const float* g, float* m, float* v, const long long* chunk_off,
                             const int* chunk_len, const int* chunk_tensor, int nchunks, const float* tnorm,
                             float lambda, float* state, float lr, double beta1, double beta2, float eps,
                             float weight_decay, float max_norm, int do_clip, double* chunk_ss_new, int scalars_ready,
                             void* stream);
MMSA_API int mmsa_scale_by_state(float* g, long n, const float* state, int idx, void* stream);
MMSA_API int mmsa_scalar_axpy(float* dst, const float* a, const float* b, float alpha, void* stream);
/* Up to four device-to-device copies in one launch: a step's inputs into the fixed buffers that a replayed launch
 * sequence reads (the reference moves each batch tensor with its own .to(DEVICE), multimodal_train.py:158-173).
 * srcs / dsts / bytes are host arrays of n <= 4 entries; sizes are multiples of 4 bytes. */
/* Diagnostic (no counterpart in the reference): buf[slot] <- the GPU's 100 MHz real-time counter when `stream` reaches
 * this point; see tools/step_stamps.py. */
MMSA_API int mmsa_stamp(unsigned long long* buf, int slot, void* stream);
/* Diagnostic (no counterpart in the reference): keeps `stream` busy for about `microseconds` (<= 10 000), so that what
 * the host enqueues behind it is complete before the GPU gets there (bench.py's event pairs around single launches). */
MMSA_API int mmsa_spin(int microseconds, void* stream);
MMSA_API int mmsa_copy_multi(const void* const* srcs, void* const* dsts, const long* bytes, int n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMSA_H */
