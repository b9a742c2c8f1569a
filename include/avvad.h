/* avvad.h -- C ABI of libavvad_hip.so: the MI355X (gfx950) kernels behind the
 * per-frame classification hot path of sp-uhh/audio-visual-vad.
 *
 * The reference has no FFI / operator registry (it is pure Python on torch.nn,
 * SURVEY.md 8b); every entry point below replaces a stock torch / torchvision
 * call site of the reference, cited as file:line relative to the reference root.
 * The Python host (audio-visual-vad_amd/avvad) binds these with ctypes; a
 * maintainer of the reference would add the same ctypes stub.
 *
 * Conventions (all entry points):
 *   - plain C types only; every pointer is a DEVICE pointer unless the
 *     parameter name ends in _h (host);
 *   - the caller allocates every buffer, including the workspace
 *     (size from the matching *_workspace() query, bytes);
 *   - asynchronous on the given hipStream_t, never synchronises, never
 *     allocates, keeps no per-call state -> re-entrant across streams and
 *     devices, capturable into a hipGraph.  The only process-wide state is the
 *     table of schedule options below (avvad_set_option): it is filled ONCE
 *     from the AVVAD_* environment variables on first use and never re-read;
 *   - returns AVVAD_OK or a negative AVVAD_E* code, never throws;
 *   - fp32 storage and fp32 arithmetic (fp32-input MFMA == ordered fmaf chain).
 */
#ifndef AVVAD_H
#define AVVAD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* avvad_stream_t; /* hipStream_t */

#define AVVAD_OK 0
#define AVVAD_EINVAL (-1)     /* bad descriptor / unsupported shape */
#define AVVAD_EWORKSPACE (-2) /* workspace too small */
#define AVVAD_ELAUNCH (-3)    /* hipGetLastError() != hipSuccess after a launch */

/* library / build identification ("gfx950", ABI version).  AVVAD_ABI_VERSION is what THIS header describes; a binding
 * must refuse a library whose avvad_abi_version() differs (signatures changed incompatibly between versions:
 * 2 = (ws, ws_bytes) in front of the stream of avvad_gemm_f32 / avvad_conv2d_*, avvad_wavenet_desc.shared_device). */
#define AVVAD_ABI_VERSION 3   /* 3 = + avvad_conv2d_*_bf16 */
const char* avvad_version(void);
int avvad_abi_version(void);

/* Schedule options (tuning / debugging; production leaves them alone).  Names:
 *   "no_streamk" (1 = whole-tile GEMM schedule), "igemm_variant", "kmajor", "no_tall", "no_stem_kernel",
 *   "no_fixup1" (1 = always the four-wave fix-up kernel), "no_buf" (1 = convolution gathers with flat addressing + validity selects, the form operands >= 2 GiB use),
 *   "lstm_no_fused_step", "lstm_no_persistent", "wn_no_fused_tail", "wn_no_fused_wgrad", "wn_bwd_t",
 *   "wn_flat" (encoder block forward: 0 by plane length, 1 flat dword kernel, 2 buffer dword kernel with resident
 *   weights, 3 dwordx4 kernel, 4 high-occupancy kernel), "wn_dx" / "wn_bwd_t" (0 by the descriptor's shared_device hint, other values force
 *   a form), "wn_grid" (workgroup cap of the encoder block kernels),
 *   "bf16" (BASELINE config 5's mixed precision, never the default.  1: the trunk runs its bf16 DATA PATH -- activations
 *   between convolutions, the gradients that feed convolutions and the packed weights are stored as bf16, the convolutions
 *   run on the bf16 engine -- and the dense GEMMs of the heads round their fp32 operands to bf16 on their way into LDS;
 *   2: round 2's form, fp32 storage everywhere and operands rounded while staging.  BatchNorm statistics, LSTM cell, loss,
 *   Adam and every accumulation stay fp32 in both),
 *   "max_cus" (cap on the CUs a persistent grid occupies, so that RCCL's kernels find free CUs during data-parallel
 *   training), "bwd_max_cus" (the same cap applied only while a backward entry point runs: the gradient all-reduce overlaps
 *   the backward pass, the forward keeps the whole chip), "no_cls" (1 = 3x3 convolutions multiply their zero padding like
 *   everything else instead of running position-major), "cls_cap", "no_fused_stats" (1 = BatchNorm statistics by separate
 *   column-reduction passes instead of the producing kernels' epilogues), "no_conv64" (1 = the 64 -> 64 channel 3x3
 *   convolutions on the GEMM engine instead of their weights-stationary / output-stationary kernels), "no_s2_cls" (1 = a
 *   stride-2 data gradient as four accumulating parity-class launches instead of one position-class product).
 * Initial values come from AVVAD_<NAME> in the environment, read once.  Returns AVVAD_EINVAL for an unknown name. */
int avvad_set_option(const char* name, int value);
int avvad_get_option(const char* name);

/* ------------------------------------------------------------------------
 * Dense GEMM on fp32 MFMA:  C[M,N] (+)= op(A) . op(B) (+ bias[N])
 *   transA=0: A is [M,K] row-major (lda);  transA=1: A is stored [K,M] (lda)
 *   transB=0: B is [K,N] row-major (ldb);  transB=1: B is stored [N,K] (ldb)
 *   accumulate: C += result.
 * Replaces: nn.LSTM input / recurrent projections and nn.Linear
 *   (packages/models/Audio_Net.py:30-35,51-59, Video_Net.py:45-51,102-116,
 *    AV_Net.py:53-58,128-140) and their autograd backward.
 * ---------------------------------------------------------------------- */
typedef struct {
  int M, N, K;
  int lda, ldb, ldc;
  int transA, transB;
  int accumulate;
  int split_k; /* >= 1: hint that K is long and the tiles few (needs accumulate) */
  int relu_a;  /* apply max(.,0) to A elements on load */
  int relu_b;
} avvad_gemm_desc;
/* Scratch of the GEMM engine, bytes (a constant: one tile per persistent worker, or -- the larger -- one [576][64] partial
 * weight gradient per CU for the 64-channel convolutions' own kernel).  Tiles whose K range is cut between
 * workers (the engine's stream-K round) leave their partial sums there and a fix-up kernel adds them in a fixed
 * order: results are bit-reproducible run to run, there are no float atomics.  Every entry point that runs a single
 * GEMM / convolution takes (ws, ws_bytes); ws == NULL (or too small) selects whole-tile scheduling -- same results up
 * to summation order, slower where the tile count quantises badly against the 256 CUs. */
size_t avvad_engine_workspace(void);
int avvad_gemm_f32(const float* A, const float* B, const float* bias, float* C,
                   const avvad_gemm_desc* d, void* ws, size_t ws_bytes, avvad_stream_t s);

/* ------------------------------------------------------------------------
 * WaveNet-style encoder (valid dilated Conv1d stack)
 * Replaces: wavenet_autoencoder._encode, packages/models/wavenet_autoencoder.py:74-93
 * Layout: activations [B][C][L] (torch NCL, time contiguous); weights exactly as
 * in the state_dict: Conv1d weight [Cout][Cin][fw], bias [Cout].
 * ---------------------------------------------------------------------- */
typedef struct {
  int B;   /* sequences                                                */
  int L;   /* input samples per sequence                               */
  int qc;  /* quantization_channel (input channels)                    */
  int R;   /* en_residual_channel                                      */
  int D;   /* en_dilation_channel                                      */
  int Bn;  /* en_bottleneck_width                                      */
  int fw;  /* filter_width                                             */
  int P;   /* en_pool_kernel_size (used as pool OUTPUT size, :91)      */
  int n_layers;
  const int* dilations_h; /* host array [n_layers]                     */
  int use_bias;
  int save_for_backward; /* forward keeps every s_i in the workspace (z_i is rebuilt by the backward) */
  int shared_device;     /* hint: another stream's kernels run beside this call (the AV model's trunk on the main stream):
                            prefer the kernel forms that interfere least with them.  0 = the device is ours        */
} avvad_wavenet_desc;

/* parameter pointers, host arrays of device pointers */
typedef struct {
  const float* causal_w;
  const float* causal_b;
  const float* const* dil_w_h; /* [n_layers] */
  const float* const* dil_b_h;
  const float* const* dense_w_h;
  const float* const* dense_b_h;
  const float* bott_w;
  const float* bott_b;
} avvad_wavenet_params;

typedef struct {
  float* causal_w;
  float* causal_b;
  float* const* dil_w_h;
  float* const* dil_b_h;
  float* const* dense_w_h;
  float* const* dense_b_h;
  float* bott_w;
  float* bott_b;
} avvad_wavenet_grads;

size_t avvad_wavenet_workspace(const avvad_wavenet_desc* d);
/* wave [B][qc][L] -> out [B][Bn][P] */
int avvad_wavenet_fwd(const float* wave, const avvad_wavenet_params* p, float* out,
                      const avvad_wavenet_desc* d, void* ws, size_t ws_bytes, avvad_stream_t s);
/* needs the workspace of a forward run with save_for_backward=1.
 * grads are ACCUMULATED (+=) into g; dwave may be NULL. */
int avvad_wavenet_bwd(const float* wave, const avvad_wavenet_params* p, const float* dout,
                      const avvad_wavenet_grads* g, float* dwave, const avvad_wavenet_desc* d,
                      void* ws, size_t ws_bytes, avvad_stream_t s);

/* One residual block of the R = D = 32, filter_width 2 encoder on its own (wavenet_autoencoder.py:80-86):
 *   s_out[b][r][t] = b_dense[r] + sum_d W_dense[r][d] relu(b_dil[d] + sum_{c,k} W_dil[d][c][k] relu(s_in[b][c][t + k*dil]))
 *                    + s_in[b][r][t + dil]          s_in [B][32][Lin] -> s_out [B][32][Lin - dil]
 * The layer-at-a-time kernel of the large dilations; bench.py times it per launch for its HBM roofline entry.
 * Biases may be NULL. */
int avvad_wavenet_block_fwd(const float* s_in, const float* w_dil, const float* b_dil, const float* w_dense,
                            const float* b_dense, float* s_out, int B, int Lin, int dil, avvad_stream_t s);

/* ------------------------------------------------------------------------
 * ResNet-18 trunk over gray lip crops
 * Replaces: self.features(video).squeeze() with the 3x channel repeat in front,
 *   packages/models/Video_Net.py:60-81, packages/models/AV_Net.py:78-94
 *   (torchvision.models.resnet18 children [:-1]).
 * Layout: input frames [N][H][W] (1 channel; the reference's 3 identical
 * channels are folded into conv1's weights), activations NHWC inside, output
 * [N][512].  Parameters are passed in torchvision state_dict layout (OIHW).
 * ---------------------------------------------------------------------- */
#define AVVAD_TRUNK_NCONV 20 /* conv1 + 16 block convs + 3 downsample convs */

typedef struct {
  int N, H, W;
  int training;   /* batch statistics + running-stat update */
  float momentum; /* 0.1 */
  float eps;      /* 1e-5 */
  int save_for_backward;
} avvad_trunk_desc;

/* conv index order: 0 = conv1; then per stage s (0..3), per block b (0..1):
 * conv1, conv2, [downsample if s>0 and b==0].  The i-th BatchNorm follows the
 * i-th conv.  (torchvision order: features.0/1, features.{4..7}.{0,1}.{conv1,bn1,conv2,bn2,downsample}) */
typedef struct {
  const float* conv_w[AVVAD_TRUNK_NCONV]; /* OIHW */
  const float* bn_w[AVVAD_TRUNK_NCONV];
  const float* bn_b[AVVAD_TRUNK_NCONV];
  float* bn_rm[AVVAD_TRUNK_NCONV]; /* running_mean (updated when training) */
  float* bn_rv[AVVAD_TRUNK_NCONV]; /* running_var                          */
} avvad_trunk_params;

typedef struct {
  float* conv_w[AVVAD_TRUNK_NCONV]; /* OIHW, accumulated (+=) */
  float* bn_w[AVVAD_TRUNK_NCONV];
  float* bn_b[AVVAD_TRUNK_NCONV];
} avvad_trunk_grads;

/* Single convolution on the implicit-GEMM engine (the trunk's building block; also what bench.py times
 * per launch for the roofline).  x [N][H][W][C] NHWC, y [N][Ho][Wo][Co]; square kernel KS, stride 1|2.
 * Weights are packed once from OIHW: wf [(kh,kw,c)][co] (forward) and wd [(kh,kw,co)][c] (dgrad; may be
 * NULL).  C must be 1 (stem, forward/wgrad only) or a multiple of 32; Co a multiple of 4.
 * wgrad writes the packed layout [(kh,kw,c)][co] (overwritten).  ws: avvad_engine_workspace() bytes.  Replaces nn.Conv2d inside
 * torchvision's resnet18 (packages/models/Video_Net.py:35-37). */
typedef struct {
  int N, H, W, C, Co, KS, stride, pad;
} avvad_conv_desc;
int avvad_conv2d_pack_weights(const float* w_oihw, float* wf, float* wd, const avvad_conv_desc* d,
                              avvad_stream_t s);
int avvad_conv2d_fwd(const float* x, const float* wf, float* y, const avvad_conv_desc* d, void* ws,
                     size_t ws_bytes, avvad_stream_t s);
int avvad_conv2d_dgrad(const float* dy, const float* wd, float* dx, const avvad_conv_desc* d,
                       int accumulate, void* ws, size_t ws_bytes, avvad_stream_t s);
int avvad_conv2d_wgrad(const float* x, const float* dy, float* dw_packed, const avvad_conv_desc* d, void* ws,
                       size_t ws_bytes, avvad_stream_t s);

/* The same convolutions on the bf16 data path (BASELINE configs[4]; what the trunk runs when option "bf16" is 1): operands
 * are bf16 IN MEMORY -- x16 / dy16 NHWC bf16, weights in the K-contiguous bf16 packs
 *   wf16[co][(cc * T + tap) * 64 + r] = w[co][cc * 64 + r][tap]     (forward; T = KS * KS taps, 64-channel chunks cc)
 *   wd16[c][(cc * T + tap) * 64 + r]  = w[cc * 64 + r][c][tap]      (data gradient; may be NULL in the pack call)
 * -- multiplied by v_mfma_f32_32x32x16_bf16 with fp32 accumulation; results (y, dx, the packed weight gradient
 * [(kh,kw,c)][co]) are fp32.  C and Co must be multiples of 64.  Same replaced call site as above. */
int avvad_conv2d_pack_weights_bf16(const float* w_oihw, void* wf16, void* wd16, const avvad_conv_desc* d, avvad_stream_t s);
int avvad_conv2d_fwd_bf16(const void* x16, const void* wf16, float* y, const avvad_conv_desc* d, void* ws, size_t ws_bytes,
                          avvad_stream_t s);
int avvad_conv2d_dgrad_bf16(const void* dy16, const void* wd16, float* dx, const avvad_conv_desc* d, int accumulate,
                            void* ws, size_t ws_bytes, avvad_stream_t s);
int avvad_conv2d_wgrad_bf16(const void* x16, const void* dy16, float* dw_packed, const avvad_conv_desc* d, void* ws,
                            size_t ws_bytes, avvad_stream_t s);

size_t avvad_trunk_workspace(const avvad_trunk_desc* d);
/* Test support: offset (floats), channels and spatial size of the post-ReLU activation `index` that a forward run with
 * save_for_backward keeps in its workspace, NHWC.  index 0: pooled stem output; 1 + 2k: block k's first activation
 * (bn1 + ReLU); 2 + 2k: block k's output (k = 0..7).  The parity tests compare sign patterns with the oracle's. */
int avvad_trunk_activation(const avvad_trunk_desc* d, int index, size_t* offset_floats, int* C, int* H, int* W);
int avvad_trunk_fwd(const float* frames, const avvad_trunk_params* p, float* feat /* [N][512] */,
                    const avvad_trunk_desc* d, void* ws, size_t ws_bytes, avvad_stream_t s);
int avvad_trunk_bwd(const float* frames, const avvad_trunk_params* p, const float* dfeat,
                    const avvad_trunk_grads* g, const avvad_trunk_desc* d, void* ws, size_t ws_bytes,
                    avvad_stream_t s);

/* ------------------------------------------------------------------------
 * Packed-sequence multi-layer LSTM (unidirectional) -- batch_first padded input,
 * padded output steps are zero, (h,c) stop at each sequence's length.
 * Replaces: pack_padded_sequence -> nn.LSTM -> pad_packed_sequence(total_length)
 *   packages/models/Audio_Net.py:50-56, Video_Net.py:102-113, AV_Net.py:127-137
 * x [B][T][In], y [B][T][H]; weights in state_dict layout: w_ih [4H][In],
 * w_hh [4H][H], b_ih/b_hh [4H], gate order i,f,g,o.
 * ---------------------------------------------------------------------- */
typedef struct {
  int B, T, In, H;
  const int* lengths; /* device int32 [B] */
  int save_for_backward;
} avvad_lstm_desc;
size_t avvad_lstm_workspace(const avvad_lstm_desc* d);
int avvad_lstm_layer_fwd(const float* x, const float* w_ih, const float* w_hh, const float* b_ih,
                         const float* b_hh, float* y, const avvad_lstm_desc* d, void* ws,
                         size_t ws_bytes, avvad_stream_t s);
/* dx may be NULL; parameter grads are accumulated (+=). dy is [B][T][H]. */
int avvad_lstm_layer_bwd(const float* x, const float* w_ih, const float* w_hh, const float* y,
                         const float* dy, float* dx, float* dw_ih, float* dw_hh, float* db_ih,
                         float* db_hh, const avvad_lstm_desc* d, void* ws, size_t ws_bytes,
                         avvad_stream_t s);

/* ------------------------------------------------------------------------
 * Multimodal compact bilinear fusion + signed sqrt + whole-tensor L2 normalisation + BatchNorm1d
 * Replaces: the use_mcb branch of DeepVAD_AV.forward, packages/models/AV_Net.py:109-121, i.e.
 *   CompactBilinearPooling (packages/models/compact_bilinear_pooling.py:7-27,140-220: count sketches
 *   psi(x,h,s)[h_i] += s_i x_i, circular convolution of the two sketches) -> sign(y)sqrt(|y|+eps) ->
 *   y / ||y||_2 (detached norm of the whole tensor) -> BatchNorm1d(D, eps) over all rows.
 * audio [rows][A], video [rows][V], out [rows][D]; rows = B*T; h1/h2 int64 bucket per input channel
 * (values in [0, D)), s1/s2 = +-1.  D % 4 == 0, D <= 1024.  bwd accumulates (+=) dbn_w / dbn_b and
 * overwrites daudio / dvideo (either may be NULL).
 * ---------------------------------------------------------------------- */
typedef struct {
  int rows, A, V, D;
  float eps;      /* used for the signed sqrt AND as the BatchNorm eps (AV_Net.py:49,114) */
  int training;
  float momentum;
  int save_for_backward;
} avvad_mcb_desc;
size_t avvad_mcb_workspace(const avvad_mcb_desc* d);
int avvad_mcb_fusion_fwd(const float* audio, const float* video, const int64_t* h1, const float* s1,
                         const int64_t* h2, const float* s2, const float* bn_w, const float* bn_b,
                         float* bn_rm, float* bn_rv, float* out, const avvad_mcb_desc* d, void* ws,
                         size_t ws_bytes, avvad_stream_t s);
int avvad_mcb_fusion_bwd(const float* audio, const float* video, const int64_t* h1, const float* s1,
                         const int64_t* h2, const float* s2, const float* bn_w, const float* dout,
                         float* daudio, float* dvideo, float* dbn_w, float* dbn_b,
                         const avvad_mcb_desc* d, void* ws, size_t ws_bytes, avvad_stream_t s);

/* The bare modules of packages/models/compact_bilinear_pooling.py, for callers that use them outside DeepVAD_AV:
 *   CountSketch.forward (:59-114 -> CountSketchFn_forward :7-27):  out[row][h[i]] += s[i] * x[row][i]
 *   CountSketchFn_backward (:30-38):                               dx[row][i] = s[i] * dout[row][h[i]]
 *   CompactBilinearPooling.forward (:222-263 -> CompactBilinearPoolingFn.forward :140-173): the raw vector
 *     y = irfft(rfft(psi(a,h1,s1)) * rfft(psi(v,h2,s2))) = circular convolution of the two sketches, [rows][D]
 *   CompactBilinearPoolingFn.backward (:175-220): da, dv (either may be NULL; overwritten).
 * D <= 2048. */
int avvad_count_sketch_fwd(const float* x, const int64_t* h, const float* s, float* out, int rows, int In, int D,
                           avvad_stream_t st);
int avvad_count_sketch_bwd(const float* dout, const int64_t* h, const float* s, float* dx, int rows, int In, int D,
                           avvad_stream_t st);
int avvad_mcb_fwd(const float* a, const float* v, const int64_t* h1, const float* s1, const int64_t* h2,
                  const float* s2, float* y, int rows, int A, int V, int D, avvad_stream_t st);
int avvad_mcb_bwd(const float* a, const float* v, const int64_t* h1, const float* s1, const int64_t* h2,
                  const float* s2, const float* dy, float* da, float* dv, int rows, int A, int V, int D,
                  avvad_stream_t st);

/* ------------------------------------------------------------------------
 * STFT log-power front-end: framing + periodic Hann + real DFT (one MFMA GEMM) + |X|^2 (+ log)
 * Replaces: stft_pytorch packages/processing/stft.py:102-151 (center=False; the optional one-hop zero pad at
 *   the end is implied by T: frames may run at most one hop past L and read zeros there) and the callers'
 *   power / log, scripts/evaluate_audio_net.py:141-148, packages/data_handling.py:454-457.
 * wave [B][L].  mode 0: out [B][T][F] = log(|X|^2 + eps); mode 1: out = |X|^2; mode 2 (B == 1): out [F][T][2] =
 * (re, im), the legacy torch.stft real view the reference's callers index.  F = n_fft/2 + 1, n_fft % 32 == 0.
 * ---------------------------------------------------------------------- */
typedef struct {
  int B;
  long L;
  int n_fft, hop, T;
  float eps;
} avvad_stft_desc;
size_t avvad_stft_workspace(const avvad_stft_desc* d);
int avvad_stft(const float* wave, float* out, const avvad_stft_desc* d, int mode, void* ws, size_t ws_bytes,
               avvad_stream_t s);

/* The evaluate scripts' feature chain in one call (scripts/evaluate_audio_net.py:131-163): STFT -> |X|^2 ->
 * log(. + d->eps) -> (x - mean[f]) / (std[f] + norm_eps), the standardisation folded into the DFT's epilogue pass.
 * mean / std: [F] train-set statistics.  out [B][T][F]. */
int avvad_stft_features(const float* wave, const float* mean, const float* std_, float* out,
                        const avvad_stft_desc* d, float norm_eps, void* ws, size_t ws_bytes, avvad_stream_t s);
/* out[b][:] = x[b][:] / max|x[b][:]|   (peak normalisation, scripts/evaluate_audio_net.py:125-127); out may alias x */
int avvad_peak_normalize(const float* x, float* out, int B, long L, avvad_stream_t s);
/* out[r][f] = (x[r][f] - mean[f]) / (std[f] + eps)  -- input standardisation of the train / evaluate loops
 * (scripts/train_AV_net.py:286-291, evaluate_audio_net.py:158-163).  nstat == F: per-bin statistics (audio,
 * 513 x 1 in the reference); nstat == 1: one scalar pair (video, 1 x 1).  out may alias x. */
int avvad_standardize(const float* x, const float* mean, const float* std_, float* out, size_t rows, int F,
                      int nstat, float eps, avvad_stream_t s);

/* ------------------------------------------------------------------------
 * Masked BCE-with-eps loss, summed over sequences
 * Replaces: binary_cross_entropy packages/models/utils.py:108-113 and its caller
 *   loop scripts/train_AV_net.py:298-301  (per-sequence mean over valid frames
 *   and y_dim, summed over the batch).
 * logits/targets [B][T][Y]; loss: one float (overwritten); dlogits [B][T][Y]
 * (d loss / d logits, zero on padded steps), may be NULL.
 * ---------------------------------------------------------------------- */
int avvad_bce_masked(const float* logits, const float* targets, const int* lengths, float* loss,
                     float* dlogits, int B, int T, int Y, float eps, avvad_stream_t s);

/* Two-output-unit BCE on probabilities: binary_cross_entropy_2classes packages/models/utils.py:115-116
 * (imported by scripts/train_video_net.py:18):
 *   loss = -mean_rows( sum_y [ x log(r1 + eps) + (1 - x) log(r2 + eps) ] ).
 * r1, r2, x [rows][Y]; loss one float; dr1 / dr2 (d loss / d r, may be NULL) [rows][Y]. */
int avvad_bce_2classes(const float* r1, const float* r2, const float* x, float* loss, float* dr1, float* dr2,
                       long rows, int Y, float eps, avvad_stream_t s);

/* ------------------------------------------------------------------------
 * Fused Adam step over a flat parameter buffer
 * Replaces: torch.optim.Adam(lr, betas=(0.9,0.999)).step()  scripts/train_AV_net.py:238,306
 * (torch semantics: eps added to sqrt(v_hat); no weight decay, no amsgrad).
 * ---------------------------------------------------------------------- */
int avvad_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                    float lr, float beta1, float beta2, float eps, int step, avvad_stream_t s);

/* ------------------------------------------------------------------------
 * Small fused elementwise helpers used by the Python host
 * ---------------------------------------------------------------------- */
/* dst[r][dst_off + c] = src[r][src_off + c], c < ncols (row strides src_ld / dst_ld) -- writes a
 * branch into the concat buffer (torch.cat, AV_Net.py:124) and splits its gradient back */
int avvad_copy_cols(const float* src, float* dst, size_t rows, int ncols, int src_ld, int src_off,
                    int dst_ld, int dst_off, avvad_stream_t s);
/* out[c] += sum_r X[r][c]   (bias gradient of nn.Linear) */
int avvad_colsum_acc(const float* X, size_t rows, int cols, float* out, avvad_stream_t s);
/* x[i] *= *scalar  (scalar lives on the device: upstream gradient of the loss) */
int avvad_scale_by_device_scalar(float* x, const float* scalar, size_t n, avvad_stream_t s);
/* out[b][t][c] = in[b][c][t]  (encoder output (B,Bn,P) -> (B,P,Bn)) and back */
int avvad_transpose_last2(const float* in, float* out, int B, int C, int T, avvad_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* AVVAD_H */
