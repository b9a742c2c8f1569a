// stft.hip -- STFT front-end on the GPU: framing + periodic Hann + real DFT as ONE fp32-MFMA GEMM against a
// windowed cos/sin basis, then |X|^2 -> log.
//
// Replaces stft_pytorch (packages/processing/stft.py:102-151: optional one-hop zero pad at the end, Hann(n_fft),
// torch.stft(n_fft, hop, center=False)) and the callers' power / log (scripts/evaluate_audio_net.py:141-148,
// packages/data_handling.py:454-457).  A 1024-point real DFT of a frame is a [1 x 1024] x [1024 x 1026] product;
// batched over all frames of all utterances it is a GEMM with M = B*T, K = n_fft, N = 2*(n_fft/2+1) -- MFMA work
// (2.1 GFLOP per 1000 frames) instead of a butterfly network, and the framing gather (hop 256 -> every sample is
// read by 4 frames) never materialises: the A functor reads wave[b][t*hop + k] directly.
#include "igemm.h"

namespace {

// A[m = (b,t)][k] = wave[b][t*hop + k]  (zero beyond the utterance: the reference's end padding)
struct FrameRows {
  static constexpr bool KCONTIG = true;
  static constexpr int VEC = 1;
  typedef igemm::NoCtx Ctx;
  const float* p;
  long L;
  int X, K, T, hop;
  __device__ __forceinline__ Ctx prep(int) const { return Ctx(); }
  __device__ __forceinline__ void load(const Ctx&, int x, int k0, int kin, float* v) const {
    const int k = k0 + kin;
    float t = 0.f;
    if (x < X && k < K) {
      const int b = x / T, fr = x - b * T;
      const long idx = (long)fr * hop + k;
      if (idx < L) t = p[(long)b * L + idx];
    }
    v[0] = t;
  }
};

// basis[k][2f] = hann[k] cos(2 pi f k / N), basis[k][2f+1] = -hann[k] sin(2 pi f k / N); columns >= 2F are zero
__global__ void dft_basis(float* __restrict__ W, int N, int F, int ld) {
  const long n = (long)N * ld;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i / ld), c = (int)(i % ld);
    float v = 0.f;
    if (c < 2 * F) {
      const int f = c >> 1;
      const double win = 0.5 - 0.5 * cospi(2.0 * (double)k / (double)N);          // periodic Hann
      const long fk = ((long)f * k) % N;                                          // exact phase reduction
      const double ang = 2.0 * (double)fk / (double)N;
      v = (float)((c & 1) ? -win * sinpi(ang) : win * cospi(ang));
    }
    W[i] = v;
  }
}

// out[m][f] = log(re^2 + im^2 + eps)   (or the power itself when take_log == 0)
__global__ void power_log(const float* __restrict__ S, float* __restrict__ out, long M, int F, int ld, float eps, int take_log) {
  const long n = M * F;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long m = i / F;
    const int f = (int)(i - m * F);
    const float2 c = *reinterpret_cast<const float2*>(S + m * ld + 2 * f);
    const float pw = c.x * c.x + c.y * c.y;
    out[i] = take_log ? logf(pw + eps) : pw;
  }
}
// the evaluate scripts' whole feature chain behind the DFT: log(re^2 + im^2 + eps) then the train-set standardisation
// (x - mean[f]) / (std[f] + eps)  (scripts/evaluate_audio_net.py:141-163) in the same pass
__global__ void power_log_standardize(const float* __restrict__ S, const float* __restrict__ mean, const float* __restrict__ stdv,
                                      float* __restrict__ out, long M, int F, int ld, float eps, float norm_eps) {
  const long n = M * F;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long m = i / F;
    const int f = (int)(i - m * F);
    const float2 c = *reinterpret_cast<const float2*>(S + m * ld + 2 * f);
    const float v = logf(c.x * c.x + c.y * c.y + eps);
    out[i] = (v - mean[f]) / (stdv[f] + norm_eps);
  }
}
// legacy torch.stft real view of ONE utterance: out[f][t][{re,im}]
__global__ void to_legacy_view(const float* __restrict__ S, float* __restrict__ out, int T, int F, int ld) {
  const long n = (long)T * F * 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int ri = (int)(i & 1);
    const long r = i >> 1;
    const int t = (int)(r % T), f = (int)(r / T);
    out[i] = S[(long)t * ld + 2 * f + ri];
  }
}

static inline int grid1(long n) { long b = (n + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }
static inline int ld_of(const avvad_stft_desc* d) { return (2 * (d->n_fft / 2 + 1) + 3) / 4 * 4; }
static bool ok_desc(const avvad_stft_desc* d) {
  return d && d->B > 0 && d->L > 0 && d->n_fft >= 32 && d->n_fft % 32 == 0 && d->hop > 0 && d->T > 0 &&
         (long)(d->T - 1) * d->hop + d->n_fft <= d->L + d->hop;   // at most the reference's one-hop end pad
}

}  // namespace

extern "C" size_t avvad_stft_workspace(const avvad_stft_desc* d) {
  if (!ok_desc(d)) return 0;
  const size_t ld = ld_of(d);
  return (align_up((size_t)d->n_fft * ld, 64) + align_up((size_t)d->B * d->T * ld, 64) + igemm::SLAB_FLOATS) * sizeof(float);
}

static int stft_impl(const float* wave, float* out, const avvad_stft_desc* d, int mode, const float* mean, const float* stdv,
                     float norm_eps, void* wsv, size_t ws_bytes, avvad_stream_t sv);

// mode 0: out [B][T][F] = log(|X|^2 + eps);  mode 1: out [B][T][F] = |X|^2;
// mode 2 (B == 1): out [F][T][2] = legacy torch.stft real view (re, im)
extern "C" int avvad_stft(const float* wave, float* out, const avvad_stft_desc* d, int mode, void* wsv, size_t ws_bytes,
                          avvad_stream_t sv) {
  return stft_impl(wave, out, d, mode, nullptr, nullptr, 0.f, wsv, ws_bytes, sv);
}

// log-power features standardised with the train-set statistics in the DFT's epilogue pass
extern "C" int avvad_stft_features(const float* wave, const float* mean, const float* stdv, float* out,
                                   const avvad_stft_desc* d, float norm_eps, void* wsv, size_t ws_bytes, avvad_stream_t sv) {
  if (!mean || !stdv) return AVVAD_EINVAL;
  return stft_impl(wave, out, d, 0, mean, stdv, norm_eps, wsv, ws_bytes, sv);
}

static int stft_impl(const float* wave, float* out, const avvad_stft_desc* d, int mode, const float* mean, const float* stdv,
                     float norm_eps, void* wsv, size_t ws_bytes, avvad_stream_t sv) {
  AVVAD_ENTER();
  if (!wave || !out || !wsv || !ok_desc(d) || mode < 0 || mode > 2 || (mode == 2 && d->B != 1)) return AVVAD_EINVAL;
  if (ws_bytes < avvad_stft_workspace(d)) return AVVAD_EWORKSPACE;
  hipStream_t s = (hipStream_t)sv;
  const int F = d->n_fft / 2 + 1, ld = ld_of(d);
  float* W = (float*)wsv;
  float* S = W + align_up((size_t)d->n_fft * ld, 64);
  const int M = d->B * d->T;
  hipLaunchKernelGGL(dft_basis, dim3(grid1((long)d->n_fft * ld)), dim3(256), 0, s, W, d->n_fft, F, ld);
  FrameRows a{wave, d->L, M, d->n_fft, d->T, d->hop};
  igemm::ColPlain<4> b{W, ld, ld, d->n_fft, 0};
  igemm::EpiStore e{S, ld, nullptr, 0};
  int rc = igemm::launch<128, 128>(a, b, e, M, ld, d->n_fft, 1, s, S + align_up((size_t)M * ld, 64), /*allow_bf16=*/false);
  if (rc) return rc;
  if (mode == 2) hipLaunchKernelGGL(to_legacy_view, dim3(grid1((long)d->T * F * 2)), dim3(256), 0, s, S, out, d->T, F, ld);
  else if (mean)
    hipLaunchKernelGGL(power_log_standardize, dim3(grid1((long)M * F)), dim3(256), 0, s, S, mean, stdv, out, (long)M, F, ld, d->eps,
                       norm_eps);
  else hipLaunchKernelGGL(power_log, dim3(grid1((long)M * F)), dim3(256), 0, s, S, out, (long)M, F, ld, d->eps, mode == 0);
  AVVAD_LAUNCH_CHECK();
  return AVVAD_OK;
}
