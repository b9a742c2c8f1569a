// misc.hip -- loss, optimiser and small layout kernels of the AV-VAD hot path.
#include "common.h"

extern "C" const char* avvad_version(void) { return "avvad-hip 0.1 (gfx950, fp32 MFMA)"; }
extern "C" int avvad_abi_version(void) { return AVVAD_ABI_VERSION; }   // include/avvad.h says what changed when

// ---------------------------------------------------------------- schedule options
#include <stdlib.h>
#include <string.h>
namespace {
struct OptName { const char* name; const char* env; int AvvadTune::*field; };
const OptName kOpts[] = {
    {"igemm_variant", "AVVAD_IGEMM_VARIANT", &AvvadTune::igemm_variant},
    {"no_streamk", "AVVAD_NO_STREAMK", &AvvadTune::no_streamk},
    {"kmajor", "AVVAD_KMAJOR", &AvvadTune::kmajor},
    {"lstm_no_fused_step", "AVVAD_LSTM_NO_FUSED_STEP", &AvvadTune::lstm_no_fused_step},
    {"lstm_no_persistent", "AVVAD_LSTM_NO_PERSISTENT", &AvvadTune::lstm_no_persistent},
    {"no_stem_kernel", "AVVAD_NO_STEM_KERNEL", &AvvadTune::no_stem_kernel},
    {"no_tall", "AVVAD_NO_TALL", &AvvadTune::no_tall},
    {"wn_no_fused_tail", "AVVAD_WN_NO_FUSED_TAIL", &AvvadTune::wn_no_fused_tail},
    {"wn_no_tail_pair", "AVVAD_WN_NO_TAIL_PAIR", &AvvadTune::wn_no_tail_pair},
    {"wn_no_fused_wgrad", "AVVAD_WN_NO_FUSED_WGRAD", &AvvadTune::wn_no_fused_wgrad},
    {"wn_bwd_t", "AVVAD_WN_BWD_T", &AvvadTune::wn_bwd_t},
    {"no_buf", "AVVAD_NO_BUF", &AvvadTune::no_buf},
    {"no_fixup1", "AVVAD_NO_FIXUP1", &AvvadTune::no_fixup1},
    {"no_conv64", "AVVAD_NO_CONV64", &AvvadTune::no_conv64},
    {"no_s2_cls", "AVVAD_NO_S2_CLS", &AvvadTune::no_s2_cls},
    {"wn_flat", "AVVAD_WN_FLAT", &AvvadTune::wn_flat},
    {"wn_grid", "AVVAD_WN_GRID", &AvvadTune::wn_grid},
    {"wn_dx", "AVVAD_WN_DX", &AvvadTune::wn_dx},
    {"bf16", "AVVAD_BF16", &AvvadTune::bf16},
    {"no_fused_stats", "AVVAD_NO_FUSED_STATS", &AvvadTune::no_fused_stats},
    {"no_cls", "AVVAD_NO_CLS", &AvvadTune::no_cls},
    {"cls_cap", "AVVAD_CLS_CAP", &AvvadTune::cls_cap},
    {"stagger", "AVVAD_STAGGER", &AvvadTune::stagger},
    {"max_cus", "AVVAD_MAX_CUS", &AvvadTune::max_cus},
    {"bwd_max_cus", "AVVAD_BWD_MAX_CUS", &AvvadTune::bwd_max_cus},
};
int parse_opt(const char* name, const char* v) {
  if (!strcmp(name, "igemm_variant")) return v[0] == 'd' ? 0 : (v[0] == 's' ? 1 : (v[0] == 'w' ? 2 : atoi(v)));
  if (!strcmp(name, "no_streamk")) return v[0] == 'a' ? 1 : (v[0] >= '0' && v[0] <= '9' && !v[1] ? 10 + (v[0] - '0') : atoi(v));
  if (!strcmp(name, "max_cus") || !strcmp(name, "bwd_max_cus") || !strcmp(name, "wn_flat") || !strcmp(name, "wn_grid") || !strcmp(name, "wn_dx") || !strcmp(name, "wn_bwd_t") || !strcmp(name, "no_fixup1") || !strcmp(name, "cls_cap")) return atoi(v);
  return (v[0] && strcmp(v, "0")) ? 1 : 0;
}
}  // namespace

AvvadTune& avvad_tune() {
  static AvvadTune t = [] {      // the environment is read ONCE, here
    AvvadTune x;
    memset(&x, 0, sizeof(x));
    x.igemm_variant = -1;
    for (const OptName& o : kOpts) {
      const char* v = getenv(o.env);
      if (v && v[0]) x.*(o.field) = parse_opt(o.name, v);
    }
    return x;
  }();
  return t;
}

extern "C" int avvad_set_option(const char* name, int value) {
  if (!name) return AVVAD_EINVAL;
  for (const OptName& o : kOpts)
    if (!strcmp(o.name, name)) { avvad_tune().*(o.field) = value; return AVVAD_OK; }
  return AVVAD_EINVAL;
}

extern "C" int avvad_get_option(const char* name) {
  if (!name) return AVVAD_EINVAL;
  for (const OptName& o : kOpts)
    if (!strcmp(o.name, name)) return avvad_tune().*(o.field);
  return AVVAD_EINVAL;
}

namespace {

// Masked BCE-with-eps, summed over sequences.
//   loss = sum_b  -(1/(len_b*Y)) sum_{t<len_b, y} [ x log(sig(r)+eps) + (1-x) log(1-sig(r)+eps) ]
// reference: binary_cross_entropy packages/models/utils.py:108-113 + caller loop scripts/train_AV_net.py:298-301.
// One workgroup, fixed summation order -> deterministic.
__global__ void __launch_bounds__(1024)
    bce_masked_kernel(const float* __restrict__ logits, const float* __restrict__ targets, const int* __restrict__ lengths,
                      float* __restrict__ loss, float* __restrict__ dlogits, int B, int T, int Y, float eps) {
  __shared__ float sm[1024];
  const int tid = threadIdx.x;
  const long n = (long)B * T * Y;
  float acc = 0.f;
  for (long i = tid; i < n; i += 1024) {
    const int b = (int)(i / ((long)T * Y));
    const int t = (int)((i / Y) % T);
    const int len = lengths[b];
    float g = 0.f;
    if (t < len) {
      const float r = logits[i], x = targets[i];
      const float sg = 1.f / (1.f + expf(-r));
      const float inv = 1.f / ((float)len * (float)Y);
      acc -= (x * logf(sg + eps) + (1.f - x) * logf(1.f - sg + eps)) * inv;
      const float ds = sg * (1.f - sg);
      g = -(x * ds / (sg + eps) - (1.f - x) * ds / (1.f - sg + eps)) * inv;
    }
    if (dlogits) dlogits[i] = g;
  }
  sm[tid] = acc;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o) sm[tid] += sm[tid + o];
    __syncthreads();
  }
  if (tid == 0) loss[0] = sm[0];
}

// Two-output-unit BCE (binary_cross_entropy_2classes, packages/models/utils.py:115-116; imported by
// scripts/train_video_net.py:18):  loss = -mean_rows( sum_y [ x log(r1 + eps) + (1 - x) log(r2 + eps) ] ),
// r1 / r2 are PROBABILITIES.  Value and both gradients in one deterministic workgroup.
__global__ void __launch_bounds__(1024)
    bce_2classes_kernel(const float* __restrict__ r1, const float* __restrict__ r2, const float* __restrict__ x,
                        float* __restrict__ loss, float* __restrict__ dr1, float* __restrict__ dr2, long rows, int Y, float eps) {
  __shared__ float sm[1024];
  const int tid = threadIdx.x;
  const long n = rows * Y;
  const float inv = 1.f / (float)rows;
  float acc = 0.f;
  for (long i = tid; i < n; i += 1024) {
    const float a = r1[i], b = r2[i], t = x[i];
    acc -= (t * logf(a + eps) + (1.f - t) * logf(b + eps)) * inv;
    if (dr1) dr1[i] = -t / (a + eps) * inv;
    if (dr2) dr2[i] = -(1.f - t) / (b + eps) * inv;
  }
  sm[tid] = acc;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o) sm[tid] += sm[tid + o];
    __syncthreads();
  }
  if (tid == 0) loss[0] = sm[0];
}

// Input standardisation with the train-set statistics (scripts/train_AV_net.py:286-291, evaluate_audio_net.py:158-163):
//   out[r][f] = (x[r][f] - mean[f % nstat]) / (std[f % nstat] + eps)     nstat = F (audio: 513 bins) or 1 (video scalar)
__global__ void standardize_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ stdv,
                                   float* __restrict__ out, size_t n, int F, int nstat, float eps) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int f = nstat == 1 ? 0 : (int)(i % F);
    out[i] = (x[i] - mean[f]) / (stdv[f] + eps);
  }
}

// x / max|x| per utterance (scripts/evaluate_audio_net.py:125-127): one workgroup per row
__global__ void __launch_bounds__(1024) peak_normalize_kernel(const float* __restrict__ x, float* __restrict__ out, long L) {
  __shared__ float sm[1024];
  const float* xr = x + (long)blockIdx.x * L;
  float* orow = out + (long)blockIdx.x * L;
  float m = 0.f;
  for (long i = threadIdx.x; i < L; i += 1024) m = fmaxf(m, fabsf(xr[i]));
  sm[threadIdx.x] = m;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (threadIdx.x < o) sm[threadIdx.x] = fmaxf(sm[threadIdx.x], sm[threadIdx.x + o]);
    __syncthreads();
  }
  const float peak = sm[0];
  for (long i = threadIdx.x; i < L; i += 1024) orow[i] = xr[i] / peak;
}

// torch.optim.Adam semantics (no weight decay, no amsgrad): scripts/train_AV_net.py:238,306
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float step_size, float b1, float b2, float eps,
                                         float bc2_sqrt) {
  const float mi = b1 * m + (1.f - b1) * g;
  const float vi = b2 * v + (1.f - b2) * g * g;
  m = mi;
  v = vi;
  p -= step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
}
// 16-byte accesses over the flat buffers (7 streams of 4 n bytes: HBM-bound); VEC = 1 for unaligned buffers and the tail
template <int VEC>
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            size_t n, float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt) {
  const float step_size = lr / bc1;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n / VEC; i += (size_t)gridDim.x * blockDim.x) {
    if constexpr (VEC == 4) {
      float4 pi = reinterpret_cast<float4*>(p)[i], mi = reinterpret_cast<float4*>(m)[i], vi = reinterpret_cast<float4*>(v)[i];
      const float4 gi = reinterpret_cast<const float4*>(g)[i];
      adam_one(pi.x, gi.x, mi.x, vi.x, step_size, b1, b2, eps, bc2_sqrt);
      adam_one(pi.y, gi.y, mi.y, vi.y, step_size, b1, b2, eps, bc2_sqrt);
      adam_one(pi.z, gi.z, mi.z, vi.z, step_size, b1, b2, eps, bc2_sqrt);
      adam_one(pi.w, gi.w, mi.w, vi.w, step_size, b1, b2, eps, bc2_sqrt);
      reinterpret_cast<float4*>(m)[i] = mi;
      reinterpret_cast<float4*>(v)[i] = vi;
      reinterpret_cast<float4*>(p)[i] = pi;
    } else {
      adam_one(p[i], g[i], m[i], v[i], step_size, b1, b2, eps, bc2_sqrt);
    }
  }
}

__global__ void copy_cols_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t rows, int ncols, int src_ld,
                                 int src_off, int dst_ld, int dst_off) {
  const size_t n = rows * (size_t)ncols;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / ncols;
    const int c = (int)(i - r * ncols);
    dst[r * dst_ld + dst_off + c] = src[r * src_ld + src_off + c];
  }
}

// out[c] += sum_r X[r][c]
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ X, size_t rows, int cols, float* __restrict__ out) {
  __shared__ float sm[256];
  const int t = threadIdx.x, c = blockIdx.x * 64 + (t & 63), rl = t >> 6;
  float s = 0.f;
  if (c < cols)
    for (size_t r = rl; r < rows; r += 4) s += X[r * cols + c];
  sm[t] = s;
  __syncthreads();
  if (rl == 0 && c < cols) out[c] += sm[t] + sm[t + 64] + sm[t + 128] + sm[t + 192];
}

__global__ void scale_kernel(float* __restrict__ x, const float* __restrict__ scalar, size_t n) {
  const float a = scalar[0];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] *= a;
}

// in [B][C][T] -> out [B][T][C] through a padded LDS tile (both sides coalesced)
__global__ void __launch_bounds__(256) transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int T) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* ib = in + (long)b * C * T;
  float* ob = out + (long)b * C * T;
  for (int j = ty; j < 32; j += 8)
    if (c0 + j < C && t0 + tx < T) tile[j][tx] = ib[(long)(c0 + j) * T + t0 + tx];
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (t0 + j < T && c0 + tx < C) ob[(long)(t0 + j) * C + c0 + tx] = tile[tx][j];
}

static inline int grid1(size_t n) { size_t b = (n + 255) / 256; return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

}  // namespace

extern "C" int avvad_bce_masked(const float* logits, const float* targets, const int* lengths, float* loss, float* dlogits,
                                int B, int T, int Y, float eps, avvad_stream_t s) {
  AVVAD_ENTER();
  if (!logits || !targets || !lengths || !loss || B <= 0 || T <= 0 || Y <= 0) return AVVAD_EINVAL;
  hipLaunchKernelGGL(bce_masked_kernel, dim3(1), dim3(1024), 0, (hipStream_t)s, logits, targets, lengths, loss, dlogits, B, T,
                     Y, eps);
  AVVAD_LAUNCH_CHECK();
  return AVVAD_OK;
}

extern "C" int avvad_bce_2classes(const float* r1, const float* r2, const float* x, float* loss, float* dr1, float* dr2,
                                  long rows, int Y, float eps, avvad_stream_t s) {
  AVVAD_ENTER();
  if (!r1 || !r2 || !x || !loss || rows <= 0 || Y <= 0) return AVVAD_EINVAL;
  hipLaunchKernelGGL(bce_2classes_kernel, dim3(1), dim3(1024), 0, (hipStream_t)s, r1, r2, x, loss, dr1, dr2, rows, Y, eps);
  AVVAD_LAUNCH_CHECK();
  return AVVAD_OK;
}

extern "C" int avvad_standardize(const float* x, const float* mean, const float* stdv, float* out, size_t rows, int F, int nstat,
                                 float eps, avvad_stream_t s) {
  AVVAD_ENTER();
  if (!x || !mean || !stdv || !out || F <= 0 || (nstat != 1 && nstat != F)) return AVVAD_EINVAL;
  if (rows == 0) return AVVAD_OK;
  hipLaunchKernelGGL(standardize_kernel, dim3(grid1(rows * F)), dim3(256), 0, (hipStream_t)s, x, mean, stdv, out, rows * (size_t)F,
                     F, nstat, eps);
  AVVAD_LAUNCH_CHECK();
  return AVVAD_OK;
}

extern "C" int avvad_peak_normalize(const float* x, float* out, int B, long L, avvad_stream_t s) {
  AVVAD_ENTER();
  if (!x || !out || B <= 0 || L <= 0) return AVVAD_EINVAL;
  hipLaunchKernelGGL(peak_normalize_kernel, dim3(B), dim3(1024), 0, (hipStream_t)s, x, out, L);
  AVVAD_LAUNCH_CHECK();
  return AVVAD_OK;
}

extern "C" int avvad_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr,
                               float beta1, float beta2, float eps, int step, avvad_stream_t s) {
  AVVAD_ENTER();
  if (!param || !grad || !exp_avg || !exp_avg_sq || step < 1) return AVVAD_EINVAL;
  if (n == 0) return AVVAD_OK;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const bool al = (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0;
  const size_t n4 = al ? n / 4 * 4 : 0;
  if (n4)
    hipLaunchKernelGGL(adam_kernel<4>, dim3(grid1(n4 / 4)), dim3(256), 0, (hipStream_t)s, param, grad, exp_avg, exp_avg_sq, n4, lr,
                       beta1, beta2, eps, (float)bc1, (float)sqrt(bc2));
  if (n > n4)
    hipLaunchKernelGGL(adam_kernel<1>, dim3(grid1(n - n4)), dim3(256), 0, (hipStream_t)s, param + n4, grad + n4, exp_avg + n4,
                       exp_avg_sq + n4, n - n4, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2));
  AVVAD_LAUNCH_CHECK();
  return AVVAD_OK;
}

extern "C" int avvad_copy_cols(const float* src, float* dst, size_t rows, int ncols, int src_ld, int src_off, int dst_ld,
                               int dst_off, avvad_stream_t s) {
  AVVAD_ENTER();
  if (!src || !dst || ncols <= 0 || src_off < 0 || dst_off < 0 || src_ld < src_off + ncols || dst_ld < dst_off + ncols)
    return AVVAD_EINVAL;
  if (rows == 0) return AVVAD_OK;
  hipLaunchKernelGGL(copy_cols_kernel, dim3(grid1(rows * ncols)), dim3(256), 0, (hipStream_t)s, src, dst, rows, ncols, src_ld,
                     src_off, dst_ld, dst_off);
  AVVAD_LAUNCH_CHECK();
  return AVVAD_OK;
}

extern "C" int avvad_colsum_acc(const float* X, size_t rows, int cols, float* out, avvad_stream_t s) {
  AVVAD_ENTER();
  if (!X || !out || cols <= 0) return AVVAD_EINVAL;
  if (rows == 0) return AVVAD_OK;
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(cols, 64)), dim3(256), 0, (hipStream_t)s, X, rows, cols, out);
  AVVAD_LAUNCH_CHECK();
  return AVVAD_OK;
}

extern "C" int avvad_scale_by_device_scalar(float* x, const float* scalar, size_t n, avvad_stream_t s) {
  AVVAD_ENTER();
  if (!x || !scalar) return AVVAD_EINVAL;
  if (n == 0) return AVVAD_OK;
  hipLaunchKernelGGL(scale_kernel, dim3(grid1(n)), dim3(256), 0, (hipStream_t)s, x, scalar, n);
  AVVAD_LAUNCH_CHECK();
  return AVVAD_OK;
}

extern "C" int avvad_transpose_last2(const float* in, float* out, int B, int C, int T, avvad_stream_t s) {
  AVVAD_ENTER();
  if (!in || !out || B <= 0 || C <= 0 || T <= 0) return AVVAD_EINVAL;
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(T, 32), cdiv(C, 32), B), dim3(256), 0, (hipStream_t)s, in, out, C, T);
  AVVAD_LAUNCH_CHECK();
  return AVVAD_OK;
}
