// misc.hip -- loss, optimiser and small layout kernels of the AV-VAD hot path.
#include "common.h"

extern "C" const char* avvad_version(void) { return "avvad-hip 0.1 (gfx950, fp32 MFMA)"; }
extern "C" int avvad_abi_version(void) {
  AVVAD_ENTER(); return 1; }

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

// torch.optim.Adam semantics (no weight decay, no amsgrad): scripts/train_AV_net.py:238,306
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            size_t n, float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt) {
  const float step_size = lr / bc1;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
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

extern "C" int avvad_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr,
                               float beta1, float beta2, float eps, int step, avvad_stream_t s) {
  AVVAD_ENTER();
  if (!param || !grad || !exp_avg || !exp_avg_sq || step < 1) return AVVAD_EINVAL;
  if (n == 0) return AVVAD_OK;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  hipLaunchKernelGGL(adam_kernel, dim3(grid1(n)), dim3(256), 0, (hipStream_t)s, param, grad, exp_avg, exp_avg_sq, n, lr,
                     beta1, beta2, eps, (float)bc1, (float)sqrt(bc2));
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
