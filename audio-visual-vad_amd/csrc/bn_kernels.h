// bn_kernels.h -- BatchNorm building blocks over a row-major [M][C] fp32 matrix (C % 4 == 0, C <= BN_MAXC):
// two-stage fp64 column reductions, statistics -> scale/shift, fused normalise(+residual)(+ReLU), and the
// backward pair.  Shared by the ResNet trunk (BatchNorm2d over NHWC pixels) and the MCB fusion (BatchNorm1d
// over (sequence, frame) rows).  Included into an anonymous namespace by each user.
#pragma once
#include "common.h"

constexpr int BN_MAXC = 1024;
constexpr int MAXC = BN_MAXC;
constexpr int STAT_CHUNKS = 512;    // workgroups of a column reduction (2 per CU; 1024 timed the same and doubled the finalize kernels' reads)

// ------------------------------------------------------------------ column reductions ([M][C] fp32 -> per-channel fp64 sums)
// MODE 0: sum x, sum x^2           (batch statistics)
// MODE 1: sum g, sum g*xhat  with g = dy * (ymask > 0 if ymask)   (BatchNorm backward)
//         msc/msh (instead of ymask): the ReLU mask of a BatchNorm whose output went straight into a ReLU is rebuilt from x
//         itself, fmaf(x, msc[c], msh[c]) > 0 -- the forward's own expression on the saved scale/shift, so the same bits --
//         and the activation tensor is not read at all (one of the three planes this pass streams)
//         qmask (instead of ymask): one byte per quad written by bn_act, bit j = (y[j] > 0)
template <int MODE>
__global__ void __launch_bounds__(256)
    col_reduce(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ ymask,
               const float* __restrict__ mean, const float* __restrict__ invstd, long M, int C, long rows_per_chunk,
               double* __restrict__ part, const float* __restrict__ msc = nullptr, const float* __restrict__ msh = nullptr,
               const unsigned char* __restrict__ qmask = nullptr) {
  __shared__ double sm[256 * 8];
  const int t = threadIdx.x;
  const int Q = C >> 2, RL = 256 / Q;
  const int q = t % Q, rl = t / Q;
  double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
  float mu[4] = {0, 0, 0, 0}, is[4] = {1, 1, 1, 1};
  float ksc[4] = {0, 0, 0, 0}, ksh[4] = {0, 0, 0, 0};
  if (MODE == 1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { mu[j] = mean[q * 4 + j]; is[j] = invstd[q * 4 + j]; }
    if (msc) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { ksc[j] = msc[q * 4 + j]; ksh[j] = msh[q * 4 + j]; }
    }
  }
  const long r0 = blockIdx.x * rows_per_chunk;
  const long r1 = min(M, r0 + rows_per_chunk);
  // 4 rows per trip: all loads of the trip are issued before any is consumed (one row per trip left a 256-block grid
  // with ~3 MB in flight -- latency-bound at a third of the HBM rate)
  constexpr int U = 4;
  if (rl < RL)
    for (long r = r0 + rl; r < r1; r += (long)U * RL) {
      float4 xv4[U], dv4[U], yv4[U];
      unsigned qm[U];
      bool live[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long rr = r + (long)u * RL;
        live[u] = rr < r1;
        const long o = (live[u] ? rr : r) * C + q * 4;
        xv4[u] = *reinterpret_cast<const float4*>(x + o);
        if (MODE == 1) {
          dv4[u] = *reinterpret_cast<const float4*>(dy + o);
          if (ymask) yv4[u] = *reinterpret_cast<const float4*>(ymask + o);
          else if (qmask) qm[u] = qmask[o >> 2];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!live[u]) continue;
        const float xv[4] = {xv4[u].x, xv4[u].y, xv4[u].z, xv4[u].w};
        if (MODE == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { s[j] += xv[j]; ss[j] += (double)xv[j] * xv[j]; }
        } else {
          float g[4] = {dv4[u].x, dv4[u].y, dv4[u].z, dv4[u].w};
          if (ymask) {
            if (!(yv4[u].x > 0.f)) g[0] = 0.f;
            if (!(yv4[u].y > 0.f)) g[1] = 0.f;
            if (!(yv4[u].z > 0.f)) g[2] = 0.f;
            if (!(yv4[u].w > 0.f)) g[3] = 0.f;
          } else if (qmask) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (!((qm[u] >> j) & 1u)) g[j] = 0.f;
          } else if (msc) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (!(fmaf(xv[j], ksc[j], ksh[j]) > 0.f)) g[j] = 0.f;
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) { s[j] += g[j]; ss[j] += (double)g[j] * ((xv[j] - mu[j]) * is[j]); }
        }
      }
    }
#pragma unroll
  for (int j = 0; j < 4; ++j) { sm[t * 8 + j] = s[j]; sm[t * 8 + 4 + j] = ss[j]; }
  __syncthreads();
  for (int c = t; c < C; c += 256) {
    const int cq = c >> 2, cj = c & 3;
    double a = 0, b = 0;
    for (int l = 0; l < RL; ++l) { a += sm[(l * Q + cq) * 8 + cj]; b += sm[(l * Q + cq) * 8 + 4 + cj]; }
    part[((long)blockIdx.x * 2 + 0) * C + c] = a;
    part[((long)blockIdx.x * 2 + 1) * C + c] = b;
  }
}

// sum the per-chunk partials of channel c = blockIdx.x*FIN_CH + (tid % FIN_CH): 256/FIN_CH chunk-lanes per channel, each
// adding its strided subset in order, then a fixed-shape pairwise tree in LDS (deterministic); every thread of the
// 256-thread block must call it; lanes tid < FIN_CH get the totals.
// (4 channels per block: the 40 finalize launches of a step are pure latency -- with 8 channels per block a 64-channel
//  layer ran on 8 CUs, every lane walked ~30 dependent loads and one lane per channel then read 32 LDS words in series:
//  9 us per launch, 0.36 ms per step.)
constexpr int FIN_CH = 4;    // channels per finalize block (grid = ceil(C / FIN_CH))
__device__ __forceinline__ void chunk_sums(const double* __restrict__ part, int nchunk, int C, double& s, double& ss) {
  constexpr int KL = 256 / FIN_CH;
  __shared__ double sm[2][KL][FIN_CH];
  const int cl = threadIdx.x % FIN_CH, kl = threadIdx.x / FIN_CH;
  const int c = blockIdx.x * FIN_CH + cl;
  double a = 0, b = 0;
  if (c < C) {
#pragma unroll 4
    for (int i = kl; i < nchunk; i += KL) { a += part[((long)i * 2 + 0) * C + c]; b += part[((long)i * 2 + 1) * C + c]; }
  }
  sm[0][kl][cl] = a; sm[1][kl][cl] = b;
  __syncthreads();
#pragma unroll
  for (int off = KL / 2; off > 0; off >>= 1) {
    if (kl < off) { sm[0][kl][cl] += sm[0][kl + off][cl]; sm[1][kl][cl] += sm[1][kl + off][cl]; }
    __syncthreads();
  }
  s = sm[0][0][cl]; ss = sm[1][0][cl];
}

// finalize batch statistics -> scale/shift (+ saved mean/invstd, running-stat update)
__global__ void bn_finalize(const double* __restrict__ part, int nchunk, long M, int C, const float* __restrict__ gamma,
                            const float* __restrict__ beta, float* __restrict__ rm, float* __restrict__ rv, int training,
                            float momentum, float eps, float* __restrict__ scale, float* __restrict__ shift,
                            float* __restrict__ mean_o, float* __restrict__ invstd_o) {
  double s = 0, ss = 0;
  if (training) chunk_sums(part, nchunk, C, s, ss);
  const int c = blockIdx.x * FIN_CH + (threadIdx.x % FIN_CH);
  if (c >= C || threadIdx.x >= FIN_CH) return;
  float mean, var;
  if (training) {
    const double m = s / (double)M;
    double v = ss / (double)M - m * m;
    if (v < 0) v = 0;
    mean = (float)m; var = (float)v;
    const double unb = M > 1 ? v * (double)M / (double)(M - 1) : v;
    rm[c] = (1.f - momentum) * rm[c] + momentum * mean;
    rv[c] = (1.f - momentum) * rv[c] + momentum * (float)unb;
  } else {
    mean = rm[c]; var = rv[c];
  }
  const float is = 1.0f / sqrtf(var + eps);
  const float sc = gamma[c] * is;
  scale[c] = sc;
  shift[c] = beta[c] - mean * sc;
  mean_o[c] = mean;
  invstd_o[c] = is;
}

// finalize BatchNorm backward sums -> dgamma/dbeta (accumulated) and the apply coefficients
//   dx = k0 * (g - k1 - xhat * k2)
__global__ void bn_bwd_finalize(const double* __restrict__ part, int nchunk, long M, int C, const float* __restrict__ gamma,
                                const float* __restrict__ invstd, int training, float* __restrict__ dgamma,
                                float* __restrict__ dbeta, float* __restrict__ coef) {
  double s = 0, ss = 0;
  chunk_sums(part, nchunk, C, s, ss);
  const int c = blockIdx.x * FIN_CH + (threadIdx.x % FIN_CH);
  if (c >= C || threadIdx.x >= FIN_CH) return;
  if (dbeta) dbeta[c] += (float)s;
  if (dgamma) dgamma[c] += (float)ss;
  coef[c] = gamma[c] * invstd[c];
  coef[MAXC + c] = training ? (float)(s / (double)M) : 0.f;
  coef[2 * MAXC + c] = training ? (float)(ss / (double)M) : 0.f;
}

// ------------------------------------------------------------------ fused elementwise (16-byte accesses, C % 4 == 0)
__device__ __forceinline__ float bn_affine(float x, float sc, float sh) { return fmaf(x, sc, sh); }

// Quads of bf16 (the trunk's bf16 data path, option "bf16"): activations that only convolutions consume are STORED as bf16
// (round to nearest even: a plain cast, v_cvt_pk_bf16_f32 on gfx950 -- NaNs stay NaNs) by the elementwise kernel that
// produces them; 8-byte accesses.
typedef __bf16 bn_bf16x4 __attribute__((ext_vector_type(4)));
typedef float bn_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_quad_bf16(void* base, long quad, const float (&o)[4]) {
  const bn_f32x4 f = {o[0], o[1], o[2], o[3]};
  reinterpret_cast<bn_bf16x4*>(base)[quad] = __builtin_convertvector(f, bn_bf16x4);
}
__device__ __forceinline__ float4 load_quad_bf16(const void* base, long quad) {
  const bn_bf16x4 h = reinterpret_cast<const bn_bf16x4*>(base)[quad];
  const bn_f32x4 f = __builtin_convertvector(h, bn_f32x4);
  return make_float4(f[0], f[1], f[2], f[3]);
}

// The elementwise kernels run 256-thread blocks with a grid-stride loop: the stride (gridDim*256 quads) is a multiple
// of C/4 whenever C/4 divides 256, so a thread's channel quad never changes and its per-channel coefficients are
// loaded ONCE into registers -- per-iteration coefficient loads (up to 20 scalar loads per float4 of data) made these
// kernels TA-issue-bound at ~1/3 of the HBM rate.  Other C fall back to reloading each iteration.
__device__ __forceinline__ float4 ld4(const float* p, int c) { return *reinterpret_cast<const float4*>(p + c); }

// y = [relu]( x*scale+shift  [+ idn | + idn*iscale+ishift] )
// OB: y is stored as bf16; IB: the identity input idn is bf16 (a block input of the bf16 data path; never with iscale)
template <bool OB = false, bool IB = false>
__global__ void __launch_bounds__(256)
    bn_act(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
           const float* __restrict__ idn, const float* __restrict__ iscale, const float* __restrict__ ishift,
           float* __restrict__ y, long nquad, int C, int relu, unsigned char* __restrict__ qmask_out = nullptr) {
  const bool fixed = (256 % (C >> 2)) == 0;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int c = (int)((i * 4) % C);
  float4 sc = ld4(scale, c), sh = ld4(shift, c), a = make_float4(0, 0, 0, 0), b = a;
  if (iscale) { a = ld4(iscale, c); b = ld4(ishift, c); }
  for (; i < nquad; i += (long)gridDim.x * blockDim.x) {
    if (!fixed) {
      c = (int)((i * 4) % C);
      sc = ld4(scale, c); sh = ld4(shift, c);
      if (iscale) { a = ld4(iscale, c); b = ld4(ishift, c); }
    }
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    float o[4] = {bn_affine(v.x, sc.x, sh.x), bn_affine(v.y, sc.y, sh.y), bn_affine(v.z, sc.z, sh.z), bn_affine(v.w, sc.w, sh.w)};
    if (idn) {
      const float4 d = IB ? load_quad_bf16(idn, i) : reinterpret_cast<const float4*>(idn)[i];
      if (iscale) {
        o[0] += bn_affine(d.x, a.x, b.x); o[1] += bn_affine(d.y, a.y, b.y);
        o[2] += bn_affine(d.z, a.z, b.z); o[3] += bn_affine(d.w, a.w, b.w);
      } else { o[0] += d.x; o[1] += d.y; o[2] += d.z; o[3] += d.w; }
    }
    if (relu) { o[0] = fmaxf(o[0], 0.f); o[1] = fmaxf(o[1], 0.f); o[2] = fmaxf(o[2], 0.f); o[3] = fmaxf(o[3], 0.f); }
    if constexpr (OB) store_quad_bf16(y, i, o);
    else reinterpret_cast<float4*>(y)[i] = make_float4(o[0], o[1], o[2], o[3]);
    // the ReLU mask of this quad as one byte (bit j: y[j] > 0): what the backward reads instead of y (1/16 of its bytes)
    if (qmask_out)
      qmask_out[i] = (unsigned char)((o[0] > 0.f ? 1 : 0) | (o[1] > 0.f ? 2 : 0) | (o[2] > 0.f ? 4 : 0) | (o[3] > 0.f ? 8 : 0));
  }
}

// dx = k0*(g - k1 - xhat*k2), g = dy*(ymask>0);  optionally also writes g (identity branch of the residual)
// (dx may alias dy: each element is read, then written, by the same thread)
// OB: dx is stored as bf16 (the bf16 data path: dx only feeds the data- and weight-gradient convolutions; it must then NOT
// alias dy)
template <bool OB = false>
__global__ void __launch_bounds__(256)
    bn_bwd_apply(const float* __restrict__ x, const float* dy, const float* __restrict__ ymask,
                 const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ coef,
                 float* dx, float* gout, long nquad, int C, const float* __restrict__ msc = nullptr,
                 const float* __restrict__ msh = nullptr, const unsigned char* __restrict__ qmask = nullptr) {
  const bool fixed = (256 % (C >> 2)) == 0;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int c = (int)((i * 4) % C);
  float4 mu = ld4(mean, c), is = ld4(invstd, c), k0 = ld4(coef, c), k1 = ld4(coef + MAXC, c), k2 = ld4(coef + 2 * MAXC, c);
  float4 ms = make_float4(0, 0, 0, 0), mh = ms;      // ReLU mask rebuilt from x (see col_reduce)
  if (msc) { ms = ld4(msc, c); mh = ld4(msh, c); }
  for (; i < nquad; i += (long)gridDim.x * blockDim.x) {
    if (!fixed) {
      c = (int)((i * 4) % C);
      mu = ld4(mean, c); is = ld4(invstd, c); k0 = ld4(coef, c); k1 = ld4(coef + MAXC, c); k2 = ld4(coef + 2 * MAXC, c);
      if (msc) { ms = ld4(msc, c); mh = ld4(msh, c); }
    }
    const float4 xv = reinterpret_cast<const float4*>(x)[i];
    const float4 d = reinterpret_cast<const float4*>(dy)[i];
    float g[4] = {d.x, d.y, d.z, d.w};
    if (ymask) {
      const float4 y = reinterpret_cast<const float4*>(ymask)[i];
      if (!(y.x > 0.f)) g[0] = 0.f;
      if (!(y.y > 0.f)) g[1] = 0.f;
      if (!(y.z > 0.f)) g[2] = 0.f;
      if (!(y.w > 0.f)) g[3] = 0.f;
    } else if (qmask) {
      const unsigned qm = qmask[i];
      if (!(qm & 1u)) g[0] = 0.f;
      if (!(qm & 2u)) g[1] = 0.f;
      if (!(qm & 4u)) g[2] = 0.f;
      if (!(qm & 8u)) g[3] = 0.f;
    } else if (msc) {
      if (!(fmaf(xv.x, ms.x, mh.x) > 0.f)) g[0] = 0.f;
      if (!(fmaf(xv.y, ms.y, mh.y) > 0.f)) g[1] = 0.f;
      if (!(fmaf(xv.z, ms.z, mh.z) > 0.f)) g[2] = 0.f;
      if (!(fmaf(xv.w, ms.w, mh.w) > 0.f)) g[3] = 0.f;
    }
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
    const float m4[4] = {mu.x, mu.y, mu.z, mu.w}, i4[4] = {is.x, is.y, is.z, is.w};
    const float a4[4] = {k0.x, k0.y, k0.z, k0.w}, b4[4] = {k1.x, k1.y, k1.z, k1.w}, c4[4] = {k2.x, k2.y, k2.z, k2.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = (xs[j] - m4[j]) * i4[j];
      o[j] = a4[j] * (g[j] - b4[j] - xh * c4[j]);
    }
    if constexpr (OB) store_quad_bf16(dx, i, o);
    else reinterpret_cast<float4*>(dx)[i] = make_float4(o[0], o[1], o[2], o[3]);
    if (gout) reinterpret_cast<float4*>(gout)[i] = make_float4(g[0], g[1], g[2], g[3]);
  }
}
