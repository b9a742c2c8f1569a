// gemm_lab.hip -- standalone kernel-structure experiments for the fp32-MFMA engine (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemm_lab gemm_lab.hip && ./gemm_lab
// C[M,N] = At[K,M]^T . B[K,N], M=N=K=4096, fp32, v_mfma_f32_32x32x2_f32.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ constexpr int mfma32_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
constexpr int BK = 32;

// ---- V0: the engine's 8-wave double-buffered structure, with ablation switches
template <int ABL>
__global__ void __launch_bounds__(512, 2) v0(const float* __restrict__ At, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
  constexpr int BM = 128, BN = 128, NTH = 512, LD = 132;
  constexpr int TILE = BK * LD * 2;
  __shared__ __attribute__((aligned(16))) float smem[2 * TILE];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 2, wn = wave & 3;
  const int ntn = N / BN;
  const int tile = blockIdx.x;
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
  const int ktiles = K / BK;
  // staging: 128x32 floats per operand / 512 thr = 8 floats = 2 float4
  float sa[2][4], sb[2][4];
  const int xl = (t & 31) * 4, kl0 = t >> 5;   // kl = kl0 + 16*i
  const int lm0 = (ABL & 16) ? 0 : m0, ln0 = (ABL & 16) ? 0 : n0;
  auto gload = [&](int kt) {
    if (ABL & 32) kt = kt & 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float4 ta = *reinterpret_cast<const float4*>(At + (long)(kt * BK + kl0 + 16 * i) * M + lm0 + xl);
      const float4 tb = *reinterpret_cast<const float4*>(B + (long)(kt * BK + kl0 + 16 * i) * N + ln0 + xl);
      sa[i][0] = ta.x; sa[i][1] = ta.y; sa[i][2] = ta.z; sa[i][3] = ta.w;
      sb[i][0] = tb.x; sb[i][1] = tb.y; sb[i][2] = tb.z; sb[i][3] = tb.w;
    }
  };
  auto to_lds = [&](float* S) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<float4*>(S + (kl0 + 16 * i) * LD + xl) = make_float4(sa[i][0], sa[i][1], sa[i][2], sa[i][3]);
      *reinterpret_cast<float4*>(S + BK * LD + (kl0 + 16 * i) * LD + xl) = make_float4(sb[i][0], sb[i][1], sb[i][2], sb[i][3]);
    }
  };
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  gload(0);
  to_lds(smem);
  gload(1);
  __syncthreads();
  for (int kt = 0; kt < ktiles; ++kt) {
    const int cur = kt & 1;
    float* As = smem + cur * TILE;
    float* Bs = As + BK * LD;
    const float* ap = As + lh * LD + wm * 64 + li;
    const float* bp = Bs + lh * LD + wn * 32 + li;
    float a[2][2], b[2];
    if (!(ABL & 8)) { a[0][0] = ap[0]; a[0][1] = ap[32]; b[0] = bp[0]; }
    else { a[0][0] = a[0][1] = a[1][0] = a[1][1] = (float)t; b[0] = b[1] = (float)lane; }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (ks == 8 && kt + 1 < ktiles) {
        float* An = smem + (cur ^ 1) * TILE;
        if (!(ABL & 2)) to_lds(An);
        if (!(ABL & 1)) { if (kt + 2 < ktiles) gload(kt + 2); }
      }
      if (!(ABL & 8) && ks + 1 < 16) {
        a[(ks + 1) & 1][0] = ap[(ks + 1) * 2 * LD];
        a[(ks + 1) & 1][1] = ap[(ks + 1) * 2 * LD + 32];
        b[(ks + 1) & 1] = bp[(ks + 1) * 2 * LD];
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = mfma32(a[ks & 1][0], b[ks & 1], acc[0]);
      acc[1] = mfma32(a[ks & 1][1], b[ks & 1], acc[1]);
    }
    if (!(ABL & 4)) __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = n0 + wn * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 64 + i * 32 + mfma32_row(r, lh);
      C[(long)m * N + n] = acc[i][r];
    }
  }
}

template <class F>
static float time_it(F launch, int iters = 10) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms / iters;
}

static std::vector<float> hA, hB;
static float *dA, *dB, *dC;
static const int M = 4096, N = 4096, K = 4096;

static void check(const char* name) {
  std::vector<float> hC(16 * N);
  CK(hipMemcpy(hC.data(), dC + (long)1000 * N, sizeof(float) * 16 * N, hipMemcpyDeviceToHost));
  double maxerr = 0;
  for (int r = 0; r < 2; ++r)
    for (int n = 0; n < N; n += 97) {
      double s = 0;
      for (int k = 0; k < K; ++k) s += (double)hA[(long)k * M + 1000 + r] * hB[(long)k * N + n];
      double e = fabs(s - hC[(long)r * N + n]);
      if (e > maxerr) maxerr = e;
    }
  printf("    check %-10s max abs err %.3e\n", name, maxerr);
}

#define RUN(name, check_it, ...)                                                        \
  do {                                                                                  \
    CK(hipMemset(dC, 0, sizeof(float) * (long)M * N));                                  \
    float ms = time_it([&]() { __VA_ARGS__; });                                         \
    printf("%-28s %8.3f ms  %7.1f TFLOP/s\n", name, ms, 2.0 * M * N * K / ms * 1e-9);   \
    if (check_it) check(name);                                                          \
    fflush(stdout);                                                                     \
  } while (0)

#include "variants.h"

int main() {
  hA.resize((long)K * M); hB.resize((long)K * N);
  srand(1);
  for (auto& v : hA) v = (rand() % 2001 - 1000) * 1e-3f;
  for (auto& v : hB) v = (rand() % 2001 - 1000) * 1e-3f;
  CK(hipMalloc(&dA, sizeof(float) * hA.size())); CK(hipMalloc(&dB, sizeof(float) * hB.size())); CK(hipMalloc(&dC, sizeof(float) * (long)M * N));
  CK(hipMemcpy(dA, hA.data(), sizeof(float) * hA.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, hB.data(), sizeof(float) * hB.size(), hipMemcpyHostToDevice));
  const int tiles = (M / 128) * (N / 128);
  RUN("v0 full", 1, hipLaunchKernelGGL(v0<0>, dim3(tiles), dim3(512), 0, 0, dA, dB, dC, M, N, K));
  RUN("v0 -gload", 0, hipLaunchKernelGGL(v0<1>, dim3(tiles), dim3(512), 0, 0, dA, dB, dC, M, N, K));
  RUN("v0 -gload-ldswr", 0, hipLaunchKernelGGL(v0<3>, dim3(tiles), dim3(512), 0, 0, dA, dB, dC, M, N, K));
  RUN("v0 -gload-ldswr-bar", 0, hipLaunchKernelGGL(v0<7>, dim3(tiles), dim3(512), 0, 0, dA, dB, dC, M, N, K));
  RUN("v0 mfma only", 0, hipLaunchKernelGGL(v0<15>, dim3(tiles), dim3(512), 0, 0, dA, dB, dC, M, N, K));
  RUN("v0 -bar only", 0, hipLaunchKernelGGL(v0<4>, dim3(tiles), dim3(512), 0, 0, dA, dB, dC, M, N, K));
  RUN("v0 -dsread only", 0, hipLaunchKernelGGL(v0<8>, dim3(tiles), dim3(512), 0, 0, dA, dB, dC, M, N, K));
  RUN("v0 same panel (L2 hot)", 0, hipLaunchKernelGGL(v0<16>, dim3(tiles), dim3(512), 0, 0, dA, dB, dC, M, N, K));
  RUN("v0 same ktile (L1 hot)", 0, hipLaunchKernelGGL(v0<48>, dim3(tiles), dim3(512), 0, 0, dA, dB, dC, M, N, K));
  run_variants();
  return 0;
}
