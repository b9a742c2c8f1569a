// ---- L1: the "untuned" shape: 256 threads, 2x2 waves of 64x64, [k][x] LDS image, b32 fragment reads, no software pipelining
__global__ void __launch_bounds__(256) l1(const float* __restrict__ At, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
  constexpr int LD = 132;
  __shared__ __attribute__((aligned(16))) float smem[2 * BK * LD];
  float* As = smem; float* Bs = smem + BK * LD;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = N / 128;
  const int m0 = (blockIdx.x / ntn) * 128, n0 = (blockIdx.x % ntn) * 128;
  const int xl = (t & 31) * 4, kl0 = t >> 5;  // kl = kl0 + 8*i, i<4
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int kt = 0; kt < K / BK; ++kt) {
    float4 sa[4], sb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sa[i] = *reinterpret_cast<const float4*>(At + (long)(kt * BK + kl0 + 8 * i) * M + m0 + xl);
      sb[i] = *reinterpret_cast<const float4*>(B + (long)(kt * BK + kl0 + 8 * i) * N + n0 + xl);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<float4*>(As + (kl0 + 8 * i) * LD + xl) = sa[i];
      *reinterpret_cast<float4*>(Bs + (kl0 + 8 * i) * LD + xl) = sb[i];
    }
    __syncthreads();
    const float* ap = As + lh * LD + wm * 64 + li;
    const float* bp = Bs + lh * LD + wn * 64 + li;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const float a0 = ap[ks * 2 * LD], a1 = ap[ks * 2 * LD + 32], b0 = bp[ks * 2 * LD], b1 = bp[ks * 2 * LD + 32];
      acc[0][0] = mfma32(a0, b0, acc[0][0]);
      acc[0][1] = mfma32(a0, b1, acc[0][1]);
      acc[1][0] = mfma32(a1, b0, acc[1][0]);
      acc[1][1] = mfma32(a1, b1, acc[1][1]);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) C[(long)(m0 + wm * 64 + i * 32 + mfma32_row(r, lh)) * N + n] = acc[i][j][r];
    }
}

// ---- L2: NT operands (A[M][K], Bt[N][K]), [x][k] swizzled LDS image, ds_read_b128 along k (k-permuted MFMA order)
template <int PIPE>
__global__ void __launch_bounds__(256) l2(const float* __restrict__ A, const float* __restrict__ Bt, float* __restrict__ C, int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 128 * BK];
  float* As = smem; float* Bs = smem + 128 * BK;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = N / 128;
  const int m0 = (blockIdx.x / ntn) * 128, n0 = (blockIdx.x % ntn) * 128;
  const int s = t & 7, r0 = t >> 3;  // row r0 + 32*i
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 sa[4], sb[4];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sa[i] = *reinterpret_cast<const float4*>(A + (long)(m0 + r0 + 32 * i) * K + kt * BK + s * 4);
      sb[i] = *reinterpret_cast<const float4*>(Bt + (long)(n0 + r0 + 32 * i) * K + kt * BK + s * 4);
    }
  };
  auto to_lds = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = r0 + 32 * i;
      *reinterpret_cast<float4*>(As + r * 32 + ((s ^ ((r >> 1) & 7)) * 4)) = sa[i];
      *reinterpret_cast<float4*>(Bs + r * 32 + ((s ^ ((r >> 1) & 7)) * 4)) = sb[i];
    }
  };
  if (PIPE) gload(0);
  for (int kt = 0; kt < K / BK; ++kt) {
    if (!PIPE) gload(kt);
    __syncthreads();
    to_lds();
    __syncthreads();
    if (PIPE && kt + 1 < K / BK) gload(kt + 1);
    const int ra = wm * 64 + li, rb = wn * 64 + li;   // +32 for the second fragment: (row>>1)&7 unchanged
    const int sw = (ra >> 1) & 7, swb = (rb >> 1) & 7;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 a0 = *reinterpret_cast<const float4*>(As + ra * 32 + (((2 * j + lh) ^ sw) * 4));
      const float4 a1 = *reinterpret_cast<const float4*>(As + (ra + 32) * 32 + (((2 * j + lh) ^ sw) * 4));
      const float4 b0 = *reinterpret_cast<const float4*>(Bs + rb * 32 + (((2 * j + lh) ^ swb) * 4));
      const float4 b1 = *reinterpret_cast<const float4*>(Bs + (rb + 32) * 32 + (((2 * j + lh) ^ swb) * 4));
      const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
      const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0][0] = mfma32(av0[e], bv0[e], acc[0][0]);
        acc[0][1] = mfma32(av0[e], bv1[e], acc[0][1]);
        acc[1][0] = mfma32(av1[e], bv0[e], acc[1][0]);
        acc[1][1] = mfma32(av1[e], bv1[e], acc[1][1]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) C[(long)(m0 + wm * 64 + i * 32 + mfma32_row(r, lh)) * N + n] = acc[i][j][r];
    }
}

// ---- v16: v0's structure on v_mfma_f32_16x16x4_f32 (same 64x32 wave tile, LD = 144 keeps the 4 k-rows of a fragment on distinct banks)
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int MFMA_ONLY>
__global__ void __launch_bounds__(512, 2) v16(const float* __restrict__ At, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K,
                                               unsigned long long* clk) {
  constexpr int BM = 128, BN = 128, LD = 144;
  constexpr int TILE = BK * LD * 2;
  __shared__ __attribute__((aligned(16))) float smem[2 * TILE];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, lq = lane >> 4;
  const int wm = wave >> 2, wn = wave & 3;
  const int ntn = N / BN;
  const int tile = blockIdx.x;
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
  const int ktiles = K / BK;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float sa[2][4], sb[2][4];
  const int xl = (t & 31) * 4, kl0 = t >> 5;
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float4 ta = *reinterpret_cast<const float4*>(At + (long)(kt * BK + kl0 + 16 * i) * M + m0 + xl);
      const float4 tb = *reinterpret_cast<const float4*>(B + (long)(kt * BK + kl0 + 16 * i) * N + n0 + xl);
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
  f32x4v acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
  gload(0);
  to_lds(smem);
  gload(1);
  __syncthreads();
  for (int kt = 0; kt < ktiles; ++kt) {
    const int cur = kt & 1;
    float* As = smem + cur * TILE;
    float* Bs = As + BK * LD;
    const float* ap = As + lq * LD + wm * 64 + l15;
    const float* bp = Bs + lq * LD + wn * 32 + l15;
    float a[2][4], b[2][2];
    if (!MFMA_ONLY) {
#pragma unroll
      for (int i = 0; i < 4; ++i) a[0][i] = ap[i * 16];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[0][j] = bp[j * 16];
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) a[0][i] = a[1][i] = At[t + i];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[0][j] = b[1][j] = B[t + j];
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (!MFMA_ONLY && ks == 4 && kt + 1 < ktiles) {
        float* An = smem + (cur ^ 1) * TILE;
        to_lds(An);
        if (kt + 2 < ktiles) gload(kt + 2);
      }
      if (!MFMA_ONLY && ks + 1 < 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[(ks + 1) & 1][i] = ap[(ks + 1) * 4 * LD + i * 16];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[(ks + 1) & 1][j] = bp[(ks + 1) * 4 * LD + j * 16];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks & 1][i], b[ks & 1][j], acc[i][j], 0, 0, 0);
    }
    if (!MFMA_ONLY) __syncthreads();
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (t == 0 && clk) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 32 + j * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) C[(long)(m0 + wm * 64 + i * 16 + 4 * lq + r) * N + n] = acc[i][j][r];
    }
}

// v0 with clock stamps (same body as v0<0> / v0<15>)
template <int ABL>
__global__ void __launch_bounds__(512, 2) v0c(const float* __restrict__ At, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K,
                                               unsigned long long* clk) {
  constexpr int BM = 128, BN = 128, LD = 132;
  constexpr int TILE = BK * LD * 2;
  __shared__ __attribute__((aligned(16))) float smem[2 * TILE];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 2, wn = wave & 3;
  const int ntn = N / BN;
  const int tile = blockIdx.x;
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
  const int ktiles = K / BK;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float sa[2][4], sb[2][4];
  const int xl = (t & 31) * 4, kl0 = t >> 5;
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float4 ta = *reinterpret_cast<const float4*>(At + (long)(kt * BK + kl0 + 16 * i) * M + m0 + xl);
      const float4 tb = *reinterpret_cast<const float4*>(B + (long)(kt * BK + kl0 + 16 * i) * N + n0 + xl);
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
    else { a[0][0] = a[1][0] = At[t]; a[0][1] = a[1][1] = At[t + 1]; b[0] = b[1] = B[t]; }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (!(ABL & 8) && ks == 8 && kt + 1 < ktiles) {
        float* An = smem + (cur ^ 1) * TILE;
        to_lds(An);
        if (kt + 2 < ktiles) gload(kt + 2);
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
    if (!(ABL & 8)) __syncthreads();
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (t == 0 && clk) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = n0 + wn * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) C[(long)(m0 + wm * 64 + i * 32 + mfma32_row(r, lh)) * N + n] = acc[i][r];
  }
}

// ---- v2: 256x128 tile, 8 waves as 4x2, each 64x64 (4 accumulators), double-buffered LDS (101 KB -> 1 workgroup / CU)
template <int PERSIST>
__global__ void __launch_bounds__(512, 1) v2(const float* __restrict__ At, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K,
                                              unsigned long long* clk) {
  constexpr int BM = 256, BN = 128, LDA = 260, LDB = 132;
  constexpr int TILE = BK * LDA + BK * LDB;
  __shared__ __attribute__((aligned(16))) float smem[2 * TILE];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = N / BN;
  const int ntiles = (M / BM) * ntn;
  const int ktiles = K / BK;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  const int xa = (t & 63) * 4, ka0 = t >> 6;    // A: 64 float4 per k-row, 8 rows per pass, 4 passes
  const int xb = (t & 31) * 4, kb0 = t >> 5;    // B: 32 float4 per k-row, 16 rows per pass, 2 passes
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    float sa[4][4], sb[2][4];
    auto gload = [&](int kt) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 ta = *reinterpret_cast<const float4*>(At + (long)(kt * BK + ka0 + 8 * i) * M + m0 + xa);
        sa[i][0] = ta.x; sa[i][1] = ta.y; sa[i][2] = ta.z; sa[i][3] = ta.w;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float4 tb = *reinterpret_cast<const float4*>(B + (long)(kt * BK + kb0 + 16 * i) * N + n0 + xb);
        sb[i][0] = tb.x; sb[i][1] = tb.y; sb[i][2] = tb.z; sb[i][3] = tb.w;
      }
    };
    auto to_lds = [&](float* S) {
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(S + (ka0 + 8 * i) * LDA + xa) = make_float4(sa[i][0], sa[i][1], sa[i][2], sa[i][3]);
#pragma unroll
      for (int i = 0; i < 2; ++i) *reinterpret_cast<float4*>(S + BK * LDA + (kb0 + 16 * i) * LDB + xb) = make_float4(sb[i][0], sb[i][1], sb[i][2], sb[i][3]);
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    gload(0);
    __syncthreads();
    to_lds(smem);
    gload(1);
    __syncthreads();
    for (int kt = 0; kt < ktiles; ++kt) {
      const int cur = kt & 1;
      float* As = smem + cur * TILE;
      float* Bs = As + BK * LDA;
      const float* ap = As + lh * LDA + wm * 64 + li;
      const float* bp = Bs + lh * LDB + wn * 64 + li;
      float a[2][2], b[2][2];
      a[0][0] = ap[0]; a[0][1] = ap[32]; b[0][0] = bp[0]; b[0][1] = bp[32];
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        if (ks == 8 && kt + 1 < ktiles) {
          float* An = smem + (cur ^ 1) * TILE;
          to_lds(An);
          if (kt + 2 < ktiles) gload(kt + 2);
        }
        if (ks + 1 < 16) {
          a[(ks + 1) & 1][0] = ap[(ks + 1) * 2 * LDA];
          a[(ks + 1) & 1][1] = ap[(ks + 1) * 2 * LDA + 32];
          b[(ks + 1) & 1][0] = bp[(ks + 1) * 2 * LDB];
          b[(ks + 1) & 1][1] = bp[(ks + 1) * 2 * LDB + 32];
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = mfma32(a[ks & 1][0], b[ks & 1][0], acc[0][0]);
        acc[0][1] = mfma32(a[ks & 1][0], b[ks & 1][1], acc[0][1]);
        acc[1][0] = mfma32(a[ks & 1][1], b[ks & 1][0], acc[1][0]);
        acc[1][1] = mfma32(a[ks & 1][1], b[ks & 1][1], acc[1][1]);
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) C[(long)(m0 + wm * 64 + i * 32 + mfma32_row(r, lh)) * N + n] = acc[i][j][r];
      }
    if (!PERSIST) break;
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (t == 0 && clk && blockIdx.x < 1024) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

static unsigned long long* dClk;
static void report_clock(const char* name, int tiles) {
  std::vector<unsigned long long> h(2 * tiles);
  CK(hipMemcpy(h.data(), dClk, sizeof(unsigned long long) * 2 * tiles, hipMemcpyDeviceToHost));
  std::vector<double> f;
  double cyc = 0;
  for (int i = 0; i < tiles; ++i) { f.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1); cyc += (double)h[2 * i]; }
  std::sort(f.begin(), f.end());
  printf("    %-24s in-kernel clock median %.3f GHz (min %.3f max %.3f), mean cycles per workgroup %.0f\n", name, f[tiles / 2], f[0], f[tiles - 1], cyc / tiles);
}
// sustained: ~1.5 s of back-to-back launches, time the last 200
#define SUSTAIN(name, ...)                                                                \
  do {                                                                                    \
    for (int i = 0; i < 1200; ++i) { __VA_ARGS__; }                                       \
    float ms = time_it([&]() { __VA_ARGS__; }, 200);                                      \
    printf("%-28s %8.3f ms  %7.1f TFLOP/s (sustained)\n", name, ms, 2.0 * M * N * K / ms * 1e-9); \
    report_clock(name, tiles);                                                            \
    fflush(stdout);                                                                       \
  } while (0)

static float *dAr, *dBt;   // A[M][K], Bt[N][K]
static void run_variants() {
  std::vector<float> hAr((long)M * K), hBt((long)N * K);
  for (int k = 0; k < K; ++k) for (int m = 0; m < M; ++m) hAr[(long)m * K + k] = hA[(long)k * M + m];
  for (int k = 0; k < K; ++k) for (int n = 0; n < N; ++n) hBt[(long)n * K + k] = hB[(long)k * N + n];
  CK(hipMalloc(&dAr, sizeof(float) * hAr.size())); CK(hipMalloc(&dBt, sizeof(float) * hBt.size()));
  CK(hipMemcpy(dAr, hAr.data(), sizeof(float) * hAr.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dBt, hBt.data(), sizeof(float) * hBt.size(), hipMemcpyHostToDevice));
  const int tiles = (M / 128) * (N / 128);
  CK(hipMalloc(&dClk, sizeof(unsigned long long) * 2 * tiles));
  SUSTAIN("v0c 32x32x2 full", hipLaunchKernelGGL(v0c<0>, dim3(tiles), dim3(512), 0, 0, dA, dB, dC, M, N, K, dClk));
  check("v0c");
  { const int tiles = 256; SUSTAIN("v2 256x128 persistent 256wg", hipLaunchKernelGGL(v2<1>, dim3(256), dim3(512), 0, 0, dA, dB, dC, M, N, K, dClk)); }
  check("v2p");
  { const int tiles = 512; SUSTAIN("v2 256x128 512 blocks", hipLaunchKernelGGL(v2<0>, dim3(512), dim3(512), 0, 0, dA, dB, dC, M, N, K, dClk)); }
  check("v2");
  SUSTAIN("v16 16x16x4 full", hipLaunchKernelGGL(v16<0>, dim3(tiles), dim3(512), 0, 0, dA, dB, dC, M, N, K, dClk));
  check("v16");
  SUSTAIN("v0c 32x32x2 mfma-only", hipLaunchKernelGGL(v0c<8>, dim3(tiles), dim3(512), 0, 0, dA, dB, dC, M, N, K, dClk));
  SUSTAIN("v16 16x16x4 mfma-only", hipLaunchKernelGGL(v16<1>, dim3(tiles), dim3(512), 0, 0, dA, dB, dC, M, N, K, dClk));
  RUN("l1 untuned 4w b32", 1, hipLaunchKernelGGL(l1, dim3(tiles), dim3(256), 0, 0, dA, dB, dC, M, N, K));
  RUN("l2 NT b128 swz", 1, hipLaunchKernelGGL(l2<0>, dim3(tiles), dim3(256), 0, 0, dAr, dBt, dC, M, N, K));
  RUN("l2 NT b128 swz +prefetch", 1, hipLaunchKernelGGL(l2<1>, dim3(tiles), dim3(256), 0, 0, dAr, dBt, dC, M, N, K));
}
