// igemm.h -- fp32-MFMA implicit-GEMM engine for gfx950.
//
//   C[M,N] = sum_k A(m,k) * B(k,n)
//
// One workgroup = 256 threads = 4 waves (2x2), each wave owns a (BM/2)x(BN/2)
// sub-tile as TMxTN v_mfma_f32_32x32x2_f32 accumulators (exact fp32, the
// result is an ordered fmaf chain: MI355X_MICROARCH "Matrix cores").
// The K loop stages BK=32 deep tiles through LDS in k-major order
// (As[k][m], Bs[k][n]) so an MFMA operand fragment is one conflict-free
// ds_read_b32 (32 consecutive floats per lane half); the next tile's global
// loads are issued before the current tile's 16 k-steps (64 MFMAs per wave
// for a 128x128 tile = 4096 matrix-pipe cycles) and land in registers.
//
// Operands are described by small functors ("Ops") so the same kernel body
// serves dense GEMMs (NN/NT/TN), NHWC convolution forward / dgrad / wgrad
// (im2col on the fly, never materialised) and the Conv1d weight gradients.
// Two staging modes:
//   KCONTIG  : memory is contiguous along K  (A[m][k]); a lane walks k, the
//              tile is transposed on its way into LDS (4B stores, stride
//              BX+1 -> conflict-free);
//   !KCONTIG : memory is contiguous along the M/N index (A[k][m]); 16-byte
//              global loads go straight to 16-byte LDS stores.
#pragma once
#include "common.h"

namespace igemm {

constexpr int BK = 32;
constexpr int NTHREADS = 256;

struct NoCtx {};

// ---------------------------------------------------------------- epilogues
struct EpiStore {
  float* C;
  long ldc;
  const float* bias;  // per column n, may be null
  int mode;           // 0 store, 1 C += v, 2 atomicAdd
  int cs = 1;         // column stride (elements)
  __device__ __forceinline__ void store(int m, int n, float v) const {
    float* p = C + (long)m * ldc + (long)n * cs;
    if (bias) v += bias[n];
    if (mode == 0) *p = v;
    else if (mode == 1) *p += v;
    else atomicAdd(p, v);
  }
};

// ---------------------------------------------------------------- dense operands
// element (x, k) = p[x*ld + k]
struct RowPlain {
  static constexpr bool KCONTIG = true;
  static constexpr int VEC = 1;
  typedef NoCtx Ctx;
  const float* p;
  long ld;
  int X, K, relu;
  __device__ __forceinline__ Ctx prep(int) const { return Ctx(); }
  __device__ __forceinline__ void load(const Ctx&, int x, int k0, int kin, float* v) const {
    const int k = k0 + kin;
    float t = 0.f;
    if (x < X && k < K) t = p[(long)x * ld + k];
    v[0] = relu ? fmaxf(t, 0.f) : t;
  }
};

// element (x, k) = p[b*bstride + x*ld + t + shift], k = b*seglen + t   (batch-segmented K:
// the Conv1d weight gradients contract over (sequence, time))
struct RowSegK {
  static constexpr bool KCONTIG = true;
  static constexpr int VEC = 1;
  typedef NoCtx Ctx;
  const float* p;
  long ld, bstride;
  int X, K, seglen, shift, relu, fw, dil;   // x = c*fw + tap (fw = 1: x = c); element += tap*dil
  __device__ __forceinline__ Ctx prep(int) const { return Ctx(); }
  __device__ __forceinline__ void load(const Ctx&, int x, int k0, int kin, float* v) const {
    const int k = k0 + kin;
    float t = 0.f;
    if (x < X && k < K) {
      const int b = k / seglen, tt = k - b * seglen;
      const int c = x / fw, tap = x - c * fw;
      t = p[(long)b * bstride + (long)c * ld + tt + shift + tap * dil];
    }
    v[0] = relu ? fmaxf(t, 0.f) : t;
  }
};

// element (k, x) = p[k*ld + x]
template <int V>
struct ColPlain {
  static constexpr bool KCONTIG = false;
  static constexpr int VEC = V;
  typedef NoCtx Ctx;
  const float* p;
  long ld;
  int X, K, relu;
  __device__ __forceinline__ Ctx prep(int) const { return Ctx(); }
  __device__ __forceinline__ void load(const Ctx&, int x, int k0, int kin, float* v) const {
    const int k = k0 + kin;
#pragma unroll
    for (int j = 0; j < V; ++j) v[j] = 0.f;
    if (k < K && x < X) {  // V==4: X % 4 == 0 is a launch precondition
      if (V == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p + (long)k * ld + x);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      } else {
        v[0] = p[(long)k * ld + x];
      }
    }
    if (relu) {
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] = fmaxf(v[j], 0.f);
    }
  }
};

// ---------------------------------------------------------------- the kernel
template <class Op, int BX>
struct Stage {
  static constexpr int VEC = Op::VEC;
  static constexpr int NV = BX * BK / (NTHREADS * VEC);
  static constexpr int LD = Op::KCONTIG ? BX + 1 : BX + 4;
  static constexpr int NCTX = Op::KCONTIG ? NV : 1;
  // thread -> (x_local, k_local) of its i-th vector
  __device__ static __forceinline__ void coord(int t, int i, int& xl, int& kl) {
    if (Op::KCONTIG) {
      if (VEC == 4) { kl = (t & 7) * 4; xl = (t >> 3) + 32 * i; }
      else { kl = t & 31; xl = (t >> 5) + 8 * i; }
    } else {
      if (VEC == 4) { constexpr int Q = BX / 4; xl = (t % Q) * 4; kl = t / Q + (NTHREADS / Q) * i; }
      else { xl = t % BX; kl = t / BX + (NTHREADS / BX) * i; }
    }
  }
  __device__ static __forceinline__ void to_lds(float* S, int t, const float (&st)[NV][VEC]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int xl, kl;
      coord(t, i, xl, kl);
      if (Op::KCONTIG) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) S[(kl + j) * LD + xl] = st[i][j];
      } else if (VEC == 4) {
        *reinterpret_cast<float4*>(S + kl * LD + xl) = make_float4(st[i][0], st[i][1], st[i][2], st[i][3]);
      } else {
        S[kl * LD + xl] = st[i][0];
      }
    }
  }
};

template <int BM, int BN, class AOp, class BOp, class Epi>
__global__ void __launch_bounds__(NTHREADS)
    kernel(const AOp A, const BOp B, const Epi E, const int M, const int N, const int K,
           const int ktiles_per_split) {
  typedef Stage<AOp, BM> SA;
  typedef Stage<BOp, BN> SB;
  constexpr int TM = BM / 64, TN = BN / 64;
  __shared__ __attribute__((aligned(16))) float smem[BK * SA::LD + BK * SB::LD];
  float* As = smem;
  float* Bs = smem + BK * SA::LD;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  const int ntn = (N + BN - 1) / BN;
  const int nblk = gridDim.x;
  const int bid = xcd_remap(blockIdx.x, nblk);
  const int m0 = (bid / ntn) * BM, n0 = (bid % ntn) * BN;

  const int ktiles = (K + BK - 1) / BK;
  const int kt0 = blockIdx.y * ktiles_per_split;
  const int kt1 = min(ktiles, kt0 + ktiles_per_split);
  if (kt0 >= kt1) return;

  typename AOp::Ctx actx[SA::NCTX];
  typename BOp::Ctx bctx[SB::NCTX];
#pragma unroll
  for (int i = 0; i < SA::NCTX; ++i) { int xl, kl; SA::coord(t, i, xl, kl); actx[i] = A.prep(m0 + xl); }
#pragma unroll
  for (int i = 0; i < SB::NCTX; ++i) { int xl, kl; SB::coord(t, i, xl, kl); bctx[i] = B.prep(n0 + xl); }

  float sa[SA::NV][SA::VEC], sb[SB::NV][SB::VEC];
  auto gload = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < SA::NV; ++i) { int xl, kl; SA::coord(t, i, xl, kl); A.load(actx[AOp::KCONTIG ? i : 0], m0 + xl, k0, kl, sa[i]); }
#pragma unroll
    for (int i = 0; i < SB::NV; ++i) { int xl, kl; SB::coord(t, i, xl, kl); B.load(bctx[BOp::KCONTIG ? i : 0], n0 + xl, k0, kl, sb[i]); }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  gload(kt0);
  for (int kt = kt0; kt < kt1; ++kt) {
    __syncthreads();  // everyone is done reading the previous tile
    SA::to_lds(As, t, sa);
    SB::to_lds(Bs, t, sb);
    __syncthreads();
    if (kt + 1 < kt1) gload(kt + 1);  // in flight during the 16 k-steps below
    const float* ap = As + lh * SA::LD + wm * (BM / 2) + li;
    const float* bp = Bs + lh * SB::LD + wn * (BN / 2) + li;
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = ap[ks * 2 * SA::LD + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = bp[ks * 2 * SB::LD + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
    }
  }

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (BN / 2) + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * (BM / 2) + i * 32 + mfma32_row(r, lh);
        if (m < M && n < N) E.store(m, n, acc[i][j][r]);
      }
    }
}

template <int BM, int BN, class AOp, class BOp, class Epi>
static inline int launch(const AOp& a, const BOp& b, const Epi& e, int M, int N, int K, int split_k,
                         hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0) return AVVAD_EINVAL;
  const int ktiles = (K + BK - 1) / BK;
  if (split_k < 1) split_k = 1;
  if (split_k > ktiles) split_k = ktiles;
  const int per = (ktiles + split_k - 1) / split_k;
  split_k = (ktiles + per - 1) / per;
  dim3 grid(cdiv(M, BM) * cdiv(N, BN), split_k);
  hipLaunchKernelGGL((kernel<BM, BN, AOp, BOp, Epi>), grid, dim3(NTHREADS), 0, s, a, b, e, M, N, K, per);
  AVVAD_LAUNCH_CHECK();
  return AVVAD_OK;
}

}  // namespace igemm
