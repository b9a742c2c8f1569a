// bgemm.h -- bf16 MFMA GEMM engine for operands that are bf16 IN HBM (BASELINE configs[4]: the trunk's bf16 data path).
//
//   C[M,N] (fp32) = sum_k A(m,k) * B(k,n),   A, B bf16, v_mfma_f32_32x32x16_bf16, fp32 accumulation.
//
// The fp32 engine's "bf16" mode of round 2 (igemm::kernel<..., BF = true>) converts fp32 operands on their way into LDS:
// it moves 4 bytes per element from HBM/L2, spends a v_cvt per pair and, for operands that are contiguous along M/N,
// transposes with 2-byte LDS stores.  Here the operands are stored as bf16 by whoever produced them (the BatchNorm /
// ReLU epilogue kernels, the weight pack), staged with 16-byte loads and 16-byte LDS stores as they are, and the
// transposition a weight-gradient product needs happens in the LDS READ (ds_read_b64_tr_b16).
//
// UNITS.  All operand addressing is in bf16 PAIRS (one dword): the gather functors of conv_ops.h are reused unchanged with
// the channel counts halved, their "float" loads move two bf16 each.  Two operand forms:
//   KK (forward, data gradient): both operands contiguous along K.  A K tile is 32 pairs = 64 k; LDS image x-major,
//      one row of 128 B (+16 B pad: the 16 lanes of a ds_read_b128 phase cover all 64 banks) per m / n; a fragment of
//      k-step s (16 k) is one ds_read_b128 at pair offset 8 s + 4 (lane >> 5).
//   MM (weight gradient): both operands contiguous along M / N (channels), K = output pixels.  A K tile is 32 pixels; LDS
//      image k-major, one row of BX bf16 (+64 B pad) per pixel; a fragment is two ds_read_b64_tr_b16 (4 pixels x 16
//      channels per 16-lane group, delivered channel-on-lane): row stride = 64 B mod 256 B makes the four rows of a
//      32-lane half cover all banks.
// Work decomposition, slabs and the ordered fix-up are the fp32 engine's (igemm.h): data-parallel rounds of whole tiles +
// one stream-K round, bit-reproducible.  One workgroup = 4 waves (2 x 2), each (BM/2) x (BN/2) as TM x TN 32x32 accumulators.
#pragma once
#include "conv_ops.h"

namespace bgemm {

using igemm::ClassTile;
using igemm::EpiCls;
using igemm::EpiStore;
using igemm::HasPrep2;
using igemm::HasSched;
using igemm::HasStat;
using igemm::HasTile;
using igemm::IsPlain;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int BKU = 32;                                   // K tile: 32 pairs (KK) or 32 pixels (MM)
constexpr int LDK = 36;                                   // dwords per LDS row of the KK image
__host__ __device__ constexpr int ldm(int BX) { return BX / 2 + 16; }   // dwords per LDS row of the MM image

// element (x, k pairs) = p[x * ld + k] (dwords), four consecutive pairs per 16-byte buffer fetch: the packed bf16 weights of
// the forward / data-gradient products, stored [n][K] in exactly the GEMM's K order
struct RowPairs {
  static constexpr bool KCONTIG = true;
  static constexpr int VEC = 4;
  typedef igemm::NoCtx Ctx;
  const float* p;      // (bf16 pairs behind a float pointer: the engine moves dwords)
  int ld, X, K;
  __device__ __forceinline__ Ctx prep(int) const { return Ctx(); }
  __device__ __forceinline__ void load(const Ctx&, int x, int k0, int kin, float* v) const {
    const int k = k0 + kin;
    const bool ok = x < X && k < K;
    const f4v t = bload4(brsrc2g(p), ok ? (int)(((unsigned)x * (unsigned)ld + (unsigned)k) * 4u) : BUF_OOB, 0);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  }
};
// the same packed weights read for ONE parity class of a stride-2 data gradient: K tile q -> (chunk cc, live tap seg)
// -> the 32 pairs at K position (cc * T + tap[seg]) * 32 of the full pack
struct RowPairsSeg {
  static constexpr bool KCONTIG = true;
  static constexpr int VEC = 4;
  typedef igemm::NoCtx Ctx;
  const float* p;
  int ld, X, K, T, ntap;
  int tap[4];
  unsigned mg_ntap;
  __device__ __forceinline__ Ctx prep(int) const { return Ctx(); }
  __device__ __forceinline__ void load(const Ctx&, int x, int k0, int kin, float* v) const {
    const int q = k0 >> 5, cc = convop::fast_div(q, mg_ntap), seg = (q - cc * ntap) & 3;
    const int tp = seg == 0 ? tap[0] : (seg == 1 ? tap[1] : (seg == 2 ? tap[2] : tap[3]));
    const bool ok = x < X && k0 < K;
    const f4v t = bload4(brsrc2g(p), ok ? (int)(((unsigned)x * (unsigned)ld + (unsigned)((cc * T + tp) * 32 + kin)) * 4u) : BUF_OOB, 0);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  }
};

// the packed weights for the live taps of a tile's position class (igemm.h "position classes"; conv_ops.h cls_tap)
struct RowPairsCls {
  static constexpr bool KCONTIG = true;
  static constexpr int VEC = 4;
  typedef igemm::NoCtx Ctx;
  const float* p;
  int ld, X, T, KS, flip;
  igemm::ClassRow cr;
  igemm::ClassSched sc;
  __device__ __forceinline__ Ctx prep(int) const { return Ctx(); }
  __device__ __forceinline__ ClassTile tile(int m0) const {
    int n, pp;
    cr.split(m0, n, pp);
    return igemm::class_tile(sc, pp, flip != 0);
  }
  __device__ __forceinline__ void load(const Ctx&, const ClassTile& t, int x, int k0, int kin, float* v) const {
    int cc, kh, kw;
    convop::cls_tap(t, k0, cc, kh, kw);
    const f4v q = bload4(brsrc2g(p), x < X ? (int)(((unsigned)x * (unsigned)ld + (unsigned)((cc * T + kh * KS + kw) * 32 + kin)) * 4u) : BUF_OOB, 0);
    v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
  }
};

template <int BM, int BN, bool MM, class AOp, class BOp, class Epi>
__global__ void __launch_bounds__(256, 2)
    kernel(const AOp A, const BOp B, const Epi E, const int M, const int N, const int ktiles, const int full_rounds,
           const int rem_tiles, float* __restrict__ const slab) {
  static_assert(AOp::VEC == 4 && BOp::VEC == 4, "16-byte operand fetches");
  static_assert(AOp::KCONTIG == !MM && BOp::KCONTIG == !MM, "operand form");
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int LDA = MM ? ldm(BM) : LDK, LDB = MM ? ldm(BN) : LDK;          // dwords per LDS row
  constexpr int ROWS_A = MM ? BKU : BM, ROWS_B = MM ? BKU : BN;
  constexpr int AOFF = ROWS_A * LDA, TILE = AOFF + ROWS_B * LDB;             // dwords
  constexpr int NVA = MM ? BM / 64 : BM / 32, NVB = MM ? BN / 64 : BN / 32;  // 16-byte vectors per thread and K tile
  constexpr int KS = MM ? 2 : 4;                                             // 16-deep k-steps per K tile
  __shared__ __attribute__((aligned(16))) float smem[2 * TILE];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = (N + BN - 1) / BN;

  // thread -> (x_local, k_local) of its i-th staged vector, in operand units
  auto coordA = [&](int i, int& xl, int& kl) {
    if constexpr (MM) { constexpr int Q = BM / 8; xl = (t % Q) * 4; kl = t / Q + (256 / Q) * i; }
    else { kl = (t & 7) * 4; xl = (t >> 3) + 32 * i; }
  };
  auto coordB = [&](int i, int& xl, int& kl) {
    if constexpr (MM) { constexpr int Q = BN / 8; xl = (t % Q) * 4; kl = t / Q + (256 / Q) * i; }
    else { kl = (t & 7) * 4; xl = (t >> 3) + 32 * i; }
  };

  const long G = gridDim.x;
  const long g = xcd_remap(blockIdx.x, gridDim.x);
  long rem_iters = (long)rem_tiles * ktiles;
  if constexpr (HasSched<Epi>::value) rem_iters = E.sched.total();      // position classes: every tile is in the pool
  long it = g * rem_iters / G;
  const long it_end = (g + 1) * rem_iters / G;
  const long ntiles_all = (long)((M + BM - 1) / BM) * ntn;
  int round = 0;

  for (;;) {
    int tile, kt0, kt1;
    int klen = ktiles;
    if (round < full_rounds) {
      const long tl = (long)round * G + g;
      ++round;
      if (tl >= ntiles_all) continue;
      tile = (int)tl; kt0 = 0; kt1 = ktiles;
    } else if (it < it_end) {
      if constexpr (HasSched<Epi>::value) {
        E.sched.locate(it, tile, kt0, klen);
      } else {
        const long tr = it / ktiles;
        tile = (int)((long)full_rounds * G + tr);
        kt0 = (int)(it - tr * ktiles);
      }
      kt1 = (int)min((long)klen, kt0 + (it_end - it));
      it += kt1 - kt0;
    } else {
      break;
    }
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    // operand coordinates of the tile origin, in operand units (MM: pairs along M / N)
    const int ax0 = MM ? m0 / 2 : m0, bx0 = MM ? n0 / 2 : n0;

    constexpr int NCA = MM ? 1 : NVA, NCB = MM ? 1 : NVB;
    typename AOp::Ctx actx[NCA];
    typename BOp::Ctx bctx[NCB];
#pragma unroll
    for (int i = 0; i < NCA; ++i) {
      int xl, kl;
      coordA(i, xl, kl);
      if constexpr (HasPrep2<AOp>::value) actx[i] = A.prep2(ax0 + xl, kl);
      else actx[i] = A.prep(ax0 + xl);
    }
#pragma unroll
    for (int i = 0; i < NCB; ++i) {
      int xl, kl;
      coordB(i, xl, kl);
      if constexpr (HasPrep2<BOp>::value) bctx[i] = B.prep2(bx0 + xl, kl);
      else bctx[i] = B.prep(bx0 + xl);
    }

    const typename igemm::TileOf<AOp>::type atile = igemm::TileOf<AOp>::get(A, m0);
    const typename igemm::TileOf<BOp>::type btile = igemm::TileOf<BOp>::get(B, m0);
    (void)atile; (void)btile;
    float sa[NVA][4], sb[NVB][4];
    auto gload = [&](int kt) {
      const int k0 = kt * BKU;
#pragma unroll
      for (int i = 0; i < NVA; ++i) {
        int xl, kl;
        coordA(i, xl, kl);
        if constexpr (HasTile<AOp>::value) A.load(actx[NCA > 1 ? i : 0], atile, ax0 + xl, k0, kl, sa[i]);
        else A.load(actx[NCA > 1 ? i : 0], ax0 + xl, k0, kl, sa[i]);
      }
#pragma unroll
      for (int i = 0; i < NVB; ++i) {
        int xl, kl;
        coordB(i, xl, kl);
        if constexpr (HasTile<BOp>::value) B.load(bctx[NCB > 1 ? i : 0], btile, bx0 + xl, k0, kl, sb[i]);
        else B.load(bctx[NCB > 1 ? i : 0], bx0 + xl, k0, kl, sb[i]);
      }
    };
    auto stage = [&](float* dst) {      // registers -> LDS, 16-byte stores, no conversion
#pragma unroll
      for (int i = 0; i < NVA; ++i) {
        int xl, kl;
        coordA(i, xl, kl);
        float* p = MM ? dst + kl * LDA + xl : dst + xl * LDA + kl;
        *reinterpret_cast<float4*>(p) = make_float4(sa[i][0], sa[i][1], sa[i][2], sa[i][3]);
      }
#pragma unroll
      for (int i = 0; i < NVB; ++i) {
        int xl, kl;
        coordB(i, xl, kl);
        float* p = MM ? dst + AOFF + kl * LDB + xl : dst + AOFF + xl * LDB + kl;
        *reinterpret_cast<float4*>(p) = make_float4(sb[i][0], sb[i][1], sb[i][2], sb[i][3]);
      }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    gload(kt0);
    __syncthreads();                       // the previous segment's LDS reads are done
    stage(smem);
    if (kt0 + 1 < kt1) gload(kt0 + 1);
    __syncthreads();

    for (int kt = kt0; kt < kt1; ++kt) {
      const int cur = (kt - kt0) & 1;
      const float* As = smem + cur * TILE;
      const float* Bs = As + AOFF;
      bf16x8 a[KS][TM], b[KS][TN];
      if constexpr (MM) {
        // transposed reads: 16-lane group gq = lane >> 4 takes pixels 8 (gq >> 1) + {0..3} (+4) x channels 16 (gq & 1) + {0..15};
        // lane 4 q + p of the group supplies row q, channels 4 p .. 4 p + 3 and receives channel (lane & 15)
        const int gq = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
        const int krow = 8 * (gq >> 1) + q;
        const int colp = 8 * (gq & 1) + 2 * p;                 // dword (pair) column inside a 32-channel block
        const float* ap = As + krow * LDA + wm * (BM / 4) + colp;
        const float* bp = Bs + krow * LDB + wn * (BN / 4) + colp;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ap + ks * 16 * LDA + i * 16));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ap + (ks * 16 + 4) * LDA + i * 16));
            const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            a[ks][i] = __builtin_bit_cast(bf16x8, v);
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(bp + ks * 16 * LDB + j * 16));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(bp + (ks * 16 + 4) * LDB + j * 16));
            const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            b[ks][j] = __builtin_bit_cast(bf16x8, v);
          }
        }
      } else {
        const float* ap = As + (wm * (BM / 2) + li) * LDA + lh * 4;
        const float* bp = Bs + (wn * (BN / 2) + li) * LDB + lh * 4;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
          for (int i = 0; i < TM; ++i) a[ks][i] = *reinterpret_cast<const bf16x8*>(ap + i * 32 * LDA + ks * 8);
#pragma unroll
          for (int j = 0; j < TN; ++j) b[ks][j] = *reinterpret_cast<const bf16x8*>(bp + j * 32 * LDB + ks * 8);
        }
      }
      if (kt + 1 < kt1) {                  // tile kt+1 -> the other buffer, tile kt+2 requested: under this tile's MFMAs
        stage(smem + (cur ^ 1) * TILE);
        if (kt + 2 < kt1) gload(kt + 2);
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
      __syncthreads();
    }

    int li_e = li, lh_e = lh;
    asm volatile("" : "+v"(li_e), "+v"(lh_e));
    if (kt0 > 0) {                         // a later piece of a split tile: partial sums to this worker's slab
      float* const S = slab + g * (long)(BM * BN);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = wn * (BN / 2) + j * 32 + li_e;
          const int rowb = wm * (BM / 2) + i * 32 + 4 * lh_e;
#pragma unroll
          for (int r = 0; r < 16; ++r) S[(rowb + mfma32_row(r, 0)) * BN + col] = acc[i][j][r];
        }
      continue;
    }
    // (BEFORE the output stores: the LDS exchange needs a workgroup barrier, and a barrier behind the stores would wait for
    //  every one of them to drain -- s_waitcnt vmcnt(0) -- on every tile)
    if constexpr (HasStat<Epi>::value) {   // fused BatchNorm statistics of tiles finished here (igemm.h, EpiStore::stat)
      if (E.stat != nullptr && kt0 == 0 && kt1 == klen) {
        double* red = reinterpret_cast<double*>(smem);          // [2][BN][2]
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float sf = 0.f, qf = 0.f;
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float v = acc[i][j][r]; sf += v; qf = fmaf(v, v, qf); }
          double sd = (double)sf, qd = (double)qf;
          sd += __shfl_xor(sd, 32, 64);
          qd += __shfl_xor(qd, 32, 64);
          if (lh_e == 0) {
            const int cl = wn * (BN / 2) + j * 32 + li_e;
            red[(wm * BN + cl) * 2 + 0] = sd;
            red[(wm * BN + cl) * 2 + 1] = qd;
          }
        }
        __syncthreads();
        if (t < BN && n0 + t < N) {
          const double a0 = red[t * 2 + 0] + red[(BN + t) * 2 + 0], b0 = red[t * 2 + 1] + red[(BN + t) * 2 + 1];
          const long tm = tile / ntn;
          E.stat[(tm * 2 + 0) * N + n0 + t] = a0;
          E.stat[(tm * 2 + 1) * N + n0 + t] = b0;
        }
      }
    }
    float* Cv = nullptr;                   // where the tile's rows live (position classes: the view of its grid position)
    int mrow0 = m0, Mv = M;
    if constexpr (HasSched<Epi>::value) { const auto v = E.view(m0); Cv = v.C; mrow0 = v.row0; Mv = v.M; }
    else if constexpr (IsPlain<Epi>::value) Cv = E.C;
    bool fast = false;
    if constexpr (IsPlain<Epi>::value) fast = E.cs == 1 && (long)Mv * E.ldc < (1L << 31) && mrow0 + BM <= Mv && n0 + BN <= N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 32 + li_e;
        const int mb = mrow0 + wm * (BM / 2) + i * 32;
        if constexpr (IsPlain<Epi>::value) {
          if (fast) {
            const unsigned ld32 = (unsigned)E.ldc;
            const unsigned base = (unsigned)(mb + 4 * lh_e) * ld32 + (unsigned)n;
            float* const Cb = Cv;
            if (E.mode == 0) {
#pragma unroll
              for (int r = 0; r < 16; ++r) Cb[base + (unsigned)mfma32_row(r, 0) * ld32] = acc[i][j][r];
            } else {
              float old[16];
#pragma unroll
              for (int r = 0; r < 16; ++r) old[r] = Cb[base + (unsigned)mfma32_row(r, 0) * ld32];
#pragma unroll
              for (int r = 0; r < 16; ++r) Cb[base + (unsigned)mfma32_row(r, 0) * ld32] = old[r] + acc[i][j][r];
            }
            continue;
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + mfma32_row(r, lh_e);
          if (m < Mv && n < N) {
            float* p;
            if constexpr (HasSched<Epi>::value) p = Cv + (long)m * E.ldc + n;
            else p = E.ptr(m, n);
            if (E.mode == 0) *p = acc[i][j][r];
            else *p += acc[i][j][r];
          }
        }
      }
  }
}

// K in operand units (KK: pairs, a multiple of 32 per tap chunk; MM: pixels).  slab: igemm::SLAB_FLOATS of scratch or null.
template <int BM, int BN, bool MM, class AOp, class BOp, class Epi>
static inline int launch(const AOp& a, const BOp& b, const Epi& e, int M, int N, int K, hipStream_t s, float* slab) {
  if (M <= 0 || N <= 0 || K <= 0) return AVVAD_EINVAL;
  const int ktiles = cdiv(K, BKU);
  const long ntiles = (long)cdiv(M, BM) * cdiv(N, BN);
  const AvvadTune& tn = avvad_tune();
  const int cus = (tn.max_cus > 0 && tn.max_cus < igemm::NUM_CU) ? tn.max_cus : igemm::NUM_CU;
  long G = (long)cus * 2;                                    // two 4-wave workgroups per CU (LDS: 2 x 72 KB at 128x128 KK)
  if (G * BM * BN > (long)igemm::SLAB_FLOATS) G = (long)(igemm::SLAB_FLOATS / ((size_t)BM * BN));
  const bool no_sk = !slab || tn.no_streamk == 1 || tn.no_streamk == 10 + e.mode || (double)ntiles * ktiles * (double)G >= 4.0e9;
  long full_rounds = ntiles / G, rem = ntiles - full_rounds * G;
  if (no_sk || rem * 10 >= G * 9) {
    if (rem > 0) ++full_rounds;
    rem = 0;
  }
  if (rem > 0 && full_rounds == 0) {
    const long iters = ntiles * ktiles;
    const long cap = iters / 4 > 0 ? iters / 4 : 1;
    if (G > cap) G = cap;
  }
  bool stat = false;
  if constexpr (std::is_same<Epi, EpiStore>::value) {
    stat = e.stat != nullptr;
    if (stat && (e.cs != 1 || (e.ldc & 3) || (N & 3) || (((uintptr_t)e.C) & 15) || e.bias != nullptr)) return AVVAD_EINVAL;
  }
  if (e.bias != nullptr) return AVVAD_EINVAL;               // (no bias in this engine's epilogue: the convolutions have none)
  const int fr = (int)full_rounds, rt = (int)rem;
  hipLaunchKernelGGL((kernel<BM, BN, MM, AOp, BOp, Epi>), dim3((int)G), dim3(256), 0, s, a, b, e, M, N, ktiles, fr, rt, slab);
  if (rt > 0 && stat) {
    if constexpr (std::is_same<Epi, EpiStore>::value)
      hipLaunchKernelGGL((igemm::fixup_tile<BM, BN>), dim3(rt), dim3(256), 0, s, e, slab, M, N, ktiles, G, fr, rt, cdiv(N, BN));
  } else if (rt > 0) {
    if ((G + rt - 1) / rt <= 6)
      hipLaunchKernelGGL((igemm::fixup1<BM, BN, Epi>), dim3(rt * (BM * BN / 1024)), dim3(256), 0, s, e, slab, M, N, ktiles, G, fr, rt, cdiv(N, BN));
    else
      hipLaunchKernelGGL((igemm::fixup<BM, BN, Epi>), dim3(rt * (BM * BN / 256)), dim3(256), 0, s, e, slab, M, N, ktiles, G, fr, rt, cdiv(N, BN));
  }
  AVVAD_LAUNCH_CHECK();
  return AVVAD_OK;
}

// position-class product on the bf16 engine (igemm::launch_cls is the fp32 twin): 128x128 KK tiles
template <bool MM, class AOp, class BOp, class Epi>
static inline int launch_cls(const AOp& a, const BOp& b, const Epi& e, int Mp, int N, hipStream_t s, float* slab) {
  constexpr int BM = 128, BN = 128;
  if (Mp <= 0 || N <= 0 || !slab) return AVVAD_EINVAL;
  const AvvadTune& tn = avvad_tune();
  const long ntiles = (long)cdiv(Mp, BM) * cdiv(N, BN);
  const int cus = (tn.max_cus > 0 && tn.max_cus < igemm::NUM_CU) ? tn.max_cus : igemm::NUM_CU;
  long G = (long)cus * 2;
  if (G * BM * BN > (long)igemm::SLAB_FLOATS) G = (long)(igemm::SLAB_FLOATS / ((size_t)BM * BN));
  const long R = e.sched.total();
  if (R <= 0 || (double)R * (double)G >= 4.0e9) return AVVAD_EINVAL;
  if (G > R / 4) G = R / 4 > 0 ? R / 4 : 1;
  bool stat = false;
  if constexpr (HasStat<Epi>::value) {
    stat = e.stat != nullptr;
    if (stat && (e.cs != 1 || (e.ldc & 3) || (N & 3) || (((uintptr_t)e.C) & 15) || (e.W & 3))) return AVVAD_EINVAL;
  }
  hipLaunchKernelGGL((kernel<BM, BN, MM, AOp, BOp, Epi>), dim3((int)G), dim3(256), 0, s, a, b, e, Mp, N, 1, 0, (int)ntiles, slab);
  if (stat) {
    if constexpr (HasStat<Epi>::value)
      hipLaunchKernelGGL((igemm::fixup_tile<BM, BN, Epi>), dim3((int)ntiles), dim3(256), 0, s, e, slab, Mp, N, 1, G, 0, (int)ntiles, cdiv(N, BN));
  } else if ((G + ntiles - 1) / ntiles <= 6) {
    hipLaunchKernelGGL((igemm::fixup1<BM, BN, Epi>), dim3((int)ntiles * (BM * BN / 1024)), dim3(256), 0, s, e, slab, Mp, N, 1, G, 0,
                       (int)ntiles, cdiv(N, BN));
  } else {
    hipLaunchKernelGGL((igemm::fixup<BM, BN, Epi>), dim3((int)ntiles * (BM * BN / 256)), dim3(256), 0, s, e, slab, Mp, N, 1, G, 0,
                       (int)ntiles, cdiv(N, BN));
  }
  AVVAD_LAUNCH_CHECK();
  return AVVAD_OK;
}

}  // namespace bgemm
