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
#include <type_traits>

namespace igemm {

constexpr int BK = 32;

struct NoCtx {};

// ---------------------------------------------------------------- epilogues
// An epilogue names WHERE element (m, n) of the product lives (ptr) and HOW it lands there: mode 0 overwrite, mode 1/2
// accumulate (C += v; 2 is the historical "K-split accumulate" spelling and means the same).  The kernel performs the
// store itself.  No epilogue adds atomically any more: a tile whose K range is split between workers is combined by the
// fix-up kernel below in a fixed order (bit-reproducible), the K-major tuning mode being the one exception.
struct EpiStore {
  static constexpr bool PLAIN = true;
  float* C;
  long ldc;
  const float* bias;  // per column n, may be null (added by the kernel: once per column, before the row loop)
  int mode;           // 0 store, 1 / 2 C += v
  int cs = 1;         // column stride (elements)
  // Column statistics of the finished product (BatchNorm's batch statistics, fused): stat[(tm * 2 + 0) * N + n] = sum over the
  // rows of M tile tm of C[m][n], [(tm * 2 + 1) * N + n] = sum of squares, as doubles; every (tm, n) is written exactly once
  // per launch -- by the kernel for tiles it finishes itself, by fixup_tile for tiles cut along K.  Needs mode 0 / 1 on a
  // row-major output with cs == 1, ldc % 4 == 0, N % 4 == 0 (checked by launch()).  null: no statistics.
  double* stat = nullptr;
  __device__ __forceinline__ float* ptr(int m, int n) const { return C + (long)m * ldc + (long)n * cs; }
};
template <class Epi, class = void>
struct HasStat : std::false_type {};
template <class Epi>
struct HasStat<Epi, std::void_t<decltype(Epi::stat)>> : std::true_type {};

// ---------------------------------------------------------------- position classes: skipping the zero padding
// A 3x3 / pad 1 convolution multiplies structural zeros wherever a tap falls into the padding: 7 % of all products on a 17x17
// grid, 14 % on 9x9, 25 % on 5x5 and 40 % on the 3x3 grid of the last stage (a corner pixel has 4 live taps of 9) -- on the
// ResNet-18 trunk at 67x67 crops about a quarter of all convolution FLOPs.  They can be skipped, exactly (0 * w adds nothing),
// when a tile's rows all sit at the SAME grid position: the GEMM's rows are therefore ordered position-major, m' = p * NP + n
// (p = ho * Wo + wo, n = image, NP = images padded to a whole number of M tiles), so that the set of live taps is a property
// of the TILE -- a(ho) x b(wo) taps, contiguous ranges kh_lo .. / kw_lo .. -- and a tile's K loop simply has fewer K tiles.
// Tiles then differ in length: the stream-K round cuts the concatenation of all tiles' K tiles into equal shares, with the
// tile <-> iteration maps in closed form (the live-tap counts are separable: a(ho) = KS - [ho == 0] fs - [ho == Ho-1] ls).
// Used where the tile count does not exceed the worker count (no data-parallel rounds: every tile is in the pool).
struct ClassSched {
  // Tile order: (image block mt, grid position p, N tile nt), nt fastest -- the 128 images of a block are walked through ALL
  // their positions by neighbouring workers (one XCD after the remap), so the taps' overlapping pixel reads stay in that XCD's
  // L2 (position-outermost order re-fetched every image once per position from beyond L2: 427 MB per launch, PMC).
  int Ho, Wo, ntn, nchunk, MB;    // grid of position classes; N tiles; channel chunks per tap; image blocks (of 128)
  int KS, fsh, lsh, fsw, lsw;     // first / last row (column) of the grid loses fsh / lsh (fsw / lsw) taps
  // s2: the data gradient of a 3x3 / STRIDE 2 / pad 1 convolution.  The classes are the INPUT positions (h, w) of a grid
  // Ho x Wo = H x W whose gradient image dy is Hq x Wq: row h is reached by the taps kh with h + 1 - kh even and
  // (h + 1 - kh) / 2 < Hq -- kh = 1 for even h, kh = 0 and 2 for odd h (kh = 0 only while (h + 1) / 2 < Hq): 1, 2, 1, 2, ...
  // live tap rows instead of 3, i.e. the four parity classes of the decomposition with 1 / 2 / 2 / 4 taps, in ONE product.
  int s2 = 0, Hq = 0, Wq = 0;
  __host__ __device__ static int odd2(int h, int q) { return ((h & 1) && ((h + 1) >> 1) < q) ? 1 : 0; }
  __host__ __device__ static int cnt2(int h, int q) { const int m = h >> 1; return h + (m < q - 1 ? m : q - 1); }   // sum of (1 + odd2) below h
  __host__ __device__ int ah(int h) const { return s2 ? 1 + odd2(h, Hq) : KS - (h == 0 ? fsh : 0) - (h == Ho - 1 ? lsh : 0); }
  __host__ __device__ int bw(int w) const { return s2 ? 1 + odd2(w, Wq) : KS - (w == 0 ? fsw : 0) - (w == Wo - 1 ? lsw : 0); }
  __host__ __device__ int Ah(int h) const { return s2 ? cnt2(h, Hq) : KS * h - (h > 0 ? fsh : 0) - (h > Ho - 1 ? lsh : 0); }   // sum of ah below h
  __host__ __device__ int Bw(int w) const { return s2 ? cnt2(w, Wq) : KS * w - (w > 0 ? fsw : 0) - (w > Wo - 1 ? lsw : 0); }
  __host__ __device__ long per_block() const { return (long)nchunk * ntn * Ah(Ho) * Bw(Wo); }
  __host__ __device__ long total() const { return per_block() * MB; }
  __host__ __device__ int len(int tile) const {
    const int p = (tile / ntn) % (Ho * Wo), ho = p / Wo, wo = p - ho * Wo;
    return ah(ho) * bw(wo) * nchunk;
  }
  __host__ __device__ long base(int tile) const {      // K tiles of all tiles in front of `tile`
    const int q = tile / ntn, nt = tile - q * ntn, P = Ho * Wo, mt = q / P, p = q - mt * P, ho = p / Wo, wo = p - ho * Wo;
    return per_block() * mt + (long)nchunk * ((long)ntn * (Ah(ho) * Bw(Wo) + ah(ho) * Bw(wo)) + (long)nt * ah(ho) * bw(wo));
  }
  __device__ void locate(long it, int& tile, int& kt, int& klen) const {
    const long pb = per_block();
    const int mt = (int)(it / pb);
    long rem = it - pb * mt;
    const long c = (long)nchunk * ntn;
    const int Bt = Bw(Wo);
    int ho = 0;
    while (ho + 1 < Ho && c * Ah(ho + 1) * Bt <= rem) ++ho;
    rem -= c * Ah(ho) * Bt;
    const int a = ah(ho);
    int wo = 0;
    while (wo + 1 < Wo && c * a * Bw(wo + 1) <= rem) ++wo;
    rem -= c * a * Bw(wo);
    klen = a * bw(wo) * nchunk;
    const int nt = (int)(rem / klen);
    kt = (int)(rem - (long)nt * klen);
    tile = ((mt * Ho + ho) * Wo + wo) * ntn + nt;
  }
};
// GEMM row m' of a position-class product -> (image n, grid position p):  m' = (mt * P + p) * 128 + r,  n = mt * 128 + r
struct ClassRow {
  int P;
  unsigned mg_P;
  __device__ __forceinline__ void split(int m, int& n, int& p) const {
    const int q = m >> 7, mt = mg_P ? (int)__umulhi((unsigned)q, mg_P) : q;
    p = q - mt * P;
    n = mt * 128 + (m & 127);
  }
};
// what a tile's K loop needs to know about its class: live tap ranges and counts (wave-uniform: kept in scalar registers)
struct ClassTile {
  int ho, wo, a, b, kh_lo, kw_lo;
  unsigned mg_ab, mg_b;           // exact-division magics for a * b and b
  int step;                       // live taps are kh_lo, kh_lo + step, ...: 1, or 2 for the stride-2 data gradient
};
__device__ __forceinline__ unsigned dev_magic(int d) { return d <= 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)d) + 1u; }
__device__ __forceinline__ int dev_div(int k, unsigned mg) { return mg ? (int)__umulhi((unsigned)k, mg) : k; }
// flip: the data gradient walks the taps the other way round (pixel = position + pad - tap): the first row loses the LAST tap
__device__ __forceinline__ ClassTile class_tile(const ClassSched& s, int p, bool flip) {
  ClassTile t;
  t.ho = p / s.Wo;
  t.wo = p - t.ho * s.Wo;
  t.a = s.ah(t.ho);
  t.b = s.bw(t.wo);
  const int cut_h = flip ? (t.ho == s.Ho - 1 ? s.lsh : 0) : (t.ho == 0 ? s.fsh : 0);
  const int cut_w = flip ? (t.wo == s.Wo - 1 ? s.lsw : 0) : (t.wo == 0 ? s.fsw : 0);
  t.kh_lo = cut_h;
  t.kw_lo = cut_w;
  t.step = 1;
  if (s.s2) {      // even position: the middle tap; odd: taps 0 and 2 (2 alone in a last row that tap 0 would take past dy)
    t.kh_lo = (t.ho & 1) ? (t.a == 2 ? 0 : 2) : 1;
    t.kw_lo = (t.wo & 1) ? (t.b == 2 ? 0 : 2) : 1;
    t.step = 2;
  }
  t.mg_ab = dev_magic(t.a * t.b);
  t.mg_b = dev_magic(t.b);
  return t;
}
// Epilogue of a position-class product: GEMM row m' = p * NP + n is row n of the [rows][ldc] view that starts at column
// block p * W of the output (NHWC: pixel (n, p) -> C + (n * P + p) * W).  Carries the schedule (the fix-up kernels need it).
struct EpiCls {
  static constexpr bool PLAIN = true;
  float* C;
  long ldc;            // P * W: elements between consecutive images at one grid position
  const float* bias;   // unused (null)
  int mode;
  int cs = 1;
  double* stat = nullptr;
  int W = 0;           // channels of the output (elements per pixel)
  int rows = 0;        // images
  ClassRow cr{0, 0u};
  ClassSched sched;
  struct View { float* C; int row0, M; };
  __device__ __forceinline__ View view(int m0) const {      // m0: a tile origin (multiple of 128)
    int n, p;
    cr.split(m0, n, p);
    return View{C + (long)p * W, n, rows};
  }
  __device__ __forceinline__ bool live(int m) const {
    int n, p;
    cr.split(m, n, p);
    return n < rows;
  }
  __device__ __forceinline__ float* ptr(int m, int nn) const {
    int n, p;
    cr.split(m, n, p);
    return C + (long)p * W + (long)n * ldc + nn;
  }
};
// The WEIGHT gradient's twin: C[(tap, c)][co] = sum over output pixels.  Rows are tap-major (a 128-row tile = one tap x 128
// channels = rows of the packed gradient as they are), K runs over the pixels in the same position-major order k' = p * NP + n,
// and a tile's K loop walks only the grid positions at which ITS tap falls inside the image: nh(kh) x nw(kw) positions of
// NP / 32 K tiles each (tap (0, 0) of a 3x3 grid sees 4 of the 9 positions).
struct TapSched {
  int Ho, Wo, tpt, kpp;           // tiles per tap (channel tiles x N tiles); K tiles per grid position (NP / 32)
  int KS, fsh, lsh, fsw, lsw;     // tap row kh = 0 misses the first fsh grid rows, kh = KS - 1 the last lsh (same for columns)
  __host__ __device__ int nh(int kh) const { return Ho - (kh == 0 ? fsh : 0) - (kh == KS - 1 ? lsh : 0); }
  __host__ __device__ int nw(int kw) const { return Wo - (kw == 0 ? fsw : 0) - (kw == KS - 1 ? lsw : 0); }
  __host__ __device__ int len_tap(int tap) const { const int kh = tap / KS; return nh(kh) * nw(tap - kh * KS) * kpp; }
  __host__ __device__ int len(int tile) const { return len_tap(tile / tpt); }
  __host__ __device__ long base(int tile) const {
    const int tap = tile / tpt;
    long b = 0;
    for (int q = 0; q < tap; ++q) b += (long)len_tap(q) * tpt;
    return b + (long)(tile - tap * tpt) * len_tap(tap);
  }
  __host__ __device__ long total() const { return base(KS * KS * tpt); }
  __device__ void locate(long it, int& tile, int& kt, int& klen) const {
    int tap = 0;
    long b = 0;
    for (; tap + 1 < KS * KS; ++tap) {
      const long l = (long)len_tap(tap) * tpt;
      if (it < b + l) break;
      b += l;
    }
    klen = len_tap(tap);
    const int r = (int)((it - b) / klen);
    kt = (int)(it - b - (long)r * klen);
    tile = tap * tpt + r;
  }
};
struct TapTile {
  int kh, kw, h_lo, w_lo, rowlen, kpp;    // rowlen: K tiles per live grid row = nw(kw) * kpp
  unsigned mg_rowlen, mg_kpp;
};
__device__ __forceinline__ TapTile tap_tile(const TapSched& s, int tap) {
  TapTile t;
  t.kh = tap / s.KS;
  t.kw = tap - t.kh * s.KS;
  t.h_lo = t.kh == 0 ? s.fsh : 0;
  t.w_lo = t.kw == 0 ? s.fsw : 0;
  t.kpp = s.kpp;
  t.rowlen = s.nw(t.kw) * s.kpp;
  t.mg_rowlen = dev_magic(t.rowlen);
  t.mg_kpp = dev_magic(s.kpp);
  return t;
}
// K tile q of a tap tile -> grid position (ho, wo) and first image n0 of its 32 pixels (all scalar)
__device__ __forceinline__ void tap_pos(const TapTile& t, int k0, int& ho, int& wo, int& n0) {
  const int q = k0 >> 5, j = dev_div(q, t.mg_rowlen), r = q - j * t.rowlen, pw = dev_div(r, t.mg_kpp);
  ho = t.h_lo + j;
  wo = t.w_lo + pw;
  n0 = (r - pw * t.kpp) * 32;
}
struct EpiTap {      // plain [M][ldc] output (the packed weight gradient) + the schedule
  static constexpr bool PLAIN = true;
  float* C;
  long ldc;
  const float* bias;   // unused (null)
  int mode;
  int cs = 1;
  int Mrows = 0;
  TapSched sched;
  struct View { float* C; int row0, M; };
  __device__ __forceinline__ View view(int m0) const { return View{C, m0, Mrows}; }
  __device__ __forceinline__ bool live(int) const { return true; }
  __device__ __forceinline__ float* ptr(int m, int n) const { return C + (long)m * ldc + n; }
};
// what a functor's tile() hook returns (nothing for ordinary operands)
struct NoTile {};
template <class Op, class = void>
struct TileOf {
  typedef NoTile type;
  __device__ static __forceinline__ type get(const Op&, int) { return NoTile(); }
};
template <class Op>
struct TileOf<Op, std::void_t<decltype(&Op::tile)>> {
  typedef decltype(std::declval<const Op&>().tile(0)) type;
  __device__ static __forceinline__ type get(const Op& o, int m0) { return o.tile(m0); }
};
template <class Epi, class = void>
struct HasSched : std::false_type {};
template <class Epi>
struct HasSched<Epi, std::void_t<decltype(Epi::sched)>> : std::true_type {};
template <class Op, class = void>
struct HasTile : std::false_type {};
template <class Op>
struct HasTile<Op, std::void_t<decltype(&Op::tile)>> : std::true_type {};
template <class Epi>
__device__ __forceinline__ bool epi_live(const Epi& E, int m, int M) {
  if constexpr (HasSched<Epi>::value) return m < M && E.live(m);
  else return m < M;
}

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
    if (x < X && k < K) t = p[(long)x * ld + k];   // scalar guarded loads: cheaper here than 16 clamped 64-bit offsets
    v[0] = t;
  }
  __device__ __forceinline__ float post(float v) const { return relu ? fmaxf(v, 0.f) : v; }
};

// element (x, k) = p[x*ld + k], four consecutive k per 16-byte buffer fetch (K % 4 == 0, ld % 4 == 0, 16-byte aligned base,
// operand < 2 GiB: checked by the caller).  The scalar RowPlain above issues 8 guarded one-dword loads per thread and K tile
// for a 64-row tile; this one issues 2 and needs no validity select (out-of-range rows carry BUF_OOB and read as 0).
struct RowVec4 {
  static constexpr bool KCONTIG = true;
  static constexpr int VEC = 4;
  typedef NoCtx Ctx;
  const float* p;
  long ld;
  int X, K, relu;
  __device__ __forceinline__ Ctx prep(int) const { return Ctx(); }
  __device__ __forceinline__ void load(const Ctx&, int x, int k0, int kin, float* v) const {
    const int k = k0 + kin;
    const bool ok = x < X && k < K;
    const f4v t = bload4(brsrc2g(p), ok ? (int)(((unsigned)x * (unsigned)ld + (unsigned)k) * 4u) : BUF_OOB, 0);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  }
  __device__ __forceinline__ float post(float v) const { return relu ? fmaxf(v, 0.f) : v; }
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
    v[0] = t;
  }
  __device__ __forceinline__ float post(float v) const { return relu ? fmaxf(v, 0.f) : v; }
};

// element (k, x) = p[k*ld + x]      (BUF: buffer-addressed fetch, see conv_ops.h -- the operand is < 2 GiB, V == 4)
template <int V, bool BUF = false>
struct ColPlain {
  static constexpr bool KCONTIG = false;
  static constexpr int VEC = V;
  typedef NoCtx Ctx;
  const float* p;
  long ld;
  int X, K, relu;
  __device__ __forceinline__ Ctx prep(int) const { return Ctx(); }
  __device__ __forceinline__ typename std::conditional<BUF, void, bool>::type load(const Ctx&, int x, int k0, int kin, float* v) const {
    const int k = k0 + kin;
    const bool ok = k < K && x < X;  // V==4: X % 4 == 0 and 16-byte aligned rows are launch preconditions
    if constexpr (BUF) {
      static_assert(V == 4, "buffer form: 16-byte fetches");
      const f4v t = bload4(brsrc2g(p), ok ? (int)(((unsigned)k * (unsigned)ld + (unsigned)x) * 4u) : BUF_OOB, 0);
      v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    } else {
      const long off = ok ? (long)k * ld + x : 0;
      if (V == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p + off);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      } else {
        v[0] = p[off];
      }
      return ok;
    }
  }
  __device__ __forceinline__ float post(float v) const { return relu ? fmaxf(v, 0.f) : v; }
};

#ifdef AVVAD_PROF
// tuning aid (never in the shipped build): per-phase shader-clock totals of wave 0 of every workgroup
__device__ unsigned long long g_prof[8];
__device__ unsigned long long g_prof_blk[2 * 2048];   // per block: s_memrealtime (100 MHz) at start / end of the LAST launch
__device__ unsigned long long g_prof_clk[2048];       // per block: shader cycles (s_memtime) spent in the kernel
__device__ unsigned long long g_prof_hw[2048];        // per block: XCC_ID << 32 | HW_ID (which CU / wave slot it landed on)   // 0 segments, 1 prologue, 2 loop, 3 barrier wait, 4 staging, 5 epilogue, 6 ktile iterations
#define PROF_T() ((wave == 0) ? __builtin_amdgcn_s_memtime() : 0ull)
#define PROF_ADD(i, v) do { if (t == 0) atomicAdd(&g_prof[i], (unsigned long long)(v)); } while (0)
#else
#define PROF_T() 0ull
#define PROF_ADD(i, v) do { } while (0)
#endif

// OPERAND CONTRACT.  load() issues its global load UNCONDITIONALLY (from offset 0 of the operand when the element is
// out of range or padding) and returns whether the element is real; it does no arithmetic on the loaded value.  The
// zero-fill of invalid elements and an operand's optional elementwise map (post(): the fused ReLU of the dense
// operands) happen when the staged registers are written to LDS.  Anything that touches a just-loaded value inside
// load() -- a select, an fmaxf, even a register shuffle behind a guarded load -- makes hipcc wait for the load on the
// spot (s_waitcnt vmcnt(0) behind each global_load), which serialises the prefetch and exposes its full latency in the
// middle of the MFMA stream.  (Older functors whose load() returns void zero their own values; they still work.)
template <class Op, class = void>
struct HasPost : std::false_type {};
template <class Op>
struct HasPost<Op, std::void_t<decltype(&Op::post)>> : std::true_type {};

// An operand functor may ask for one context PER STAGED VECTOR (static constexpr bool PERVEC = true) and build it with
// prep2(x, k_local): everything about a vector's address that does not change from K tile to K tile -- its row / column,
// its k position inside the tile -- is then folded into the context once per tile segment, and load() is left with one
// add of a wave-uniform (SALU-computed) offset.  On this chip VALU time adds to fp32-MFMA time (the matrix instruction
// runs on the vector lanes): the per-K-tile address arithmetic of the im2col gathers was ~17 % of the convolution time.
template <class Op, class = void>
struct HasPrep2 : std::false_type {};
template <class Op>
struct HasPrep2<Op, std::void_t<decltype(&Op::prep2)>> : std::true_type {};
template <class Op, class = void>
struct PerVec : std::false_type {};
template <class Op>
struct PerVec<Op, std::void_t<decltype(Op::PERVEC)>> : std::integral_constant<bool, Op::PERVEC> {};

// An epilogue may FINISH elements: `finish4(m, n0, v, N)` receives the final sums of up to four consecutive columns of row m
// from the fix-up kernel (the one place where a split tile's total exists) and does whatever follows the product -- the
// LSTM backward runs the previous step's gate arithmetic there, one launch less per time step.  It is only honoured when
// EVERY tile of the product passes through the stream-K pool (launch() tells the caller through `finished`); `active` is
// set by launch().
// epilogues whose output is a plain row-major matrix (C, ldc, cs, mode) take the kernel's fast store path
template <class Epi, class = void>
struct IsPlain : std::false_type {};
template <class Epi>
struct IsPlain<Epi, std::void_t<decltype(Epi::PLAIN)>> : std::integral_constant<bool, Epi::PLAIN> {};
template <class Epi, class = void>
struct HasFinish : std::false_type {};
template <class Epi>
struct HasFinish<Epi, std::void_t<decltype(&Epi::finish4)>> : std::true_type {};

// ---------------------------------------------------------------- the kernel
// bf16 operands (option "bf16", BASELINE config 5): the fp32 values are rounded to bf16 (RNE) on their way into LDS and
// multiplied by v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- 16x the matrix rate of the fp32-input instruction.  That
// instruction wants 8 CONSECUTIVE k per lane (lane half h: k = 8h .. 8h+7 of a 16-deep step), so the bf16 LDS image is
// x-major: one row per m (or n) of BK = 32 bf16 = 64 bytes, padded to 80 bytes (20 words: the 16 lanes of a ds_read_b128
// phase then cover all 64 banks exactly once).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int LDH = 40;   // halfwords per LDS row of the bf16 image

template <class Op, int BX, int NTH>
struct Stage {
  static constexpr int VEC = Op::VEC;
  static constexpr int NV = BX * BK / (NTH * VEC);
  static constexpr int LD = Op::KCONTIG ? BX + 1 : BX + 4;
  static constexpr int NCTX = (Op::KCONTIG || PerVec<Op>::value) ? NV : 1;
  static constexpr int TILE_F32 = BK * LD;                 // floats of the fp32 (k-major) image
  static constexpr int TILE_BF = BX * LDH / 2;             // floats' worth of the bf16 (x-major) image
  // thread -> (x_local, k_local) of its i-th vector
  __device__ static __forceinline__ void coord(int t, int i, int& xl, int& kl) {
    if (Op::KCONTIG) {
      if (VEC == 4) { kl = (t & 7) * 4; xl = (t >> 3) + (NTH / 8) * i; }
      else { kl = t & 31; xl = (t >> 5) + (NTH / 32) * i; }
    } else {
      if (VEC == 4) { constexpr int Q = BX / 4; xl = (t % Q) * 4; kl = t / Q + (NTH / Q) * i; }
      else { xl = t % BX; kl = t / BX + (NTH / BX) * i; }
    }
  }
  __device__ static __forceinline__ void to_lds(float* S, int t, const float (&st0)[NV][VEC], const Op& op, unsigned okmask) {
    float st[NV][VEC];
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float v = st0[i][j];
        if constexpr (HasPost<Op>::value) v = op.post(v);
#ifdef AVVAD_ABL_NOSEL
        st[i][j] = v;
#else
        st[i][j] = ((okmask >> i) & 1u) ? v : 0.f;
#endif
      }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int xl, kl;
      coord(t, i, xl, kl);
      if constexpr (Op::KCONTIG) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) S[(kl + j) * LD + xl] = st[i][j];
      } else if constexpr (VEC == 4) {
        *reinterpret_cast<float4*>(S + kl * LD + xl) = make_float4(st[i][0], st[i][1], st[i][2], st[i][3]);
      } else {
        S[kl * LD + xl] = st[i][0];
      }
    }
  }
  // the bf16 image: S16[x][k]
  __device__ static __forceinline__ void to_lds_bf(__bf16* S, int t, const float (&st0)[NV][VEC], const Op& op, unsigned okmask) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int xl, kl;
      coord(t, i, xl, kl);
      float st[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float v = st0[i][j];
        if constexpr (HasPost<Op>::value) v = op.post(v);
        st[j] = ((okmask >> i) & 1u) ? v : 0.f;
      }
      if (Op::KCONTIG) {
        if constexpr (VEC == 4) {                // four consecutive k of one row: one 8-byte store
          const f32x4v f = {st[0], st[1], st[2], st[3]};
          *reinterpret_cast<bf16x4*>(S + xl * LDH + kl) = __builtin_convertvector(f, bf16x4);
        } else {
          S[xl * LDH + kl] = (__bf16)st[0];
        }
      } else {                                   // consecutive x at one k: the transposition costs 2-byte stores
#pragma unroll
        for (int j = 0; j < VEC; ++j) S[(xl + j) * LDH + kl] = (__bf16)st[j];
      }
    }
  }
};

// Work decomposition: data-parallel rounds + one "stream-K" round.  The launch creates G persistent workgroups
// (= the number resident at once: 256 CUs x workgroups/CU).  Worker g first takes whole tiles g, g+G, ... for
// `full_rounds` rounds (plain stores, all workers walk K in step -> weight panels are shared in L2); the
// remaining rem_tiles < G tiles are then treated as ONE pool of rem_tiles x ktiles BK-deep iterations cut into G
// equal contiguous shares: a tile may be split between workers, and split pieces are added with float atomics onto
// a pre-zeroed (or accumulating) output.  That removes the tile-quantisation loss of e.g. 648 or 288 tiles on 512
// resident slots (37 % / 44 % idle) while only the last round pays the ~1.3 TB/s float-atomic rate.
// Measured balance of this static split (tools/lab/prof_conv.py, -DAVVAD_PROF): all G workers start within 0.5 us; the
// two workgroups of a CU are raw block ids b and b+256 and the OLDER always finishes first (192 vs 237 us: the matrix
// pipe is arbitrated oldest-first); XCDs finish up to 17 % apart on equal shares (the same two are slowest for every
// shape on a given device).  Tried and dropped: age-skewed shares (+-1 %), a dynamically pulled tail pool of 8-K-tile
// pieces (1-5 % slower: every piece pays a prologue and a 64 KB atomic flush), K-major cells for the weight gradients
// (less HBM traffic, 2 % slower end to end), staggering the co-residents' phases (no effect), requesting the NEXT
// segment's first K tile before the epilogue (the prefetched registers live across the epilogue: past the 128-VGPR
// budget of 4 waves/SIMD the 8-wave kernels spill, 5 % slower end to end).
// NTH = 256: 4 waves as 2x2, each (BM/2)x(BN/2);  NTH = 512: 8 waves as 2x4, each (BM/2)x(BN/4) -- half the
// accumulators and staging registers per wave, so twice the waves per SIMD fit next to the same LDS tile.
template <int BM, int BN, bool DB, int NTH, class AOp, class BOp, class Epi, bool BF = false>
// (HIP's second launch-bounds argument is WAVES PER SIMD, not blocks per CU.  The 8-wave instantiation is pinned to
//  4 waves/SIMD = the 2 workgroups/CU launch() assumes: a build drifting past 128 VGPRs would otherwise silently halve
//  the residency -- the instrumented -DAVVAD_PROF build did exactly that.  The 4-wave instantiations are left looser:
//  pinning them to their per_cu made the 64x64 kernels spill.)
__global__ void __launch_bounds__(NTH, NTH == 512 ? 4 : ((DB || BM * BN < 128 * 128) ? 2 : 3))
    kernel(const AOp A, const BOp B, const Epi E, const int M, const int N, const int flags, const int ktiles,
           const int full_rounds, const int rem_tiles, const int kchunks, float* __restrict__ const slab) {
  const bool stagger = (flags & 1) != 0;     // (flags: bit 0 = stagger the two wave halves, see the K loop)
  typedef Stage<AOp, BM, NTH> SA;
  typedef Stage<BOp, BN, NTH> SB;
  // wave grid WGM x WGN: 2 x 2 (256 threads), 2 x 4 (512 threads, 128x128) or 4 x 2 (512 threads, 256x64: the tall tile
  // of the 64-channel convolutions); every wave owns a (BM/WGM) x (BN/WGN) = TM x TN grid of 32x32 accumulators
  constexpr int WGN = (BM == 256) ? 2 : NTH / 128;
  constexpr int WGM = NTH / 64 / WGN;
  constexpr int TM = BM / (32 * WGM), TN = BN / (32 * WGN);
  constexpr int TILE = BF ? SA::TILE_BF + SB::TILE_BF : BK * SA::LD + BK * SB::LD;
  constexpr int AOFF = BF ? SA::TILE_BF : BK * SA::LD;     // floats from a tile's A image to its B image
  const int t_stage = threadIdx.x;
  __shared__ __attribute__((aligned(16))) float smem[(DB ? 2 : 1) * TILE];
  auto stage = [&](float* dst, const float (&ra)[SA::NV][SA::VEC], const float (&rb)[SB::NV][SB::VEC], unsigned ma, unsigned mb) {
    if constexpr (BF) {
      SA::to_lds_bf(reinterpret_cast<__bf16*>(dst), t_stage, ra, A, ma);
      SB::to_lds_bf(reinterpret_cast<__bf16*>(dst + AOFF), t_stage, rb, B, mb);
    } else {
      SA::to_lds(dst, t_stage, ra, A, ma);
      SB::to_lds(dst + AOFF, t_stage, rb, B, mb);
    }
  };

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / WGN, wn = wave % WGN;
  const int ntn = (N + BN - 1) / BN;

#ifdef AVVAD_PROF
  if (t == 0 && blockIdx.x < 2048) {
    g_prof_blk[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    g_prof_clk[blockIdx.x] = __builtin_amdgcn_s_memtime();
    g_prof_hw[blockIdx.x] = ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 32) |
                            (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
  }
#endif
  const long G = gridDim.x;
  const long g = xcd_remap(blockIdx.x, gridDim.x);
  long rem_iters = (long)rem_tiles * ktiles;
  if constexpr (HasSched<Epi>::value) rem_iters = E.sched.total();      // tiles of different lengths: every tile is in the pool
  long it = g * rem_iters / G;
  const long it_end = (g + 1) * rem_iters / G;
  const long ntiles_all = (long)((M + BM - 1) / BM) * ntn;
  int round = 0;
  long cell = g;   // K-major mode (kchunks > 0): cell = chunk * ntiles + tile, workers take cells g, g+G, ...

  for (;;) {
    int tile, kt0, kt1;
    int klen = ktiles;      // K tiles of this tile
    if (kchunks > 0) {
      // K-MAJOR CELLS (few tiles, long K: the conv weight gradients).  K is cut into kchunks ranges and the cells
      // (range, tile) are dealt out tile-fastest, so neighbouring workers -- one XCD after the remap -- contract the
      // SAME pixel range for different output tiles and share its dy / x lines in L2.  Tile-major stream-K gave every
      // XCD its own private walk over all pixels: the wgrad kernel fetched 460 MB per launch from beyond L2.
      if (cell >= (long)kchunks * ntiles_all) break;
      const int c = (int)(cell / ntiles_all);
      tile = (int)(cell - (long)c * ntiles_all);
      kt0 = (int)((long)c * ktiles / kchunks);
      kt1 = (int)((long)(c + 1) * ktiles / kchunks);
      cell += G;
      if (kt1 <= kt0) continue;
    } else if (round < full_rounds) {                       // data-parallel rounds: whole tiles
      const long tl = (long)round * G + g;
      ++round;
      if (tl >= ntiles_all) continue;                // last, partially filled round
      tile = (int)tl; kt0 = 0; kt1 = ktiles;
    } else if (it < it_end) {                        // stream-K round over the remainder tiles
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

    typename AOp::Ctx actx[SA::NCTX];
    typename BOp::Ctx bctx[SB::NCTX];
#pragma unroll
    for (int i = 0; i < SA::NCTX; ++i) {
      int xl, kl;
      SA::coord(t, i, xl, kl);
      if constexpr (HasPrep2<AOp>::value) actx[i] = A.prep2(m0 + xl, kl);
      else actx[i] = A.prep(m0 + xl);
    }
#pragma unroll
    for (int i = 0; i < SB::NCTX; ++i) {
      int xl, kl;
      SB::coord(t, i, xl, kl);
      if constexpr (HasPrep2<BOp>::value) bctx[i] = B.prep2(n0 + xl, kl);
      else bctx[i] = B.prep(n0 + xl);
    }

    // position-class operands: what the tile's class means for its K loop (live taps), wave-uniform
    const typename TileOf<AOp>::type atile = TileOf<AOp>::get(A, m0);
    const typename TileOf<BOp>::type btile = TileOf<BOp>::get(B, m0);
    (void)atile; (void)btile;

    float sa[SA::NV][SA::VEC], sb[SB::NV][SB::VEC];
    unsigned oka = ~0u, okb = ~0u;   // bit i: staged vector i holds a real element (see OPERAND CONTRACT)
    static_assert(SA::NV <= 32 && SB::NV <= 32, "validity masks are 32 bits");
    auto gload = [&](int kt) {
#ifdef AVVAD_ABL_HOT
      const int k0 = (kt & 1) * BK;
#else
      const int k0 = kt * BK;
#endif
      if constexpr (HasTile<AOp>::value) {           // (class operands: buffer form, nothing to return)
#pragma unroll
        for (int i = 0; i < SA::NV; ++i) {
          int xl, kl;
          SA::coord(t, i, xl, kl);
          A.load(actx[SA::NCTX > 1 ? i : 0], atile, m0 + xl, k0, kl, sa[i]);
        }
      } else {
        constexpr bool AB = std::is_same<decltype(A.load(actx[0], 0, 0, 0, sa[0])), bool>::value;
        if (AB) oka = 0u;
#pragma unroll
        for (int i = 0; i < SA::NV; ++i) {
          int xl, kl;
          SA::coord(t, i, xl, kl);
          if constexpr (AB) oka |= (unsigned)A.load(actx[SA::NCTX > 1 ? i : 0], m0 + xl, k0, kl, sa[i]) << i;
          else A.load(actx[SA::NCTX > 1 ? i : 0], m0 + xl, k0, kl, sa[i]);
        }
      }
      if constexpr (HasTile<BOp>::value) {
#pragma unroll
        for (int i = 0; i < SB::NV; ++i) {
          int xl, kl;
          SB::coord(t, i, xl, kl);
          B.load(bctx[SB::NCTX > 1 ? i : 0], btile, n0 + xl, k0, kl, sb[i]);
        }
      } else {
        constexpr bool BB = std::is_same<decltype(B.load(bctx[0], 0, 0, 0, sb[0])), bool>::value;
        if (BB) okb = 0u;
#pragma unroll
        for (int i = 0; i < SB::NV; ++i) {
          int xl, kl;
          SB::coord(t, i, xl, kl);
          if constexpr (BB) okb |= (unsigned)B.load(bctx[SB::NCTX > 1 ? i : 0], n0 + xl, k0, kl, sb[i]) << i;
          else B.load(bctx[SB::NCTX > 1 ? i : 0], n0 + xl, k0, kl, sb[i]);
        }
      }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // prologue: tile kt0 -> LDS buffer 0, tile kt0+1 -> registers
    const unsigned long long tp0 = PROF_T();
    unsigned long long tbar = 0, tstage = 0;
    gload(kt0);
    __syncthreads();  // the previous segment's LDS reads are done
    stage(smem, sa, sb, oka, okb);
    if (kt0 + 1 < kt1) gload(kt0 + 1);
    __syncthreads();
    const unsigned long long tp1 = PROF_T();

    for (int kt = kt0; kt < kt1; ++kt) {
      const int cur = DB ? ((kt - kt0) & 1) : 0;
      if constexpr (BF) {
        // bf16: two 16-deep k-steps per K tile; a fragment is one ds_read_b128 (8 bf16 of row m / column n)
        const __bf16* As16 = reinterpret_cast<const __bf16*>(smem + cur * TILE);
        const __bf16* Bs16 = reinterpret_cast<const __bf16*>(smem + cur * TILE + AOFF);
        const __bf16* ap = As16 + (wm * (BM / WGM) + li) * LDH + lh * 8;
        const __bf16* bp = Bs16 + (wn * (BN / WGN) + li) * LDH + lh * 8;
        bf16x8 a[2][TM], b[2][TN];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int i = 0; i < TM; ++i) a[ks][i] = *reinterpret_cast<const bf16x8*>(ap + i * 32 * LDH + ks * 16);
#pragma unroll
          for (int j = 0; j < TN; ++j) b[ks][j] = *reinterpret_cast<const bf16x8*>(bp + j * 32 * LDH + ks * 16);
        }
        if (DB && kt + 1 < kt1) {        // stage tile kt+1 into the other buffer, request tile kt+2: under this tile's MFMAs
          stage(smem + (cur ^ 1) * TILE, sa, sb, oka, okb);
          if (kt + 2 < kt1) gload(kt + 2);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
      } else {
      float* As = smem + cur * TILE;
      float* Bs = As + BK * SA::LD;
      const float* ap = As + lh * SA::LD + wm * (BM / WGM) + li;
      const float* bp = Bs + lh * SB::LD + wn * (BN / WGN) + li;
      float a[2][TM], b[2][TN];
      // STAGGER (MI355X_MICROARCH "Two waves per SIMD" item 9): the 8 waves of a workgroup run the same program with one
      // barrier per K tile, i.e. in lockstep -- both waves of a SIMD reach their staging burst (LDS writes, the next loads'
      // issue) and their MFMA runs together.  Waves 4-7 therefore stage at the START of the K tile, waves 0-3 in its middle
      // (any point of the tile is legal: the other LDS buffer's readers finished before the last barrier).  MEASURED on
      // this kernel (bench.py --ab, same process): the step gets 0.14 ms SLOWER -- two workgroups share a CU here and are
      // out of phase with each other already -- so the stagger is off unless option "stagger" asks for it.
      const int stage_ks = (NTH == 512 && stagger && __builtin_amdgcn_readfirstlane(wave) >= 4) ? 0 : BK / 4;
#pragma unroll
      for (int i = 0; i < TM; ++i) a[0][i] = ap[i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[0][j] = bp[j * 32];
#pragma unroll
      for (int ks = 0; ks < BK / 2; ++ks) {
        if (DB && ks == stage_ks && kt + 1 < kt1) {
          // MID-tile staging (double-buffered LDS): the wait for tile kt+1's global loads, its LDS writes (other
          // buffer: its readers passed the last barrier) and the issue of tile kt+2's loads sit between the two
          // halves of this tile's MFMAs instead of right behind the barrier, where every wave of both co-resident
          // workgroups would do them at the same time with the matrix pipe idle.
          float* An = smem + (cur ^ 1) * TILE;
          const unsigned long long ts0 = PROF_T();
#ifndef AVVAD_ABL_NOLDSW
          stage(An, sa, sb, oka, okb);
#endif
#ifndef AVVAD_ABL_NOGLOAD
          if (kt + 2 < kt1) gload(kt + 2);
#endif
          tstage += PROF_T() - ts0;
        }
#ifndef AVVAD_ABL_NODSR
        if (ks + 1 < BK / 2) {  // fragments of the next k-step are in flight under this step's MFMAs
#pragma unroll
          for (int i = 0; i < TM; ++i) a[(ks + 1) & 1][i] = ap[(ks + 1) * 2 * SA::LD + i * 32];
#pragma unroll
          for (int j = 0; j < TN; ++j) b[(ks + 1) & 1][j] = bp[(ks + 1) * 2 * SB::LD + j * 32];
        }
#endif
        // pin the prefetch ABOVE this step's MFMAs: left alone, hipcc sinks the ds_reads below them (operand
        // registers get reused) and every k-step then eats the LDS latency in front of its MFMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(a[ks & 1][i], b[ks & 1][j], acc[i][j]);
      }
      }
#ifndef AVVAD_ABL_NOBAR
      const unsigned long long tb0 = PROF_T();
      __syncthreads();
      tbar += PROF_T() - tb0;
#endif
      if (!DB && kt + 1 < kt1) {  // single LDS buffer (more workgroups per CU): restage after everyone has read
        stage(smem, sa, sb, oka, okb);
        if (kt + 2 < kt1) gload(kt + 2);
        __syncthreads();
      }
    }

    const unsigned long long tp2 = PROF_T();
    // (the epilogue's lane-dependent index math must not be hoisted above the K loop: opaque copies of the lane
    //  coordinates keep its address registers out of the loop's live set -- hoisted, they pushed the 8-wave kernels past
    //  their 128-VGPR budget and the spills landed around every tile segment)
    int li_e = li, lh_e = lh;
    asm volatile("" : "+v"(li_e), "+v"(lh_e));
    if (kchunks == 0 && kt0 > 0) {
      // Piece of a split tile that does not start the tile's K range (at most ONE per worker: the first segment of its
      // stream-K share): the raw partial sums go to this worker's slab, tile-local [BM][BN]; the fix-up kernel adds the
      // slabs of a tile onto its output in ascending worker (= ascending K) order.  No atomics, no pre-zeroed output.
      float* const S = slab + g * (long)(BM * BN);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = wn * (BN / WGN) + j * 32 + li_e;
          const int rowb = wm * (BM / WGM) + i * 32 + 4 * lh_e;
#pragma unroll
          for (int r = 0; r < 16; ++r) S[(rowb + mfma32_row(r, 0)) * BN + col] = acc[i][j][r];
        }
      continue;
    }
    // (BEFORE the output stores: the LDS exchange needs a workgroup barrier, and a barrier behind the stores would wait for
    //  every one of them to drain -- s_waitcnt vmcnt(0) -- on every tile)
    if constexpr (HasStat<Epi>::value) {
      // Fused BatchNorm statistics (see EpiStore::stat): this worker finished the tile by itself, so the sums are final.
      // Lane (li, lh) holds column n of its wave's TN column blocks and rows (r, lh) of TM x 16 registers: 16 * TM terms in
      // fp32, then doubles -- across the lane halves by a cross-lane move, across the WGM row waves through LDS (free: every
      // wave is past the K loop's last barrier), in a fixed order.  Rows beyond M hold exact zeros (their operand rows were
      // out of range) and add nothing.
      if (E.stat != nullptr && kt0 == 0 && kt1 == klen) {
        double* red = reinterpret_cast<double*>(smem);          // [WGM][BN][2]
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
            const int cl = wn * (BN / WGN) + j * 32 + li_e;
            red[(wm * BN + cl) * 2 + 0] = sd;
            red[(wm * BN + cl) * 2 + 1] = qd;
          }
        }
        __syncthreads();
        if (t < BN && n0 + t < N) {
          double a = 0, b = 0;
#pragma unroll
          for (int w = 0; w < WGM; ++w) { a += red[(w * BN + t) * 2 + 0]; b += red[(w * BN + t) * 2 + 1]; }
          const long tm = tile / ntn;
          E.stat[(tm * 2 + 0) * N + n0 + t] = a;
          E.stat[(tm * 2 + 1) * N + n0 + t] = b;
        }
        // (no barrier behind the reads: the next segment's prologue has one in front of its first LDS write)
      }
    }
    // Whole tile, or the piece that starts the tile's K range (it owns the output until the fix-up kernel runs): plain
    // stores.  The column's bias is fetched once (a load + wait per element costs a memory round trip each).  Plain
    // row-major outputs that fit 32-bit offsets take the fast path: interior tiles skip the per-element range checks
    // and every store is one v_add on a 32-bit offset against the scalar base (global_store_dword v, v, s[C]).
    const bool atomic = kchunks > 0;       // K-major cells (tuning mode, off by default): several cells share a tile
    // where the tile's rows live: the whole output, or (position classes) the view of the tile's grid position
    float* Cv = nullptr;
    int mrow0 = m0, Mv = M;
    if constexpr (HasSched<Epi>::value) { const auto v = E.view(m0); Cv = v.C; mrow0 = v.row0; Mv = v.M; }
    else if constexpr (IsPlain<Epi>::value) Cv = E.C;
    bool fast = false;
    if constexpr (IsPlain<Epi>::value) fast = E.cs == 1 && (long)Mv * E.ldc < (1L << 31) && mrow0 + BM <= Mv && n0 + BN <= N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / WGN) + j * 32 + li_e;
        const float bv = (E.bias && n < N && kt0 == 0) ? E.bias[n] : 0.f;
        const int mb = mrow0 + wm * (BM / WGM) + i * 32;       // (row in the tile's view)
        if constexpr (IsPlain<Epi>::value) {
          if (fast) {
            const unsigned ld32 = (unsigned)E.ldc;
            const unsigned base = (unsigned)(mb + 4 * lh_e) * ld32 + (unsigned)n;
            float* const Cb = Cv;
            if (atomic) {
#pragma unroll
              for (int r = 0; r < 16; ++r) atomicAdd(Cb + (base + (unsigned)mfma32_row(r, 0) * ld32), acc[i][j][r] + bv);
            } else if (E.mode == 0) {
#pragma unroll
              for (int r = 0; r < 16; ++r) Cb[base + (unsigned)mfma32_row(r, 0) * ld32] = acc[i][j][r] + bv;
            } else {
              float old[16];
#pragma unroll
              for (int r = 0; r < 16; ++r) old[r] = Cb[base + (unsigned)mfma32_row(r, 0) * ld32];
#pragma unroll
              for (int r = 0; r < 16; ++r) Cb[base + (unsigned)mfma32_row(r, 0) * ld32] = old[r] + acc[i][j][r] + bv;
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
            const float v = acc[i][j][r] + bv;
            if (atomic) atomicAdd(p, v);
            else if (E.mode == 0) *p = v;
            else *p += v;
          }
        }
      }
#ifdef AVVAD_PROF
    const unsigned long long tp3 = PROF_T();
    PROF_ADD(0, 1); PROF_ADD(1, tp1 - tp0); PROF_ADD(2, tp2 - tp1); PROF_ADD(3, tbar); PROF_ADD(4, tstage); PROF_ADD(5, tp3 - tp2);
    PROF_ADD(6, kt1 - kt0);
#endif
  }
#ifdef AVVAD_PROF
  if (t == 0 && blockIdx.x < 2048) {
    g_prof_blk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    g_prof_clk[blockIdx.x] = __builtin_amdgcn_s_memtime() - g_prof_clk[blockIdx.x];
  }
#endif
}

// Fix-up of the stream-K round: tile tr of the remainder pool was cut between workers ga .. gb (ascending K).  Worker ga
// stored (or accumulated) its piece onto the output, workers ga+1 .. gb left theirs in their slabs; this kernel adds those
// slabs in a FIXED order -- so the result is bit-reproducible run to run.  It replaces both the zero-fill launch in front
// of the GEMM and the float atomics inside it.
// One workgroup = 4 waves per 256-element strip of a tile (lane = one float4): wave w sums the tile's slabs ga+1+w,
// ga+1+w+4, ... with four loads in flight, the four partial sums meet in LDS in wave order.  (First version: one thread
// walked all of a tile's slabs in a serial loop -- latency-bound, 37 us for the 31-way split tiles of the LSTM's
// recurrent products.)
template <int BM, int BN, class Epi>
__global__ void __launch_bounds__(256)
    fixup(const Epi E, const float* __restrict__ slab, const int M, const int N, const int ktiles, const long G,
          const int full_rounds, const int rem_tiles, const int ntn) {
  constexpr int STRIPS = BM * BN / 256;                  // 256-element strips per tile
  __shared__ float4 part[3][64];
  const int tr = blockIdx.x / STRIPS, strip = blockIdx.x % STRIPS;
  // worker of iteration it: shares are [g*R/G, (g+1)*R/G)  ->  g = ceil((it+1)*G/R) - 1.  All of this is block-uniform and
  // in 32 bits (the host launches this kernel only when R * G < 2^32: a 64-bit software division per thread and slab was
  // most of this kernel's time in its first version).
  unsigned R = (unsigned)rem_tiles * (unsigned)ktiles, it0 = (unsigned)tr * (unsigned)ktiles, it1 = it0 + (unsigned)ktiles - 1u;
  const unsigned Gu = (unsigned)G;
  if constexpr (HasSched<Epi>::value) {            // tiles of different lengths, all of them in the pool
    R = (unsigned)E.sched.total();
    it0 = (unsigned)E.sched.base(tr);
    it1 = it0 + (unsigned)E.sched.len(tr) - 1u;
  }
  const int ga = (int)(((it0 + 1u) * Gu + R - 1u) / R) - 1, gb = (int)(((it1 + 1u) * Gu + R - 1u) / R) - 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int e0 = strip * 256 + lane * 4;                 // tile-local element (row-major [BM][BN]) of this lane's float4
  if (gb <= ga) {                                        // the tile was not split: its output is final already
    if constexpr (HasFinish<Epi>::value) {
      if (E.active && wave == 0) {
        const unsigned tile_u = (unsigned)full_rounds * Gu + (unsigned)tr;
        const int mu = (int)(tile_u / (unsigned)ntn) * BM + e0 / BN, nu = (int)(tile_u % (unsigned)ntn) * BN + e0 % BN;
        if (epi_live(E, mu, M)) {
          float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (nu + e < N) v[e] = *E.ptr(mu, nu + e);
          E.finish4(mu, nu, v, N);
        }
      }
    }
    return;
  }
  const bool sparse = R < Gu;                            // fewer iterations than workers: some shares are empty
  const float* S0 = slab + e0;
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  // (a worker whose share is empty wrote no slab: its slot is skipped; shares are contiguous, so every non-empty worker
  //  between ga and gb lies inside this tile)
  for (int g = ga + 1 + wave; g <= gb; g += 16) {
    float4 v[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int gu = g + 4 * u;
      ok[u] = gu <= gb && (!sparse || ((unsigned)(gu + 1) * R) / Gu > ((unsigned)gu * R) / Gu);
      v[u] = *reinterpret_cast<const float4*>(S0 + (long)(ok[u] ? gu : ga + 1) * (BM * BN));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (ok[u]) { sum.x += v[u].x; sum.y += v[u].y; sum.z += v[u].z; sum.w += v[u].w; }
  }
  if (wave > 0) part[wave - 1][lane] = sum;
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int w = 0; w < 3; ++w) { const float4 p = part[w][lane]; sum.x += p.x; sum.y += p.y; sum.z += p.z; sum.w += p.w; }
  const unsigned tile = (unsigned)full_rounds * Gu + (unsigned)tr;
  const int m = (int)(tile / (unsigned)ntn) * BM + e0 / BN, n0 = (int)(tile % (unsigned)ntn) * BN + e0 % BN;
  if (!epi_live(E, m, M)) return;
  const float sv[4] = {sum.x, sum.y, sum.z, sum.w};
  if constexpr (HasFinish<Epi>::value) {
    if (E.active) {
      float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n0 + e < N) v[e] = *E.ptr(m, n0 + e) + sv[e];
      E.finish4(m, n0, v, N);
      return;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (n0 + e < N) *E.ptr(m, n0 + e) += sv[e];
}

// The same fix-up for SHALLOW splits (a tile cut between a handful of workers: the convolutions' stream-K rounds): one WAVE
// per strip, four strips per workgroup, the wave walks its tile's slabs itself.  The kernel above spends four waves, an LDS
// exchange and a barrier on every strip; with ~3 slabs per tile three of them load one slab each -- a 136-tile round was
// 34816 waves for 43 MB, 15 us.  (Deep splits -- the LSTM's 31-way, the weight gradients' 100-way tiles -- keep the
// four-wave form.)
template <int BM, int BN, class Epi>
__global__ void __launch_bounds__(256)
    fixup1(const Epi E, const float* __restrict__ slab, const int M, const int N, const int ktiles, const long G,
           const int full_rounds, const int rem_tiles, const int ntn) {
  constexpr int STRIPS = BM * BN / 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sg = blockIdx.x * 4 + wave;
  const int tr = sg / STRIPS, strip = sg % STRIPS;
  if (tr >= rem_tiles) return;
  unsigned R = (unsigned)rem_tiles * (unsigned)ktiles, it0 = (unsigned)tr * (unsigned)ktiles, it1 = it0 + (unsigned)ktiles - 1u;
  const unsigned Gu = (unsigned)G;
  if constexpr (HasSched<Epi>::value) {            // position classes: tiles of different lengths, all of them in the pool
    R = (unsigned)E.sched.total();
    it0 = (unsigned)E.sched.base(tr);
    it1 = it0 + (unsigned)E.sched.len(tr) - 1u;
  }
  const int ga = (int)(((it0 + 1u) * Gu + R - 1u) / R) - 1, gb = (int)(((it1 + 1u) * Gu + R - 1u) / R) - 1;
  const int e0 = strip * 256 + lane * 4;
  const unsigned tile = (unsigned)full_rounds * Gu + (unsigned)tr;
  const int m = (int)(tile / (unsigned)ntn) * BM + e0 / BN, n0 = (int)(tile % (unsigned)ntn) * BN + e0 % BN;
  if (!epi_live(E, m, M)) return;
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  if (gb > ga) {
    const bool sparse = R < Gu;
    const float* S0 = slab + e0;
    for (int g = ga + 1; g <= gb; g += 4) {
      float4 v[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int gu = g + u;
        ok[u] = gu <= gb && (!sparse || ((unsigned)(gu + 1) * R) / Gu > ((unsigned)gu * R) / Gu);
        v[u] = *reinterpret_cast<const float4*>(S0 + (long)(ok[u] ? gu : ga + 1) * (BM * BN));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (ok[u]) { sum.x += v[u].x; sum.y += v[u].y; sum.z += v[u].z; sum.w += v[u].w; }
    }
  } else if (!HasFinish<Epi>::value) {
    return;                                              // unsplit tile, nothing to finish
  }
  const float sv[4] = {sum.x, sum.y, sum.z, sum.w};
  if constexpr (HasFinish<Epi>::value) {
    if (E.active) {
      float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n0 + e < N) v[e] = *E.ptr(m, n0 + e) + sv[e];
      E.finish4(m, n0, v, N);
      return;
    }
    if (gb <= ga) return;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (n0 + e < N) *E.ptr(m, n0 + e) += sv[e];
}

// Fix-up of split tiles WITH the fused column statistics (EpiStore::stat): one workgroup per remainder tile.  Thread
// (cq, rg) owns the float4 column cq of rows rg, rg + RG, ...: it adds the tile's slabs in ascending worker order onto the
// piece the first worker stored, writes the total back and keeps sum / sum of squares of its columns (fp32 over its BM / RG
// rows, then doubles); the RG row groups meet in LDS in a fixed order.  Tiles that one worker finished by itself are skipped
// (the kernel wrote their statistics).
template <int BM, int BN, class Epi = EpiStore>
__global__ void __launch_bounds__(256)
    fixup_tile(const Epi E, const float* __restrict__ slab, const int M, const int N, const int ktiles, const long G,
               const int full_rounds, const int rem_tiles, const int ntn) {
  constexpr int CQ = BN / 4, RG = 256 / CQ, RPT = BM / RG;
  static_assert(RPT % 4 == 0, "four rows in flight per thread");
  __shared__ double red[RG * BN * 2];
  const int tr = blockIdx.x;
  unsigned R = (unsigned)rem_tiles * (unsigned)ktiles, it0 = (unsigned)tr * (unsigned)ktiles, it1 = it0 + (unsigned)ktiles - 1u;
  const unsigned Gu = (unsigned)G;
  if constexpr (HasSched<Epi>::value) {
    R = (unsigned)E.sched.total();
    it0 = (unsigned)E.sched.base(tr);
    it1 = it0 + (unsigned)E.sched.len(tr) - 1u;
  }
  const int ga = (int)(((it0 + 1u) * Gu + R - 1u) / R) - 1, gb = (int)(((it1 + 1u) * Gu + R - 1u) / R) - 1;
  if (gb <= ga) return;                                  // block-uniform
  const bool sparse = R < Gu;
  const unsigned tile = (unsigned)full_rounds * Gu + (unsigned)tr;
  const int tm = (int)(tile / (unsigned)ntn);
  const int m0 = tm * BM, n0 = (int)(tile % (unsigned)ntn) * BN;
  float* Cv = E.C;
  int mrow0 = m0, Mv = M;
  if constexpr (HasSched<Epi>::value) { const auto v = E.view(m0); Cv = v.C; mrow0 = v.row0; Mv = v.M; }
  const int cq = threadIdx.x % CQ, rg = threadIdx.x / CQ;
  const int n = n0 + cq * 4;
  float s4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};
  if (n < N) {                                           // (N % 4 == 0: a float4 column is inside or outside as a whole)
#pragma unroll 1
    for (int rr = 0; rr < RPT; rr += 4) {
      float4 tot[4];
      bool live[4];
      const float* sp[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int row = rg + RG * (rr + u);
        live[u] = mrow0 + row < Mv;
        sp[u] = slab + (long)(live[u] ? row : 0) * BN + cq * 4;
        tot[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      for (int g = ga + 1; g <= gb; ++g) {
        if (sparse && !(((unsigned)(g + 1) * R) / Gu > ((unsigned)g * R) / Gu)) continue;     // this worker's share was empty
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(sp[u] + (long)g * (BM * BN));
#pragma unroll
        for (int u = 0; u < 4; ++u) { tot[u].x += v[u].x; tot[u].y += v[u].y; tot[u].z += v[u].z; tot[u].w += v[u].w; }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (!live[u]) continue;
        float4* const cp = reinterpret_cast<float4*>(Cv + (long)(mrow0 + rg + RG * (rr + u)) * E.ldc + n);
        float4 o = *cp;
        o.x += tot[u].x; o.y += tot[u].y; o.z += tot[u].z; o.w += tot[u].w;
        *cp = o;
        s4[0] += o.x; s4[1] += o.y; s4[2] += o.z; s4[3] += o.w;
        q4[0] = fmaf(o.x, o.x, q4[0]); q4[1] = fmaf(o.y, o.y, q4[1]); q4[2] = fmaf(o.z, o.z, q4[2]); q4[3] = fmaf(o.w, o.w, q4[3]);
      }
    }
  }
  if (E.stat == nullptr) return;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[(rg * BN + cq * 4 + e) * 2 + 0] = (double)s4[e];
    red[(rg * BN + cq * 4 + e) * 2 + 1] = (double)q4[e];
  }
  __syncthreads();
  const int c = threadIdx.x;
  if (c < BN && n0 + c < N) {
    double a = 0, b = 0;
#pragma unroll 4
    for (int w = 0; w < RG; ++w) { a += red[(w * BN + c) * 2 + 0]; b += red[(w * BN + c) * 2 + 1]; }
    E.stat[((long)tm * 2 + 0) * N + n0 + c] = a;
    E.stat[((long)tm * 2 + 1) * N + n0 + c] = b;
  }
}

constexpr int NUM_CU = 256;  // MI355X

// Floats of slab workspace a launch may need: one BM x BN tile per persistent worker (512 x 128x128 = 1024 x 128x64 =
// 1536 x 64x64 at most).  Callers carve this out of their own workspace (the C ABI's caller-allocates rule).
// (... or one [576][64] weight-gradient partial per CU, conv64.h: 256 x 36864, the larger of the two)
constexpr size_t SLAB_FLOATS = (size_t)256 * 576 * 64;
static_assert(SLAB_FLOATS >= (size_t)512 * 128 * 128, "one 128x128 tile per persistent worker");

// split_k_hint > 1 marks a "few tiles, very long K" product (the conv weight gradients); with the K-major tuning option
// it is cut K-major (see the kernel); everything else is scheduled as data-parallel rounds + one stream-K round.
// `slab`: SLAB_FLOATS floats of scratch for the split tiles of the stream-K round; nullptr selects whole-tile scheduling
// (slower on tile counts that quantise badly, still deterministic).
// `allow_bf16`: the process-wide option "bf16" applies to this product (false for the STFT, whose DFT stays exact fp32).
template <int BM, int BN, class AOp, class BOp, class Epi>
static inline int launch(const AOp& a, const BOp& b, const Epi& e_in, int M, int N, int K, int split_k_hint,
                         hipStream_t s, float* slab = nullptr, bool allow_bf16 = true, bool* finished = nullptr) {
  Epi e = e_in;
  if (M <= 0 || N <= 0 || K <= 0) return AVVAD_EINVAL;
  const int ktiles = (K + BK - 1) / BK;
  const long ntiles = (long)cdiv(M, BM) * cdiv(N, BN);
  const long iters = ntiles * ktiles;
  const AvvadTune& tn = avvad_tune();
  const int var = tn.igemm_variant;  // tuning aid: 0/1 = 4-wave double/single LDS buffer, 2 = 8 waves
  constexpr bool BIG = (BM == 128 && BN == 128);
  constexpr bool TALL = (BM == 256);             // 256x64, 8 waves as 4x2, single LDS buffer (41.6 KB), 2 workgroups / CU
  const int variant = TALL ? 3 : (var >= 0 ? var : (BIG ? 2 : 1));
  // resident workgroups per CU (LDS footprint / VGPR budget of the instantiation)
  int per_cu;
  if (variant == 3) per_cu = 2;
  else if (variant == 0) per_cu = BIG ? 2 : (BM * BN >= 128 * 64 ? 3 : 4);
  else if (variant == 1 || !BIG) per_cu = BIG ? 3 : (BM * BN >= 128 * 64 ? 4 : 6);
  else per_cu = 2;
  const int cus = (tn.max_cus > 0 && tn.max_cus < NUM_CU) ? tn.max_cus : NUM_CU;   // room for RCCL's kernels at N > 1
  long G = (long)cus * per_cu;
  if (G * BM * BN > (long)SLAB_FLOATS) G = (long)(SLAB_FLOATS / ((size_t)BM * BN));
  const bool no_sk = !slab || tn.no_streamk == 1 || tn.no_streamk == 10 + e.mode ||   // whole-tile schedule
                     (double)ntiles * ktiles * (double)G >= 4.0e9;                     // (fix-up index math is 32-bit)
  long full_rounds = ntiles / G, rem = ntiles - full_rounds * G;
  int kchunks = 0;
  // OFF by default (option "kmajor"): it removes most of the wgrad kernel's beyond-L2 fetches, but the
  // kernels are MFMA-bound -- isolated they time the same (+-2 %), and the whole training step measured 0.45 ms
  // (2 %) SLOWER in three A/B/A/B pairs on one device (23.7 vs 23.25 ms), so tile-major stream-K stays the default.
  // (Cells of one tile are added with float atomics onto an accumulating output: the one non-reproducible schedule left.)
  if (split_k_hint > 1 && e.mode != 0 && ntiles <= G && tn.kmajor) {
    // r cells per worker (r = 1 or 2): kchunks = floor(r*G / ntiles); cost in K-tile iterations incl. ~8 per atomic flush
    long best = -1;
    for (int r = 1; r <= 2; ++r) {
      long nc = r * G / ntiles;
      if (nc > ktiles) nc = ktiles;
      if (nc < 1) nc = 1;
      const long rr = (nc * ntiles + G - 1) / G;
      const long cost = rr * ((ktiles + nc - 1) / nc) + rr * 8;
      if (best < 0 || cost < best) { best = cost; kchunks = (int)nc; }
    }
    full_rounds = 0; rem = 0;
    if ((long)kchunks * ntiles < G) G = (long)kchunks * ntiles;
  } else {
    if (no_sk || rem * 10 >= G * 9) {   // (nearly) full last round: keep it data-parallel
      if (rem > 0) ++full_rounds;
      rem = 0;
    }
    if (rem > 0 && full_rounds == 0) {  // fewer tiles than workers: >= 4 iterations per worker amortise prologue/epilogue
      const long cap = iters / 4 > 0 ? iters / 4 : 1;
      if (G > cap) G = cap;
    }
  }
  const int fr = (int)full_rounds, rt = (int)rem;
  const bool bf = allow_bf16 && tn.bf16;
  bool stat = false;
  if constexpr (std::is_same<Epi, EpiStore>::value) {
    stat = e.stat != nullptr;
    if (stat && (e.cs != 1 || (e.ldc & 3) || (N & 3) || (((uintptr_t)e.C) & 15) || kchunks > 0 || e.bias != nullptr)) return AVVAD_EINVAL;
  }
  if constexpr (HasFinish<Epi>::value) {
    // finish4() runs in the fix-up kernel: only when every tile of the product goes through the stream-K pool
    e.active = (rt > 0 && kchunks == 0 && fr == 0 && (long)rt == ntiles) ? 1 : 0;
    if (finished) *finished = e.active != 0;
  } else if (finished) {
    *finished = false;
  }
  const int kflags = tn.stagger ? 1 : 0;
#define AVVAD_IGEMM_LAUNCH(DBV, NTHV)                                                                                       \
  do {                                                                                                                      \
    if (bf)                                                                                                                 \
      hipLaunchKernelGGL((kernel<BM, BN, DBV, NTHV, AOp, BOp, Epi, true>), dim3((int)G), dim3(NTHV), 0, s, a, b, e, M, N, kflags, \
                         ktiles, fr, rt, kchunks, slab);                                                                    \
    else                                                                                                                    \
      hipLaunchKernelGGL((kernel<BM, BN, DBV, NTHV, AOp, BOp, Epi, false>), dim3((int)G), dim3(NTHV), 0, s, a, b, e, M, N, kflags, \
                         ktiles, fr, rt, kchunks, slab);                                                                    \
  } while (0)
  if constexpr (TALL) {
    AVVAD_IGEMM_LAUNCH(false, 512);
  } else {
    if (variant == 0) AVVAD_IGEMM_LAUNCH(true, 256);
    else if (variant == 1 || !BIG) AVVAD_IGEMM_LAUNCH(false, 256);
    else AVVAD_IGEMM_LAUNCH(true, (BIG ? 512 : 256));
  }
#undef AVVAD_IGEMM_LAUNCH
  if (rt > 0 && kchunks == 0 && stat) {
    if constexpr (std::is_same<Epi, EpiStore>::value)
      hipLaunchKernelGGL((fixup_tile<BM, BN>), dim3(rt), dim3(256), 0, s, e, slab, M, N, ktiles, G, fr, rt, cdiv(N, BN));
  } else if (rt > 0 && kchunks == 0) {
    const int deep = tn.no_fixup1 > 1 ? tn.no_fixup1 : 6;     // (option values > 1: the depth threshold, tuning aid)
    if ((G + rt - 1) / rt <= deep && tn.no_fixup1 != 1)      // a handful of slabs per tile: one wave per strip
      hipLaunchKernelGGL((fixup1<BM, BN, Epi>), dim3(rt * (BM * BN / 1024)), dim3(256), 0, s, e, slab, M, N, ktiles, G, fr, rt, cdiv(N, BN));
    else
      hipLaunchKernelGGL((fixup<BM, BN, Epi>), dim3(rt * (BM * BN / 256)), dim3(256), 0, s, e, slab, M, N, ktiles, G, fr, rt, cdiv(N, BN));
  }
  AVVAD_LAUNCH_CHECK();
  return AVVAD_OK;
}

// Position-class product (ClassSched / EpiCls above): every tile is in the stream-K pool, tiles differ in length.  Needs the
// scratch and a tile count that does not exceed the worker count; the caller falls back to launch() otherwise (returns
// AVVAD_EINVAL without launching anything).  128x128 tiles, 8 waves, fp32.
template <class AOp, class BOp, class Epi>
static inline int launch_cls(const AOp& a, const BOp& b, const Epi& e, int Mp, int N, hipStream_t s, float* slab) {
  constexpr int BM = 128, BN = 128;
  if (Mp <= 0 || N <= 0 || !slab) return AVVAD_EINVAL;
  const AvvadTune& tn = avvad_tune();
  const long ntiles = (long)cdiv(Mp, BM) * cdiv(N, BN);
  const int cus = (tn.max_cus > 0 && tn.max_cus < NUM_CU) ? tn.max_cus : NUM_CU;
  long G = (long)cus * 2;
  if (G * BM * BN > (long)SLAB_FLOATS) G = (long)(SLAB_FLOATS / ((size_t)BM * BN));
  const long R = e.sched.total();
  if (R <= 0 || (double)R * (double)G >= 4.0e9) return AVVAD_EINVAL;
  if (G > R / 4) G = R / 4 > 0 ? R / 4 : 1;
  bool stat = false;
  if constexpr (HasStat<Epi>::value) {
    stat = e.stat != nullptr;
    if (stat && (e.cs != 1 || (e.ldc & 3) || (N & 3) || (((uintptr_t)e.C) & 15) || (e.W & 3))) return AVVAD_EINVAL;
  }
  hipLaunchKernelGGL((kernel<BM, BN, true, 512, AOp, BOp, Epi, false>), dim3((int)G), dim3(512), 0, s, a, b, e, Mp, N,
                     tn.stagger ? 1 : 0, 1, 0, (int)ntiles, 0, slab);
  if (stat) {
    if constexpr (HasStat<Epi>::value)
      hipLaunchKernelGGL((fixup_tile<BM, BN, Epi>), dim3((int)ntiles), dim3(256), 0, s, e, slab, Mp, N, 1, G, 0, (int)ntiles, cdiv(N, BN));
  } else if ((G + ntiles - 1) / ntiles <= 6) {
    hipLaunchKernelGGL((fixup1<BM, BN, Epi>), dim3((int)ntiles * (BM * BN / 1024)), dim3(256), 0, s, e, slab, Mp, N, 1, G, 0,
                       (int)ntiles, cdiv(N, BN));
  } else {
    hipLaunchKernelGGL((fixup<BM, BN, Epi>), dim3((int)ntiles * (BM * BN / 256)), dim3(256), 0, s, e, slab, Mp, N, 1, G, 0, (int)ntiles,
                       cdiv(N, BN));
  }
  AVVAD_LAUNCH_CHECK();
  return AVVAD_OK;
}

}  // namespace igemm
