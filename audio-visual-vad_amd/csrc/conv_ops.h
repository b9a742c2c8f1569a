// conv_ops.h -- NHWC convolution operand functors for the igemm engine.
//
// Geometry of one Conv2d (square kernel / stride / pad, no bias, groups=1):
//   x  [N][H][W][C]   ->   y [N][Ho][Wo][Co]
// forward : M = N*Ho*Wo pixels, N = Co, K = KH*KW*C   A = im2col(x)      B = Wf[(kh,kw,c)][co]
// dgrad   : M = N*H*W pixels,   N = C,  K = KH*KW*Co  A = im2col'(dy)    B = Wd[(kh,kw,co)][c]
// wgrad   : M = KH*KW*C,        N = Co, K = N*Ho*Wo   A = x gathered^T   B = dy[(pixel)][co]
// C and Co are multiples of 32 for every trunk conv but the stem (C = 1),
// which has its own scalar-gather functors.
//
// K ORDER.  The contraction index is walked as (32-channel chunk, tap, channel-in-chunk): consecutive BK=32
// K-tiles are the 9 taps of ONE 128-byte channel chunk, i.e. the same few input pixels shifted by one -- the
// re-reads hit L1/L2.  With the textbook (tap, channel) order a tap's re-read comes C/32 K-tiles later, by
// which time an XCD's 64 resident workgroups have streamed > 4 MiB through its L2: rocprof showed the forward
// kernel fetching 7.5x its input from beyond L2 (profiles/r01_pmc_fetch_size_per_kernel.csv).  Weights stay in
// the plain packed layouts; the B functors remap rows.
//
// BUF.  Every gather functor comes in two addressing forms.  BUF = true (operands < 2 GiB, checked by the launcher): the
// fetch is a buffer load against a descriptor over the operand, an element that is padding / out of range carries the
// out-of-range offset BUF_OOB and comes back as 0 from the bounds check, and load() returns nothing -- the engine then
// compiles no validity selects into the staging at all (3 % of the forward kernel's time, VALU instructions that add to
// fp32-MFMA time on this chip) and the address is a 32-bit register instead of a 64-bit add per load.  BUF = false: flat
// loads from a clamped address + a validity bit, zero-filled at the LDS store (operands of 2 GiB and more).
#pragma once
#include "igemm.h"

namespace convop {

// 16 bytes at byte offset `off` of operand p (BUF: off == BUF_OOB -> zeros; flat: the caller clamped off to a valid address)
template <bool BUF>
__device__ __forceinline__ void fetch4(const float* p, unsigned off, float* v) {
  if constexpr (BUF) {
    const f4v t = bload4(brsrc2g(p), (int)off, 0);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  } else {
    const float4 t = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p) + off);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
}
template <bool BUF> struct LoadRet { typedef bool type; };
template <> struct LoadRet<true> { typedef void type; };
#define CONVOP_RETURN(ok) do { if constexpr (!BUF) return ok; else return; } while (0)
constexpr unsigned INVALID_OFF(bool buf) { return buf ? 0x80000000u : 0u; }

struct Geom {
  int N, H, W, C, Ho, Wo, Co, KS, stride, pad;
};

// exact unsigned division by a runtime constant: q = mulhi(k, magic), magic = floor(2^32/d) + 1, valid for
// k * d < 2^32 (checked on the host) -- one v_mul_hi_u32 instead of the ~30-instruction software divide that the
// pixel decomposition of every gathered element would otherwise cost
static inline unsigned div_magic(unsigned d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull / d) + 1ull); }   // 0: divide by 1
__device__ __forceinline__ int fast_div(int k, unsigned magic) { return magic ? (int)__umulhi((unsigned)k, magic) : k; }

// magic numbers of the divisors that appear in the per-K-tile tap decomposition
struct TapDiv {
  unsigned mg_T, mg_KS;
};
static inline TapDiv tap_div(int T, int KS) { return TapDiv{div_magic((unsigned)T), div_magic((unsigned)KS)}; }

// ---- forward A: rows = output pixels, K = (kh,kw,c), c contiguous (16-byte loads)
// Context per staged vector: the BYTE offset of (window origin, channel kin) and one validity bit per tap.  A K tile is
// one tap of one 32-channel chunk (wave-uniform), so load() adds a scalar tap offset and tests one bit.
// (host, trunk.hip conv_fwd_t / conv_dgrad_t: the activation tensor is < 4 GiB -- 32-bit byte offsets -- and KS*KS <= 32,
//  one tap-validity bit each; larger kernels are refused with AVVAD_EINVAL)
template <bool BUF>
struct Im2colFwd {
  static constexpr bool KCONTIG = true;
  static constexpr int VEC = 4;
  struct Ctx { unsigned boff, mask; };
  const float* x;
  Geom g;
  int M;
  TapDiv td;
  __device__ __forceinline__ Ctx prep(int m) const { return prep2(m, 0); }
  __device__ __forceinline__ Ctx prep2(int m, int kin) const {
    Ctx c;
    c.boff = 0u; c.mask = 0u;
    if (m >= M) return c;
    const int hw = g.Ho * g.Wo;
    const int n = m / hw, r = m - n * hw;
    const int ho = r / g.Wo, wo = r - ho * g.Wo;
    const int hi0 = ho * g.stride - g.pad, wi0 = wo * g.stride - g.pad;
    // (unsigned arithmetic: the sum may wrap below zero for a window that starts in the padding; it is used only with a
    //  valid tap, whose offset brings it back in range)
    c.boff = ((((unsigned)n * (unsigned)g.H + (unsigned)hi0) * (unsigned)g.W + (unsigned)wi0) * (unsigned)g.C + (unsigned)kin) * 4u;
    for (int kh = 0; kh < g.KS; ++kh)
      for (int kw = 0; kw < g.KS; ++kw)
        if ((unsigned)(hi0 + kh) < (unsigned)g.H && (unsigned)(wi0 + kw) < (unsigned)g.W) c.mask |= 1u << (kh * g.KS + kw);
    return c;
  }
  __device__ __forceinline__ typename LoadRet<BUF>::type load(const Ctx& c, int, int k0, int, float* v) const {
    const int T = g.KS * g.KS, q = k0 >> 5;  // wave-uniform (k0 % 32 == 0, C % 32 == 0)
    const int cc = fast_div(q, td.mg_T), tap = q - cc * T;
    const int kh = fast_div(tap, td.mg_KS), kw = tap - kh * g.KS;
    const unsigned soff = ((unsigned)(kh * g.W + kw) * (unsigned)g.C + (unsigned)cc * 32u) * 4u;   // scalar
    const bool ok = (c.mask >> tap) & 1u;
    fetch4<BUF>(x, ok ? c.boff + soff : INVALID_OFF(BUF), v);
    CONVOP_RETURN(ok);
  }
};

// ---- dgrad A: rows = INPUT pixels, K = (kh,kw,co), gathers dy  (same per-vector context as the forward gather)
template <bool BUF>
struct Im2colDgrad {
  static constexpr bool KCONTIG = true;
  static constexpr int VEC = 4;
  struct Ctx { unsigned boff[2]; unsigned mask; };   // stride 1: boff[0] only; stride 2: tap (kh,kw) -> pixel ((hp-kh)/2, (wp-kw)/2)
  const float* dy;
  Geom g;
  int M;
  TapDiv td;
  __device__ __forceinline__ Ctx prep(int m) const { return prep2(m, 0); }
  __device__ __forceinline__ Ctx prep2(int m, int kin) const {
    Ctx c;
    c.boff[0] = c.boff[1] = 0u; c.mask = 0u;
    if (m >= M) return c;
    const int hw = g.H * g.W;
    const int n = m / hw, r = m - n * hw;
    const int hi = r / g.W, wi = r - hi * g.W;
    const int hp = hi + g.pad, wp = wi + g.pad;
    for (int kh = 0; kh < g.KS; ++kh)
      for (int kw = 0; kw < g.KS; ++kw) {
        int ho = hp - kh, wo = wp - kw;
        bool ok = ho >= 0 && wo >= 0;
        if (g.stride == 2) { ok = ok && !((ho | wo) & 1); ho >>= 1; wo >>= 1; }
        if (ok && ho < g.Ho && wo < g.Wo) c.mask |= 1u << (kh * g.KS + kw);
      }
    // stride 1: pixel (hp - kh, wp - kw) = origin (hp, wp) minus the tap; stride 2: the per-tap pixel is not affine in the
    // tap, so the origin keeps (hp, wp) and load() halves the (scalar) tap shift -- valid taps have even hp-kh / wp-kw
    if (g.stride == 1) c.boff[0] = ((((unsigned)n * (unsigned)g.Ho + (unsigned)hp) * (unsigned)g.Wo + (unsigned)wp) * (unsigned)g.Co + (unsigned)kin) * 4u;   // affine in the tap
    else c.boff[0] = ((unsigned)n * (unsigned)g.Ho * (unsigned)g.Wo * (unsigned)g.Co + (unsigned)kin) * 4u;
    c.boff[1] = (unsigned)(hp * 65536 + wp);
    return c;
  }
  __device__ __forceinline__ typename LoadRet<BUF>::type load(const Ctx& c, int, int k0, int, float* v) const {
    const int T = g.KS * g.KS, q = k0 >> 5;
    const int cc = fast_div(q, td.mg_T), tap = q - cc * T;
    const int kh = fast_div(tap, td.mg_KS), kw = tap - kh * g.KS;
    const bool ok = (c.mask >> tap) & 1u;
    unsigned off;
    if (g.stride == 1) {                   // one add of a scalar: pixel (hp - kh, wp - kw)
      off = ok ? c.boff[0] + ((unsigned)cc * 32u - (unsigned)(kh * g.Wo + kw) * (unsigned)g.Co) * 4u : INVALID_OFF(BUF);
    } else {
      const int ho = ((int)(c.boff[1] >> 16) - kh) >> 1, wo = ((int)(c.boff[1] & 0xffffu) - kw) >> 1;
      off = ok ? c.boff[0] + ((unsigned)(ho * g.Wo + wo) * (unsigned)g.Co + (unsigned)cc * 32u) * 4u : INVALID_OFF(BUF);
    }
    fetch4<BUF>(dy, off, v);
    CONVOP_RETURN(ok);
  }
};


// ---- stride-2 dgrad, one input-pixel parity class (ph, pw) at a time.
// For stride 2 only the taps with kh = (hi + pad) mod 2 (step 2) reach an input row hi, so the 3x3 kernel splits
// into 4 classes with 1, 2, 2 and 4 live taps (the 1x1 downsample: 1 class with 1 tap) -- 2.25/9 of the MACs of
// the dense formulation, which multiplies structural zeros.  Rows of this GEMM enumerate the class's pixels
// (hi = 2i + ph, wi = 2j + pw); K enumerates (live tap, co).
struct S2Class {
  int ph, pw, Hc, Wc;   // class grid
  int kh0, kw0, nkh, nkw;
  int oh, ow;           // ho = i + oh - a, wo = j + ow - b for tap (kh0 + 2a, kw0 + 2b)
  unsigned mg_ntap, mg_nkw;   // div_magic(nkh*nkw), div_magic(nkw)
};
template <bool BUF>
struct Im2colDgradS2 {
  static constexpr bool KCONTIG = true;
  static constexpr int VEC = 4;
  struct Ctx { int base, i, j; };
  const float* dy;
  Geom g;
  S2Class c;
  int M;
  __device__ __forceinline__ Ctx prep(int m) const {
    Ctx x;
    if (m >= M) { x.base = -1; x.i = x.j = 0; return x; }
    const int hw = c.Hc * c.Wc;
    const int n = m / hw, r = m - n * hw;
    x.i = r / c.Wc;
    x.j = r - x.i * c.Wc;
    x.base = n * g.Ho * g.Wo;
    return x;
  }
  __device__ __forceinline__ typename LoadRet<BUF>::type load(const Ctx& x, int, int k0, int kin, float* v) const {
    const int ntap = c.nkh * c.nkw, q = k0 >> 5;
    const int cc = fast_div(q, c.mg_ntap), seg = q - cc * ntap;
    const int c0 = cc * 32 + kin;
    const int a = fast_div(seg, c.mg_nkw), b = seg - a * c.nkw;
    const int ho = x.i + c.oh - a, wo = x.j + c.ow - b;
    const bool ok = x.base >= 0 && (unsigned)ho < (unsigned)g.Ho && (unsigned)wo < (unsigned)g.Wo;
    fetch4<BUF>(dy, ok ? (unsigned)((x.base + ho * g.Wo + wo) * g.Co + c0) * 4u : INVALID_OFF(BUF), v);
    CONVOP_RETURN(ok);
  }
};
// weight rows of the live taps in (chunk, tap, channel) K order:
//   k = (cc*ntap + tap)*32 + r  ->  row rowbase[tap] + cc*32 + r
template <bool BUF>
struct ColSegRows {
  static constexpr bool KCONTIG = false;
  static constexpr int VEC = 4;
  typedef igemm::NoCtx Ctx;
  const float* p;
  long ld;
  int X, K, ntap;
  int rowbase[4];
  unsigned mg_ntap;
  __device__ __forceinline__ Ctx prep(int) const { return Ctx(); }
  __device__ __forceinline__ typename LoadRet<BUF>::type load(const Ctx&, int x, int k0, int kin, float* v) const {
    const int k = k0 + kin;
    const bool ok = k < K && x < X;
    const int q = k >> 5, cc = fast_div(q, mg_ntap), seg = (q - cc * ntap) & 3;
    const int rb = seg == 0 ? rowbase[0] : (seg == 1 ? rowbase[1] : (seg == 2 ? rowbase[2] : rowbase[3]));  // no scratch array
    fetch4<BUF>(p, ok ? (unsigned)((rb + cc * 32 + (k & 31)) * (int)ld + x) * 4u : INVALID_OFF(BUF), v);
    CONVOP_RETURN(ok);
  }
};
// epilogue: class-local row m -> input pixel (n, 2i+ph, 2j+pw); always accumulating (the caller zero-fills; plain RMW:
// the four parity classes write disjoint pixels and run one after the other on the stream)
struct EpiS2 {
  float* C;
  long ldc;
  const float* bias;   // unused (interface symmetry)
  int mode;            // 1
  int cs;
  int H, W, Hc, Wc, ph, pw;
  unsigned mg_hw, mg_wc;   // div_magic(Hc*Wc), div_magic(Wc)
  __device__ __forceinline__ float* ptr(int m, int n) const {
    const int hw = Hc * Wc;
    const int img = fast_div(m, mg_hw), r = m - img * hw;
    const int i = fast_div(r, mg_wc), j = r - i * Wc;
    return C + ((long)(img * H + 2 * i + ph) * W + 2 * j + pw) * ldc + n;
  }
};

// packed weights [(tap, ch)][x] read in (chunk, tap, channel) K order: k = (cc*T + tap)*32 + r -> row tap*C + cc*32 + r.
// Per-vector context: byte offset of (row r = k_local, column x); a K tile adds the scalar (tap*C + cc*32) * ld.
// (K % 32 == 0 for every convolution that uses this functor: a K tile is valid as a whole)
template <bool BUF>
struct ColTapRows {
  static constexpr bool KCONTIG = false;
  static constexpr int VEC = 4;
  static constexpr bool PERVEC = true;
  struct Ctx { unsigned boff; int ok; };
  const float* p;
  long ld;
  int X, K, C, T;
  unsigned mg_T;
  __device__ __forceinline__ Ctx prep(int x) const { return prep2(x, 0); }
  __device__ __forceinline__ Ctx prep2(int x, int kl) const {
    Ctx c;
    c.ok = x < X;
    c.boff = c.ok ? (unsigned)((kl * (int)ld + x) * 4) : 0u;
    return c;
  }
  __device__ __forceinline__ typename LoadRet<BUF>::type load(const Ctx& c, int, int k0, int, float* v) const {
    const int q = k0 >> 5, cc = fast_div(q, mg_T), tap = q - cc * T;         // wave-uniform
    const bool ok = c.ok && k0 < K;
    fetch4<BUF>(p, ok ? c.boff + (unsigned)((tap * C + cc * 32) * (int)ld * 4) : INVALID_OFF(BUF), v);
    CONVOP_RETURN(ok);
  }
};
// wgrad epilogue: GEMM row m' = (cc*T + tap)*32 + r (chunk-major, see WgradX) -> packed row tap*C + cc*32 + r
struct EpiWgrad {
  float* C;
  long ldc;
  const float* bias;  // unused
  int mode;           // 0: the packed gradient is overwritten (split tiles are combined by igemm::fixup)
  int cs;
  int Cch, T;
  unsigned mg_T;
  int sh = 5;         // log2 of the channel chunk: 5 (fp32 engine, 32-channel chunks) or 6 (bf16 engine, 64)
  __device__ __forceinline__ float* ptr(int m, int n) const {
    const int q = m >> sh, cc = fast_div(q, mg_T), tap = q - cc * T;
    return C + (long)(tap * Cch + (cc << sh) + (m & ((1 << sh) - 1))) * ldc + n;
  }
};

// ---- wgrad A: A[m][k = output pixel] = x[n, ho*s-p+kh, wo*s-p+kw, c] with m = (cc*T + tap)*32 + r, c = cc*32 + r:
// a 128-row M tile is 4 taps of one 32-channel chunk, so the taps' overlapping pixel reads share L1/L2 lines
template <bool BUF>
struct WgradX {
  static constexpr bool KCONTIG = false;
  static constexpr int VEC = 4;
  struct Ctx { int kh, kw, c; };
  const float* x;
  Geom g;
  int M, K;
  unsigned mg_hw, mg_wo;
  __device__ __forceinline__ Ctx prep(int m) const {
    Ctx c;
    if (m >= M) { c.kh = -1; c.kw = c.c = 0; return c; }
    const int T = g.KS * g.KS, q = m >> 5;
    const int cc = q / T, tap = q - cc * T;
    c.c = cc * 32 + (m & 31);
    c.kh = tap / g.KS;
    c.kw = tap - c.kh * g.KS;
    return c;
  }
  __device__ __forceinline__ typename LoadRet<BUF>::type load(const Ctx& c, int, int k0, int kin, float* v) const {
    const int k = k0 + kin;
    const int hw = g.Ho * g.Wo;
    const int n = fast_div(k, mg_hw), r = k - n * hw;
    const int ho = fast_div(r, mg_wo), wo = r - ho * g.Wo;
    const int hi = ho * g.stride - g.pad + c.kh, wi = wo * g.stride - g.pad + c.kw;
    const bool ok = c.kh >= 0 && k < K && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
    fetch4<BUF>(x, ok ? (unsigned)(((n * g.H + hi) * g.W + wi) * g.C + c.c) * 4u : INVALID_OFF(BUF), v);
    CONVOP_RETURN(ok);
  }
};

// ---- position-class operands (igemm.h "position classes"): a tile's 128 rows are 128 images at ONE grid position (ClassRow), so
// that they share their set of live taps and the K loop walks only those -- no validity masks at all.  3x3 / pad 1.
// K tile q of a tile of class (a x b live taps from (kh_lo, kw_lo)):  q = cc * (a b) + (kh - kh_lo) * b + (kw - kw_lo).
__device__ __forceinline__ void cls_tap(const igemm::ClassTile& t, int k0, int& cc, int& kh, int& kw) {
  const int q = k0 >> 5, ab = t.a * t.b;
  cc = igemm::dev_div(q, t.mg_ab);
  const int j = q - cc * ab, dkh = igemm::dev_div(j, t.mg_b);
  kh = t.kh_lo + dkh * t.step;
  kw = t.kw_lo + (j - dkh * t.b) * t.step;
}
// forward A: classes = OUTPUT positions (ho, wo); row n of class p reads x[n, ho s - 1 + kh, wo s - 1 + kw, :]
struct Im2colFwdCls {
  static constexpr bool KCONTIG = true;
  static constexpr int VEC = 4;
  struct Ctx { unsigned boff; int ok; };
  const float* x;
  Geom g;
  int rows;
  igemm::ClassRow cr;
  igemm::ClassSched sc;
  __device__ __forceinline__ Ctx prep(int m) const { return prep2(m, 0); }
  __device__ __forceinline__ Ctx prep2(int m, int kin) const {
    int n, p;
    cr.split(m, n, p);
    Ctx c;
    c.ok = n < rows;
    c.boff = ((unsigned)n * (unsigned)(g.H * g.W) * (unsigned)g.C + (unsigned)kin) * 4u;
    return c;
  }
  __device__ __forceinline__ igemm::ClassTile tile(int m0) const {
    int n, p;
    cr.split(m0, n, p);
    return igemm::class_tile(sc, p, false);
  }
  __device__ __forceinline__ void load(const Ctx& c, const igemm::ClassTile& t, int, int k0, int, float* v) const {
    int cc, kh, kw;
    cls_tap(t, k0, cc, kh, kw);
    const int hi = t.ho * g.stride - g.pad + kh, wi = t.wo * g.stride - g.pad + kw;         // inside the image by construction
    const unsigned soff = ((unsigned)(hi * g.W + wi) * (unsigned)g.C + (unsigned)cc * 32u) * 4u;   // scalar
    fetch4<true>(x, c.ok ? c.boff + soff : INVALID_OFF(true), v);
  }
};
// data gradient A: classes = INPUT positions (hi, wi); row n of class p reads dy[n, (hi + 1 - kh) / s, (wi + 1 - kw) / s, :]
// (stride 2: the schedule's live taps are exactly those for which the quotient is whole and inside dy, ClassSched::s2)
struct Im2colDgradCls {
  static constexpr bool KCONTIG = true;
  static constexpr int VEC = 4;
  struct Ctx { unsigned boff; int ok; };
  const float* dy;
  Geom g;
  int rows;
  igemm::ClassRow cr;
  igemm::ClassSched sc;
  __device__ __forceinline__ Ctx prep(int m) const { return prep2(m, 0); }
  __device__ __forceinline__ Ctx prep2(int m, int kin) const {
    int n, p;
    cr.split(m, n, p);
    Ctx c;
    c.ok = n < rows;
    c.boff = ((unsigned)n * (unsigned)(g.Ho * g.Wo) * (unsigned)g.Co + (unsigned)kin) * 4u;
    return c;
  }
  __device__ __forceinline__ igemm::ClassTile tile(int m0) const {
    int n, p;
    cr.split(m0, n, p);
    return igemm::class_tile(sc, p, true);
  }
  __device__ __forceinline__ void load(const Ctx& c, const igemm::ClassTile& t, int, int k0, int, float* v) const {
    int cc, kh, kw;
    cls_tap(t, k0, cc, kh, kw);
    const int sh = g.stride == 2 ? 1 : 0;
    const int ho = (t.ho + g.pad - kh) >> sh, wo = (t.wo + g.pad - kw) >> sh;
    const unsigned soff = ((unsigned)(ho * g.Wo + wo) * (unsigned)g.Co + (unsigned)cc * 32u) * 4u;
    fetch4<true>(dy, c.ok ? c.boff + soff : INVALID_OFF(true), v);
  }
};
// packed weights [(tap, ch)][x] for the live taps of the tile's class (rows tap * C + cc * 32 + r)
struct ColTapRowsCls {
  static constexpr bool KCONTIG = false;
  static constexpr int VEC = 4;
  static constexpr bool PERVEC = true;
  struct Ctx { unsigned boff; int ok; };
  const float* p;
  long ld;
  int X, C, KS, flip;
  igemm::ClassRow cr;
  igemm::ClassSched sc;
  __device__ __forceinline__ Ctx prep(int x) const { return prep2(x, 0); }
  __device__ __forceinline__ Ctx prep2(int x, int kl) const {
    Ctx c;
    c.ok = x < X;
    c.boff = c.ok ? (unsigned)((kl * (int)ld + x) * 4) : 0u;
    return c;
  }
  __device__ __forceinline__ igemm::ClassTile tile(int m0) const {
    int n, pp;
    cr.split(m0, n, pp);
    return igemm::class_tile(sc, pp, flip != 0);
  }
  __device__ __forceinline__ void load(const Ctx& c, const igemm::ClassTile& t, int, int k0, int, float* v) const {
    int cc, kh, kw;
    cls_tap(t, k0, cc, kh, kw);
    fetch4<true>(p, c.ok ? c.boff + (unsigned)(((kh * KS + kw) * C + cc * 32) * (int)ld * 4) : INVALID_OFF(true), v);
  }
};

// ---- weight gradient by taps (igemm.h TapSched): A[m = tap * Cu + c][k' = p * NP + n] = x[n, ho s - 1 + kh, wo s - 1 + kw, c],
// B[k'][co] = dy[n, ho, wo, co]; only the grid positions at which the tile's tap is inside the image are walked.
// Units: Cu / X count operand ELEMENTS (fp32 engine: channels; bf16 engine: channel pairs); Ct = REAL rows per tap (what the
// kernel's m0 counts).  Both operands are contiguous along M / N (16-byte fetches of 4 elements).
struct WgradXTap {
  static constexpr bool KCONTIG = false;
  static constexpr int VEC = 4;
  struct Ctx { unsigned boff; int ok; };
  const float* x;
  Geom g;                       // g.C in operand elements
  int M, Ct, rows;              // M in operand elements; Ct real rows per tap; rows = images
  unsigned mg_Cu;               // magic of g.C
  igemm::TapSched sc;
  __device__ __forceinline__ Ctx prep(int m) const {
    Ctx c;
    c.ok = m < M;
    c.boff = (unsigned)(m - fast_div(m, mg_Cu) * g.C) * 4u;             // the channel (pair) inside its tap
    return c;
  }
  __device__ __forceinline__ igemm::TapTile tile(int m0) const { return igemm::tap_tile(sc, m0 / Ct); }
  __device__ __forceinline__ void load(const Ctx& c, const igemm::TapTile& t, int, int k0, int kin, float* v) const {
    int ho, wo, n0;
    igemm::tap_pos(t, k0, ho, wo, n0);
    const int hi = ho * g.stride - g.pad + t.kh, wi = wo * g.stride - g.pad + t.kw;       // inside the image by construction
    const unsigned soff = (unsigned)(hi * g.W + wi) * (unsigned)g.C * 4u;                 // scalar
    const int n = n0 + kin;
    fetch4<true>(x, (c.ok && n < rows) ? (unsigned)n * (unsigned)(g.H * g.W * g.C) * 4u + c.boff + soff : INVALID_OFF(true), v);
  }
};
struct ColDyTap {
  static constexpr bool KCONTIG = false;
  static constexpr int VEC = 4;
  struct Ctx { unsigned boff; int ok; };
  const float* dy;
  Geom g;                       // g.Co in operand elements
  int X, Ct, rows;
  igemm::TapSched sc;
  __device__ __forceinline__ Ctx prep(int x) const {
    Ctx c;
    c.ok = x < X;
    c.boff = (unsigned)x * 4u;
    return c;
  }
  __device__ __forceinline__ igemm::TapTile tile(int m0) const { return igemm::tap_tile(sc, m0 / Ct); }
  __device__ __forceinline__ void load(const Ctx& c, const igemm::TapTile& t, int, int k0, int kin, float* v) const {
    int ho, wo, n0;
    igemm::tap_pos(t, k0, ho, wo, n0);
    const unsigned soff = (unsigned)(ho * g.Wo + wo) * (unsigned)g.Co * 4u;
    const int n = n0 + kin;
    fetch4<true>(dy, (c.ok && n < rows) ? (unsigned)n * (unsigned)(g.Ho * g.Wo * g.Co) * 4u + c.boff + soff : INVALID_OFF(true), v);
  }
};

// ---- stem (C = 1): scalar gathers
struct StemFwd {  // rows = output pixels, K = KS*KS (49), element = x[n, ho*2-3+kh, wo*2-3+kw]
  static constexpr bool KCONTIG = true;
  static constexpr int VEC = 1;
  struct Ctx { int base, hi0, wi0; };
  const float* x;
  Geom g;
  int M, K;
  __device__ __forceinline__ Ctx prep(int m) const {
    Ctx c;
    if (m >= M) { c.base = -1; c.hi0 = c.wi0 = 0; return c; }
    const int hw = g.Ho * g.Wo;
    const int n = m / hw, r = m - n * hw;
    const int ho = r / g.Wo, wo = r - ho * g.Wo;
    c.base = n * g.H * g.W;
    c.hi0 = ho * g.stride - g.pad;
    c.wi0 = wo * g.stride - g.pad;
    return c;
  }
  __device__ __forceinline__ void load(const Ctx& c, int, int k0, int kin, float* v) const {
    const int k = k0 + kin;
    const int kh = k / g.KS, kw = k - kh * g.KS;
    const int hi = c.hi0 + kh, wi = c.wi0 + kw;
    v[0] = 0.f;
    if (c.base >= 0 && k < K && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W)
      v[0] = x[c.base + hi * g.W + wi];
  }
};

struct StemWgradX {  // A[m = (kh,kw)][k = output pixel]
  static constexpr bool KCONTIG = false;
  static constexpr int VEC = 1;
  struct Ctx { int kh, kw; };
  const float* x;
  Geom g;
  int M, K;
  __device__ __forceinline__ Ctx prep(int m) const {
    Ctx c;
    if (m >= M) { c.kh = -1; c.kw = 0; return c; }
    c.kh = m / g.KS;
    c.kw = m - c.kh * g.KS;
    return c;
  }
  __device__ __forceinline__ void load(const Ctx& c, int, int k0, int kin, float* v) const {
    const int k = k0 + kin;
    v[0] = 0.f;
    if (c.kh < 0 || k >= K) return;
    const int hw = g.Ho * g.Wo;
    const int n = k / hw, r = k - n * hw;
    const int ho = r / g.Wo, wo = r - ho * g.Wo;
    const int hi = ho * g.stride - g.pad + c.kh, wi = wo * g.stride - g.pad + c.kw;
    if ((unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W) v[0] = x[(n * g.H + hi) * g.W + wi];
  }
};

#undef CONVOP_RETURN

}  // namespace convop
