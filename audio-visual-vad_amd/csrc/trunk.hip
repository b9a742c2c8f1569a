// trunk.hip -- ResNet-18 trunk over gray lip crops: forward and backward.
//
// Replaces `self.features(video).squeeze()` (+ the 3x channel repeat in front of it)
// of the reference: packages/models/Video_Net.py:60-81, packages/models/AV_Net.py:78-94,
// i.e. torchvision.models.resnet18 children [:-1].
//
// Data layout in HBM: activations NHWC fp32 (channel-contiguous: an im2col row
// is a run of 128-byte lines), weights re-packed once per call from the
// state_dict's OIHW into [(kh,kw,c)][co] (forward B operand) and
// [(kh,kw,co)][c] (dgrad B operand).  The reference's 3 identical input
// channels are folded into the stem's weights (sum over c), so the frame is
// read once as a 1-channel image.
//
// Kernels: every convolution (forward, dgrad, wgrad) is the fp32-MFMA
// implicit GEMM of igemm.h; BatchNorm statistics are a two-stage column
// reduction in fp64; normalisation / ReLU / residual add / pooling are fused
// elementwise kernels with 16-byte accesses.
#include "bgemm.h"
#include "conv64.h"

using convop::Geom;

namespace {

#include "bn_kernels.h"

constexpr int NCONV = AVVAD_TRUNK_NCONV;

// ------------------------------------------------------------------ static network description
struct ConvSpec {
  int cin, cout, ks, stride, pad;
  int stage;    // -1 stem
  int role;     // 0 stem, 1 block conv1, 2 block conv2, 3 downsample
};

struct Plan {
  int N, H, W;
  int h[6], w[6];  // 0: input, 1: stem out, 2: pooled / stage0, 3..5: stage1..3
  ConvSpec conv[NCONV];
  Geom geom[NCONV];
  // workspace offsets in floats
  size_t wf[NCONV], wd[NCONV];
  size_t wf16[NCONV], wd16[NCONV];   // bf16 packs of the bf16 data path: [co][(cc,tap,r)] and [c][(cc,tap,r)] (64-channel chunks cc)
  size_t bn_scale, bn_shift, bn_mean, bn_invstd;  // [NCONV][MAXC]
  size_t coef;                                    // [3][MAXC] backward coefficients
  size_t part;                                    // doubles: [STAT_CHUNKS][2][MAXC]
  size_t c0, p0, am;   // am: arg-max position (0..8, one byte per channel) of every pooled stem element
  size_t blk[8][5];  // c1, a1, c2, cd, out
  size_t qm[8];      // bytes (as floats): one ReLU-mask byte per quad of a block's output (save_for_backward)
  size_t G[4], g0, wg[NCONV];   // wg[i]: conv i's packed weight gradient (all 20 kept: ONE unpack launch at the end)
  size_t slab;     // igemm::SLAB_FLOATS: partial tiles of the engine's stream-K round
  size_t total;
};

static void make_plan(const avvad_trunk_desc* d, Plan* p) {
  p->N = d->N; p->H = d->H; p->W = d->W;
  p->h[0] = d->H; p->w[0] = d->W;
  p->h[1] = (d->H + 6 - 7) / 2 + 1; p->w[1] = (d->W + 6 - 7) / 2 + 1;
  p->h[2] = (p->h[1] + 2 - 3) / 2 + 1; p->w[2] = (p->w[1] + 2 - 3) / 2 + 1;
  for (int s = 3; s < 6; ++s) { p->h[s] = (p->h[s - 1] + 2 - 3) / 2 + 1; p->w[s] = (p->w[s - 1] + 2 - 3) / 2 + 1; }
  int i = 0;
  p->conv[i++] = {1, 64, 7, 2, 3, -1, 0};
  int cin = 64;
  const int widths[4] = {64, 128, 256, 512};
  for (int s = 0; s < 4; ++s)
    for (int b = 0; b < 2; ++b) {
      const int c = widths[s], st = (b == 0 && s > 0) ? 2 : 1;
      p->conv[i++] = {cin, c, 3, st, 1, s, 1};
      p->conv[i++] = {c, c, 3, 1, 1, s, 2};
      if (b == 0 && s > 0) p->conv[i++] = {cin, c, 1, 2, 0, s, 3};
      cin = c;
    }
  // geometry
  i = 0;
  p->geom[i++] = {d->N, p->h[0], p->w[0], 1, p->h[1], p->w[1], 64, 7, 2, 3};
  for (int s = 0; s < 4; ++s)
    for (int b = 0; b < 2; ++b) {
      const int hin = (b == 0 && s > 0) ? p->h[s + 1] : p->h[s + 2], win = (b == 0 && s > 0) ? p->w[s + 1] : p->w[s + 2];
      const int ho = p->h[s + 2], wo = p->w[s + 2];
      const ConvSpec c1 = p->conv[i];
      p->geom[i++] = {d->N, hin, win, c1.cin, ho, wo, c1.cout, 3, c1.stride, 1};
      const ConvSpec c2 = p->conv[i];
      p->geom[i++] = {d->N, ho, wo, c2.cin, ho, wo, c2.cout, 3, 1, 1};
      if (b == 0 && s > 0) { const ConvSpec cd = p->conv[i]; p->geom[i++] = {d->N, hin, win, cd.cin, ho, wo, cd.cout, 1, 2, 0}; }
    }
  size_t off = 0;
  auto take = [&](size_t n) { size_t o = off; off += align_up(n, 64); return o; };
  for (i = 0; i < NCONV; ++i) {
    const ConvSpec& c = p->conv[i];
    const size_t n = (size_t)c.ks * c.ks * c.cin * c.cout;
    p->wf[i] = take(n);
    p->wd[i] = (i == 0) ? 0 : take(n);
    p->wf16[i] = (i == 0) ? 0 : take(n / 2);
    p->wd16[i] = (i == 0) ? 0 : take(n / 2);
  }
  p->bn_scale = take(NCONV * MAXC); p->bn_shift = take(NCONV * MAXC);
  p->bn_mean = take(NCONV * MAXC); p->bn_invstd = take(NCONV * MAXC);
  p->coef = take(3 * MAXC);
  p->part = take((size_t)STAT_CHUNKS * 2 * MAXC * 2);  // doubles
  p->slab = take(igemm::SLAB_FLOATS);
  const size_t N = d->N;
  p->c0 = take(N * p->h[1] * p->w[1] * 64);
  p->p0 = take(N * p->h[2] * p->w[2] * 64);
  p->am = take(N * p->h[2] * p->w[2] * 16);     // 4 bytes (one channel quad) per word
  for (int s = 0; s < 4; ++s)
    for (int b = 0; b < 2; ++b) {
      const size_t n = N * p->h[s + 2] * p->w[s + 2] * widths[s];
      for (int j = 0; j < 5; ++j) p->blk[s * 2 + b][j] = (j == 3 && !(b == 0 && s > 0)) ? 0 : take(n);
      p->qm[s * 2 + b] = d->save_for_backward ? take((n / 4 + 3) / 4) : 0;
    }
  if (d->save_for_backward) {
    const size_t gmax = N * p->h[2] * p->w[2] * 64;  // stage0 is the largest block tensor
    for (int j = 0; j < 4; ++j) p->G[j] = take(gmax);
    p->g0 = take(N * p->h[1] * p->w[1] * 64);
    for (int j = 0; j < NCONV; ++j) p->wg[j] = take((size_t)p->conv[j].ks * p->conv[j].ks * p->conv[j].cin * p->conv[j].cout);
  }
  p->total = off;
}

// ------------------------------------------------------------------ weight (un)packing
__global__ void pack_weights(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wd, int Co, int C, int KS) {
  const int n = Co * C * KS * KS;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    // i indexes OIHW
    int r = i;
    const int kw = r % KS; r /= KS;
    const int kh = r % KS; r /= KS;
    const int c = r % C; const int co = r / C;
    const float v = w[i];
    wf[((long)(kh * KS + kw) * C + c) * Co + co] = v;
    if (wd) wd[((long)(kh * KS + kw) * Co + co) * C + c] = v;
  }
}
// all 20 convolutions' weights in ONE launch (19 separate launches of ~9 us each were 0.17 ms of a 22 ms step): the block
// index selects the convolution through a small table of block ranges.  A workgroup owns a 32 (co) x 32 (c) tile of one
// convolution: its OIHW source is 32 contiguous runs of 32*T floats, read coalesced into LDS; each of the T taps then
// leaves as a 32x32 tile whose rows are contiguous in the packed layouts (co for the forward pack, c for the dgrad pack).
// (One thread per element wrote 4-byte scatters a whole channel plane apart: 92 us for 135 MB.)
struct PackTab {
  const float* w[NCONV];
  float* wf[NCONV];
  float* wd[NCONV];      // may be null
  __bf16* wf16[NCONV];   // bf16 data path (may be null): wf16[co][(cc * T + tap) * 64 + r] = w[co][cc * 64 + r][tap]
  __bf16* wd16[NCONV];   //                               wd16[c][(cc * T + tap) * 64 + r] = w[cc * 64 + r][c][tap]
  int cout[NCONV], cin[NCONV], ks[NCONV];
  int blk0[NCONV + 1];   // first block of conv i (conv 0 = stem: its 3 input channels are folded, see pack_stem)
};
constexpr int PACK_LD = 32 * 9 + 1;
__global__ void __launch_bounds__(256) pack_all(const PackTab tab) {
  __shared__ float tl[32 * PACK_LD];
  int i = 0;
#pragma unroll
  for (int j = 1; j < NCONV; ++j) i += (int)blockIdx.x >= tab.blk0[j];      // block-uniform
  const int lb = (int)blockIdx.x - tab.blk0[i];
  const float* w = tab.w[i];
  float* wf = tab.wf[i];
  if (i == 0) {
    const int e = lb * 256 + threadIdx.x;
    if (e >= 64 * 49) return;
    const int co = e / 49, k = e % 49;
    wf[k * 64 + co] = w[(co * 3 + 0) * 49 + k] + w[(co * 3 + 1) * 49 + k] + w[(co * 3 + 2) * 49 + k];
    return;
  }
  const int Co = tab.cout[i], C = tab.cin[i], T = tab.ks[i] * tab.ks[i];     // T = 9 or 1; Co, C multiples of 32
  const int ctiles = C >> 5;
  const int co0 = (lb / ctiles) * 32, c0 = (lb % ctiles) * 32;
  const int run = 32 * T;
  for (int idx = threadIdx.x; idx < 32 * run; idx += 256) {
    const int col = idx / run, r = idx - col * run;
    tl[col * PACK_LD + r] = w[((long)(co0 + col) * C + c0) * T + r];
  }
  __syncthreads();
  float* wd = tab.wd[i];
  __bf16* wf16 = tab.wf16[i];
  __bf16* wd16 = tab.wd16[i];
  if (wf16) {     // the bf16 data path takes only these (K-contiguous rows in the GEMM's own K order; C, Co multiples of 64)
    for (int idx = threadIdx.x; idx < T * 1024; idx += 256) {
      const int tap = idx >> 10, a = (idx >> 5) & 31, b = idx & 31;
      wf16[(long)(co0 + a) * (T * C) + ((c0 >> 6) * T + tap) * 64 + (c0 & 63) + b] = (__bf16)tl[a * PACK_LD + b * T + tap];
      if (wd16) wd16[(long)(c0 + a) * (T * Co) + ((co0 >> 6) * T + tap) * 64 + (co0 & 63) + b] = (__bf16)tl[b * PACK_LD + a * T + tap];
    }
    return;
  }
  for (int idx = threadIdx.x; idx < T * 1024; idx += 256) {
    const int tap = idx >> 10, a = (idx >> 5) & 31, b = idx & 31;
    wf[((long)(tap * C + c0 + a)) * Co + co0 + b] = tl[b * PACK_LD + a * T + tap];           // row (tap, c = a), column co = b
    if (wd) wd[((long)(tap * Co + co0 + a)) * C + c0 + b] = tl[a * PACK_LD + b * T + tap];    // row (tap, co = a), column c = b
  }
}
// stem: fold the 3 identical input channels (Video_Net.py:64): Wf[kh*7+kw][co] = sum_c w[co][c][kh][kw]
__global__ void pack_stem(const float* __restrict__ w, float* __restrict__ wf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 64 * 49) return;
  const int co = i / 49, k = i % 49;
  wf[k * 64 + co] = w[(co * 3 + 0) * 49 + k] + w[(co * 3 + 1) * 49 + k] + w[(co * 3 + 2) * 49 + k];
}
// every convolution's packed gradient -> += the OIHW gradient, ONE launch behind the last weight-gradient GEMM (19 launches
// of ~8 us each were 0.15 ms of the step; the whole trunk is one autograd node, nobody consumes a gradient earlier)
struct UnpackTab {
  const float* pk[NCONV];
  float* dw[NCONV];        // null: this convolution's gradient is not wanted
  int cout[NCONV], cin[NCONV], ks[NCONV];
  int blk0[NCONV + 1];
};
__global__ void __launch_bounds__(256) unpack_all(const UnpackTab tab) {
  __shared__ float tl[32 * PACK_LD];
  int i = 0;
#pragma unroll
  for (int j = 1; j < NCONV; ++j) i += (int)blockIdx.x >= tab.blk0[j];      // block-uniform
  const int lb = (int)blockIdx.x - tab.blk0[i];
  float* dw = tab.dw[i];
  if (!dw) return;
  const float* pk = tab.pk[i];
  if (i == 0) {
    const int e = lb * 256 + threadIdx.x;
    if (e >= 64 * 3 * 49) return;
    dw[e] += pk[(e % 49) * 64 + e / 147];
    return;
  }
  // the mirror image of pack_all: a 32 (co) x 32 (c) tile, T taps read as rows of co, added onto 32 contiguous OIHW runs
  const int Co = tab.cout[i], C = tab.cin[i], T = tab.ks[i] * tab.ks[i];
  const int ctiles = C >> 5;
  const int co0 = (lb / ctiles) * 32, c0 = (lb % ctiles) * 32;
  for (int idx = threadIdx.x; idx < T * 1024; idx += 256) {
    const int tap = idx >> 10, a = (idx >> 5) & 31, b = idx & 31;
    tl[b * PACK_LD + a * T + tap] = pk[((long)(tap * C + c0 + a)) * Co + co0 + b];
  }
  __syncthreads();
  const int run = 32 * T;
  for (int idx = threadIdx.x; idx < 32 * run; idx += 256) {
    const int col = idx / run, r = idx - col * run;
    dw[((long)(co0 + col) * C + c0) * T + r] += tl[col * PACK_LD + r];
  }
}

// ------------------------------------------------------------------ stem convolution 7x7 / 2, 1 -> 64 channels (frames that fit LDS)
// K = 49 is two K tiles of the engine: prologue + epilogue outweighed the loop and it ran at a quarter of the output-write
// rate.  Here one workgroup owns one frame: the zero-padded frame is staged in LDS once (22 KB at 67x67), the 49x64
// weights sit in registers as MFMA B fragments, and a wave walks 32-pixel tiles: the A fragment of k-step ks is ONE
// ds_read_b32 (pixel base + tap offset: the im2col gather happens inside LDS), 25 k-steps x 2 column tiles = 50 MFMAs,
// then 32 coalesced 128-byte row stores.
__global__ void __launch_bounds__(256)
    stem_fwd_mfma(const float* __restrict__ x, const float* __restrict__ wf, float* __restrict__ y, int H, int W, int Ho, int Wo,
                  int LDW, double* __restrict__ stat) {
  // stat (may be null): BatchNorm batch statistics fused -- stat[(n * 2 + 0) * 64 + c] = sum over frame n's pixels of y[.][c],
  // [(n * 2 + 1) * 64 + c] = sum of squares (one chunk per frame; bn_finalize adds the chunks in order)
  __shared__ double red[4 * 64 * 2];
  extern __shared__ float img[];                     // (H + 6) x LDW, zero border of 3
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int PH = H + 6;
  for (int i = threadIdx.x; i < PH * LDW; i += 256) img[i] = 0.f;
  __syncthreads();
  const float* xn = x + (long)n * H * W;
  for (int i = threadIdx.x; i < H * W; i += 256) {
    const int r = i / W, c = i - r * W;
    img[(r + 3) * LDW + c + 3] = xn[i];
  }
  float b0[25], b1[25];
  int toff[25];
#pragma unroll
  for (int ks = 0; ks < 25; ++ks) {
    const int tap = 2 * ks + lh;
    const bool live = tap < 49;
    const int tc = live ? tap : 0;
    b0[ks] = live ? wf[tc * 64 + li] : 0.f;
    b1[ks] = live ? wf[tc * 64 + 32 + li] : 0.f;
    toff[ks] = (tc / 7) * LDW + (tc % 7);
  }
  __syncthreads();
  const int npix = Ho * Wo;
  const int ntile = (npix + 31) >> 5;
  float* yn = y + (long)n * npix * 64;
  float sf0 = 0.f, qf0 = 0.f, sf1 = 0.f, qf1 = 0.f;
  for (int tile = wave; tile < ntile; tile += 4) {
    const int pix = tile * 32 + li;
    const int pc = pix < npix ? pix : 0;              // clamped: rows beyond the frame are computed and dropped
    const int ho = pc / Wo, wo = pc - ho * Wo;
    const int base = 2 * ho * LDW + 2 * wo;           // padded coordinates: input row 2*ho - 3 + 3
    float a[25];
#pragma unroll
    for (int ks = 0; ks < 25; ++ks) a[ks] = img[base + toff[ks]];
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 25; ++ks) {
      acc0 = mfma32(a[ks], b0[ks], acc0);
      acc1 = mfma32(a[ks], b1[ks], acc1);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = tile * 32 + mfma32_row(r, lh);
      if (p < npix) {
        yn[(long)p * 64 + li] = acc0[r];
        yn[(long)p * 64 + 32 + li] = acc1[r];
        sf0 += acc0[r]; qf0 = fmaf(acc0[r], acc0[r], qf0);
        sf1 += acc1[r]; qf1 = fmaf(acc1[r], acc1[r], qf1);
      }
    }
  }
  if (stat == nullptr) return;
  double d[4] = {(double)sf0, (double)qf0, (double)sf1, (double)qf1};
#pragma unroll
  for (int k = 0; k < 4; ++k) d[k] += __shfl_xor(d[k], 32, 64);
  if (lh == 0) {
    red[(wave * 64 + li) * 2 + 0] = d[0]; red[(wave * 64 + li) * 2 + 1] = d[1];
    red[(wave * 64 + 32 + li) * 2 + 0] = d[2]; red[(wave * 64 + 32 + li) * 2 + 1] = d[3];
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int c = threadIdx.x;
    double a = 0, b = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { a += red[(w * 64 + c) * 2 + 0]; b += red[(w * 64 + c) * 2 + 1]; }
    stat[((long)n * 2 + 0) * 64 + c] = a;
    stat[((long)n * 2 + 1) * 64 + c] = b;
  }
}

// stem: p0 = maxpool3x3/2 pad1 ( relu(bn(c0)) ).  Also records, per pooled element, WHICH window position (dh*3+dw) holds
// the maximum -- the first one in row-major scan order, torch's rule (max_pool2d updates on a strict '>') -- so that the
// backward sends the gradient to exactly that element even when flat image regions produce exact ties.
template <bool OB = false>     // OB: p0 is stored as bf16 (the bf16 data path)
__global__ void stem_bn_relu_pool(const float* __restrict__ c0, const float* __restrict__ scale, const float* __restrict__ shift,
                                  float* __restrict__ p0, unsigned* __restrict__ am, int N, int Hc, int Wc, int Hp, int Wp) {
  const long total = (long)N * Hp * Wp * 16;  // 64 channels = 16 quads
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i & 15);
    long r = i >> 4;
    const int pw = (int)(r % Wp); r /= Wp;
    const int ph = (int)(r % Hp); const int n = (int)(r / Hp);
    const float4 sc = *reinterpret_cast<const float4*>(scale + q * 4), sh = *reinterpret_cast<const float4*>(shift + q * 4);
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    unsigned idx[4] = {0, 0, 0, 0};
    for (int dh = 0; dh < 3; ++dh) {
      const int h = ph * 2 - 1 + dh;
      if ((unsigned)h >= (unsigned)Hc) continue;
      for (int dw = 0; dw < 3; ++dw) {
        const int w = pw * 2 - 1 + dw;
        if ((unsigned)w >= (unsigned)Wc) continue;
        const float4 v = *reinterpret_cast<const float4*>(c0 + ((long)(n * Hc + h) * Wc + w) * 64 + q * 4);
        const float y[4] = {fmaxf(bn_affine(v.x, sc.x, sh.x), 0.f), fmaxf(bn_affine(v.y, sc.y, sh.y), 0.f),
                            fmaxf(bn_affine(v.z, sc.z, sh.z), 0.f), fmaxf(bn_affine(v.w, sc.w, sh.w), 0.f)};
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (y[k] > m[k]) { m[k] = y[k]; idx[k] = (unsigned)(dh * 3 + dw); }
      }
    }
    if constexpr (OB) store_quad_bf16(p0, i, m);
    else reinterpret_cast<float4*>(p0)[i] = make_float4(m[0], m[1], m[2], m[3]);
    if (am) am[i] = idx[0] | (idx[1] << 8) | (idx[2] << 16) | (idx[3] << 24);
  }
}

// stem backward of pool+relu: g0[n,h,w,c] = sum of dp0 over the pooled windows whose recorded arg-max is this element,
// masked by the ReLU (an all-zero window's arg-max carries no gradient).
// part (may be null; needs gridDim.x * blockDim.x % 16 == 0): the BatchNorm backward's column sums of the stem, fused -- per
// workgroup sum g0 and sum g0 * xhat per channel (what col_reduce<1> would read c0 and g0 again for), chunk = workgroup
__global__ void __launch_bounds__(256)
    stem_pool_relu_bwd(const float* __restrict__ c0, const float* __restrict__ scale, const float* __restrict__ shift,
                       const unsigned* __restrict__ am, const float* __restrict__ dp0, float* __restrict__ g0,
                       int N, int Hc, int Wc, int Hp, int Wp, const float* __restrict__ mean = nullptr,
                       const float* __restrict__ invstd = nullptr, double* __restrict__ part = nullptr) {
  __shared__ double sm[256 * 8];
  const long total = (long)N * Hc * Wc * 16;
  float s4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};
  float mu[4] = {0.f, 0.f, 0.f, 0.f}, is[4] = {1.f, 1.f, 1.f, 1.f};
  if (part) {
    const int qc = (int)(((long)blockIdx.x * blockDim.x + threadIdx.x) & 15);
#pragma unroll
    for (int k = 0; k < 4; ++k) { mu[k] = mean[qc * 4 + k]; is[k] = invstd[qc * 4 + k]; }
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i & 15);
    long r = i >> 4;
    const int w = (int)(r % Wc); r /= Wc;
    const int h = (int)(r % Hc); const int n = (int)(r / Hc);
    const float4 sc = *reinterpret_cast<const float4*>(scale + q * 4), sh = *reinterpret_cast<const float4*>(shift + q * 4);
    const float4 v = reinterpret_cast<const float4*>(c0)[i];
    const bool pos[4] = {bn_affine(v.x, sc.x, sh.x) > 0.f, bn_affine(v.y, sc.y, sh.y) > 0.f, bn_affine(v.z, sc.z, sh.z) > 0.f,
                         bn_affine(v.w, sc.w, sh.w) > 0.f};
    float g[4] = {0, 0, 0, 0};
    const int ph0 = max(0, h / 2), ph1 = min(Hp - 1, (h + 1) / 2);
    const int pw0 = max(0, w / 2), pw1 = min(Wp - 1, (w + 1) / 2);
    for (int ph = ph0; ph <= ph1; ++ph)
      for (int pw = pw0; pw <= pw1; ++pw) {
        // window of (ph,pw) covers rows 2ph-1..2ph+1
        if (h < 2 * ph - 1 || h > 2 * ph + 1 || w < 2 * pw - 1 || w > 2 * pw + 1) continue;
        const unsigned me = (unsigned)((h - (2 * ph - 1)) * 3 + (w - (2 * pw - 1)));
        const long o = ((long)(n * Hp + ph) * Wp + pw) * 16 + q;
        const unsigned a = am[o];
        const float4 dp = reinterpret_cast<const float4*>(dp0)[o];
        if (pos[0] && (a & 255u) == me) g[0] += dp.x;
        if (pos[1] && ((a >> 8) & 255u) == me) g[1] += dp.y;
        if (pos[2] && ((a >> 16) & 255u) == me) g[2] += dp.z;
        if (pos[3] && (a >> 24) == me) g[3] += dp.w;
      }
    reinterpret_cast<float4*>(g0)[i] = make_float4(g[0], g[1], g[2], g[3]);
    if (part) {
      const float xv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) { s4[k] += g[k]; q4[k] = fmaf(g[k], (xv[k] - mu[k]) * is[k], q4[k]); }
    }
  }
  if (part == nullptr) return;
  // 256 threads = 16 row groups x 16 channel quads (thread t: quad t & 15): fixed-order sums in LDS
#pragma unroll
  for (int k = 0; k < 4; ++k) { sm[threadIdx.x * 8 + k] = (double)s4[k]; sm[threadIdx.x * 8 + 4 + k] = (double)q4[k]; }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int c = threadIdx.x, cq = c >> 2, cj = c & 3;
    double a = 0, b = 0;
    for (int l = 0; l < 16; ++l) { a += sm[(l * 16 + cq) * 8 + cj]; b += sm[(l * 16 + cq) * 8 + 4 + cj]; }
    part[((long)blockIdx.x * 2 + 0) * 64 + c] = a;
    part[((long)blockIdx.x * 2 + 1) * 64 + c] = b;
  }
}

// feat[n][c] = mean over HW of y[n][hw][c]
template <bool IB = false>     // IB: y is bf16 (the bf16 data path)
__global__ void avgpool_fwd(const float* __restrict__ y, float* __restrict__ feat, int N, int HW, int C) {
  const long total = (long)N * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); const long n = i / C;
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += IB ? (float)reinterpret_cast<const __bf16*>(y)[(n * HW + p) * C + c] : y[(n * HW + p) * C + c];
    feat[i] = s / (float)HW;
  }
}
__global__ void avgpool_bwd(const float* __restrict__ dfeat, float* __restrict__ dy, int N, int HW, int C) {
  const long total = (long)N * HW * C;
  const float inv = 1.f / (float)HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); const long n = i / ((long)HW * C);
    dy[i] = dfeat[n * C + c] * inv;
  }
}
__global__ void zero_f32(float* p, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = 0.f;
}

static inline int ew_grid(long n) { long b = (n + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

// ------------------------------------------------------------------ stem weight gradient 7x7 / 2, 1 -> 64 channels
//   pk[k = kh*7 + kw][co] = sum_{n, ho, wo} x[n][2 ho - 3 + kh][2 wo - 3 + kw] * dy[n][ho][wo][co]
// On the engine this 49 x 64 x (N*Ho*Wo) product ran on scalar gathers (198 us, 37 TFLOP/s).  Here a workgroup walks whole
// frames: the zero-padded frame sits in LDS (as in stem_fwd_mfma), the contraction index is the output pixel -- lane half h
// takes pixel (ho, 2 s + h) -- so the A operand of tap (kh, kw) is the LDS word  kh*LDW + kw + 2h  +  (2 ho LDW + 4 s):
// a per-lane constant plus a wave-uniform offset, conflict-free (LDW odd), and the B operand dy[pixel][co] is a coalesced
// row.  Wave (mt, nt) owns taps 32 mt .. 32 mt + 31 (49 live) x channels 32 nt .. 32 nt + 31 in one accumulator for all its
// frames; per-workgroup sums leave through slabs and are added in workgroup order (deterministic).
constexpr int STEM_WH = 17;      // Wo / 2 of the production geometry (34 output columns); the kernel takes Wo <= 34, even
__global__ void __launch_bounds__(256)
    stem_wgrad_mfma(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ slab, int N, int H, int W,
                    int Ho, int Wo, int LDW) {
  extern __shared__ float img[];                     // (H + 6) x LDW, zero border of 3
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int mt = wave >> 1, nt = wave & 1;
  const int tap = mt * 32 + li;
  const int tc = tap < 49 ? tap : 0;                 // rows >= 49 of the accumulator are never stored
  const int abase = (tc / 7) * LDW + (tc % 7) + 2 * lh;
  const int PH = H + 6, wh = Wo >> 1;
  for (int i = threadIdx.x; i < PH * LDW; i += 256) img[i] = 0.f;      // the border stays zero for every frame
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int voff = (lh * 64 + nt * 32 + li) * 4;
  for (int n = blockIdx.x; n < N; n += gridDim.x) {
    __syncthreads();                                 // the previous frame's reads are done
    const float* xn = x + (long)n * H * W;
    for (int i = threadIdx.x; i < H * W; i += 256) {
      const int r = i / W, c = i - r * W;
      img[(r + 3) * LDW + c + 3] = xn[i];
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rdy = brsrc(dy + (long)n * Ho * Wo * 64, Ho * Wo * 64 * 4);
    // dy rows in batches of RB: the NEXT batch (RB x 17 loads) is in flight under this batch's RB x 17 MFMAs (~1.8 us: one
    // row ahead, 0.45 us, was shorter than the memory latency and the kernel ran 150 us instead of the engine's 198)
    constexpr int RB = 4;
    float b0[RB][STEM_WH], b1[RB][STEM_WH];
    auto rowload = [&](int ho0, float (&b)[RB][STEM_WH]) {
#pragma unroll
      for (int rr = 0; rr < RB; ++rr)
#pragma unroll
        for (int sw = 0; sw < STEM_WH; ++sw)
          b[rr][sw] = bload(rdy, (sw < wh && ho0 + rr < Ho) ? voff : BUF_OOB, ((ho0 + rr) * Wo + 2 * sw) * 256);
    };
    auto rowmul = [&](int ho0, const float (&b)[RB][STEM_WH]) {
#pragma unroll
      for (int rr = 0; rr < RB; ++rr) {
        if (ho0 + rr >= Ho) break;
        const float* ap = img + abase + 2 * (ho0 + rr) * LDW;
#pragma unroll
        for (int sw = 0; sw < STEM_WH; ++sw)
          if (sw < wh) acc = mfma32(ap[4 * sw], b[rr][sw], acc);
      }
    };
    rowload(0, b0);
    for (int ho = 0; ho < Ho; ho += 2 * RB) {
      if (ho + RB < Ho) rowload(ho + RB, b1);
      __builtin_amdgcn_sched_barrier(0);
      rowmul(ho, b0);
      if (ho + RB >= Ho) break;
      if (ho + 2 * RB < Ho) rowload(ho + 2 * RB, b0);
      __builtin_amdgcn_sched_barrier(0);
      rowmul(ho + RB, b1);
    }
  }
  float* out = slab + (long)blockIdx.x * (49 * 64);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int t = mt * 32 + mfma32_row(r, lh);
    if (t < 49) out[t * 64 + nt * 32 + li] = acc[r];
  }
}
// pk[k][co] = sum over workgroups of slab[.][k][co], ascending: 32 elements x 8 slab groups per workgroup, combined in order
__global__ void __launch_bounds__(256) stem_wgrad_reduce(const float* __restrict__ slab, int nslab, float* __restrict__ pk) {
  __shared__ float sm[8][32];
  const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + el;                // 49 * 64 = 98 * 32
  float s = 0.f;
#pragma unroll 4
  for (int b = grp; b < nslab; b += 8) s += slab[(long)b * (49 * 64) + i];
  sm[grp][el] = s;
  __syncthreads();
  if (grp != 0) return;
#pragma unroll
  for (int k = 1; k < 8; ++k) s += sm[k][el];
  pk[i] = s;
}

// ------------------------------------------------------------------ conv launchers
static inline bool fits_u31(long n) { return n >= 0 && n < (1L << 31); }
// gathered operands are addressed with 32-bit BYTE offsets (Ctx.boff, fetch4): at most 2^30 floats (4 GiB).  Between the
// buffer form's 2 GiB limit and this one the flat form serves; beyond it the launchers refuse (AVVAD_EINVAL).
static inline bool fits_u30(long n) { return n >= 0 && n < (1L << 30); }
// the im2col gathers keep one tap-validity bit per (kh, kw) in a 32-bit word
static inline bool taps_fit(const Geom& g) { return g.KS * g.KS <= 32; }
// operands below 2 GiB take the buffer-addressed gathers (conv_ops.h "BUF")
static inline bool fits_buf(long n_floats) { return n_floats >= 0 && n_floats < (1L << 29) - 64; }
// the stem's LDS-resident-frame kernels serve this geometry (7x7 / 2, 1 -> 64 channels, a padded frame within 48 KB of LDS)
static inline bool stem_kernel_ok(const Geom& g) {
  const int LDW = (g.W + 6) | 1;
  return g.C == 1 && g.KS == 7 && g.stride == 2 && g.pad == 3 && g.Co == 64 && (size_t)(g.H + 6) * LDW * sizeof(float) <= 48 * 1024 &&
         !avvad_tune().no_stem_kernel;
}
// rows per M tile of the forward GEMM of convolution g (what the fused BatchNorm statistics are laid out by), 0: the stem
static inline int fwd_tile_rows(const Geom& g) {
  if (g.C == 1) return 0;
  if (avvad_tune().bf16 == 1) return 128;                   // the bf16 engine's tiles (bgemm.h)
  return g.Co <= 64 && !avvad_tune().no_tall ? 256 : 128;
}
// most tiles a position-class product may have (every tile is in the stream-K pool: about one split tile per worker, so the
// fix-up's traffic grows with the tile count; option "cls_cap" overrides)
static inline long cls_tile_cap() {
  if (avvad_tune().cls_cap > 0) return avvad_tune().cls_cap;
  return 4L * ((avvad_tune().max_cus > 0 && avvad_tune().max_cus < 256) ? avvad_tune().max_cus : 256);     // (measured: 2/CU -> 4/CU: step -0.07 ms)
}
// ---- position classes (igemm.h): 3x3 / pad 1 convolutions whose tile count fits the stream-K pool skip the zero padding
static inline bool cls_common(const Geom& g, float* slab) {
  return avvad_tune().bf16 == 0 && !avvad_tune().no_cls && slab && g.KS == 3 && g.pad == 1 && g.N >= 128 && g.C % 32 == 0 &&
         g.Co % 32 == 0 && g.Ho >= 2 && g.Wo >= 2 && avvad_tune().igemm_variant < 0 && !avvad_tune().no_buf &&
         avvad_tune().no_streamk == 0;
}
static inline bool conv_fwd_cls_ok(const Geom& g, float* slab) {
  if (!cls_common(g, slab) || g.Co % 4 || g.Co < 128) return false;
  const long tiles = (long)g.Ho * g.Wo * cdiv(g.N, 128) * cdiv(g.Co, 128);
  return tiles <= cls_tile_cap() &&
         fits_buf((long)g.N * g.H * g.W * g.C) && fits_buf(9L * g.C * g.Co);
}
static inline bool conv_dgrad_cls_ok(const Geom& g, float* slab) {
  // (stride 2: the parity classes as position classes of ONE product, igemm::ClassSched::s2, instead of four accumulating launches
  //  over a zero-filled dx; option no_s2_cls keeps the four launches)
  if (!cls_common(g, slab) || (g.stride != 1 && (g.stride != 2 || avvad_tune().no_s2_cls)) || g.C % 4 || g.C < 128) return false;
  const long tiles = (long)g.H * g.W * cdiv(g.N, 128) * cdiv(g.C, 128);
  return tiles <= cls_tile_cap() &&
         fits_buf((long)g.N * g.Ho * g.Wo * g.Co) && fits_buf(9L * g.C * g.Co);
}
static int conv_fwd_cls(const float* x, const float* wf, float* y, const Geom& g, hipStream_t s, float* slab, double* stat) {
  const int MB = cdiv(g.N, 128), P = g.Ho * g.Wo;
  const int lsh = ((g.Ho - 1) * g.stride - g.pad + 2 > g.H - 1) ? 1 : 0, lsw = ((g.Wo - 1) * g.stride - g.pad + 2 > g.W - 1) ? 1 : 0;
  const igemm::ClassSched sc{g.Ho, g.Wo, cdiv(g.Co, 128), g.C / 32, MB, 3, 1, lsh, 1, lsw};
  const igemm::ClassRow cr{P, convop::div_magic(P)};
  convop::Im2colFwdCls a{x, g, g.N, cr, sc};
  convop::ColTapRowsCls b{wf, g.Co, g.Co, g.C, 3, 0, cr, sc};
  igemm::EpiCls e{y, (long)P * g.Co, nullptr, 0};
  e.stat = stat; e.W = g.Co; e.rows = g.N; e.cr = cr; e.sched = sc;
  return igemm::launch_cls(a, b, e, MB * P * 128, g.Co, s, slab);
}
static int conv_dgrad_cls(const float* dy, const float* wd, float* dx, const Geom& g, int accumulate, hipStream_t s, float* slab) {
  const int MB = cdiv(g.N, 128), P = g.H * g.W;
  igemm::ClassSched sc{g.H, g.W, cdiv(g.C, 128), g.Co / 32, MB, 3, 1, 1, 1, 1};
  if (g.stride == 2) { sc.s2 = 1; sc.Hq = g.Ho; sc.Wq = g.Wo; }
  const igemm::ClassRow cr{P, convop::div_magic(P)};
  convop::Im2colDgradCls a{dy, g, g.N, cr, sc};
  convop::ColTapRowsCls b{wd, g.C, g.C, g.Co, 3, 1, cr, sc};
  igemm::EpiCls e{dx, (long)P * g.C, nullptr, accumulate ? 1 : 0};
  e.W = g.C; e.rows = g.N; e.cr = cr; e.sched = sc;
  return igemm::launch_cls(a, b, e, MB * P * 128, g.C, s, slab);
}
// ---- the 64 -> 64 channel 3x3 / 1 / 1 convolutions (ResNet layer1): weights-stationary kernel, conv64.h
static inline int conv64_cus() { return (avvad_tune().max_cus > 0 && avvad_tune().max_cus < 256) ? avvad_tune().max_cus : 256; }
static inline bool conv64_ok(const Geom& g) {
  const long M = (long)g.N * g.H * g.W;
  return avvad_tune().bf16 == 0 && !avvad_tune().no_conv64 && g.C == 64 && g.Co == 64 && g.KS == 3 && g.stride == 1 && g.pad == 1 &&
         g.Ho == g.H && g.Wo == g.W && M > 0 && fits_buf(M * 64) && (unsigned long)(M + 64) * (unsigned long)(g.H * g.W) < 0x100000000ull;
}
// y = conv(x) (flip = false, wpk = forward pack) or dx (+)= dgrad(dy) (flip = true, wpk = dgrad pack)
static int conv64_launch(bool flip, const float* x, const float* wpk, float* y, const Geom& g, int accumulate, hipStream_t s, double* stat) {
  const int M = g.N * g.H * g.W;
  const int grid = conv64::grid_for(M, conv64_cus());
  const unsigned mg_hw = convop::div_magic((unsigned)(g.H * g.W)), mg_w = convop::div_magic((unsigned)g.W);
  if (flip)
    hipLaunchKernelGGL(conv64::kernel<true>, dim3(grid), dim3(conv64::NW * 64), 0, s, x, wpk, y, M, g.H, g.W, mg_hw, mg_w, accumulate, stat);
  else
    hipLaunchKernelGGL(conv64::kernel<false>, dim3(grid), dim3(conv64::NW * 64), 0, s, x, wpk, y, M, g.H, g.W, mg_hw, mg_w, accumulate, stat);
  AVVAD_LAUNCH_CHECK();
  return AVVAD_OK;
}
// the same two products on the bf16 data path (bf16 activations / packs in, fp32 out): conv64::kernel16
static inline bool conv64_16_ok(const Geom& g) {
  const long M = (long)g.N * g.H * g.W;
  return !avvad_tune().no_conv64 && g.C == 64 && g.Co == 64 && g.KS == 3 && g.stride == 1 && g.pad == 1 &&
         g.Ho == g.H && g.Wo == g.W && M > 0 && fits_buf(M * 64) && (unsigned long)(M + 64) * (unsigned long)(g.H * g.W) < 0x100000000ull;
}
static int conv64_16_launch(bool flip, const float* x16, const float* wpk16, float* y, const Geom& g, int accumulate, hipStream_t s, double* stat) {
  const int M = g.N * g.H * g.W;
  const int grid = conv64::grid_for(M, conv64_cus());
  const unsigned mg_hw = convop::div_magic((unsigned)(g.H * g.W)), mg_w = convop::div_magic((unsigned)g.W);
  if (flip)
    hipLaunchKernelGGL(conv64::kernel16<true>, dim3(grid), dim3(conv64::NW * 64), 0, s, x16, wpk16, y, M, g.H, g.W, mg_hw, mg_w, accumulate, stat);
  else
    hipLaunchKernelGGL(conv64::kernel16<false>, dim3(grid), dim3(conv64::NW * 64), 0, s, x16, wpk16, y, M, g.H, g.W, mg_hw, mg_w, accumulate, stat);
  AVVAD_LAUNCH_CHECK();
  return AVVAD_OK;
}
// number of partial-sum chunks the forward of convolution g leaves in its `stat` buffer (one per M tile of its GEMM)
static inline int fwd_stat_chunks(const Geom& g, float* slab);

// stat: per-M-tile column sums / sums of squares of y (igemm::EpiStore::stat), or null
template <bool BUF>
static int conv_fwd_t(const float* x, const float* wf, float* y, const Geom& g, hipStream_t s, float* slab, double* stat) {
  const int M = g.N * g.Ho * g.Wo, K = g.KS * g.KS * g.C;
  igemm::EpiStore e{y, g.Co, nullptr, 0};
  e.stat = g.C == 1 ? nullptr : stat;       // (the stem's own kernel takes `stat` directly, see below)
  if (g.C == 1) {
    const int LDW = (g.W + 6) | 1;
    const size_t lds = (size_t)(g.H + 6) * LDW * sizeof(float);
    if (g.KS == 7 && g.stride == 2 && g.pad == 3 && g.Co == 64 && lds <= 48 * 1024 && !avvad_tune().no_stem_kernel) {
      hipLaunchKernelGGL(stem_fwd_mfma, dim3(g.N), dim3(256), lds, s, x, wf, y, g.H, g.W, g.Ho, g.Wo, LDW, stat);
      AVVAD_LAUNCH_CHECK();
      return AVVAD_OK;
    }
    igemm::ColPlain<4> b{wf, g.Co, g.Co, K, 0};
    convop::StemFwd a{x, g, M, K};
    return igemm::launch<128, 64>(a, b, e, M, g.Co, K, 1, s, slab);
  }
  if (g.C % 32 || g.Co % 4 || !taps_fit(g)) return AVVAD_EINVAL;
  if (!fits_u30((long)g.N * g.H * g.W * g.C) || !fits_u30((long)K * g.Co)) return AVVAD_EINVAL;   // 32-bit BYTE offsets in the gathers
  const int T = g.KS * g.KS;
  convop::ColTapRows<BUF> b{wf, g.Co, g.Co, K, g.C, T, convop::div_magic(T)};
  convop::Im2colFwd<BUF> a{x, g, M, convop::tap_div(T, g.KS)};
  if (g.Co <= 64) return avvad_tune().no_tall ? igemm::launch<128, 64>(a, b, e, M, g.Co, K, 1, s, slab) : igemm::launch<256, 64>(a, b, e, M, g.Co, K, 1, s, slab);
  return igemm::launch<128, 128>(a, b, e, M, g.Co, K, 1, s, slab);
}
static inline bool conv_fwd16_cls_ok(const Geom& g, float* slab);
static inline int fwd_stat_chunks(const Geom& g, float* slab) {
  if (avvad_tune().bf16 == 1 && conv64_16_ok(g)) return conv64::grid_for((long)g.N * g.H * g.W, conv64_cus());      // one chunk per workgroup
  if (avvad_tune().bf16 == 1)
    return (g.C % 64 == 0 && g.Co % 64 == 0 && conv_fwd16_cls_ok(g, slab)) ? g.Ho * g.Wo * cdiv(g.N, 128) : cdiv((long)g.N * g.Ho * g.Wo, 128);
  if (conv64_ok(g)) return conv64::grid_for((long)g.N * g.H * g.W, conv64_cus());      // one chunk per workgroup
  if (conv_fwd_cls_ok(g, slab)) return g.Ho * g.Wo * cdiv(g.N, 128);
  const int rows = fwd_tile_rows(g);
  return rows ? cdiv((long)g.N * g.Ho * g.Wo, rows) : 0;
}
static int conv_fwd(const float* x, const float* wf, float* y, const Geom& g, hipStream_t s, float* slab, double* stat = nullptr) {
  if (conv64_ok(g)) return conv64_launch(false, x, wf, y, g, 0, s, stat);
  if (conv_fwd_cls_ok(g, slab)) return conv_fwd_cls(x, wf, y, g, s, slab, stat);
  const bool buf = fits_buf((long)g.N * g.H * g.W * g.C) && fits_buf((long)g.KS * g.KS * g.C * g.Co) && !avvad_tune().no_buf;
  return buf ? conv_fwd_t<true>(x, wf, y, g, s, slab, stat) : conv_fwd_t<false>(x, wf, y, g, s, slab, stat);
}
// dx (+)= dgrad
template <bool BUF>
static int conv_dgrad_t(const float* dy, const float* wd, float* dx, const Geom& g, int accumulate, hipStream_t s, float* slab) {
  const int M = g.N * g.H * g.W, K = g.KS * g.KS * g.Co;
  if (g.Co % 32 || g.C % 4 || !taps_fit(g)) return AVVAD_EINVAL;
  if (!fits_u30((long)g.N * g.Ho * g.Wo * g.Co) || !fits_u30((long)K * g.C)) return AVVAD_EINVAL;   // 32-bit BYTE offsets in the gathers
  if (g.stride == 2) {
    // parity-class decomposition (conv_ops.h): 4 small GEMMs over live taps only, accumulating into dx
    if (!accumulate) hipLaunchKernelGGL(zero_f32, dim3(ew_grid((long)M * g.C)), dim3(256), 0, s, dx, (long)M * g.C);
    for (int ph = 0; ph < 2; ++ph)
      for (int pw = 0; pw < 2; ++pw) {
        convop::S2Class c;
        c.ph = ph; c.pw = pw;
        c.Hc = (g.H - ph + 1) / 2; c.Wc = (g.W - pw + 1) / 2;
        c.kh0 = (ph + g.pad) & 1; c.kw0 = (pw + g.pad) & 1;
        c.nkh = c.kh0 < g.KS ? (g.KS - c.kh0 + 1) / 2 : 0;
        c.nkw = c.kw0 < g.KS ? (g.KS - c.kw0 + 1) / 2 : 0;
        c.oh = (ph + g.pad - c.kh0) / 2; c.ow = (pw + g.pad - c.kw0) / 2;
        const int Mc = g.N * c.Hc * c.Wc, ntap = c.nkh * c.nkw;
        if (Mc <= 0 || ntap <= 0) continue;
        if (ntap > 4) return AVVAD_EINVAL;
        c.mg_ntap = convop::div_magic(ntap); c.mg_nkw = convop::div_magic(c.nkw);
        convop::Im2colDgradS2<BUF> a{dy, g, c, Mc};
        convop::ColSegRows<BUF> b{wd, g.C, g.C, ntap * g.Co, ntap, {0, 0, 0, 0}, convop::div_magic(ntap)};
        for (int ia = 0; ia < c.nkh; ++ia)
          for (int ib = 0; ib < c.nkw; ++ib)
            b.rowbase[ia * c.nkw + ib] = ((c.kh0 + 2 * ia) * g.KS + (c.kw0 + 2 * ib)) * g.Co;
        convop::EpiS2 e{dx, g.C, nullptr, 1, 1, g.H, g.W, c.Hc, c.Wc, ph, pw, convop::div_magic(c.Hc * c.Wc), convop::div_magic(c.Wc)};
        if ((unsigned long)(Mc + 128) * (unsigned long)(c.Hc * c.Wc) >= 0x100000000ull) return AVVAD_EINVAL;   // fast_div range
        int rc;
        if (g.C <= 64) rc = igemm::launch<128, 64>(a, b, e, Mc, g.C, ntap * g.Co, 1, s, slab);
        else rc = igemm::launch<128, 128>(a, b, e, Mc, g.C, ntap * g.Co, 1, s, slab);
        if (rc) return rc;
      }
    return AVVAD_OK;
  }
  igemm::EpiStore e{dx, g.C, nullptr, accumulate ? 1 : 0};
  const int T = g.KS * g.KS;
  convop::ColTapRows<BUF> b{wd, g.C, g.C, K, g.Co, T, convop::div_magic(T)};
  convop::Im2colDgrad<BUF> a{dy, g, M, convop::tap_div(T, g.KS)};
  if (g.C <= 64) return avvad_tune().no_tall ? igemm::launch<128, 64>(a, b, e, M, g.C, K, 1, s, slab) : igemm::launch<256, 64>(a, b, e, M, g.C, K, 1, s, slab);
  return igemm::launch<128, 128>(a, b, e, M, g.C, K, 1, s, slab);
}
static int conv_dgrad(const float* dy, const float* wd, float* dx, const Geom& g, int accumulate, hipStream_t s, float* slab) {
  if (conv64_ok(g)) return conv64_launch(true, dy, wd, dx, g, accumulate, s, nullptr);
  if (conv_dgrad_cls_ok(g, slab)) return conv_dgrad_cls(dy, wd, dx, g, accumulate, s, slab);
  const bool buf = fits_buf((long)g.N * g.Ho * g.Wo * g.Co) && fits_buf((long)g.KS * g.KS * g.C * g.Co) && !avvad_tune().no_buf;
  return buf ? conv_dgrad_t<true>(dy, wd, dx, g, accumulate, s, slab) : conv_dgrad_t<false>(dy, wd, dx, g, accumulate, s, slab);
}
// pk[(kh,kw,c)][co] = wgrad (overwritten; tiles split along K are combined by the engine's fix-up kernel, in a fixed order)
template <bool BUF>
static int conv_wgrad_t(const float* x, const float* dy, float* pk, const Geom& g, hipStream_t s, float* slab) {
  const int M = g.KS * g.KS * g.C, K = g.N * g.Ho * g.Wo;
  igemm::ColPlain<4, BUF> b{dy, g.Co, g.Co, K, 0};
  const int ktiles = cdiv(K, igemm::BK);
  if (g.C == 1) {
    const int LDW = (g.W + 6) | 1;
    const size_t lds = (size_t)(g.H + 6) * LDW * sizeof(float);
    if (g.KS == 7 && g.stride == 2 && g.pad == 3 && g.Co == 64 && (g.Wo & 1) == 0 && g.Wo <= 2 * STEM_WH && lds <= 48 * 1024 &&
        slab && !avvad_tune().no_stem_kernel && (long)g.Ho * g.Wo * 64 * 4 < (1L << 31)) {
      const int nb = g.N < 256 ? g.N : 256;                       // one workgroup per CU, whole frames each
      hipLaunchKernelGGL(stem_wgrad_mfma, dim3(nb), dim3(256), lds, s, x, dy, slab, g.N, g.H, g.W, g.Ho, g.Wo, LDW);
      hipLaunchKernelGGL(stem_wgrad_reduce, dim3(49 * 64 / 32), dim3(256), 0, s, slab, nb, pk);
      AVVAD_LAUNCH_CHECK();
      return AVVAD_OK;
    }
    igemm::EpiStore e{pk, g.Co, nullptr, 0};
    convop::StemWgradX a{x, g, M, K};
    int split = 1024; if (split > ktiles) split = ktiles;
    return igemm::launch<64, 64>(a, b, e, M, g.Co, K, split, s, slab);
  }
  if (g.C % 32) return AVVAD_EINVAL;
  convop::EpiWgrad e{pk, g.Co, nullptr, 0, 1, g.C, g.KS * g.KS, convop::div_magic(g.KS * g.KS)};
  if ((unsigned long)(K + igemm::BK) * (unsigned long)(g.Ho * g.Wo) >= 0x100000000ull) return AVVAD_EINVAL;   // fast_div range
  if (!fits_u30((long)g.N * g.H * g.W * g.C) || !fits_u30((long)K * g.Co)) return AVVAD_EINVAL;                // 32-bit BYTE offsets in the gathers
  convop::WgradX<BUF> a{x, g, M, K, convop::div_magic(g.Ho * g.Wo), convop::div_magic(g.Wo)};
  const bool small = g.Co <= 64;
  const int nb = cdiv(M, 128) * cdiv(g.Co, small ? 64 : 128);
  int split = cdiv(1024, nb); if (split > ktiles) split = ktiles;
  if (small) return igemm::launch<128, 64>(a, b, e, M, g.Co, K, split, s, slab);
  return igemm::launch<128, 128>(a, b, e, M, g.Co, K, split, s, slab);
}
// weight gradient by taps (igemm.h TapSched): only the grid positions at which a tap is inside the image are contracted
static inline bool conv_wgrad_tap_ok(const Geom& g, float* slab) {
  if (!cls_common(g, slab) || g.C % 128 || g.Co % 4 || g.Co < 128) return false;
  return 9L * (g.C / 128) * cdiv(g.Co, 128) <= cls_tile_cap() &&
         fits_buf((long)g.N * g.H * g.W * g.C) && fits_buf((long)g.N * g.Ho * g.Wo * g.Co);
}
static inline igemm::TapSched tap_sched(const Geom& g, int tiles_per_tap) {
  const int NP = cdiv(g.N, 128) * 128;
  const int lsh = ((g.Ho - 1) * g.stride - g.pad + 2 > g.H - 1) ? 1 : 0, lsw = ((g.Wo - 1) * g.stride - g.pad + 2 > g.W - 1) ? 1 : 0;
  return igemm::TapSched{g.Ho, g.Wo, tiles_per_tap, NP / 32, 3, 1, lsh, 1, lsw};
}
static int conv_wgrad_tap(const float* x, const float* dy, float* pk, const Geom& g, hipStream_t s, float* slab) {
  const int M = 9 * g.C;
  const igemm::TapSched sc = tap_sched(g, (g.C / 128) * cdiv(g.Co, 128));
  convop::WgradXTap a{x, g, M, g.C, g.N, convop::div_magic(g.C), sc};
  convop::ColDyTap b{dy, g, g.Co, g.C, g.N, sc};
  igemm::EpiTap e{pk, g.Co, nullptr, 0};
  e.Mrows = M; e.sched = sc;
  return igemm::launch_cls(a, b, e, M, g.Co, s, slab);
}
// the 64 -> 64 channel 3x3 convolutions: output-stationary kernel + ordered reduction of the per-CU partials (conv64.h)
static inline bool conv64_wgrad_ok(const Geom& g, float* slab) {
  return conv64_ok(g) && slab && g.W <= conv64::WG_MAXW && (size_t)conv64_cus() * conv64::WG_PART <= igemm::SLAB_FLOATS &&
         (long)g.N * g.H * g.W * 256 < (1L << 31);
}
static int conv64_wgrad(const float* x, const float* dy, float* pk, const Geom& g, hipStream_t s, float* slab) {
  const int NR = g.N * g.H;
  const int grid = NR < conv64_cus() ? NR : conv64_cus();
  hipLaunchKernelGGL(conv64::wgrad_kernel, dim3(grid), dim3(conv64::WG_WAVES * 64), 0, s, x, dy, slab, NR, g.H, g.W,
                     convop::div_magic((unsigned)g.H));
  hipLaunchKernelGGL(conv64::wgrad_reduce, dim3(conv64::WG_PART / 256), dim3(256), 0, s, (const float*)slab, grid, pk);
  AVVAD_LAUNCH_CHECK();
  return AVVAD_OK;
}
static int conv_wgrad(const float* x, const float* dy, float* pk, const Geom& g, hipStream_t s, float* slab) {
  if (conv64_wgrad_ok(g, slab)) return conv64_wgrad(x, dy, pk, g, s, slab);
  if (conv_wgrad_tap_ok(g, slab)) return conv_wgrad_tap(x, dy, pk, g, s, slab);
  const bool buf = g.C > 1 && fits_buf((long)g.N * g.H * g.W * g.C) && fits_buf((long)g.N * g.Ho * g.Wo * g.Co) && !avvad_tune().no_buf;
  return buf ? conv_wgrad_t<true>(x, dy, pk, g, s, slab) : conv_wgrad_t<false>(x, dy, pk, g, s, slab);
}

// ------------------------------------------------------------------ the bf16 data path (option "bf16" = 1; bgemm.h)
// Operands are bf16 in HBM: activations / output gradients NHWC bf16 (written by the BatchNorm elementwise kernels), weights
// in the K-contiguous bf16 packs of pack_all.  Outputs are fp32 (raw convolution sums feed BatchNorm statistics; gradients
// accumulate).  Channel counts must be multiples of 64 (every trunk convolution behind the stem), operands < 2 GiB.
static inline bool native_bf16() { return avvad_tune().bf16 == 1; }
static inline bool bf16_conv_ok(const Geom& g) {
  return g.C % 64 == 0 && g.Co % 64 == 0 && taps_fit(g) && fits_buf((long)g.N * g.H * g.W * g.C / 2) &&
         fits_buf((long)g.N * g.Ho * g.Wo * g.Co / 2) && fits_buf((long)g.KS * g.KS * g.C * g.Co / 2);
}
// position classes on the bf16 engine (the zero padding skipped: igemm.h); 64-channel chunks
// (the bf16 kernels are bound by operand traffic, not by the matrix pipe: skipped products buy time only where they are 40 %
//  of the tile -- the 3x3 grid -- and on larger grids the all-tiles-in-the-pool fix-up costs more than they save: measured
//  per forward launch, 9x9 grid 55 -> 74 us, 5x5 66 -> 73 us, 3x3 78 -> 65 us)
static inline bool cls16_common(const Geom& g, float* slab) {
  return !avvad_tune().no_cls && slab && g.KS == 3 && g.pad == 1 && g.N >= 128 && g.Ho >= 2 && g.Wo >= 2 && g.Ho * g.Wo <= 9 &&
         avvad_tune().no_streamk == 0;
}
static inline bool conv_fwd16_cls_ok(const Geom& g, float* slab) {
  if (!cls16_common(g, slab) || g.Co < 128) return false;
  return (long)g.Ho * g.Wo * cdiv(g.N, 128) * cdiv(g.Co, 128) <= cls_tile_cap();
}
static inline bool conv_dgrad16_cls_ok(const Geom& g, float* slab) {
  if (!cls16_common(g, slab) || g.stride != 1 || g.C < 128) return false;
  return (long)g.H * g.W * cdiv(g.N, 128) * cdiv(g.C, 128) <= cls_tile_cap();
}
static int conv_fwd16(const float* x16, const float* wf16, float* y, const Geom& g, hipStream_t s, float* slab, double* stat) {
  if (!bf16_conv_ok(g)) return AVVAD_EINVAL;
  if (conv64_16_ok(g)) return conv64_16_launch(false, x16, wf16, y, g, 0, s, stat);
  if (conv_fwd16_cls_ok(g, slab)) {
    const int MB = cdiv(g.N, 128), P = g.Ho * g.Wo;
    const int lsh = ((g.Ho - 1) * g.stride - g.pad + 2 > g.H - 1) ? 1 : 0, lsw = ((g.Wo - 1) * g.stride - g.pad + 2 > g.W - 1) ? 1 : 0;
    const igemm::ClassSched sc{g.Ho, g.Wo, cdiv(g.Co, 128), g.C / 64, MB, 3, 1, lsh, 1, lsw};
    const igemm::ClassRow cr{P, convop::div_magic(P)};
    Geom gp = g;
    gp.C = g.C / 2;
    convop::Im2colFwdCls a{x16, gp, g.N, cr, sc};
    bgemm::RowPairsCls b{wf16, 9 * g.C / 2, g.Co, 9, 3, 0, cr, sc};
    igemm::EpiCls e{y, (long)P * g.Co, nullptr, 0};
    e.stat = stat; e.W = g.Co; e.rows = g.N; e.cr = cr; e.sched = sc;
    return bgemm::launch_cls<false>(a, b, e, MB * P * 128, g.Co, s, slab);
  }
  const int M = g.N * g.Ho * g.Wo, T = g.KS * g.KS, Kp = T * g.C / 2;
  Geom gp = g;
  gp.C = g.C / 2;                                            // the gather addresses bf16 PAIRS
  convop::Im2colFwd<true> a{x16, gp, M, convop::tap_div(T, g.KS)};
  bgemm::RowPairs b{wf16, Kp, g.Co, Kp};
  igemm::EpiStore e{y, g.Co, nullptr, 0};
  e.stat = stat;
  if (g.Co <= 64) return bgemm::launch<128, 64, false>(a, b, e, M, g.Co, Kp, s, slab);
  return bgemm::launch<128, 128, false>(a, b, e, M, g.Co, Kp, s, slab);
}
static int conv_dgrad16(const float* dy16, const float* wd16, float* dx, const Geom& g, int accumulate, hipStream_t s, float* slab) {
  if (!bf16_conv_ok(g)) return AVVAD_EINVAL;
  if (conv64_16_ok(g)) return conv64_16_launch(true, dy16, wd16, dx, g, accumulate, s, nullptr);
  const int M = g.N * g.H * g.W, T = g.KS * g.KS, Kp = T * g.Co / 2;
  Geom gp = g;
  gp.Co = g.Co / 2;
  // stride 2, 3x3: the four parity classes as position classes of ONE product (igemm::ClassSched::s2) -- on this engine the four
  // separate launches are ~40 us each whatever their size, so grouping pays on every grid (unlike the stride-1 classes)
  if (g.stride == 2 && g.KS == 3 && g.pad == 1 && !avvad_tune().no_cls && !avvad_tune().no_s2_cls && avvad_tune().no_streamk == 0 && slab &&
      g.N >= 128 && g.C >= 128 && (long)g.H * g.W * cdiv(g.N, 128) * cdiv(g.C, 128) <= cls_tile_cap()) {
    const int MB = cdiv(g.N, 128), P = g.H * g.W;
    igemm::ClassSched sc{g.H, g.W, cdiv(g.C, 128), g.Co / 64, MB, 3, 1, 1, 1, 1};
    sc.s2 = 1; sc.Hq = g.Ho; sc.Wq = g.Wo;
    const igemm::ClassRow cr{P, convop::div_magic(P)};
    convop::Im2colDgradCls a{dy16, gp, g.N, cr, sc};
    bgemm::RowPairsCls b{wd16, Kp, g.C, 9, 3, 1, cr, sc};
    igemm::EpiCls e{dx, (long)P * g.C, nullptr, accumulate ? 1 : 0};
    e.W = g.C; e.rows = g.N; e.cr = cr; e.sched = sc;
    return bgemm::launch_cls<false>(a, b, e, MB * P * 128, g.C, s, slab);
  }
  if (g.stride == 2) {
    if (!accumulate) hipLaunchKernelGGL(zero_f32, dim3(ew_grid((long)M * g.C)), dim3(256), 0, s, dx, (long)M * g.C);
    for (int ph = 0; ph < 2; ++ph)
      for (int pw = 0; pw < 2; ++pw) {
        convop::S2Class c;
        c.ph = ph; c.pw = pw;
        c.Hc = (g.H - ph + 1) / 2; c.Wc = (g.W - pw + 1) / 2;
        c.kh0 = (ph + g.pad) & 1; c.kw0 = (pw + g.pad) & 1;
        c.nkh = c.kh0 < g.KS ? (g.KS - c.kh0 + 1) / 2 : 0;
        c.nkw = c.kw0 < g.KS ? (g.KS - c.kw0 + 1) / 2 : 0;
        c.oh = (ph + g.pad - c.kh0) / 2; c.ow = (pw + g.pad - c.kw0) / 2;
        const int Mc = g.N * c.Hc * c.Wc, ntap = c.nkh * c.nkw;
        if (Mc <= 0 || ntap <= 0) continue;
        if (ntap > 4) return AVVAD_EINVAL;
        c.mg_ntap = convop::div_magic(ntap); c.mg_nkw = convop::div_magic(c.nkw);
        convop::Im2colDgradS2<true> a{dy16, gp, c, Mc};
        bgemm::RowPairsSeg b{wd16, Kp, g.C, ntap * g.Co / 2, T, ntap, {0, 0, 0, 0}, convop::div_magic(ntap)};
        for (int ia = 0; ia < c.nkh; ++ia)
          for (int ib = 0; ib < c.nkw; ++ib) b.tap[ia * c.nkw + ib] = (c.kh0 + 2 * ia) * g.KS + (c.kw0 + 2 * ib);
        convop::EpiS2 e{dx, g.C, nullptr, 1, 1, g.H, g.W, c.Hc, c.Wc, ph, pw, convop::div_magic(c.Hc * c.Wc), convop::div_magic(c.Wc)};
        if ((unsigned long)(Mc + 128) * (unsigned long)(c.Hc * c.Wc) >= 0x100000000ull) return AVVAD_EINVAL;   // fast_div range
        int rc;
        if (g.C <= 64) rc = bgemm::launch<128, 64, false>(a, b, e, Mc, g.C, ntap * g.Co / 2, s, slab);
        else rc = bgemm::launch<128, 128, false>(a, b, e, Mc, g.C, ntap * g.Co / 2, s, slab);
        if (rc) return rc;
      }
    return AVVAD_OK;
  }
  if (conv_dgrad16_cls_ok(g, slab)) {
    const int MB = cdiv(g.N, 128), P = g.H * g.W;
    const igemm::ClassSched sc{g.H, g.W, cdiv(g.C, 128), g.Co / 64, MB, 3, 1, 1, 1, 1};
    const igemm::ClassRow cr{P, convop::div_magic(P)};
    convop::Im2colDgradCls a{dy16, gp, g.N, cr, sc};
    bgemm::RowPairsCls b{wd16, Kp, g.C, 9, 3, 1, cr, sc};
    igemm::EpiCls e{dx, (long)P * g.C, nullptr, accumulate ? 1 : 0};
    e.W = g.C; e.rows = g.N; e.cr = cr; e.sched = sc;
    return bgemm::launch_cls<false>(a, b, e, MB * P * 128, g.C, s, slab);
  }
  convop::Im2colDgrad<true> a{dy16, gp, M, convop::tap_div(T, g.KS)};
  bgemm::RowPairs b{wd16, Kp, g.C, Kp};
  igemm::EpiStore e{dx, g.C, nullptr, accumulate ? 1 : 0};
  if (g.C <= 64) return bgemm::launch<128, 64, false>(a, b, e, M, g.C, Kp, s, slab);
  return bgemm::launch<128, 128, false>(a, b, e, M, g.C, Kp, s, slab);
}
static int conv_wgrad16(const float* x16, const float* dy16, float* pk, const Geom& g, hipStream_t s, float* slab) {
  if (!bf16_conv_ok(g)) return AVVAD_EINVAL;
  const int T = g.KS * g.KS, M = T * g.C, K = g.N * g.Ho * g.Wo;
  if ((unsigned long)(K + bgemm::BKU) * (unsigned long)(g.Ho * g.Wo) >= 0x100000000ull) return AVVAD_EINVAL;   // fast_div range
  if (cls16_common(g, slab) && g.C % 128 == 0 && g.Co >= 128 &&
      9L * (g.C / 128) * cdiv(g.Co, 128) <= cls_tile_cap()) {
    const igemm::TapSched sc = tap_sched(g, (g.C / 128) * cdiv(g.Co, 128));
    Geom gq = g;
    gq.C = g.C / 2; gq.Co = g.Co / 2;                        // operand elements are bf16 pairs
    convop::WgradXTap a{x16, gq, M / 2, g.C, g.N, convop::div_magic(g.C / 2), sc};
    convop::ColDyTap b{dy16, gq, g.Co / 2, g.C, g.N, sc};
    igemm::EpiTap e{pk, g.Co, nullptr, 0};
    e.Mrows = M; e.sched = sc;
    return bgemm::launch_cls<true>(a, b, e, M, g.Co, s, slab);
  }
  Geom gp = g;
  gp.C = g.C / 2;
  convop::WgradX<true> a{x16, gp, M / 2, K, convop::div_magic(g.Ho * g.Wo), convop::div_magic(g.Wo)};
  igemm::ColPlain<4, true> b{dy16, g.Co / 2, g.Co / 2, K, 0};
  convop::EpiWgrad e{pk, g.Co, nullptr, 0, 1, g.C, T, convop::div_magic(T), 6};
  if (g.Co <= 64) return bgemm::launch<128, 64, true>(a, b, e, M, g.Co, K, s, slab);
  return bgemm::launch<128, 128, true>(a, b, e, M, g.Co, K, s, slab);
}

struct StatCtx { double* part; int nchunk; long rows_per_chunk; };
static StatCtx stat_ctx(Plan* p, float* ws, long M, int C) {
  StatCtx c;
  c.part = reinterpret_cast<double*>(ws + p->part);
  const int RL = 256 / (C / 4);
  long per = (M + STAT_CHUNKS - 1) / STAT_CHUNKS;
  if (per < 16L * RL) per = 16L * RL;   // >= 16 rows per thread: fewer, fuller chunks for the small deep layers (the finalize
                                         // kernels read every chunk's partial sums)
  per = (per + RL - 1) / RL * RL;
  c.rows_per_chunk = per;
  c.nchunk = cdiv(M, per);
  return c;
}

// batch statistics of conv output i -> scale/shift/mean/invstd slots i (+ running stats)
// where conv i's forward can leave its fused statistics (null: take the separate column-reduction pass)
static double* fused_stat(Plan* p, float* ws, int i, const avvad_trunk_desc* d) {
  const int rows = fwd_tile_rows(p->geom[i]);
  if (i == 0 && d->training && !avvad_tune().no_fused_stats && stem_kernel_ok(p->geom[0]) && (long)p->N * 64 <= (long)STAT_CHUNKS * MAXC)
    return reinterpret_cast<double*>(ws + p->part);      // one chunk per frame, written by stem_fwd_mfma
  if (!d->training || rows == 0 || avvad_tune().no_fused_stats) return nullptr;
  const long chunks = fwd_stat_chunks(p->geom[i], ws + p->slab);
  if (chunks * p->conv[i].cout > (long)STAT_CHUNKS * MAXC) return nullptr;      // the partial-sum buffer's size
  return reinterpret_cast<double*>(ws + p->part);
}
static int bn_prepare(Plan* p, float* ws, int i, const float* craw, long M, const avvad_trunk_params* prm,
                      const avvad_trunk_desc* d, hipStream_t s) {
  const int C = p->conv[i].cout;
  StatCtx sc = stat_ctx(p, ws, M, C);
  if (fused_stat(p, ws, i, d)) {
    sc.nchunk = i == 0 ? p->N : fwd_stat_chunks(p->geom[i], ws + p->slab);   // one chunk per M tile of the conv's GEMM (stem: per frame)
  } else if (d->training) {
    hipLaunchKernelGGL(col_reduce<0>, dim3(sc.nchunk), dim3(256), 0, s, craw, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, M, C, sc.rows_per_chunk, sc.part);
  }
  hipLaunchKernelGGL(bn_finalize, dim3(cdiv(C, FIN_CH)), dim3(256), 0, s, sc.part, sc.nchunk, M, C, prm->bn_w[i], prm->bn_b[i],
                     prm->bn_rm[i], prm->bn_rv[i], d->training, d->momentum, d->eps, ws + p->bn_scale + i * MAXC,
                     ws + p->bn_shift + i * MAXC, ws + p->bn_mean + i * MAXC, ws + p->bn_invstd + i * MAXC);
  AVVAD_LAUNCH_CHECK();
  return AVVAD_OK;
}

// BatchNorm backward of conv i:  dx = BN'(x_raw; dy masked by ymask>0); optional gout = masked dy
// own_relu: this BatchNorm's output went straight through a ReLU (the first BN of a block): the mask is rebuilt from xraw
// with the forward's saved scale / shift instead of reading the activation (ymask is then ignored)
static int bn_backward(Plan* p, float* ws, int i, const float* xraw, const float* dy, const float* ymask, float* dx,
                       float* gout, long M, const avvad_trunk_params* prm, const avvad_trunk_grads* g,
                       const avvad_trunk_desc* d, hipStream_t s, bool own_relu = false, const unsigned char* qmask = nullptr,
                       bool out16 = false, int reduced_chunks = 0) {
  // reduced_chunks > 0: the column sums are already in the partial-sum buffer (that many chunks), written by the producer of dy
  // out16 (the bf16 data path): dx is stored as bf16, and the block output's ReLU mask comes from the byte mask in BOTH
  // passes (the activation itself is bf16 there)
  const int C = p->conv[i].cout;
  StatCtx sc = stat_ctx(p, ws, M, C);
  const float* mean = ws + p->bn_mean + i * MAXC;
  const float* invstd = ws + p->bn_invstd + i * MAXC;
  const float* msc = own_relu ? ws + p->bn_scale + i * MAXC : (const float*)nullptr;
  const float* msh = own_relu ? ws + p->bn_shift + i * MAXC : (const float*)nullptr;
  // qmask: the block output's ReLU mask as one byte per quad, written by bn_act.  The apply kernel reads it instead of the
  // activation (32.5 -> 28.2 us); the column reduction keeps the float activation -- with byte loads it got 9 % SLOWER.
  if (own_relu) ymask = nullptr;
  const bool qred = out16 && qmask != nullptr;
  if (reduced_chunks > 0)
    sc.nchunk = reduced_chunks;
  else
    hipLaunchKernelGGL(col_reduce<1>, dim3(sc.nchunk), dim3(256), 0, s, xraw, dy, qred ? (const float*)nullptr : ymask, mean, invstd, M, C,
                       sc.rows_per_chunk, sc.part, msc, msh, qred ? qmask : (const unsigned char*)nullptr);
  hipLaunchKernelGGL(bn_bwd_finalize, dim3(cdiv(C, FIN_CH)), dim3(256), 0, s, sc.part, sc.nchunk, M, C, prm->bn_w[i], invstd,
                     d->training, g->bn_w[i], g->bn_b[i], ws + p->coef);
  const long nq = M * C / 4;
  if (out16)
    hipLaunchKernelGGL(bn_bwd_apply<true>, dim3(ew_grid(nq)), dim3(256), 0, s, xraw, dy, qmask ? (const float*)nullptr : ymask, mean, invstd,
                       ws + p->coef, dx, gout, nq, C, msc, msh, qmask);
  else
    hipLaunchKernelGGL(bn_bwd_apply<false>, dim3(ew_grid(nq)), dim3(256), 0, s, xraw, dy, qmask ? (const float*)nullptr : ymask, mean, invstd,
                       ws + p->coef, dx, gout, nq, C, msc, msh, qmask);
  AVVAD_LAUNCH_CHECK();
  return AVVAD_OK;
}

}  // namespace

// ====================================================================== C ABI
// ---- single-convolution entry points (NHWC activations, packed weights): the building blocks of the trunk,
// exported for unit tests and for bench.py's per-launch roofline timing.
static bool conv_geom(const avvad_conv_desc* d, Geom* g) {
  if (!d || d->N <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->Co <= 0 || d->KS <= 0 || d->stride <= 0 || d->pad < 0)
    return false;
  const int Ho = (d->H + 2 * d->pad - d->KS) / d->stride + 1, Wo = (d->W + 2 * d->pad - d->KS) / d->stride + 1;
  if (Ho <= 0 || Wo <= 0 || (d->stride != 1 && d->stride != 2)) return false;
  *g = Geom{d->N, d->H, d->W, d->C, Ho, Wo, d->Co, d->KS, d->stride, d->pad};
  return true;
}
extern "C" int avvad_conv2d_pack_weights(const float* w_oihw, float* wf, float* wd, const avvad_conv_desc* d,
                                         avvad_stream_t s) {
  AVVAD_ENTER();
  Geom g;
  if (!w_oihw || !wf || !conv_geom(d, &g)) return AVVAD_EINVAL;
  hipLaunchKernelGGL(pack_weights, dim3(ew_grid((long)g.Co * g.C * g.KS * g.KS)), dim3(256), 0, (hipStream_t)s, w_oihw, wf, wd,
                     g.Co, g.C, g.KS);
  AVVAD_LAUNCH_CHECK();
  return AVVAD_OK;
}
static float* slab_of(void* ws, size_t ws_bytes) { return (ws && ws_bytes >= igemm::SLAB_FLOATS * sizeof(float)) ? (float*)ws : nullptr; }
extern "C" size_t avvad_engine_workspace(void) { return igemm::SLAB_FLOATS * sizeof(float); }
extern "C" int avvad_conv2d_fwd(const float* x, const float* wf, float* y, const avvad_conv_desc* d, void* ws, size_t ws_bytes,
                                avvad_stream_t s) {
  AVVAD_ENTER();
  Geom g;
  if (!x || !wf || !y || !conv_geom(d, &g)) return AVVAD_EINVAL;
  return conv_fwd(x, wf, y, g, (hipStream_t)s, slab_of(ws, ws_bytes));
}
#ifdef AVVAD_PROF
extern "C" int avvad_debug_prof(unsigned long long* out, int reset) {
  if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(igemm::g_prof), sizeof(unsigned long long) * 8);
  if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(igemm::g_prof), z, sizeof(z)); }
  return 0;
}
extern "C" int avvad_debug_prof_blocks(unsigned long long* out, int n) {
  hipMemcpyFromSymbol(out, HIP_SYMBOL(igemm::g_prof_blk), sizeof(unsigned long long) * 2 * n);
  return 0;
}
extern "C" int avvad_debug_prof_clk(unsigned long long* out, int n) {
  hipMemcpyFromSymbol(out, HIP_SYMBOL(igemm::g_prof_clk), sizeof(unsigned long long) * n);
  return 0;
}
extern "C" int avvad_debug_prof_hw(unsigned long long* out, int n) {
  hipMemcpyFromSymbol(out, HIP_SYMBOL(igemm::g_prof_hw), sizeof(unsigned long long) * n);
  return 0;
}
#endif
extern "C" int avvad_conv2d_dgrad(const float* dy, const float* wd, float* dx, const avvad_conv_desc* d, int accumulate,
                                  void* ws, size_t ws_bytes, avvad_stream_t s) {
  AVVAD_ENTER();
  Geom g;
  if (!dy || !wd || !dx || !conv_geom(d, &g)) return AVVAD_EINVAL;
  return conv_dgrad(dy, wd, dx, g, accumulate, (hipStream_t)s, slab_of(ws, ws_bytes));
}
extern "C" int avvad_conv2d_wgrad(const float* x, const float* dy, float* dw_packed, const avvad_conv_desc* d, void* ws,
                                  size_t ws_bytes, avvad_stream_t s) {
  AVVAD_ENTER();
  Geom g;
  if (!x || !dy || !dw_packed || !conv_geom(d, &g)) return AVVAD_EINVAL;
  return conv_wgrad(x, dy, dw_packed, g, (hipStream_t)s, slab_of(ws, ws_bytes));
}

// ---- the bf16 data path's convolutions on their own (bgemm.h): operands bf16 in HBM, fp32 results.  Exported for the
// parity tests and bench.py's bf16 roofline probe; the trunk calls the same launchers when option "bf16" is 1.
namespace {
__global__ void pack_weights_bf16(const float* __restrict__ w, __bf16* __restrict__ wf16, __bf16* __restrict__ wd16, int Co, int C, int KS) {
  const int T = KS * KS, n = Co * C * T;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    int r = i;                                                 // i indexes OIHW
    const int tap = r % T; r /= T;
    const int c = r % C; const int co = r / C;
    const __bf16 v = (__bf16)w[i];
    wf16[(long)co * (T * C) + ((c >> 6) * T + tap) * 64 + (c & 63)] = v;
    if (wd16) wd16[(long)c * (T * Co) + ((co >> 6) * T + tap) * 64 + (co & 63)] = v;
  }
}
}  // namespace
extern "C" int avvad_conv2d_pack_weights_bf16(const float* w_oihw, void* wf16, void* wd16, const avvad_conv_desc* d, avvad_stream_t s) {
  AVVAD_ENTER();
  Geom g;
  if (!w_oihw || !wf16 || !conv_geom(d, &g) || g.C % 64 || g.Co % 64) return AVVAD_EINVAL;
  hipLaunchKernelGGL(pack_weights_bf16, dim3(ew_grid((long)g.Co * g.C * g.KS * g.KS)), dim3(256), 0, (hipStream_t)s, w_oihw,
                     (__bf16*)wf16, (__bf16*)wd16, g.Co, g.C, g.KS);
  AVVAD_LAUNCH_CHECK();
  return AVVAD_OK;
}
extern "C" int avvad_conv2d_fwd_bf16(const void* x16, const void* wf16, float* y, const avvad_conv_desc* d, void* ws, size_t ws_bytes,
                                     avvad_stream_t s) {
  AVVAD_ENTER();
  Geom g;
  if (!x16 || !wf16 || !y || !conv_geom(d, &g)) return AVVAD_EINVAL;
  return conv_fwd16((const float*)x16, (const float*)wf16, y, g, (hipStream_t)s, slab_of(ws, ws_bytes), nullptr);
}
extern "C" int avvad_conv2d_dgrad_bf16(const void* dy16, const void* wd16, float* dx, const avvad_conv_desc* d, int accumulate,
                                       void* ws, size_t ws_bytes, avvad_stream_t s) {
  AVVAD_ENTER();
  Geom g;
  if (!dy16 || !wd16 || !dx || !conv_geom(d, &g)) return AVVAD_EINVAL;
  return conv_dgrad16((const float*)dy16, (const float*)wd16, dx, g, accumulate, (hipStream_t)s, slab_of(ws, ws_bytes));
}
extern "C" int avvad_conv2d_wgrad_bf16(const void* x16, const void* dy16, float* dw_packed, const avvad_conv_desc* d, void* ws,
                                       size_t ws_bytes, avvad_stream_t s) {
  AVVAD_ENTER();
  Geom g;
  if (!x16 || !dy16 || !dw_packed || !conv_geom(d, &g)) return AVVAD_EINVAL;
  return conv_wgrad16((const float*)x16, (const float*)dy16, dw_packed, g, (hipStream_t)s, slab_of(ws, ws_bytes));
}

extern "C" size_t avvad_trunk_workspace(const avvad_trunk_desc* d) {
  if (!d || d->N <= 0 || d->H < 32 || d->W < 32) return 0;
  Plan p;
  make_plan(d, &p);
  return p.total * sizeof(float);
}

// Where the post-ReLU activations a forward run with save_for_backward keeps in its workspace live (test support: the
// parity tests compare their sign patterns with the oracle's to tell a flipped ReLU unit from an indexing error).
// index 0: pooled stem output; 1 + 2k: block k's first activation (after bn1 + ReLU); 2 + 2k: block k's output. NHWC.
extern "C" int avvad_trunk_activation(const avvad_trunk_desc* d, int index, size_t* offset_floats, int* C, int* H, int* W) {
  if (!d || d->N <= 0 || d->H < 32 || d->W < 32 || index < 0 || index > 16 || !offset_floats || !C || !H || !W) return AVVAD_EINVAL;
  Plan p;
  make_plan(d, &p);
  const int widths[4] = {64, 128, 256, 512};
  if (index == 0) { *offset_floats = p.p0; *C = 64; *H = p.h[2]; *W = p.w[2]; return AVVAD_OK; }
  const int k = (index - 1) / 2, st = k / 2;
  *offset_floats = p.blk[k][(index - 1) % 2 == 0 ? 1 : 4];
  *C = widths[st]; *H = p.h[st + 2]; *W = p.w[st + 2];
  return AVVAD_OK;
}

extern "C" int avvad_trunk_fwd(const float* frames, const avvad_trunk_params* prm, float* feat, const avvad_trunk_desc* d,
                               void* wsv, size_t ws_bytes, avvad_stream_t sv) {
  AVVAD_ENTER();
  if (!frames || !prm || !feat || !d || !wsv || d->N <= 0 || d->H < 32 || d->W < 32) return AVVAD_EINVAL;
  hipStream_t s = (hipStream_t)sv;
  Plan p;
  make_plan(d, &p);
  if (ws_bytes < p.total * sizeof(float)) return AVVAD_EWORKSPACE;
  float* ws = (float*)wsv;
  int rc;
  const bool b16 = native_bf16();        // the bf16 data path: activations between convolutions are stored as bf16
  // weights: OIHW -> packed
  {
    PackTab tab;
    int nb = 0;
    for (int i = 0; i < NCONV; ++i) {
      const ConvSpec& c = p.conv[i];
      tab.w[i] = prm->conv_w[i];
      tab.wf[i] = ws + p.wf[i];
      tab.wd[i] = (i > 0 && d->save_for_backward) ? ws + p.wd[i] : (float*)nullptr;
      tab.wf16[i] = (b16 && i > 0) ? reinterpret_cast<__bf16*>(ws + p.wf16[i]) : (__bf16*)nullptr;
      tab.wd16[i] = (b16 && i > 0 && d->save_for_backward) ? reinterpret_cast<__bf16*>(ws + p.wd16[i]) : (__bf16*)nullptr;
      tab.cout[i] = c.cout; tab.cin[i] = c.cin; tab.ks[i] = c.ks;
      tab.blk0[i] = nb;
      nb += i == 0 ? cdiv(64 * 49, 256) : (c.cout / 32) * (c.cin / 32);
    }
    tab.blk0[NCONV] = nb;
    hipLaunchKernelGGL(pack_all, dim3(nb), dim3(256), 0, s, tab);
  }
  // stem
  const long N = d->N;
  if ((rc = conv_fwd(frames, ws + p.wf[0], ws + p.c0, p.geom[0], s, ws + p.slab, fused_stat(&p, ws, 0, d)))) return rc;
  const long M0 = N * p.h[1] * p.w[1];
  if ((rc = bn_prepare(&p, ws, 0, ws + p.c0, M0, prm, d, s))) return rc;
  {
    unsigned* am = d->save_for_backward ? reinterpret_cast<unsigned*>(ws + p.am) : (unsigned*)nullptr;
    if (b16)
      hipLaunchKernelGGL(stem_bn_relu_pool<true>, dim3(ew_grid(N * p.h[2] * p.w[2] * 16)), dim3(256), 0, s, ws + p.c0, ws + p.bn_scale,
                         ws + p.bn_shift, ws + p.p0, am, d->N, p.h[1], p.w[1], p.h[2], p.w[2]);
    else
      hipLaunchKernelGGL(stem_bn_relu_pool<false>, dim3(ew_grid(N * p.h[2] * p.w[2] * 16)), dim3(256), 0, s, ws + p.c0, ws + p.bn_scale,
                         ws + p.bn_shift, ws + p.p0, am, d->N, p.h[1], p.w[1], p.h[2], p.w[2]);
  }
  auto cfwd = [&](const float* xin, int i, float* yout) -> int {
    if (b16) return conv_fwd16(xin, ws + p.wf16[i], yout, p.geom[i], s, ws + p.slab, fused_stat(&p, ws, i, d));
    return conv_fwd(xin, ws + p.wf[i], yout, p.geom[i], s, ws + p.slab, fused_stat(&p, ws, i, d));
  };
  // residual stages
  const float* x = ws + p.p0;
  int ci = 1;
  const int widths[4] = {64, 128, 256, 512};
  for (int st = 0; st < 4; ++st)
    for (int b = 0; b < 2; ++b) {
      const int C = widths[st];
      const size_t* o = p.blk[st * 2 + b];
      const bool ds = (b == 0 && st > 0);
      const long M = N * p.h[st + 2] * p.w[st + 2];
      const long nq = M * C / 4;
      const int i1 = ci, i2 = ci + 1, id = ci + 2;
      unsigned char* qmo = d->save_for_backward ? reinterpret_cast<unsigned char*>(ws + p.qm[st * 2 + b]) : (unsigned char*)nullptr;
      if ((rc = cfwd(x, i1, ws + o[0]))) return rc;
      if ((rc = bn_prepare(&p, ws, i1, ws + o[0], M, prm, d, s))) return rc;
      if (b16)
        hipLaunchKernelGGL((bn_act<true, false>), dim3(ew_grid(nq)), dim3(256), 0, s, ws + o[0], ws + p.bn_scale + i1 * MAXC,
                           ws + p.bn_shift + i1 * MAXC, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                           ws + o[1], nq, C, 1, (unsigned char*)nullptr);
      else
        hipLaunchKernelGGL((bn_act<false, false>), dim3(ew_grid(nq)), dim3(256), 0, s, ws + o[0], ws + p.bn_scale + i1 * MAXC,
                           ws + p.bn_shift + i1 * MAXC, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                           ws + o[1], nq, C, 1, (unsigned char*)nullptr);
      if ((rc = cfwd(ws + o[1], i2, ws + o[2]))) return rc;
      if ((rc = bn_prepare(&p, ws, i2, ws + o[2], M, prm, d, s))) return rc;
      if (ds) {
        if ((rc = cfwd(x, id, ws + o[3]))) return rc;
        if ((rc = bn_prepare(&p, ws, id, ws + o[3], M, prm, d, s))) return rc;
        // (the identity input is the downsample convolution's RAW fp32 output in either data path)
        if (b16)
          hipLaunchKernelGGL((bn_act<true, false>), dim3(ew_grid(nq)), dim3(256), 0, s, ws + o[2], ws + p.bn_scale + i2 * MAXC,
                             ws + p.bn_shift + i2 * MAXC, ws + o[3], ws + p.bn_scale + id * MAXC, ws + p.bn_shift + id * MAXC,
                             ws + o[4], nq, C, 1, qmo);
        else
          hipLaunchKernelGGL((bn_act<false, false>), dim3(ew_grid(nq)), dim3(256), 0, s, ws + o[2], ws + p.bn_scale + i2 * MAXC,
                             ws + p.bn_shift + i2 * MAXC, ws + o[3], ws + p.bn_scale + id * MAXC, ws + p.bn_shift + id * MAXC,
                             ws + o[4], nq, C, 1, qmo);
      } else {
        if (b16)     // the identity input is the block's bf16 input
          hipLaunchKernelGGL((bn_act<true, true>), dim3(ew_grid(nq)), dim3(256), 0, s, ws + o[2], ws + p.bn_scale + i2 * MAXC,
                             ws + p.bn_shift + i2 * MAXC, x, (const float*)nullptr, (const float*)nullptr, ws + o[4], nq, C, 1, qmo);
        else
          hipLaunchKernelGGL((bn_act<false, false>), dim3(ew_grid(nq)), dim3(256), 0, s, ws + o[2], ws + p.bn_scale + i2 * MAXC,
                             ws + p.bn_shift + i2 * MAXC, x, (const float*)nullptr, (const float*)nullptr, ws + o[4], nq, C, 1, qmo);
      }
      x = ws + o[4];
      ci += ds ? 3 : 2;
    }
  if (b16) hipLaunchKernelGGL(avgpool_fwd<true>, dim3(ew_grid(N * 512)), dim3(256), 0, s, x, feat, d->N, p.h[5] * p.w[5], 512);
  else hipLaunchKernelGGL(avgpool_fwd<false>, dim3(ew_grid(N * 512)), dim3(256), 0, s, x, feat, d->N, p.h[5] * p.w[5], 512);
  AVVAD_LAUNCH_CHECK();
  return AVVAD_OK;
}

extern "C" int avvad_trunk_bwd(const float* frames, const avvad_trunk_params* prm, const float* dfeat,
                               const avvad_trunk_grads* g, const avvad_trunk_desc* d, void* wsv, size_t ws_bytes,
                               avvad_stream_t sv) {
  AVVAD_ENTER();
  BwdCuCap cu_cap;
  if (!frames || !prm || !dfeat || !g || !d || !wsv || !d->save_for_backward) return AVVAD_EINVAL;
  hipStream_t s = (hipStream_t)sv;
  Plan p;
  make_plan(d, &p);
  if (ws_bytes < p.total * sizeof(float)) return AVVAD_EWORKSPACE;
  float* ws = (float*)wsv;
  int rc;
  const long N = d->N;
  const int widths[4] = {64, 128, 256, 512};
  float* G0 = ws + p.G[0];
  float* G1 = ws + p.G[1];
  float* G2 = ws + p.G[2];
  float* G3 = ws + p.G[3];
  // d(out of last block) from the average pool
  hipLaunchKernelGGL(avgpool_bwd, dim3(ew_grid(N * p.h[5] * p.w[5] * 512)), dim3(256), 0, s, dfeat, G0, d->N,
                     p.h[5] * p.w[5], 512);
  const bool b16 = native_bf16();        // the forward ran the bf16 data path (same process-wide option): activations,
                                         // and the BatchNorm-backward outputs that feed convolutions, are bf16
  auto wgrad = [&](int i, const float* xin, const float* dyraw) -> int {
    if (!g->conv_w[i]) return AVVAD_OK;
    if (b16 && i > 0) return conv_wgrad16(xin, dyraw, ws + p.wg[i], p.geom[i], s, ws + p.slab);
    return conv_wgrad(xin, dyraw, ws + p.wg[i], p.geom[i], s, ws + p.slab);   // unpacked at the end (unpack_all)
  };
  auto dgrad = [&](const float* dyv, int i, float* dxv, int accumulate) -> int {
    if (b16) return conv_dgrad16(dyv, ws + p.wd16[i], dxv, p.geom[i], accumulate, s, ws + p.slab);
    return conv_dgrad(dyv, ws + p.wd[i], dxv, p.geom[i], accumulate, s, ws + p.slab);
  };
  int ci = NCONV;
  for (int st = 3; st >= 0; --st)
    for (int b = 1; b >= 0; --b) {
      const bool ds = (b == 0 && st > 0);
      ci -= ds ? 3 : 2;
      const int i1 = ci, i2 = ci + 1, id = ci + 2;
      const size_t* o = p.blk[st * 2 + b];
      const float* x = (st == 0 && b == 0) ? ws + p.p0 : (b == 1 ? ws + p.blk[st * 2][4] : ws + p.blk[st * 2 - 1][4]);
      const long M = N * p.h[st + 2] * p.w[st + 2];
      (void)widths;
      const unsigned char* qmb = reinterpret_cast<const unsigned char*>(ws + p.qm[st * 2 + b]);
      // G0 = d(block output, post-ReLU).  main branch: BN2 backward (mask out>0) -> d c2 in G1
      if (ds) {
        if ((rc = bn_backward(&p, ws, i2, ws + o[2], G0, ws + o[4], G1, nullptr, M, prm, g, d, s, false, qmb, b16))) return rc;
        // identity branch through downsample BN + 1x1 conv: d cd in G2, d x in G3
        if ((rc = bn_backward(&p, ws, id, ws + o[3], G0, ws + o[4], G2, nullptr, M, prm, g, d, s, false, qmb, b16))) return rc;
        if ((rc = wgrad(id, x, G2))) return rc;
        if ((rc = dgrad(G2, id, G3, 0))) return rc;
      } else {
        // identity branch: d x = masked d out (written to G3 by the apply kernel)
        if ((rc = bn_backward(&p, ws, i2, ws + o[2], G0, ws + o[4], G1, G3, M, prm, g, d, s, false, qmb, b16))) return rc;
      }
      if ((rc = wgrad(i2, ws + o[1], G1))) return rc;
      if ((rc = dgrad(G1, i2, G2, 0))) return rc;  // d a1 in G2
      if ((rc = bn_backward(&p, ws, i1, ws + o[0], G2, ws + o[1], G1, nullptr, M, prm, g, d, s, true, nullptr, b16))) return rc;  // d c1 in G1
      if ((rc = wgrad(i1, x, G1))) return rc;
      if ((rc = dgrad(G1, i1, G3, 1))) return rc;  // d x += ...
      float* t = G0; G0 = G3; G3 = t;
    }
  // stem: G0 = d p0
  if (g->conv_w[0] || g->bn_w[0] || g->bn_b[0]) {
    float* g0 = ws + p.g0;
    // the BatchNorm backward's column sums (sum g0, sum g0 xhat) are formed here, where g0 is produced: 2048 workgroups = chunks
    const bool fuse0 = !avvad_tune().no_fused_stats;
    int grid0 = ew_grid(N * p.h[1] * p.w[1] * 16);
    if (fuse0 && grid0 > 2048) grid0 = 2048;
    hipLaunchKernelGGL(stem_pool_relu_bwd, dim3(grid0), dim3(256), 0, s, ws + p.c0,
                       ws + p.bn_scale, ws + p.bn_shift, reinterpret_cast<const unsigned*>(ws + p.am), G0, g0, d->N, p.h[1], p.w[1],
                       p.h[2], p.w[2], fuse0 ? ws + p.bn_mean : (const float*)nullptr, fuse0 ? ws + p.bn_invstd : (const float*)nullptr,
                       fuse0 ? reinterpret_cast<double*>(ws + p.part) : (double*)nullptr);
    const long M0 = N * p.h[1] * p.w[1];
    // in place: d c0 overwrites g0
    if ((rc = bn_backward(&p, ws, 0, ws + p.c0, g0, nullptr, g0, nullptr, M0, prm, g, d, s, false, nullptr, false, fuse0 ? grid0 : 0))) return rc;
    if ((rc = wgrad(0, frames, g0))) return rc;
  }
  {
    UnpackTab tab;
    int nb = 0;
    bool any = false;
    for (int i = 0; i < NCONV; ++i) {
      const ConvSpec& c = p.conv[i];
      const bool done = g->conv_w[i] && (i > 0 || g->conv_w[0]);
      tab.pk[i] = ws + p.wg[i];
      tab.dw[i] = done ? g->conv_w[i] : (float*)nullptr;
      tab.cout[i] = c.cout; tab.cin[i] = c.cin; tab.ks[i] = c.ks;
      tab.blk0[i] = nb;
      nb += i == 0 ? cdiv(64 * 3 * 49, 256) : (c.cout / 32) * (c.cin / 32);
      any = any || done;
    }
    tab.blk0[NCONV] = nb;
    if (any) hipLaunchKernelGGL(unpack_all, dim3(nb), dim3(256), 0, s, tab);
  }
  AVVAD_LAUNCH_CHECK();
  return AVVAD_OK;
}
