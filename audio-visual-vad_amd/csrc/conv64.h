// conv64.h -- the 64 -> 64 channel 3x3 / stride 1 / pad 1 convolutions of the trunk's first stage (ResNet layer1: four forward
// products and four data gradients per step, 17x17 grid at 67x67 crops), weights-stationary.
//
// Why not the engine (igemm.h): with 64 output channels a GEMM tile is 256 x 64, so the im2col gather of A -- the expensive
// operand: address arithmetic, validity, LDS transposition -- is amortised over half the columns of a 128 x 128 tile, and the
// kernel ran at 66 % of the fp32 MFMA peak against 80 % for the wider stages (profiles/r03_bench_n1_kernel_stats*.csv).
// Here the whole weight matrix [(tap, c)][co] = 576 x 64 floats = 144 KB stays in LDS for the life of the workgroup (one per
// CU, LDS is 160 KB) and A never touches LDS at all: a wave owns 32 pixels x 64 output channels, lane (i, h) of
// v_mfma_f32_32x32x2_f32 supplies A[row i][k = h], and since the contraction order is free it fetches, per tap, the eight
// 16-byte pieces "channels 8 j + 4 h .. + 3 of ITS pixel" straight into registers -- k-step (tap, j, s) then contracts channel
// 8 j + 4 h + s, which is row tap * 64 + 8 j + 4 h + s of the weight image as it lies in memory.  No staging stores, no
// barriers in the main loop, no split tiles, no fix-up launch; the next tap's pieces are in flight under this tap's 64 MFMAs.
// Padding: a tap that falls outside the image carries the buffer descriptor's out-of-range offset and reads zeros.
//
// Work split: tiles of 32 consecutive pixels; workgroup b owns the contiguous range [b T / G, (b + 1) T / G) and its 8 waves
// walk it interleaved (at any time they work on 8 neighbouring tiles = 256 consecutive pixels, whose halo lines are shared in
// L1).  Two waves share a SIMD; a last, partly filled round leaves one wave per SIMD, which then has the matrix pipe alone.
//
// FLIP = the data gradient: dx[n, h, w, :] = sum over taps of dy[n, h + 1 - kh, w + 1 - kw, :] . Wd[(kh, kw, co)][c] -- the
// same kernel with the gather mirrored and the dgrad pack for its weights.
// stat (forward, training): per-workgroup column sums / sums of squares of y for the fused BatchNorm statistics, laid out as
// igemm::EpiStore::stat -- chunk = workgroup: stat[(b * 2 + {0, 1}) * 64 + co] (fp32 over a tile's 16 rows per lane, doubles
// across tiles, lane halves, waves: a fixed order).
#pragma once
#include "common.h"

namespace conv64 {

constexpr int NW = 8;                 // waves per workgroup
constexpr int WROWS = 9 * 64;         // rows of the weight image

template <bool FLIP>
__global__ void __launch_bounds__(NW * 64, 2)
    kernel(const float* __restrict__ x, const float* __restrict__ wpk, float* __restrict__ y, const int M, const int H, const int W,
           const unsigned mg_hw, const unsigned mg_w, const int accumulate, double* __restrict__ stat) {
  __shared__ __attribute__((aligned(16))) float wl[WROWS * 64 + NW * 64 * 4];     // weights + the statistics exchange (doubles)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const __amdgpu_buffer_rsrc_t rx = brsrc2g(x);
  const __amdgpu_buffer_rsrc_t ry = brsrc2g(y);
  const long T = ((long)M + 31) / 32, G = gridDim.x;
  const long t_lo = blockIdx.x * T / G, t_hi = (blockIdx.x + 1) * T / G;
  const int HW = H * W;
  const float* const bl = wl + lh * 256 + li;           // this lane's column of the weight image (+ 4 h rows)

  // voffset of tap (kh, kw) for the pixel this lane gathers in tile `tile` (BUF_OOB outside the image / past M)
  int mbase = 0, rmask = 0, cmask = 0;
  auto tile_ctx = [&](long tile) {
    const int m = (int)tile * 32 + li;
    const int n = mg_hw ? (int)__umulhi((unsigned)m, mg_hw) : m, r = m - n * HW;
    const int h = mg_w ? (int)__umulhi((unsigned)r, mg_w) : r, w = r - h * W;
    const bool in = m < M;
    // bit d (0..2) of rmask: row h + d - 1 is inside the image
    rmask = in ? ((h > 0 ? 1 : 0) | 2 | (h + 1 < H ? 4 : 0)) : 0;
    cmask = (w > 0 ? 1 : 0) | 2 | (w + 1 < W ? 4 : 0);
    mbase = m * 256 + lh * 16;
  };
  auto tap_off = [&](int tap) -> int {       // (tap: wave-uniform)
    const int kh = (tap * 11) >> 5, kw = tap - kh * 3;
    const int dh = FLIP ? 1 - kh : kh - 1, dw = FLIP ? 1 - kw : kw - 1;
    const bool ok = ((rmask >> (dh + 1)) & 1) && ((cmask >> (dw + 1)) & 1);
    return ok ? mbase + (dh * W + dw) * 256 : BUF_OOB;
  };

  double ds0 = 0, dq0 = 0, ds1 = 0, dq1 = 0;
  f4v a0[8], a1[8];                  // the two taps in flight: one being contracted, one being fetched
  f32x16 acc0, acc1;
  // one tap: 32 k-steps = 64 MFMAs against rows tap * 64 .. + 63 of the weight image; the B values of the next group of four
  // k-steps are read while this group's MFMAs run (the barrier keeps the compiler from hoisting all 64 reads to the top)
  auto contract = [&](const f4v (&av)[8], int tap) {
    const float* const bt = bl + tap * 64 * 64;
    float b[2][8];
#pragma unroll
    for (int s = 0; s < 4; ++s) { b[0][2 * s] = bt[s * 64]; b[0][2 * s + 1] = bt[s * 64 + 32]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < 7) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          b[(j + 1) & 1][2 * s] = bt[((j + 1) * 8 + s) * 64];
          b[(j + 1) & 1][2 * s + 1] = bt[((j + 1) * 8 + s) * 64 + 32];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc0 = mfma32(av[j][s], b[j & 1][2 * s], acc0);
        acc1 = mfma32(av[j][s], b[j & 1][2 * s + 1], acc1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto fetch = [&](f4v (&av)[8], int tap) {
    const int vo = tap_off(tap);
#pragma unroll
    for (int j = 0; j < 8; ++j) av[j] = bload4(rx, vo + j * 32, 0);
  };
  long tile = t_lo + wave;
  if (tile < t_hi) {
    tile_ctx(tile);
    fetch(a0, 0);
  }
  {  // the weight image, verbatim (the first tap's pieces are already in flight)
    const float4* src = reinterpret_cast<const float4*>(wpk);
    float4* dst = reinterpret_cast<float4*>(wl);
#pragma unroll 6
    for (int i = t; i < WROWS * 16; i += NW * 64) dst[i] = src[i];
  }
  __syncthreads();
  for (; tile < t_hi; tile += NW) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const int m0 = (int)tile * 32;
#pragma unroll 1
    for (int tp = 0; tp < 4; ++tp) {          // taps 2 tp (slot 0) and 2 tp + 1 (slot 1)
      fetch(a1, 2 * tp + 1);
      contract(a0, 2 * tp);
      fetch(a0, 2 * tp + 2);
      contract(a1, 2 * tp + 1);
    }
    // tap 8, with the NEXT tile's first tap in flight
    const bool more = tile + NW < t_hi;
    if (more) {
      tile_ctx(tile + NW);
      fetch(a1, 0);
    }
    contract(a0, 8);
    if (more) {
#pragma unroll
      for (int j = 0; j < 8; ++j) a0[j] = a1[j];
    }

    // epilogue: column li (+ 32), rows mfma32_row(r, lh) of the tile
    if (stat != nullptr) {
      float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0 += acc0[r]; q0 = fmaf(acc0[r], acc0[r], q0);
        s1 += acc1[r]; q1 = fmaf(acc1[r], acc1[r], q1);
      }
      ds0 += (double)s0; dq0 += (double)q0; ds1 += (double)s1; dq1 += (double)q1;
    }
    if (accumulate) {
      float o0[16], o1[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mfma32_row(r, lh);
        const int vo = m < M ? (m * 64 + li) * 4 : BUF_OOB;
        o0[r] = bload(ry, vo, 0);
        o1[r] = bload(ry, vo, 128);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mfma32_row(r, lh);
        const int vo = m < M ? (m * 64 + li) * 4 : BUF_OOB;
        bstore(o0[r] + acc0[r], ry, vo, 0);
        bstore(o1[r] + acc1[r], ry, vo, 128);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mfma32_row(r, lh);
        const int vo = m < M ? (m * 64 + li) * 4 : BUF_OOB;
        bstore(acc0[r], ry, vo, 0);
        bstore(acc1[r], ry, vo, 128);
      }
    }
  }
  if (stat != nullptr) {
    ds0 += __shfl_xor(ds0, 32, 64); dq0 += __shfl_xor(dq0, 32, 64);
    ds1 += __shfl_xor(ds1, 32, 64); dq1 += __shfl_xor(dq1, 32, 64);
    double* red = reinterpret_cast<double*>(wl + WROWS * 64);          // [NW][64][2]
    if (lh == 0) {
      red[(wave * 64 + li) * 2 + 0] = ds0; red[(wave * 64 + li) * 2 + 1] = dq0;
      red[(wave * 64 + li + 32) * 2 + 0] = ds1; red[(wave * 64 + li + 32) * 2 + 1] = dq1;
    }
    __syncthreads();
    if (t < 64) {
      double sa = 0, sb = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) { sa += red[(w * 64 + t) * 2 + 0]; sb += red[(w * 64 + t) * 2 + 1]; }
      stat[((long)blockIdx.x * 2 + 0) * 64 + t] = sa;
      stat[((long)blockIdx.x * 2 + 1) * 64 + t] = sb;
    }
  }
}

// ---------------------------------------------------------------- the same kernel on the bf16 data path
// x / dy are bf16 NHWC (128 bytes per pixel), the weights the K-contiguous bf16 packs [co][(tap, c)] / [c][(tap, co)] of the
// bf16 engine (576 bf16 = 1152 bytes per row), y / dx fp32.  v_mfma_f32_32x32x16_bf16 takes 8 consecutive k per lane: lane
// (i, h) fetches, per tap, the four 16-byte pieces "channels 16 s + 8 h .. + 7 of its pixel" (s = 0..3) and k-step (tap, s)
// contracts them against the weight image's 16 bytes at k = tap * 64 + 16 s + 8 h of row co -- one ds_read_b128 each; the
// image's rows are padded to 1168 bytes so that the 16 lanes of a read phase cover all 64 banks.  74 KB of LDS.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int W16_LD = 292;           // dwords per row of the bf16 weight image (288 + 4)

template <bool FLIP>
__global__ void __launch_bounds__(NW * 64, 2)
    kernel16(const float* __restrict__ x16, const float* __restrict__ wpk16, float* __restrict__ y, const int M, const int H, const int W,
             const unsigned mg_hw, const unsigned mg_w, const int accumulate, double* __restrict__ stat) {
  __shared__ __attribute__((aligned(16))) float wl[64 * W16_LD + NW * 64 * 4];     // weights + the statistics exchange (doubles)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const __amdgpu_buffer_rsrc_t rx = brsrc2g(x16);
  const __amdgpu_buffer_rsrc_t ry = brsrc2g(y);
  const long T = ((long)M + 31) / 32, G = gridDim.x;
  const long t_lo = blockIdx.x * T / G, t_hi = (blockIdx.x + 1) * T / G;
  const int HW = H * W;
  const float* const bl = wl + li * W16_LD + lh * 4;            // this lane's weight row (+ 8 h in k)

  int mbase = 0, rmask = 0, cmask = 0;
  auto tile_ctx = [&](long tile) {
    const int m = (int)tile * 32 + li;
    const int n = mg_hw ? (int)__umulhi((unsigned)m, mg_hw) : m, r = m - n * HW;
    const int h = mg_w ? (int)__umulhi((unsigned)r, mg_w) : r, w = r - h * W;
    const bool in = m < M;
    rmask = in ? ((h > 0 ? 1 : 0) | 2 | (h + 1 < H ? 4 : 0)) : 0;
    cmask = (w > 0 ? 1 : 0) | 2 | (w + 1 < W ? 4 : 0);
    mbase = m * 128 + lh * 16;
  };
  auto tap_off = [&](int tap) -> int {       // (tap: wave-uniform)
    const int kh = (tap * 11) >> 5, kw = tap - kh * 3;
    const int dh = FLIP ? 1 - kh : kh - 1, dw = FLIP ? 1 - kw : kw - 1;
    const bool ok = ((rmask >> (dh + 1)) & 1) && ((cmask >> (dw + 1)) & 1);
    return ok ? mbase + (dh * W + dw) * 128 : BUF_OOB;
  };

  double ds0 = 0, dq0 = 0, ds1 = 0, dq1 = 0;
  f4v a0[4], a1[4];
  f32x16 acc0, acc1;
  auto contract = [&](const f4v (&av)[4], int tap) {      // one tap: 4 k-steps of 16, 8 MFMAs
    const float* const bt = bl + tap * 32;
    bf16x8 b[4][2];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      b[s][0] = *reinterpret_cast<const bf16x8*>(bt + s * 8);
      b[s][1] = *reinterpret_cast<const bf16x8*>(bt + 32 * W16_LD + s * 8);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 a = __builtin_bit_cast(bf16x8, av[s]);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[s][0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[s][1], acc1, 0, 0, 0);
    }
  };
  auto fetch = [&](f4v (&av)[4], int tap) {
    const int vo = tap_off(tap);
#pragma unroll
    for (int s = 0; s < 4; ++s) av[s] = bload4(rx, vo + s * 32, 0);
  };
  long tile = t_lo + wave;
  if (tile < t_hi) {
    tile_ctx(tile);
    fetch(a0, 0);
  }
  {  // the weight image: 64 rows of 72 sixteen-byte pieces, padded rows
    const float4* src = reinterpret_cast<const float4*>(wpk16);
    for (int i = t; i < 64 * 72; i += NW * 64) {
      const int row = i / 72, c4 = i - row * 72;
      *reinterpret_cast<float4*>(wl + row * W16_LD + c4 * 4) = src[i];
    }
  }
  __syncthreads();
  for (; tile < t_hi; tile += NW) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const int m0 = (int)tile * 32;
#pragma unroll 1
    for (int tp = 0; tp < 4; ++tp) {
      fetch(a1, 2 * tp + 1);
      contract(a0, 2 * tp);
      fetch(a0, 2 * tp + 2);
      contract(a1, 2 * tp + 1);
    }
    const bool more = tile + NW < t_hi;
    if (more) {
      tile_ctx(tile + NW);
      fetch(a1, 0);
    }
    contract(a0, 8);
    if (more) {
#pragma unroll
      for (int j = 0; j < 4; ++j) a0[j] = a1[j];
    }
    if (stat != nullptr) {
      float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0 += acc0[r]; q0 = fmaf(acc0[r], acc0[r], q0);
        s1 += acc1[r]; q1 = fmaf(acc1[r], acc1[r], q1);
      }
      ds0 += (double)s0; dq0 += (double)q0; ds1 += (double)s1; dq1 += (double)q1;
    }
    if (accumulate) {
      float o0[16], o1[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mfma32_row(r, lh);
        const int vo = m < M ? (m * 64 + li) * 4 : BUF_OOB;
        o0[r] = bload(ry, vo, 0);
        o1[r] = bload(ry, vo, 128);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mfma32_row(r, lh);
        const int vo = m < M ? (m * 64 + li) * 4 : BUF_OOB;
        bstore(o0[r] + acc0[r], ry, vo, 0);
        bstore(o1[r] + acc1[r], ry, vo, 128);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mfma32_row(r, lh);
        const int vo = m < M ? (m * 64 + li) * 4 : BUF_OOB;
        bstore(acc0[r], ry, vo, 0);
        bstore(acc1[r], ry, vo, 128);
      }
    }
  }
  if (stat != nullptr) {
    ds0 += __shfl_xor(ds0, 32, 64); dq0 += __shfl_xor(dq0, 32, 64);
    ds1 += __shfl_xor(ds1, 32, 64); dq1 += __shfl_xor(dq1, 32, 64);
    double* red = reinterpret_cast<double*>(wl + 64 * W16_LD);          // [NW][64][2]
    if (lh == 0) {
      red[(wave * 64 + li) * 2 + 0] = ds0; red[(wave * 64 + li) * 2 + 1] = dq0;
      red[(wave * 64 + li + 32) * 2 + 0] = ds1; red[(wave * 64 + li + 32) * 2 + 1] = dq1;
    }
    __syncthreads();
    if (t < 64) {
      double sa = 0, sb = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) { sa += red[(w * 64 + t) * 2 + 0]; sb += red[(w * 64 + t) * 2 + 1]; }
      stat[((long)blockIdx.x * 2 + 0) * 64 + t] = sa;
      stat[((long)blockIdx.x * 2 + 1) * 64 + t] = sb;
    }
  }
}

// ---------------------------------------------------------------- the weight gradient of the same convolutions
// dW[(kh, kw, c)][co] = sum over pixels of x[n, h + kh - 1, w + kw - 1, c] . dy[n, h, w, co]: a 576 x 64 output contracted over
// ~300 k pixels.  On the engine this is 4.5 row tiles of 128 (10 % of the last one's work multiplies nothing) cut 100 ways
// along K, each cut staging its own copies of x through LDS once PER TAP; 59 % of the fp32 MFMA peak.  Here the output is
// stationary: a workgroup of 12 waves holds ALL 36 blocks of 32 x 32 in registers -- wave (quadrant (c half, co half), kh)
// owns the three kw blocks -- and walks a contiguous range of image rows.  Per image row the three x rows it touches and the
// dy row go through LDS ONCE (16-byte loads, double-buffered, one barrier per row; zero columns either side and zero rows
// outside the image stand in for the padding) and serve all nine taps: lane (i, h) reads A[c = i][k = pixel w + h] and
// B[k][co = j] as single conflict-free dwords.  Each workgroup leaves one partial [576][64]; a second kernel adds the
// partials in workgroup order (bit-reproducible, no atomics).
constexpr int WG_WAVES = 12;
constexpr int WG_MAXW = 18;           // widest grid row the static LDS image holds
constexpr int WG_PART = WROWS * 64;   // floats per partial

__global__ void __launch_bounds__(WG_WAVES * 64, 3)
    wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part, const int NR, const int H,
                 const int W, const unsigned mg_H) {
  constexpr int XP = (WG_MAXW + 2) * 64;            // floats per staged x row (pitch), columns -1 .. W
  constexpr int DP = (WG_MAXW + 2) * 64;            // floats per staged dy row (an even number of pixels + spare)
  __shared__ __attribute__((aligned(16))) float lds[2 * (3 * XP + DP)];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int q = wave / 3, kh = wave - q * 3, cb = q >> 1, nb = q & 1;
  for (int i = t; i < 2 * (3 * XP + DP) / 4; i += WG_WAVES * 64) reinterpret_cast<float4*>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const __amdgpu_buffer_rsrc_t rx = brsrc2g(x);
  const __amdgpu_buffer_rsrc_t rd = brsrc2g(dy);
  const long G = gridDim.x;
  const int r_lo = (int)(blockIdx.x * (long)NR / G), r_hi = (int)((blockIdx.x + 1) * (long)NR / G);
  // this thread's two staging items (the same for every row): item = (source row 0..2 of x | 3 = dy, pixel, 16-byte piece)
  const int per_row = W * 16, items = 4 * per_row;
  int sel[2], goff[2], loff[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int idx = t + u * WG_WAVES * 64;
    const int rs = idx / per_row, rem = idx - rs * per_row, px = rem >> 4, c4 = rem & 15;
    sel[u] = idx < items ? rs : -1;
    goff[u] = ((rs < 3 ? rs - 1 : 0) * W + px) * 256 + c4 * 16;           // bytes, relative to the output row's first pixel
    loff[u] = rs < 3 ? rs * XP + (px + 1) * 64 + c4 * 4 : 3 * XP + px * 64 + c4 * 4;
  }
  f4v st[2];
  auto fetch = [&](int r) {
    const int n = mg_H ? (int)__umulhi((unsigned)r, mg_H) : r, h = r - n * H;
    const int rb = r * W * 256;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const bool ok = sel[u] == 3 || (sel[u] >= 0 && (unsigned)(h + sel[u] - 1) < (unsigned)H);
      const int vo = ok ? rb + goff[u] : BUF_OOB;
      st[u] = sel[u] == 3 ? bload4(rd, vo, 0) : bload4(rx, vo, 0);
    }
  };
  auto put = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (sel[u] >= 0) *reinterpret_cast<f4v*>(lds + buf * (3 * XP + DP) + loff[u]) = st[u];
  };
  f32x16 acc[3];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  const int nks = (W + 1) >> 1;
  __syncthreads();
  if (r_lo < r_hi) {
    fetch(r_lo);
    put(0);
  }
  __syncthreads();
  for (int r = r_lo; r < r_hi; ++r) {
    const int buf = (r - r_lo) & 1;
    if (r + 1 < r_hi) fetch(r + 1);
    const float* const ap = lds + buf * (3 * XP + DP) + kh * XP + lh * 64 + cb * 32 + li;      // column w + h + kw (image column + 1)
    const float* const bp = lds + buf * (3 * XP + DP) + 3 * XP + lh * 64 + nb * 32 + li;
#pragma unroll 3
    for (int ks = 0; ks < nks; ++ks) {
      const float b = bp[ks * 128];
      const float a0 = ap[ks * 128], a1 = ap[ks * 128 + 64], a2 = ap[ks * 128 + 128];
      acc[0] = mfma32(a0, b, acc[0]);
      acc[1] = mfma32(a1, b, acc[1]);
      acc[2] = mfma32(a2, b, acc[2]);
    }
    if (r + 1 < r_hi) put(buf ^ 1);
    __syncthreads();
  }
  float* const P = part + (long)blockIdx.x * WG_PART;
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      P[((kh * 3 + kw) * 64 + cb * 32 + mfma32_row(r, lh)) * 64 + nb * 32 + li] = acc[kw][r];
}

// out[e] = sum over workgroups b of part[b][e], in ascending b within four interleaved groups that meet in a fixed order
__global__ void __launch_bounds__(256)
    wgrad_reduce(const float* __restrict__ part, const int G, float* __restrict__ out) {
  __shared__ float4 red[3][64];
  const int c = threadIdx.x & 63, gq = threadIdx.x >> 6;
  const long e4 = (long)blockIdx.x * 64 + c;                      // float4 index inside a partial
  const float4* p = reinterpret_cast<const float4*>(part) + e4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int b = gq; b < G; b += 32) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (b + 4 * u < G) ? p[(long)(b + 4 * u) * (WG_PART / 4)] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  if (gq > 0) red[gq - 1][c] = s;
  __syncthreads();
  if (gq > 0) return;
#pragma unroll
  for (int w = 0; w < 3; ++w) { const float4 o = red[w][c]; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
  reinterpret_cast<float4*>(out)[e4] = s;
}

// workgroups of a launch over M pixels (= chunks of the fused statistics): one per CU, fewer when there is less than a tile each
static inline int grid_for(long M, int cus) {
  const long T = (M + 31) / 32;
  return (int)(T < cus ? (T > 0 ? T : 1) : cus);
}

}  // namespace conv64
