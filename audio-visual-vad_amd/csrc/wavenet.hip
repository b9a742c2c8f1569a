// wavenet.hip -- WaveNet-style encoder (valid dilated Conv1d stack): forward and backward.
//
// Replaces wavenet_autoencoder._encode, packages/models/wavenet_autoencoder.py:74-93:
//   s0 = causal(wave)                                            (:75)
//   s_{i+1} = dense_i(relu(dil_i(relu(s_i)))) + s_i[:, :, -len:]   (:78-86)
//   out = AdaptiveAvgPool1d(P)(relu(bottleneck(s_N)))            (:88-92)
// Layout: [B][C][L] fp32, time contiguous (the reference's NCL).
//
// Production shape (R = D = 32, filter_width 2) runs a fused MFMA kernel per
// residual block: one wave owns a 32-sample time tile; lane half h of the wave
// streams tap h (x[c][t + h*d]) straight from HBM into the B operand of
// v_mfma_f32_32x32x2_f32 -- the dilation gather is the lane-half assignment,
// no LDS, no shuffles -- and the first GEMM's accumulator tile (rows = dilation
// channels in registers, column = time on the lane) is consumed in place as the
// B operand of the 1x1 "dense" GEMM (k-order permuted to the accumulator's row
// order).  ReLUs, both biases and the left-cropped residual are fused; each
// activation is read once (+ once for the residual, an L2 hit) and written once.
// The block kernels come in several forms of the same arithmetic (options wn_flat,
// wn_dx, wn_bwd_t): buffer-addressed with the weights read
// from LDS and 4 waves per SIMD (the default for short planes), 16-byte memory
// instructions over 128-sample super-tiles (long planes), resident weights with a
// cross-tile prefetch (beside another stream's kernels), flat addressing (round 1).
// Any other (R, D, filter_width) runs generic direct kernels.
// Weight gradients contract over (sequence, time) on the igemm engine.
#include "gemm_api.h"

namespace {

static inline int grid1(long n) { long b = (n + 255) / 256; return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

// ------------------------------------------------------------------ generic direct kernels
// y[b][co][t] = bias[co] + sum_{ci,k} w[co][ci][k] * f(x[b][ci][t + k*dil])  (+ res[b][co][t + res_off])
__global__ void conv1d_fwd_generic(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                   const float* __restrict__ res, float* __restrict__ y, int B, int Cin, int Cout, int Lin,
                                   int Lout, int fw, int dil, int relu_in, int res_off, int res_L) {
  const long n = (long)B * Cout * Lout;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i % Lout);
    const int co = (int)((i / Lout) % Cout);
    const int b = (int)(i / ((long)Lout * Cout));
    float acc = bias ? bias[co] : 0.f;
    const float* xb = x + (long)b * Cin * Lin + t;
    const float* wr = w + (long)co * Cin * fw;
    for (int ci = 0; ci < Cin; ++ci)
      for (int k = 0; k < fw; ++k) {
        float v = xb[(long)ci * Lin + k * dil];
        if (relu_in) v = fmaxf(v, 0.f);
        acc = fmaf(wr[ci * fw + k], v, acc);
      }
    if (res) acc += res[((long)b * Cout + co) * res_L + t + res_off];
    y[i] = acc;
  }
}

// dx[b][ci][ti] = mask(xmask>0) * sum_{co,k} w[co][ci][k] dy[b][co][ti - k*dil]   (+ res[b][ci][ti - res_off])
__global__ void conv1d_bwd_data_generic(const float* __restrict__ dy, const float* __restrict__ w,
                                        const float* __restrict__ xmask, const float* __restrict__ res, float* __restrict__ dx,
                                        int B, int Cin, int Cout, int Lin, int Lout, int fw, int dil, int res_off, int res_L) {
  const long n = (long)B * Cin * Lin;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int ti = (int)(i % Lin);
    const int ci = (int)((i / Lin) % Cin);
    const int b = (int)(i / ((long)Lin * Cin));
    float acc = 0.f;
    if (!xmask || xmask[i] > 0.f) {
      for (int k = 0; k < fw; ++k) {
        const int to = ti - k * dil;
        if (to < 0 || to >= Lout) continue;
        const float* dyb = dy + (long)b * Cout * Lout + to;
        for (int co = 0; co < Cout; ++co) acc = fmaf(w[((long)co * Cin + ci) * fw + k], dyb[(long)co * Lout], acc);
      }
    }
    if (res && ti >= res_off && ti - res_off < res_L) acc += res[((long)b * Cin + ci) * res_L + ti - res_off];
    dx[i] = acc;
  }
}

// db[c] += sum_{b,t} dy[b][c][t]
__global__ void __launch_bounds__(256) chan_sum_acc(const float* __restrict__ dy, float* __restrict__ db, int B, int C, int L) {
  __shared__ float sm[256];
  const int c = blockIdx.x;
  float s = 0.f;
  for (int b = blockIdx.y; b < B; b += gridDim.y) {
    const float* p = dy + ((long)b * C + c) * L;
    for (int t = threadIdx.x; t < L; t += 256) s += p[t];
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) db[c] += sm[0];     // gridDim.y == 1: one workgroup per channel, a fixed summation order
}

// Causal layer (few input channels, Cin*fw <= 4 weights per output channel): weight AND bias gradient in one pass over dy
//   dw[co][ci][k] += sum_{b,t} dy[b][co][t] x[b][ci][t + k]      db[co] += sum_{b,t} dy[b][co][t]
// (the engine's 64x64 tile on this 32 x 2 product ran 168 us; this reads dy once at HBM rate)
__global__ void __launch_bounds__(256)
    narrow_conv1d_grads(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ part,
                        int B, int Cout, int Cin, int Lout, int Lin, int fw) {
  __shared__ float sm[5][256];
  const int co = blockIdx.x, nw = Cin * fw;
  float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int b = blockIdx.y; b < B; b += gridDim.y) {
    const float* p = dy + ((long)b * Cout + co) * Lout;
    const float* xb = x + (long)b * Cin * Lin;
    for (int t = threadIdx.x; t < Lout; t += 256) {
      const float g = p[t];
      acc[4] += g;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < nw) acc[j] = fmaf(g, xb[(long)(j / fw) * Lin + t + (j % fw)], acc[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 5; ++j) sm[j][threadIdx.x] = acc[j];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o)
#pragma unroll
      for (int j = 0; j < 5; ++j) sm[j][threadIdx.x] += sm[j][threadIdx.x + o];
    __syncthreads();
  }
  // partial sums of this (channel, sequence group); narrow_conv1d_grads_sum adds the groups in order (deterministic)
  if (threadIdx.x < 5) part[((long)blockIdx.y * Cout + co) * 5 + threadIdx.x] = sm[threadIdx.x][0];
}
__global__ void narrow_conv1d_grads_sum(const float* __restrict__ part, int ny, int Cout, int nw, float* __restrict__ dw,
                                        float* __restrict__ db) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Cout * 5) return;
  const int co = i / 5, j = i - co * 5;
  float s = 0.f;
  for (int y = 0; y < ny; ++y) s += part[((long)y * Cout + co) * 5 + j];
  if (j < 4) { if (j < nw && dw) dw[(long)co * nw + j] += s; }
  else if (db) db[co] += s;
}

__device__ __forceinline__ void pool_bin(int p, int Lv, int P, int& a, int& e) {
  a = (int)(((long)p * Lv) / P);
  e = (int)((((long)(p + 1)) * Lv + P - 1) / P);
}

// out[b][bn][p] = mean_{t in bin p} relu(bb[bn] + sum_c Wb[bn][c] s[b][c][t])
__global__ void __launch_bounds__(256)
    tail_fwd_generic(const float* __restrict__ s, const float* __restrict__ wb, const float* __restrict__ bb,
                     float* __restrict__ out, int R, int Bn, int Lv, int P) {
  const int b = blockIdx.y, p = blockIdx.x;
  int a, e;
  pool_bin(p, Lv, P, a, e);
  const float* sb = s + (long)b * R * Lv;
  for (int bn = threadIdx.x; bn < Bn; bn += 256) {
    const float* wr = wb + (long)bn * R;
    const float bias = bb ? bb[bn] : 0.f;
    float sum = 0.f;
    for (int t = a; t < e; ++t) {
      float z = bias;
      for (int c = 0; c < R; ++c) z = fmaf(wr[c], sb[(long)c * Lv + t], z);
      sum += fmaxf(z, 0.f);
    }
    out[((long)b * Bn + bn) * P + p] = sum / (float)(e - a);
  }
}

// dz[b][bn][t] = (z>0) * sum_{p: t in bin p} dout[b][bn][p] / len_p
__global__ void tail_bwd_dz_generic(const float* __restrict__ s, const float* __restrict__ wb, const float* __restrict__ bb,
                                    const float* __restrict__ dout, float* __restrict__ dz, int B, int R, int Bn, int Lv, int P) {
  const long n = (long)B * Bn * Lv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i % Lv);
    const int bn = (int)((i / Lv) % Bn);
    const int b = (int)(i / ((long)Lv * Bn));
    float z = bb ? bb[bn] : 0.f;
    const float* sb = s + (long)b * R * Lv + t;
    for (int c = 0; c < R; ++c) z = fmaf(wb[(long)bn * R + c], sb[(long)c * Lv], z);
    float g = 0.f;
    if (z > 0.f) {
      const int pc = (int)(((long)t * P) / Lv);
      for (int p = max(0, pc - 1); p <= min(P - 1, pc + 1); ++p) {
        int a, e;
        pool_bin(p, Lv, P, a, e);
        if (t >= a && t < e) g += dout[((long)b * Bn + bn) * P + p] / (float)(e - a);
      }
    }
    dz[i] = g;
  }
}

// ------------------------------------------------------------------ fused MFMA residual block (R = D = 32, fw = 2)
// MODE 0: write s_out only; MODE 1: write s_out and z (pre-ReLU dilation output); MODE 2: write z only
template <int MODE>
__global__ void __launch_bounds__(256)
    wn_block_fwd_mfma(const float* __restrict__ s_in, const float* __restrict__ w_dil, const float* __restrict__ b_dil,
                      const float* __restrict__ w_dense, const float* __restrict__ b_dense, float* __restrict__ s_out,
                      float* __restrict__ z_out, int B, int Lin, int dil) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int Lo = Lin - dil;
  const int tiles_per_seq = (Lo + 31) >> 5;
  const long ntiles = (long)B * tiles_per_seq;
  const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;

  // A operands, resident for the whole kernel.  The per-lane fragment pattern is a 256-byte-strided gather, so
  // the block first copies the 3072 weights coalesced into (padded, conflict-free) LDS and picks them from there.
  __shared__ float wl[32 * 65 + 32 * 33 + 64];
  for (int i = threadIdx.x; i < 2048; i += 256) wl[(i >> 6) * 65 + (i & 63)] = w_dil[i];
  for (int i = threadIdx.x; i < 1024; i += 256) wl[2080 + (i >> 5) * 33 + (i & 31)] = w_dense[i];
  if (threadIdx.x < 32) {   // biases stay in LDS and seed the accumulators per tile (32 registers saved)
    wl[3136 + threadIdx.x] = b_dil ? b_dil[threadIdx.x] : 0.f;
    wl[3168 + threadIdx.x] = b_dense ? b_dense[threadIdx.x] : 0.f;
  }
  __syncthreads();
  float wd[32];   // W_dil[d = li][c = s][tap = lh]
#pragma unroll
  for (int s = 0; s < 32; ++s) wd[s] = wl[li * 65 + s * 2 + lh];
  float we[16];   // W_dense[r = li][d = row(r', lh)]
#pragma unroll
  for (int r = 0; r < 16; ++r) we[r] = wl[2080 + li * 33 + mfma32_row(r, lh)];
  const float* bzl = wl + 3136;
  const float* bsl = wl + 3168;

  // Software pipeline across tiles: the NEXT tile's 48 loads are issued before the current tile's 48 MFMAs, so
  // HBM latency hides under the matrix work inside one wave (the kernel is balanced HBM <-> MFMA).
  // Loads are UNCONDITIONAL on a clamped (always valid) address and masked afterwards: a per-element
  // "ok ? load : 0" makes hipcc branch around every load and wait for each one (32 serial round trips).
  float xn[32], rn[16];
  auto issue = [&](long tile) {
    const long tl = tile < ntiles ? tile : ntiles - 1;
    const int b = (int)(tl / tiles_per_seq);
    const int t = (int)(tl - (long)b * tiles_per_seq) * 32 + li;
    const int tcl = t < Lo ? t : 0;
    const float* xp = s_in + (long)b * 32 * Lin + tcl + lh * dil;
#pragma unroll
    for (int c = 0; c < 32; ++c) xn[c] = xp[(long)c * Lin];
    if (MODE <= 1) {
      const float* rp = s_in + (long)b * 32 * Lin + tcl + dil;
#pragma unroll
      for (int r = 0; r < 16; ++r) rn[r] = rp[(long)mfma32_row(r, lh) * Lin];
    }
  };
  if (wave0 < ntiles) issue(wave0);
  for (long tile = wave0; tile < ntiles; tile += nwaves) {
    const int b = (int)(tile / tiles_per_seq);
    const int t = (int)(tile - (long)b * tiles_per_seq) * 32 + li;
    const bool ok = t < Lo;
    float x[32], rv[16];
#pragma unroll
    for (int c = 0; c < 32; ++c) x[c] = ok ? xn[c] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) rv[r] = (MODE <= 1 && ok) ? rn[r] : 0.f;
    issue(tile + nwaves);     // in flight during the MFMAs below (clamped to the last tile when there is none)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bzl[mfma32_row(r, lh)];
#pragma unroll
    for (int s = 0; s < 32; ++s) acc = mfma32(wd[s], relu1(x[s]), acc);
    if (MODE >= 1) {
      float* zp = z_out + (long)b * 32 * Lo + t;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (ok) zp[(long)mfma32_row(r, lh) * Lo] = acc[r];
    }
    if (MODE <= 1) {
      f32x16 acc2;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[r] = bsl[mfma32_row(r, lh)] + rv[r];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2 = mfma32(we[r], relu1(acc[r]), acc2);
      float* op = s_out + (long)b * 32 * Lo + t;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (ok) op[(long)mfma32_row(r, lh) * Lo] = acc2[r];
    }
  }
}


// BUFFER-ADDRESSED form of the same block (the default; option "wn_flat" selects the kernel above).  On this chip the fp32
// MFMA runs on the vector lanes: every VALU instruction costs ~4 matrix-pipe cycles whatever the occupancy (tools/lab/
// mfvar_lab.py: a wave's MFMA chain at 0.90 of the pipe with ~50 VALU per tile, 0.71 with ~220, the same at 1, 2 and 3
// waves/SIMD).  The flat kernel above carries 718 VALU instructions per 48 MFMAs -- 139 64-bit address adds, 96 v_max (two
// per ReLU), 85 validity selects -- i.e. as many vector cycles as matrix cycles.  Here a tile's addressing is wave-uniform:
//   * one buffer descriptor per tensor and tile (SALU), ONE per-lane byte offset, the row in the instruction's scalar offset;
//   * lanes past the end of a sequence carry an out-of-range offset: their loads return 0 and their stores are dropped by
//     the bounds check, so no select touches the data;
//   * ReLU is one integer max (relu1);
//   * the cross-tile prefetch alternates between two register sets instead of copying.
template <int MODE>
__global__ void __launch_bounds__(256)
    wn_block_fwd_buf(const float* __restrict__ s_in, const float* __restrict__ w_dil, const float* __restrict__ b_dil,
                     const float* __restrict__ w_dense, const float* __restrict__ b_dense, float* __restrict__ s_out,
                     float* __restrict__ z_out, int B, int Lin, int dil) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int Lo = Lin - dil;
  const int tiles_per_seq = (Lo + 31) >> 5;
  const int ntiles = B * tiles_per_seq;
  const int wave0 = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  const int nwaves = (int)((gridDim.x * blockDim.x) >> 6);
  __shared__ float wl[32 * 65 + 32 * 33 + 64];
  float wd[32];   // W_dil[d = li][c = s][tap = lh]
  float we[16];   // W_dense[r = li][d = row(r', lh)]
  const float* bzl = wl + 3136 + 4 * lh;
  const float* bsl = wl + 3168 + 4 * lh;
  const int rowL = Lin * 4, rowO = Lo * 4;     // bytes per row of the input / output tensors

  // request tile `tile`'s 48 values (clamped to the last tile when there is none: the values are then never used)
  auto issue = [&](int tile, float (&xn)[32], float (&rn)[16]) {
    const int tl = tile < ntiles ? tile : ntiles - 1;
    const int b = tl / tiles_per_seq;
    const int t = (tl - b * tiles_per_seq) * 32 + li;
    const bool ok = t < Lo;
    const __amdgpu_buffer_rsrc_t rx = brsrc(s_in + (long)b * 32 * Lin, 32 * rowL);
    const int offx = ok ? (t + lh * dil) * 4 : BUF_OOB;
#pragma unroll
    for (int c = 0; c < 32; ++c) xn[c] = bload(rx, offx, c * rowL);
    if (MODE <= 1) {
      const int offr = ok ? (t + dil) * 4 + 4 * lh * rowL : BUF_OOB;
#pragma unroll
      for (int r = 0; r < 16; ++r) rn[r] = bload(rx, offr, mfma32_row(r, 0) * rowL);
    }
  };
  auto compute = [&](int tile, const float (&x)[32], const float (&rv)[16]) {
    const int b = tile / tiles_per_seq;
    const int t = (tile - b * tiles_per_seq) * 32 + li;
    const int offo = t < Lo ? t * 4 + 4 * lh * rowO : BUF_OOB;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bzl[mfma32_row(r, 0)];
#pragma unroll
    for (int s = 0; s < 32; ++s) acc = mfma32(wd[s], relu1(x[s]), acc);
    if (MODE >= 1) {
      const __amdgpu_buffer_rsrc_t rz = brsrc(z_out + (long)b * 32 * Lo, 32 * rowO);
#pragma unroll
      for (int r = 0; r < 16; ++r) bstore(acc[r], rz, offo, mfma32_row(r, 0) * rowO);
    }
    if (MODE <= 1) {
      f32x16 acc2;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[r] = bsl[mfma32_row(r, 0)] + rv[r];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2 = mfma32(we[r], relu1(acc[r]), acc2);
      const __amdgpu_buffer_rsrc_t ro = brsrc(s_out + (long)b * 32 * Lo, 32 * rowO);
#pragma unroll
      for (int r = 0; r < 16; ++r) bstore(acc2[r], ro, offo, mfma32_row(r, 0) * rowO);
    }
  };
  // XCD-aware walk: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so XCD x takes the contiguous
  // eighth [lo, hi) of the tiles and its waves sweep it side by side -- the second tap and the residual of a tile (the
  // same plane, `dil` samples later) are then lines that a neighbouring wave of the SAME XCD fetched a moment ago.  Dealt
  // out by raw wave id, tile j's neighbours j + dil/32 sat in other XCDs and every plane crossed the fabric ~1.7 times.
  int first = wave0, last = ntiles, stride = nwaves;
  if ((gridDim.x & 7) == 0) {
    const int xcd = blockIdx.x & 7, nbx = gridDim.x >> 3;
    first = (int)((long)ntiles * xcd / 8) + ((blockIdx.x >> 3) * 4 + (int)(threadIdx.x >> 6));
    last = (int)((long)ntiles * (xcd + 1) / 8);
    stride = nbx * 4;
  }
  first = __builtin_amdgcn_readfirstlane(first);
  float xa[32], ra[16], xb[32], rb[16];
  if (first < last) issue(first, xa, ra);
  // the first tile's requests are on their way while the block's weights make their own round trip (global -> LDS ->
  // registers): at the bench shape a launch lasts ~37 us and each of the two latencies is ~2 us of it
  for (int i = threadIdx.x; i < 2048; i += 256) wl[(i >> 6) * 65 + (i & 63)] = w_dil[i];
  for (int i = threadIdx.x; i < 1024; i += 256) wl[2080 + (i >> 5) * 33 + (i & 31)] = w_dense[i];
  if (threadIdx.x < 32) {
    wl[3136 + threadIdx.x] = b_dil ? b_dil[threadIdx.x] : 0.f;
    wl[3168 + threadIdx.x] = b_dense ? b_dense[threadIdx.x] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 32; ++s) wd[s] = wl[li * 65 + s * 2 + lh];
#pragma unroll
  for (int r = 0; r < 16; ++r) we[r] = wl[2080 + li * 33 + mfma32_row(r, lh)];
  for (int tile = first; tile < last; tile += 2 * stride) {
    issue(tile + stride, xb, rb);          // in flight during the MFMAs below (a tile past `last` is fetched, never used)
    compute(tile, xa, ra);
    if (tile + stride >= last) break;
    issue(tile + 2 * stride, xa, ra);
    compute(tile + stride, xb, rb);
  }
}

// HIGH-OCCUPANCY form (the default for planes shorter than 8192 samples): the same block with nothing held across tiles
// -- weights read from LDS per MFMA instead of 48 resident registers, no cross-tile prefetch registers -- so the kernel
// fits 4 waves per SIMD (< 128 VGPRs) and the memory latency of a tile hides under the OTHER waves' MFMAs instead of under
// this wave's own.  Measured (tools/lab/clk_lab2.py, same body): bench-shape layer 28.4 us at 4 waves/SIMD, 28.5 at 3,
// 31.0 at 2; C2-shape layer 273 / 303 / 380 us.  Earlier attempts at this shape (round 2, first half) ran the flat
// addressing: with ~700 VALU instructions per tile the extra waves only queued for the vector ALUs.
template <int MODE>
__global__ void __launch_bounds__(256, 4)
    wn_block_fwd_occ(const float* __restrict__ s_in, const float* __restrict__ w_dil, const float* __restrict__ b_dil,
                     const float* __restrict__ w_dense, const float* __restrict__ b_dense, float* __restrict__ s_out,
                     float* __restrict__ z_out, int B, int Lin, int dil) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int Lo = Lin - dil;
  const int tiles_per_seq = (Lo + 31) >> 5;
  const int ntiles = B * tiles_per_seq;
  __shared__ float wl[32 * 65 + 32 * 33 + 64];
  for (int i = threadIdx.x; i < 2048; i += 256) wl[(i >> 6) * 65 + (i & 63)] = w_dil[i];
  for (int i = threadIdx.x; i < 1024; i += 256) wl[2080 + (i >> 5) * 33 + (i & 31)] = w_dense[i];
  if (threadIdx.x < 32) {
    wl[3136 + threadIdx.x] = b_dil ? b_dil[threadIdx.x] : 0.f;
    wl[3168 + threadIdx.x] = b_dense ? b_dense[threadIdx.x] : 0.f;
  }
  __syncthreads();
  const float* wdl = wl + li * 65 + lh;              // W_dil[d = li][c = s][tap = lh] at wdl[2 s]
  const float* wel = wl + 2080 + li * 33 + 4 * lh;   // W_dense[r = li][d = row(r', lh)] at wel[row(r', 0)]
  const float* bzl = wl + 3136 + 4 * lh;
  const float* bsl = wl + 3168 + 4 * lh;
  const int rowL = Lin * 4, rowO = Lo * 4;
  const TileWalk tw = xcd_walk(ntiles);
  const int first = __builtin_amdgcn_readfirstlane((int)tw.first), last = (int)tw.last, stride = (int)tw.stride;
  for (int tile = first; tile < last; tile += stride) {
    const int b = tile / tiles_per_seq;
    const int t = (tile - b * tiles_per_seq) * 32 + li;
    const bool ok = t < Lo;
    const __amdgpu_buffer_rsrc_t rx = brsrc(s_in + (long)b * 32 * Lin, 32 * rowL);
    const int offx = ok ? (t + lh * dil) * 4 : BUF_OOB;
    float x[32], rv[16];
#pragma unroll
    for (int c = 0; c < 32; ++c) x[c] = bload(rx, offx, c * rowL);
    if (MODE <= 1) {
      const int offr = ok ? (t + dil) * 4 + 4 * lh * rowL : BUF_OOB;
#pragma unroll
      for (int r = 0; r < 16; ++r) rv[r] = bload(rx, offr, mfma32_row(r, 0) * rowL);
    }
    const int offo = ok ? t * 4 + 4 * lh * rowO : BUF_OOB;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bzl[mfma32_row(r, 0)];
#pragma unroll
    for (int s = 0; s < 32; ++s) acc = mfma32(wdl[2 * s], relu1(x[s]), acc);
    if (MODE >= 1) {
      const __amdgpu_buffer_rsrc_t rz = brsrc(z_out + (long)b * 32 * Lo, 32 * rowO);
#pragma unroll
      for (int r = 0; r < 16; ++r) bstore(acc[r], rz, offo, mfma32_row(r, 0) * rowO);
    }
    if (MODE <= 1) {
      f32x16 acc2;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[r] = bsl[mfma32_row(r, 0)] + rv[r];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2 = mfma32(wel[mfma32_row(r, 0)], relu1(acc[r]), acc2);
      const __amdgpu_buffer_rsrc_t ro = brsrc(s_out + (long)b * 32 * Lo, 32 * rowO);
#pragma unroll
      for (int r = 0; r < 16; ++r) bstore(acc2[r], ro, offo, mfma32_row(r, 0) * rowO);
    }
  }
}

// LDS-DMA form.  What bounds the dword kernels above is the CU's ADDRESS unit, shared by the four SIMDs: a wave memory
// instruction costs it ~16 cycles whatever its width, a 32-sample tile issues 64 of them (32 operand loads, 16 residual loads,
// 16 stores) = ~1000 cycles per tile and CU against 768 cycles of matrix work (48 MFMAs x 64 cycles / 4 SIMDs) -- the memory
// phase alone takes 20 us per bench-shape layer (12 200 tiles / 256 CUs x 1024 cycles), and at launch every resident wave
// queues its 48 loads at once, so the first MFMA of a CU waits for ~12 000 cycles of address work.  Here a tile's operands
// arrive by EIGHT buffer_load_dwordx4 ... lds instructions (1 KiB each: 8 channel rows x 128 B, straight into the wave's own
// 8 KB LDS tile, no VGPR destination): tap 0 and tap 1 as two [32 channels][32 samples] images.  The MFMA B operand of
// k-step s is then one conflict-free ds_read_b32 (lane half h = tap h, row s), the residual rows are 16 more reads of the
// tap-1 image, and the address unit sees 8 + 16 instructions per tile (384 cycles) -- the kernel is bound by the matrix
// pipe.  Same MFMA sequence and accumulator initialisation as wn_block_fwd_occ: the results agree bit for bit.
// One workgroup = 8 waves (weights once per workgroup in LDS, 64 KB of wave tiles): two per CU = 4 waves per SIMD.
__global__ void __launch_bounds__(512, 4)
    wn_block_fwd_dma(const float* __restrict__ s_in, const float* __restrict__ w_dil, const float* __restrict__ b_dil,
                     const float* __restrict__ w_dense, const float* __restrict__ b_dense, float* __restrict__ s_out, int B, int Lin,
                     int dil) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int Lo = Lin - dil;
  const int tiles_per_seq = (Lo + 31) >> 5;
  const int ntiles = B * tiles_per_seq;
  __shared__ __attribute__((aligned(16))) float wl[8 * 2048 + 32 * 65 + 32 * 33 + 64];     // wave tiles first (16-byte aligned DMA targets)
  float* const wts = wl + 8 * 2048;
  for (int i = threadIdx.x; i < 2048; i += 512) wts[(i >> 6) * 65 + (i & 63)] = w_dil[i];
  for (int i = threadIdx.x; i < 1024; i += 512) wts[2080 + (i >> 5) * 33 + (i & 31)] = w_dense[i];
  if (threadIdx.x < 32) {
    wts[3136 + threadIdx.x] = b_dil ? b_dil[threadIdx.x] : 0.f;
    wts[3168 + threadIdx.x] = b_dense ? b_dense[threadIdx.x] : 0.f;
  }
  __syncthreads();
  const float* wdl = wts + li * 65 + lh;              // W_dil[d = li][c = s][tap = lh] at wdl[2 s]
  const float* wel = wts + 2080 + li * 33 + 4 * lh;   // W_dense[r = li][d = row(r', lh)] at wel[row(r', 0)]
  const float* bzl = wts + 3136 + 4 * lh;
  const float* bsl = wts + 3168 + 4 * lh;
  const int rowL = Lin * 4, rowO = Lo * 4;
  float* const tb = wl + __builtin_amdgcn_readfirstlane(wave) * 2048;        // this wave's tile: [tap][32 channels][32 samples]
  const float* const xs = tb + lh * 1024 + li;                               // B operand of k-step s: xs[32 s]
  const float* const rs = tb + 1024 + 4 * lh * 32 + li;                       // residual row (r, lh): rs[32 row(r, 0)]
  // the DMA's per-lane source: channel row (lane >> 3) of the instruction's 8, samples 4 (lane & 7) .. + 3
  const int dvo = (lane >> 3) * rowL + (lane & 7) * 16;
  // tile walk: waves of one XCD sweep a contiguous eighth of the tiles (see xcd_walk; 8 waves per workgroup here)
  int first = (int)(blockIdx.x * 8 + wave), last = ntiles, stride = (int)(gridDim.x * 8);
  if ((gridDim.x & 7) == 0) {
    const int xcd = blockIdx.x & 7;
    first = (int)((long)ntiles * xcd / 8) + ((blockIdx.x >> 3) * 8 + wave);
    last = (int)((long)ntiles * (xcd + 1) / 8);
    stride = (gridDim.x >> 3) * 8;
  }
  first = __builtin_amdgcn_readfirstlane(first);
  for (int tile = first; tile < last; tile += stride) {
    const int b = tile / tiles_per_seq;
    const int t0 = (tile - b * tiles_per_seq) * 32;
    const __amdgpu_buffer_rsrc_t rx = brsrc(s_in + (long)b * 32 * Lin, 32 * rowL);
    // samples past the end of a row read into the next row, past the slab nothing is written: garbage COLUMNS, whose
    // results are never stored (an output column depends on its own input columns only)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(tb + q * 256), 16, dvo + t0 * 4,
                                               8 * q * rowL, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)(tb + 1024 + q * 256), 16,
                                               dvo + (t0 + dil) * 4, 8 * q * rowL, 0, 0);
    const int t = t0 + li;
    const bool ok = t < Lo;
    const int offo = ok ? t * 4 + 4 * lh * rowO : BUF_OOB;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bzl[mfma32_row(r, 0)];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the tile has landed (and the previous tile's stores have drained)
    __builtin_amdgcn_sched_barrier(0);
    // Operands of 8 k-steps at a time, the NEXT group's LDS reads in flight under this group's MFMAs (pinned: left alone,
    // hipcc issues two reads, waits for them and multiplies twice -- the LDS latency in front of every MFMA pair)
    float wv[2][8], xv[2][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { wv[0][k] = wdl[2 * k]; xv[0][k] = xs[32 * k]; }
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      if (gq < 3) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { wv[(gq + 1) & 1][k] = wdl[2 * (8 * (gq + 1) + k)]; xv[(gq + 1) & 1][k] = xs[32 * (8 * (gq + 1) + k)]; }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc = mfma32(wv[gq & 1][k], relu1(xv[gq & 1][k]), acc);
    }
    // the 1x1 product: its 16 weights and the 16 residual rows are read while the last dilated MFMAs run
    float we16[16];
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { we16[r] = wel[mfma32_row(r, 0)]; acc2[r] = bsl[mfma32_row(r, 0)] + rs[32 * mfma32_row(r, 0)]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2 = mfma32(we16[r], relu1(acc[r]), acc2);
    const __amdgpu_buffer_rsrc_t ro = brsrc(s_out + (long)b * 32 * Lo, 32 * rowO);
#pragma unroll
    for (int r = 0; r < 16; ++r) bstore(acc2[r], ro, offo, mfma32_row(r, 0) * rowO);
    // (the next tile's DMA overwrites this wave's LDS tile: every read of it above has been consumed by an MFMA or an add
    //  that precedes the DMA in program order)
    __builtin_amdgcn_sched_barrier(0);
  }
}

// WIDE form (the default for planes of >= 128 samples): the same block with 16-byte memory instructions.  The dword
// kernels above are ISSUE-bound in the memory pipe -- 64 one-dword wave-instructions per 32-sample tile keep the CU's
// address unit busier (~1000 cycles) than the tile's 48 MFMAs keep the matrix pipe (768), and a wave that is stuck issuing
// its loads issues no MFMAs behind them either -- so here one wave owns a 128-sample SUPER-tile as four interleaved
// sub-tiles (sub-tile j = samples t0 + 4n + j: columns are independent, so any set of 32 columns is a valid MFMA tile):
// lane n holds 4 consecutive samples, every load / store is a dwordx4 (rows start at any sample: 4-byte aligned 16-byte
// accesses), and the instruction count per sample drops 4x.  The weights come from LDS per MFMA group (one ds_read feeds
// four MFMAs), the residual is fetched (an L1/L2 hit: the tap-1 lanes just read those lines) into the x registers once the
// dilated product is done.
__global__ void __launch_bounds__(256, 2)
    wn_block_fwd_w4(const float* __restrict__ s_in, const float* __restrict__ w_dil, const float* __restrict__ b_dil,
                    const float* __restrict__ w_dense, const float* __restrict__ b_dense, float* __restrict__ s_out, int B,
                    int Lin, int dil) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int Lo = Lin - dil;
  const int tiles_per_seq = (Lo + 127) >> 7;
  const int ntiles = B * tiles_per_seq;
  __shared__ float wl[32 * 65 + 32 * 33 + 64];
  for (int i = threadIdx.x; i < 2048; i += 256) wl[(i >> 6) * 65 + (i & 63)] = w_dil[i];
  for (int i = threadIdx.x; i < 1024; i += 256) wl[2080 + (i >> 5) * 33 + (i & 31)] = w_dense[i];
  if (threadIdx.x < 32) {
    wl[3136 + threadIdx.x] = b_dil ? b_dil[threadIdx.x] : 0.f;
    wl[3168 + threadIdx.x] = b_dense ? b_dense[threadIdx.x] : 0.f;
  }
  __syncthreads();
  const float* wdl = wl + li * 65 + lh;            // W_dil[d = li][c = s][tap = lh] at wdl[2 s]
  const float* wel = wl + 2080 + li * 33 + 4 * lh; // W_dense[r = li][d = row(r', lh)] at wel[row(r', 0)]
  const float* bzl = wl + 3136 + 4 * lh;
  const float* bsl = wl + 3168 + 4 * lh;
  const int rowL = Lin * 4, rowO = Lo * 4;

  // XCD-aware walk (see wn_block_fwd_buf)
  int first = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), last = ntiles, stride = (int)((gridDim.x * blockDim.x) >> 6);
  if ((gridDim.x & 7) == 0) {
    const int xcd = blockIdx.x & 7, nbx = gridDim.x >> 3;
    first = (int)((long)ntiles * xcd / 8) + ((blockIdx.x >> 3) * 4 + (int)(threadIdx.x >> 6));
    last = (int)((long)ntiles * (xcd + 1) / 8);
    stride = nbx * 4;
  }
  first = __builtin_amdgcn_readfirstlane(first);
  for (int tile = first; tile < last; tile += stride) {
    const int b = tile / tiles_per_seq;
    const int t0 = (tile - b * tiles_per_seq) * 128;
    const int t = t0 + 4 * li;                     // this lane's first sample
    const __amdgpu_buffer_rsrc_t rx = brsrc(s_in + (long)b * 32 * Lin, 32 * rowL);
    const __amdgpu_buffer_rsrc_t ro = brsrc(s_out + (long)b * 32 * Lo, 32 * rowO);
    // samples past the end of a row read into the next row (or return 0 past the slab): garbage columns, never stored
    const int offx = (t + lh * dil) * 4;
    f4v x[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) x[c] = bload4(rx, offx, c * rowL);
    // all 32 requests go out BEFORE the first MFMA (left alone, hipcc hoists the weights into 48 registers, has too few left
    // for x and re-fetches it two loads at a time between the MFMAs: a memory round trip per four MFMAs).  The weight
    // pointers are laundered per tile so that their LDS reads stay inside the loop.
    __builtin_amdgcn_sched_barrier(0);
    const float* wd_t = wdl;
    const float* we_t = wel;
    asm volatile("" : "+v"(wd_t), "+v"(we_t));
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = bzl[mfma32_row(r, 0)];
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      const float w = wd_t[2 * s];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = mfma32(w, relu1(x[s][j]), acc[j]);
    }
    // residual rows (same plane, `dil` samples later: the lines the tap-1 half just read), into registers x no longer needs
    const int offr = (t + dil) * 4 + 4 * lh * rowL;
    f4v rv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rv[r] = bload4(rx, offr, mfma32_row(r, 0) * rowL);
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[j][r] = bsl[mfma32_row(r, 0)];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float w = we_t[mfma32_row(r, 0)];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc2[j] = mfma32(w, relu1(acc[j][r]), acc2[j]);
    }
    const int offo = t * 4 + 4 * lh * rowO;
    if (t0 + 128 <= Lo) {
      // All 16 sums first (in place of rv), THEN the 16 stores, nothing writing their data registers behind them.
      // gfx950 hazard the compiler does not know: a buffer_store_dwordx4 whose soffset is an SGPR still reads its data
      // registers a moment after issue, exactly like the immediate-soffset form hipcc pads with a wait state.  With the sum
      // of row r+1 computed into the registers of row r's store right behind it, 16 lanes of the store's first dword were
      // the NEXT row's value -- only with two waves per SIMD (the other wave's traffic delays the read), only rows whose
      // store had an SGPR soffset and a successor: r = 1..14 (tests: test_wavenet_block_kernel_forms_agree).
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const f4v o = {acc2[0][r] + rv[r][0], acc2[1][r] + rv[r][1], acc2[2][r] + rv[r][2], acc2[3][r] + rv[r][3]};
        rv[r] = o;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 16; ++r) bstore4(rv[r], ro, offo, mfma32_row(r, 0) * rowO);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 1");
    } else {          // last super-tile of a sequence: element-wise, out-of-range samples dropped by the bounds check
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int oj = t + j < Lo ? offo + 4 * j : BUF_OOB;
#pragma unroll
        for (int r = 0; r < 16; ++r) bstore(acc2[j][r] + rv[r][j], ro, oj, mfma32_row(r, 0) * rowO);
      }
    }
  }
}

// (Round 2 also tried two other shapes of the FLAT kernel -- tools/lab/mem_lab.py holds the measurements: a 3-waves-per-SIMD form
//  with the weight fragments read from LDS and no cross-tile prefetch, and one with wave-uniform scalar/buffer addressing
//  (the flat form spends ~1900 VALU cycles per tile on 64-bit per-lane address arithmetic, and on this chip VALU time adds
//  to fp32-MFMA time: the matrix instruction runs on the vector lanes).  Both timed within 3 % of this kernel at the bench
//  and C2 shapes: its ~36 us per bench-shape layer are the SUM of the access pattern's streaming time (20 us alone) and the
//  matrix time (15 us alone), and neither occupancy nor fewer VALU instructions made the two overlap.  Nor did pinning the
//  next tile's 48 loads in front of this tile's MFMAs with sched_barrier -- hipcc sinks them below the first 32 MFMAs
//  otherwise -- which timed 3 % slower.)

// ------------------------------------------------------------------ fused MFMA backward of a residual block (R = D = 32, fw = 2)
// (A)  dz[d][t] = (z[d][t] > 0) * sum_r W_dense[r][d] * dS[r][t]            -- 16 MFMAs per 32-sample tile
// Same lane geometry as the forward: column = time on the lane; the K index r = 2s+h is fed by lane half h.
__global__ void __launch_bounds__(256)
    wn_block_bwd_dz_mfma(const float* __restrict__ dS, const float* __restrict__ Z, const float* __restrict__ w_dense,
                         float* __restrict__ DZ, int B, int Lo) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int tiles_per_seq = (Lo + 31) >> 5;
  const long ntiles = (long)B * tiles_per_seq;
  const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  __shared__ float wl[32 * 33];
  for (int i = threadIdx.x; i < 1024; i += 256) wl[(i >> 5) * 33 + (i & 31)] = w_dense[i];
  __syncthreads();
  float wt[16];  // A[i = d = li][k = r = 2s+lh] = W_dense[r][d]
#pragma unroll
  for (int s = 0; s < 16; ++s) wt[s] = wl[(2 * s + lh) * 33 + li];
  for (long tile = wave0; tile < ntiles; tile += nwaves) {
    const int b = (int)(tile / tiles_per_seq);
    const int t = (int)(tile - (long)b * tiles_per_seq) * 32 + li;
    const bool ok = t < Lo;
    const long base = (long)b * 32 * Lo + (ok ? t : 0);   // clamped: loads are unconditional, masked below
    float g[16], z[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) g[s] = dS[base + (long)(2 * s + lh) * Lo];
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = Z[base + (long)mfma32_row(r, lh) * Lo];
#pragma unroll
    for (int s = 0; s < 16; ++s) g[s] = ok ? g[s] : 0.f;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = mfma32(wt[s], g[s], acc);
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (ok) DZ[base + (long)mfma32_row(r, lh) * Lo] = z[r] > 0.f ? acc[r] : 0.f;
  }
}

// (B)  dS_in[c][t'] = (s_in[c][t'] > 0) * sum_{d,k} W_dil[d][c][k] * dz[d][t' - k*dil]  +  dS_out[c][t' - dil]
//      (terms whose index falls outside [0, Lo) are zero)                        -- 32 MFMAs per tile
__global__ void __launch_bounds__(256)
    wn_block_bwd_dx_mfma(const float* __restrict__ DZ, const float* __restrict__ s_in, const float* __restrict__ dS_out,
                         const float* __restrict__ w_dil, float* __restrict__ dS_in, int B, int Lin, int dil) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int Lo = Lin - dil;
  const int tiles_per_seq = (Lin + 31) >> 5;
  const long ntiles = (long)B * tiles_per_seq;
  const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  __shared__ float wl[32 * 65];
  for (int i = threadIdx.x; i < 2048; i += 256) wl[(i >> 6) * 65 + (i & 63)] = w_dil[i];
  __syncthreads();
  float wt[32];  // A[i = c = li][k = (d = s, tap = lh)] = W_dil[d][c][tap]
#pragma unroll
  for (int s = 0; s < 32; ++s) wt[s] = wl[s * 65 + li * 2 + lh];
  // software pipeline across tiles (see wn_block_fwd_mfma): next tile's loads fly under this tile's MFMAs
  float dzn[32], svn[16], rvn[16];
  auto issue = [&](long tile) {
    const long tl = tile < ntiles ? tile : ntiles - 1;
    const int b = (int)(tl / tiles_per_seq);
    const int t = (int)(tl - (long)b * tiles_per_seq) * 32 + li;
    const bool ok = t < Lin;
    const int to = t - lh * dil;
    const bool okz = ok && to >= 0 && to < Lo;
    const float* zp = DZ + (long)b * 32 * Lo + (okz ? to : 0);
#pragma unroll
    for (int s = 0; s < 32; ++s) dzn[s] = zp[(long)s * Lo];
    const bool okr = ok && t >= dil;
    const float* sp = s_in + (long)b * 32 * Lin + (ok ? t : 0);
    const float* rp = dS_out + (long)b * 32 * Lo + (okr ? t - dil : 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) { svn[r] = sp[(long)mfma32_row(r, lh) * Lin]; rvn[r] = rp[(long)mfma32_row(r, lh) * Lo]; }
  };
  const TileWalk tw = xcd_walk(ntiles);
  if (tw.first < tw.last) issue(tw.first);
  for (long tile = tw.first; tile < tw.last; tile += tw.stride) {
    const int b = (int)(tile / tiles_per_seq);
    const int t = (int)(tile - (long)b * tiles_per_seq) * 32 + li;   // t' (input time)
    const bool ok = t < Lin;
    const int to = t - lh * dil;                                      // dz time of this lane half's tap
    const bool okz = ok && to >= 0 && to < Lo;
    const bool okr = ok && t >= dil;                                  // residual: dS_out[c][t' - dil]
    float dz[32], sv[16], rv[16];
#pragma unroll
    for (int s = 0; s < 32; ++s) dz[s] = okz ? dzn[s] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sv[r] = svn[r]; rv[r] = okr ? rvn[r] : 0.f; }
    issue(tile + tw.stride);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 32; ++s) acc = mfma32(wt[s], dz[s], acc);
    float* op = dS_in + (long)b * 32 * Lin + t;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (ok) op[(long)mfma32_row(r, lh) * Lin] = (sv[r] > 0.f ? acc[r] : 0.f) + rv[r];
  }
}

// BUFFER-ADDRESSED form of (B) with the weights resident and the next tile's requests in flight under this tile's MFMAs:
// the low-occupancy form picked beside another stream's kernels (descriptor hint shared_device), where the flat kernel
// above ran until now -- 720 VALU instructions per 32 MFMAs, most of them 64-bit address arithmetic and validity selects.
__global__ void __launch_bounds__(256)
    wn_block_bwd_dx_buf(const float* __restrict__ DZ, const float* __restrict__ s_in, const float* __restrict__ dS_out,
                        const float* __restrict__ w_dil, float* __restrict__ dS_in, int B, int Lin, int dil) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int Lo = Lin - dil;
  const int tiles_per_seq = (Lin + 31) >> 5;
  const int ntiles = B * tiles_per_seq;
  __shared__ float wl[32 * 65];
  const int rowL = Lin * 4, rowO = Lo * 4;
  // request tile `tile` (clamped to the last tile when there is none: the values are then never used)
  auto issue = [&](int tile, float (&dzn)[32], float (&svn)[16], float (&rvn)[16]) {
    const int tc = tile < ntiles ? tile : ntiles - 1;
    const int b = tc / tiles_per_seq;
    const int t = (tc - b * tiles_per_seq) * 32 + li;
    const bool ok = t < Lin;
    const int to = t - lh * dil;
    const bool okz = ok && to >= 0 && to < Lo;
    const bool okr = ok && t >= dil;
    const __amdgpu_buffer_rsrc_t rz = brsrc(DZ + (long)b * 32 * Lo, 32 * rowO);
    const __amdgpu_buffer_rsrc_t rs = brsrc(s_in + (long)b * 32 * Lin, 32 * rowL);
    const __amdgpu_buffer_rsrc_t rg = brsrc(dS_out + (long)b * 32 * Lo, 32 * rowO);
    const int offz = okz ? to * 4 : BUF_OOB;
    const int offs = ok ? t * 4 + 4 * lh * rowL : BUF_OOB;
    const int offg = okr ? (t - dil) * 4 + 4 * lh * rowO : BUF_OOB;
#pragma unroll
    for (int s = 0; s < 32; ++s) dzn[s] = bload(rz, offz, s * rowO);
#pragma unroll
    for (int r = 0; r < 16; ++r) { svn[r] = bload(rs, offs, mfma32_row(r, 0) * rowL); rvn[r] = bload(rg, offg, mfma32_row(r, 0) * rowO); }
  };
  const TileWalk tw = xcd_walk(ntiles);
  const int first = __builtin_amdgcn_readfirstlane((int)tw.first), last = (int)tw.last, stride = (int)tw.stride;
  float dza[32], sva[16], rva[16], dzb[32], svb[16], rvb[16];
  if (first < last) issue(first, dza, sva, rva);
  for (int i = threadIdx.x; i < 2048; i += 256) wl[(i >> 6) * 65 + (i & 63)] = w_dil[i];
  __syncthreads();
  float wt[32];  // A[i = c = li][k = (d = s, tap = lh)] = W_dil[d][c][tap]
#pragma unroll
  for (int s = 0; s < 32; ++s) wt[s] = wl[s * 65 + li * 2 + lh];
  auto compute = [&](int tile, const float (&dz)[32], const float (&sv)[16], const float (&rv)[16]) {
    const int b = tile / tiles_per_seq;
    const int t = (tile - b * tiles_per_seq) * 32 + li;
    const __amdgpu_buffer_rsrc_t ro = brsrc(dS_in + (long)b * 32 * Lin, 32 * rowL);
    const int offs = t < Lin ? t * 4 + 4 * lh * rowL : BUF_OOB;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 32; ++s) acc = mfma32(wt[s], dz[s], acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) bstore((sv[r] > 0.f ? acc[r] : 0.f) + rv[r], ro, offs, mfma32_row(r, 0) * rowL);
  };
  for (int tile = first; tile < last; tile += 2 * stride) {
    issue(tile + stride, dzb, svb, rvb);
    compute(tile, dza, sva, rva);
    if (tile + stride >= last) break;
    issue(tile + 2 * stride, dza, sva, rva);
    compute(tile + stride, dzb, svb, rvb);
  }
}

// HIGH-OCCUPANCY form of (B), the default (see wn_block_fwd_occ): buffer addressing, weights read from LDS per MFMA,
// nothing held across tiles -> < 128 VGPRs, 4 waves per SIMD.  Same arithmetic, same summation order.
__global__ void __launch_bounds__(256, 4)
    wn_block_bwd_dx_occ(const float* __restrict__ DZ, const float* __restrict__ s_in, const float* __restrict__ dS_out,
                        const float* __restrict__ w_dil, float* __restrict__ dS_in, int B, int Lin, int dil) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int Lo = Lin - dil;
  const int tiles_per_seq = (Lin + 31) >> 5;
  const int ntiles = B * tiles_per_seq;
  __shared__ float wl[32 * 65];
  for (int i = threadIdx.x; i < 2048; i += 256) wl[(i >> 6) * 65 + (i & 63)] = w_dil[i];
  __syncthreads();
  const float* wtl = wl + li * 2 + lh;   // A[i = c = li][k = (d = s, tap = lh)] = W_dil[d][c][tap] at wtl[65 s]
  const int rowL = Lin * 4, rowO = Lo * 4;
  const TileWalk tw = xcd_walk(ntiles);
  const int first = __builtin_amdgcn_readfirstlane((int)tw.first), last = (int)tw.last, stride = (int)tw.stride;
  for (int tile = first; tile < last; tile += stride) {
    const int b = tile / tiles_per_seq;
    const int t = (tile - b * tiles_per_seq) * 32 + li;   // t' (input time)
    const bool ok = t < Lin;
    const int to = t - lh * dil;                          // dz time of this lane half's tap
    const bool okz = ok && to >= 0 && to < Lo;
    const bool okr = ok && t >= dil;                      // residual: dS_out[c][t' - dil]
    const __amdgpu_buffer_rsrc_t rz = brsrc(DZ + (long)b * 32 * Lo, 32 * rowO);
    const __amdgpu_buffer_rsrc_t rs = brsrc(s_in + (long)b * 32 * Lin, 32 * rowL);
    const __amdgpu_buffer_rsrc_t rg = brsrc(dS_out + (long)b * 32 * Lo, 32 * rowO);
    const __amdgpu_buffer_rsrc_t ro = brsrc(dS_in + (long)b * 32 * Lin, 32 * rowL);
    const int offz = okz ? to * 4 : BUF_OOB;
    const int offs = ok ? t * 4 + 4 * lh * rowL : BUF_OOB;
    const int offg = okr ? (t - dil) * 4 + 4 * lh * rowO : BUF_OOB;
    float dz[32], sv[16], rv[16];
#pragma unroll
    for (int s = 0; s < 32; ++s) dz[s] = bload(rz, offz, s * rowO);
#pragma unroll
    for (int r = 0; r < 16; ++r) { sv[r] = bload(rs, offs, mfma32_row(r, 0) * rowL); rv[r] = bload(rg, offg, mfma32_row(r, 0) * rowO); }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 32; ++s) acc = mfma32(wtl[65 * s], dz[s], acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) bstore((sv[r] > 0.f ? acc[r] : 0.f) + rv[r], ro, offs, mfma32_row(r, 0) * rowL);
  }
}

// ------------------------------------------------------------------ MFMA bottleneck + ReLU + adaptive average pool (R = 32, Bn % 32 == 0)
// One wave per (sequence, pool bin, group of NT bn-tiles).  Transposed product z^T[t][bn] = s^T . Wb^T: the
// activation tile is the A operand (lane = time, loaded as it lies in memory), the weights the B operand, so the
// accumulator has TIME IN ITS REGISTERS and bn on the lane: the pool is 16 register adds + one lane-half swap.
template <int NT>
__global__ void __launch_bounds__(256)
    tail_fwd_mfma(const float* __restrict__ s, const float* __restrict__ wb, const float* __restrict__ bb,
                  float* __restrict__ out, int B, int Bn, int Lv, int P) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int ngrp = Bn / (32 * NT);
  const long nitems = (long)B * P * ngrp;
  const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  for (long item = wave0; item < nitems; item += nwaves) {
    const int grp = (int)(item % ngrp);
    const int p = (int)((item / ngrp) % P);
    const int b = (int)(item / ((long)ngrp * P));
    const int bn0 = grp * 32 * NT;
    int a, e;
    pool_bin(p, Lv, P, a, e);
    float w[NT][16], bias[NT], sum[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
#pragma unroll
      for (int k = 0; k < 16; ++k) w[n][k] = wb[(long)(bn0 + n * 32 + li) * 32 + 2 * k + lh];
      bias[n] = bb ? bb[bn0 + n * 32 + li] : 0.f;
      sum[n] = 0.f;
    }
    const float* sp = s + (long)b * 32 * Lv;
    for (int t0 = a; t0 < e; t0 += 32) {
      const int t = t0 + li;
      const bool ok = t < e;
      const int tcl = ok ? t : a;
      float x[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) x[k] = sp[(long)(2 * k + lh) * Lv + tcl];
#pragma unroll
      for (int k = 0; k < 16; ++k) x[k] = ok ? x[k] : 0.f;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = mfma32(x[k], w[n][k], acc);
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t0 + mfma32_row(r, lh) < e) sum[n] += relu1(acc[r] + bias[n]);
      }
    }
    const float inv = 1.f / (float)(e - a);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const float v = sum[n] + __shfl_xor(sum[n], 32, 64);
      if (lh == 0) out[((long)b * Bn + bn0 + n * 32 + li) * P + p] = v * inv;
    }
  }
}

// backward of the tail for one 32-sample tile per wave: for every bn-tile recompute z (16 MFMAs), form
// dz = (z>0) * pooled upstream gradient (written out for the weight gradients) and accumulate
// dS[c][t] += sum_bn Wb[bn][c] dz[bn][t] (16 MFMAs, the z accumulator tile reused in place as the B operand).
__global__ void __launch_bounds__(256)
    tail_bwd_mfma(const float* __restrict__ s, const float* __restrict__ wb, const float* __restrict__ bb,
                  const float* __restrict__ dout, float* __restrict__ dzt, float* __restrict__ dS, int B, int Bn, int Lv,
                  int P) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int tiles_per_seq = (Lv + 31) >> 5;
  const long ntiles = (long)B * tiles_per_seq;
  const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  // the bottleneck weights (Bn x 32) are copied coalesced into padded LDS once per workgroup: the per-lane fragment
  // pattern is a 128-byte-strided gather (32 cache lines per load instruction) that made this kernel TA-bound
  extern __shared__ float wl[];
  for (int i = threadIdx.x; i < Bn * 32; i += 256) wl[(i >> 5) * 33 + (i & 31)] = wb[i];
  __syncthreads();
  for (long tile = wave0; tile < ntiles; tile += nwaves) {
    const int b = (int)(tile / tiles_per_seq);
    const int t = (int)(tile - (long)b * tiles_per_seq) * 32 + li;
    const bool ok = t < Lv;
    const float* sp = s + (long)b * 32 * Lv + (ok ? t : 0);
    float x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = sp[(long)(2 * k + lh) * Lv];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = ok ? x[k] : 0.f;
    // pool bins that contain t (at most two adjacent ones overlap)
    int pb[3];
    float pc[3];
    {
      const int c0 = ok ? (int)(((long)t * P) / Lv) : 0;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int p = c0 - 1 + j;
        int a = 0, e = 0;
        const bool in = ok && p >= 0 && p < P;
        if (in) pool_bin(p, Lv, P, a, e);
        const bool hit = in && t >= a && t < e;
        pb[j] = hit ? p : 0;
        pc[j] = hit ? 1.f / (float)(e - a) : 0.f;
      }
    }
    f32x16 accS;
#pragma unroll
    for (int r = 0; r < 16; ++r) accS[r] = 0.f;
    for (int nb = 0; nb < Bn; nb += 32) {
      float wz[16], wt[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        wz[k] = wl[(nb + li) * 33 + 2 * k + lh];                  // A[i = bn][k = c] for z
        wt[k] = wl[(nb + mfma32_row(k, lh)) * 33 + li];            // A[i = c][k = bn] for dS
      }
      f32x16 accZ;
#pragma unroll
      for (int r = 0; r < 16; ++r) accZ[r] = bb ? bb[nb + mfma32_row(r, lh)] : 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) accZ = mfma32(wz[k], x[k], accZ);
      float dz[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long o = ((long)b * Bn + nb + mfma32_row(r, lh)) * P;
        float g = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) g = fmaf(dout[o + pb[j]], pc[j], g);   // pb is clamped to a valid bin, pc = 0 when not a member
        dz[r] = accZ[r] > 0.f ? g : 0.f;
        if (ok) dzt[((long)b * Bn + nb + mfma32_row(r, lh)) * Lv + t] = dz[r];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) accS = mfma32(wt[r], dz[r], accS);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (ok) dS[((long)b * 32 + mfma32_row(r, lh)) * Lv + t] = accS[r];
  }
}


// ------------------------------------------------------------------ tail backward + bottleneck weight/bias gradients in ONE pass
// tail_bwd_mfma wrote dz_t (B x Bn x Lv: 268 MB at the bench shape) only so that a GEMM and a column sum could read it
// back for dW_b = dz_t . s^T and db = sum_t dz_t: 800 MB of traffic for 8 K numbers.  Here each dz tile goes from the
// accumulator through a per-wave 32x33 LDS transpose straight into NBT resident 32x32 weight-gradient accumulators
// (dW_b[bn][c], one per 32-row bn tile), with the s tile transposed once per time tile; dz_t is never written.
// Partial sums leave through per-workgroup slabs + a fixed-order reduce, like the residual blocks' gradients.
template <int NBT>   // Bn / 32
__global__ void __launch_bounds__(256)
    tail_bwd_wgrad_mfma(const float* __restrict__ s, const float* __restrict__ wb, const float* __restrict__ bb,
                        const float* __restrict__ dout, float* __restrict__ dS, float* __restrict__ slab, int B, int Lv, int P) {
  constexpr int Bn = NBT * 32;
  constexpr int SLAB = Bn * 32 + Bn;
  extern __shared__ float dyn[];
  float* wl = dyn;                       // Bn x 33 padded weights
  float* red = dyn + Bn * 33;            // SLAB partial sums of this workgroup
  float* tiles = red + SLAB;             // 4 waves x 2 transpose tiles of 32 x 33
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int tiles_per_seq = (Lv + 31) >> 5;
  const long ntiles = (long)B * tiles_per_seq;
  const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  float* T0 = tiles + wave * 2 * 1056;   // s tile
  float* T1 = T0 + 1056;                 // dz tile
  for (int i = threadIdx.x; i < Bn * 32; i += 256) wl[(i >> 5) * 33 + (i & 31)] = wb[i];
  for (int i = threadIdx.x; i < SLAB; i += 256) red[i] = 0.f;
  __syncthreads();
  f32x16 accW[NBT];
  float bsum[NBT];
#pragma unroll
  for (int n = 0; n < NBT; ++n) {
    bsum[n] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) accW[n][r] = 0.f;
  }
  // buffer addressing (common.h): one wave per SIMD is resident, so every VALU instruction adds to the MFMA time; the flat
  // form spent ~1150 of them per tile on the 64-bit addresses of the 384 pooled-gradient gathers alone
  const int rowV = Lv * 4;
  const bool single = (Lv % P) == 0;
  for (long tile = wave0; tile < ntiles; tile += nwaves) {
    const int b = __builtin_amdgcn_readfirstlane((int)(tile / tiles_per_seq));
    const int t = (int)(tile - (long)b * tiles_per_seq) * 32 + li;
    const bool ok = t < Lv;
    const __amdgpu_buffer_rsrc_t rs = brsrc(s + (long)b * 32 * Lv, 32 * rowV);
    const __amdgpu_buffer_rsrc_t rd = brsrc(dS + (long)b * 32 * Lv, 32 * rowV);
    const __amdgpu_buffer_rsrc_t rg = brsrc(dout + (long)b * Bn * P, Bn * P * 4);
    const int offs = ok ? t * 4 + lh * rowV : BUF_OOB;            // s row 2k + lh
    const int offd = ok ? t * 4 + 4 * lh * rowV : BUF_OOB;        // dS row mfma32_row(r, lh)
    float x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = bload(rs, offs, 2 * k * rowV);   // 0 past the sequence
    __builtin_amdgcn_sched_barrier(0);     // keep the 16 loads together (see wn_block_bwd_dz_wgrad_mfma)
    int pb[3];
    float pc[3];
    {
      const int c0 = ok ? (int)(((long)t * P) / Lv) : 0;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int p = c0 - 1 + j;
        int a = 0, e = 0;
        const bool in = ok && p >= 0 && p < P;
        if (in) pool_bin(p, Lv, P, a, e);
        const bool hit = in && t >= a && t < e;
        pb[j] = hit ? p : 0;
        pc[j] = hit ? 1.f / (float)(e - a) : 0.f;
      }
    }
    int pbo[3];       // byte offset of (row 4 lh, bin pb[j]) inside this sequence's dout slab; the row of register r is scalar
#pragma unroll
    for (int j = 0; j < 3; ++j) pbo[j] = (pb[j] + 4 * lh * P) * 4;
    // s tile -> fragment fs[q] = s[c = li][t = 2q + lh]  (B operand of the weight-gradient products)
    float fs[16];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 16; ++k) T0[(2 * k + lh) * 33 + li] = x[k];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 16; ++q) fs[q] = T0[li * 33 + 2 * q + lh];
    f32x16 accS;
#pragma unroll
    for (int r = 0; r < 16; ++r) accS[r] = 0.f;
#pragma unroll
    for (int n = 0; n < NBT; ++n) {
      const int nb = n * 32;
      float wz[16], wt[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        wz[k] = wl[(nb + li) * 33 + 2 * k + lh];                  // A[i = bn][k = c] for z
        wt[k] = wl[(nb + mfma32_row(k, lh)) * 33 + li];            // A[i = c][k = bn] for dS
      }
      f32x16 accZ;
#pragma unroll
      for (int r = 0; r < 16; ++r) accZ[r] = bb ? bb[nb + mfma32_row(r, lh)] : 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) accZ = mfma32(wz[k], x[k], accZ);
      float dz[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float g = 0.f;
        if (single) {          // Lv % P == 0: the bins do not overlap, candidate 1 (bin t*P/Lv) is the only member
          g = bload(rg, pbo[1], (nb + mfma32_row(r, 0)) * P * 4) * pc[1];
        } else {
#pragma unroll
          for (int j = 0; j < 3; ++j) g = fmaf(bload(rg, pbo[j], (nb + mfma32_row(r, 0)) * P * 4), pc[j], g);   // pb clamped to a valid bin, pc = 0 when not a member
        }
        dz[r] = accZ[r] > 0.f ? g : 0.f;                                   // (0 outside the sequence: pc = 0)
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) accS = mfma32(wt[r], dz[r], accS);
      // dz tile (rows bn, column time) -> fragment fz[q] = dz[bn = li][t = 2q + lh]
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 16; ++r) T1[mfma32_row(r, lh) * 33 + li] = dz[r];
      __builtin_amdgcn_wave_barrier();
      float fz[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) { fz[q] = T1[li * 33 + 2 * q + lh]; bsum[n] += fz[q]; }
#pragma unroll
      for (int q = 0; q < 16; ++q) accW[n] = mfma32(fz[q], fs[q], accW[n]);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) bstore(accS[r], rd, offd, mfma32_row(r, 0) * rowV);
  }
  // the four waves' sums meet in LDS in WAVE ORDER (plain adds between barriers, not LDS atomics in arrival order)
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int n = 0; n < NBT; ++n) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(n * 32 + mfma32_row(r, lh)) * 32 + li] += accW[n][r];
      }
      if (lh == 0) {
#pragma unroll
        for (int n = 0; n < NBT; ++n) red[Bn * 32 + n * 32 + li] += bsum[n];
      }
      __builtin_amdgcn_wave_barrier();
      if (lh == 1) {
#pragma unroll
        for (int n = 0; n < NBT; ++n) red[Bn * 32 + n * 32 + li] += bsum[n];
      }
    }
  }
  __syncthreads();
  float* out = slab + (long)blockIdx.x * SLAB;
  for (int i = threadIdx.x; i < SLAB; i += 256) out[i] = red[i];
}

// The same pass with TWO waves per time tile (8 waves per workgroup, two per SIMD).  The kernel above keeps all Bn / 32 = 8
// weight-gradient accumulators in one wave: 416 registers, one wave per SIMD, and its chain -- load, MFMA, LDS transpose, MFMA --
// runs at a quarter of the matrix rate (310 us for 12.9 GFLOP at the bench shape).  Here waves 2p and 2p + 1 share tile after
// tile: each takes HALF of the bottleneck rows (four 32-row tiles: z, dz, its four weight-gradient accumulators) and a partial
// sum of d s over them; wave 2p + 1 hands its partial through LDS to wave 2p, which adds (fixed order) and stores.  Half the
// registers per wave, twice the waves: the chains of two waves interleave on every SIMD.
template <int NBT>   // Bn / 32 (even)
__global__ void __launch_bounds__(512, 2)
    tail_bwd_wgrad_pair(const float* __restrict__ s, const float* __restrict__ wb, const float* __restrict__ bb,
                        const float* __restrict__ dout, float* __restrict__ dS, float* __restrict__ slab, int B, int Lv, int P) {
  constexpr int Bn = NBT * 32, NH = NBT / 2;
  constexpr int SLAB = Bn * 32 + Bn;
  extern __shared__ float dyn[];
  float* wl = dyn;                       // Bn x 33 padded weights
  float* red = dyn + Bn * 33;            // SLAB partial sums of this workgroup
  float* tiles = red + SLAB;             // 8 waves x 2 transpose tiles of 32 x 33
  float* psum = tiles + 8 * 2 * 1056;    // 4 pairs x one 32 x 33 tile: the odd wave's partial d s
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int pair = wave >> 1, half = wave & 1;
  const int tiles_per_seq = (Lv + 31) >> 5;
  const long ntiles = (long)B * tiles_per_seq;
  const long npairs = (long)gridDim.x * 4;
  float* T0 = tiles + wave * 2 * 1056;   // s tile
  float* T1 = T0 + 1056;                 // dz tile
  float* PS = psum + pair * 1056;
  for (int i = threadIdx.x; i < Bn * 32; i += 512) wl[(i >> 5) * 33 + (i & 31)] = wb[i];
  for (int i = threadIdx.x; i < SLAB; i += 512) red[i] = 0.f;
  __syncthreads();
  f32x16 accW[NH];
  float bsum[NH];
#pragma unroll
  for (int n = 0; n < NH; ++n) {
    bsum[n] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) accW[n][r] = 0.f;
  }
  const int rowV = Lv * 4;
  const bool single = (Lv % P) == 0;
  // every pair of the workgroup runs the same number of rounds (the barriers below are workgroup barriers)
  const long rounds = (ntiles + npairs - 1) / npairs;
  for (long rd_ = 0; rd_ < rounds; ++rd_) {
    const long tile = rd_ * npairs + (long)blockIdx.x * 4 + pair;
    const bool live = tile < ntiles;
    const int b = __builtin_amdgcn_readfirstlane((int)((live ? tile : 0) / tiles_per_seq));
    const int t = (int)((live ? tile : 0) - (long)b * tiles_per_seq) * 32 + li;
    const bool ok = live && t < Lv;
    const __amdgpu_buffer_rsrc_t rs = brsrc(s + (long)b * 32 * Lv, 32 * rowV);
    const __amdgpu_buffer_rsrc_t rd = brsrc(dS + (long)b * 32 * Lv, 32 * rowV);
    const __amdgpu_buffer_rsrc_t rg = brsrc(dout + (long)b * Bn * P, Bn * P * 4);
    const int offs = ok ? t * 4 + lh * rowV : BUF_OOB;            // s row 2k + lh
    const int offd = ok ? t * 4 + 4 * lh * rowV : BUF_OOB;        // dS row mfma32_row(r, lh)
    float x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = bload(rs, offs, 2 * k * rowV);   // 0 past the sequence
    __builtin_amdgcn_sched_barrier(0);
    int pb[3];
    float pc[3];
    {
      const int c0 = ok ? (int)(((long)t * P) / Lv) : 0;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int p = c0 - 1 + j;
        int a = 0, e = 0;
        const bool in = ok && p >= 0 && p < P;
        if (in) pool_bin(p, Lv, P, a, e);
        const bool hit = in && t >= a && t < e;
        pb[j] = hit ? p : 0;
        pc[j] = hit ? 1.f / (float)(e - a) : 0.f;
      }
    }
    int pbo[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) pbo[j] = (pb[j] + 4 * lh * P) * 4;
    float fs[16];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 16; ++k) T0[(2 * k + lh) * 33 + li] = x[k];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 16; ++q) fs[q] = T0[li * 33 + 2 * q + lh];
    f32x16 accS;
#pragma unroll
    for (int r = 0; r < 16; ++r) accS[r] = 0.f;
#pragma unroll
    for (int n = 0; n < NH; ++n) {
      const int nb = (half * NH + n) * 32;
      f32x16 accZ;
#pragma unroll
      for (int r = 0; r < 16; ++r) accZ[r] = bb ? bb[nb + mfma32_row(r, lh)] : 0.f;
      {
        float wz[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) wz[k] = wl[(nb + li) * 33 + 2 * k + lh];
#pragma unroll
        for (int k = 0; k < 16; ++k) accZ = mfma32(wz[k], x[k], accZ);
      }
      __builtin_amdgcn_sched_barrier(0);      // (phase by phase: hoisted, the next phases' operands spill)
      float dz[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float g = 0.f;
        if (single) {
          g = bload(rg, pbo[1], (nb + mfma32_row(r, 0)) * P * 4) * pc[1];
        } else {
#pragma unroll
          for (int j = 0; j < 3; ++j) g = fmaf(bload(rg, pbo[j], (nb + mfma32_row(r, 0)) * P * 4), pc[j], g);
        }
        dz[r] = accZ[r] > 0.f ? g : 0.f;
      }
      __builtin_amdgcn_sched_barrier(0);
      {
        float wt[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) wt[k] = wl[(nb + mfma32_row(k, lh)) * 33 + li];
#pragma unroll
        for (int r = 0; r < 16; ++r) accS = mfma32(wt[r], dz[r], accS);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 16; ++r) T1[mfma32_row(r, lh) * 33 + li] = dz[r];
      __builtin_amdgcn_wave_barrier();
      float fz[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) { fz[q] = T1[li * 33 + 2 * q + lh]; bsum[n] += fz[q]; }
#pragma unroll
      for (int q = 0; q < 16; ++q) accW[n] = mfma32(fz[q], fs[q], accW[n]);
    }
    // d s = (rows of the first half) + (rows of the second half): the odd wave's partial crosses LDS, the even wave adds and stores
    if (half == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) PS[mfma32_row(r, lh) * 33 + li] = accS[r];
    }
    __syncthreads();
    if (half == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) bstore(accS[r] + PS[mfma32_row(r, lh) * 33 + li], rd, offd, mfma32_row(r, 0) * rowV);
    }
    __syncthreads();                     // (PS is free for the next round)
  }
  // the eight waves' sums meet in LDS in WAVE ORDER (plain adds between barriers)
  for (int w = 0; w < 8; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int n = 0; n < NH; ++n) {
        const int nt = half * NH + n;
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(nt * 32 + mfma32_row(r, lh)) * 32 + li] += accW[n][r];
      }
      if (lh == 0) {
#pragma unroll
        for (int n = 0; n < NH; ++n) red[Bn * 32 + (half * NH + n) * 32 + li] += bsum[n];
      }
      __builtin_amdgcn_wave_barrier();
      if (lh == 1) {
#pragma unroll
        for (int n = 0; n < NH; ++n) red[Bn * 32 + (half * NH + n) * 32 + li] += bsum[n];
      }
    }
  }
  __syncthreads();
  float* out = slab + (long)blockIdx.x * SLAB;
  for (int i = threadIdx.x; i < SLAB; i += 512) out[i] = red[i];
}

// dW_b[bn][c] += sum_blocks slab[bn*32 + c]; db[bn] += slab[Bn*32 + bn]   -- fixed order: 32 elements x 8 slab groups per
// workgroup, ascending slabs inside a group, the groups combined in LDS in order (no atomics, see wn_wgrad_reduce_all)
__global__ void __launch_bounds__(256)
    tail_wgrad_reduce(const float* __restrict__ slab, int nslab, int Bn, float* __restrict__ dW, float* __restrict__ db) {
  __shared__ float sm[8][32];
  const int n = Bn * 32 + Bn;
  const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + el;
  float sum = 0.f;
  if (i < n) {
#pragma unroll 4
    for (int b = grp; b < nslab; b += 8) sum += slab[(long)b * n + i];
  }
  sm[grp][el] = sum;
  __syncthreads();
  if (grp != 0 || i >= n) return;
#pragma unroll
  for (int k = 1; k < 8; ++k) sum += sm[k][el];
  if (i < Bn * 32) { if (dW) dW[i] += sum; }
  else if (db) db[i - Bn * 32] += sum;
}

constexpr int WG_SLAB = 3 * 1024 + 64;   // per-workgroup partial sums of wn_block_wgrad_mfma
constexpr int WG_MAXBLK = 512;

// ------------------------------------------------------------------ fused MFMA weight/bias gradients of a residual block (R = D = 32, fw = 2)
//   dW_dil[d][c][k] += sum_t dz[d][t] relu(s_in[c][t + k*dil])      db_dil[d]   += sum_t dz[d][t]
//   dW_dense[r][d]  += sum_t dS[r][t] relu(z[d][t])                 db_dense[r] += sum_t dS[r][t]
// The contraction index is TIME, so the MFMA wants channel-on-lane fragments (A[i = channel][k = t]).  Tiles are
// loaded the coalesced way (lane = time), transposed through a per-wave 32x33 LDS tile (conflict-free both ways)
// and fed to three 32x32 accumulators that live in registers across the wave's whole time range; one LDS +
// global float-atomic reduction per workgroup at the end.  Replaces two split-K GEMM launches and two bias
// reductions per layer.
__global__ void __launch_bounds__(256)
    wn_block_wgrad_mfma(const float* __restrict__ dS, const float* __restrict__ Z, const float* __restrict__ DZ,
                        const float* __restrict__ s_in, float* __restrict__ slab, int B, int Lin, int dil) {
  __shared__ float tile[4][32 * 33];
  __shared__ float red[3 * 1024 + 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int Lo = Lin - dil;
  const int tiles_per_seq = (Lo + 31) >> 5;
  const long ntiles = (long)B * tiles_per_seq;
  const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  float* T = tile[wave];
  for (int i = threadIdx.x; i < 3 * 1024 + 64; i += 256) red[i] = 0.f;

  f32x16 acc0, acc1, acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; }
  float bs_dz = 0.f, bs_ds = 0.f;

  // fragment of X: f[s] = X[channel = li][t0 + 2s + lh]; X is read as X[c][t0 + li] (c = 2s' + lh), coalesced,
  // unconditionally from a clamped address (masked afterwards: no per-load branches / waits)
  auto gl = [&](const float* base, long rstride, float (&v)[16]) {
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = base[(long)(2 * q + lh) * rstride];
  };
  auto xpose = [&](const float (&v)[16], bool ok, bool relu, float (&f)[16]) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float x = ok ? v[q] : 0.f;
      T[(2 * q + lh) * 33 + li] = relu ? relu1(x) : x;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 16; ++q) f[q] = T[li * 33 + 2 * q + lh];
    __builtin_amdgcn_wave_barrier();
  };

  for (long tl = wave0; tl < ntiles; tl += nwaves) {
    const int b = (int)(tl / tiles_per_seq);
    const int t = (int)(tl - (long)b * tiles_per_seq) * 32 + li;
    const bool ok = t < Lo;
    const long oo = (long)b * 32 * Lo + (ok ? t : 0);
    const long oi = (long)b * 32 * Lin + (ok ? t : 0);
    float v0[16], v1[16], v2[16], v3[16], v4[16];   // all five operand tiles in flight at once
    gl(DZ + oo, Lo, v0);
    gl(s_in + oi, Lin, v1);
    gl(s_in + oi + dil, Lin, v2);
    gl(dS + oo, Lo, v3);
    gl(Z + oo, Lo, v4);
    float fa[16], fb[16];
    xpose(v0, ok, false, fa);
#pragma unroll
    for (int q = 0; q < 16; ++q) bs_dz += fa[q];
    xpose(v1, ok, true, fb);
#pragma unroll
    for (int q = 0; q < 16; ++q) acc0 = mfma32(fa[q], fb[q], acc0);
    xpose(v2, ok, true, fb);
#pragma unroll
    for (int q = 0; q < 16; ++q) acc1 = mfma32(fa[q], fb[q], acc1);
    xpose(v3, ok, false, fa);
#pragma unroll
    for (int q = 0; q < 16; ++q) bs_ds += fa[q];
    xpose(v4, ok, true, fb);
#pragma unroll
    for (int q = 0; q < 16; ++q) acc2 = mfma32(fa[q], fb[q], acc2);
  }

  __syncthreads();  // red[] zeroed
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = mfma32_row(r, lh);
    atomicAdd(&red[row * 32 + li], acc0[r]);
    atomicAdd(&red[1024 + row * 32 + li], acc1[r]);
    atomicAdd(&red[2048 + row * 32 + li], acc2[r]);
  }
  atomicAdd(&red[3072 + li], bs_dz);
  atomicAdd(&red[3104 + li], bs_ds);
  __syncthreads();
  // one slab of 3136 partial sums per workgroup (plain stores); wn_wgrad_reduce adds the slabs in a fixed order.
  // (640 workgroups doing float atomics onto the SAME 3136 addresses ran at the contended-atomic rate: 90 us.)
  float* out = slab + (long)blockIdx.x * WG_SLAB;
  for (int i = threadIdx.x; i < WG_SLAB; i += 256) out[i] = red[i];
}

// ------------------------------------------------------------------ (A) + weight gradients in ONE pass, z recomputed
// wn_block_bwd_dz_mfma and wn_block_wgrad_mfma fused, and the forward no longer stores z: per 32-sample tile a wave loads
// dS (16 values) and the two dilation taps of s_in (lane half h = tap h, 32 channels: the forward's own operand), rebuilds
// z = b + W_dil (*) relu(s_in) with the forward's exact MFMA sequence (bit-identical, so the ReLU masks agree), forms
// dz = (z>0) * W_dense^T dS, stores it, and pushes dz / relu(s taps) / dS / relu(z) through per-wave 32x33 LDS
// transposes into the three resident weight-gradient accumulators.  Per layer this replaces 350 MB of HBM traffic
// (z written by the forward, dS + Z read twice, DZ re-read) by 150 MB; the encoder shares HBM with the trunk's
// convolutions running on the other stream, so the bytes matter beyond this kernel's own time.
__global__ void __launch_bounds__(256)
    wn_block_bwd_dz_wgrad_mfma(const float* __restrict__ dS, const float* __restrict__ w_dil, const float* __restrict__ b_dil,
                               const float* __restrict__ w_dense, const float* __restrict__ s_in, float* __restrict__ DZ,
                               float* __restrict__ slab, int B, int Lin, int dil) {
  __shared__ float tile[4][2][32 * 33];
  __shared__ float red[3 * 1024 + 64];
  __shared__ float wl[32 * 65 + 32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int Lo = Lin - dil;
  const int tiles_per_seq = (Lo + 31) >> 5;
  const int ntiles = B * tiles_per_seq;
  float* T = tile[wave][0];
  float* Th = tile[wave][lh];           // this lane half's tile for the two-tap transpose
  for (int i = threadIdx.x; i < 2048; i += 256) wl[(i >> 6) * 65 + (i & 63)] = w_dil[i];
  if (threadIdx.x < 32) wl[2080 + threadIdx.x] = b_dil ? b_dil[threadIdx.x] : 0.f;
  // W_dense through this wave's transpose tile: A[i = d = li][k = r = 2s+lh] = W_dense[r][d]
  {
    float wv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) wv[u] = w_dense[lane + 64 * u];
#pragma unroll
    for (int u = 0; u < 16; ++u) { const int i = lane + 64 * u; T[(i >> 5) * 33 + (i & 31)] = wv[u]; }
  }
  for (int i = threadIdx.x; i < 3 * 1024 + 64; i += 256) red[i] = 0.f;
  __syncthreads();
  float wt[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) wt[s] = T[(2 * s + lh) * 33 + li];
  float wd[32];   // W_dil[d = li][c = s][tap = lh]  (the forward's fragment)
#pragma unroll
  for (int s = 0; s < 32; ++s) wd[s] = wl[li * 65 + s * 2 + lh];
  const float* bzl = wl + 2080 + 4 * lh;
  __builtin_amdgcn_wave_barrier();

  f32x16 acc0, acc1, acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; }
  float bs_dz = 0.f, bs_ds = 0.f;
  const int rowL = Lin * 4, rowO = Lo * 4;

  // One wave per SIMD is resident (368 registers), so VALU instructions add to the 96 MFMAs of a tile one for one (~4
  // cycles each): the flat-addressed version of this kernel carried 1225 of them per tile -- 64-bit address arithmetic
  // for 64 memory instructions, a validity select on every loaded value, two-instruction ReLUs.  Buffer addressing
  // (common.h): samples past the end of a sequence carry an out-of-range offset, their loads return 0 and their stores are
  // dropped; a zero dS column makes every product of that column zero, so nothing downstream needs masking.
  // Software pipeline across tiles: the NEXT tile's 48 loads are issued before this tile's 96 MFMAs and LDS transposes
  // (nothing else hides their latency), alternating between two register sets instead of copying.
  auto issue = [&](int tile, float (&gn)[16], float (&xn)[32]) {
    const int tc = tile < ntiles ? tile : ntiles - 1;
    const int b = tc / tiles_per_seq;
    const int t = (tc - b * tiles_per_seq) * 32 + li;
    const bool ok = t < Lo;
    const __amdgpu_buffer_rsrc_t rg = brsrc(dS + (long)b * 32 * Lo, 32 * rowO);
    const __amdgpu_buffer_rsrc_t rx = brsrc(s_in + (long)b * 32 * Lin, 32 * rowL);
    const int offg = ok ? t * 4 + lh * rowO : BUF_OOB;
    const int offx = ok ? (t + lh * dil) * 4 : BUF_OOB;
#pragma unroll
    for (int q = 0; q < 16; ++q) gn[q] = bload(rg, offg, 2 * q * rowO);
#pragma unroll
    for (int c = 0; c < 32; ++c) xn[c] = bload(rx, offx, c * rowL);
  };
  auto compute = [&](int tidx, const float (&g)[16], const float (&xraw)[32]) {
    const int b = tidx / tiles_per_seq;
    const int t = (tidx - b * tiles_per_seq) * 32 + li;
    const int offo = t < Lo ? t * 4 + 4 * lh * rowO : BUF_OOB;
    const __amdgpu_buffer_rsrc_t rz = brsrc(DZ + (long)b * 32 * Lo, 32 * rowO);
    float x[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) x[c] = relu1(xraw[c]);      // relu(s tap lh); 0 outside the sequence
    // z exactly as the forward builds it
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = bzl[mfma32_row(r, 0)];
#pragma unroll
    for (int s = 0; s < 32; ++s) z = mfma32(wd[s], x[s], z);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc = mfma32(wt[q], g[q], acc);
    float f0[16], f1[16];
    // ---- dz (D layout: rows mfma32_row(r, lh), column = time li) -> store, transpose
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float dz = z[r] > 0.f ? acc[r] : 0.f;            // (a column past the sequence has dS = 0, so acc = 0)
      bstore(dz, rz, offo, mfma32_row(r, 0) * rowO);
      T[mfma32_row(r, lh) * 33 + li] = dz;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 16; ++q) { f0[q] = T[li * 33 + 2 * q + lh]; bs_dz += f0[q]; }   // f0 = dz[d = li][t = 2q+lh]
    __builtin_amdgcn_wave_barrier();
    // ---- both taps of relu(s): lane half h writes its 32 channels into tile h
#pragma unroll
    for (int c = 0; c < 32; ++c) Th[c * 33 + li] = x[c];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 16; ++q) f1[q] = tile[wave][0][li * 33 + 2 * q + lh];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc0 = mfma32(f0[q], f1[q], acc0);
#pragma unroll
    for (int q = 0; q < 16; ++q) f1[q] = tile[wave][1][li * 33 + 2 * q + lh];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc1 = mfma32(f0[q], f1[q], acc1);
    __builtin_amdgcn_wave_barrier();
    // ---- dS (channel 2q+lh on register q) and relu(z) (D layout; multiplied by dS = 0 past the sequence)
#pragma unroll
    for (int q = 0; q < 16; ++q) T[(2 * q + lh) * 33 + li] = g[q];
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[wave][1][mfma32_row(r, lh) * 33 + li] = relu1(z[r]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 16; ++q) { f0[q] = T[li * 33 + 2 * q + lh]; bs_ds += f0[q]; }
#pragma unroll
    for (int q = 0; q < 16; ++q) f1[q] = tile[wave][1][li * 33 + 2 * q + lh];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc2 = mfma32(f0[q], f1[q], acc2);
    __builtin_amdgcn_wave_barrier();
  };
  const TileWalk tw = xcd_walk(ntiles);
  const int first = __builtin_amdgcn_readfirstlane((int)tw.first), last = (int)tw.last, stride = (int)tw.stride;
  float ga[16], xa[32], gb[16], xb[32];
  if (first < last) issue(first, ga, xa);
  for (int tl = first; tl < last; tl += 2 * stride) {
    issue(tl + stride, gb, xb);
    // pin the 48 loads HERE: left alone hipcc sinks each one down to its consumer (load, wait, mfma, load, wait, ...),
    // 32 serial memory round trips per tile -- the kernel ran 2.5x slower than the two it replaces
    __builtin_amdgcn_sched_barrier(0);
    compute(tl, ga, xa);
    if (tl + stride >= last) break;
    issue(tl + 2 * stride, ga, xa);
    __builtin_amdgcn_sched_barrier(0);
    compute(tl + stride, gb, xb);
  }

  // the four waves' sums meet in LDS in WAVE ORDER (plain adds between barriers: float atomics here added them in
  // arrival order, the one place left where this kernel's result could differ in the last bit from run to run)
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mfma32_row(r, lh);
        red[row * 32 + li] += acc0[r];
        red[1024 + row * 32 + li] += acc1[r];
        red[2048 + row * 32 + li] += acc2[r];
      }
      // the bias sums: both lane halves hold partial sums of the same 32 channels -> half 0 first, then half 1
      if (lh == 0) { red[3072 + li] += bs_dz; red[3104 + li] += bs_ds; }
      __builtin_amdgcn_wave_barrier();
      if (lh == 1) { red[3072 + li] += bs_dz; red[3104 + li] += bs_ds; }
    }
  }
  __syncthreads();
  float* out = slab + (long)blockIdx.x * WG_SLAB;
  for (int i = threadIdx.x; i < WG_SLAB; i += 256) out[i] = red[i];
}

// HIGH-OCCUPANCY form of the fused pass (option wn_bwd_t = 2): weights read from LDS per MFMA, no cross-tile prefetch -> under
// 256 registers, two waves per SIMD (two workgroups per CU).  (Tried for the beside-the-trunk case: this body WITH the cross-tile
// prefetch, 243 registers and no AGPR shuffling, one workgroup per CU -- 0.2-0.5 ms/step slower than the resident-weights form.)
__global__ void __launch_bounds__(256, 2)
    wn_block_bwd_dz_wgrad_occ(const float* __restrict__ dS, const float* __restrict__ w_dil, const float* __restrict__ b_dil,
                               const float* __restrict__ w_dense, const float* __restrict__ s_in, float* __restrict__ DZ,
                               float* __restrict__ slab, int B, int Lin, int dil) {
  __shared__ float tile[4][2][32 * 33];
  __shared__ float red[3 * 1024 + 64];
  __shared__ float wl[32 * 65 + 32];
  __shared__ float wtl[32 * 33];        // W_dense[r][d], padded rows: A[i = d = li][k = r = 2s+lh] = wtl[(2s+lh)*33 + li]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int Lo = Lin - dil;
  const int tiles_per_seq = (Lo + 31) >> 5;
  const int ntiles = B * tiles_per_seq;
  float* T = tile[wave][0];
  float* Th = tile[wave][lh];           // this lane half's tile for the two-tap transpose
  for (int i = threadIdx.x; i < 2048; i += 256) wl[(i >> 6) * 65 + (i & 63)] = w_dil[i];
  if (threadIdx.x < 32) wl[2080 + threadIdx.x] = b_dil ? b_dil[threadIdx.x] : 0.f;
  for (int i = threadIdx.x; i < 1024; i += 256) wtl[(i >> 5) * 33 + (i & 31)] = w_dense[i];
  for (int i = threadIdx.x; i < 3 * 1024 + 64; i += 256) red[i] = 0.f;
  __syncthreads();
  const float* wtp = wtl + lh * 33 + li;       // + 66 s
  const float* wdp = wl + li * 65 + lh;        // + 2 s: W_dil[d = li][c = s][tap = lh]  (the forward's fragment)
  const float* bzl = wl + 2080 + 4 * lh;
  __builtin_amdgcn_wave_barrier();

  f32x16 acc0, acc1, acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; }
  float bs_dz = 0.f, bs_ds = 0.f;
  const int rowL = Lin * 4, rowO = Lo * 4;

  // One wave per SIMD is resident (368 registers), so VALU instructions add to the 96 MFMAs of a tile one for one (~4
  // cycles each): the flat-addressed version of this kernel carried 1225 of them per tile -- 64-bit address arithmetic
  // for 64 memory instructions, a validity select on every loaded value, two-instruction ReLUs.  Buffer addressing
  // (common.h): samples past the end of a sequence carry an out-of-range offset, their loads return 0 and their stores are
  // dropped; a zero dS column makes every product of that column zero, so nothing downstream needs masking.
  // Software pipeline across tiles: the NEXT tile's 48 loads are issued before this tile's 96 MFMAs and LDS transposes
  // (nothing else hides their latency), alternating between two register sets instead of copying.
  auto issue = [&](int tile, float (&gn)[16], float (&xn)[32]) {
    const int tc = tile < ntiles ? tile : ntiles - 1;
    const int b = tc / tiles_per_seq;
    const int t = (tc - b * tiles_per_seq) * 32 + li;
    const bool ok = t < Lo;
    const __amdgpu_buffer_rsrc_t rg = brsrc(dS + (long)b * 32 * Lo, 32 * rowO);
    const __amdgpu_buffer_rsrc_t rx = brsrc(s_in + (long)b * 32 * Lin, 32 * rowL);
    const int offg = ok ? t * 4 + lh * rowO : BUF_OOB;
    const int offx = ok ? (t + lh * dil) * 4 : BUF_OOB;
#pragma unroll
    for (int q = 0; q < 16; ++q) gn[q] = bload(rg, offg, 2 * q * rowO);
#pragma unroll
    for (int c = 0; c < 32; ++c) xn[c] = bload(rx, offx, c * rowL);
  };
  auto compute = [&](int tidx, const float (&g)[16], const float (&xraw)[32]) {
    const int b = tidx / tiles_per_seq;
    const int t = (tidx - b * tiles_per_seq) * 32 + li;
    const int offo = t < Lo ? t * 4 + 4 * lh * rowO : BUF_OOB;
    const __amdgpu_buffer_rsrc_t rz = brsrc(DZ + (long)b * 32 * Lo, 32 * rowO);
    float x[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) x[c] = relu1(xraw[c]);      // relu(s tap lh); 0 outside the sequence
    // z exactly as the forward builds it
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = bzl[mfma32_row(r, 0)];
#pragma unroll
    for (int s = 0; s < 32; ++s) z = mfma32(wdp[2 * s], x[s], z);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc = mfma32(wtp[66 * q], g[q], acc);
    float f0[16], f1[16];
    // ---- dz (D layout: rows mfma32_row(r, lh), column = time li) -> store, transpose
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float dz = z[r] > 0.f ? acc[r] : 0.f;            // (a column past the sequence has dS = 0, so acc = 0)
      bstore(dz, rz, offo, mfma32_row(r, 0) * rowO);
      T[mfma32_row(r, lh) * 33 + li] = dz;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 16; ++q) { f0[q] = T[li * 33 + 2 * q + lh]; bs_dz += f0[q]; }   // f0 = dz[d = li][t = 2q+lh]
    __builtin_amdgcn_wave_barrier();
    // ---- both taps of relu(s): lane half h writes its 32 channels into tile h
#pragma unroll
    for (int c = 0; c < 32; ++c) Th[c * 33 + li] = x[c];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 16; ++q) f1[q] = tile[wave][0][li * 33 + 2 * q + lh];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc0 = mfma32(f0[q], f1[q], acc0);
#pragma unroll
    for (int q = 0; q < 16; ++q) f1[q] = tile[wave][1][li * 33 + 2 * q + lh];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc1 = mfma32(f0[q], f1[q], acc1);
    __builtin_amdgcn_wave_barrier();
    // ---- dS (channel 2q+lh on register q) and relu(z) (D layout; multiplied by dS = 0 past the sequence)
#pragma unroll
    for (int q = 0; q < 16; ++q) T[(2 * q + lh) * 33 + li] = g[q];
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[wave][1][mfma32_row(r, lh) * 33 + li] = relu1(z[r]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 16; ++q) { f0[q] = T[li * 33 + 2 * q + lh]; bs_ds += f0[q]; }
#pragma unroll
    for (int q = 0; q < 16; ++q) f1[q] = tile[wave][1][li * 33 + 2 * q + lh];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc2 = mfma32(f0[q], f1[q], acc2);
    __builtin_amdgcn_wave_barrier();
  };
  const TileWalk tw = xcd_walk(ntiles);
  const int first = __builtin_amdgcn_readfirstlane((int)tw.first), last = (int)tw.last, stride = (int)tw.stride;
  for (int tl = first; tl < last; tl += stride) {       // nothing held across tiles: the other wave of the SIMD hides the loads
    float ga[16], xa[32];
    issue(tl, ga, xa);
    __builtin_amdgcn_sched_barrier(0);
    compute(tl, ga, xa);
  }

  // the four waves' sums meet in LDS in WAVE ORDER (plain adds between barriers: float atomics here added them in
  // arrival order, the one place left where this kernel's result could differ in the last bit from run to run)
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mfma32_row(r, lh);
        red[row * 32 + li] += acc0[r];
        red[1024 + row * 32 + li] += acc1[r];
        red[2048 + row * 32 + li] += acc2[r];
      }
      // the bias sums: both lane halves hold partial sums of the same 32 channels -> half 0 first, then half 1
      if (lh == 0) { red[3072 + li] += bs_dz; red[3104 + li] += bs_ds; }
      __builtin_amdgcn_wave_barrier();
      if (lh == 1) { red[3072 + li] += bs_dz; red[3104 + li] += bs_ds; }
    }
  }
  __syncthreads();
  float* out = slab + (long)blockIdx.x * WG_SLAB;
  for (int i = threadIdx.x; i < WG_SLAB; i += 256) out[i] = red[i];
}

// ------------------------------------------------------------------ the same pass with TRANSPOSED products: no LDS transposes
// wn_block_bwd_dz_wgrad_mfma pushes five 32x32 tiles per time tile through LDS to turn "time on the lane" (how the data
// lies in memory) into "channel on the lane" (what a contraction over TIME wants), holds 368 registers and runs one wave
// per SIMD.  Here z and dz are formed TRANSPOSED instead -- the activation tile is the A operand (lane = time, loaded
// coalesced exactly as before), the weights are the B operand:
//     zT[t][d]  = b_dil[d] + sum_{c,k} relu(s[c][t + k*dil]) * W_dil[d][c][k]      (bit-identical to the forward's z: same
//     dzT[t][d] = (z > 0) * sum_r dS[r][t] * W_dense[r][d]                          products, same accumulation order)
// so the accumulators come out with the CHANNEL on the lane and time in the registers (register r of lane-half h is sample
// t0 + mfma32_row(r, h)) -- which is precisely the A / B fragment of the weight-gradient products, whose k-steps may
// enumerate the 32 samples of the tile in any order:
//     dW_dil[d][c][k] += sum_t dzT[t][d] * relu(s[c][t + k*dil])      A = dzT registers,  B = s, channel on the lane
//     dW_dense[r][d]  += sum_t dS[r][t]  * relu(zT[t][d])             A = dS, channel on the lane,  B = relu(zT) registers
// The three channel-on-lane operands (dS, both taps of s) are a SECOND load of lines the time-on-lane loads fetch anyway:
// lane c reads its own row as four 16-byte pieces (samples 8q + 4h .. +3), L1/L2 hits.  dz leaves as four 16-byte stores
// per lane.  No LDS traffic inside the loop, under 256 registers: two waves per SIMD cover each other's memory latency.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // rows start at any sample: 4-byte aligned 16-byte accesses

__global__ void __launch_bounds__(256, 2)
    wn_block_bwd_dzw_t(const float* __restrict__ dS, const float* __restrict__ w_dil, const float* __restrict__ b_dil,
                       const float* __restrict__ w_dense, const float* __restrict__ s_in, float* __restrict__ DZ,
                       float* __restrict__ slab, int B, int Lin, int dil) {
  __shared__ float red[3 * 1024 + 64];
  __shared__ float wl[32 * 65 + 32 * 33 + 32];
  const int lane = threadIdx.x & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int Lo = Lin - dil;
  const int tiles_per_seq = (Lo + 31) >> 5;
  const long ntiles = (long)B * tiles_per_seq;
  const long wave0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  for (int i = threadIdx.x; i < 2048; i += 256) wl[(i >> 6) * 65 + (i & 63)] = w_dil[i];
  for (int i = threadIdx.x; i < 1024; i += 256) wl[2080 + (i >> 5) * 33 + (i & 31)] = w_dense[i];
  if (threadIdx.x < 32) wl[3136 + threadIdx.x] = b_dil ? b_dil[threadIdx.x] : 0.f;
  for (int i = threadIdx.x; i < 3 * 1024 + 64; i += 256) red[i] = 0.f;
  __syncthreads();
  // The weight fragments stay in LDS and are read per k-step (one conflict-free ds_read_b32 each, 48 per tile against 96
  // MFMAs of 64 cycles): 48 registers less than keeping them resident -- that is what brings the kernel under the 256
  // registers of two waves per SIMD.
  //   wdp[2s]  : B[k = (c = s, tap = lh)][j = d = li] = W_dil[d][c][tap]   (the forward's A fragment)
  //   wtp[66s] : B[k = r = 2s+lh][j = d = li]         = W_dense[r][d]
  const float* wdp = wl + li * 65 + lh;
  const float* wtp = wl + 2080 + lh * 33 + li;
  const float bz = wl[3136 + li];

  f32x16 acc0, acc1, acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; }
  float bs_dz = 0.f, bs_ds = 0.f;

  for (long tl = wave0; tl < ntiles; tl += nwaves) {
    const int b = (int)(tl / tiles_per_seq);
    const int t0 = (int)(tl - (long)b * tiles_per_seq) * 32;
    const int t = t0 + li;
    const bool ok = t < Lo;
    const int tcl = ok ? t : 0;
    const bool full = t0 + 32 <= Lo;                        // wave-uniform: every sample of the tile exists
    // ---- time on the lane (coalesced; unconditional loads from a clamped address, masked below)
    float x[32], g[16];
    {
      const float* xp = s_in + (long)b * 32 * Lin + tcl + lh * dil;
#pragma unroll
      for (int c = 0; c < 32; ++c) x[c] = xp[(long)c * Lin];
      const float* gp = dS + (long)b * 32 * Lo + tcl;
#pragma unroll
      for (int q = 0; q < 16; ++q) g[q] = gp[(long)(2 * q + lh) * Lo];
    }
    // ---- channel on the lane: lane li reads its own row, samples t0 + mfma32_row(s, lh) = t0 + 8q + 4lh + e (s = 4q + e)
    float gc[16], xc0[16], xc1[16];
    const float* rowS = dS + ((long)b * 32 + li) * Lo + t0 + 4 * lh;
    const float* rowX = s_in + ((long)b * 32 + li) * Lin + t0 + 4 * lh;
    if (full) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f4u a = *reinterpret_cast<const f4u*>(rowS + 8 * q);
        const f4u u = *reinterpret_cast<const f4u*>(rowX + 8 * q);
        const f4u v = *reinterpret_cast<const f4u*>(rowX + dil + 8 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) { gc[4 * q + e] = a[e]; xc0[4 * q + e] = u[e]; xc1[4 * q + e] = v[e]; }
      }
    } else {                                                // ragged last tile of a sequence: element-wise, clamped + masked
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int o = 8 * (s >> 2) + (s & 3);
        const bool in = t0 + 4 * lh + o < Lo;
        const int oc = in ? o : -(t0 + 4 * lh);             // clamp to the row's first sample
        const float a = rowS[oc], u = rowX[oc], v = rowX[oc + dil];
        gc[s] = in ? a : 0.f; xc0[s] = in ? u : 0.f; xc1[s] = in ? v : 0.f;
      }
    }
    __builtin_amdgcn_sched_barrier(0);                      // keep every load of the tile above the MFMAs
#pragma unroll
    for (int c = 0; c < 32; ++c) x[c] = ok ? relu1(x[c]) : 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) g[q] = ok ? g[q] : 0.f;
    // ---- zT, dzT
    f32x16 zT;
#pragma unroll
    for (int r = 0; r < 16; ++r) zT[r] = bz;
#pragma unroll
    for (int s = 0; s < 32; ++s) zT = mfma32(x[s], wdp[2 * s], zT);
    f32x16 dzT;
#pragma unroll
    for (int r = 0; r < 16; ++r) dzT[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) dzT = mfma32(g[s], wtp[66 * s], dzT);
#pragma unroll
    for (int r = 0; r < 16; ++r) dzT[r] = zT[r] > 0.f ? dzT[r] : 0.f;     // rows of samples beyond the sequence are zero (g = 0)
    // ---- dz out: lane li owns row d = li
    float* rowZ = DZ + ((long)b * 32 + li) * Lo + t0 + 4 * lh;
    if (full) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f4u o = {dzT[4 * q], dzT[4 * q + 1], dzT[4 * q + 2], dzT[4 * q + 3]};
        *reinterpret_cast<f4u*>(rowZ + 8 * q) = o;
      }
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int o = 8 * (s >> 2) + (s & 3);
        if (t0 + 4 * lh + o < Lo) rowZ[o] = dzT[s];
      }
    }
    // ---- parameter gradients: k-step s stands for sample t0 + mfma32_row(s, lh) on both operands
#pragma unroll
    for (int s = 0; s < 16; ++s) { bs_dz += dzT[s]; bs_ds += gc[s]; }
#pragma unroll
    for (int s = 0; s < 16; ++s) acc2 = mfma32(gc[s], relu1(zT[s]), acc2);
#pragma unroll
    for (int s = 0; s < 16; ++s) acc0 = mfma32(dzT[s], relu1(xc0[s]), acc0);
#pragma unroll
    for (int s = 0; s < 16; ++s) acc1 = mfma32(dzT[s], relu1(xc1[s]), acc1);
  }

#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = mfma32_row(r, lh);
    atomicAdd(&red[row * 32 + li], acc0[r]);
    atomicAdd(&red[1024 + row * 32 + li], acc1[r]);
    atomicAdd(&red[2048 + row * 32 + li], acc2[r]);
  }
  atomicAdd(&red[3072 + li], bs_dz);
  atomicAdd(&red[3104 + li], bs_ds);
  __syncthreads();
  float* out = slab + (long)blockIdx.x * WG_SLAB;
  for (int i = threadIdx.x; i < WG_SLAB; i += 256) out[i] = red[i];
}

// dW_dil[d][c][k] += sum_blocks slab[k*1024 + d*32 + c]; dW_dense[r][d] += slab[2048 + r*32 + d]; biases likewise.
// ALL residual blocks in one launch behind the last of them (blockIdx.y = layer; 20 launches of ~5 us before), and in a
// FIXED order: 32 elements x 8 slab groups per workgroup, each thread adds its group's slabs in ascending order, the eight
// partial sums meet in LDS in group order, one thread adds the total onto the gradient.  (The per-layer version spread
// the slabs over gridDim.y and added with float atomics: the last place where the encoder's weight gradients could
// differ in the last bit from run to run.)
constexpr int WN_MAXL = 64;       // residual blocks one reduce launch covers
struct WgradTab {
  const float* slab[WN_MAXL];
  int nslab[WN_MAXL];
  float* dW_dil[WN_MAXL];
  float* db_dil[WN_MAXL];
  float* dW_dense[WN_MAXL];
  float* db_dense[WN_MAXL];
};
__global__ void __launch_bounds__(256) wn_wgrad_reduce_all(const WgradTab tab) {
  __shared__ float sm[8][32];
  const int l = blockIdx.y;
  const int el = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + el;                     // WG_SLAB = 98 * 32
  const int nslab = tab.nslab[l];
  float s = 0.f;
  if (nslab > 0) {
    const float* sl = tab.slab[l] + i;
#pragma unroll 4
    for (int b = grp; b < nslab; b += 8) s += sl[(long)b * WG_SLAB];
  }
  sm[grp][el] = s;
  __syncthreads();
  if (grp != 0 || nslab <= 0) return;
#pragma unroll
  for (int k = 1; k < 8; ++k) s += sm[k][el];
  if (i < 2048) { if (tab.dW_dil[l]) tab.dW_dil[l][(i & 1023) * 2 + (i >> 10)] += s; }   // every pointer may be NULL (frozen)
  else if (i < 3072) { if (tab.dW_dense[l]) tab.dW_dense[l][i - 2048] += s; }
  else if (i < 3104) { if (tab.db_dil[l]) tab.db_dil[l][i - 3072] += s; }
  else if (tab.db_dense[l]) tab.db_dense[l][i - 3104] += s;
}

// ------------------------------------------------------------------ plan / workspace
struct Plan {
  int n;
  int L[66];        // L[0] = causal output length, L[i+1] = after block i
  size_t s[66];     // offsets of s_i (floats); without save_for_backward only 2 ping-pong buffers
  size_t z, dz, ga, gb, dzt;
  size_t zs[66];    // per-layer pre-ReLU dilation outputs kept for backward (MFMA shape); else 0
  size_t slab;      // weight-gradient partial slabs
  size_t gslab;     // GEMM engine scratch (weight gradients of shapes / switches that go through igemm)
  size_t total;
};
static int make_plan(const avvad_wavenet_desc* d, Plan* p) {
  if (!d || d->n_layers < 0 || d->n_layers > 64 || d->B <= 0 || d->fw < 1 || d->qc < 1 || d->R < 1 || d->D < 1 ||
      d->Bn < 1 || d->P < 1 || (d->n_layers > 0 && !d->dilations_h))
    return AVVAD_EINVAL;
  p->n = d->n_layers;
  p->L[0] = d->L - (d->fw - 1);
  for (int i = 0; i < p->n; ++i) p->L[i + 1] = p->L[i] - d->dilations_h[i] * (d->fw - 1);
  if (p->L[p->n] < 1) return AVVAD_EINVAL;
  size_t off = 0;
  auto take = [&](size_t n) { size_t o = off; off += align_up(n, 64); return o; };
  const size_t B = d->B;
  if (d->save_for_backward) {
    for (int i = 0; i <= p->n; ++i) p->s[i] = take(B * d->R * p->L[i]);
    const size_t big = B * (size_t)(d->R > d->D ? d->R : d->D) * p->L[0];
    p->z = take(big); p->dz = take(big); p->ga = take(big); p->gb = take(big);
    p->dzt = take(B * d->Bn * p->L[p->n]);
    {   // one scratch z buffer (only the unfused backward fallback rebuilds z into it)
      const size_t zb = (d->R == 32 && d->D == 32 && d->fw == 2 && p->n > 0) ? take(B * d->D * p->L[1]) : 0;
      for (int i = 0; i < p->n; ++i) p->zs[i] = zb;
    }
    {   // weight-gradient slabs: residual blocks (WG_MAXBLK x WG_SLAB) or the tail (256 workgroups x (Bn*32 + Bn))
      const bool perlayer = d->R == 32 && d->D == 32 && d->fw == 2;      // one slab set per residual block: ONE reduce launch
      const size_t a = (size_t)WG_MAXBLK * WG_SLAB * (perlayer ? (size_t)p->n : 1), b2 = (size_t)256 * ((size_t)d->Bn * 33);
      p->slab = take(a > b2 ? a : b2);
    }
    p->gslab = take(igemm::SLAB_FLOATS);
  } else {
    for (int i = 0; i < p->n; ++i) p->zs[i] = 0;
    const size_t a = take(B * d->R * p->L[0]), b2 = take(B * d->R * p->L[0]);
    for (int i = 0; i <= p->n; ++i) p->s[i] = (i & 1) ? b2 : a;
    p->z = take(B * d->D * p->L[0]);  // scratch of the generic (non-MFMA-shape) block path
    p->dz = p->ga = p->gb = p->dzt = 0;
  }
  p->total = off;
  return AVVAD_OK;
}

static inline bool mfma_shape(const avvad_wavenet_desc* d) { return d->R == 32 && d->D == 32 && d->fw == 2; }
// forward of one residual block, MFMA shape: wide kernel when the plane has >= 128 output samples, else the dword forms
static void launch_block_fwd(const float* s_in, const float* wd, const float* bd, const float* we, const float* be, float* s_out,
                             int B, int Lin, int dil, hipStream_t s);
// the buffer-addressed block kernels index a sequence's [32][L] slab with 31-bit byte offsets and count tiles in an int
static inline bool buf_ok(int B, int Lin) {
  return avvad_tune().wn_flat != 1 && (long)Lin * 32 * 4 + 4096 < (1L << 31) && (long)B * cdiv(Lin, 32) < (1L << 31);
}
// option "wn_flat": 0 by plane length (wide dwordx4 kernel from 8192 samples, else the high-occupancy dword kernel),
// 1 flat dword kernel, 2 dword buffer kernel with resident weights and cross-tile prefetch, 3 wide kernel, 4 high-occupancy kernel
constexpr int WIDE_FROM = 8192;    // plane length from which the dwordx4 kernel runs
static void launch_block_fwd(const float* s_in, const float* wd, const float* bd, const float* we, const float* be, float* s_out,
                             int B, int Lin, int dil, hipStream_t s) {
  const int Lo = Lin - dil;
  // wide kernel for long planes (C2's one-second chunks: 290 vs 363 us per layer); at the bench shape (64 planes of ~6000
  // samples, 3008 super-tiles) the dword kernel's finer tiles balance better and the two time the same (37-38 us).  The
  // choice depends on the plane LENGTH only: the two forms round the residual add differently (last bit), and a sequence's
  // result must not depend on how many others share its batch.
  if (buf_ok(B, Lin) && (avvad_tune().wn_flat == 0 || avvad_tune().wn_flat == 3) && Lo >= 128 &&
      (Lo >= WIDE_FROM || avvad_tune().wn_flat == 3)) {
    const long ntiles = (long)B * cdiv(Lo, 128);
    long blocks = (ntiles + 3) / 4;
    const long cap = avvad_tune().wn_grid > 0 ? avvad_tune().wn_grid : 1024;
    if (blocks > cap) blocks = cap;
    blocks = (blocks + 7) / 8 * 8;                  // whole XCD groups (surplus waves find no tile)
    hipLaunchKernelGGL(wn_block_fwd_w4, dim3((int)blocks), dim3(256), 0, s, s_in, wd, bd, we, be, s_out, B, Lin, dil);
    return;
  }
  const long ntiles = (long)B * cdiv(Lo, 32);
  long blocks = (ntiles + 3) / 4;
  if (blocks > 512) blocks = 512;    // 2 waves per SIMD resident; each wave walks >= 5 tiles at the bench shape
  if (avvad_tune().wn_grid > 0 && blocks > avvad_tune().wn_grid) blocks = avvad_tune().wn_grid;
  if (buf_ok(B, Lin) && avvad_tune().wn_flat == 5 && Lo >= 32) {       // LDS-DMA form
    long wb = (ntiles + 7) / 8;                       // 8 waves per workgroup, two workgroups per CU
    if (wb > 512) wb = 512;
    if (avvad_tune().wn_grid > 0 && wb > avvad_tune().wn_grid) wb = avvad_tune().wn_grid;
    if (wb >= 8) wb = wb / 8 * 8 + ((wb & 7) ? 8 : 0);
    hipLaunchKernelGGL(wn_block_fwd_dma, dim3((int)wb), dim3(512), 0, s, s_in, wd, bd, we, be, s_out, B, Lin, dil);
    return;
  }
  if (buf_ok(B, Lin) && (avvad_tune().wn_flat == 0 || avvad_tune().wn_flat == 4)) {
    long ob = (ntiles + 3) / 4;                       // 4 waves per SIMD resident: 1024 workgroups
    if (ob > 1024) ob = 1024;
    if (avvad_tune().wn_grid > 0 && ob > avvad_tune().wn_grid) ob = avvad_tune().wn_grid;
    if (ob >= 8) ob = ob / 8 * 8 + ((ob & 7) ? 8 : 0);
    hipLaunchKernelGGL(wn_block_fwd_occ<0>, dim3((int)ob), dim3(256), 0, s, s_in, wd, bd, we, be, s_out, (float*)nullptr, B, Lin,
                       dil);
    return;
  }
  if (buf_ok(B, Lin))
    hipLaunchKernelGGL(wn_block_fwd_buf<0>, dim3((int)blocks), dim3(256), 0, s, s_in, wd, bd, we, be, s_out, (float*)nullptr, B, Lin,
                       dil);
  else
    hipLaunchKernelGGL(wn_block_fwd_mfma<0>, dim3((int)blocks), dim3(256), 0, s, s_in, wd, bd, we, be, s_out, (float*)nullptr, B, Lin,
                       dil);
}

// dw[co][ci][k] += sum_{b,t} dy[b][co][t] * f(x[b][ci][t + k*dil])  for every tap, on the engine
static int wgrad_conv1d(const float* dy, const float* x, float* dw, int B, int Cout, int Cin, int Lout, int Lin, int fw,
                        int dil, int relu_in, hipStream_t s, float* slab) {
  const long K = (long)B * Lout;
  if (K > 0x7fffffffL) return AVVAD_EINVAL;
  const int ktiles = cdiv(K, igemm::BK);
  int split = 1024 / (cdiv(Cout, 64) * cdiv(Cin * fw, 64));
  if (split > ktiles) split = ktiles;
  if (split < 1) split = 1;
  // one launch for all taps: column n = ci*fw + k is exactly the [Cout][Cin][fw] weight layout
  igemm::RowSegK a{dy, Lout, (long)Cout * Lout, Cout, (int)K, Lout, 0, 0, 1, 0};
  igemm::RowSegK b{x, Lin, (long)Cin * Lin, Cin * fw, (int)K, Lout, 0, relu_in, fw, dil};
  igemm::EpiStore e{dw, (long)Cin * fw, nullptr, 1};     // dw += (split tiles combined by the engine's fix-up kernel)
  return igemm::launch<64, 64>(a, b, e, Cout, Cin * fw, (int)K, split, s, slab);
}

static void bias_grad(const float* dy, float* db, int B, int C, int L, hipStream_t s) {
  if (!db) return;
  // (one workgroup per channel: with several per channel adding atomically the sum depended on their arrival order --
  //  the one gradient of the generic encoder path that was not bit-reproducible run to run)
  hipLaunchKernelGGL(chan_sum_acc, dim3(C, 1), dim3(256), 0, s, dy, db, B, C, L);
}

}  // namespace

extern "C" size_t avvad_wavenet_workspace(const avvad_wavenet_desc* d) {
  Plan p;
  if (make_plan(d, &p)) return 0;
  return p.total * sizeof(float);
}

extern "C" int avvad_wavenet_fwd(const float* wave, const avvad_wavenet_params* prm, float* out, const avvad_wavenet_desc* d,
                                 void* wsv, size_t ws_bytes, avvad_stream_t sv) {
  AVVAD_ENTER();
  Plan p;
  if (!wave || !prm || !out || !wsv || make_plan(d, &p)) return AVVAD_EINVAL;
  if (ws_bytes < p.total * sizeof(float)) return AVVAD_EWORKSPACE;
  hipStream_t s = (hipStream_t)sv;
  float* ws = (float*)wsv;
  const int B = d->B, R = d->R, D = d->D, fw = d->fw;
  // causal layer: Conv1d(qc -> R, k = fw), no input ReLU   (:75)
  hipLaunchKernelGGL(conv1d_fwd_generic, dim3(grid1((long)B * R * p.L[0])), dim3(256), 0, s, wave, prm->causal_w,
                     d->use_bias ? prm->causal_b : (const float*)nullptr, (const float*)nullptr, ws + p.s[0], B, d->qc, R,
                     d->L, p.L[0], fw, 1, 0, 0, 0);
  for (int i = 0; i < p.n; ++i) {
    const int dil = d->dilations_h[i];
    const float* bd = d->use_bias ? prm->dil_b_h[i] : nullptr;
    const float* be = d->use_bias ? prm->dense_b_h[i] : nullptr;
    if (mfma_shape(d)) {
      // z (pre-ReLU dilation output) is NOT kept: backward rebuilds it from s_i inside its fused kernel, bit-identically
      launch_block_fwd(ws + p.s[i], prm->dil_w_h[i], bd, prm->dense_w_h[i], be, ws + p.s[i + 1], B, p.L[i], dil, s);
    } else {
      float* z = ws + p.z;
      hipLaunchKernelGGL(conv1d_fwd_generic, dim3(grid1((long)B * D * p.L[i + 1])), dim3(256), 0, s, ws + p.s[i],
                         prm->dil_w_h[i], bd, (const float*)nullptr, z, B, R, D, p.L[i], p.L[i + 1], fw, dil, 1, 0, 0);
      hipLaunchKernelGGL(conv1d_fwd_generic, dim3(grid1((long)B * R * p.L[i + 1])), dim3(256), 0, s, z, prm->dense_w_h[i], be,
                         ws + p.s[i], ws + p.s[i + 1], B, D, R, p.L[i + 1], p.L[i + 1], 1, 1, 1, p.L[i] - p.L[i + 1], p.L[i]);
    }
  }
  const float* bbp = d->use_bias ? prm->bott_b : (const float*)nullptr;
  if (R == 32 && d->Bn % 32 == 0) {
    const int nt = (d->Bn % 128 == 0) ? 4 : 1;
    long blocks = ((long)B * d->P * (d->Bn / (32 * nt)) + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    if (nt == 4)
      hipLaunchKernelGGL(tail_fwd_mfma<4>, dim3((int)blocks), dim3(256), 0, s, ws + p.s[p.n], prm->bott_w, bbp, out, B, d->Bn,
                         p.L[p.n], d->P);
    else
      hipLaunchKernelGGL(tail_fwd_mfma<1>, dim3((int)blocks), dim3(256), 0, s, ws + p.s[p.n], prm->bott_w, bbp, out, B, d->Bn,
                         p.L[p.n], d->P);
  } else {
    hipLaunchKernelGGL(tail_fwd_generic, dim3(d->P, B), dim3(256), 0, s, ws + p.s[p.n], prm->bott_w, bbp, out, R, d->Bn,
                       p.L[p.n], d->P);
  }
  AVVAD_LAUNCH_CHECK();
  return AVVAD_OK;
}

// One residual block alone (the layer-at-a-time kernel the large dilations run on): what bench.py times per launch for
// the HBM roofline entry, like avvad_conv2d_fwd for the MFMA one.
extern "C" int avvad_wavenet_block_fwd(const float* s_in, const float* w_dil, const float* b_dil, const float* w_dense,
                                       const float* b_dense, float* s_out, int B, int Lin, int dil, avvad_stream_t sv) {
  AVVAD_ENTER();
  if (!s_in || !w_dil || !w_dense || !s_out || B <= 0 || dil < 1 || Lin - dil < 1) return AVVAD_EINVAL;
  launch_block_fwd(s_in, w_dil, b_dil, w_dense, b_dense, s_out, B, Lin, dil, (hipStream_t)sv);
  AVVAD_LAUNCH_CHECK();
  return AVVAD_OK;
}

extern "C" int avvad_wavenet_bwd(const float* wave, const avvad_wavenet_params* prm, const float* dout,
                                 const avvad_wavenet_grads* g, float* dwave, const avvad_wavenet_desc* d, void* wsv,
                                 size_t ws_bytes, avvad_stream_t sv) {
  AVVAD_ENTER();
  Plan p;
  if (!wave || !prm || !dout || !g || !wsv || make_plan(d, &p) || !d->save_for_backward) return AVVAD_EINVAL;
  if (ws_bytes < p.total * sizeof(float)) return AVVAD_EWORKSPACE;
  hipStream_t s = (hipStream_t)sv;
  float* ws = (float*)wsv;
  const int B = d->B, R = d->R, D = d->D, fw = d->fw, Bn = d->Bn;
  int rc;
  const int Lv = p.L[p.n];
  float* Z = ws + p.z;
  float* DZ = ws + p.dz;
  float* GA = ws + p.ga;  // d s_{i+1}
  float* GB = ws + p.gb;  // d s_i
  // ---- tail: dz_t = relu'(z) * pooled-grad ; d s_N = Wb^T dz_t
  float* DZT = ws + p.dzt;
  const bool tail_mfma = (R == 32 && Bn % 32 == 0 && Bn <= 1024);
  const bool tail_fused = tail_mfma && Bn == 256 && (g->bott_w || (d->use_bias && g->bott_b)) && !avvad_tune().wn_no_fused_tail;
  if (tail_fused) {
    // d s_N and the bottleneck's weight + bias gradients in one pass; dz_t is never materialised
    long blocks = ((long)B * cdiv(Lv, 32) + 3) / 4;
    if (blocks > 256) blocks = 256;        // 1 workgroup / CU is resident: one round
    if (!avvad_tune().wn_no_tail_pair) {
      // two waves per time tile, 8 waves per workgroup (tail_bwd_wgrad_pair)
      const size_t lds = ((size_t)Bn * 33 + (size_t)Bn * 33 + 16 * 1056 + 4 * 1056) * sizeof(float);
      static bool attr_set2 = false;
      if (!attr_set2) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(tail_bwd_wgrad_pair<8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) return AVVAD_ELAUNCH;
        attr_set2 = true;
      }
      hipLaunchKernelGGL(tail_bwd_wgrad_pair<8>, dim3((int)blocks), dim3(512), lds, s, ws + p.s[p.n], prm->bott_w,
                         d->use_bias ? prm->bott_b : (const float*)nullptr, dout, GA, ws + p.slab, B, Lv, d->P);
    } else {
    const size_t lds = ((size_t)Bn * 33 + (size_t)Bn * 33 + 8 * 1056) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(tail_bwd_wgrad_mfma<8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds) != hipSuccess) return AVVAD_ELAUNCH;
      attr_set = true;
    }
    hipLaunchKernelGGL(tail_bwd_wgrad_mfma<8>, dim3((int)blocks), dim3(256), lds, s, ws + p.s[p.n], prm->bott_w,
                       d->use_bias ? prm->bott_b : (const float*)nullptr, dout, GA, ws + p.slab, B, Lv, d->P);
    }
    hipLaunchKernelGGL(tail_wgrad_reduce, dim3(cdiv(Bn * 33, 32)), dim3(256), 0, s, ws + p.slab, (int)blocks, Bn, g->bott_w,
                       d->use_bias ? g->bott_b : (float*)nullptr);
  } else if (tail_mfma) {
    long blocks = ((long)B * cdiv(Lv, 32) + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(tail_bwd_mfma, dim3((int)blocks), dim3(256), (size_t)Bn * 33 * sizeof(float), s, ws + p.s[p.n], prm->bott_w,
                       d->use_bias ? prm->bott_b : (const float*)nullptr, dout, DZT, GA, B, Bn, Lv, d->P);
  } else {
    hipLaunchKernelGGL(tail_bwd_dz_generic, dim3(grid1((long)B * Bn * Lv)), dim3(256), 0, s, ws + p.s[p.n], prm->bott_w,
                       d->use_bias ? prm->bott_b : (const float*)nullptr, dout, DZT, B, R, Bn, Lv, d->P);
  }
  if (!tail_fused) {
    if (g->bott_w && (rc = wgrad_conv1d(DZT, ws + p.s[p.n], g->bott_w, B, Bn, R, Lv, Lv, 1, 1, 0, s, ws + p.gslab))) return rc;
    if (d->use_bias) bias_grad(DZT, g->bott_b, B, Bn, Lv, s);
  }
  if (!tail_mfma)
    hipLaunchKernelGGL(conv1d_bwd_data_generic, dim3(grid1((long)B * R * Lv)), dim3(256), 0, s, DZT, prm->bott_w,
                       (const float*)nullptr, (const float*)nullptr, GA, B, R, Bn, Lv, Lv, 1, 1, 0, 0);
  // ---- residual blocks, last to first
  WgradTab wtab;
  bool any_slab = false;
  for (int l = 0; l < WN_MAXL; ++l) {
    wtab.slab[l] = nullptr; wtab.nslab[l] = 0;
    wtab.dW_dil[l] = wtab.db_dil[l] = wtab.dW_dense[l] = wtab.db_dense[l] = nullptr;
  }
  auto note_slabs = [&](int i, float* slab_i, long wb) {
    wtab.slab[i] = slab_i; wtab.nslab[i] = (int)wb;
    wtab.dW_dil[i] = g->dil_w_h[i]; wtab.dW_dense[i] = g->dense_w_h[i];
    wtab.db_dil[i] = d->use_bias ? g->dil_b_h[i] : (float*)nullptr;
    wtab.db_dense[i] = d->use_bias ? g->dense_b_h[i] : (float*)nullptr;
    any_slab = true;
  };
  for (int i = p.n - 1; i >= 0; --i) {
    float* const slab_i = ws + p.slab + (size_t)i * WG_MAXBLK * WG_SLAB;     // this block's partial sums (reduced at the end)
    const int dil = d->dilations_h[i];
    const int Li = p.L[i], Lo = p.L[i + 1];
    const float* si = ws + p.s[i];
    const float* bd = d->use_bias ? prm->dil_b_h[i] : nullptr;
    const bool fast = mfma_shape(d);
    float* Zi = fast ? ws + p.zs[i] : Z;
    if (!fast)  // recompute z = dil_i(relu(s_i)) (pre-ReLU)
      hipLaunchKernelGGL(conv1d_fwd_generic, dim3(grid1((long)B * D * Lo)), dim3(256), 0, s, si, prm->dil_w_h[i], bd,
                         (const float*)nullptr, Zi, B, R, D, Li, Lo, fw, dil, 1, 0, 0);
    if (fast) {
      // dz = (z>0) * W_dense^T dS ; then all four parameter gradients in one pass ; then d s_i
      long blocks = ((long)B * cdiv(Lo, 32) + 3) / 4;
      // a block's four parameter gradients are wanted independently (partially frozen blocks): the fused kernel always
      // forms all four sums, the reduce skips the NULL destinations
      const bool any_grad = g->dil_w_h[i] || g->dense_w_h[i] || (d->use_bias && (g->dil_b_h[i] || g->dense_b_h[i]));
      if (any_grad && !avvad_tune().wn_no_fused_wgrad) {
        // dz AND the four parameter gradients in one pass over dS, Z and the two shifted s_i tiles
        long wb = ((long)B * cdiv(Lo, 32) + 15) / 16;   // >= 4 tiles per wave
        // one workgroup per CU is resident (368 VGPRs): ONE round of 256, not two of 512 -- every workgroup pays a weight
        // staging prologue and a 12.5 KB slab epilogue (step 22.57 -> 22.06 ms; 128..256 workgroups time the same)
        if (wb > 256) wb = 256;
        if (wb > WG_MAXBLK) wb = WG_MAXBLK;
        if (wb < 1) wb = 1;
        // option wn_bwd_t: 0 by the descriptor's shared_device hint, 1 transposed-product alternate, 2 high-occupancy form,
        // 3 resident-weights form.  Alone on the device the two-waves-per-SIMD form wins (encoder fwd+bwd 3.09 -> 2.90 ms);
        // beside the trunk's backward on the other stream it costs the STEP 0.25 ms (same-process A/B), like the dx kernel.
        const int bt = avvad_tune().wn_bwd_t == 0 ? (d->shared_device ? 3 : 2) : avvad_tune().wn_bwd_t;
        if (bt == 3) {
          hipLaunchKernelGGL(wn_block_bwd_dz_wgrad_mfma, dim3((int)wb), dim3(256), 0, s, GA, prm->dil_w_h[i], bd,
                             prm->dense_w_h[i], si, DZ, slab_i, B, Li, dil);
        } else if (bt == 2) {
          wb = ((long)B * cdiv(Lo, 32) + 15) / 16;
          if (wb > 512) wb = 512;
          if (wb >= 8) wb = wb / 8 * 8;
          if (wb < 1) wb = 1;
          hipLaunchKernelGGL(wn_block_bwd_dz_wgrad_occ, dim3((int)wb), dim3(256), 0, s, GA, prm->dil_w_h[i], bd,
                             prm->dense_w_h[i], si, DZ, slab_i, B, Li, dil);
        } else {
          // alternate: transposed-product form, under 256 registers, two workgroups per CU (one wave per SIMD each)
          wb = ((long)B * cdiv(Lo, 32) + 15) / 16;
          if (wb > 512) wb = 512;
          if (wb < 1) wb = 1;
          hipLaunchKernelGGL(wn_block_bwd_dzw_t, dim3((int)wb), dim3(256), 0, s, GA, prm->dil_w_h[i], bd, prm->dense_w_h[i], si,
                             DZ, slab_i, B, Li, dil);
        }
        note_slabs(i, slab_i, wb);
      } else {
        // (frozen weights / tuning switch) separate kernels: z is not kept by the forward -> rebuild it first
        long fb = blocks > 512 ? 512 : blocks;
        hipLaunchKernelGGL(wn_block_fwd_mfma<2>, dim3((int)fb), dim3(256), 0, s, si, prm->dil_w_h[i], bd, prm->dense_w_h[i],
                           (const float*)nullptr, (float*)nullptr, Zi, B, Li, dil);
        if (blocks > 768) blocks = 768;
        hipLaunchKernelGGL(wn_block_bwd_dz_mfma, dim3((int)blocks), dim3(256), 0, s, GA, Zi, prm->dense_w_h[i], DZ, B, Lo);
        if (any_grad) {
          long wb = ((long)B * cdiv(Lo, 32) + 15) / 16;   // >= 4 tiles per wave
          if (wb > WG_MAXBLK) wb = WG_MAXBLK;
          if (wb < 1) wb = 1;
          hipLaunchKernelGGL(wn_block_wgrad_mfma, dim3((int)wb), dim3(256), 0, s, GA, Zi, DZ, si, slab_i, B, Li, dil);
          note_slabs(i, slab_i, wb);
        }
      }
      blocks = ((long)B * cdiv(Li, 32) + 3) / 4;
      // The high-occupancy dx kernel everywhere.  (Round 2 measured it 0.16 ms/step SLOWER beside the trunk's backward on another
      // stream, and the shared_device hint picked the resident-weights form there; with this round's trunk kernels -- layer 1 off
      // the persistent engine, fewer and longer class launches -- the same A/B reads 0.06 ms FASTER: bench.py --ab wn_dx=1.)
      // Option wn_dx: 1 forces the resident-weights kernel, 2 the high-occupancy one.
      const bool dx_occ = avvad_tune().wn_dx == 2 || avvad_tune().wn_dx == 0;
      if (buf_ok(B, Li) && avvad_tune().wn_flat != 1 && dx_occ) {
        if (blocks > 1024) blocks = 1024;                 // 4 waves per SIMD resident
        if (blocks >= 8) blocks = (blocks + 7) / 8 * 8;
        hipLaunchKernelGGL(wn_block_bwd_dx_occ, dim3((int)blocks), dim3(256), 0, s, DZ, si, GA, prm->dil_w_h[i], GB, B, Li, dil);
      } else if (buf_ok(B, Li) && avvad_tune().wn_flat != 1 && avvad_tune().wn_dx != 3) {     // (wn_dx = 3: the flat kernel)
        if (blocks > 512) blocks = 512;
        hipLaunchKernelGGL(wn_block_bwd_dx_buf, dim3((int)blocks), dim3(256), 0, s, DZ, si, GA, prm->dil_w_h[i], GB, B, Li, dil);
      } else {
        if (blocks > 512) blocks = 512;
        hipLaunchKernelGGL(wn_block_bwd_dx_mfma, dim3((int)blocks), dim3(256), 0, s, DZ, si, GA, prm->dil_w_h[i], GB, B, Li, dil);
      }
    } else {
      // dense (1x1) layer: d W_dense = GA . relu(z)^T ; d b ; dz = (z>0) * W_dense^T GA
      if (g->dense_w_h[i] && (rc = wgrad_conv1d(GA, Zi, g->dense_w_h[i], B, R, D, Lo, Lo, 1, 1, 1, s, ws + p.gslab))) return rc;
      if (d->use_bias) bias_grad(GA, g->dense_b_h[i], B, R, Lo, s);
      hipLaunchKernelGGL(conv1d_bwd_data_generic, dim3(grid1((long)B * D * Lo)), dim3(256), 0, s, GA, prm->dense_w_h[i], Zi,
                         (const float*)nullptr, DZ, B, D, R, Lo, Lo, 1, 1, 0, 0);
      // dilated layer: d W_dil = dz . relu(s_i shifted)^T ; d b ; d s_i = (s_i>0) * W_dil^T (*) dz + left-padded GA
      if (g->dil_w_h[i] && (rc = wgrad_conv1d(DZ, si, g->dil_w_h[i], B, D, R, Lo, Li, fw, dil, 1, s, ws + p.gslab))) return rc;
      if (d->use_bias) bias_grad(DZ, g->dil_b_h[i], B, D, Lo, s);
      hipLaunchKernelGGL(conv1d_bwd_data_generic, dim3(grid1((long)B * R * Li)), dim3(256), 0, s, DZ, prm->dil_w_h[i], si, GA,
                         GB, B, R, D, Li, Lo, fw, dil, Li - Lo, Lo);
    }
    float* t = GA; GA = GB; GB = t;
  }
  if (any_slab) hipLaunchKernelGGL(wn_wgrad_reduce_all, dim3(WG_SLAB / 32, p.n), dim3(256), 0, s, wtab);
  // ---- causal layer
  if (d->qc * fw <= 4) {   // narrow causal layer: weight + bias gradient in one pass over GA
    if (g->causal_w || (d->use_bias && g->causal_b))
    {
      const int ny = B < 32 ? B : 32;
      float* part = ws + p.gslab;          // the engine scratch is idle here
      hipLaunchKernelGGL(narrow_conv1d_grads, dim3(R, ny), dim3(256), 0, s, GA, wave, part, B, R, d->qc, p.L[0], d->L, fw);
      hipLaunchKernelGGL(narrow_conv1d_grads_sum, dim3(cdiv(R * 5, 256)), dim3(256), 0, s, part, ny, R, d->qc * fw, g->causal_w,
                         d->use_bias ? g->causal_b : (float*)nullptr);
    }
  } else {
    if (g->causal_w && (rc = wgrad_conv1d(GA, wave, g->causal_w, B, R, d->qc, p.L[0], d->L, fw, 1, 0, s, ws + p.gslab))) return rc;
    if (d->use_bias) bias_grad(GA, g->causal_b, B, R, p.L[0], s);
  }
  if (dwave)
    hipLaunchKernelGGL(conv1d_bwd_data_generic, dim3(grid1((long)B * d->qc * d->L)), dim3(256), 0, s, GA, prm->causal_w,
                       (const float*)nullptr, (const float*)nullptr, dwave, B, d->qc, R, d->L, p.L[0], fw, 1, 0, 0);
  AVVAD_LAUNCH_CHECK();
  return AVVAD_OK;
}
