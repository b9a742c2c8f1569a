// mem_lab.hip -- what HBM rate do the encoder kernels' ACCESS PATTERNS reach on their own (no MFMA)?
//   pattern 0: the layer kernels' direct feed: per wave a 32-sample x 32-channel tile, one dword per lane per load, lane half h
//              reads tap h (two 128-byte row pieces per instruction), 48 loads in flight, 16 dword stores (rows 128 B)
//   pattern 1: row-contiguous float4: a wave reads 256 consecutive samples of one row per instruction (1 KB), 8 in flight
//   pattern 2: like 0 but each wave walks TWO adjacent tiles per step (256 B contiguous per row per wave)
// out = in (+ dilated tap) so the compiler cannot drop anything; planes [B][32][L].
#include <hip/hip_runtime.h>
extern "C" __global__ void __launch_bounds__(256) pat0(const float* __restrict__ in, float* __restrict__ out, int B, int L, int dil) {
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int Lo = L - dil, tps = (Lo + 31) >> 5;
  const long ntiles = (long)B * tps;
  const long w0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
  for (long tile = w0; tile < ntiles; tile += nw) {
    const int b = (int)(tile / tps), t = (int)(tile - (long)b * tps) * 32 + li;
    const int tc = t < Lo ? t : 0;
    const float* xp = in + (long)b * 32 * L + tc + lh * dil;
    float x[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) x[c] = xp[(long)c * L];
    float* op = out + (long)b * 32 * Lo + tc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = x[2 * r] + x[2 * r + 1] + __shfl_xor(x[2 * r], 32, 64);
      if (t < Lo) op[(long)(2 * r + lh) * Lo] = v;
    }
  }
}
extern "C" __global__ void __launch_bounds__(256) pat1(const float* __restrict__ in, float* __restrict__ out, int B, int L, int dil) {
  // rows of Lo4 = Lo/4 float4 (L, Lo multiples of 4 here); a wave takes 64 float4 = 256 samples of one row
  const int lane = threadIdx.x & 63;
  const int Lo = L - dil;
  const long chunks_per_row = (Lo + 255) / 256;
  const long nchunks = (long)B * 32 * chunks_per_row;
  const long w0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
  for (long ch = w0; ch < nchunks; ch += nw * 4) {
    float4 v[4], u[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long c2 = ch + k * nw < nchunks ? ch + k * nw : ch;
      const long row = c2 / chunks_per_row;
      const int t = (int)(c2 - row * chunks_per_row) * 256 + lane * 4;
      const int tc = t + 3 < Lo ? t : 0;
      v[k] = *reinterpret_cast<const float4*>(in + row * L + tc);
      u[k] = *reinterpret_cast<const float4*>(in + row * L + tc + dil);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long c2 = ch + k * nw;
      if (c2 >= nchunks) continue;
      const long row = c2 / chunks_per_row;
      const int t = (int)(c2 - row * chunks_per_row) * 256 + lane * 4;
      if (t + 3 < Lo) *reinterpret_cast<float4*>(out + row * Lo + t) = make_float4(v[k].x + u[k].x, v[k].y + u[k].y, v[k].z + u[k].z, v[k].w + u[k].w);
    }
  }
}
extern "C" __global__ void __launch_bounds__(256) pat2(const float* __restrict__ in, float* __restrict__ out, int B, int L, int dil) {
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int Lo = L - dil, tps = (Lo + 63) >> 6;     // 64-sample double tiles
  const long ntiles = (long)B * tps;
  const long w0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
  for (long tile = w0; tile < ntiles; tile += nw) {
    const int b = (int)(tile / tps), t0 = (int)(tile - (long)b * tps) * 64;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int t = t0 + 32 * h2 + li;
      const int tc = t < Lo ? t : 0;
      const float* xp = in + (long)b * 32 * L + tc + lh * dil;
      float x[32];
#pragma unroll
      for (int c = 0; c < 32; ++c) x[c] = xp[(long)c * L];
      float* op = out + (long)b * 32 * Lo + tc;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = x[2 * r] + x[2 * r + 1] + __shfl_xor(x[2 * r], 32, 64);
        if (t < Lo) op[(long)(2 * r + lh) * Lo] = v;
      }
    }
  }
}
#define LAUNCH(n) extern "C" void launch_##n(const float* a, float* b, int B, int L, int dil, int blocks, hipStream_t s) { hipLaunchKernelGGL(n, dim3(blocks), dim3(256), 0, s, a, b, B, L, dil); }
LAUNCH(pat0) LAUNCH(pat1) LAUNCH(pat2)

// ---- the forward block with parts switched off: what do loads / MFMAs / stores cost in situ?
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ constexpr int mrow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
template <int LOADS, int MFMA, int RES, int WPS>
__global__ void __launch_bounds__(256, WPS) fwdlab(const float* __restrict__ s_in, float* __restrict__ s_out, int B, int Lin, int dil) {
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int Lo = Lin - dil, tps = (Lo + 31) >> 5;
  const long ntiles = (long)B * tps;
  const long w0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
  __shared__ float wl[32 * 65 + 32 * 33 + 64];
  for (int i = threadIdx.x; i < 32 * 65 + 32 * 33 + 64; i += 256) wl[i] = 0.001f * (i % 17);
  __syncthreads();
  const float* wdp = wl + li * 65 + lh;
  const float* wep = wl + 2080 + li * 33;
  for (long tile = w0; tile < ntiles; tile += nw) {
    const int b = (int)(tile / tps), t = (int)(tile - (long)b * tps) * 32 + li;
    const bool ok = t < Lo;
    const int tcl = ok ? t : 0;
    float x[32], rv[16];
    const float* xp = s_in + (long)b * 32 * Lin + tcl + lh * dil;
    if (LOADS) {
#pragma unroll
      for (int c = 0; c < 32; ++c) x[c] = xp[(long)c * Lin];
    } else {
#pragma unroll
      for (int c = 0; c < 32; ++c) x[c] = (float)(t + c);
    }
    const float* rp = s_in + (long)b * 32 * Lin + tcl + dil;
    if (RES && LOADS) {
#pragma unroll
      for (int r = 0; r < 16; ++r) rv[r] = rp[(long)mrow(r, lh) * Lin];
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) rv[r] = 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (MFMA) {
#pragma unroll
      for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wdp[2 * s], ok ? fmaxf(x[s], 0.f) : 0.f, acc, 0, 0, 0);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = x[2 * r] + x[2 * r + 1];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = (ok ? rv[r] : 0.f);
    if (MFMA) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wep[mrow(r, lh)], fmaxf(acc[r], 0.f), acc2, 0, 0, 0);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[r] += acc[r];
    }
    float* op = s_out + (long)b * 32 * Lo + t;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (ok) op[(long)mrow(r, lh) * Lo] = acc2[r];
  }
}
#define LAUNCHT(n, L, M, R, W) extern "C" void launch_##n(const float* a, float* b, int B, int Lin, int dil, int blocks, hipStream_t s) { hipLaunchKernelGGL((fwdlab<L, M, R, W>), dim3(blocks), dim3(256), 0, s, a, b, B, Lin, dil); }
LAUNCHT(full3, 1, 1, 1, 3) LAUNCHT(nomfma3, 1, 0, 1, 3) LAUNCHT(noload3, 0, 1, 1, 3) LAUNCHT(nores3, 1, 1, 0, 3) LAUNCHT(full2, 1, 1, 1, 2) LAUNCHT(noresnomfma3, 1, 0, 0, 3)

// ---- MFMA-phase ablations (no global loads): what limits the matrix rate of a block tile?
//   WREG: weight fragments in registers instead of a ds_read per k-step;  RELU: the v_max / v_cndmask in front of every MFMA;
//   DUAL: two tiles interleaved per wave (two independent accumulator chains)
template <int WREG, int RELU, int DUAL, int WPS, int NOSTORE = 0>
__global__ void __launch_bounds__(256, WPS) mfmalab(const float* __restrict__ s_in, float* __restrict__ s_out, int B, int Lin, int dil) {
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int Lo = Lin - dil, tps = (Lo + 31) >> 5;
  const long ntiles = (long)B * tps;
  const long w0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
  __shared__ float wl[32 * 65 + 32 * 33 + 64];
  for (int i = threadIdx.x; i < 32 * 65 + 32 * 33 + 64; i += 256) wl[i] = 0.001f * (i % 17);
  __syncthreads();
  const float* wdp = wl + li * 65 + lh;
  const float* wep = wl + 2080 + li * 33;
  float wd[32], we[16];
  if (WREG) {
#pragma unroll
    for (int s = 0; s < 32; ++s) wd[s] = wdp[2 * s];
#pragma unroll
    for (int r = 0; r < 16; ++r) we[r] = wep[mrow(r, lh)];
  }
  for (long tile = w0; tile < ntiles; tile += nw * (DUAL ? 2 : 1)) {
    f32x16 accA[2], accB[2];
    float x[2][32];
    bool ok[2];
    int tt[2], bb[2];
#pragma unroll
    for (int u = 0; u < (DUAL ? 2 : 1); ++u) {
      const long tl = tile + u * nw < ntiles ? tile + u * nw : tile;
      bb[u] = (int)(tl / tps); tt[u] = (int)(tl - (long)bb[u] * tps) * 32 + li;
      ok[u] = tt[u] < Lo;
#pragma unroll
      for (int c = 0; c < 32; ++c) x[u][c] = (float)(tt[u] - 3000 + c);
#pragma unroll
      for (int r = 0; r < 16; ++r) { accA[u][r] = 0.f; accB[u][r] = 0.f; }
    }
#pragma unroll
    for (int s = 0; s < 32; ++s)
#pragma unroll
      for (int u = 0; u < (DUAL ? 2 : 1); ++u) {
        const float bv = RELU ? (ok[u] ? fmaxf(x[u][s], 0.f) : 0.f) : x[u][s];
        accA[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(WREG ? wd[s] : wdp[2 * s], bv, accA[u], 0, 0, 0);
      }
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int u = 0; u < (DUAL ? 2 : 1); ++u) {
        const float bv = RELU ? fmaxf(accA[u][r], 0.f) : accA[u][r];
        accB[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(WREG ? we[r] : wep[mrow(r, lh)], bv, accB[u], 0, 0, 0);
      }
#pragma unroll
    for (int u = 0; u < (DUAL ? 2 : 1); ++u) {
      if (tile + u * nw >= ntiles) continue;
      float* op = s_out + (long)bb[u] * 32 * Lo + tt[u];
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (ok[u] && (!NOSTORE || accB[u][r] == 12345.678f)) op[(long)mrow(r, lh) * Lo] = accB[u][r];
    }
  }
}
#define LAUNCHM(n, a, b, c, d) extern "C" void launch_##n(const float* x, float* y, int B, int Lin, int dil, int blocks, hipStream_t s) { hipLaunchKernelGGL((mfmalab<a, b, c, d>), dim3(blocks), dim3(256), 0, s, x, y, B, Lin, dil); }
LAUNCHM(m_lds_relu, 0, 1, 0, 3) LAUNCHM(m_reg_relu, 1, 1, 0, 3) LAUNCHM(m_reg_norelu, 1, 0, 0, 3) LAUNCHM(m_reg_relu_dual, 1, 1, 1, 2) LAUNCHM(m_reg_norelu_dual, 1, 0, 1, 2) LAUNCHM(m_lds_relu_dual, 0, 1, 1, 2) LAUNCHM(m_reg_relu_2w, 1, 1, 0, 2) LAUNCHM(m_reg_relu_1w, 1, 1, 0, 1)

extern "C" void launch_m_nostore(const float* x, float* y, int B, int Lin, int dil, int blocks, hipStream_t s) { hipLaunchKernelGGL((mfmalab<1, 1, 0, 3, 1>), dim3(blocks), dim3(256), 0, s, x, y, B, Lin, dil); }
extern "C" void launch_m_nostore_dual(const float* x, float* y, int B, int Lin, int dil, int blocks, hipStream_t s) { hipLaunchKernelGGL((mfmalab<1, 0, 1, 2, 1>), dim3(blocks), dim3(256), 0, s, x, y, B, Lin, dil); }

// ---- in-kernel clock of an MFMA-dense fp32 loop (MI355X_MICROARCH.md, DVFS give-back item 6): shader cycles (s_memtime)
// per 100 MHz reference tick (s_memrealtime), stamped around a long back-to-back chain; out[2*wg] = cycles, out[2*wg+1] = ref ticks
extern "C" __global__ void __launch_bounds__(256) clk_mfma(unsigned long long* __restrict__ out, float* __restrict__ sink, int iters, float seed) {
  const int lane = threadIdx.x & 63;
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  float a = seed + lane * 0.37f, b = 1.0f - lane * 0.011f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
    }
    a = a * 0.999f + 0.001f; b = b * 1.0001f - 0.0001f;
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  if (s == 12345.678f) sink[threadIdx.x] = s;
}
extern "C" void launch_clk(unsigned long long* out, float* sink, int iters, float seed, int blocks, hipStream_t s) {
  hipLaunchKernelGGL(clk_mfma, dim3(blocks), dim3(256), 0, s, out, sink, iters, seed);
}

// the same stamped loop on RANDOM operands (per lane, changing every MFMA): data-dependent power -> clock
extern "C" __global__ void __launch_bounds__(256) clk_mfma_rand(unsigned long long* __restrict__ out, float* __restrict__ sink, const float* __restrict__ rnd, int iters) {
  __shared__ float tab[64 * 33];
  for (int i = threadIdx.x; i < 64 * 33; i += 256) tab[i] = rnd[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  float a[16], b[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) { a[u] = tab[lane * 33 + u]; b[u] = tab[lane * 33 + 16 + u]; }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[u], a[(u + 5) & 15], acc1, 0, 0, 0);
    }
    if ((i & 63) == 63) {     // keep the sums bounded
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] *= 1e-3f; acc1[r] *= 1e-3f; }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  if (s == 12345.678f) sink[threadIdx.x] = s;
}
extern "C" void launch_clk_rand(unsigned long long* out, float* sink, const float* rnd, int iters, int blocks, hipStream_t s) {
  hipLaunchKernelGGL(clk_mfma_rand, dim3(blocks), dim3(256), 0, s, out, sink, rnd, iters);
}

// ---- in-kernel clock of the forward block's work mix (loads + MFMA chain + stores), stamped around the whole tile loop
template <int LOADS, int MFMA>
__global__ void __launch_bounds__(256, 3) fwdclk(const float* __restrict__ s_in, float* __restrict__ s_out, unsigned long long* __restrict__ stamp, int B, int Lin, int dil) {
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int Lo = Lin - dil, tps = (Lo + 31) >> 5;
  const long ntiles = (long)B * tps;
  const long w0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
  __shared__ float wl[32 * 65 + 32 * 33 + 64];
  for (int i = threadIdx.x; i < 32 * 65 + 32 * 33 + 64; i += 256) wl[i] = 0.001f * (i % 17);
  __syncthreads();
  const float* wdp = wl + li * 65 + lh;
  const float* wep = wl + 2080 + li * 33;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (long tile = w0; tile < ntiles; tile += nw) {
    const int b = (int)(tile / tps), t = (int)(tile - (long)b * tps) * 32 + li;
    const bool ok = t < Lo;
    const int tcl = ok ? t : 0;
    float x[32], rv[16];
    const float* xp = s_in + (long)b * 32 * Lin + tcl + lh * dil;
    if (LOADS) {
#pragma unroll
      for (int c = 0; c < 32; ++c) x[c] = xp[(long)c * Lin];
      const float* rp = s_in + (long)b * 32 * Lin + tcl + dil;
#pragma unroll
      for (int r = 0; r < 16; ++r) rv[r] = rp[(long)mrow(r, lh) * Lin];
    } else {
#pragma unroll
      for (int c = 0; c < 32; ++c) x[c] = (float)(t + c) * 0.37f;
#pragma unroll
      for (int r = 0; r < 16; ++r) rv[r] = 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (MFMA) {
#pragma unroll
      for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wdp[2 * s], ok ? fmaxf(x[s], 0.f) : 0.f, acc, 0, 0, 0);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = x[2 * r] + x[2 * r + 1];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = (ok ? rv[r] : 0.f);
    if (MFMA) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wep[mrow(r, lh)], fmaxf(acc[r], 0.f), acc2, 0, 0, 0);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[r] += acc[r];
    }
    float* op = s_out + (long)b * 32 * Lo + t;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (ok && (LOADS || acc2[r] == 12345.678f)) op[(long)mrow(r, lh) * Lo] = acc2[r];
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) {
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    stamp[4 * w] = c1 - c0; stamp[4 * w + 1] = r1 - r0; stamp[4 * w + 2] = r0;
    stamp[4 * w + 3] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492);
  }
}
#define LAUNCHC(n, L, M) extern "C" void launch_##n(const float* a, float* b, unsigned long long* st, int B, int Lin, int dil, int blocks, hipStream_t s) { hipLaunchKernelGGL((fwdclk<L, M>), dim3(blocks), dim3(256), 0, s, a, b, st, B, Lin, dil); }
LAUNCHC(clk_full, 1, 1) LAUNCHC(clk_mfmaonly, 0, 1) LAUNCHC(clk_memonly, 1, 0)

// ---- what keeps the block's MFMA chain below the pipe's rate: variants of the MFMA-only body
// LDSW: weights from LDS (1) or registers (0); RELU: max+select on the B operand; TWO: two tiles' chains interleaved
template <int LDSW, int RELU, int TWO, int XGEN>
__global__ void __launch_bounds__(256, 2) mfvar(float* __restrict__ s_out, int ntiles_per_wave) {
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  __shared__ float wl[32 * 65 + 32 * 33 + 64];
  for (int i = threadIdx.x; i < 32 * 65 + 32 * 33 + 64; i += 256) wl[i] = 0.001f * (i % 17);
  __syncthreads();
  const float* wdp = wl + li * 65 + lh;
  const float* wep = wl + 2080 + li * 33;
  float wreg[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) wreg[i] = wdp[i];
  float keep = 0.f;
  for (int it = 0; it < ntiles_per_wave; it += (TWO ? 2 : 1)) {
    const int t = it * 32 + li;
    const bool ok = (t & 1023) != 1023;
    float x[32], x2[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) { x[c] = XGEN ? (float)(t + c) * 0.37f : keep + (float)c; x2[c] = XGEN ? (float)(t + c) * 0.11f : keep - (float)c; }
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc, acc2, bcc, bcc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; bcc[r] = 0.f; }
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      const float a = LDSW ? wdp[2 * s] : wreg[s & 7];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, RELU ? (ok ? fmaxf(x[s], 0.f) : 0.f) : x[s], acc, 0, 0, 0);
      if (TWO) bcc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, RELU ? (ok ? fmaxf(x2[s], 0.f) : 0.f) : x2[s], bcc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc2[r] = 0.f; bcc2[r] = 0.f; }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float a = LDSW ? wep[mrow(r, lh)] : wreg[r & 7];
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, RELU ? fmaxf(acc[r], 0.f) : acc[r], acc2, 0, 0, 0);
      if (TWO) bcc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, RELU ? fmaxf(bcc[r], 0.f) : bcc[r], bcc2, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) keep += acc2[r] + (TWO ? bcc2[r] : 0.f);
  }
  if (keep == 12345.678f) s_out[threadIdx.x] = keep;
}
#define LAUNCHV(n, A, B, C, D) extern "C" void launch_##n(float* b, int ntw, int blocks, hipStream_t s) { hipLaunchKernelGGL((mfvar<A, B, C, D>), dim3(blocks), dim3(256), 0, s, b, ntw); }
LAUNCHV(v_base, 1, 1, 0, 1) LAUNCHV(v_regw, 0, 1, 0, 1) LAUNCHV(v_regw_norelu, 0, 0, 0, 1) LAUNCHV(v_two, 1, 1, 1, 1) LAUNCHV(v_pure, 0, 0, 0, 0)
LAUNCHV(v_lds_norelu, 1, 0, 0, 1) LAUNCHV(v_two_pure, 0, 0, 1, 0)

// ---- the same work mix with buffer addressing: one descriptor per tensor and tile (SALU), ONE per-lane byte offset, the row
// offset in the instruction's scalar offset -> no per-load VALU address math; masked lanes carry an out-of-range offset (loads
// return 0, stores are dropped) -> no validity selects; ReLU as a one-instruction integer max
__device__ __forceinline__ float relu_i(float x) { const int v = __builtin_bit_cast(int, x); return __builtin_bit_cast(float, v > 0 ? v : 0); }
template <int LOADS, int MFMA>
__global__ void __launch_bounds__(256, 2) fwdclk2(const float* __restrict__ s_in, float* __restrict__ s_out, unsigned long long* __restrict__ stamp, int B, int Lin, int dil) {
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int Lo = Lin - dil, tps = (Lo + 31) >> 5;
  const int ntiles = B * tps;
  const int w0 = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6)), nw = (int)((gridDim.x * blockDim.x) >> 6);
  __shared__ float wl[32 * 65 + 32 * 33 + 64];
  for (int i = threadIdx.x; i < 32 * 65 + 32 * 33 + 64; i += 256) wl[i] = 0.001f * (i % 17);
  __syncthreads();
  float wd[32], we[16];
#pragma unroll
  for (int s = 0; s < 32; ++s) wd[s] = wl[li * 65 + lh + 2 * s];
#pragma unroll
  for (int r = 0; r < 16; ++r) we[r] = wl[2080 + li * 33 + mrow(r, lh)];
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  const int rowL = Lin * 4, rowO = Lo * 4;   // bytes per row (uniform)
  for (int tile = w0; tile < ntiles; tile += nw) {
    const int b = tile / tps, t0 = (tile - b * tps) * 32;          // scalar
    const int t = t0 + li;
    const bool ok = t < Lo;
    const int offx = ok ? (t + lh * dil) * 4 : (int)0x80000000;
    const int offr = ok ? (t + dil) * 4 + 4 * lh * rowL : (int)0x80000000;
    const int offo = ok ? t * 4 + 4 * lh * rowO : (int)0x80000000;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(s_in + (long)b * 32 * Lin), 0, 32 * rowL, 0x00020000);
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(s_out + (long)b * 32 * Lo), 0, 32 * rowO, 0x00020000);
    float x[32], rv[16];
    if (LOADS) {
#pragma unroll
      for (int c = 0; c < 32; ++c) x[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, offx, c * rowL, 0));
#pragma unroll
      for (int r = 0; r < 16; ++r) rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, offr, mrow(r, 0) * rowL, 0));
    } else {
#pragma unroll
      for (int c = 0; c < 32; ++c) x[c] = (float)(t + c) * 0.37f;
#pragma unroll
      for (int r = 0; r < 16; ++r) rv[r] = 0.f;
    }
    f32x16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (MFMA) {
#pragma unroll
      for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wd[s], relu_i(x[s]), acc, 0, 0, 0);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = x[2 * r] + x[2 * r + 1];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = rv[r];
    if (MFMA) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(we[r], relu_i(acc[r]), acc2, 0, 0, 0);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[r] += acc[r];
    }
    if (LOADS || acc2[0] == 12345.678f) {
#pragma unroll
      for (int r = 0; r < 16; ++r) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, acc2[r]), ro, offo, mrow(r, 0) * rowO, 0);
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) {
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    stamp[4 * w] = c1 - c0; stamp[4 * w + 1] = r1 - r0; stamp[4 * w + 2] = r0;
    stamp[4 * w + 3] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492);
  }
}
template <int LOADS, int MFMA, int OCC>
__global__ void __launch_bounds__(256, OCC) fwdclk3(const float* __restrict__ s_in, float* __restrict__ s_out, unsigned long long* __restrict__ stamp, int B, int Lin, int dil) {
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int Lo = Lin - dil, tps = (Lo + 31) >> 5;
  const int ntiles = B * tps;
  const int w0 = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6)), nw = (int)((gridDim.x * blockDim.x) >> 6);
  __shared__ float wl[32 * 65 + 32 * 33 + 64];
  for (int i = threadIdx.x; i < 32 * 65 + 32 * 33 + 64; i += 256) wl[i] = 0.001f * (i % 17);
  __syncthreads();
  const float* wd = wl + li * 65 + lh;
  const float* we = wl + 2080 + li * 33 + 4 * lh;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  const int rowL = Lin * 4, rowO = Lo * 4;   // bytes per row (uniform)
  int first = w0, last = ntiles, stride = nw;
  if ((gridDim.x & 7) == 0) {
    const int xcd = blockIdx.x & 7;
    first = (int)((long)ntiles * xcd / 8) + ((blockIdx.x >> 3) * 4 + (int)(threadIdx.x >> 6));
    last = (int)((long)ntiles * (xcd + 1) / 8);
    stride = (gridDim.x >> 3) * 4;
  }
  first = __builtin_amdgcn_readfirstlane(first);
  for (int tile = first; tile < last; tile += stride) {
    const int b = tile / tps, t0 = (tile - b * tps) * 32;          // scalar
    const int t = t0 + li;
    const bool ok = t < Lo;
    const int offx = ok ? (t + lh * dil) * 4 : (int)0x80000000;
    const int offr = ok ? (t + dil) * 4 + 4 * lh * rowL : (int)0x80000000;
    const int offo = ok ? t * 4 + 4 * lh * rowO : (int)0x80000000;
    __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(s_in + (long)b * 32 * Lin), 0, 32 * rowL, 0x00020000);
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(s_out + (long)b * 32 * Lo), 0, 32 * rowO, 0x00020000);
    float x[32], rv[16];
    if (LOADS) {
#pragma unroll
      for (int c = 0; c < 32; ++c) x[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, offx, c * rowL, 0));
#pragma unroll
      for (int r = 0; r < 16; ++r) rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, offr, mrow(r, 0) * rowL, 0));
    } else {
#pragma unroll
      for (int c = 0; c < 32; ++c) x[c] = (float)(t + c) * 0.37f;
#pragma unroll
      for (int r = 0; r < 16; ++r) rv[r] = 0.f;
    }
    f32x16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (MFMA) {
#pragma unroll
      for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wd[2 * s], relu_i(x[s]), acc, 0, 0, 0);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = x[2 * r] + x[2 * r + 1];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = rv[r];
    if (MFMA) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(we[mrow(r, 0)], relu_i(acc[r]), acc2, 0, 0, 0);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[r] += acc[r];
    }
    if (LOADS || acc2[0] == 12345.678f) {
#pragma unroll
      for (int r = 0; r < 16; ++r) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, acc2[r]), ro, offo, mrow(r, 0) * rowO, 0);
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) {
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    stamp[4 * w] = c1 - c0; stamp[4 * w + 1] = r1 - r0; stamp[4 * w + 2] = r0;
    stamp[4 * w + 3] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492);
  }
}
#define LAUNCHC2(n, L, M) extern "C" void launch_##n(const float* a, float* b, unsigned long long* st, int B, int Lin, int dil, int blocks, hipStream_t s) { hipLaunchKernelGGL((fwdclk2<L, M>), dim3(blocks), dim3(256), 0, s, a, b, st, B, Lin, dil); }
LAUNCHC2(clk2_full, 1, 1) LAUNCHC2(clk2_mfmaonly, 0, 1) LAUNCHC2(clk2_memonly, 1, 0)

#define LAUNCHC3(n, L, M, O) extern "C" void launch_##n(const float* a, float* b, unsigned long long* st, int B, int Lin, int dil, int blocks, hipStream_t s) { hipLaunchKernelGGL((fwdclk3<L, M, O>), dim3(blocks), dim3(256), 0, s, a, b, st, B, Lin, dil); }
LAUNCHC3(clk3_occ2, 1, 1, 2) LAUNCHC3(clk3_occ3, 1, 1, 3) LAUNCHC3(clk3_occ4, 1, 1, 4)
