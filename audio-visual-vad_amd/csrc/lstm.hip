// lstm.hip -- packed-sequence LSTM layer, forward and backward (BPTT).
//
// Replaces pack_padded_sequence -> nn.LSTM -> pad_packed_sequence(total_length=T):
//   packages/models/Audio_Net.py:50-56, Video_Net.py:102-113, AV_Net.py:127-137.
// Packed semantics as masking: a sequence's state stops at its length, padded
// output steps are zero (so h_{t-1} can simply be read back from y[:, t-1]).
//
// Structure per layer: one big input-projection GEMM over all (b,t) rows, then
// per step a small recurrent GEMM (stream-K over the long K, accumulated onto the
// pre-activations) + one fused gate kernel.  The activated gates overwrite the
// pre-activations in the workspace and are what backward consumes.
#include "gemm_api.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void add_bias2(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = a[i] + b[i];
}

// gates G[b][t][4H] (pre-activation in, activated out), cell Cs[b][t][H], y[b][t][H]
__global__ void lstm_gates_fwd(float* __restrict__ G, float* __restrict__ Cs, float* __restrict__ y,
                               const int* __restrict__ lengths, int B, int T, int H, int t) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * H) return;
  const int b = idx / H, j = idx - b * H;
  float* g = G + ((long)b * T + t) * 4 * H;
  const long o = ((long)b * T + t) * H + j;
  if (t >= lengths[b]) {
    g[j] = 0.f; g[H + j] = 0.f; g[2 * H + j] = 0.f; g[3 * H + j] = 0.f;
    Cs[o] = 0.f; y[o] = 0.f;
    return;
  }
  const float ig = sigmoidf_(g[j]), fg = sigmoidf_(g[H + j]), gg = tanhf(g[2 * H + j]), og = sigmoidf_(g[3 * H + j]);
  const float cp = t > 0 ? Cs[o - H] : 0.f;
  const float c = fg * cp + ig * gg;
  g[j] = ig; g[H + j] = fg; g[2 * H + j] = gg; g[3 * H + j] = og;
  Cs[o] = c;
  y[o] = og * tanhf(c);
}

// ------------------------------------------------------------------ fused recurrent step (t >= 1), B % 16 == 0, H % 16 == 0
// One launch does what the stream-K GEMM (M = B rows only: 64 x 4096 x 1024, atomics onto the pre-activations) plus
// lstm_gates_fwd did in two.  Workgroup blk owns hidden units 4*blk .. 4*blk+3, i.e. the 16 gate rows
// {gate*H + 4*blk + u}; wave w owns sequences 16w .. 16w+15.  v_mfma_f32_16x16x4_f32 with D[16 gate rows][16 sequences]:
// lane (i = l&15, q = l>>4) feeds A = W_hh[row(i)][k] and B = h_{t-1}[b][k]; both are read as float4 along k straight
// from L2 (no LDS: nothing is shared between waves but the 64 KB weight slab, which L1 serves) and the 4 elements go to 4
// MFMAs -- the k-slot q of MFMA e stands for k = 16*kk + 4*q + e on both operands, a permutation of the contraction
// order only.  Row order 4*u + gate puts the four gate sums of unit q in the four accumulator registers of lane (b, q),
// so the gate arithmetic (identical to lstm_gates_fwd) runs in registers.
// The workgroup always runs 16 waves: B/16 sequence blocks x KS = 16/(B/16) K-slices (the loads are L2-latency-bound, so
// a wave keeps 16 float4 loads in flight over a short slice instead of walking all of K); the slices' partial sums meet
// in LDS and the slice-0 waves finish the step.
// (Round 1 tried ONE persistent launch per layer with agent-scope release/acquire fences around a grid barrier: correct,
//  0.4 ms/step SLOWER -- the fences write back and invalidate whole L2s every step.  Round 2's lstm_persistent_fwd below
//  hands h_t over with write-through stores and a flag barrier instead and is the default where the shape allows; this
//  kernel serves the other shapes and option lstm_no_persistent.  Also tried: the same
//  skinny treatment for the backward product dh = dG W_hh (64 columns x a K-split per workgroup, slabs summed by the gate
//  kernel): equal to the stream-K GEMM within 0.1 ms -- both stream all of W_hh every step.  Round 2 tried a fused backward
//  step in this kernel's image -- W_hh transposed once, workgroup = 16 units x 64 sequences, dh in the accumulators, gate
//  backward of step t-1 in registers: correct, one launch instead of three, but only H/16 = 64 workgroups each streaming
//  1.25 MB of gate gradients: 1.3 ms/step SLOWER at the bench shape, 2.5 ms at C2.)
typedef float f32x4v __attribute__((ext_vector_type(4)));
// blockIdx.y selects a group of BG = min(B, 64) sequences, blockIdx.x a run of 4*UB hidden units.  At B = 256 (BASELINE
// config 1) the step is bound by L2 traffic: every workgroup re-reads h_{t-1} of its 64 sequences (256 KB) next to its
// weight slab (64 KB per 4 units), 328 MB per step with UB = 1 -- 72 us.  UB = 4 shares one read of h between 16 units:
// 131 MB per step.
template <int UB>
__global__ void __launch_bounds__(1024)
    lstm_step_fwd_mfma(float* __restrict__ G, float* __restrict__ Cs, float* __restrict__ y, const float* __restrict__ w_hh,
                       const int* __restrict__ lengths, int B, int T, int H, int t) {
  __shared__ f32x4v part[UB][16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int BG = B < 64 ? B : 64;                            // sequences per workgroup
  const int nb = BG >> 4, KS = 16 / nb;
  const int wb = wave % nb, ks = wave / nb;
  const int i = lane & 15, q = lane >> 4;
  const int blk = blockIdx.x;
  const int b = blockIdx.y * BG + wb * 16 + i;               // this lane's sequence (B operand column / D column)
  const int kspan = H / KS;                                  // host guarantees H % (16*KS) == 0
  // A row i = 4*u + gate of unit-quad ub: W_hh row gate*H + 4*(blk*UB + ub) + u
  const float* wrow = w_hh + (long)((i & 3) * H + 4 * blk * UB + (i >> 2)) * H + ks * kspan + 4 * q;
  const float* hrow = y + ((long)b * T + (t - 1)) * H + ks * kspan + 4 * q;
  f32x4v acc[UB];
#pragma unroll
  for (int ub = 0; ub < UB; ++ub) acc[ub] = f32x4v{0.f, 0.f, 0.f, 0.f};
  const int nk = kspan >> 4;
  for (int k0 = 0; k0 < nk; k0 += 8) {
    float4 h[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int kk = (k0 + u < nk) ? k0 + u : nk - 1;       // clamped (nk % 8 != 0 shapes); masked below
      h[u] = *reinterpret_cast<const float4*>(hrow + 16 * kk);
    }
#pragma unroll
    for (int ub = 0; ub < UB; ++ub) {
      float4 a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int kk = (k0 + u < nk) ? k0 + u : nk - 1;
        a[u] = *reinterpret_cast<const float4*>(wrow + (long)ub * 4 * H + 16 * kk);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (k0 + u < nk) {
          acc[ub] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, h[u].x, acc[ub], 0, 0, 0);
          acc[ub] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, h[u].y, acc[ub], 0, 0, 0);
          acc[ub] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, h[u].z, acc[ub], 0, 0, 0);
          acc[ub] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, h[u].w, acc[ub], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int ub = 0; ub < UB; ++ub) part[ub][wave][lane] = acc[ub];
  __syncthreads();
  if (ks != 0) return;
  const bool live = t < lengths[b];
#pragma unroll
  for (int ub = 0; ub < UB; ++ub) {
    f32x4v r = acc[ub];
    for (int s2 = 1; s2 < KS; ++s2) {
      const f32x4v p = part[ub][s2 * nb + wb][lane];
      r[0] += p[0]; r[1] += p[1]; r[2] += p[2]; r[3] += p[3];
    }
    // r[gate] = recurrent part of the gate of unit j for sequence b
    const int j = 4 * (blk * UB + ub) + q;
    float* g = G + ((long)b * T + t) * 4 * H;
    const long o = ((long)b * T + t) * H + j;
    if (!live) {
      g[j] = 0.f; g[H + j] = 0.f; g[2 * H + j] = 0.f; g[3 * H + j] = 0.f;
      Cs[o] = 0.f; y[o] = 0.f;
      continue;
    }
    const float ig = sigmoidf_(g[j] + r[0]), fg = sigmoidf_(g[H + j] + r[1]), gg = tanhf(g[2 * H + j] + r[2]),
                og = sigmoidf_(g[3 * H + j] + r[3]);
    const float cp = Cs[o - H];
    const float c = fg * cp + ig * gg;
    g[j] = ig; g[H + j] = fg; g[2 * H + j] = gg; g[3 * H + j] = og;
    Cs[o] = c;
    y[o] = og * tanhf(c);
  }
}

// ------------------------------------------------------------------ persistent forward: ONE launch for steps 1 .. T-1
// Grid = H/4 workgroups of 4 waves, all resident (a workgroup needs ~150 registers per lane and 4 KB of LDS: the chip
// holds several per CU).  Workgroup blk owns hidden units 4 blk .. 4 blk + 3 for every sequence (B <= 64); wave w holds
// the K quarter w of its 16 gate rows of W_hh as MFMA A fragments IN REGISTERS for the whole launch (NK float4 per lane),
// so W_hh is read once per layer instead of once per step, and a step costs no launch.
// What carries h_t between workgroups (per-XCD L2s are not coherent): h_t is stored with sc1 (write-through) stores into
// one of TWO padded hand-off copies (re-used every second step) and read with sc1 loads.  Correctness of the re-use rests
// on the sc1 forms: an sc1 store leaves no line behind in the storing XCD's L2 and an sc1 load bypasses the reading CU's
// L1 and is served from beyond the XCD's L2 for lines that L2 does not hold -- no L2 ever holds a copy of the hand-off
// buffers, because every access to them in this kernel is an sc1 access (MI355X_MICROARCH "stores of each flavour") -- and the
// step barrier is one agent-scope atomic add per workgroup (behind s_waitcnt vmcnt(0) of every storing wave and a workgroup
// barrier) polled by one lane with sc1 loads: MI355X_MICROARCH "Valid forms", first table row.  No agent-scope fences: the
// release/acquire fences of round 1's persistent attempt wrote back and invalidated whole L2s every step and lost 0.4 ms.
// (Tried: 8 waves per workgroup, each a K eighth, held to 128 registers so that two workgroups still fit a CU: 18 spills and
//  the stack forward went from 0.68 to 0.73 ms.)
// Every spin is BOUNDED: if the grid ever failed to be co-resident the launch ends (status word set, results wrong and
// loudly so) instead of hanging the device.
template <int NK>   // H / 64: float4 fragments per lane and K quarter
__global__ void __launch_bounds__(256)
    lstm_persistent_fwd(float* __restrict__ G, float* __restrict__ Cs, float* __restrict__ y, const float* __restrict__ w_hh,
                        const int* __restrict__ lengths, int B, int T, int H, unsigned* __restrict__ sync) {
  __shared__ f32x4v part[4][4][64];      // [sequence group][K quarter][lane]
  __shared__ int give_up;
  const int lane = threadIdx.x & 63, ks = threadIdx.x >> 6;
  const int i = lane & 15, q = lane >> 4;
  const int blk = blockIdx.x, nblk = gridDim.x;      // nblk <= 256: one 16-byte poll per lane covers every flag
  const int nsg = B >> 4;
  const int kspan = H >> 2;
  if (threadIdx.x == 0) give_up = 0;
  // resident A fragments: row i = 4 u + gate  <->  W_hh row gate*H + 4 blk + u
  f32x4v a[NK];
  {
    const float* wrow = w_hh + (long)((i & 3) * H + 4 * blk + (i >> 2)) * H + ks * kspan + 4 * q;
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
      const float4 v = *reinterpret_cast<const float4*>(wrow + 16 * kk);
      a[kk] = f32x4v{v.x, v.y, v.z, v.w};
    }
  }
  // the finishing wave of sequence group sg is wave sg: lane (i, q) owns (sequence 16 sg + i, unit 4 blk + q)
  const int bfin = ks * 16 + i;
  const bool fin = ks < nsg;
  const int j = 4 * blk + q;
  float cprev = fin ? Cs[((long)bfin * T + 0) * H + j] : 0.f;     // c_0 (written by lstm_gates_fwd before this launch)
  const int len = fin ? lengths[bfin] : 0;
  const __amdgpu_buffer_rsrc_t ry = brsrc2g(y);
  const __amdgpu_buffer_rsrc_t rf = brsrc(reinterpret_cast<const float*>(sync + 64), nblk * 4);   // one flag word per workgroup
  // The hand-off copy of h: hb[t & 1][b][H + 32].  In y a sequence's rows are T*H*4 bytes apart (64 KB at the bench shape):
  // the 16 sequences of every load instruction, and every workgroup of an XCD at the same moment, fell on ONE L2 channel
  // (the h reads + MFMAs of a step took 8.9 us for 3.4 us of MFMA).  Padded rows spread them over the channels.
  const int LDHB = H + 32;
  float* const hb = reinterpret_cast<float*>(sync + 1024);
  const __amdgpu_buffer_rsrc_t rh = brsrc(hb, 2 * B * LDHB * 4);
  // h_{t-1} of sequence group sg, this wave's K quarter: NK 16-byte sc1 loads (step 1 reads y[:, 0], written before the launch)
  auto hload = [&](int sg, int t, f32x4v (&h)[NK]) {
    if (t == 1) {
      const int off = (int)((((long)(sg * 16 + i) * T) * H + ks * kspan + 4 * q) * 4);
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) h[kk] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(ry, off, 64 * kk, 16));
    } else {
      const int off = ((((t - 1) & 1) * B + sg * 16 + i) * LDHB + ks * kspan + 4 * q) * 4;
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) h[kk] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rh, off, 64 * kk, 16));
    }
  };
  auto hmul = [&](const f32x4v (&h)[NK], int sg) {
    // two accumulators: a dependent chain of this instruction issues every 40 cycles, independent ones every 32
    f32x4v acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk][0], h[kk][0], acc, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk][1], h[kk][1], acc1, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk][2], h[kk][2], acc, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk][3], h[kk][3], acc1, 0, 0, 0);
    }
    acc[0] += acc1[0]; acc[1] += acc1[1]; acc[2] += acc1[2]; acc[3] += acc1[3];
    part[sg][ks][lane] = acc;
  };
  __syncthreads();
#ifdef AVVAD_LSTM_PROF
  unsigned long long tA = 0, tB = 0, tC = 0;
#endif
  for (int t = 1; t < T; ++t) {
#ifdef AVVAD_LSTM_PROF
    const unsigned long long s0 = __builtin_amdgcn_s_memrealtime();
#endif
    float gin[4] = {0.f, 0.f, 0.f, 0.f};
    float* g = G + ((long)bfin * T + t) * 4 * H;
    // two register sets: group sg+1's requests are in flight under group sg's MFMAs (one wave per SIMD: nothing else
    // would hide them); sched_barriers keep hipcc from pairing each load with its four MFMAs
    f32x4v h0[NK], h1[NK];
    hload(0, t, h0);
    __builtin_amdgcn_sched_barrier(0);
    for (int sg = 0; sg < nsg; sg += 2) {
      if (sg + 1 < nsg) hload(sg + 1, t, h1);
      else if (fin) {       // last group: this step's pre-activations ride under its MFMAs
#pragma unroll
        for (int e = 0; e < 4; ++e) gin[e] = g[e * H + j];
      }
      __builtin_amdgcn_sched_barrier(0);
      hmul(h0, sg);
      if (sg + 1 >= nsg) break;
      if (sg + 2 < nsg) hload(sg + 2, t, h0);
      else if (fin) {
#pragma unroll
        for (int e = 0; e < 4; ++e) gin[e] = g[e * H + j];
      }
      __builtin_amdgcn_sched_barrier(0);
      hmul(h1, sg + 1);
    }
#ifdef AVVAD_LSTM_PROF
    const unsigned long long s1 = __builtin_amdgcn_s_memrealtime();
#endif
    __syncthreads();
    if (fin) {
      f32x4v r = part[ks][0][lane];
#pragma unroll
      for (int s2 = 1; s2 < 4; ++s2) {
        const f32x4v pp = part[ks][s2][lane];
        r[0] += pp[0]; r[1] += pp[1]; r[2] += pp[2]; r[3] += pp[3];
      }
      const long o = ((long)bfin * T + t) * H + j;
      float hv = 0.f;
      if (t < len) {
        const float ig = sigmoidf_(gin[0] + r[0]), fg = sigmoidf_(gin[1] + r[1]), gg = tanhf(gin[2] + r[2]),
                    og = sigmoidf_(gin[3] + r[3]);
        const float c = fg * cprev + ig * gg;
        hv = og * tanhf(c);
        // h_t first (sc1: for every workgroup), then the layer's output and the state the backward reads
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, hv), rh, (((t & 1) * B + bfin) * LDHB + j) * 4, 0, 16);
        y[o] = hv;
        g[j] = ig; g[H + j] = fg; g[2 * H + j] = gg; g[3 * H + j] = og;
        Cs[o] = c;
        cprev = c;
      } else {
        __builtin_amdgcn_raw_buffer_store_b32(0, rh, (((t & 1) * B + bfin) * LDHB + j) * 4, 0, 16);
        y[o] = 0.f;
        g[j] = 0.f; g[H + j] = 0.f; g[2 * H + j] = 0.f; g[3 * H + j] = 0.f;
        Cs[o] = 0.f;
        cprev = 0.f;
      }
    }
    if (t + 1 == T) break;
    // ---- step barrier: every storing wave drains, the workgroup meets, lane 0 publishes flag[blk] = t (sc1 store, no
    // atomics: 256 adds onto one counter serialise, 6.6 us per step), wave 0 polls ALL flags with one 16-byte sc1 load per
    // lane and joins the workgroup barrier only once every flag shows this step
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef AVVAD_LSTM_PROF
    const unsigned long long s2 = __builtin_amdgcn_s_memrealtime();
#endif
    if (ks == 0 && !give_up) {
      if (lane == 0) __builtin_amdgcn_raw_buffer_store_b32(t, rf, blk * 4, 0, 16);
      int spins = 0;
      for (;;) {
        // lanes past the last flag read out of range = 0: treated as arrived
        const i4v f = __builtin_amdgcn_raw_buffer_load_b128(rf, lane * 16, 0, 16);
        const bool mine = (lane * 4 + 0 >= nblk || f[0] >= t) && (lane * 4 + 1 >= nblk || f[1] >= t) &&
                          (lane * 4 + 2 >= nblk || f[2] >= t) && (lane * 4 + 3 >= nblk || f[3] >= t);
        if (__all(mine)) break;
        if (++spins > (1 << 21)) { if (lane == 0) { give_up = 1; sync[1] = 1u; } break; }       // never hang the device
      }
    }
    __syncthreads();
#ifdef AVVAD_LSTM_PROF
    const unsigned long long s3 = __builtin_amdgcn_s_memrealtime();
    tA += s1 - s0; tB += s2 - s1; tC += s3 - s2;
#endif
  }
#ifdef AVVAD_LSTM_PROF
  if (threadIdx.x == 0 && blk < 8) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(sync + 16) + 3 * blk;
    o[0] = tA; o[1] = tB; o[2] = tC;
  }
#endif
}

// Workgroups of lstm_persistent_fwd<NK> the device can hold at once, by the occupancy query (one block per CU taken off
// where it reports more than one: on ROCm 7.2 the query over-reports by one for SGPR-heavy 256-thread kernels,
// MI355X_MICROARCH "Residency and cooperative launch").  The grid barrier needs the whole grid resident; a grid beyond
// this falls back to the per-step kernels.  What the query cannot see is OTHER work on the device (a second process, RCCL's
// kernels, the trunk's persistent grids on another stream): their workgroups retire on their own -- nothing they wait for
// is behind this launch -- so the grid still becomes resident, late; the spins are bounded all the same.
static int persistent_capacity(int NKp) {
  static int cap[3] = {-1, -1, -1};
  const int i = NKp == 16 ? 0 : (NKp == 8 ? 1 : 2);
  if (cap[i] < 0) {
    int nb = 0, dev = 0, cus = 0;
    hipError_t e = NKp == 16 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, lstm_persistent_fwd<16>, 256, 0)
                   : NKp == 8 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, lstm_persistent_fwd<8>, 256, 0)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, lstm_persistent_fwd<4>, 256, 0);
    if (e != hipSuccess || hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    cap[i] = (nb > 1 ? nb - 1 : nb) * cus;
  }
  return cap[i];
}

// behind the persistent launch: had a step barrier ever timed out (status word set), the layer's output is poisoned
// with NaNs -- a wrong result must not pass for a right one
__global__ void lstm_persistent_check(const unsigned* __restrict__ sync, float* __restrict__ y, long n) {
  if (sync[1] == 0u) return;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = __builtin_nanf("");
}

// in: activated gates in G, dy[b][t], DH (recurrent dh from step t+1, consumed and zeroed), DC (dc from t+1)
// out: d(pre-activation gates) in G (in place), DC for step t-1
__global__ void lstm_gates_bwd(float* __restrict__ G, const float* __restrict__ Cs, const float* __restrict__ dy,
                               float* __restrict__ DH, float* __restrict__ DC, const int* __restrict__ lengths, int B,
                               int T, int H, int t) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * H) return;
  const int b = idx / H, j = idx - b * H;
  float* g = G + ((long)b * T + t) * 4 * H;
  const long o = ((long)b * T + t) * H + j;
  const float dhr = DH[idx];
  DH[idx] = 0.f;
  if (t >= lengths[b]) {
    g[j] = 0.f; g[H + j] = 0.f; g[2 * H + j] = 0.f; g[3 * H + j] = 0.f;
    DC[idx] = 0.f;
    return;
  }
  const float ig = g[j], fg = g[H + j], gg = g[2 * H + j], og = g[3 * H + j];
  const float c = Cs[o], cp = t > 0 ? Cs[o - H] : 0.f;
  const float tc = tanhf(c);
  const float dh = dy[o] + dhr;
  const float dc = DC[idx] + dh * og * (1.f - tc * tc);
  g[j] = dc * gg * ig * (1.f - ig);
  g[H + j] = dc * cp * fg * (1.f - fg);
  g[2 * H + j] = dc * ig * (1.f - gg * gg);
  g[3 * H + j] = dh * tc * og * (1.f - og);
  DC[idx] = dc * fg;
}

// Epilogue of the backward recurrent product DH = dG_t . W_hh (rows = sequences, columns = hidden units): the engine's
// fix-up kernel, which is where a split tile's total exists, hands each finished dh to finish4(), and that runs the gate
// backward of step t - 1 on the spot (lstm_gates_bwd's arithmetic) -- two launches per time step instead of three.
struct EpiLstmBwd {
  static constexpr bool PLAIN = true;
  float* C;            // DH [B][H]
  long ldc;
  const float* bias;   // null
  int mode;            // 0: the first piece of a tile overwrites DH
  int cs;
  int active;          // set by igemm::launch(): every tile goes through the fix-up
  float* G;
  const float* Cs;
  const float* dy;
  float* DC;
  const int* lengths;
  int T, H, t;         // t: the step whose gates are differentiated (the product's step minus one)
  __device__ __forceinline__ float* ptr(int m, int n) const { return C + (long)m * ldc + n; }
  __device__ __forceinline__ void finish4(int b, int j0, const float (&dhr)[4], int N) const {
    const bool live = t < lengths[b];
    float* g = G + ((long)b * T + t) * 4 * H;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = j0 + e;
      if (j >= N) continue;
      const int idx = b * H + j;
      const long o = ((long)b * T + t) * H + j;
      if (!live) {
        g[j] = 0.f; g[H + j] = 0.f; g[2 * H + j] = 0.f; g[3 * H + j] = 0.f;
        DC[idx] = 0.f;
        continue;
      }
      const float ig = g[j], fg = g[H + j], gg = g[2 * H + j], og = g[3 * H + j];
      const float c = Cs[o], cp = t > 0 ? Cs[o - H] : 0.f;
      const float tc = tanhf(c);
      const float dh = dy[o] + dhr[e];
      const float dc = DC[idx] + dh * og * (1.f - tc * tc);
      g[j] = dc * gg * ig * (1.f - ig);
      g[H + j] = dc * cp * fg * (1.f - fg);
      g[2 * H + j] = dc * ig * (1.f - gg * gg);
      g[3 * H + j] = dh * tc * og * (1.f - og);
      DC[idx] = dc * fg;
    }
  }
};

__global__ void shift_time(const float* __restrict__ y, float* __restrict__ ys, int B, int T, int H) {
  const long n = (long)B * T * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)((i / H) % T);
    ys[i] = t > 0 ? y[i - H] : 0.f;
  }
}

__global__ void fill0(float* p, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = 0.f;
}

// out[c] += sum_r X[r][c]  (and out2 if given): bias gradients over all (sequence, step) rows.  Two deterministic stages:
// (column quad, row chunk) workgroups leave partial sums, a second kernel adds the chunks in order.  (The first version
// walked all rows with cols/64 workgroups: 1.4 ms per call at the C2 shape, 15 360 rows x 4096 columns.)
constexpr int CS_CHUNKS = 64;
__global__ void __launch_bounds__(256)
    colsum_stage1(const float* __restrict__ X, int rows, int cols, float* __restrict__ part) {
  __shared__ float4 sm[4][64];
  const int q = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + q) * 4;
  const int per = (rows + CS_CHUNKS - 1) / CS_CHUNKS;
  const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < cols)
    for (int r = r0 + rl; r < r1; r += 4) {
      const float4 v = *reinterpret_cast<const float4*>(X + (long)r * cols + c);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  sm[rl][q] = s;
  __syncthreads();
  if (rl == 0 && c < cols) {
    float4 t = sm[0][q];
#pragma unroll
    for (int k = 1; k < 4; ++k) { t.x += sm[k][q].x; t.y += sm[k][q].y; t.z += sm[k][q].z; t.w += sm[k][q].w; }
    *reinterpret_cast<float4*>(part + (long)blockIdx.y * cols + c) = t;
  }
}
__global__ void colsum_stage2(const float* __restrict__ part, int cols, float* __restrict__ out, float* __restrict__ out2) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float s = 0.f;
  for (int k = 0; k < CS_CHUNKS; ++k) s += part[(long)k * cols + c];
  out[c] += s;
  if (out2) out2[c] += s;
}

struct Ws {
  float *G, *Cs, *bias, *DH, *DC, *Ys, *slab;
  size_t total;
};
static Ws carve(const avvad_lstm_desc* d, float* base) {
  Ws w;
  size_t off = 0;
  auto take = [&](size_t n) { size_t o = off; off += align_up(n, 64); return base ? base + o : (float*)nullptr; };
  const size_t B = d->B, T = d->T, H = d->H;
  w.G = take(B * T * 4 * H);
  w.Cs = take(B * T * H);
  w.bias = take(4 * H);
  w.DH = take(B * H);
  w.DC = take(B * H);
  w.Ys = take(B * T * H);
  w.slab = take(igemm::SLAB_FLOATS);      // partial tiles of the GEMM engine's stream-K round
  w.total = off;
  return w;
}
static inline int grid1(long n) { long b = (n + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }
static inline int pick_split(int M, int N, int K) {
  const int tiles = cdiv(M, (M <= 64 || N <= 64) ? 64 : 128) * cdiv(N, (M <= 64 || N <= 64) ? 64 : 128);
  int s = 512 / (tiles > 0 ? tiles : 1);
  const int kt = cdiv(K, igemm::BK);
  if (s > kt) s = kt;
  if (s > 16) s = 16;
  return s < 1 ? 1 : s;
}

}  // namespace

extern "C" size_t avvad_lstm_workspace(const avvad_lstm_desc* d) {
  if (!d || d->B <= 0 || d->T <= 0 || d->H <= 0) return 0;
  return carve(d, nullptr).total * sizeof(float);
}

extern "C" int avvad_lstm_layer_fwd(const float* x, const float* w_ih, const float* w_hh, const float* b_ih,
                                    const float* b_hh, float* y, const avvad_lstm_desc* d, void* wsv, size_t ws_bytes,
                                    avvad_stream_t sv) {
  AVVAD_ENTER();
  if (!x || !w_ih || !w_hh || !b_ih || !b_hh || !y || !d || !wsv || !d->lengths || d->B <= 0 || d->T <= 0 || d->In <= 0 ||
      d->H <= 0)
    return AVVAD_EINVAL;
  hipStream_t s = (hipStream_t)sv;
  Ws w = carve(d, (float*)wsv);
  if (ws_bytes < w.total * sizeof(float)) return AVVAD_EWORKSPACE;
  const int B = d->B, T = d->T, H = d->H, In = d->In;
  hipLaunchKernelGGL(add_bias2, dim3(cdiv(4 * H, 256)), dim3(256), 0, s, b_ih, b_hh, w.bias, 4 * H);
  int rc;
  avvad_gemm_desc gd = gemm_desc(B * T, 4 * H, In, In, In, 4 * H, 0, 1, 0, 1);
  if ((rc = avvad_gemm_impl(x, w_ih, w.bias, w.G, &gd, s, w.slab))) return rc;
  const int split = pick_split(B, 4 * H, H);
  // sequences are handled in groups of BG = min(B, 64) (blockIdx.y); a group's 16 waves are BG/16 sequence blocks x KS K-slices
  const int BG = B < 64 ? B : 64;
  const bool fused_step = (B == 16 || B == 32 || (B % 64 == 0 && B <= 65535 * 64)) && (H % (16 * (256 / BG)) == 0) &&
                          !avvad_tune().lstm_no_fused_step;
  // ONE persistent launch for steps 1 .. T-1 (option lstm_no_persistent: the per-step kernels)
  const int NKp = H / 64;
  const bool persistent = fused_step && !avvad_tune().lstm_no_persistent && B <= 64 && B % 16 == 0 && T > 1 &&
                          (NKp == 16 || NKp == 8 || NKp == 4) && H % 64 == 0 && H / 4 <= 256 &&
                          (long)B * T * H * 4 < (1L << 31);
  if (persistent && H / 4 <= persistent_capacity(NKp)) {
    hipLaunchKernelGGL(lstm_gates_fwd, dim3(cdiv(B * H, 256)), dim3(256), 0, s, w.G, w.Cs, y, d->lengths, B, T, H, 0);
    unsigned* sync = reinterpret_cast<unsigned*>(w.slab);          // the engine scratch is idle during the time loop
    if (hipMemsetAsync(sync, 0, 2048, s) != hipSuccess) return AVVAD_ELAUNCH;   // status words + one flag per workgroup
    // (behind them, from word 1024: the two padded hand-off copies of h, 2 x B x (H + 32) floats)
    if (NKp == 16) hipLaunchKernelGGL(lstm_persistent_fwd<16>, dim3(H / 4), dim3(256), 0, s, w.G, w.Cs, y, w_hh, d->lengths, B, T, H, sync);
    else if (NKp == 8) hipLaunchKernelGGL(lstm_persistent_fwd<8>, dim3(H / 4), dim3(256), 0, s, w.G, w.Cs, y, w_hh, d->lengths, B, T, H, sync);
    else hipLaunchKernelGGL(lstm_persistent_fwd<4>, dim3(H / 4), dim3(256), 0, s, w.G, w.Cs, y, w_hh, d->lengths, B, T, H, sync);
    hipLaunchKernelGGL(lstm_persistent_check, dim3(64), dim3(256), 0, s, sync, y, (long)B * T * H);
    AVVAD_LAUNCH_CHECK();
    return AVVAD_OK;
  }
  for (int t = 0; t < T; ++t) {
    if (t > 0 && fused_step) {
      if (B >= 128 && H % 16 == 0)        // many sequences: one read of h_{t-1} serves 16 hidden units
        hipLaunchKernelGGL(lstm_step_fwd_mfma<4>, dim3(H / 16, B / BG), dim3(1024), 0, s, w.G, w.Cs, y, w_hh, d->lengths, B, T, H, t);
      else
        hipLaunchKernelGGL(lstm_step_fwd_mfma<1>, dim3(H / 4, B / BG), dim3(1024), 0, s, w.G, w.Cs, y, w_hh, d->lengths, B, T, H, t);
      continue;
    }
    if (t > 0) {
      avvad_gemm_desc rd = gemm_desc(B, 4 * H, H, T * H, H, T * 4 * H, 0, 1, 1, split);
      if ((rc = avvad_gemm_impl(y + (long)(t - 1) * H, w_hh, nullptr, w.G + (long)t * 4 * H, &rd, s, w.slab))) return rc;
    }
    hipLaunchKernelGGL(lstm_gates_fwd, dim3(cdiv(B * H, 256)), dim3(256), 0, s, w.G, w.Cs, y, d->lengths, B, T, H, t);
  }
  AVVAD_LAUNCH_CHECK();
  return AVVAD_OK;
}

extern "C" int avvad_lstm_layer_bwd(const float* x, const float* w_ih, const float* w_hh, const float* y, const float* dy,
                                    float* dx, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh,
                                    const avvad_lstm_desc* d, void* wsv, size_t ws_bytes, avvad_stream_t sv) {
  AVVAD_ENTER();
  BwdCuCap cu_cap;
  if (!x || !w_ih || !w_hh || !y || !dy || !d || !wsv || !d->lengths) return AVVAD_EINVAL;
  hipStream_t s = (hipStream_t)sv;
  Ws w = carve(d, (float*)wsv);
  if (ws_bytes < w.total * sizeof(float)) return AVVAD_EWORKSPACE;
  const int B = d->B, T = d->T, H = d->H, In = d->In;
  int rc;
  hipLaunchKernelGGL(fill0, dim3(grid1(B * H)), dim3(256), 0, s, w.DH, (long)B * H);
  hipLaunchKernelGGL(fill0, dim3(grid1(B * H)), dim3(256), 0, s, w.DC, (long)B * H);
  const int split = pick_split(B, H, 4 * H);
  // fused form: the product's fix-up finishes step t - 1's gates (EpiLstmBwd); needs the 64x64 tile path with 16-byte rows
  const bool fuse = (B <= 64 || H <= 64) && H % 4 == 0 && ((uintptr_t)w_hh % 16 == 0) && !avvad_tune().lstm_no_fused_step &&
                    (long)B * T * 4 * H < (1L << 29) - 64;
  bool gates_done = false;     // step t's gate backward already ran inside the previous product's fix-up
  for (int t = T - 1; t >= 0; --t) {
    if (!gates_done)
      hipLaunchKernelGGL(lstm_gates_bwd, dim3(cdiv(B * H, 256)), dim3(256), 0, s, w.G, w.Cs, dy, w.DH, w.DC, d->lengths, B, T,
                         H, t);
    gates_done = false;
    if (t > 0) {  // DH = dG_t . W_hh   ([B][4H] x [4H][H])
      if (fuse) {
        igemm::RowVec4 a{w.G + (long)t * 4 * H, (long)T * 4 * H, B, 4 * H, 0};     // (4H % 4 == 0, rows 16-byte aligned)
        igemm::ColPlain<4> bop{w_hh, H, H, 4 * H, 0};
        EpiLstmBwd e{w.DH, H, nullptr, 0, 1, 0, w.G, w.Cs, dy, w.DC, d->lengths, T, H, t - 1};
        if ((rc = igemm::launch<64, 64>(a, bop, e, B, H, 4 * H, split, s, w.slab, true, &gates_done))) return rc;
      } else {
        avvad_gemm_desc rd = gemm_desc(B, H, 4 * H, T * 4 * H, H, H, 0, 0, 1, split);
        if ((rc = avvad_gemm_impl(w.G + (long)t * 4 * H, w_hh, nullptr, w.DH, &rd, s, w.slab))) return rc;
      }
    }
  }
  const int R = B * T;
  if (dx) {
    avvad_gemm_desc g1 = gemm_desc(R, In, 4 * H, 4 * H, In, In, 0, 0, 0, 1);
    if ((rc = avvad_gemm_impl(w.G, w_ih, nullptr, dx, &g1, s, w.slab))) return rc;
  }
  if (dw_ih) {
    avvad_gemm_desc g2 = gemm_desc(4 * H, In, R, 4 * H, In, In, 1, 0, 1, pick_split(4 * H, In, R));
    if ((rc = avvad_gemm_impl(w.G, x, nullptr, dw_ih, &g2, s, w.slab))) return rc;
  }
  if (dw_hh) {
    hipLaunchKernelGGL(shift_time, dim3(grid1((long)R * H)), dim3(256), 0, s, y, w.Ys, B, T, H);
    avvad_gemm_desc g3 = gemm_desc(4 * H, H, R, 4 * H, H, H, 1, 0, 1, pick_split(4 * H, H, R));
    if ((rc = avvad_gemm_impl(w.G, w.Ys, nullptr, dw_hh, &g3, s, w.slab))) return rc;
  }
  if (db_ih || db_hh) {          // (4H % 4 == 0; the engine scratch is free here and holds the CS_CHUNKS x 4H partial sums)
    hipLaunchKernelGGL(colsum_stage1, dim3(cdiv(4 * H, 256), CS_CHUNKS), dim3(256), 0, s, w.G, R, 4 * H, w.slab);
    hipLaunchKernelGGL(colsum_stage2, dim3(cdiv(4 * H, 256)), dim3(256), 0, s, w.slab, 4 * H, db_ih ? db_ih : db_hh,
                       (db_ih && db_hh) ? db_hh : (float*)nullptr);
  }
  AVVAD_LAUNCH_CHECK();
  return AVVAD_OK;
}
