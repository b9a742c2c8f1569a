// common.h -- shared device helpers for libavvad_hip.so (gfx950 / CDNA4 only)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "avvad.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#include <stdio.h>
#define AVVAD_LAUNCH_CHECK()                                                                      \
  do {                                                                                            \
    hipError_t e_ = hipGetLastError();                                                            \
    if (e_ != hipSuccess) {                                                                       \
      fprintf(stderr, "avvad: HIP error '%s' after launch at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return AVVAD_ELAUNCH;                                                                       \
    }                                                                                             \
  } while (0)

// hipGetLastError() is per host thread and also reports errors left behind by OTHER users of the runtime
// in this process (e.g. a not-ready event query inside PyTorch): clear it on entry so that the check after
// our launches only sees our launches.
#define AVVAD_ENTER() (void)hipGetLastError()

// Process-wide schedule options (tuning / debugging switches).  They are NOT read from the environment per launch: the
// table is filled once, on first use, from the AVVAD_* variables and can afterwards only be changed through the C ABI
// (avvad_set_option).  Defined in misc.hip.
struct AvvadTune {
  int igemm_variant;        // -1 default; 0 "db" 4-wave double LDS buffer, 1 "sb" single buffer, 2 "w8" 8 waves
  int no_streamk;           // 0 production; 1 whole-tile schedule for every GEMM; 10+m only for epilogue mode m
  int kmajor;               // K-major cells for the conv weight gradients
  int lstm_no_fused_step;   // per-step GEMM + gate kernel instead of the fused recurrent step
  int lstm_no_persistent;   // one launch per time step instead of the persistent recurrent kernel
  int no_stem_kernel;       // 7x7 stem on the engine instead of the LDS-resident frame kernel
  int no_tall;              // 128x64 tiles instead of 256x64 for the 64-channel convolutions
  int wn_no_fused_tail;     // unfused encoder tail backward
  int wn_no_tail_pair;      // fused encoder tail backward with one wave per time tile (4 waves / workgroup) instead of two
  int wn_no_fused_wgrad;    // unfused encoder block backward (dz, weight gradients as separate kernels)
  int no_fixup1;             // tuning aid: always the four-wave fix-up kernel
  int no_buf;                // convolution gathers with flat addressing + validity selects (the form operands >= 2 GiB use)
  int wn_flat;               // residual-block forward: 0 by size, 1 flat dword kernel, 2 dword buffer kernel, 3 wide (dwordx4) buffer kernel
  int wn_dx;                 // dx kernel: 0 by the descriptor's shared_device hint, 1 resident weights + cross-tile prefetch, 2 high occupancy, 3 round 1's flat kernel
  int wn_grid;               // tuning aid: workgroup cap of the wide residual-block kernels (0 = default)
  int wn_bwd_t;             // fused block backward: 0 by the descriptor's shared_device hint, 1 transposed products, 2 high occupancy, 3 resident weights
  int bf16;                 // bf16-input MFMA (fp32 accumulate) for the convolutions and dense GEMMs: BASELINE config 5's arithmetic
  int stagger;              // the engine's 8-wave kernels WITH the half-tile stagger of waves 4-7 (measured: step +0.14 ms; off)
  int cls_cap;              // tuning aid: most tiles a position-class product may have (0 = two per CU)
  int no_cls;               // 3x3 convolutions without the position-class schedule (zero padding multiplied like everything else)
  int no_fused_stats;       // BatchNorm batch statistics by the separate column-reduction pass instead of the convolution's epilogue
  int no_s2_cls;            // stride-2 data gradients as four accumulating parity-class launches instead of one position-class product
  int no_conv64;            // the 64 -> 64 channel 3x3 convolutions on the engine instead of the weights-stationary kernel (conv64.h)
  int bwd_max_cus;          // the same cap, applied to the BACKWARD entry points only (that is when the gradient all-reduce runs)
  int max_cus;              // cap on the CUs a persistent grid occupies (0 = all 256): leaves room for RCCL kernels
};
AvvadTune& avvad_tune();
// Backward entry points: while one is on the host's call stack the persistent grids leave CUs to RCCL (option bwd_max_cus).
// (The options table is process-wide and the host side of a process is single-threaded per device: DESIGN.md 5.)
struct BwdCuCap {
  int saved;
  BwdCuCap() : saved(avvad_tune().max_cus) {
    const int b = avvad_tune().bwd_max_cus;
    if (b > 0 && (saved <= 0 || b < saved)) avvad_tune().max_cus = b;
  }
  ~BwdCuCap() { avvad_tune().max_cus = saved; }
};

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// fp32-input MFMA 32x32x2: D(32x32) += A(32x2) . B(2x32).
//   lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
//   D element reg r of lane l is row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31.
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// max(x, 0) as ONE instruction: v_max_i32 on the float's bits (a negative float is a negative integer; -0 -> +0; a
// positive NaN stays a NaN).  fmaxf() costs two (hipcc first canonicalises a possible signalling NaN with v_max_f32 x, x),
// and so does v_med3_f32 x, 0, +inf, which it folds back into the same pair; on this chip the fp32 MFMA runs on the vector
// ALUs' own lanes, so a VALU instruction per matrix operand is not free (measured: ~4 matrix-pipe cycles each).  (An
// inline-asm v_max_f32 is NOT an option: the compiler's hazard recogniser does not see a VALU write inside an asm statement
// and leaves out the wait states an MFMA needs before it reads that register -- the first version did exactly that and
// computed garbage, run-to-run different.)
__device__ __forceinline__ float relu1(float x) {
  const int v = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, v > 0 ? v : 0);
}
// ---- buffer addressing.  A tile's addressing as (descriptor in SGPRs, ONE 32-bit per-lane byte offset, scalar offset in
// the instruction): no 64-bit per-lane address arithmetic, and a lane that must not touch memory carries BUF_OOB -- the
// bounds check returns 0 to its load and drops its store, so no select ever touches the data.  num_records is < 2^31 for
// every descriptor built here (launch preconditions), so an offset with bit 31 set is out of range for all of them.
// HAZARD (gfx950, unknown to hipcc 7.2): buffer_store_dwordx4 with an SGPR soffset reads its data registers a moment after
// issue, like the immediate-soffset form the compiler pads; do not write those registers in the next instructions.
typedef float f4v __attribute__((ext_vector_type(4)));
typedef int i4v __attribute__((ext_vector_type(4)));
constexpr int BUF_OOB = (int)0x80000000;
constexpr int BUF_WORD3 = 0x00020000;         // raw buffer, 32-bit data format
__device__ __forceinline__ __amdgpu_buffer_rsrc_t brsrc(const float* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, bytes, BUF_WORD3);
}
// descriptor over "everything within 2 GiB of p": for operands whose real extent the host has checked to be < 2^31 bytes
__device__ __forceinline__ __amdgpu_buffer_rsrc_t brsrc2g(const float* p) { return brsrc(p, (int)0x80000000); }
__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void bstore(float v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, voff, soff, 0);
}
__device__ __forceinline__ f4v bload4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void bstore4(f4v v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4v, v), r, voff, soff, 0);
}
// row of D register r for lane-half h
__device__ __forceinline__ constexpr int mfma32_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Bijective XCD-aware remap of a linear block id: blocks that the dispatcher
// round-robins onto one XCD (ids congruent mod 8) get a contiguous chunk of
// logical tiles, so neighbouring tiles share that XCD's L2.  Speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7, j = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

// XCD-aware walk of a persistent grid of 4-wave workgroups over `ntiles` wave-tiles.  Workgroups are dealt round-robin to
// the 8 XCDs (each with its own L2), so XCD x takes the contiguous eighth [x*ntiles/8, (x+1)*ntiles/8) and its waves sweep it
// side by side: tile j + d (the other tap, the residual, the neighbouring halo) is then a line that a wave of the SAME XCD
// fetched a moment ago.  Dealt out by raw wave id, neighbouring tiles sat in different XCDs and every activation plane
// crossed the fabric ~1.4-1.7 times (FETCH_SIZE: 117 -> 94 MB per encoder-layer launch for 86 MB algorithmic).  Falls back
// to the plain strided walk when the grid is not a whole number of XCD groups.  Speed only: every tile is visited once.
struct TileWalk { long first, last, stride; };
__device__ __forceinline__ TileWalk xcd_walk(long ntiles) {
  TileWalk w;
  w.first = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  w.last = ntiles;
  w.stride = ((long)gridDim.x * blockDim.x) >> 6;
  if ((gridDim.x & 7) == 0 && blockDim.x == 256) {
    const int xcd = blockIdx.x & 7;
    w.first = ntiles * xcd / 8 + ((blockIdx.x >> 3) * 4 + (threadIdx.x >> 6));
    w.last = ntiles * (xcd + 1) / 8;
    w.stride = (gridDim.x >> 3) * 4;
  }
  return w;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
