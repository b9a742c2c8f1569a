// gemm_api.h -- internal (C++) handle on the dense GEMM so other translation
// units (LSTM, WaveNet) reuse the instantiations compiled in gemm.hip.
#pragma once
#include "igemm.h"

// slab: igemm::SLAB_FLOATS floats of scratch (partial tiles of the stream-K round) or nullptr (whole-tile schedule)
int avvad_gemm_impl(const float* A, const float* B, const float* bias, float* C, const avvad_gemm_desc* d, hipStream_t s,
                    float* slab);

static inline avvad_gemm_desc gemm_desc(int M, int N, int K, int lda, int ldb, int ldc, int tA, int tB, int acc, int split) {
  avvad_gemm_desc d;
  d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldb = ldb; d.ldc = ldc;
  d.transA = tA; d.transB = tB; d.accumulate = acc; d.split_k = split; d.relu_a = 0; d.relu_b = 0;
  return d;
}
