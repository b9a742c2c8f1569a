// gemm.hip -- dense fp32 GEMM entry point on the igemm engine (NN / NT / TN / TT).
// Replaces the nn.LSTM projections and nn.Linear of the reference heads
// (packages/models/Audio_Net.py:30-35,51-59; Video_Net.py:45-51,102-116; AV_Net.py:53-58,128-140)
// and their autograd backward.
#include "gemm_api.h"

namespace {

template <int BM, int BN, class AOp, class BOp>
int run2(const AOp& a, const BOp& b, const igemm::EpiStore& e, const avvad_gemm_desc* d, hipStream_t s, float* slab) {
  return igemm::launch<BM, BN>(a, b, e, d->M, d->N, d->K, d->split_k, s, slab);
}

template <class AOp, class BOp>
int run1(const AOp& a, const BOp& b, const igemm::EpiStore& e, const avvad_gemm_desc* d, hipStream_t s, float* slab) {
  if (d->M <= 64 || d->N <= 64) return run2<64, 64>(a, b, e, d, s, slab);
  return run2<128, 128>(a, b, e, d, s, slab);
}

template <class AOp>
int runA(const AOp& a, const float* B, const igemm::EpiStore& e, const avvad_gemm_desc* d, hipStream_t s, float* slab) {
  if (d->transB) {  // stored [N][K]
    igemm::RowPlain b{B, d->ldb, d->N, d->K, d->relu_b};
    return run1(a, b, e, d, s, slab);
  }
  const bool v4 = (d->ldb % 4 == 0) && (d->N % 4 == 0) && ((uintptr_t)B % 16 == 0);
  if (v4) {
    igemm::ColPlain<4> b{B, d->ldb, d->N, d->K, d->relu_b};
    return run1(a, b, e, d, s, slab);
  }
  igemm::ColPlain<1> b{B, d->ldb, d->N, d->K, d->relu_b};
  return run1(a, b, e, d, s, slab);
}

}  // namespace

int avvad_gemm_impl(const float* A, const float* B, const float* bias, float* C, const avvad_gemm_desc* d, hipStream_t s, float* slab) {
  if (!A || !B || !C || !d || d->M <= 0 || d->N <= 0 || d->K <= 0) return AVVAD_EINVAL;
  if (d->split_k > 1 && !d->accumulate) return AVVAD_EINVAL;  // historical: a K-split product accumulates onto C
  igemm::EpiStore e{C, d->ldc, bias, d->split_k > 1 ? 2 : (d->accumulate ? 1 : 0)};
  if (bias && d->split_k > 1) return AVVAD_EINVAL;
  if (!d->transA) {  // [M][K]
    if (d->lda % 4 == 0 && d->K % 4 == 0 && (uintptr_t)A % 16 == 0 && (long)d->M * d->lda < (1L << 29) - 64) {
      igemm::RowVec4 a{A, d->lda, d->M, d->K, d->relu_a};     // 16-byte buffer fetches along K
      return runA(a, B, e, d, s, slab);
    }
    igemm::RowPlain a{A, d->lda, d->M, d->K, d->relu_a};
    return runA(a, B, e, d, s, slab);
  }
  const bool v4 = (d->lda % 4 == 0) && (d->M % 4 == 0) && ((uintptr_t)A % 16 == 0);
  if (v4) {
    igemm::ColPlain<4> a{A, d->lda, d->M, d->K, d->relu_a};
    return runA(a, B, e, d, s, slab);
  }
  igemm::ColPlain<1> a{A, d->lda, d->M, d->K, d->relu_a};
  return runA(a, B, e, d, s, slab);
}

extern "C" int avvad_gemm_f32(const float* A, const float* B, const float* bias, float* C, const avvad_gemm_desc* d, void* ws,
                              size_t ws_bytes, avvad_stream_t s) {
  AVVAD_ENTER();
  float* slab = (ws && ws_bytes >= igemm::SLAB_FLOATS * sizeof(float)) ? (float*)ws : nullptr;
  return avvad_gemm_impl(A, B, bias, C, d, (hipStream_t)s, slab);
}
