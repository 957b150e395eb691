// Fused multi-tensor RAdam step for the decoder's ~70 parameter tensors (2.85 M fp32).
//
// Reference: utils/optimization_utils.py:31-97 -- a Python loop over every parameter, ~10 elementwise kernels each
// (~700 launches per optimiser step for the decoder).  Here all tensors of one (step count, hyper-parameter) group are updated
// by a handful of launches: the host packs up to MAX_TENSORS tensor pointer quadruples and a block -> (tensor, chunk) map into
// the kernel ARGUMENT (no device-side table to keep in sync with the gradient tensors, which autograd re-allocates every
// step), apex-style.  Per element, exactly the reference's update:
//     v <- beta2 v + (1 - beta2) g g ;   m <- beta1 m + (1 - beta1) g                       (:57-58)
//     mode 2 (N_sma >= 5):  p <- p - wd lr p ;  p <- p - step_size lr  m / (sqrt(v) + eps)   (:83-87)
//     mode 1 (SGD-like)  :  p <- p - wd lr p ;  p <- p - step_size lr  m                     (:89-92)
//     mode 0             :  moments only (step_size < 0: degenerated_to_sgd = False and N_sma < 5)
// HBM-bound streaming: 4 reads + 3 writes of 4 bytes per element.
#include "common.h"

namespace qagnn {

constexpr int RADAM_MAX_TENSORS = 24;
constexpr int RADAM_MAX_BLOCKS = 320;
constexpr int RADAM_CHUNK = 4096;  // elements per block

struct radam_pack {
  float* p[RADAM_MAX_TENSORS];
  const float* g[RADAM_MAX_TENSORS];
  float* m[RADAM_MAX_TENSORS];
  float* v[RADAM_MAX_TENSORS];
  int numel[RADAM_MAX_TENSORS];
  int block_chunk[RADAM_MAX_BLOCKS];
  unsigned char block_tensor[RADAM_MAX_BLOCKS];
  float beta1, beta2, ob1, ob2, eps, decay, s;  // ob = 1 - beta, decay = -wd * lr, s = -step_size * lr: derived in DOUBLE on the host,
  int mode;                                     // like the reference's Python scalars (1 - 0.999 in fp32 is off by 1.3e-5 relative)
};

__global__ __launch_bounds__(256) void k_radam_multi(const radam_pack a) {
  const int t = a.block_tensor[blockIdx.x];
  const int base = a.block_chunk[blockIdx.x] * RADAM_CHUNK;
  const int n = min(a.numel[t] - base, RADAM_CHUNK);
  float* __restrict__ p = a.p[t] + base;
  const float* __restrict__ g = a.g[t] + base;
  float* __restrict__ m = a.m[t] + base;
  float* __restrict__ v = a.v[t] + base;
  const float ob1 = a.ob1, ob2 = a.ob2, decay = a.decay, s = a.s;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float gi = g[i];
    const float vi = v[i] * a.beta2 + ob2 * gi * gi;
    const float mi = m[i] * a.beta1 + ob1 * gi;
    v[i] = vi;
    m[i] = mi;
    if (a.mode) {
      float pi = p[i];
      if (decay != 0.0f) pi += decay * pi;
      pi += a.mode == 2 ? s * (mi / (sqrtf(vi) + a.eps)) : s * mi;
      p[i] = pi;
    }
  }
}

}  // namespace qagnn

using namespace qagnn;

extern "C" int qagnn_radam_step_f32(int32_t n_tensors, float* const* p, const float* const* g, float* const* m, float* const* v,
                                    const int64_t* numel, double beta1, double beta2, double eps, double lr, double weight_decay,
                                    double step_size, int32_t mode, qagnn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  QAGNN_REQUIRE(n_tensors >= 0 && (n_tensors == 0 || (p && g && m && v && numel)), QAGNN_EINVAL, "radam_step: null table");
  QAGNN_REQUIRE(mode >= 0 && mode <= 2, QAGNN_EINVAL, "radam_step: mode %d", mode);
  radam_pack a;
  a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.ob1 = (float)(1.0 - beta1); a.ob2 = (float)(1.0 - beta2); a.eps = (float)eps;
  a.decay = (float)(-weight_decay * lr); a.s = (float)(-step_size * lr); a.mode = mode;
  int nt = 0, nb = 0;
  auto flush = [&]() -> int {
    if (nb == 0) { nt = 0; return QAGNN_OK; }
    k_radam_multi<<<nb, 256, 0, stream>>>(a);
    QAGNN_LAUNCH_CHECK("k_radam_multi");
    nt = 0; nb = 0;
    return QAGNN_OK;
  };
  for (int i = 0; i < n_tensors; ++i) {
    QAGNN_REQUIRE(p[i] && g[i] && m[i] && v[i] && numel[i] >= 0 && numel[i] < (1ll << 31), QAGNN_EINVAL, "radam_step: tensor %d: null pointer or bad size", i);
    const int chunks = cdiv(numel[i], RADAM_CHUNK);
    int c = 0;
    while (c < chunks) {
      if (nt == RADAM_MAX_TENSORS || nb == RADAM_MAX_BLOCKS) {
        int rc = flush();
        if (rc != QAGNN_OK) return rc;
      }
      // (re-)open tensor i in the current pack
      const int slot = nt++;
      a.p[slot] = p[i]; a.g[slot] = g[i]; a.m[slot] = m[i]; a.v[slot] = v[i]; a.numel[slot] = (int)numel[i];
      while (c < chunks && nb < RADAM_MAX_BLOCKS) {
        a.block_tensor[nb] = (unsigned char)slot;
        a.block_chunk[nb] = c++;
        ++nb;
      }
    }
  }
  return flush();
}
