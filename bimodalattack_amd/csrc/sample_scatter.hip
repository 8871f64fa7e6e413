// a3, second half -- candidate construction (reference bimodal_attack.py:142, :150-162).
//
//   pos  = argsort(rand(B, n_opt))[:, :n_rep]            (bma_rand_positions)
//   out  = ids.repeat(B,1).scatter_(1, pos, topk_idx[pos, rank])   (bma_sample_scatter)
//
// Index work on a few KB per step: launch-bound.  One lane per candidate; the
// n_opt random keys of a candidate are selected with an n_rep-pass running minimum
// (n_rep is 1 in every BASELINE config), ties by position, which is what a stable
// argsort yields.  Any suffix length.

#include "bma_common.h"
#include "bma_profile.h"

namespace {

__global__ __launch_bounds__(256) void rand_positions_kernel(const float* __restrict__ rnd, int B, int n_opt,
                                                             int n_rep, int64_t* __restrict__ pos_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* r = rnd + static_cast<int64_t>(b) * n_opt;
  // pass j picks the smallest (key, position) pair that is larger than pass j-1's pick: the
  // order a stable argsort yields, for any n_opt and without a "taken" set
  float pv = 0.0f;
  int pp = -1;
  for (int j = 0; j < n_rep; ++j) {
    int best = -1;
    float bv = 0.0f;
    for (int p = 0; p < n_opt; ++p) {
      const float v = r[p];
      if (pp >= 0 && !(v > pv || (v == pv && p > pp))) continue;   // picked already (or in front of the last pick)
      if (best < 0 || v < bv) {
        best = p;
        bv = v;
      }
    }
    pv = bv;
    pp = best;
    pos_out[static_cast<int64_t>(b) * n_rep + j] = best;
  }
}

__global__ __launch_bounds__(256) void sample_scatter_kernel(const int64_t* __restrict__ ids,
                                                             const int64_t* __restrict__ topk_idx,
                                                             const int64_t* __restrict__ pos,
                                                             const int64_t* __restrict__ rank, int B, int n_opt,
                                                             int n_rep, int k, int64_t* __restrict__ out) {
  // one lane per (candidate, position): coalesced int64 row writes
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= static_cast<int64_t>(B) * n_opt) return;
  const int b = static_cast<int>(i / n_opt), p = static_cast<int>(i % n_opt);
  int64_t v = ids[p];
  // later j overwrite earlier ones, as consecutive scatter writes would; positions
  // from an argsort are distinct, so at most one j matches
  for (int j = 0; j < n_rep; ++j) {
    const int64_t pj = pos[static_cast<int64_t>(b) * n_rep + j];
    if (pj == p) {
      int64_t r = rank[static_cast<int64_t>(b) * n_rep + j];
      r = r < 0 ? 0 : (r >= k ? k - 1 : r);  // never read outside the table
      v = topk_idx[static_cast<int64_t>(p) * k + r];
    }
  }
  out[i] = v;
}

}  // namespace

extern "C" int bma_rand_positions(const float* rnd, int B, int n_opt, int n_rep, int64_t* pos_out,
                                  void* stream) {
  if (B < 0 || n_opt <= 0 || n_rep <= 0 || n_rep > n_opt) return BMA_EINVAL;
  if (B == 0) return BMA_OK;
  if (!rnd || !pos_out) return BMA_EINVAL;
  hipLaunchKernelGGL(rand_positions_kernel, dim3((B + 255) / 256), dim3(256), 0,
                     static_cast<hipStream_t>(stream), rnd, B, n_opt, n_rep, pos_out);
  BMA_LAUNCH_CHECK();
  return BMA_OK;
}

extern "C" int bma_sample_scatter(const int64_t* ids, const int64_t* topk_idx, const int64_t* pos,
                                  const int64_t* rank, int B, int n_opt, int n_rep, int k, int64_t* out,
                                  void* stream) {
  if (B < 0 || n_opt <= 0 || n_rep <= 0 || n_rep > n_opt || k <= 0) return BMA_EINVAL;
  if (B == 0) return BMA_OK;
  if (!ids || !topk_idx || !pos || !rank || !out) return BMA_EINVAL;
  const int64_t n = static_cast<int64_t>(B) * n_opt;
  BMA_PROF_BEGIN(BMA_K_SCATTER, static_cast<hipStream_t>(stream), 8.0 * n + 16.0 * B * n_rep);
  hipLaunchKernelGGL(sample_scatter_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), ids, topk_idx, pos, rank, B, n_opt, n_rep, k, out);
  BMA_PROF_END(BMA_K_SCATTER, static_cast<hipStream_t>(stream));
  BMA_LAUNCH_CHECK();
  return BMA_OK;
}
