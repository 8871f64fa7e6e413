// a5 -- PGD L-inf step (reference bimodal_attack.py:1030-1037).
//
//   out = clamp(clamp(x - step*sign(g), x0-eps, x0+eps), 0, 1)      fp32, exact
//
// HBM-bound, 16 algorithmic bytes per pixel (3 reads + 1 write).  At the
// LLaVA image size (338,688 px, 5.4 MB) the launch is latency-bound; at the
// Gemma size (2,408,448 px, 38.5 MB) it streams.  One float4 per lane per
// access (16 B/lane, 1 KiB per wave-instruction), grid-stride over
// <= 2048 workgroups.
//
// Bit-exactness notes: sign() is torch's ((g>0) - (g<0): 0 for NaN and -0.0);
// step*sign is exact, so contracting it into an FMA cannot change the result;
// both clamps are written as compare+select so NaN and -0.0 propagate exactly
// as torch.clamp propagates them.

#include "bma_common.h"
#include "bma_profile.h"

namespace {

__device__ __forceinline__ float linf_one(float x, float g, float x0, float eps, float step) {
  const float s = (g > 0.0f ? 1.0f : 0.0f) - (g < 0.0f ? 1.0f : 0.0f);
  float y = x - step * s;
  const float lo = x0 - eps, hi = x0 + eps;
  y = y < lo ? lo : y;
  y = y > hi ? hi : y;
  y = y < 0.0f ? 0.0f : y;
  y = y > 1.0f ? 1.0f : y;
  return y;
}

__global__ __launch_bounds__(256) void linf_step_vec4(const bma::float4_t* x,
                                                      const bma::float4_t* __restrict__ g,
                                                      const bma::float4_t* __restrict__ x0,
                                                      bma::float4_t* out, int64_t n4,
                                                      float eps, float step) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const bma::float4_t vx = x[i], vg = g[i], v0 = x0[i];
    bma::float4_t r;
    r.x = linf_one(vx.x, vg.x, v0.x, eps, step);
    r.y = linf_one(vx.y, vg.y, v0.y, eps, step);
    r.z = linf_one(vx.z, vg.z, v0.z, eps, step);
    r.w = linf_one(vx.w, vg.w, v0.w, eps, step);
    out[i] = r;
  }
}

__global__ __launch_bounds__(256) void linf_step_scalar(const float* x, const float* g, const float* x0,
                                                        float* out, int64_t begin, int64_t n, float eps,
                                                        float step) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = begin + static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = linf_one(x[i], g[i], x0[i], eps, step);
}

}  // namespace

extern "C" int bma_linf_step(const float* x, const float* g, const float* x0, int64_t n, float eps,
                             float step, float* out, void* stream) {
  if (n < 0 || (n > 0 && (!x || !g || !x0 || !out))) return BMA_EINVAL;
  if (n == 0) return BMA_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const uintptr_t bits = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(g) |
                         reinterpret_cast<uintptr_t>(x0) | reinterpret_cast<uintptr_t>(out);
  if (bits & 3) return BMA_EALIGN;
  int64_t done = 0;
  if ((bits & 15) == 0 && n >= 4) {
    const int64_t n4 = n / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    BMA_PROF_BEGIN(BMA_K_LINF, st, 16.0 * static_cast<double>(n4) * 4.0);
    hipLaunchKernelGGL(linf_step_vec4, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st,
                       reinterpret_cast<const bma::float4_t*>(x), reinterpret_cast<const bma::float4_t*>(g),
                       reinterpret_cast<const bma::float4_t*>(x0), reinterpret_cast<bma::float4_t*>(out), n4,
                       eps, step);
    BMA_PROF_END(BMA_K_LINF, st);
    BMA_LAUNCH_CHECK();
    done = n4 * 4;
  }
  if (done < n) {
    int64_t blocks = (n - done + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(linf_step_scalar, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, x, g, x0, out,
                       done, n, eps, step);
    BMA_LAUNCH_CHECK();
  }
  return BMA_OK;
}
