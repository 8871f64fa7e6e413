// a3, first half -- forbidden-token mask + top-k of the negated token gradient
//   reference bimodal_attack.py:144-147:  grad[:, not_allowed] = inf ; (-grad).topk(k).indices
//
// Per suffix position, pick the k allowed tokens with the most negative gradient,
// ordered (gradient ascending, token id ascending).  Exact selection, not a sort of
// the row: an MSB-first radix SELECT over the 64-bit composite key
//       [ order-preserving image of the gradient : 32 | token id : 32 ]
// which makes every element distinct, so ties at the k-th value are resolved by
// token id with no special case.  11-bit digits: fp32 gradients need 3 histogram
// passes, bf16/fp16 need 2 (their low 16 key bits are zero); the id digits are only
// visited when the k-th value is tied.  A last pass collects the k winners into LDS,
// a bitonic network orders them, and lane-contiguous int64 stores write them out.
//
// The data is tiny (rows*V*2 B = 1.2 MB for LLaVA, 10 MB for Gemma) and stays in L2
// across passes; the launch is latency-bound, not HBM-bound.  One 1024-thread
// workgroup per row keeps 16 waves of loads in flight on the row's CU.
//
// Unlike the reference, the gradient is not overwritten with +inf (:145): nothing
// reads it afterwards.

#include "bma_common.h"
#include "bma_profile.h"

namespace {

using bma::uint4_t;

constexpr int kTPB = 1024;
constexpr int kBins = 2048;
constexpr int kMaxK = 2048;

// DT only matters for the low key bits: a widened bf16 (fp16) value has 16 (13) zero
// low bits, which "~u" below would turn into ones for negative values; clearing them
// keeps the order and lets 16-bit rows skip the digit that covers key bits 9..0.
template <int DT>
__device__ __forceinline__ uint32_t order_key(float v, bool masked) {
  if (masked) return 0xffff0000u;          // +inf by decree: behind every real value (+inf maps to 0xff800000);
                                           // low 16 bits zero so 16-bit rows may skip that digit
  uint32_t u = __float_as_uint(v);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0u;   // NaN: torch.topk ranks it first in -grad
  if (u == 0x80000000u) u = 0u;                      // -0.0 == +0.0
  const uint32_t key = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending in v
  return DT == BMA_F32 ? key : (DT == BMA_BF16 ? (key & 0xffff0000u) : (key & 0xffffe000u));
}

template <int DT>
__device__ __forceinline__ float elem_at(const void* base, int i) {
  if (DT == BMA_F32) return static_cast<const float*>(base)[i];
  const uint32_t h = static_cast<const uint16_t*>(base)[i];
  return DT == BMA_BF16 ? bma::bf16_bits_to_f32(h) : bma::f16_bits_to_f32(h);
}

// Calls f(composite) for every element of the row, 16 bytes per lane when aligned.
template <int DT, bool VEC, typename F>
__device__ __forceinline__ void for_each_key(const void* base, int V, const uint32_t* __restrict__ mask, F&& f) {
  const int tid = threadIdx.x;
  if (VEC) {
    constexpr int NE = 16 / bma::elem_bytes<DT>::value;
    const uint4_t* p = static_cast<const uint4_t*>(base);
    const int nvec = V / NE;
    for (int i = tid; i < nvec; i += kTPB) {
      const uint4_t w = p[i];
      const int e0 = i * NE;
      const uint32_t mw = mask ? (mask[e0 >> 5] >> (e0 & 31)) : 0u;  // NE divides 32
      float v[NE];
      if (DT == BMA_F32) {
        v[0] = __uint_as_float(w.x); v[1] = __uint_as_float(w.y);
        v[2] = __uint_as_float(w.z); v[3] = __uint_as_float(w.w);
      } else {
        v[0] = bma::unpack16<DT>(w.x, 0); v[1] = bma::unpack16<DT>(w.x, 1);
        v[2 % NE] = bma::unpack16<DT>(w.y, 0); v[3 % NE] = bma::unpack16<DT>(w.y, 1);
        v[4 % NE] = bma::unpack16<DT>(w.z, 0); v[5 % NE] = bma::unpack16<DT>(w.z, 1);
        v[6 % NE] = bma::unpack16<DT>(w.w, 0); v[7 % NE] = bma::unpack16<DT>(w.w, 1);
      }
#pragma unroll
      for (int j = 0; j < NE; ++j)
        f((static_cast<uint64_t>(order_key<DT>(v[j], (mw >> j) & 1u)) << 32) | static_cast<uint32_t>(e0 + j));
    }
  } else {
    for (int i = tid; i < V; i += kTPB) {
      const bool m = mask ? ((mask[i >> 5] >> (i & 31)) & 1u) : false;
      f((static_cast<uint64_t>(order_key<DT>(elem_at<DT>(base, i), m)) << 32) | static_cast<uint32_t>(i));
    }
  }
}

template <int DT, bool VEC>
__global__ __launch_bounds__(kTPB) void mask_topk_kernel(const void* __restrict__ grad, int64_t ld_row, int V,
                                                         const uint32_t* __restrict__ mask, int k, int npow2,
                                                         int64_t* __restrict__ idx_out) {
  __shared__ uint32_t hist[kBins];
  __shared__ uint32_t wave_tot[kTPB / 64];
  __shared__ uint64_t sel[kMaxK];
  __shared__ uint32_t s_bin, s_below, s_cnt, s_count;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const void* base = static_cast<const char*>(grad) +
                     static_cast<int64_t>(blockIdx.x) * ld_row * bma::elem_bytes<DT>::value;

  uint64_t prefix = 0, pmask = 0;
  uint32_t need = static_cast<uint32_t>(k);
  bool done = false;

  // digit p covers composite bits [shift, shift+width)
  const int shifts[6] = {53, 42, 31, 20, 9, 0};
  const int widths[6] = {11, 11, 11, 11, 11, 9};
#pragma unroll 1
  for (int p = 0; p < 6 && !done; ++p) {
    const int shift = shifts[p];
    const uint64_t dmask = (1ull << widths[p]) - 1ull;
    if (DT != BMA_F32 && p == 2) {  // key bits 9..0 and id bit 31 are zero for 16-bit gradients
      pmask |= dmask << shift;
      continue;
    }
    for (int i = tid; i < kBins; i += kTPB) hist[i] = 0;
    __syncthreads();
    for_each_key<DT, VEC>(base, V, mask, [&](uint64_t c) {
      if ((c & pmask) == prefix) atomicAdd(&hist[static_cast<uint32_t>((c >> shift) & dmask)], 1u);
    });
    __syncthreads();
    // block-wide inclusive scan over the 2048 bins, two bins per lane
    const uint32_t h0 = hist[2 * tid], h1 = hist[2 * tid + 1];
    uint32_t incl = h0 + h1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(incl, o, BMA_WAVE);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wave; ++w) before += wave_tot[w];
    const uint32_t excl = before + incl - (h0 + h1);
    // the pair of bins in which the running count first reaches `need`
    if (excl < need && need <= excl + h0 + h1) {
      if (need <= excl + h0) { s_bin = 2 * tid; s_below = excl; s_cnt = h0; }
      else { s_bin = 2 * tid + 1; s_below = excl + h0; s_cnt = h1; }
    }
    __syncthreads();
    need -= s_below;
    prefix |= static_cast<uint64_t>(s_bin) << shift;
    pmask |= dmask << shift;
    done = (s_cnt == need);  // the whole bin is wanted: nothing left to split
    __syncthreads();
  }

  // collect the k winners: everything whose decided digits are <= the threshold's
  if (tid == 0) s_count = 0;
  __syncthreads();
  for_each_key<DT, VEC>(base, V, mask, [&](uint64_t c) {
    if ((c & pmask) <= prefix) {
      const uint32_t slot = atomicAdd(&s_count, 1u);
      if (slot < static_cast<uint32_t>(kMaxK)) sel[slot] = c;
    }
  });
  __syncthreads();
  const int got = static_cast<int>(s_count);  // == k by construction
  for (int i = tid; i < npow2; i += kTPB)
    if (i >= got) sel[i] = ~0ull;
  __syncthreads();

  // bitonic sort of npow2 composites, ascending
  for (int size = 2; size <= npow2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = tid; i < (npow2 >> 1); i += kTPB) {
        const int lo = 2 * i - (i & (stride - 1));
        const int hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const uint64_t a = sel[lo], b = sel[hi];
        if ((a > b) == up) { sel[lo] = b; sel[hi] = a; }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < k; i += kTPB)
    idx_out[static_cast<int64_t>(blockIdx.x) * k + i] = static_cast<int64_t>(sel[i] & 0xffffffffull);
}

template <int DT>
int launch(const void* grad, int64_t ld_row, int rows, int V, const uint32_t* mask, int k, int64_t* idx_out,
           hipStream_t st) {
  constexpr int ES = bma::elem_bytes<DT>::value;
  int npow2 = 2;
  while (npow2 < k) npow2 <<= 1;
  const bool vec = (reinterpret_cast<uintptr_t>(grad) % 16 == 0) && ((ld_row * ES) % 16 == 0) &&
                   ((static_cast<int64_t>(V) * ES) % 16 == 0);
  BMA_PROF_BEGIN(BMA_K_TOPK, st, static_cast<double>(rows) * V * ES + static_cast<double>(rows) * k * 8.0);
  if (vec)
    hipLaunchKernelGGL((mask_topk_kernel<DT, true>), dim3(rows), dim3(kTPB), 0, st, grad, ld_row, V, mask, k, npow2,
                       idx_out);
  else
    hipLaunchKernelGGL((mask_topk_kernel<DT, false>), dim3(rows), dim3(kTPB), 0, st, grad, ld_row, V, mask, k,
                       npow2, idx_out);
  BMA_PROF_END(BMA_K_TOPK, st);
  BMA_LAUNCH_CHECK();
  return BMA_OK;
}

}  // namespace

extern "C" int bma_mask_topk(const void* grad, int64_t ld_row, int rows, int V, int dtype,
                             const uint32_t* mask_bits, int k, int64_t* idx_out, void* stream) {
  if (rows < 0 || V <= 0 || k <= 0 || k > V || ld_row < V) return BMA_EINVAL;
  if (k > kMaxK) return BMA_ELIMIT;
  if (rows == 0) return BMA_OK;
  if (!grad || !idx_out) return BMA_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case BMA_F32: return launch<BMA_F32>(grad, ld_row, rows, V, mask_bits, k, idx_out, st);
    case BMA_BF16: return launch<BMA_BF16>(grad, ld_row, rows, V, mask_bits, k, idx_out, st);
    case BMA_F16: return launch<BMA_F16>(grad, ld_row, rows, V, mask_bits, k, idx_out, st);
    default: return BMA_EDTYPE;
  }
}
