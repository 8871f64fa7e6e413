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
// The data is tiny (rows*V*2 B = 1.2 MB for LLaVA, 10 MB for Gemma): latency-bound, not
// HBM-bound.  19 rows on 256 CUs is what limits a row-per-workgroup kernel, so rows longer
// than one slice (4096 tokens) are cut across workgroups in two stages, every element read
// from memory exactly ONCE:
//   topk_slice_kernel   grid (slices, rows), 256 threads: a slice's 4096 composites live in
//                       registers; the same radix select, on them, finds the slice's own k
//                       smallest -- the row's k smallest are among those -- and writes them
//                       to the workspace as 64-bit composites;
//   topk_merge_kernel   one 1024-thread workgroup per row: slices*k composites in
//                       registers, radix select of the k smallest, LDS bitonic sort, ids out.
// Exact like the one-workgroup kernel (same composite order); that kernel still serves short
// rows (one slice) and k*slices beyond the merge kernel's registers.
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

// ---------------------------------------------------------------------------------------------
// Two-stage select.  wg_select: the `need` smallest of NT*EPT distinct composites held in registers
// (nbits significant bits each; unused slots hold ~0).  MSB-first, 11-bit digits, LDS histogram.
// Returns (threshold, low): an element c is selected iff (c >> low) <= threshold.
constexpr int kSlice = 4096;
constexpr int kSliceThreads = 256;
constexpr int kMergeThreads = 1024;
constexpr int kMergeMaxEpt = 32;

template <int NT, int EPT>
__device__ __forceinline__ void wg_select(const uint64_t (&c)[EPT], int nbits, uint32_t need, uint32_t* hist /*2048*/,
                                          uint32_t* wave_tot /*NT/64*/, uint32_t* s_pick /*3*/, uint64_t& thr, int& low) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int BPT = kBins / NT;               // bins per thread in the scan (8 at 256 threads, 2 at 1024)
  uint64_t prefix = 0;                          // the decided high bits, right-aligned
  int hi = nbits;
  while (hi > 0) {
    const int lo = hi > 11 ? hi - 11 : 0;
    const int width = hi - lo;
    const uint32_t dmask = (1u << width) - 1u;
    for (int i = tid; i < kBins; i += NT) hist[i] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < EPT; ++j)
      if ((c[j] >> hi) == prefix && c[j] != ~0ull) atomicAdd(&hist[static_cast<uint32_t>(c[j] >> lo) & dmask], 1u);
    __syncthreads();
    uint32_t hb[BPT];
    uint32_t mine = 0;
#pragma unroll
    for (int b = 0; b < BPT; ++b) { hb[b] = hist[BPT * tid + b]; mine += hb[b]; }
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(incl, o, BMA_WAVE);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wave; ++w) before += wave_tot[w];
    uint32_t run = before + incl - mine;        // elements in bins below this thread's
    if (run < need && need <= run + mine) {
#pragma unroll
      for (int b = 0; b < BPT; ++b) {
        if (run < need && need <= run + hb[b]) { s_pick[0] = BPT * tid + b; s_pick[1] = run; s_pick[2] = hb[b]; }
        run += hb[b];
      }
    }
    __syncthreads();
    const uint32_t bin = s_pick[0], below = s_pick[1], cnt = s_pick[2];
    need -= below;
    prefix = (prefix << width) | bin;
    hi = lo;
    __syncthreads();
    if (cnt == need) break;                     // the whole bin is wanted: nothing left to split
  }
  thr = prefix;
  low = hi;
}

template <int DT> struct key_bits { static constexpr int value = DT == BMA_F32 ? 32 : (DT == BMA_BF16 ? 16 : 19); };

// stage 1: grid (slices, rows).  Workspace layout: ws[(row*slices + slice)*k + j], ~0 in unused slots.
template <int DT, bool VEC>
__global__ __launch_bounds__(kSliceThreads) void topk_slice_kernel(const void* __restrict__ grad, int64_t ld_row, int V,
                                                                   const uint32_t* __restrict__ mask, int k,
                                                                   uint64_t* __restrict__ ws) {
  constexpr int EPT = kSlice / kSliceThreads;   // 16
  constexpr int KB = key_bits<DT>::value;
  constexpr int NE = 16 / bma::elem_bytes<DT>::value;
  __shared__ uint32_t hist[kBins];
  __shared__ uint32_t wave_tot[kSliceThreads / 64];
  __shared__ uint32_t s_pick[3];
  __shared__ uint32_t s_count;
  const int tid = threadIdx.x;
  const int slice = blockIdx.x, row = blockIdx.y, slices = gridDim.x;
  const int e_base = slice * kSlice;
  const void* base = static_cast<const char*>(grad) + static_cast<int64_t>(row) * ld_row * bma::elem_bytes<DT>::value;
  const int n_valid = V - e_base < kSlice ? V - e_base : kSlice;

  // local composite: [key, KB bits | index inside the slice, 12 bits]
  uint64_t c[EPT];
  auto put = [&](int j, float v, int e) {
    if (e < V) {
      const bool m = mask ? ((mask[e >> 5] >> (e & 31)) & 1u) : false;
      const uint32_t key = order_key<DT>(v, m) >> (32 - KB);
      c[j] = (static_cast<uint64_t>(key) << 12) | static_cast<uint32_t>(e - e_base);
    } else {
      c[j] = ~0ull;
    }
  };
  if (VEC) {
    const uint4_t* p = static_cast<const uint4_t*>(base);
#pragma unroll
    for (int it = 0; it < EPT / NE; ++it) {
      const int e0 = e_base + (tid + it * kSliceThreads) * NE;
      if (e0 < V) {                              // V is a multiple of NE here: a vector is all in or all out
        const uint4_t w = p[e0 / NE];
        if (DT == BMA_F32) {
          put(it * NE + 0, __uint_as_float(w.x), e0); put(it * NE + 1, __uint_as_float(w.y), e0 + 1);
          put(it * NE + 2, __uint_as_float(w.z), e0 + 2); put(it * NE + 3, __uint_as_float(w.w), e0 + 3);
        } else {
          put(it * NE + 0, bma::unpack16<DT>(w.x, 0), e0); put(it * NE + 1, bma::unpack16<DT>(w.x, 1), e0 + 1);
          put(it * NE + 2 % NE, bma::unpack16<DT>(w.y, 0), e0 + 2); put(it * NE + 3 % NE, bma::unpack16<DT>(w.y, 1), e0 + 3);
          put(it * NE + 4 % NE, bma::unpack16<DT>(w.z, 0), e0 + 4); put(it * NE + 5 % NE, bma::unpack16<DT>(w.z, 1), e0 + 5);
          put(it * NE + 6 % NE, bma::unpack16<DT>(w.w, 0), e0 + 6); put(it * NE + 7 % NE, bma::unpack16<DT>(w.w, 1), e0 + 7);
        }
      } else {
#pragma unroll
        for (int q = 0; q < NE; ++q) c[it * NE + q] = ~0ull;
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      const int e = e_base + tid + j * kSliceThreads;
      put(j, e < V ? elem_at<DT>(base, e) : 0.0f, e);
    }
  }

  const uint32_t want = static_cast<uint32_t>(k < n_valid ? k : n_valid);
  uint64_t thr = ~0ull;
  int low = 0;
  if (want < static_cast<uint32_t>(n_valid)) wg_select<kSliceThreads, EPT>(c, KB + 12, want, hist, wave_tot, s_pick, thr, low);
  if (tid == 0) s_count = 0;
  __syncthreads();
  uint64_t* out = ws + (static_cast<int64_t>(row) * slices + slice) * k;
#pragma unroll
  for (int j = 0; j < EPT; ++j) {
    if (c[j] != ~0ull && (c[j] >> low) <= thr) {
      const uint32_t slot = atomicAdd(&s_count, 1u);
      // global composite: [full 32-bit key | token id]
      const uint32_t key = static_cast<uint32_t>(c[j] >> 12) << (32 - KB);
      if (slot < static_cast<uint32_t>(k)) out[slot] = (static_cast<uint64_t>(key) << 32) | static_cast<uint32_t>(e_base + (c[j] & 0xfffu));
    }
  }
  __syncthreads();
  for (int i = static_cast<int>(s_count) + tid; i < k; i += kSliceThreads) out[i] = ~0ull;
}

// stage 2: one workgroup per row over its slices*k stage-1 composites.
template <int DT, int EPT>
__global__ __launch_bounds__(kMergeThreads) void topk_merge_kernel(const uint64_t* __restrict__ ws, int n_in, int idbits, int k,
                                                                   int npow2, int64_t* __restrict__ idx_out) {
  constexpr int KB = key_bits<DT>::value;
  __shared__ uint32_t hist[kBins];
  __shared__ uint32_t wave_tot[kMergeThreads / 64];
  __shared__ uint32_t s_pick[3];
  __shared__ uint32_t s_count;
  __shared__ uint64_t sel[kMaxK];
  const int tid = threadIdx.x;
  const uint64_t* in = ws + static_cast<int64_t>(blockIdx.x) * n_in;
  // select on the compact form [key, KB bits | id, idbits]: fewer digits than the 64-bit composite
  uint64_t c[EPT];
#pragma unroll
  for (int j = 0; j < EPT; ++j) {
    const int i = tid + j * kMergeThreads;
    const uint64_t g = i < n_in ? in[i] : ~0ull;
    c[j] = g == ~0ull ? ~0ull : (((g >> (64 - KB)) << idbits) | (g & 0xffffffffull));
  }
  uint64_t thr;
  int low;
  wg_select<kMergeThreads, EPT>(c, KB + idbits, static_cast<uint32_t>(k), hist, wave_tot, s_pick, thr, low);
  if (tid == 0) s_count = 0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < EPT; ++j)
    if (c[j] != ~0ull && (c[j] >> low) <= thr) {
      const uint32_t slot = atomicAdd(&s_count, 1u);
      if (slot < static_cast<uint32_t>(kMaxK)) sel[slot] = c[j];          // the compact form orders like the full one
    }
  __syncthreads();
  const int got = static_cast<int>(s_count);  // == k by construction
  for (int i = tid; i < npow2; i += kMergeThreads)
    if (i >= got) sel[i] = ~0ull;
  __syncthreads();
  for (int size = 2; size <= npow2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = tid; i < (npow2 >> 1); i += kMergeThreads) {
        const int lo = 2 * i - (i & (stride - 1));
        const int hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const uint64_t a = sel[lo], b = sel[hi];
        if ((a > b) == up) { sel[lo] = b; sel[hi] = a; }
      }
      __syncthreads();
    }
  }
  const uint64_t idmask = (1ull << idbits) - 1ull;
  for (int i = tid; i < k; i += kMergeThreads)
    idx_out[static_cast<int64_t>(blockIdx.x) * k + i] = static_cast<int64_t>(sel[i] & idmask);
}

inline int slices_of(int V) { return (V + kSlice - 1) / kSlice; }
inline bool two_stage_ok(int V, int k) {
  const int s = slices_of(V);
  return s >= 2 && static_cast<int64_t>(s) * k <= static_cast<int64_t>(kMergeMaxEpt) * kMergeThreads;
}

template <int DT>
int launch_two_stage(const void* grad, int64_t ld_row, int rows, int V, const uint32_t* mask, int k, int64_t* idx_out,
                     uint64_t* ws, bool vec, int npow2, hipStream_t st) {
  const int s = slices_of(V);
  const dim3 grid(static_cast<unsigned>(s), static_cast<unsigned>(rows));
  if (vec)
    hipLaunchKernelGGL((topk_slice_kernel<DT, true>), grid, dim3(kSliceThreads), 0, st, grad, ld_row, V, mask, k, ws);
  else
    hipLaunchKernelGGL((topk_slice_kernel<DT, false>), grid, dim3(kSliceThreads), 0, st, grad, ld_row, V, mask, k, ws);
  const int n_in = s * k;
  int idbits = 1;
  while ((1ll << idbits) < V) ++idbits;
  const int ept = (n_in + kMergeThreads - 1) / kMergeThreads;
#define BMA_MERGE(E) hipLaunchKernelGGL((topk_merge_kernel<DT, E>), dim3(rows), dim3(kMergeThreads), 0, st, ws, n_in, idbits, k, npow2, idx_out)
  if (ept <= 2) BMA_MERGE(2);
  else if (ept <= 4) BMA_MERGE(4);
  else if (ept <= 8) BMA_MERGE(8);
  else if (ept <= 16) BMA_MERGE(16);
  else if (ept <= 20) BMA_MERGE(20);
  else BMA_MERGE(32);
#undef BMA_MERGE
  return BMA_OK;
}

template <int DT>
int launch(const void* grad, int64_t ld_row, int rows, int V, const uint32_t* mask, int k, int64_t* idx_out,
           uint64_t* ws, hipStream_t st) {
  constexpr int ES = bma::elem_bytes<DT>::value;
  int npow2 = 2;
  while (npow2 < k) npow2 <<= 1;
  const bool vec = (reinterpret_cast<uintptr_t>(grad) % 16 == 0) && ((ld_row * ES) % 16 == 0) &&
                   ((static_cast<int64_t>(V) * ES) % 16 == 0);
  BMA_PROF_BEGIN(BMA_K_TOPK, st, static_cast<double>(rows) * V * ES + static_cast<double>(rows) * k * 8.0);
  if (ws && two_stage_ok(V, k)) {
    launch_two_stage<DT>(grad, ld_row, rows, V, mask, k, idx_out, ws, vec, npow2, st);
  } else if (vec)
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

extern "C" size_t bma_mask_topk_ws_bytes(int rows, int V, int k) {
  if (rows <= 0 || V <= 0 || k <= 0 || !two_stage_ok(V, k)) return 0;
  return static_cast<size_t>(rows) * slices_of(V) * k * sizeof(uint64_t);
}

extern "C" int bma_mask_topk(const void* grad, int64_t ld_row, int rows, int V, int dtype,
                             const uint32_t* mask_bits, int k, int64_t* idx_out, void* ws, void* stream) {
  if (rows < 0 || V <= 0 || k <= 0 || k > V || ld_row < V) return BMA_EINVAL;
  if (k > kMaxK) return BMA_ELIMIT;
  if (rows == 0) return BMA_OK;
  if (!grad || !idx_out) return BMA_EINVAL;
  if (ws && reinterpret_cast<uintptr_t>(ws) % 8) return BMA_EALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  uint64_t* w = static_cast<uint64_t*>(ws);
  switch (dtype) {
    case BMA_F32: return launch<BMA_F32>(grad, ld_row, rows, V, mask_bits, k, idx_out, w, st);
    case BMA_BF16: return launch<BMA_BF16>(grad, ld_row, rows, V, mask_bits, k, idx_out, w, st);
    case BMA_F16: return launch<BMA_F16>(grad, ld_row, rows, V, mask_bits, k, idx_out, w, st);
    default: return BMA_EDTYPE;
  }
}
