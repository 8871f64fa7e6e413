// a2 -- cross-entropy over the target slice
//   candidate scoring: reference bimodal_attack.py:1289-1306 (CE "none" -> view(B,T).mean(-1),
//                      early-stop argmax test)
//   gradient pass:     reference bimodal_attack.py:1006-1012 (mean CE; autograd supplies
//                      (softmax - onehot)/T, produced here directly)
//
// HBM-bound: the B*T rows of V logits are read exactly once (B*T*V*sizeof(elem)
// algorithmic bytes; 657 MB per step at B=512, T=20, V=32064, bf16).  One
// 256-thread workgroup per row, B*T = 10,240 workgroups >> 256 CUs.  Each lane
// streams 16-byte vectors (8 bf16) with four loads in flight and keeps an online
// (max, sum-of-exp) pair in fp32; the row maximum never needs a second pass.  The
// pairs are merged across the wave with shuffles and across the four waves through
// 64 bytes of LDS.  The argmax needed by the early-stop test rides along as a
// (value, first index) pair.  A second tiny launch folds the T row losses of a
// candidate in fixed order (bitwise reproducible; no atomics).
//
// Scratch layout (`ws`, 3*B*T 4-byte words): [0,BT) row loss f32 | [BT,2BT) row
// argmax==label i32 | [2BT,3BT) row logsumexp f32 (consumed by the dlogits pass).

#include "bma_common.h"
#include "bma_profile.h"

namespace {

using bma::uint4_t;

constexpr int kThreads = 256;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kNegBig = -3.0e38f;

template <int DT>
struct V16 {
  static constexpr int NE = 16 / bma::elem_bytes<DT>::value;
  __device__ static __forceinline__ void unpack(const uint4_t& w, float* f) {
    if (DT == BMA_F32) {
      f[0] = __uint_as_float(w.x);
      f[1] = __uint_as_float(w.y);
      f[2] = __uint_as_float(w.z);
      f[3] = __uint_as_float(w.w);
    } else {
      f[0] = bma::unpack16<DT>(w.x, 0); f[1] = bma::unpack16<DT>(w.x, 1);
      f[2] = bma::unpack16<DT>(w.y, 0); f[3] = bma::unpack16<DT>(w.y, 1);
      f[4] = bma::unpack16<DT>(w.z, 0); f[5] = bma::unpack16<DT>(w.z, 1);
      f[6] = bma::unpack16<DT>(w.w, 0); f[7] = bma::unpack16<DT>(w.w, 1);
    }
  }
};

template <int DT>
__device__ __forceinline__ float load_elem(const void* base, int64_t i) {
  if (DT == BMA_F32) return static_cast<const float*>(base)[i];
  const uint32_t h = static_cast<const uint16_t*>(base)[i];
  return DT == BMA_BF16 ? bma::bf16_bits_to_f32(h) : bma::f16_bits_to_f32(h);
}

struct RowStat {
  float m, s;   // running maximum, sum of exp(x - m)
  float av;     // argmax value
  int ai;       // argmax index (first occurrence)
};

__device__ __forceinline__ void stat_init(RowStat& r) {
  r.m = kNegBig; r.s = 0.0f; r.av = -INFINITY; r.ai = 0x7fffffff;
}

template <int N, bool ARGMAX>
__device__ __forceinline__ void stat_push(RowStat& r, const float* f, int first_index) {
  float cm = f[0];
#pragma unroll
  for (int j = 1; j < N; ++j) cm = fmaxf(cm, f[j]);
  const float mn = fmaxf(r.m, cm);
  float acc = r.s * __builtin_amdgcn_exp2f((r.m - mn) * kLog2e);
#pragma unroll
  for (int j = 0; j < N; ++j) acc += __builtin_amdgcn_exp2f((f[j] - mn) * kLog2e);
  r.s = acc;
  r.m = mn;
  if (ARGMAX) {
    if (cm > r.av || (cm == r.av && first_index < r.ai)) {  // this lane's indices are not monotone across pushes
#pragma unroll
      for (int j = N - 1; j >= 0; --j)
        if (f[j] == cm) { r.ai = first_index + j; }
      r.av = cm;
    }
  }
}

__device__ __forceinline__ void stat_merge(RowStat& a, float m, float s, float av, int ai) {
  const float mn = fmaxf(a.m, m);
  a.s = a.s * __builtin_amdgcn_exp2f((a.m - mn) * kLog2e) + s * __builtin_amdgcn_exp2f((m - mn) * kLog2e);
  a.m = mn;
  if (av > a.av || (av == a.av && ai < a.ai)) { a.av = av; a.ai = ai; }
}

// One workgroup per row.  VEC: rows are 16-byte aligned and V % NE == 0.
template <int DT, bool VEC, bool ARGMAX>
__global__ __launch_bounds__(kThreads) void ce_rows_kernel(const void* __restrict__ logits, int64_t ld_cand,
                                                           int64_t ld_row, const int64_t* __restrict__ labels,
                                                           int T, int V, float* __restrict__ ws, int64_t BT) {
  const int64_t row = blockIdx.x;
  const int b = static_cast<int>(row / T), t = static_cast<int>(row % T);
  const int64_t off = static_cast<int64_t>(b) * ld_cand + static_cast<int64_t>(t) * ld_row;
  const char* base = static_cast<const char*>(logits) + off * bma::elem_bytes<DT>::value;
  const int tid = threadIdx.x;

  RowStat st;
  stat_init(st);
  if (VEC) {
    constexpr int NE = V16<DT>::NE;
    const uint4_t* p = reinterpret_cast<const uint4_t*>(base);
    const int nvec = V / NE;
    int i = tid;
    // four 16-byte loads in flight per lane (4 KiB per wave per trip)
    for (; i + 3 * kThreads < nvec; i += 4 * kThreads) {
      const uint4_t w0 = p[i], w1 = p[i + kThreads], w2 = p[i + 2 * kThreads], w3 = p[i + 3 * kThreads];
      float f[NE];
      V16<DT>::unpack(w0, f); stat_push<NE, ARGMAX>(st, f, i * NE);
      V16<DT>::unpack(w1, f); stat_push<NE, ARGMAX>(st, f, (i + kThreads) * NE);
      V16<DT>::unpack(w2, f); stat_push<NE, ARGMAX>(st, f, (i + 2 * kThreads) * NE);
      V16<DT>::unpack(w3, f); stat_push<NE, ARGMAX>(st, f, (i + 3 * kThreads) * NE);
    }
    for (; i < nvec; i += kThreads) {
      float f[NE];
      V16<DT>::unpack(p[i], f);
      stat_push<NE, ARGMAX>(st, f, i * NE);
    }
  } else {
    for (int i = tid; i < V; i += kThreads) {
      float f[1] = {load_elem<DT>(base, i)};
      stat_push<1, ARGMAX>(st, f, i);
    }
  }

  // wave merge (butterfly), then the four waves through LDS
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m = __shfl_xor(st.m, o, BMA_WAVE), s = __shfl_xor(st.s, o, BMA_WAVE);
    const float av = __shfl_xor(st.av, o, BMA_WAVE);
    const int ai = __shfl_xor(st.ai, o, BMA_WAVE);
    stat_merge(st, m, s, av, ai);
  }
  __shared__ float sm[4][4];
  const int wave = tid >> 6;
  if ((tid & 63) == 0) {
    sm[wave][0] = st.m; sm[wave][1] = st.s; sm[wave][2] = st.av; sm[wave][3] = __int_as_float(st.ai);
  }
  __syncthreads();
  if (tid == 0) {
    RowStat r;
    r.m = sm[0][0]; r.s = sm[0][1]; r.av = sm[0][2]; r.ai = __float_as_int(sm[0][3]);
#pragma unroll
    for (int w = 1; w < kThreads / 64; ++w) stat_merge(r, sm[w][0], sm[w][1], sm[w][2], __float_as_int(sm[w][3]));
    const int64_t lab = labels[t];
    const float lse = r.m + __logf(r.s);
    float xl = NAN;  // an out-of-range label poisons the loss instead of reading out of bounds
    if (lab >= 0 && lab < V) xl = load_elem<DT>(base, lab);
    ws[row] = lse - xl;
    reinterpret_cast<int32_t*>(ws)[BT + row] = ARGMAX ? (r.ai == static_cast<int>(lab) ? 1 : 0) : 0;
    ws[2 * BT + row] = lse;
  }
}

// loss[b] = (row_loss[b,0] + ... + row_loss[b,T-1]) / T, summed in row order.
__global__ __launch_bounds__(256) void ce_fold_kernel(const float* __restrict__ ws, int B, int T, int64_t BT,
                                                      float* __restrict__ loss, int32_t* __restrict__ match) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float acc = 0.0f;
  int all = 1;
  for (int t = 0; t < T; ++t) {
    acc += ws[static_cast<int64_t>(b) * T + t];
    all &= reinterpret_cast<const int32_t*>(ws)[BT + static_cast<int64_t>(b) * T + t];
  }
  loss[b] = acc / static_cast<float>(T);
  if (match) match[b] = all;
}

// dlogits[row, v] = (exp(x - lse) - [v == label]) * scale   (scale = grad_scale / T)
template <int DT, bool VEC>
__global__ __launch_bounds__(kThreads) void ce_dlogits_kernel(const void* __restrict__ logits, int64_t ld_cand,
                                                              int64_t ld_row, const int64_t* __restrict__ labels,
                                                              int T, int V, const float* __restrict__ ws, int64_t BT,
                                                              void* __restrict__ dlogits, float scale) {
  const int64_t row = blockIdx.x;
  const int b = static_cast<int>(row / T), t = static_cast<int>(row % T);
  const int64_t off = static_cast<int64_t>(b) * ld_cand + static_cast<int64_t>(t) * ld_row;
  const char* base = static_cast<const char*>(logits) + off * bma::elem_bytes<DT>::value;
  char* obase = static_cast<char*>(dlogits) + row * V * bma::elem_bytes<DT>::value;
  const float lse = ws[2 * BT + row];
  const int lab = static_cast<int>(labels[t]);
  const int tid = threadIdx.x;
  if (VEC) {
    constexpr int NE = V16<DT>::NE;
    const uint4_t* p = reinterpret_cast<const uint4_t*>(base);
    uint4_t* q = reinterpret_cast<uint4_t*>(obase);
    const int nvec = V / NE;
    for (int i = tid; i < nvec; i += kThreads) {
      float f[NE];
      V16<DT>::unpack(p[i], f);
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        const float pr = __builtin_amdgcn_exp2f((f[j] - lse) * kLog2e);
        f[j] = (pr - ((i * NE + j) == lab ? 1.0f : 0.0f)) * scale;
      }
      uint4_t w;
      if (DT == BMA_F32) {
        w.x = __float_as_uint(f[0]); w.y = __float_as_uint(f[1]);
        w.z = __float_as_uint(f[2]); w.w = __float_as_uint(f[3]);
      } else {
        w.x = bma::pack16<DT>(f[0], f[1]); w.y = bma::pack16<DT>(f[2], f[3]);
        w.z = bma::pack16<DT>(f[4 % NE], f[5 % NE]); w.w = bma::pack16<DT>(f[6 % NE], f[7 % NE]);
      }
      q[i] = w;
    }
  } else {
    for (int i = tid; i < V; i += kThreads) {
      const float x = load_elem<DT>(base, i);
      const float v = (__builtin_amdgcn_exp2f((x - lse) * kLog2e) - (i == lab ? 1.0f : 0.0f)) * scale;
      if (DT == BMA_F32) reinterpret_cast<float*>(obase)[i] = v;
      else if (DT == BMA_BF16) reinterpret_cast<uint16_t*>(obase)[i] = static_cast<uint16_t>(bma::f32_to_bf16_bits(v));
      else reinterpret_cast<uint16_t*>(obase)[i] = static_cast<uint16_t>(bma::f32_to_f16_bits(v));
    }
  }
}

template <int DT>
int launch(const void* logits, int64_t ld_cand, int64_t ld_row, const int64_t* labels, int B, int T, int V,
           float* ws, float* loss, int32_t* match, void* dlogits, float grad_scale, hipStream_t st) {
  constexpr int ES = bma::elem_bytes<DT>::value;
  const int64_t BT = static_cast<int64_t>(B) * T;
  const bool vec = (reinterpret_cast<uintptr_t>(logits) % 16 == 0) && ((ld_cand * ES) % 16 == 0) &&
                   ((ld_row * ES) % 16 == 0) && ((static_cast<int64_t>(V) * ES) % 16 == 0);
  const dim3 grid(static_cast<unsigned>(BT)), block(kThreads);
  const bool am = match != nullptr;
#define BMA_CE_GO(VEC_, AM_) \
  hipLaunchKernelGGL((ce_rows_kernel<DT, VEC_, AM_>), grid, block, 0, st, logits, ld_cand, ld_row, labels, T, V, ws, BT)
  const int prof_slot = B > 1 ? BMA_K_CE_ROWS : BMA_K_CE_ROWS_B1;
  BMA_PROF_BEGIN(prof_slot, st, static_cast<double>(BT) * V * ES);
  if (vec) { if (am) BMA_CE_GO(true, true); else BMA_CE_GO(true, false); }
  else     { if (am) BMA_CE_GO(false, true); else BMA_CE_GO(false, false); }
#undef BMA_CE_GO
  BMA_PROF_END(prof_slot, st);
  BMA_LAUNCH_CHECK();
  hipLaunchKernelGGL(ce_fold_kernel, dim3((B + 255) / 256), dim3(256), 0, st, ws, B, T, BT, loss, match);
  BMA_LAUNCH_CHECK();
  if (dlogits) {
    const bool ovec = vec && (reinterpret_cast<uintptr_t>(dlogits) % 16 == 0);
    const float scale = grad_scale / static_cast<float>(T);
    BMA_PROF_BEGIN(BMA_K_CE_DLOGITS, st, 2.0 * static_cast<double>(BT) * V * ES);
    if (ovec)
      hipLaunchKernelGGL((ce_dlogits_kernel<DT, true>), grid, block, 0, st, logits, ld_cand, ld_row, labels, T, V,
                         ws, BT, dlogits, scale);
    else
      hipLaunchKernelGGL((ce_dlogits_kernel<DT, false>), grid, block, 0, st, logits, ld_cand, ld_row, labels, T, V,
                         ws, BT, dlogits, scale);
    BMA_PROF_END(BMA_K_CE_DLOGITS, st);
    BMA_LAUNCH_CHECK();
  }
  return BMA_OK;
}

}  // namespace

extern "C" size_t bma_ce_target_ws_bytes(int B, int T) {
  if (B < 0 || T < 0) return 0;
  return static_cast<size_t>(3) * static_cast<size_t>(B) * static_cast<size_t>(T) * 4;
}

extern "C" int bma_ce_target(const void* logits, int64_t ld_cand, int64_t ld_row, const int64_t* labels, int B,
                             int T, int V, int dtype, float* ws, float* loss, int32_t* match, void* dlogits,
                             float grad_scale, void* stream) {
  if (B < 0 || T <= 0 || V <= 0 || ld_row < V || ld_cand < 0) return BMA_EINVAL;
  if (B == 0) return BMA_OK;
  if (!logits || !labels || !ws || !loss) return BMA_EINVAL;
  if (static_cast<int64_t>(B) * T > 0x7fffffffLL) return BMA_ELIMIT;
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case BMA_F32: return launch<BMA_F32>(logits, ld_cand, ld_row, labels, B, T, V, ws, loss, match, dlogits, grad_scale, st);
    case BMA_BF16: return launch<BMA_BF16>(logits, ld_cand, ld_row, labels, B, T, V, ws, loss, match, dlogits, grad_scale, st);
    case BMA_F16: return launch<BMA_F16>(logits, ld_cand, ld_row, labels, B, T, V, ws, loss, match, dlogits, grad_scale, st);
    default: return BMA_EDTYPE;
  }
}
