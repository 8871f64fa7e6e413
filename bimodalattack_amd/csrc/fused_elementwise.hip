// a6 (candidate scoring) -- the elementwise tail of a Llama-family decoder layer, fused.
//
// The reference scores candidates with `self.model(inputs_embeds=...)` (bimodal_attack.py:1287);
// inside the HuggingFace model every RMSNorm, rotary embedding and SwiGLU gate is a
// chain of 5-8 eager elementwise launches that re-read and re-write the whole activation
// (B*L x D, 184 MB at B=512, L=44, D=4096): ~40 % of the scoring phase on MI355X
// (profiles/archive/r1_bench_gcg_kernel_stats.csv).  These three kernels do each chain in ONE
// pass over HBM.  They reproduce the eager chain's ROUNDING POINTS (every intermediate the
// eager code materialises in the model dtype is rounded to the model dtype here too), so
// SwiGLU and RoPE are bit-identical to the eager modules and RMSNorm differs only through
// the summation order of the mean (<= 1 ulp of the model dtype, rarely).
//
//   rmsnorm : y = w * dt(x * rsqrt(mean(x^2) + eps))         (Llama; HF LlamaRMSNorm.forward)
//             y = dt(x * rsqrt(mean(x^2) + eps) * (1 + w))   (Gemma; HF Gemma3RMSNorm.forward)
//   swiglu  : y = dt(dt(silu(g)) * u)                        (HF LlamaMLP: act_fn(gate_proj(x)) * up_proj(x))
//   rope    : q <- dt(dt(q*cos) + dt(rotate_half(q)*sin))    (HF apply_rotary_pos_emb), in place
//
// All HBM-bound: 2 (rmsnorm, rope) or 3 (swiglu) x rows x D x es algorithmic bytes.

#include "bma_common.h"
#include "bma_profile.h"

// The eager chains round every intermediate to the model dtype.  With contraction on,
// LLVM narrows (half)((float)a*(float)b) + ... into half-precision mul+add and then fuses
// them into an FMA, silently dropping one of those roundings (seen as 1-ulp product
// differences amplified by cancellation in fp16 RoPE).  No contraction in this file.
#pragma clang fp contract(off)

namespace {

using bma::uint4_t;

template <int DT>
__device__ __forceinline__ float rnd(float v) {  // round to the model dtype and back
  if (DT == BMA_F32) return v;
  if (DT == BMA_BF16) return bma::bf16_bits_to_f32(bma::f32_to_bf16_bits(v));
  return bma::f16_bits_to_f32(bma::f32_to_f16_bits(v));
}

template <int DT>
struct Chunk {  // one 16-byte chunk = NE elements
  static constexpr int NE = 16 / bma::elem_bytes<DT>::value;
  __device__ static __forceinline__ void unpack(const uint4_t& w, float* f) {
    if (DT == BMA_F32) {
      f[0] = __uint_as_float(w.x); f[1] = __uint_as_float(w.y);
      f[2 % NE] = __uint_as_float(w.z); f[3 % NE] = __uint_as_float(w.w);
    } else {
      f[0] = bma::unpack16<DT>(w.x, 0); f[1] = bma::unpack16<DT>(w.x, 1);
      f[2] = bma::unpack16<DT>(w.y, 0); f[3] = bma::unpack16<DT>(w.y, 1);
      f[4 % NE] = bma::unpack16<DT>(w.z, 0); f[5 % NE] = bma::unpack16<DT>(w.z, 1);
      f[6 % NE] = bma::unpack16<DT>(w.w, 0); f[7 % NE] = bma::unpack16<DT>(w.w, 1);
    }
  }
  __device__ static __forceinline__ uint4_t pack(const float* f) {
    uint4_t w;
    if (DT == BMA_F32) {
      w.x = __float_as_uint(f[0]); w.y = __float_as_uint(f[1]);
      w.z = __float_as_uint(f[2 % NE]); w.w = __float_as_uint(f[3 % NE]);
    } else {
      w.x = bma::pack16<DT>(f[0], f[1]); w.y = bma::pack16<DT>(f[2], f[3]);
      w.z = bma::pack16<DT>(f[4 % NE], f[5 % NE]); w.w = bma::pack16<DT>(f[6 % NE], f[7 % NE]);
    }
    return w;
  }
};

// ---------------------------------------------------------------------------- rmsnorm
// One 256-thread workgroup per row; the row (<= 4 chunks per lane) stays in registers
// between the sum of squares and the scaling, so it is read once and written once.
constexpr int kNormThreads = 256;
constexpr int kNormMaxChunks = 4;

template <int DT, int NCH, bool GEMMA>
__global__ __launch_bounds__(kNormThreads) void rmsnorm_kernel(const uint4_t* __restrict__ x,
                                                               const uint4_t* __restrict__ w, float eps, int cpr,
                                                               int D, uint4_t* __restrict__ y) {
  constexpr int NE = Chunk<DT>::NE;
  const int64_t row = blockIdx.x;
  const int tid = threadIdx.x;
  const uint4_t* xr = x + row * cpr;
  float v[NCH][NE];
  float ss = 0.0f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int i = tid + c * kNormThreads;
    if (i < cpr) {
      Chunk<DT>::unpack(xr[i], v[c]);
#pragma unroll
      for (int j = 0; j < NE; ++j) ss += v[c][j] * v[c][j];
    }
  }
  ss = bma::wave_sum(ss);
  __shared__ float part[kNormThreads / 64];
  if ((tid & 63) == 0) part[tid >> 6] = ss;
  __syncthreads();
  float tot = 0.0f;
#pragma unroll
  for (int i = 0; i < kNormThreads / 64; ++i) tot += part[i];
  const float rstd = 1.0f / sqrtf(tot / static_cast<float>(D) + eps);
  uint4_t* yr = y + row * cpr;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int i = tid + c * kNormThreads;
    if (i < cpr) {
      float wf[NE], o[NE];
      Chunk<DT>::unpack(w[i], wf);
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        if (GEMMA) o[j] = v[c][j] * rstd * (1.0f + wf[j]);   // fp32 throughout, one rounding at the end
        else o[j] = wf[j] * rnd<DT>(v[c][j] * rstd);         // normalised x is rounded before the weight
      }
      yr[i] = Chunk<DT>::pack(o);
    }
  }
}

// Residual add + RMSNorm in one pass (HF decoder layer: `h = residual + h` followed by a norm of the sum):
//   s = dt(res + a)           a = h, or with PRE the Gemma-3 sandwich norm of h:  a = dt(h * rstd(h) * (1 + wp))
//   y = rmsnorm(s; w)         same arithmetic and rounding points as rmsnorm_kernel on s
// Both s (the next residual) and y are written: 2 reads + 2 writes per row instead of the 3 + 2 (+2 with PRE) of
// the separate launches.  Bit-identical to aten's add followed by rmsnorm_kernel (same reduction order).
template <int DT, int NCH, bool GEMMA, bool PRE>
__global__ __launch_bounds__(kNormThreads) void add_rmsnorm_kernel(const uint4_t* __restrict__ res,
                                                                   const uint4_t* __restrict__ h,
                                                                   const uint4_t* __restrict__ wp, float eps_pre,
                                                                   const uint4_t* __restrict__ w, float eps, int cpr,
                                                                   int D, uint4_t* __restrict__ s_out,
                                                                   uint4_t* __restrict__ y) {
  constexpr int NE = Chunk<DT>::NE;
  const int64_t row = blockIdx.x;
  const int tid = threadIdx.x;
  __shared__ float part[2][kNormThreads / 64];
  float v[NCH][NE];
  if (PRE) {
    float ss = 0.0f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int i = tid + c * kNormThreads;
      if (i < cpr) {
        Chunk<DT>::unpack(h[row * cpr + i], v[c]);
#pragma unroll
        for (int j = 0; j < NE; ++j) ss += v[c][j] * v[c][j];
      }
    }
    ss = bma::wave_sum(ss);
    if ((tid & 63) == 0) part[0][tid >> 6] = ss;
    __syncthreads();
    float tot = 0.0f;
#pragma unroll
    for (int i = 0; i < kNormThreads / 64; ++i) tot += part[0][i];
    const float rstd = 1.0f / sqrtf(tot / static_cast<float>(D) + eps_pre);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int i = tid + c * kNormThreads;
      if (i < cpr) {
        float wf[NE];
        Chunk<DT>::unpack(wp[i], wf);
#pragma unroll
        for (int j = 0; j < NE; ++j) {
          if (GEMMA) v[c][j] = rnd<DT>(v[c][j] * rstd * (1.0f + wf[j]));
          else v[c][j] = rnd<DT>(wf[j] * rnd<DT>(v[c][j] * rstd));
        }
      }
    }
  }
  float ss = 0.0f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int i = tid + c * kNormThreads;
    if (i < cpr) {
      float r[NE];
      Chunk<DT>::unpack(res[row * cpr + i], r);
      if (!PRE) Chunk<DT>::unpack(h[row * cpr + i], v[c]);
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        v[c][j] = rnd<DT>(r[j] + v[c][j]);            // the sum as the eager add leaves it in the model dtype
        ss += v[c][j] * v[c][j];
      }
      s_out[row * cpr + i] = Chunk<DT>::pack(v[c]);
    }
  }
  ss = bma::wave_sum(ss);
  if ((tid & 63) == 0) part[1][tid >> 6] = ss;
  __syncthreads();
  float tot = 0.0f;
#pragma unroll
  for (int i = 0; i < kNormThreads / 64; ++i) tot += part[1][i];
  const float rstd = 1.0f / sqrtf(tot / static_cast<float>(D) + eps);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int i = tid + c * kNormThreads;
    if (i < cpr) {
      float wf[NE], o[NE];
      Chunk<DT>::unpack(w[i], wf);
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        if (GEMMA) o[j] = v[c][j] * rstd * (1.0f + wf[j]);
        else o[j] = wf[j] * rnd<DT>(v[c][j] * rstd);
      }
      y[row * cpr + i] = Chunk<DT>::pack(o);
    }
  }
}

// Short rows (per-head q/k norms: D = 128 or 256): LPR lanes per row, 256/LPR rows per workgroup,
// one chunk per lane, the sum of squares reduced inside the LPR-lane group with xor shuffles.
template <int DT, int LPR, bool GEMMA>
__global__ __launch_bounds__(kNormThreads) void rmsnorm_short_kernel(const uint4_t* __restrict__ x,
                                                                     const uint4_t* __restrict__ w, float eps,
                                                                     int64_t rows, int cpr, int D,
                                                                     uint4_t* __restrict__ y) {
  constexpr int NE = Chunk<DT>::NE;
  constexpr int RPB = kNormThreads / LPR;
  const int tid = threadIdx.x;
  const int sub = tid % LPR;
  const int64_t row = static_cast<int64_t>(blockIdx.x) * RPB + tid / LPR;
  const bool live = row < rows && sub < cpr;
  float v[NE];
  float ss = 0.0f;
  if (live) {
    Chunk<DT>::unpack(x[row * cpr + sub], v);
#pragma unroll
    for (int j = 0; j < NE; ++j) ss += v[j] * v[j];
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, BMA_WAVE);
  if (!live) return;
  const float rstd = 1.0f / sqrtf(ss / static_cast<float>(D) + eps);
  float wf[NE], o[NE];
  Chunk<DT>::unpack(w[sub], wf);
#pragma unroll
  for (int j = 0; j < NE; ++j) {
    if (GEMMA) o[j] = v[j] * rstd * (1.0f + wf[j]);
    else o[j] = wf[j] * rnd<DT>(v[j] * rstd);
  }
  y[row * cpr + sub] = Chunk<DT>::pack(o);
}

// ---------------------------------------------------------------------------- swiglu
#ifdef BMA_FAST_SILU
#define BMA_SILU(x) __fdividef((x), 1.0f + __expf(-(x)))
#else
#define BMA_SILU(x) ((x) / (1.0f + expf(-(x))))   // accurate exp + IEEE division, as aten's silu kernel
#endif

// gelu(x, approximate="tanh") exactly as the main loop of aten's device kernel evaluates it in
// fp32 (including the fused multiply-add its compiler forms for x + kappa*x^3): Gemma's gated
// MLP.  (aten's partial last block is compiled with one more contraction and is 1 fp32 ulp off
// its own main loop; tests/test_fused_gpu.py pins both facts.)
__device__ __forceinline__ float gelu_tanh(float x) {
  constexpr float kBeta = 0.7978845608028654f;     // sqrt(2/pi)
  constexpr float kKappa = 0.044715f;
  const float x3 = x * x * x;
  const float inner = kBeta * fmaf(kKappa, x3, x);
  return 0.5f * x * (1.0f + tanhf(inner));
}

// ACT: 0 = SiLU (Llama family), 1 = GELU-tanh (Gemma family)
template <int DT, int ACT>
__device__ __forceinline__ uint4_t swiglu_chunk(const uint4_t& gw, const uint4_t& uw) {
  constexpr int NE = Chunk<DT>::NE;
  float gf[NE], uf[NE], o[NE];
  Chunk<DT>::unpack(gw, gf);
  Chunk<DT>::unpack(uw, uf);
#pragma unroll
  for (int j = 0; j < NE; ++j) {
    const float s = ACT == 0 ? BMA_SILU(gf[j]) : gelu_tanh(gf[j]);
    o[j] = rnd<DT>(s) * uf[j];
  }
  return Chunk<DT>::pack(o);
}

// Each workgroup owns ONE contiguous slab of kSwiChunks*256 chunks (16 KiB per array at 4 chunks
// per lane) and there is no grid-stride loop: all eight loads of a lane are issued before the
// first use, and the DRAM pages a workgroup touches are adjacent.  (A grid-stride version of this
// kernel ran at 4.4 TB/s where this layout -- the one torch's elementwise kernels use -- reaches
// the 6 TB/s a 2-read + 1-write stream gets on this part.)
constexpr int kSwiChunks = 4;

// IL: gate and up come as alternating 16-byte chunks of ONE array (the output of the gate_proj/up_proj
// product against their chunk-interleaved weights, fused.py): chunk i of the output reads chunks 2i and 2i+1.
template <int DT, int ACT, bool IL>
__global__ __launch_bounds__(256) void swiglu_kernel(const uint4_t* __restrict__ g, const uint4_t* __restrict__ u,
                                                     int64_t n_chunks, uint4_t* __restrict__ y) {
  const int64_t base = static_cast<int64_t>(blockIdx.x) * (kSwiChunks * 256) + threadIdx.x;
  constexpr int S = IL ? 2 : 1;
  if (IL) u = g + 1;
  if (base + (kSwiChunks - 1) * 256 < n_chunks) {
    uint4_t gw[kSwiChunks], uw[kSwiChunks];
#pragma unroll
    for (int j = 0; j < kSwiChunks; ++j) { gw[j] = g[S * (base + j * 256)]; uw[j] = u[S * (base + j * 256)]; }
#pragma unroll
    for (int j = 0; j < kSwiChunks; ++j) y[base + j * 256] = swiglu_chunk<DT, ACT>(gw[j], uw[j]);
  } else {
#pragma unroll
    for (int j = 0; j < kSwiChunks; ++j) {
      const int64_t i = base + j * 256;
      if (i < n_chunks) y[i] = swiglu_chunk<DT, ACT>(g[S * i], u[S * i]);
    }
  }
}

// ---------------------------------------------------------------------------- QuickGELU (CLIP's MLP)
// y = x * sigmoid(1.702 x) as HuggingFace's QuickGELUActivation evaluates it on a 16-bit tensor -- three aten kernels,
// each rounding to the tensor dtype: t = rnd(1.702 x), s = rnd(sigmoid(t)), y = rnd(x s) -- and the five kernels of its
// autograd backward -- dy s, dy x, sigmoid_backward, the scalar product, the accumulation -- in one launch each, with
// the roundings where aten has them (bit-identical; tests/test_fused_gpu.py).
#define BMA_SIGMOID(x) (1.0f / (1.0f + expf(-(x))))   // accurate exp + IEEE division, as aten's sigmoid kernel
template <int DT, bool BWD>
__global__ __launch_bounds__(256) void quick_gelu_kernel(const uint4_t* __restrict__ x, const uint4_t* __restrict__ dy,
                                                         int64_t n_chunks, uint4_t* __restrict__ out) {
  constexpr int NE = Chunk<DT>::NE;
  const int64_t base = static_cast<int64_t>(blockIdx.x) * (kSwiChunks * 256) + threadIdx.x;
#pragma unroll
  for (int j = 0; j < kSwiChunks; ++j) {
    const int64_t i = base + j * 256;
    if (i >= n_chunks) continue;
    float xf[NE], gf[NE], o[NE];
    Chunk<DT>::unpack(x[i], xf);
    if (BWD) Chunk<DT>::unpack(dy[i], gf);
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const float t = rnd<DT>(1.702f * xf[e]);
      const float sg = rnd<DT>(BMA_SIGMOID(t));
      if (!BWD) {
        o[e] = xf[e] * sg;
      } else {
        const float g1 = rnd<DT>(gf[e] * sg);                 // d/dx through the product's first factor
        const float gs = rnd<DT>(gf[e] * xf[e]);              // gradient arriving at the sigmoid
        // aten's sigmoid_backward, `a * (scalar_t(1) - b) * b`, is written on the TENSOR type: every operator rounds
        const float gt = rnd<DT>(rnd<DT>(gs * rnd<DT>(1.0f - sg)) * sg);
        const float g2 = rnd<DT>(gt * 1.702f);
        o[e] = g1 + g2;
      }
    }
    out[i] = Chunk<DT>::pack(o);
  }
}

static int quick_gelu_launch(const void* x, const void* dy, bool bwd, int64_t n, int dtype, void* out, void* stream) {
  if (n < 0) return BMA_EINVAL;
  if (n == 0) return BMA_OK;
  if (!x || (bwd && !dy) || !out) return BMA_EINVAL;
  if (dtype != BMA_F32 && dtype != BMA_BF16 && dtype != BMA_F16) return BMA_EDTYPE;
  const int es = dtype == BMA_F32 ? 4 : 2;
  if ((n * es) % 16) return BMA_EALIGN;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(out)) % 16) return BMA_EALIGN;
  const int64_t chunks = n * es / 16;
  const int64_t blocks = (chunks + kSwiChunks * 256 - 1) / (kSwiChunks * 256);
  if (blocks > 0x7fffffffLL) return BMA_ELIMIT;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(static_cast<unsigned>(blocks)), block(256);
  const uint4_t* xp = static_cast<const uint4_t*>(x);
  const uint4_t* gp = static_cast<const uint4_t*>(dy);
  uint4_t* yp = static_cast<uint4_t*>(out);
#define BMA_QG_GO(DT_)                                                                                     \
  do {                                                                                                     \
    if (bwd) hipLaunchKernelGGL((quick_gelu_kernel<DT_, true>), grid, block, 0, st, xp, gp, chunks, yp);    \
    else hipLaunchKernelGGL((quick_gelu_kernel<DT_, false>), grid, block, 0, st, xp, gp, chunks, yp);       \
  } while (0)
  if (dtype == BMA_F32) BMA_QG_GO(BMA_F32);
  else if (dtype == BMA_BF16) BMA_QG_GO(BMA_BF16);
  else BMA_QG_GO(BMA_F16);
#undef BMA_QG_GO
  BMA_LAUNCH_CHECK();
  return BMA_OK;
}

// ---------------------------------------------------------------------------- rope
// q is addressed through strides (elements): element (b,h,l,d) at q + b*sb + h*sh + l*sl + d.
// One lane owns ONE 16-byte chunk of a head vector; the chunk of the other half of the head
// (its rotate_half partner) lives in the lane `cph/2` lanes away, and the two exchange their
// values with one shuffle.  Consecutive lanes therefore walk whole head vectors: every wave
// instruction loads/stores 1 KiB of contiguous memory when heads are adjacent (a projection
// output), and the update is safely in place.  cos/sin: [cb][L][Dh], cb = 1 or B.
// Requires cph = Dh*es/16 to be a power of two <= 64 (Dh = 64..512 for 16-bit dtypes).
// Sum over the aligned group of `cph` lanes (a power of two <= 64) that hold one head: the xor butterfly of
// rmsnorm_short_kernel -- partners cph/2, ..., 2, 1 lanes away, in that order, so the sum is the same fp32 number -- but
// without the LDS crossbar: the 32- and 16-lane exchanges are gfx950's row swaps, and below 16 a ROTATION by 8, 4, 2, 1
// inside the 16-lane row (one DPP add each) meets the same partner value, because after the step before it the partial
// sums repeat with that period (heads of at least 16 lanes; narrower ones keep the shuffles).  Five ds_bpermute round trips in a kernel with one 16-byte load per lane were most of a
// workgroup's life.
__device__ __forceinline__ float head_sum(float ss, int cph) {
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  if (cph < 16) {                               // (a head narrower than a 16-lane row: the rotations would reach into its neighbour)
    for (int o = cph >> 1; o > 0; o >>= 1) ss += __shfl_xor(ss, o, BMA_WAVE);
    return ss;
  }
  if (cph >= 64) {
    const u32x2_t p = __builtin_amdgcn_permlane32_swap(__float_as_uint(ss), __float_as_uint(ss), false, false);
    ss = __uint_as_float(p.x) + __uint_as_float(p.y);
  }
  if (cph >= 32) {
    const u32x2_t p = __builtin_amdgcn_permlane16_swap(__float_as_uint(ss), __float_as_uint(ss), false, false);
    ss = __uint_as_float(p.x) + __uint_as_float(p.y);
  }
  ss += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(ss), 0x128, 0xf, 0xf, false));   // row_ror:8
  ss += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(ss), 0x124, 0xf, 0xf, false));   // row_ror:4
  ss += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(ss), 0x122, 0xf, 0xf, false));   // row_ror:2
  ss += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(ss), 0x121, 0xf, 0xf, false));   // row_ror:1
  return ss;
}

// NORM: the per-head RMSNorm of Gemma-3's q_norm / k_norm (weight `nw` [Dh], `neps`, `ngemma` = the (1 + w) form) is
// applied to the head vector first, with rmsnorm_short_kernel's arithmetic -- the lane's eight squares, then the xor
// shuffles over the head's lanes, the normalised value rounded to the model dtype -- so the result is, bit for bit, that
// kernel followed by the rotation, in one pass over q and k instead of two.
template <int DT, bool NORM = false>
__device__ __forceinline__ void rope_row(const void* q, int64_t sb, int64_t sh, int64_t sl, void* dst,
                                         int64_t db, int64_t dh, int64_t dl, int B, int H, int L, int Dh,
                                         const void* __restrict__ cosp, const void* __restrict__ sinp,
                                         int cos_batch, int cph_log2, float sin_sign,
                                         const void* __restrict__ nw = nullptr, float neps = 0.0f, int ngemma = 0) {
  constexpr int NE = Chunk<DT>::NE;
  constexpr int ES = bma::elem_bytes<DT>::value;
  const int cph = 1 << cph_log2;               // chunks per head vector
  const int half = cph >> 1;
  // one workgroup per (b, l): no per-chunk division, only shifts and masks
  const int row = blockIdx.x;
  const int b = row / L, l = row - b * L;
  const char* qrow = static_cast<const char*>(q) + (static_cast<int64_t>(b) * sb + static_cast<int64_t>(l) * sl) * ES;
  char* drow = static_cast<char*>(dst) + (static_cast<int64_t>(b) * db + static_cast<int64_t>(l) * dl) * ES;
  const int64_t cs = (static_cast<int64_t>(cos_batch > 1 ? b : 0) * L + l) * Dh;
  const uint4_t* crow = reinterpret_cast<const uint4_t*>(static_cast<const char*>(cosp) + cs * ES);
  const uint4_t* srow = reinterpret_cast<const uint4_t*>(static_cast<const char*>(sinp) + cs * ES);
  const int n = H << cph_log2;                 // chunks in this row (a multiple of cph: whole heads per wave)
  for (int i = threadIdx.x; i < n; i += 256) {
    const int c = i & (cph - 1);
    const int h = i >> cph_log2;
    const uint4_t* px = reinterpret_cast<const uint4_t*>(qrow + static_cast<int64_t>(h) * sh * ES) + c;
    uint4_t* pd = reinterpret_cast<uint4_t*>(drow + static_cast<int64_t>(h) * dh * ES) + c;
    const uint4_t cw = crow[c], sw = srow[c];
    uint4_t xw = *px;
    if (NORM) {
      float v[NE], wf[NE], o[NE];
      Chunk<DT>::unpack(xw, v);
      float ss = 0.0f;
#pragma unroll
      for (int j = 0; j < NE; ++j) ss += v[j] * v[j];
      ss = head_sum(ss, cph);
      const float rstd = 1.0f / sqrtf(ss / static_cast<float>(Dh) + neps);
      Chunk<DT>::unpack(reinterpret_cast<const uint4_t*>(nw)[c], wf);
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        if (ngemma) o[j] = v[j] * rstd * (1.0f + wf[j]);
        else o[j] = wf[j] * rnd<DT>(v[j] * rstd);
      }
      xw = Chunk<DT>::pack(o);
    }
    uint4_t pw;                                 // the partner half's chunk
    pw.x = __shfl_xor(xw.x, half, BMA_WAVE);
    pw.y = __shfl_xor(xw.y, half, BMA_WAVE);
    pw.z = __shfl_xor(xw.z, half, BMA_WAVE);
    pw.w = __shfl_xor(xw.w, half, BMA_WAVE);
    float x[NE], p[NE], cf[NE], sf[NE], o[NE];
    Chunk<DT>::unpack(xw, x);
    Chunk<DT>::unpack(pw, p);
    Chunk<DT>::unpack(cw, cf);
    Chunk<DT>::unpack(sw, sf);
    // rotate_half(x) = cat(-x2, x1); sin_sign = -1 turns the rotation into its inverse (the backward)
    const float sign = ((c < half) ? -1.0f : 1.0f) * sin_sign;
#pragma unroll
    for (int j = 0; j < NE; ++j) o[j] = rnd<DT>(rnd<DT>(x[j] * cf[j]) + rnd<DT>(sign * p[j] * sf[j]));
    *pd = Chunk<DT>::pack(o);
  }
}

template <int DT>
__global__ __launch_bounds__(256) void rope_kernel(const void* q, int64_t sb, int64_t sh, int64_t sl, void* dst,
                                                   int64_t db, int64_t dh, int64_t dl, int B, int H, int L, int Dh,
                                                   const void* __restrict__ cosp, const void* __restrict__ sinp,
                                                   int cos_batch, int cph_log2, float sin_sign) {
  rope_row<DT>(q, sb, sh, sl, dst, db, dh, dl, B, H, L, Dh, cosp, sinp, cos_batch, cph_log2, sin_sign);
}

// q AND k of one attention block in ONE launch (blockIdx.y picks the tensor): they share cos/sin, B, L and Dh and
// differ in base pointer, strides and head count (grouped key/value heads).
struct RopeTensor {
  const void* src;
  void* dst;
  int64_t sb, sh, sl, db, dh, dl;
  int H;
};

template <int DT>
__global__ __launch_bounds__(256) void rope2_kernel(RopeTensor tq, RopeTensor tk, int B, int L, int Dh,
                                                    const void* __restrict__ cosp, const void* __restrict__ sinp,
                                                    int cos_batch, int cph_log2, float sin_sign) {
  const RopeTensor& t = blockIdx.y ? tk : tq;
  rope_row<DT>(t.src, t.sb, t.sh, t.sl, t.dst, t.db, t.dh, t.dl, B, t.H, L, Dh, cosp, sinp, cos_batch, cph_log2, sin_sign);
}

template <int DT>
__global__ __launch_bounds__(256) void qknorm_rope2_kernel(RopeTensor tq, RopeTensor tk, int B, int L, int Dh,
                                                           const void* __restrict__ cosp, const void* __restrict__ sinp,
                                                           int cos_batch, int cph_log2, const void* __restrict__ wq,
                                                           const void* __restrict__ wk, float eps, int gemma) {
  const RopeTensor& t = blockIdx.y ? tk : tq;
  rope_row<DT, true>(t.src, t.sb, t.sh, t.sl, t.dst, t.db, t.dh, t.dl, B, t.H, L, Dh, cosp, sinp, cos_batch, cph_log2, 1.0f,
                     blockIdx.y ? wk : wq, eps, gemma);
}

template <int DT>
int launch_rmsnorm(const void* x, const void* w, float eps, int64_t rows, int D, int gemma, void* y, hipStream_t st) {
  constexpr int ES = bma::elem_bytes<DT>::value;
  const int cpr = static_cast<int>(static_cast<int64_t>(D) * ES / 16);
  const int nch = (cpr + kNormThreads - 1) / kNormThreads;
  if (nch > kNormMaxChunks) return BMA_ELIMIT;
  const dim3 grid(static_cast<unsigned>(rows)), block(kNormThreads);
  const uint4_t* xp = static_cast<const uint4_t*>(x);
  const uint4_t* wp = static_cast<const uint4_t*>(w);
  uint4_t* yp = static_cast<uint4_t*>(y);
  BMA_PROF_BEGIN(BMA_K_RMSNORM, st, 2.0 * static_cast<double>(rows) * D * ES);
  if (cpr <= 64) {
#define BMA_NORM_SHORT(L)                                                                                          \
  do {                                                                                                             \
    const dim3 g(static_cast<unsigned>((rows + kNormThreads / L - 1) / (kNormThreads / L)));                        \
    if (gemma) hipLaunchKernelGGL((rmsnorm_short_kernel<DT, L, true>), g, block, 0, st, xp, wp, eps, rows, cpr, D, yp); \
    else hipLaunchKernelGGL((rmsnorm_short_kernel<DT, L, false>), g, block, 0, st, xp, wp, eps, rows, cpr, D, yp); \
  } while (0)
    if (cpr <= 8) BMA_NORM_SHORT(8);
    else if (cpr <= 16) BMA_NORM_SHORT(16);
    else if (cpr <= 32) BMA_NORM_SHORT(32);
    else BMA_NORM_SHORT(64);
#undef BMA_NORM_SHORT
    BMA_PROF_END(BMA_K_RMSNORM, st);
    BMA_LAUNCH_CHECK();
    return BMA_OK;
  }
#define BMA_NORM_GO(N)                                                                                          \
  do {                                                                                                          \
    if (gemma) hipLaunchKernelGGL((rmsnorm_kernel<DT, N, true>), grid, block, 0, st, xp, wp, eps, cpr, D, yp);   \
    else hipLaunchKernelGGL((rmsnorm_kernel<DT, N, false>), grid, block, 0, st, xp, wp, eps, cpr, D, yp);        \
  } while (0)
  switch (nch) {
    case 1: BMA_NORM_GO(1); break;
    case 2: BMA_NORM_GO(2); break;
    case 3: BMA_NORM_GO(3); break;
    default: BMA_NORM_GO(4); break;
  }
#undef BMA_NORM_GO
  BMA_PROF_END(BMA_K_RMSNORM, st);
  BMA_LAUNCH_CHECK();
  return BMA_OK;
}

}  // namespace

extern "C" int bma_rmsnorm(const void* x, const void* weight, float eps, int64_t rows, int D, int dtype,
                           int gemma_style, void* out, void* stream) {
  if (rows < 0 || D <= 0 || rows > 0x7fffffffLL) return BMA_EINVAL;
  if (rows == 0) return BMA_OK;
  if (!x || !weight || !out) return BMA_EINVAL;
  const int es = dtype == BMA_F32 ? 4 : 2;
  if ((static_cast<int64_t>(D) * es) % 16) return BMA_EALIGN;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(weight) | reinterpret_cast<uintptr_t>(out)) % 16)
    return BMA_EALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case BMA_F32: return launch_rmsnorm<BMA_F32>(x, weight, eps, rows, D, gemma_style, out, st);
    case BMA_BF16: return launch_rmsnorm<BMA_BF16>(x, weight, eps, rows, D, gemma_style, out, st);
    case BMA_F16: return launch_rmsnorm<BMA_F16>(x, weight, eps, rows, D, gemma_style, out, st);
    default: return BMA_EDTYPE;
  }
}

namespace {
template <int DT>
int launch_add_rmsnorm(const void* res, const void* h, const void* wp, float eps_pre, const void* w, float eps,
                       int64_t rows, int D, int gemma, void* s_out, void* y, hipStream_t st) {
  constexpr int ES = bma::elem_bytes<DT>::value;
  const int cpr = static_cast<int>(static_cast<int64_t>(D) * ES / 16);
  const int nch = (cpr + kNormThreads - 1) / kNormThreads;
  if (nch > kNormMaxChunks) return BMA_ELIMIT;
  const dim3 grid(static_cast<unsigned>(rows)), block(kNormThreads);
  const uint4_t* rp = static_cast<const uint4_t*>(res);
  const uint4_t* hp = static_cast<const uint4_t*>(h);
  const uint4_t* pp = static_cast<const uint4_t*>(wp);
  const uint4_t* wq = static_cast<const uint4_t*>(w);
  uint4_t* sp = static_cast<uint4_t*>(s_out);
  uint4_t* yp = static_cast<uint4_t*>(y);
  BMA_PROF_BEGIN(BMA_K_ADD_RMSNORM, st, 4.0 * static_cast<double>(rows) * D * ES);
#define BMA_AN_GO(N)                                                                                                      \
  do {                                                                                                                    \
    if (gemma && wp) hipLaunchKernelGGL((add_rmsnorm_kernel<DT, N, true, true>), grid, block, 0, st, rp, hp, pp, eps_pre, wq, eps, cpr, D, sp, yp);   \
    else if (gemma) hipLaunchKernelGGL((add_rmsnorm_kernel<DT, N, true, false>), grid, block, 0, st, rp, hp, pp, eps_pre, wq, eps, cpr, D, sp, yp);   \
    else if (wp) hipLaunchKernelGGL((add_rmsnorm_kernel<DT, N, false, true>), grid, block, 0, st, rp, hp, pp, eps_pre, wq, eps, cpr, D, sp, yp);      \
    else hipLaunchKernelGGL((add_rmsnorm_kernel<DT, N, false, false>), grid, block, 0, st, rp, hp, pp, eps_pre, wq, eps, cpr, D, sp, yp);             \
  } while (0)
  switch (nch) {
    case 1: BMA_AN_GO(1); break;
    case 2: BMA_AN_GO(2); break;
    case 3: BMA_AN_GO(3); break;
    default: BMA_AN_GO(4); break;
  }
#undef BMA_AN_GO
  BMA_PROF_END(BMA_K_ADD_RMSNORM, st);
  BMA_LAUNCH_CHECK();
  return BMA_OK;
}
}  // namespace

extern "C" int bma_add_rmsnorm(const void* residual, const void* h, const void* pre_weight, float pre_eps,
                               const void* weight, float eps, int64_t rows, int D, int dtype, int gemma_style,
                               void* sum_out, void* out, void* stream) {
  if (rows < 0 || D <= 0 || rows > 0x7fffffffLL) return BMA_EINVAL;
  if (rows == 0) return BMA_OK;
  if (!residual || !h || !weight || !sum_out || !out) return BMA_EINVAL;
  const int es = dtype == BMA_F32 ? 4 : 2;
  if ((static_cast<int64_t>(D) * es) % 16) return BMA_EALIGN;
  if ((reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(pre_weight) |
       reinterpret_cast<uintptr_t>(weight) | reinterpret_cast<uintptr_t>(sum_out) | reinterpret_cast<uintptr_t>(out)) % 16)
    return BMA_EALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case BMA_F32: return launch_add_rmsnorm<BMA_F32>(residual, h, pre_weight, pre_eps, weight, eps, rows, D, gemma_style, sum_out, out, st);
    case BMA_BF16: return launch_add_rmsnorm<BMA_BF16>(residual, h, pre_weight, pre_eps, weight, eps, rows, D, gemma_style, sum_out, out, st);
    case BMA_F16: return launch_add_rmsnorm<BMA_F16>(residual, h, pre_weight, pre_eps, weight, eps, rows, D, gemma_style, sum_out, out, st);
    default: return BMA_EDTYPE;
  }
}

static int gated_act_launch(const void* gate, const void* up, bool interleaved, int64_t n, int dtype, int act, void* out,
                            void* stream) {
  if (n < 0 || (act != 0 && act != 1)) return BMA_EINVAL;
  if (n == 0) return BMA_OK;
  if (!gate || (!interleaved && !up) || !out) return BMA_EINVAL;
  if (dtype != BMA_F32 && dtype != BMA_BF16 && dtype != BMA_F16) return BMA_EDTYPE;
  const int es = dtype == BMA_F32 ? 4 : 2;
  if ((n * es) % 16) return BMA_EALIGN;
  if ((reinterpret_cast<uintptr_t>(gate) | reinterpret_cast<uintptr_t>(up) | reinterpret_cast<uintptr_t>(out)) % 16)
    return BMA_EALIGN;
  const int64_t chunks = n * es / 16;
  const int64_t blocks = (chunks + kSwiChunks * 256 - 1) / (kSwiChunks * 256);
  if (blocks > 0x7fffffffLL) return BMA_ELIMIT;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(static_cast<unsigned>(blocks)), block(256);
  const uint4_t* g = static_cast<const uint4_t*>(gate);
  const uint4_t* u = static_cast<const uint4_t*>(up);
  uint4_t* y = static_cast<uint4_t*>(out);
  BMA_PROF_BEGIN(BMA_K_SWIGLU, st, 3.0 * static_cast<double>(n) * es);
#define BMA_GA_GO(DT_)                                                                                          \
  do {                                                                                                          \
    if (interleaved) {                                                                                          \
      if (act == 0) hipLaunchKernelGGL((swiglu_kernel<DT_, 0, true>), grid, block, 0, st, g, u, chunks, y);      \
      else hipLaunchKernelGGL((swiglu_kernel<DT_, 1, true>), grid, block, 0, st, g, u, chunks, y);               \
    } else {                                                                                                    \
      if (act == 0) hipLaunchKernelGGL((swiglu_kernel<DT_, 0, false>), grid, block, 0, st, g, u, chunks, y);     \
      else hipLaunchKernelGGL((swiglu_kernel<DT_, 1, false>), grid, block, 0, st, g, u, chunks, y);              \
    }                                                                                                           \
  } while (0)
  if (dtype == BMA_F32) BMA_GA_GO(BMA_F32);
  else if (dtype == BMA_BF16) BMA_GA_GO(BMA_BF16);
  else BMA_GA_GO(BMA_F16);
#undef BMA_GA_GO
  BMA_PROF_END(BMA_K_SWIGLU, st);
  BMA_LAUNCH_CHECK();
  return BMA_OK;
}

extern "C" int bma_gated_act(const void* gate, const void* up, int64_t n, int dtype, int act, void* out,
                             void* stream) {
  return gated_act_launch(gate, up, false, n, dtype, act, out, stream);
}

extern "C" int bma_gated_act_il(const void* gate_up, int64_t n, int dtype, int act, void* out, void* stream) {
  return gated_act_launch(gate_up, nullptr, true, n, dtype, act, out, stream);
}

extern "C" int bma_quick_gelu(const void* x, int64_t n, int dtype, void* out, void* stream) {
  return quick_gelu_launch(x, nullptr, false, n, dtype, out, stream);
}

extern "C" int bma_quick_gelu_bwd(const void* x, const void* dy, int64_t n, int dtype, void* dx, void* stream) {
  return quick_gelu_launch(x, dy, true, n, dtype, dx, stream);
}

extern "C" int bma_swiglu(const void* gate, const void* up, int64_t n, int dtype, void* out, void* stream) {
  return bma_gated_act(gate, up, n, dtype, 0, out, stream);
}

extern "C" int bma_rope(const void* q, int64_t stride_b, int64_t stride_h, int64_t stride_l, void* dst, int64_t dst_b,
                        int64_t dst_h, int64_t dst_l, int B, int H, int L, int Dh, const void* cos, const void* sin,
                        int cos_batch, float sin_sign, int dtype, void* stream) {
  if (B < 0 || H <= 0 || L < 0 || Dh <= 0 || (cos_batch != 1 && cos_batch != B)) return BMA_EINVAL;
  if (sin_sign != 1.0f && sin_sign != -1.0f) return BMA_EINVAL;
  if (B == 0 || L == 0) return BMA_OK;
  if (!q || !dst || !cos || !sin) return BMA_EINVAL;
  if (dtype != BMA_F32 && dtype != BMA_BF16 && dtype != BMA_F16) return BMA_EDTYPE;
  const int es = dtype == BMA_F32 ? 4 : 2;
  const int ne = 16 / es;
  if (Dh % (2 * ne)) return BMA_EALIGN;                       // half a head must be whole 16-byte chunks
  const int cph_host = Dh / ne;
  if (cph_host > 64 || (cph_host & (cph_host - 1))) return BMA_ELIMIT;   // partner exchange stays inside a wave
  if ((stride_b * es) % 16 || (stride_h * es) % 16 || (stride_l * es) % 16) return BMA_EALIGN;
  if ((dst_b * es) % 16 || (dst_h * es) % 16 || (dst_l * es) % 16) return BMA_EALIGN;
  if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(cos) |
       reinterpret_cast<uintptr_t>(sin)) % 16)
    return BMA_EALIGN;
  int cph_log2 = 0;
  while ((1 << cph_log2) < cph_host) ++cph_log2;
  // every lane of a wave must be active while partners exchange values: H*cph is a multiple of
  // cph and 256 % cph == 0, so a head vector never straddles the loop's tail
  const int64_t rows = static_cast<int64_t>(B) * L;
  if (rows > 0x7fffffffLL) return BMA_ELIMIT;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(static_cast<unsigned>(rows)), block(256);
  BMA_PROF_BEGIN(BMA_K_ROPE, st, 2.0 * static_cast<double>(B) * H * L * Dh * es);
#define BMA_ROPE_GO(DT_)                                                                                            \
  hipLaunchKernelGGL((rope_kernel<DT_>), grid, block, 0, st, q, stride_b, stride_h, stride_l, dst, dst_b, dst_h, dst_l, \
                     B, H, L, Dh, cos, sin, cos_batch, cph_log2, sin_sign)
  if (dtype == BMA_F32) BMA_ROPE_GO(BMA_F32);
  else if (dtype == BMA_BF16) BMA_ROPE_GO(BMA_BF16);
  else BMA_ROPE_GO(BMA_F16);
#undef BMA_ROPE_GO
  BMA_PROF_END(BMA_K_ROPE, st);
  BMA_LAUNCH_CHECK();
  return BMA_OK;
}

extern "C" int bma_rope2(const void* q, int64_t q_b, int64_t q_h, int64_t q_l, void* qd, int64_t qd_b, int64_t qd_h,
                         int64_t qd_l, int Hq, const void* k, int64_t k_b, int64_t k_h, int64_t k_l, void* kd,
                         int64_t kd_b, int64_t kd_h, int64_t kd_l, int Hk, int B, int L, int Dh, const void* cos,
                         const void* sin, int cos_batch, float sin_sign, int dtype, void* stream) {
  if (B < 0 || Hq <= 0 || Hk <= 0 || L < 0 || Dh <= 0 || (cos_batch != 1 && cos_batch != B)) return BMA_EINVAL;
  if (sin_sign != 1.0f && sin_sign != -1.0f) return BMA_EINVAL;
  if (B == 0 || L == 0) return BMA_OK;
  if (!q || !qd || !k || !kd || !cos || !sin) return BMA_EINVAL;
  if (dtype != BMA_F32 && dtype != BMA_BF16 && dtype != BMA_F16) return BMA_EDTYPE;
  const int es = dtype == BMA_F32 ? 4 : 2;
  const int ne = 16 / es;
  if (Dh % (2 * ne)) return BMA_EALIGN;
  const int cph_host = Dh / ne;
  if (cph_host > 64 || (cph_host & (cph_host - 1))) return BMA_ELIMIT;
  const int64_t strides[12] = {q_b, q_h, q_l, qd_b, qd_h, qd_l, k_b, k_h, k_l, kd_b, kd_h, kd_l};
  for (int i = 0; i < 12; ++i)
    if ((strides[i] * es) % 16) return BMA_EALIGN;
  if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(qd) | reinterpret_cast<uintptr_t>(k) |
       reinterpret_cast<uintptr_t>(kd) | reinterpret_cast<uintptr_t>(cos) | reinterpret_cast<uintptr_t>(sin)) % 16)
    return BMA_EALIGN;
  int cph_log2 = 0;
  while ((1 << cph_log2) < cph_host) ++cph_log2;
  const int64_t rows = static_cast<int64_t>(B) * L;
  if (rows > 0x7fffffffLL) return BMA_ELIMIT;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(static_cast<unsigned>(rows), 2), block(256);
  const RopeTensor tq{q, qd, q_b, q_h, q_l, qd_b, qd_h, qd_l, Hq};
  const RopeTensor tk{k, kd, k_b, k_h, k_l, kd_b, kd_h, kd_l, Hk};
  BMA_PROF_BEGIN(BMA_K_ROPE, st, 2.0 * static_cast<double>(B) * (Hq + Hk) * L * Dh * es);
#define BMA_ROPE2_GO(DT_) \
  hipLaunchKernelGGL((rope2_kernel<DT_>), grid, block, 0, st, tq, tk, B, L, Dh, cos, sin, cos_batch, cph_log2, sin_sign)
  if (dtype == BMA_F32) BMA_ROPE2_GO(BMA_F32);
  else if (dtype == BMA_BF16) BMA_ROPE2_GO(BMA_BF16);
  else BMA_ROPE2_GO(BMA_F16);
#undef BMA_ROPE2_GO
  BMA_PROF_END(BMA_K_ROPE, st);
  BMA_LAUNCH_CHECK();
  return BMA_OK;
}

extern "C" int bma_qknorm_rope2(const void* q, int64_t q_b, int64_t q_h, int64_t q_l, void* qd, int64_t qd_b, int64_t qd_h,
                                int64_t qd_l, int Hq, const void* k, int64_t k_b, int64_t k_h, int64_t k_l, void* kd,
                                int64_t kd_b, int64_t kd_h, int64_t kd_l, int Hk, int B, int L, int Dh, const void* wq,
                                const void* wk, float eps, int gemma, const void* cos, const void* sin, int cos_batch,
                                int dtype, void* stream) {
  if (B < 0 || Hq <= 0 || Hk <= 0 || L < 0 || Dh <= 0 || (cos_batch != 1 && cos_batch != B)) return BMA_EINVAL;
  if (B == 0 || L == 0) return BMA_OK;
  if (!q || !qd || !k || !kd || !cos || !sin || !wq || !wk) return BMA_EINVAL;
  if (dtype != BMA_F32 && dtype != BMA_BF16 && dtype != BMA_F16) return BMA_EDTYPE;
  const int es = dtype == BMA_F32 ? 4 : 2;
  const int ne = 16 / es;
  if (Dh % (2 * ne)) return BMA_EALIGN;
  const int cph_host = Dh / ne;
  if (cph_host > 64 || (cph_host & (cph_host - 1))) return BMA_ELIMIT;   // a head's chunks: one aligned group of lanes
  const int64_t strides[12] = {q_b, q_h, q_l, qd_b, qd_h, qd_l, k_b, k_h, k_l, kd_b, kd_h, kd_l};
  for (int i = 0; i < 12; ++i)
    if ((strides[i] * es) % 16) return BMA_EALIGN;
  if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(qd) | reinterpret_cast<uintptr_t>(k) |
       reinterpret_cast<uintptr_t>(kd) | reinterpret_cast<uintptr_t>(cos) | reinterpret_cast<uintptr_t>(sin) |
       reinterpret_cast<uintptr_t>(wq) | reinterpret_cast<uintptr_t>(wk)) % 16)
    return BMA_EALIGN;
  int cph_log2 = 0;
  while ((1 << cph_log2) < cph_host) ++cph_log2;
  const int64_t rows = static_cast<int64_t>(B) * L;
  if (rows > 0x7fffffffLL) return BMA_ELIMIT;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(static_cast<unsigned>(rows), 2), block(256);
  const RopeTensor tq{q, qd, q_b, q_h, q_l, qd_b, qd_h, qd_l, Hq};
  const RopeTensor tk{k, kd, k_b, k_h, k_l, kd_b, kd_h, kd_l, Hk};
  BMA_PROF_BEGIN(BMA_K_ROPE, st, 2.0 * static_cast<double>(B) * (Hq + Hk) * L * Dh * es);
#define BMA_QKR_GO(DT_) \
  hipLaunchKernelGGL((qknorm_rope2_kernel<DT_>), grid, block, 0, st, tq, tk, B, L, Dh, cos, sin, cos_batch, cph_log2, wq, wk, eps, gemma)
  if (dtype == BMA_F32) BMA_QKR_GO(BMA_F32);
  else if (dtype == BMA_BF16) BMA_QKR_GO(BMA_BF16);
  else BMA_QKR_GO(BMA_F16);
#undef BMA_QKR_GO
  BMA_PROF_END(BMA_K_ROPE, st);
  BMA_LAUNCH_CHECK();
  return BMA_OK;
}

extern "C" int bma_rope_inplace(void* q, int64_t stride_b, int64_t stride_h, int64_t stride_l, int B, int H, int L,
                                int Dh, const void* cos, const void* sin, int cos_batch, int dtype, void* stream) {
  return bma_rope(q, stride_b, stride_h, stride_l, q, stride_b, stride_h, stride_l, B, H, L, Dh, cos, sin, cos_batch,
                  1.0f, dtype, stream);
}

// ---------------------------------------------------------------------------- attention merge
// Shared-prefix attention (hf_adapter / prefix_attention.py): a candidate's new tokens attend
// (1) to the prompt prefix, whose keys/values are THE SAME for every candidate, and (2) causally
// to themselves.  The two partial softmaxes come from two flash-attention launches with their
// log-sum-exps; this kernel merges them,  out = w*o1 + (1-w)*o2,  w = 1/(1+exp(l2-l1)),
// in one pass.  o1, o2, out: [B][L][H][Dh] contiguous; lse1: [H][B*L] (the prefix launch runs
// with batch 1 and B*L queries), lse2: [B][H][L]; both fp32.
namespace {

// `map` (optional): row n of o1/lse1/out pairs with row map[n] of o2/lse2 -- the ragged scoring
// layout, where o1 holds only the tokens that were computed and o2 the padded (B,L) block.
template <int DT>
__global__ __launch_bounds__(256) void attn_merge_kernel(const uint4_t* __restrict__ o1, const uint4_t* __restrict__ o2,
                                                         const float* __restrict__ lse1, const float* __restrict__ lse2,
                                                         const int* __restrict__ map, int64_t N, int64_t rows2, int L,
                                                         int H, int cph, uint4_t* __restrict__ out) {
  constexpr int NE = Chunk<DT>::NE;
  const int row = blockIdx.x;                  // row of o1 / out
  int64_t r2 = row;
  if (map) {
    r2 = map[row];
    r2 = r2 < 0 ? 0 : (r2 >= rows2 ? rows2 - 1 : r2);
  }
  const int b = static_cast<int>(r2 / L), l = static_cast<int>(r2 - static_cast<int64_t>(b) * L);
  const int n = H * cph;                       // chunks of this row
  const int64_t base = static_cast<int64_t>(row) * n;
  const int64_t base2 = r2 * n;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int h = i / cph;                     // 32-bit, once per chunk
    const float l1 = lse1[static_cast<int64_t>(h) * N + row];
    const float l2 = lse2[(static_cast<int64_t>(b) * H + h) * L + l];
    const float w = 1.0f / (1.0f + expf(l2 - l1));
    float a[NE], c[NE], o[NE];
    Chunk<DT>::unpack(o1[base + i], a);
    Chunk<DT>::unpack(o2[base2 + i], c);
#pragma unroll
    for (int j = 0; j < NE; ++j) o[j] = c[j] + w * (a[j] - c[j]);
    out[base + i] = Chunk<DT>::pack(o);
  }
}

// out[r] = src[idx[r]]: rows of `cpr` 16-byte chunks, one workgroup per output row
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint4_t* __restrict__ src, const int* __restrict__ idx,
                                                          int64_t n_src, int cpr, uint4_t* __restrict__ out) {
  const int64_t r = blockIdx.x;
  int64_t s = idx[r];
  s = s < 0 ? 0 : (s >= n_src ? n_src - 1 : s);
  const uint4_t* in = src + s * cpr;
  uint4_t* o = out + r * cpr;
  for (int i = threadIdx.x; i < cpr; i += 256) o[i] = in[i];
}

int merge_common(const void* o1, const void* o2, const float* lse1, const float* lse2, const int* map, int64_t N,
                 int64_t rows2, int L, int H, int Dh, int dtype, void* out, void* stream) {
  if (N == 0) return BMA_OK;
  if (!o1 || !o2 || !lse1 || !lse2 || !out) return BMA_EINVAL;
  if (dtype != BMA_F32 && dtype != BMA_BF16 && dtype != BMA_F16) return BMA_EDTYPE;
  const int es = dtype == BMA_F32 ? 4 : 2;
  if ((Dh * es) % 16) return BMA_EALIGN;
  if ((reinterpret_cast<uintptr_t>(o1) | reinterpret_cast<uintptr_t>(o2) | reinterpret_cast<uintptr_t>(out)) % 16)
    return BMA_EALIGN;
  const int cph = Dh * es / 16;
  if (N > 0x7fffffffLL || rows2 > 0x7fffffffLL) return BMA_ELIMIT;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(static_cast<unsigned>(N)), block(256);
  const uint4_t* a = static_cast<const uint4_t*>(o1);
  const uint4_t* c = static_cast<const uint4_t*>(o2);
  uint4_t* y = static_cast<uint4_t*>(out);
  BMA_PROF_BEGIN(BMA_K_ATTN_MERGE, st, 3.0 * static_cast<double>(N) * H * Dh * es);
  if (dtype == BMA_F32) hipLaunchKernelGGL((attn_merge_kernel<BMA_F32>), grid, block, 0, st, a, c, lse1, lse2, map, N, rows2, L, H, cph, y);
  else if (dtype == BMA_BF16) hipLaunchKernelGGL((attn_merge_kernel<BMA_BF16>), grid, block, 0, st, a, c, lse1, lse2, map, N, rows2, L, H, cph, y);
  else hipLaunchKernelGGL((attn_merge_kernel<BMA_F16>), grid, block, 0, st, a, c, lse1, lse2, map, N, rows2, L, H, cph, y);
  BMA_PROF_END(BMA_K_ATTN_MERGE, st);
  BMA_LAUNCH_CHECK();
  return BMA_OK;
}

}  // namespace

extern "C" int bma_attn_merge(const void* o1, const void* o2, const float* lse1, const float* lse2, int B, int L,
                              int H, int Dh, int dtype, void* out, void* stream) {
  if (B < 0 || L < 0 || H <= 0 || Dh <= 0) return BMA_EINVAL;
  if (B == 0 || L == 0) return BMA_OK;
  const int64_t rows = static_cast<int64_t>(B) * L;
  return merge_common(o1, o2, lse1, lse2, nullptr, rows, rows, L, H, Dh, dtype, out, stream);
}

extern "C" int bma_attn_merge_rows(const void* o1, const void* o2, const float* lse1, const float* lse2,
                                   const int* map, int64_t N, int B2, int L, int H, int Dh, int dtype, void* out,
                                   void* stream) {
  if (N < 0 || B2 <= 0 || L <= 0 || H <= 0 || Dh <= 0 || !map) return BMA_EINVAL;
  return merge_common(o1, o2, lse1, lse2, map, N, static_cast<int64_t>(B2) * L, L, H, Dh, dtype, out, stream);
}

extern "C" int bma_gather_rows(const void* src, const int* idx, int64_t n_out, int64_t n_src, int64_t row_bytes,
                               void* out, void* stream) {
  if (n_out < 0 || n_src <= 0 || row_bytes <= 0) return BMA_EINVAL;
  if (n_out == 0) return BMA_OK;
  if (!src || !idx || !out) return BMA_EINVAL;
  if (row_bytes % 16 || (reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(out)) % 16) return BMA_EALIGN;
  if (n_out > 0x7fffffffLL || row_bytes / 16 > 0x7fffffffLL) return BMA_ELIMIT;
  hipStream_t st = static_cast<hipStream_t>(stream);
  BMA_PROF_BEGIN(BMA_K_GATHER_ROWS, st, 2.0 * static_cast<double>(n_out) * row_bytes);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(static_cast<unsigned>(n_out)), dim3(256), 0, st,
                     static_cast<const uint4_t*>(src), idx, n_src, static_cast<int>(row_bytes / 16),
                     static_cast<uint4_t*>(out));
  BMA_PROF_END(BMA_K_GATHER_ROWS, st);
  BMA_LAUNCH_CHECK();
  return BMA_OK;
}

// ---------------------------------------------------------------------------- backward passes
// The gradient pass (reference :953-1028) runs the same layers under autograd at batch 1:
// ~2900 tiny eager kernels around 13 ms of weight-streaming GEMMs.  These kernels are the
// backward halves of the fused forward ops, so the whole pass (captured in a hipGraph) needs
// one launch per op and direction.  All math in fp32, one rounding to the model dtype.
//
//   rmsnorm_bwd : g = dy*w (Llama) or dy*(1+w) (Gemma);  r = rsqrt(mean(x^2)+eps)
//                 dx = r*g - x * r^3 * sum(g*x)/D          (weights are constants here: no dw)
//   swiglu_bwd  : s = sigmoid(g);  d_up = dy * dt(g*s);  d_gate = dy * u * s*(1 + g*(1-s))
//   rope bwd    : the rotation is orthogonal -- backward is the forward kernel with -sin.
namespace {

// `add` (optional): a gradient arriving at x by another path (the residual stream past a fused add + norm):
// dx = rnd(add + rnd(dx_norm)) -- the sum autograd's accumulation would form with one more launch.
template <int DT, int NCH, bool GEMMA>
__global__ __launch_bounds__(kNormThreads) void rmsnorm_bwd_kernel(const uint4_t* __restrict__ x,
                                                                   const uint4_t* __restrict__ w,
                                                                   const uint4_t* __restrict__ dy,
                                                                   const uint4_t* __restrict__ add, float eps, int cpr,
                                                                   int D, uint4_t* __restrict__ dx) {
  constexpr int NE = Chunk<DT>::NE;
  const int64_t row = blockIdx.x;
  const int tid = threadIdx.x;
  float xv[NCH][NE], gv[NCH][NE];
  float ss = 0.0f, sg = 0.0f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int i = tid + c * kNormThreads;
    if (i < cpr) {
      float wf[NE], dyf[NE];
      Chunk<DT>::unpack(x[row * cpr + i], xv[c]);
      Chunk<DT>::unpack(dy[row * cpr + i], dyf);
      Chunk<DT>::unpack(w[i], wf);
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        gv[c][j] = dyf[j] * (GEMMA ? 1.0f + wf[j] : wf[j]);
        ss += xv[c][j] * xv[c][j];
        sg += gv[c][j] * xv[c][j];
      }
    }
  }
  ss = bma::wave_sum(ss);
  sg = bma::wave_sum(sg);
  __shared__ float part[2][kNormThreads / 64];
  if ((tid & 63) == 0) { part[0][tid >> 6] = ss; part[1][tid >> 6] = sg; }
  __syncthreads();
  float tss = 0.0f, tsg = 0.0f;
#pragma unroll
  for (int i = 0; i < kNormThreads / 64; ++i) { tss += part[0][i]; tsg += part[1][i]; }
  const float r = 1.0f / sqrtf(tss / static_cast<float>(D) + eps);
  const float k = r * r * r * tsg / static_cast<float>(D);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int i = tid + c * kNormThreads;
    if (i < cpr) {
      float o[NE];
#pragma unroll
      for (int j = 0; j < NE; ++j) o[j] = r * gv[c][j] - xv[c][j] * k;
      if (add) {
        float af[NE];
        Chunk<DT>::unpack(add[row * cpr + i], af);
#pragma unroll
        for (int j = 0; j < NE; ++j) o[j] = af[j] + rnd<DT>(o[j]);
      }
      dx[row * cpr + i] = Chunk<DT>::pack(o);
    }
  }
}

template <int DT, int ACT, bool IL>
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const uint4_t* __restrict__ g, const uint4_t* __restrict__ u,
                                                         const uint4_t* __restrict__ dy, int64_t n_chunks,
                                                         uint4_t* __restrict__ dg, uint4_t* __restrict__ du) {
  constexpr int NE = Chunk<DT>::NE;
  constexpr int S = IL ? 2 : 1;                 // IL: gate/up and their gradients as alternating chunks of one array
  if (IL) { u = g + 1; du = dg + 1; }
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n_chunks; i += stride) {
    float gf[NE], uf[NE], df[NE], og[NE], ou[NE];
    Chunk<DT>::unpack(g[S * i], gf);
    Chunk<DT>::unpack(u[S * i], uf);
    Chunk<DT>::unpack(dy[i], df);
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      if (ACT == 0) {
        const float s = 1.0f / (1.0f + expf(-gf[j]));
        ou[j] = df[j] * rnd<DT>(gf[j] * s);                     // the forward kept act(g) in the model dtype
        og[j] = df[j] * uf[j] * (s * (1.0f + gf[j] * (1.0f - s)));
      } else {
        constexpr float kBeta = 0.7978845608028654f, kKappa = 0.044715f;
        const float x = gf[j], x2 = x * x;
        const float t = tanhf(kBeta * fmaf(kKappa, x2 * x, x));
        const float d = 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * (kBeta * (1.0f + 3.0f * kKappa * x2));
        ou[j] = df[j] * rnd<DT>(0.5f * x * (1.0f + t));
        og[j] = df[j] * uf[j] * d;
      }
    }
    dg[S * i] = Chunk<DT>::pack(og);
    du[S * i] = Chunk<DT>::pack(ou);
  }
}

template <int DT>
int launch_rmsnorm_bwd(const void* x, const void* w, const void* dy, const void* add, float eps, int64_t rows, int D,
                       int gemma, void* dx, hipStream_t st) {
  constexpr int ES = bma::elem_bytes<DT>::value;
  const int cpr = static_cast<int>(static_cast<int64_t>(D) * ES / 16);
  const int nch = (cpr + kNormThreads - 1) / kNormThreads;
  if (nch > kNormMaxChunks) return BMA_ELIMIT;   // x and g rows stay in registers: D*es <= 16 KiB
  const dim3 grid(static_cast<unsigned>(rows)), block(kNormThreads);
  const uint4_t* xp = static_cast<const uint4_t*>(x);
  const uint4_t* wp = static_cast<const uint4_t*>(w);
  const uint4_t* dp = static_cast<const uint4_t*>(dy);
  const uint4_t* ap = static_cast<const uint4_t*>(add);
  uint4_t* op = static_cast<uint4_t*>(dx);
#define BMA_NB_GO(N)                                                                                                    \
  do {                                                                                                                  \
    if (gemma) hipLaunchKernelGGL((rmsnorm_bwd_kernel<DT, N, true>), grid, block, 0, st, xp, wp, dp, ap, eps, cpr, D, op); \
    else hipLaunchKernelGGL((rmsnorm_bwd_kernel<DT, N, false>), grid, block, 0, st, xp, wp, dp, ap, eps, cpr, D, op);      \
  } while (0)
  switch (nch) {
    case 1: BMA_NB_GO(1); break;
    case 2: BMA_NB_GO(2); break;
    case 3: BMA_NB_GO(3); break;
    default: BMA_NB_GO(4); break;
  }
#undef BMA_NB_GO
  BMA_LAUNCH_CHECK();
  return BMA_OK;
}

}  // namespace

extern "C" int bma_add_rmsnorm_bwd(const void* x, const void* weight, const void* dy, const void* dsum, float eps,
                                   int64_t rows, int D, int dtype, int gemma_style, void* dx, void* stream) {
  if (rows < 0 || D <= 0 || rows > 0x7fffffffLL) return BMA_EINVAL;
  if (rows == 0) return BMA_OK;
  if (!x || !weight || !dy || !dx) return BMA_EINVAL;
  const int es = dtype == BMA_F32 ? 4 : 2;
  if ((static_cast<int64_t>(D) * es) % 16) return BMA_EALIGN;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(weight) | reinterpret_cast<uintptr_t>(dy) |
       reinterpret_cast<uintptr_t>(dsum) | reinterpret_cast<uintptr_t>(dx)) % 16)
    return BMA_EALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case BMA_F32: return launch_rmsnorm_bwd<BMA_F32>(x, weight, dy, dsum, eps, rows, D, gemma_style, dx, st);
    case BMA_BF16: return launch_rmsnorm_bwd<BMA_BF16>(x, weight, dy, dsum, eps, rows, D, gemma_style, dx, st);
    case BMA_F16: return launch_rmsnorm_bwd<BMA_F16>(x, weight, dy, dsum, eps, rows, D, gemma_style, dx, st);
    default: return BMA_EDTYPE;
  }
}

extern "C" int bma_rmsnorm_bwd(const void* x, const void* weight, const void* dy, float eps, int64_t rows, int D,
                               int dtype, int gemma_style, void* dx, void* stream) {
  return bma_add_rmsnorm_bwd(x, weight, dy, nullptr, eps, rows, D, dtype, gemma_style, dx, stream);
}

static int gated_act_bwd_launch(const void* gate, const void* up, bool interleaved, const void* dy, int64_t n, int dtype,
                                int act, void* dgate, void* dup, void* stream) {
  if (n < 0 || (act != 0 && act != 1)) return BMA_EINVAL;
  if (n == 0) return BMA_OK;
  if (!gate || !dy || !dgate || (!interleaved && (!up || !dup))) return BMA_EINVAL;
  if (dtype != BMA_F32 && dtype != BMA_BF16 && dtype != BMA_F16) return BMA_EDTYPE;
  const int es = dtype == BMA_F32 ? 4 : 2;
  if ((n * es) % 16) return BMA_EALIGN;
  if ((reinterpret_cast<uintptr_t>(gate) | reinterpret_cast<uintptr_t>(up) | reinterpret_cast<uintptr_t>(dy) |
       reinterpret_cast<uintptr_t>(dgate) | reinterpret_cast<uintptr_t>(dup)) % 16)
    return BMA_EALIGN;
  const int64_t chunks = n * es / 16;
  int64_t blocks = (chunks + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(static_cast<unsigned>(blocks)), block(256);
  const uint4_t* g = static_cast<const uint4_t*>(gate);
  const uint4_t* u = static_cast<const uint4_t*>(up);
  const uint4_t* d = static_cast<const uint4_t*>(dy);
  uint4_t* og = static_cast<uint4_t*>(dgate);
  uint4_t* ou = static_cast<uint4_t*>(dup);
#define BMA_GB_GO(DT_)                                                                                                  \
  do {                                                                                                                  \
    if (interleaved) {                                                                                                  \
      if (act == 0) hipLaunchKernelGGL((swiglu_bwd_kernel<DT_, 0, true>), grid, block, 0, st, g, u, d, chunks, og, ou);  \
      else hipLaunchKernelGGL((swiglu_bwd_kernel<DT_, 1, true>), grid, block, 0, st, g, u, d, chunks, og, ou);           \
    } else {                                                                                                            \
      if (act == 0) hipLaunchKernelGGL((swiglu_bwd_kernel<DT_, 0, false>), grid, block, 0, st, g, u, d, chunks, og, ou); \
      else hipLaunchKernelGGL((swiglu_bwd_kernel<DT_, 1, false>), grid, block, 0, st, g, u, d, chunks, og, ou);          \
    }                                                                                                                   \
  } while (0)
  if (dtype == BMA_F32) BMA_GB_GO(BMA_F32);
  else if (dtype == BMA_BF16) BMA_GB_GO(BMA_BF16);
  else BMA_GB_GO(BMA_F16);
#undef BMA_GB_GO
  BMA_LAUNCH_CHECK();
  return BMA_OK;
}

extern "C" int bma_gated_act_bwd(const void* gate, const void* up, const void* dy, int64_t n, int dtype, int act,
                                 void* dgate, void* dup, void* stream) {
  return gated_act_bwd_launch(gate, up, false, dy, n, dtype, act, dgate, dup, stream);
}

extern "C" int bma_gated_act_il_bwd(const void* gate_up, const void* dy, int64_t n, int dtype, int act, void* dgate_up,
                                    void* stream) {
  return gated_act_bwd_launch(gate_up, nullptr, true, dy, n, dtype, act, dgate_up, nullptr, stream);
}

extern "C" int bma_swiglu_bwd(const void* gate, const void* up, const void* dy, int64_t n, int dtype, void* dgate,
                              void* dup, void* stream) {
  return bma_gated_act_bwd(gate, up, dy, n, dtype, 0, dgate, dup, stream);
}

// ---------------------------------------------------------------------------- LayerNorm (+ residual add): CLIP's pre-LN blocks
// The vision tower of LLaVA at batch 1 (577 x 1024) is launch-bound: per encoder layer HuggingFace issues a residual add and a
// LayerNorm twice (modeling_clip.CLIPEncoderLayer.forward), and autograd an add and a LayerNorm backward twice more.  Here:
//   s = dt(res + h)                      (skipped without `res`: a plain LayerNorm of h)
//   y = dt(w * (rstd * (s - mean)) + b)  mean / rstd over the row in fp32 (two passes over the registers), aten's expression
// one launch, the row in registers throughout; (mean, rstd) are kept for the backward, which is one launch as well:
//   dx = dt(rstd * (g - mean(g) - xhat * mean(g * xhat)))  with g = dy * w, xhat = (x - mean) * rstd,  + dsum if given
// (the eager chain's rounding points: LayerNorm's input gradient rounded to the dtype, then the add).  Weights are constants
// of the attack: no weight / bias gradient is formed.
template <int DT, int NCH, bool ADD>
__global__ __launch_bounds__(kNormThreads) void add_layernorm_kernel(const uint4_t* __restrict__ res, const uint4_t* __restrict__ h,
                                                                     const uint4_t* __restrict__ w, const uint4_t* __restrict__ b,
                                                                     float eps, int cpr, int D, uint4_t* __restrict__ s_out,
                                                                     uint4_t* __restrict__ y, float* __restrict__ stats) {
  constexpr int NE = Chunk<DT>::NE;
  const int64_t row = blockIdx.x;
  const int tid = threadIdx.x;
  __shared__ float part[2][kNormThreads / 64];
  float v[NCH][NE];
  float sum = 0.0f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int i = tid + c * kNormThreads;
    if (i < cpr) {
      Chunk<DT>::unpack(h[row * cpr + i], v[c]);
      if (ADD) {
        float r[NE];
        Chunk<DT>::unpack(res[row * cpr + i], r);
#pragma unroll
        for (int j = 0; j < NE; ++j) v[c][j] = rnd<DT>(r[j] + v[c][j]);   // the sum as the eager add leaves it in the model dtype
        s_out[row * cpr + i] = Chunk<DT>::pack(v[c]);
      }
#pragma unroll
      for (int j = 0; j < NE; ++j) sum += v[c][j];
    }
  }
  sum = bma::wave_sum(sum);
  if ((tid & 63) == 0) part[0][tid >> 6] = sum;
  __syncthreads();
  float tot = 0.0f;
#pragma unroll
  for (int i = 0; i < kNormThreads / 64; ++i) tot += part[0][i];
  const float mean = tot / static_cast<float>(D);
  float sq = 0.0f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int i = tid + c * kNormThreads;
    if (i < cpr) {
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        const float d = v[c][j] - mean;
        sq += d * d;
      }
    }
  }
  sq = bma::wave_sum(sq);
  if ((tid & 63) == 0) part[1][tid >> 6] = sq;
  __syncthreads();
  float tot2 = 0.0f;
#pragma unroll
  for (int i = 0; i < kNormThreads / 64; ++i) tot2 += part[1][i];
  const float rstd = rsqrtf(tot2 / static_cast<float>(D) + eps);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int i = tid + c * kNormThreads;
    if (i < cpr) {
      float wf[NE], bf[NE], o[NE];
      Chunk<DT>::unpack(w[i], wf);
      Chunk<DT>::unpack(b[i], bf);
#pragma unroll
      for (int j = 0; j < NE; ++j) o[j] = wf[j] * (rstd * (v[c][j] - mean)) + bf[j];
      y[row * cpr + i] = Chunk<DT>::pack(o);
    }
  }
  if (stats && tid == 0) {
    stats[2 * row] = mean;
    stats[2 * row + 1] = rstd;
  }
}

template <int DT, int NCH>
__global__ __launch_bounds__(kNormThreads) void add_layernorm_bwd_kernel(const uint4_t* __restrict__ x, const uint4_t* __restrict__ w,
                                                                         const uint4_t* __restrict__ dy, const uint4_t* __restrict__ add,
                                                                         const float* __restrict__ stats, int cpr, int D,
                                                                         uint4_t* __restrict__ dx) {
  constexpr int NE = Chunk<DT>::NE;
  const int64_t row = blockIdx.x;
  const int tid = threadIdx.x;
  const float mean = stats[2 * row], rstd = stats[2 * row + 1];
  float xh[NCH][NE], g[NCH][NE];
  float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int i = tid + c * kNormThreads;
    if (i < cpr) {
      float xf[NE], wf[NE], df[NE];
      Chunk<DT>::unpack(x[row * cpr + i], xf);
      Chunk<DT>::unpack(w[i], wf);
      Chunk<DT>::unpack(dy[row * cpr + i], df);
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        xh[c][j] = (xf[j] - mean) * rstd;
        g[c][j] = df[j] * wf[j];
        s1 += g[c][j];
        s2 += g[c][j] * xh[c][j];
      }
    }
  }
  s1 = bma::wave_sum(s1);
  s2 = bma::wave_sum(s2);
  __shared__ float part[2][kNormThreads / 64];
  if ((tid & 63) == 0) { part[0][tid >> 6] = s1; part[1][tid >> 6] = s2; }
  __syncthreads();
  float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
  for (int i = 0; i < kNormThreads / 64; ++i) { t1 += part[0][i]; t2 += part[1][i]; }
  const float c1 = t1 / static_cast<float>(D), c2 = t2 / static_cast<float>(D);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int i = tid + c * kNormThreads;
    if (i < cpr) {
      float o[NE];
#pragma unroll
      for (int j = 0; j < NE; ++j) o[j] = rstd * (g[c][j] - c1 - xh[c][j] * c2);
      if (add) {
        float af[NE];
        Chunk<DT>::unpack(add[row * cpr + i], af);
#pragma unroll
        for (int j = 0; j < NE; ++j) o[j] = af[j] + rnd<DT>(o[j]);
      }
      dx[row * cpr + i] = Chunk<DT>::pack(o);
    }
  }
}

namespace {
template <int DT>
int launch_add_layernorm(const void* res, const void* h, const void* w, const void* b, float eps, int64_t rows, int D, void* s_out,
                         void* y, float* stats, hipStream_t st) {
  constexpr int ES = bma::elem_bytes<DT>::value;
  const int cpr = static_cast<int>(static_cast<int64_t>(D) * ES / 16);
  const int nch = (cpr + kNormThreads - 1) / kNormThreads;
  if (nch > kNormMaxChunks) return BMA_ELIMIT;
  const dim3 grid(static_cast<unsigned>(rows)), block(kNormThreads);
  const uint4_t* rp = static_cast<const uint4_t*>(res);
  const uint4_t* hp = static_cast<const uint4_t*>(h);
  const uint4_t* wp = static_cast<const uint4_t*>(w);
  const uint4_t* bp = static_cast<const uint4_t*>(b);
  uint4_t* sp = static_cast<uint4_t*>(s_out);
  uint4_t* yp = static_cast<uint4_t*>(y);
#define BMA_LN_GO(N)                                                                                                               \
  do {                                                                                                                             \
    if (res) hipLaunchKernelGGL((add_layernorm_kernel<DT, N, true>), grid, block, 0, st, rp, hp, wp, bp, eps, cpr, D, sp, yp, stats);  \
    else hipLaunchKernelGGL((add_layernorm_kernel<DT, N, false>), grid, block, 0, st, rp, hp, wp, bp, eps, cpr, D, sp, yp, stats);     \
  } while (0)
  switch (nch) {
    case 1: BMA_LN_GO(1); break;
    case 2: BMA_LN_GO(2); break;
    case 3: BMA_LN_GO(3); break;
    default: BMA_LN_GO(4); break;
  }
#undef BMA_LN_GO
  BMA_LAUNCH_CHECK();
  return BMA_OK;
}

template <int DT>
int launch_add_layernorm_bwd(const void* x, const void* w, const void* dy, const void* add, const float* stats, int64_t rows, int D,
                             void* dx, hipStream_t st) {
  constexpr int ES = bma::elem_bytes<DT>::value;
  const int cpr = static_cast<int>(static_cast<int64_t>(D) * ES / 16);
  const int nch = (cpr + kNormThreads - 1) / kNormThreads;
  if (nch > kNormMaxChunks) return BMA_ELIMIT;
  const dim3 grid(static_cast<unsigned>(rows)), block(kNormThreads);
  const uint4_t* xp = static_cast<const uint4_t*>(x);
  const uint4_t* wp = static_cast<const uint4_t*>(w);
  const uint4_t* dp = static_cast<const uint4_t*>(dy);
  const uint4_t* ap = static_cast<const uint4_t*>(add);
  uint4_t* op = static_cast<uint4_t*>(dx);
  switch (nch) {
    case 1: hipLaunchKernelGGL((add_layernorm_bwd_kernel<DT, 1>), grid, block, 0, st, xp, wp, dp, ap, stats, cpr, D, op); break;
    case 2: hipLaunchKernelGGL((add_layernorm_bwd_kernel<DT, 2>), grid, block, 0, st, xp, wp, dp, ap, stats, cpr, D, op); break;
    case 3: hipLaunchKernelGGL((add_layernorm_bwd_kernel<DT, 3>), grid, block, 0, st, xp, wp, dp, ap, stats, cpr, D, op); break;
    default: hipLaunchKernelGGL((add_layernorm_bwd_kernel<DT, 4>), grid, block, 0, st, xp, wp, dp, ap, stats, cpr, D, op); break;
  }
  BMA_LAUNCH_CHECK();
  return BMA_OK;
}
}  // namespace

extern "C" int bma_add_layernorm(const void* residual, const void* h, const void* weight, const void* bias, float eps, int64_t rows,
                                 int D, int dtype, void* sum_out, void* out, float* stats, void* stream) {
  if (rows < 0 || D <= 0 || rows > 0x7fffffffLL) return BMA_EINVAL;
  if (rows == 0) return BMA_OK;
  if (!h || !weight || !bias || !out || (residual && !sum_out)) return BMA_EINVAL;
  const int es = dtype == BMA_F32 ? 4 : 2;
  if ((static_cast<int64_t>(D) * es) % 16) return BMA_EALIGN;
  if ((reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(weight) |
       reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(sum_out) | reinterpret_cast<uintptr_t>(out)) % 16 ||
      reinterpret_cast<uintptr_t>(stats) % 4)
    return BMA_EALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case BMA_F32: return launch_add_layernorm<BMA_F32>(residual, h, weight, bias, eps, rows, D, sum_out, out, stats, st);
    case BMA_BF16: return launch_add_layernorm<BMA_BF16>(residual, h, weight, bias, eps, rows, D, sum_out, out, stats, st);
    case BMA_F16: return launch_add_layernorm<BMA_F16>(residual, h, weight, bias, eps, rows, D, sum_out, out, stats, st);
    default: return BMA_EDTYPE;
  }
}

extern "C" int bma_add_layernorm_bwd(const void* x, const void* weight, const void* dy, const void* dsum, const float* stats,
                                     int64_t rows, int D, int dtype, void* dx, void* stream) {
  if (rows < 0 || D <= 0 || rows > 0x7fffffffLL) return BMA_EINVAL;
  if (rows == 0) return BMA_OK;
  if (!x || !weight || !dy || !stats || !dx) return BMA_EINVAL;
  const int es = dtype == BMA_F32 ? 4 : 2;
  if ((static_cast<int64_t>(D) * es) % 16) return BMA_EALIGN;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(weight) | reinterpret_cast<uintptr_t>(dy) |
       reinterpret_cast<uintptr_t>(dsum) | reinterpret_cast<uintptr_t>(dx)) % 16 || reinterpret_cast<uintptr_t>(stats) % 4)
    return BMA_EALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case BMA_F32: return launch_add_layernorm_bwd<BMA_F32>(x, weight, dy, dsum, stats, rows, D, dx, st);
    case BMA_BF16: return launch_add_layernorm_bwd<BMA_BF16>(x, weight, dy, dsum, stats, rows, D, dx, st);
    case BMA_F16: return launch_add_layernorm_bwd<BMA_F16>(x, weight, dy, dsum, stats, rows, D, dx, st);
    default: return BMA_EDTYPE;
  }
}
