// a1 (gradient pass, text-only prompt) -- rotary embedding + causal self-attention of ONE short sequence, forward and
// backward, one launch each:  bma_b1_attention / bma_b1_attention_bwd.
//
// The reference's compute_gradient (bimodal_attack.py:953-1028) runs the language model at batch 1 over a 66-token
// prompt; every decoder layer's attention there is 32 heads of a 65 x 65 causal problem -- a few MFLOP -- which the
// HuggingFace code reaches through six launches forward (three reshapes folded away, rotary, attention) and, under
// autograd, through the library's attention backward, a counter fill, the rotary backward and a concatenation: 46 us of
// launch-bound kernels per layer, 1.5 ms of the 10.6 ms pass (profiles/archive/r3_gradient_pass_gcg_by_grid.txt).  Here the
// fused q/k/v projection output goes in as it is and the attention output comes out in the layout o_proj reads; the
// backward takes d(out) and returns d(qkv) in the projection's own layout.
//
// One workgroup (8 waves) per head; S <= 80 tokens, 128-wide heads, as many key/value heads as query heads, 16-bit types.
// Everything lives in LDS: images [row][d] (272-byte rows) for products that reduce over d, images [d][row] and
// score-shaped [row][row] images (208-byte rows, the row index zero-padded to 96) for products that reduce over a row
// index.  All products are v_mfma_f32_16x16x32 with both operands read K-contiguous (ds_read_b128); the product's
// register layout -- a lane holds D[4*(lane/16) + e][lane%16] -- makes one of {P, P^T} an 8-byte store and the other a
// 2-byte scatter, so scores are formed in BOTH orientations (A and B swapped: 16 more MFMAs per tile) and each
// orientation stores the image it gets for free.
//
//   forward : S^T = K.Q^T (row i in lane%16)  ->  softmax over the lane's registers + two cross-lane steps  ->  P [i][j]
//             O = P.V  (V^T image);  lse[h][i] = max + log(sum)
//   backward: P = exp(scale*S - lse), dP = dO.V^T, dS = scale * P o (dP - delta), delta[i] = dO[i].O[i]
//             dV = P^T.dO   dK = dS^T.Q   dQ = dS.K   (transposed images of dO, Q, K re-read from global in a second phase
//             over the natural images' LDS), inverse rotation of dQ / dK in registers (d and d +- 64 sit in one lane)
// Rounding points are the unfused route's: rotary as bma_rope2 -- dt(dt(x cos) + dt(+-x' sin)) --, P and dS rounded to
// the 16-bit type before their products (as the library's flash kernels do), fp32 accumulation, one rounding of every
// output.  Algorithmic bytes: forward 4*S*H*128*es, backward 8*S*H*128*es -- latency-bound, not bandwidth-bound.

#include "bma_common.h"
#include "bma_profile.h"

namespace {

using bma::uint4_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kDh = 128;
constexpr int kRT = 5;                // 16-row tiles: S <= 80
constexpr int kSR = 16 * kRT;
constexpr int kSK = 96;               // a row index used as a REDUCTION index: zero-padded to a multiple of 32
constexpr int kLdN = kDh + 8;         // [row][d] images, elements per row (272 B: 16-byte aligned, rows 4 banks apart)
constexpr int kLdT = kSK + 8;         // [d][row] and [row][row] images (208 B)

struct Args {
  const char* qkv;   // [S][ld_qkv]: q heads | k heads | v heads, each H x 128
  const char* cos;   // [S][128]
  const char* sin;
  char* out;         // [S][ld_out]: H x 128            (forward: written; backward: read)
  float* lse;        // [H][S]                           (forward: written; backward: read)
  const char* dout;  // [S][ld_dout]                     (backward)
  char* dqkv;        // [S][ld_dqkv]                     (backward)
  int64_t ld_qkv, ld_out, ld_dout, ld_dqkv;   // elements
  int S, H;
  float scale;
};

template <int DT>
__device__ __forceinline__ f32x4 mfma16(const uint4_t& a, const uint4_t& b, const f32x4& c) {
  if (DT == BMA_BF16)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int DT>
__device__ __forceinline__ float rnd(float v) {
  return bma::unpack16<DT>(bma::pack16<DT>(v, 0.0f), 0);
}

template <int DT>
__device__ __forceinline__ void unpack8(const uint4_t& w, float (&v)[8]) {
  v[0] = bma::unpack16<DT>(w.x, 0); v[1] = bma::unpack16<DT>(w.x, 1);
  v[2] = bma::unpack16<DT>(w.y, 0); v[3] = bma::unpack16<DT>(w.y, 1);
  v[4] = bma::unpack16<DT>(w.z, 0); v[5] = bma::unpack16<DT>(w.z, 1);
  v[6] = bma::unpack16<DT>(w.w, 0); v[7] = bma::unpack16<DT>(w.w, 1);
}

template <int DT>
__device__ __forceinline__ uint4_t pack8(const float (&v)[8]) {
  uint4_t w;
  w.x = bma::pack16<DT>(v[0], v[1]);
  w.y = bma::pack16<DT>(v[2], v[3]);
  w.z = bma::pack16<DT>(v[4], v[5]);
  w.w = bma::pack16<DT>(v[6], v[7]);
  return w;
}

// chunk c (8 elements) of a head vector after the rotation, bma_rope2's arithmetic: o = dt(dt(x cos) + dt(+-x' sin)),
// x' the element 64 away, minus for the first half (rotate_half = cat(-x2, x1)); sin_sign = -1 gives the inverse
template <int DT>
__device__ __forceinline__ void rope_chunk(const char* head, const char* cosrow, const char* sinrow, int c, float sin_sign,
                                           float (&o)[8]) {
  float x[8], p[8], cf[8], sf[8];
  unpack8<DT>(*reinterpret_cast<const uint4_t*>(head + c * 16), x);
  unpack8<DT>(*reinterpret_cast<const uint4_t*>(head + (c ^ 8) * 16), p);
  unpack8<DT>(*reinterpret_cast<const uint4_t*>(cosrow + c * 16), cf);
  unpack8<DT>(*reinterpret_cast<const uint4_t*>(sinrow + c * 16), sf);
  const float sign = (c < 8 ? -1.0f : 1.0f) * sin_sign;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = rnd<DT>(rnd<DT>(x[j] * cf[j]) + rnd<DT>(sign * p[j] * sf[j]));
}

// the same from chunks already in registers (x: the chunk, p: its partner 64 elements away)
template <int DT>
__device__ __forceinline__ uint4_t rope_regs(const uint4_t& xw, const uint4_t& pw, const uint4_t& cw, const uint4_t& sw, int c) {
  float x[8], p[8], cf[8], sf[8], o[8];
  unpack8<DT>(xw, x);
  unpack8<DT>(pw, p);
  unpack8<DT>(cw, cf);
  unpack8<DT>(sw, sf);
  const float sign = c < 8 ? -1.0f : 1.0f;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = rnd<DT>(rnd<DT>(x[j] * cf[j]) + rnd<DT>(sign * p[j] * sf[j]));
  return pack8<DT>(o);
}

__device__ __forceinline__ float elem(const f32x4& v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); }

constexpr int kNT = 512;              // threads per workgroup: 8 waves share one head's tiles
constexpr int kNWv = kNT / 64;

__device__ __forceinline__ void zero_lds(unsigned char* p, int bytes, int tid) {
  uint4_t z;
  z.x = z.y = z.z = z.w = 0u;
  for (int o = tid * 16; o < bytes; o += kNT * 16) *reinterpret_cast<uint4_t*>(p + o) = z;
}

// 16 B of row (tile*16 + lane%16) at reduction offset k0 + 8*(lane/16): the A- or B-operand fragment of one MFMA
__device__ __forceinline__ uint4_t frag(const uint16_t* img, int ld, int tile, int k0, int frow, int fg) {
  return *reinterpret_cast<const uint4_t*>(img + (tile * 16 + frow) * ld + k0 + fg * 8);
}

template <int DT>
__device__ __forceinline__ void store4(uint16_t* dst, float a, float b, float c, float d) {   // 8-byte LDS store
  bma::uint2_t o;
  o.x = bma::pack16<DT>(a, b);
  o.y = bma::pack16<DT>(c, d);
  *reinterpret_cast<bma::uint2_t*>(dst) = o;
}

// [row][d] image -> its [d][row] image, eight rows at a time: consecutive lanes take consecutive d, so the eight 2-byte
// reads of a lane group are contiguous and the 16-byte writes land 52 banks apart (no conflicts either way; a 2-byte
// scatter straight from the global loads measured 32-way conflicts).  Rows past kSR do not exist: zeros.
__device__ __forceinline__ uint4_t gather_col(const uint16_t* nat, int d, int ic) {
  uint32_t w[4] = {0u, 0u, 0u, 0u};
  if (ic * 8 < kSR) {
#pragma unroll
    for (int e = 0; e < 8; ++e) w[e >> 1] |= static_cast<uint32_t>(nat[(ic * 8 + e) * kLdN + d]) << ((e & 1) * 16);
  }
  uint4_t v;
  v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
  return v;
}
constexpr int kTI = kDh * (kSK / 8);          // (d, 8-row chunk) items of one transposed image
constexpr int kTN = (kTI + kNT - 1) / kNT;    // ... per thread

// ------------------------------------------------------------------------------------------------------------ forward
template <int DT>
__global__ __launch_bounds__(kNT) void b1_attn_fwd_kernel(Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[(3 * kSR * kLdN + kDh * kLdT + kSR * kLdT) * 2];
  uint16_t* Qs = reinterpret_cast<uint16_t*>(lds);
  uint16_t* Ks = Qs + kSR * kLdN;
  uint16_t* Vs = Ks + kSR * kLdN;
  uint16_t* Vt = Vs + kSR * kLdN;            // [d][j]
  uint16_t* Ps = Vt + kDh * kLdT;            // [i][j]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fg = lane >> 4;
  const int h = blockIdx.x, S = a.S, H = a.H;

  zero_lds(reinterpret_cast<unsigned char*>(Ps), kSR * kLdT * 2, tid);
  {
    // every global load of this thread first ((row, chunk) items: q and k chunks with their partners, v, cos, sin), then
    // the arithmetic: one exposed memory latency
    constexpr int NI = (kSR * 16 + kNT - 1) / kNT;
    uint4_t rq[NI][2], rk[NI][2], rv[NI], rc[NI], rs[NI];
    const uint4_t z4 = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int n = 0; n < NI; ++n) {
      const int item = tid + kNT * n, i = item >> 4, c = item & 15;
      rq[n][0] = rq[n][1] = rk[n][0] = rk[n][1] = rv[n] = rc[n] = rs[n] = z4;
      if (i < S) {
        const char* row = a.qkv + static_cast<int64_t>(i) * a.ld_qkv * 2;
        const char* qh = row + static_cast<int64_t>(h) * kDh * 2;
        const char* kh = row + static_cast<int64_t>(H + h) * kDh * 2;
        rq[n][0] = *reinterpret_cast<const uint4_t*>(qh + c * 16);
        rq[n][1] = *reinterpret_cast<const uint4_t*>(qh + (c ^ 8) * 16);
        rk[n][0] = *reinterpret_cast<const uint4_t*>(kh + c * 16);
        rk[n][1] = *reinterpret_cast<const uint4_t*>(kh + (c ^ 8) * 16);
        rv[n] = *reinterpret_cast<const uint4_t*>(row + static_cast<int64_t>(2 * H + h) * kDh * 2 + c * 16);
        rc[n] = *reinterpret_cast<const uint4_t*>(a.cos + static_cast<int64_t>(i) * kDh * 2 + c * 16);
        rs[n] = *reinterpret_cast<const uint4_t*>(a.sin + static_cast<int64_t>(i) * kDh * 2 + c * 16);
      }
    }
#pragma unroll
    for (int n = 0; n < NI; ++n) {
      const int item = tid + kNT * n, i = item >> 4, c = item & 15;
      if (i < kSR) {
        *reinterpret_cast<uint4_t*>(Qs + i * kLdN + c * 8) = rope_regs<DT>(rq[n][0], rq[n][1], rc[n], rs[n], c);
        *reinterpret_cast<uint4_t*>(Ks + i * kLdN + c * 8) = rope_regs<DT>(rk[n][0], rk[n][1], rc[n], rs[n], c);
        *reinterpret_cast<uint4_t*>(Vs + i * kLdN + c * 8) = rv[n];
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int n = 0; n < kTN; ++n) {
    const int item = tid + kNT * n;
    if (item < kTI) {
      const int d = item & (kDh - 1), ic = item / kDh;
      *reinterpret_cast<uint4_t*>(Vt + d * kLdT + ic * 8) = gather_col(Vs, d, ic);
    }
  }

  // scores with the query row in lane%16: D[j' = 4 fg + e][i' = frow] = K[j'] . Q[i']   (one query tile per wave)
  for (int it = wave; it < kRT; it += kNWv) {
    f32x4 s[kRT];
#pragma unroll
    for (int jt = 0; jt < kRT; ++jt) s[jt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int kk = 0; kk < kDh / 32; ++kk) {
      const uint4_t qf = frag(Qs, kLdN, it, kk * 32, frow, fg);
#pragma unroll
      for (int jt = 0; jt < kRT; ++jt)
        if (jt <= it) s[jt] = mfma16<DT>(frag(Ks, kLdN, jt, kk * 32, frow, fg), qf, s[jt]);
    }
    const int i = it * 16 + frow;
    float m = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < kRT; ++jt) {
      if (jt > it) continue;
      float v[4] = {s[jt].x, s[jt].y, s[jt].z, s[jt].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = jt * 16 + fg * 4 + e;
        v[e] = (j <= i && i < S) ? v[e] * a.scale : -INFINITY;
        m = fmaxf(m, v[e]);
      }
      s[jt] = f32x4{v[0], v[1], v[2], v[3]};
    }
    m = fmaxf(m, __shfl_xor(m, 16, BMA_WAVE));
    m = fmaxf(m, __shfl_xor(m, 32, BMA_WAVE));
    const float mm = (m == -INFINITY) ? 0.0f : m;            // (a padding row: every score masked)
    float l = 0.0f;
#pragma unroll
    for (int jt = 0; jt < kRT; ++jt) {
      if (jt > it) continue;
      float v[4] = {s[jt].x, s[jt].y, s[jt].z, s[jt].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = expf(v[e] - mm);                               // exp(-inf) = 0 for the masked ones
        l += v[e];
      }
      s[jt] = f32x4{v[0], v[1], v[2], v[3]};
    }
    l += __shfl_xor(l, 16, BMA_WAVE);
    l += __shfl_xor(l, 32, BMA_WAVE);
    const float inv = l > 0.0f ? 1.0f / l : 0.0f;
#pragma unroll
    for (int jt = 0; jt < kRT; ++jt)
      if (jt <= it) store4<DT>(Ps + i * kLdT + jt * 16 + fg * 4, s[jt].x * inv, s[jt].y * inv, s[jt].z * inv, s[jt].w * inv);
    if (fg == 0 && i < S) a.lse[static_cast<int64_t>(h) * S + i] = mm + logf(l);
  }
  __syncthreads();

  // O = P.V with the roles swapped -- A = V^T rows (d), B = P rows (i): D[d' = 4 fg + e][i' = frow] -- so that a lane holds four
  // consecutive d of one output row: 8-byte global stores.  Ten jobs: (query tile, half of d).
  for (int job = wave; job < 2 * kRT; job += kNWv) {
    const int it = job >> 1, d0 = (job & 1) * (kDh / 32);
    f32x4 o[kDh / 32];
#pragma unroll
    for (int dt = 0; dt < kDh / 32; ++dt) o[dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const int ksteps = ((it + 1) * 16 + 31) / 32;
    for (int kk = 0; kk < ksteps; ++kk) {
      const uint4_t pf = frag(Ps, kLdT, it, kk * 32, frow, fg);
#pragma unroll
      for (int dt = 0; dt < kDh / 32; ++dt) o[dt] = mfma16<DT>(frag(Vt, kLdT, d0 + dt, kk * 32, frow, fg), pf, o[dt]);
    }
    const int i = it * 16 + frow;
    if (i < S) {
      char* dst = a.out + (static_cast<int64_t>(i) * a.ld_out + h * kDh + fg * 4) * 2;
#pragma unroll
      for (int dt = 0; dt < kDh / 32; ++dt) {
        bma::uint2_t w;
        w.x = bma::pack16<DT>(o[dt].x, o[dt].y);
        w.y = bma::pack16<DT>(o[dt].z, o[dt].w);
        *reinterpret_cast<bma::uint2_t*>(dst + (d0 + dt) * 32) = w;
      }
    }
  }
}

// ----------------------------------------------------------------------------------------------------------- backward
template <int DT>
__global__ __launch_bounds__(kNT) void b1_attn_bwd_kernel(Args a) {
  constexpr int kNat = kSR * kLdN;           // elements of a [row][d] image
  constexpr int kTr = kDh * kLdT;            // ... of a [d][row] image
  constexpr int kSc = kSR * kLdT;            // ... of a [row][row] image
  static_assert(3 * kTr <= 4 * kNat, "the transposed images live where the natural ones were");
  __shared__ __attribute__((aligned(16))) unsigned char lds[(4 * kNat + 3 * kSc) * 2 + 2 * kSR * 4];
  uint16_t* Qs = reinterpret_cast<uint16_t*>(lds);   // phase A: natural images
  uint16_t* Ks = Qs + kNat;
  uint16_t* Vs = Ks + kNat;
  uint16_t* Gs = Vs + kNat;                          // dO
  uint16_t* Gt = reinterpret_cast<uint16_t*>(lds);   // phase B: [d][i] images over the same bytes
  uint16_t* Qt = Gt + kTr;
  uint16_t* Kt = Qt + kTr;
  uint16_t* Pt = Qs + 4 * kNat;                      // P^T  [j][i]
  uint16_t* Dt = Pt + kSc;                           // dS^T [j][i]
  uint16_t* Ds = Dt + kSc;                           // dS   [i][j]
  float* delta = reinterpret_cast<float*>(Ds + kSc);
  float* lses = delta + kSR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fg = lane >> 4;
  const int h = blockIdx.x, S = a.S, H = a.H;

  zero_lds(reinterpret_cast<unsigned char*>(Pt), 3 * kSc * 2, tid);
  {
    // every global load of this thread first, then the arithmetic
    constexpr int NI = (kSR * 16 + kNT - 1) / kNT;
    uint4_t rq[NI][2], rk[NI][2], rv[NI], rc[NI], rs[NI], ro[NI], rg[NI];
    float ls[NI];
    const uint4_t z4 = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int n = 0; n < NI; ++n) {
      const int item = tid + kNT * n, i = item >> 4, c = item & 15;
      rq[n][0] = rq[n][1] = rk[n][0] = rk[n][1] = rv[n] = rc[n] = rs[n] = ro[n] = rg[n] = z4;
      ls[n] = 0.0f;
      if (i < S) {
        const char* row = a.qkv + static_cast<int64_t>(i) * a.ld_qkv * 2;
        const char* qh = row + static_cast<int64_t>(h) * kDh * 2;
        const char* kh = row + static_cast<int64_t>(H + h) * kDh * 2;
        rq[n][0] = *reinterpret_cast<const uint4_t*>(qh + c * 16);
        rq[n][1] = *reinterpret_cast<const uint4_t*>(qh + (c ^ 8) * 16);
        rk[n][0] = *reinterpret_cast<const uint4_t*>(kh + c * 16);
        rk[n][1] = *reinterpret_cast<const uint4_t*>(kh + (c ^ 8) * 16);
        rv[n] = *reinterpret_cast<const uint4_t*>(row + static_cast<int64_t>(2 * H + h) * kDh * 2 + c * 16);
        rc[n] = *reinterpret_cast<const uint4_t*>(a.cos + static_cast<int64_t>(i) * kDh * 2 + c * 16);
        rs[n] = *reinterpret_cast<const uint4_t*>(a.sin + static_cast<int64_t>(i) * kDh * 2 + c * 16);
        rg[n] = *reinterpret_cast<const uint4_t*>(a.dout + (static_cast<int64_t>(i) * a.ld_dout + h * kDh) * 2 + c * 16);
        ro[n] = *reinterpret_cast<const uint4_t*>(a.out + (static_cast<int64_t>(i) * a.ld_out + h * kDh) * 2 + c * 16);
        if (c == 0) ls[n] = a.lse[static_cast<int64_t>(h) * S + i];
      }
    }
#pragma unroll
    for (int n = 0; n < NI; ++n) {
      const int item = tid + kNT * n, i = item >> 4, c = item & 15;
      float g[8], o[8], part = 0.0f;
      unpack8<DT>(rg[n], g);
      unpack8<DT>(ro[n], o);
#pragma unroll
      for (int j = 0; j < 8; ++j) part += g[j] * o[j];
      // the 16 chunks of a row sit in 16 consecutive lanes: delta[i] = sum over them (every lane takes part)
      part += __shfl_xor(part, 1, BMA_WAVE);
      part += __shfl_xor(part, 2, BMA_WAVE);
      part += __shfl_xor(part, 4, BMA_WAVE);
      part += __shfl_xor(part, 8, BMA_WAVE);
      if (i < kSR) {
        if (c == 0) {
          delta[i] = part;
          lses[i] = ls[n];
        }
        *reinterpret_cast<uint4_t*>(Qs + i * kLdN + c * 8) = rope_regs<DT>(rq[n][0], rq[n][1], rc[n], rs[n], c);
        *reinterpret_cast<uint4_t*>(Ks + i * kLdN + c * 8) = rope_regs<DT>(rk[n][0], rk[n][1], rc[n], rs[n], c);
        *reinterpret_cast<uint4_t*>(Vs + i * kLdN + c * 8) = rv[n];
        *reinterpret_cast<uint4_t*>(Gs + i * kLdN + c * 8) = rg[n];
      }
    }
  }
  __syncthreads();

  // ---- phase A: P and dS of every (query tile, key tile) at or below the diagonal, in both orientations -----------------
  for (int t = wave; t < kRT * (kRT + 1) / 2; t += kNWv) {
    int it = 0, rest = t;
    while (rest > it) { rest -= it + 1; ++it; }
    const int jt = rest;
    f32x4 s1 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, p1 = s1, s2 = s1, p2 = s1;
#pragma unroll
    for (int kk = 0; kk < kDh / 32; ++kk) {
      const uint4_t qf = frag(Qs, kLdN, it, kk * 32, frow, fg), kf = frag(Ks, kLdN, jt, kk * 32, frow, fg);
      const uint4_t gf = frag(Gs, kLdN, it, kk * 32, frow, fg), vf = frag(Vs, kLdN, jt, kk * 32, frow, fg);
      s1 = mfma16<DT>(qf, kf, s1);           // D[i' = 4 fg + e][j' = frow]
      p1 = mfma16<DT>(gf, vf, p1);
      s2 = mfma16<DT>(kf, qf, s2);           // D[j' = 4 fg + e][i' = frow]
      p2 = mfma16<DT>(vf, gf, p2);
    }
    {
      const float sv[4] = {s1.x, s1.y, s1.z, s1.w}, pv[4] = {p1.x, p1.y, p1.z, p1.w};
      float P[4], D[4];
      const int j = jt * 16 + frow;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = it * 16 + fg * 4 + e;
        const bool ok = j <= i && i < S;
        P[e] = ok ? expf(sv[e] * a.scale - lses[i]) : 0.0f;
        D[e] = ok ? P[e] * (pv[e] - delta[i]) * a.scale : 0.0f;
      }
      store4<DT>(Pt + j * kLdT + it * 16 + fg * 4, P[0], P[1], P[2], P[3]);
      store4<DT>(Dt + j * kLdT + it * 16 + fg * 4, D[0], D[1], D[2], D[3]);
    }
    {
      const float sv[4] = {s2.x, s2.y, s2.z, s2.w}, pv[4] = {p2.x, p2.y, p2.z, p2.w};
      float D[4];
      const int i = it * 16 + frow;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = jt * 16 + fg * 4 + e;
        const bool ok = j <= i && i < S;
        const float P = ok ? expf(sv[e] * a.scale - lses[i]) : 0.0f;
        D[e] = ok ? P * (pv[e] - delta[i]) * a.scale : 0.0f;
      }
      store4<DT>(Ds + i * kLdT + jt * 16 + fg * 4, D[0], D[1], D[2], D[3]);
    }
  }
  __syncthreads();

  // ---- phase B: the [d][row] images of dO, Q, K go where the natural ones were: read into registers, barrier, write ----------
  {
    uint4_t tg[kTN], tq[kTN], tk[kTN];
#pragma unroll
    for (int n = 0; n < kTN; ++n) {
      const int item = tid + kNT * n, d = item & (kDh - 1), ic = item / kDh;
      if (item < kTI) {
        tg[n] = gather_col(Gs, d, ic);
        tq[n] = gather_col(Qs, d, ic);
        tk[n] = gather_col(Ks, d, ic);
      }
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < kTN; ++n) {
      const int item = tid + kNT * n, d = item & (kDh - 1), ic = item / kDh;
      if (item < kTI) {
        *reinterpret_cast<uint4_t*>(Gt + d * kLdT + ic * 8) = tg[n];
        *reinterpret_cast<uint4_t*>(Qt + d * kLdT + ic * 8) = tq[n];
        *reinterpret_cast<uint4_t*>(Kt + d * kLdT + ic * 8) = tk[n];
      }
    }
  }
  __syncthreads();

  // ---- dV = P^T.dO, dK = dS^T.Q, dQ = dS.K: 30 jobs of (one row tile, half of d), roles swapped -- A = the [d][row] image, B =
  // the score-shaped one: D[d' = 4 fg + e][row' = frow] -- so that a lane holds four consecutive d of one row (8-byte
  // stores).  The halves of d are tiles {0,1,4,5} and {2,3,6,7}: an element and its rotation partner 64 away stay together.
  for (int job = wave; job < 6 * kRT; job += kNWv) {
    const int kind = job / (2 * kRT), rest = job - kind * 2 * kRT, rt = rest >> 1, half = rest & 1;   // 0: dV, 1: dK, 2: dQ
    const uint16_t* Bm = kind == 0 ? Pt : (kind == 1 ? Dt : Ds);
    const uint16_t* Am = kind == 0 ? Gt : (kind == 1 ? Qt : Kt);
    f32x4 acc[4];                                                // tiles half*2 + {0, 1, 4, 5}
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int kk = 0; kk < kSK / 32; ++kk) {
      const uint4_t bf = frag(Bm, kLdT, rt, kk * 32, frow, fg);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        acc[u] = mfma16<DT>(frag(Am, kLdT, half * 2 + (u & 1) + (u >> 1) * 4, kk * 32, frow, fg), bf, acc[u]);
    }
    const int r = rt * 16 + frow;
    if (r >= S) continue;
    const int head = kind == 0 ? 2 * H + h : (kind == 1 ? H + h : h);
    char* dst = a.dqkv + (static_cast<int64_t>(r) * a.ld_dqkv + head * kDh + fg * 4) * 2;
    const char* cr = a.cos + (static_cast<int64_t>(r) * kDh + fg * 4) * 2;
    const char* sr = a.sin + (static_cast<int64_t>(r) * kDh + fg * 4) * 2;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int dt = half * 2 + (u & 1) + (u >> 1) * 4;
      float o[4] = {rnd<DT>(acc[u].x), rnd<DT>(acc[u].y), rnd<DT>(acc[u].z), rnd<DT>(acc[u].w)};
      if (kind != 0) {
        // the rotation's backward: the inverse rotation (bma_rope2 with the sign of sin flipped) of the 16-bit gradient;
        // the partner 64 elements away is the tile four further, u ^ 2 here
        const float p[4] = {rnd<DT>(acc[u ^ 2].x), rnd<DT>(acc[u ^ 2].y), rnd<DT>(acc[u ^ 2].z), rnd<DT>(acc[u ^ 2].w)};
        const bma::uint2_t cw = *reinterpret_cast<const bma::uint2_t*>(cr + dt * 32);
        const bma::uint2_t sw = *reinterpret_cast<const bma::uint2_t*>(sr + dt * 32);
        const float cf[4] = {bma::unpack16<DT>(cw.x, 0), bma::unpack16<DT>(cw.x, 1), bma::unpack16<DT>(cw.y, 0), bma::unpack16<DT>(cw.y, 1)};
        const float sf[4] = {bma::unpack16<DT>(sw.x, 0), bma::unpack16<DT>(sw.x, 1), bma::unpack16<DT>(sw.y, 0), bma::unpack16<DT>(sw.y, 1)};
        const float sign = dt < 4 ? 1.0f : -1.0f;              // -(rotate_half's sign)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rnd<DT>(rnd<DT>(o[e] * cf[e]) + rnd<DT>(sign * p[e] * sf[e]));
      }
      bma::uint2_t w;
      w.x = bma::pack16<DT>(o[0], o[1]);
      w.y = bma::pack16<DT>(o[2], o[3]);
      *reinterpret_cast<bma::uint2_t*>(dst + dt * 32) = w;
    }
  }
}

int check(const Args& a, int dtype, bool bwd) {
  if (a.S < 0 || a.H <= 0) return BMA_EINVAL;
  if (a.S == 0) return BMA_OK;
  if (dtype != BMA_BF16 && dtype != BMA_F16) return BMA_EDTYPE;
  if (a.S > kSR) return BMA_ELIMIT;
  if (!a.qkv || !a.cos || !a.sin || !a.out || !a.lse || (bwd && (!a.dout || !a.dqkv))) return BMA_EINVAL;
  if (a.ld_qkv < 3 * static_cast<int64_t>(a.H) * kDh || a.ld_out < static_cast<int64_t>(a.H) * kDh) return BMA_EINVAL;
  if (bwd && (a.ld_dout < static_cast<int64_t>(a.H) * kDh || a.ld_dqkv < 3 * static_cast<int64_t>(a.H) * kDh)) return BMA_EINVAL;
  if ((a.ld_qkv * 2) % 16 || (a.ld_out * 2) % 16 || (bwd && ((a.ld_dout * 2) % 16 || (a.ld_dqkv * 2) % 8))) return BMA_EALIGN;
  if ((reinterpret_cast<uintptr_t>(a.qkv) | reinterpret_cast<uintptr_t>(a.cos) | reinterpret_cast<uintptr_t>(a.sin) |
       reinterpret_cast<uintptr_t>(a.out)) % 16 || reinterpret_cast<uintptr_t>(a.lse) % 4)
    return BMA_EALIGN;
  if (bwd && (reinterpret_cast<uintptr_t>(a.dout) % 16 || reinterpret_cast<uintptr_t>(a.dqkv) % 8)) return BMA_EALIGN;
  return 1;
}

}  // namespace

extern "C" int bma_b1_attention(const void* qkv, int64_t ld_qkv, const void* cos, const void* sin, int S, int H, int dtype,
                                float scale, void* out, int64_t ld_out, float* lse, void* stream) {
  Args a{};
  a.qkv = static_cast<const char*>(qkv); a.cos = static_cast<const char*>(cos); a.sin = static_cast<const char*>(sin);
  a.out = static_cast<char*>(out); a.lse = lse; a.ld_qkv = ld_qkv; a.ld_out = ld_out; a.S = S; a.H = H; a.scale = scale;
  const int rc = check(a, dtype, false);
  if (rc != 1) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  BMA_PROF_BEGIN(BMA_K_B1_ATTN, st, 4.0 * S * H * kDh * 2.0);
  if (dtype == BMA_BF16) hipLaunchKernelGGL((b1_attn_fwd_kernel<BMA_BF16>), dim3(H), dim3(kNT), 0, st, a);
  else hipLaunchKernelGGL((b1_attn_fwd_kernel<BMA_F16>), dim3(H), dim3(kNT), 0, st, a);
  BMA_PROF_END(BMA_K_B1_ATTN, st);
  BMA_LAUNCH_CHECK();
  return BMA_OK;
}

extern "C" int bma_b1_attention_bwd(const void* qkv, int64_t ld_qkv, const void* cos, const void* sin, const void* out,
                                    int64_t ld_out, const float* lse, const void* dout, int64_t ld_dout, int S, int H, int dtype,
                                    float scale, void* dqkv, int64_t ld_dqkv, void* stream) {
  Args a{};
  a.qkv = static_cast<const char*>(qkv); a.cos = static_cast<const char*>(cos); a.sin = static_cast<const char*>(sin);
  a.out = static_cast<char*>(const_cast<void*>(out)); a.lse = const_cast<float*>(lse);
  a.dout = static_cast<const char*>(dout); a.dqkv = static_cast<char*>(dqkv);
  a.ld_qkv = ld_qkv; a.ld_out = ld_out; a.ld_dout = ld_dout; a.ld_dqkv = ld_dqkv; a.S = S; a.H = H; a.scale = scale;
  const int rc = check(a, dtype, true);
  if (rc != 1) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  BMA_PROF_BEGIN(BMA_K_B1_ATTN, st, 8.0 * S * H * kDh * 2.0);
  if (dtype == BMA_BF16) hipLaunchKernelGGL((b1_attn_bwd_kernel<BMA_BF16>), dim3(H), dim3(kNT), 0, st, a);
  else hipLaunchKernelGGL((b1_attn_bwd_kernel<BMA_F16>), dim3(H), dim3(kNT), 0, st, a);
  BMA_PROF_END(BMA_K_B1_ATTN, st);
  BMA_LAUNCH_CHECK();
  return BMA_OK;
}
