// a1 (gradient pass) -- causal self-attention of ONE long sequence at batch 1, forward and backward.
//
// With PGD on, the reference's compute_gradient (bimodal_attack.py:953-1028) runs the decoder over 576 image rows plus the
// prompt at batch 1: per layer one causal attention of 599-644 tokens x 32 heads x 128 and its backward.  That is 3.4 + 8.5
// GFLOP -- nothing -- and the library's kernels spend 51 us (forward) and ~100 us (backward + its helper launches) on
// it because a (head, 64-query) grid of a few hundred workgroups walks its key tiles one latency at a time.  The same
// shape occurs behind a reused prefix (joint mode: 44 new rows against 643 keys).
//
// Three launches, all built like prefix_attention.hip's flash forward (v_mfma_f32_16x16x32, 32-row chunks staged
// L2 -> registers -> LDS while the previous ones are multiplied, the exponentiated accumulator packed straight into the
// next product's B operand, `ds_read_b64_tr_b16` for the transposed operand); a workgroup is 8 waves, 4 row tiles x the
// even / odd chunks of the other side, two chunks per barrier:
//   forward   (head, 64 queries): S^T = K Q^T, online softmax per query lane, O^T += V^T P^T; writes o and
//             lse2 = m*scale*log2(e) + log2(l)
//   dq        (head, 64 queries): S^T and dP^T = V dO^T recomputed per key chunk, dS^T = P^T (dP^T - delta) scale,
//             dQ^T += K^T dS^T; also writes delta = rowsum(dO o) for the third launch
//   dk, dv    (head, 64 keys), a wave per 16 keys: per 32-query chunk S = Q K^T and dP = dO V^T (queries as rows, so that
//             P and dS come out as the B operands of) dV^T += dO^T P, dK^T += Q^T dS
// No atomics: every output element has one owner, results are bitwise reproducible.
//
// Queries are the LAST Lq positions of the Lk keys (P = Lk - Lq keys of prefix in front): query i sees keys 0 .. P + i.
// H query heads over H / rep key/value heads (rep = 1: LLaVA, the towers; rep = 2: Gemma-3's decoder) of 64, 72, 128 or 256;
// q/k/v through (row, head) strides (views of a fused projection), o / dO / dq [rows][H][D] and dk / dv [rows][H / rep][D]
// contiguous.  At 256-wide heads (Gemma-3's decoder, ~320 rows: the library's four launches take 119 us per layer, all
// latency) a chunk is two 16-byte pieces per thread and the dk/dv launch splits the OUTPUT dims between its two wave halves
// instead of the chunks (each accumulator pair then fits the register file without spilling).
#include <type_traits>

#include "bma_common.h"
#include "bma_profile.h"

namespace {

using bma::uint4_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short short4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// per head width DH (64 or 128): KS = DH / 32 k-steps of a product over the head dimension, NT = DH / 16 16-dim tiles of an
// output, PITCH = DH + 16 elements per LDS row (row + 32 B: conflict-free row and transposing reads), IMG = 32 * PITCH one
// 32-row chunk image; a chunk is 32 * DH / 8 16-byte pieces, at most one per thread
constexpr int NTHR = 512;         // 8 waves

struct CArgs {
  const uint16_t *q, *k, *v, *o, *d_o;
  uint16_t *out, *dq, *dk, *dv;
  float *lse2, *delta;            // [H][Lq]
  int64_t q_rs, q_hs, k_rs, k_hs, v_rs, v_hs;
  int64_t d_rs;                   // row stride of dq (elements)
  int64_t dkv_rs;                 // row stride of dk / dv
  int Lq, Lk, H, P;
  int rep;                        // query heads per key/value head (grouped-query attention); H counts QUERY heads
  float scale, scale_log2e;
};

template <int DT>
__device__ __forceinline__ f32x4 cmfma(const uint4_t& a, const uint4_t& b, const f32x4& c) {
  if (DT == BMA_BF16)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ float vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// all-reduce over the four 16-lane rows of a wave (prefix_attention.hip)
__device__ __forceinline__ float rows_max(float x) {
  u32x2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const float m = vmax(__uint_as_float(a.x), __uint_as_float(a.y));
  u32x2 b = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  return vmax(__uint_as_float(b.x), __uint_as_float(b.y));
}
__device__ __forceinline__ float rows_sum(float x) {
  u32x2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const float m = __uint_as_float(a.x) + __uint_as_float(a.y);
  u32x2 b = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  return __uint_as_float(b.x) + __uint_as_float(b.y);
}

// ---- a 32-row chunk of a [rows][heads][DH] tensor: L2 -> registers -> LDS image [32][PITCH] ----------------------------
template <int DR>
struct Chunk {
  static constexpr int N = (32 * (DR / 8) + NTHR - 1) / NTHR;    // 16-byte pieces per thread: 1, or 2 at 256-wide heads
  uint4_t reg[N];
};
// (DR: the head's REAL width in memory, DH: the width of its LDS image and of the products -- DR = 72, SigLIP's heads, rides
// in 96-wide images whose last 24 columns are zero: zero dims add nothing to q.k and give zero outputs, which are not stored)
template <int DH, int DR>
__device__ __forceinline__ void fetch_chunk(Chunk<DR>& ch, const uint16_t* base, int64_t rs, int row0, int rows, int tid) {
#pragma unroll
  for (int j = 0; j < Chunk<DR>::N; ++j) {
    const int i = tid + j * NTHR;
    if ((32 * (DR / 8)) % NTHR != 0 && i >= 32 * (DR / 8)) return;
    int row = row0 + i / (DR / 8);
    const int piece = i % (DR / 8);
    row = row < rows ? row : rows - 1;                          // rows past the end repeat the last one (masked by the caller)
    ch.reg[j] = *reinterpret_cast<const uint4_t*>(base + static_cast<int64_t>(row) * rs + 8 * piece);
  }
}
template <int DH, int DR>
__device__ __forceinline__ void stash_chunk(const Chunk<DR>& ch, uint16_t* img, int tid) {
#pragma unroll
  for (int j = 0; j < Chunk<DR>::N; ++j) {
    const int i = tid + j * NTHR;
    if ((32 * (DR / 8)) % NTHR != 0 && i >= 32 * (DR / 8)) return;
    *reinterpret_cast<uint4_t*>(img + (i / (DR / 8)) * (DH + 16) + 8 * (i % (DR / 8))) = ch.reg[j];
  }
}
// the columns DR .. DH-1 of `n_img` chunk images, once, before the first stash (a stash never touches them)
template <int DH, int DR>
__device__ __forceinline__ void zero_pad_columns(uint16_t* lds, int n_img, int tid) {
  if (DR == DH) return;
  constexpr int PP = (DH - DR) / 8;                              // 16-byte pad pieces per row
  for (int i = tid; i < n_img * 32 * PP; i += NTHR) {
    const int row = i / PP, piece = i % PP;                      // row over all images: the images are contiguous, 32 rows each
    *reinterpret_cast<uint4_t*>(lds + row * (DH + 16) + DR + 8 * piece) = uint4_t{0u, 0u, 0u, 0u};
  }
}
// operand with the image's rows 16t .. 16t+15 as the M / N index and dims 32ks .. as k (lane: row r, dims 8g ..)
template <int DH>
__device__ __forceinline__ uint4_t row_frag(const uint16_t* img, int t, int ks, int r, int g) {
  return *reinterpret_cast<const uint4_t*>(img + (16 * t + r) * (DH + 16) + 8 * g + 32 * ks);
}
// operand with dims 16dt .. 16dt+15 as the M index and the image's 32 rows as k, in the order pack_acc() leaves them
template <int DH>
__device__ __forceinline__ uint4_t tr_frag(const uint16_t* img, int dt, int r, int g) {
  constexpr int PITCH = DH + 16;
  const int q4 = r >> 2, p4 = r & 3;
  const uint16_t* rd = img + (4 * g + q4) * PITCH + 4 * p4 + 16 * dt;
  const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(rd));
  const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(rd + 16 * PITCH));
  const bma::uint2_t l2 = __builtin_bit_cast(bma::uint2_t, lo), h2 = __builtin_bit_cast(bma::uint2_t, hi);
  uint4_t f;
  f.x = l2.x; f.y = l2.y; f.z = h2.x; f.w = h2.y;
  return f;
}
// two accumulators (chunk rows 4g+rr and 16+4g+rr of one column) as the B operand whose k runs over the chunk's rows
template <int DT>
__device__ __forceinline__ uint4_t pack_acc(const float (&e)[2][4]) {
  uint4_t p;
  p.x = bma::pack16<DT>(e[0][0], e[0][1]);
  p.y = bma::pack16<DT>(e[0][2], e[0][3]);
  p.z = bma::pack16<DT>(e[1][0], e[1][1]);
  p.w = bma::pack16<DT>(e[1][2], e[1][3]);
  return p;
}
// this lane's dims (8g + 32ks ..) of row `row` of a [rows][H][DH] tensor, as the K-contiguous operand of its column
template <int KS, int DR>
__device__ __forceinline__ void load_row_frags(uint4_t (&f)[KS], const uint16_t* base, int64_t rs, int row, int g) {
  const uint16_t* p = base + static_cast<int64_t>(row) * rs + 8 * g;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (32 * ks + 32 <= DR || 8 * g + 32 * ks < DR) f[ks] = *reinterpret_cast<const uint4_t*>(p + 32 * ks);
    else f[ks] = uint4_t{0u, 0u, 0u, 0u};                        // dims past the real width: the image's zero columns
  }
}

// The operands a wave loads ONCE must have landed before the chunk loop is entered: hipcc's wait insertion otherwise carries
// "still pending" into the loop and puts counted `s_waitcnt vmcnt` in front of their first uses INSIDE it -- waits which, from
// the second trip on, drain the chunk prefetch that was issued a few instructions earlier.  A use in an empty asm statement
// makes it wait here.
template <int KS>
__device__ __forceinline__ void landed(const uint4_t (&f)[KS]) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) asm volatile("" ::"v"(f[ks]));
}

// A workgroup is 8 waves: wave w works on the 16 rows (w & 3) of the block, and the two halves (w >> 2) take the even and
// the odd 32-row chunks of the other side -- a trip of the loop stages TWO chunks (one barrier) and each wave's chain of
// dependent products and softmax steps is half as long as the sequence; the halves' partial results meet in LDS at the end.
// (One wave per 16 rows over all chunks: 1.07 us per chunk, 25 us forward and 58 us backward at 643 tokens.)
template <int DR>
struct Pair {
  Chunk<DR> x0, y0, x1, y1;                                      // (K, V) or (Q, dO) images of chunks 2t and 2t+1
};
template <int DH, int DR>
__device__ __forceinline__ void fetch_pair(Pair<DR>& p, const uint16_t* xb, int64_t x_rs, const uint16_t* yb, int64_t y_rs, int t,
                                           int rows, int tid) {
  fetch_chunk<DH, DR>(p.x0, xb, x_rs, 64 * t, rows, tid);
  fetch_chunk<DH, DR>(p.y0, yb, y_rs, 64 * t, rows, tid);
  fetch_chunk<DH, DR>(p.x1, xb, x_rs, 64 * t + 32, rows, tid);
  fetch_chunk<DH, DR>(p.y1, yb, y_rs, 64 * t + 32, rows, tid);
}
template <int DH, int DR>
__device__ __forceinline__ void stash_pair(const Pair<DR>& p, uint16_t* buf, int tid) {
  constexpr int IMG = 32 * (DH + 16);
  stash_chunk<DH, DR>(p.x0, buf, tid);
  stash_chunk<DH, DR>(p.y0, buf + IMG, tid);
  stash_chunk<DH, DR>(p.x1, buf + 2 * IMG, tid);
  stash_chunk<DH, DR>(p.y1, buf + 3 * IMG, tid);
}
// does this lane's 4-dim group of output tile dt (dims 16dt + 4g ..) exist in memory?
template <int DR>
__device__ __forceinline__ bool dims_real(int dt, int g) {
  return 16 * dt + 16 <= DR || 16 * dt + 4 * g < DR;
}

// ------------------------------------------------------------------------------------------------------------------------
// TQ: 16-query tiles per wave.  1 is the latency shape (643 tokens: as many workgroups as possible, the shortest chain per
// wave); 2 makes every K fragment (ds_read_b128) and V^T fragment (ds_read_b64_tr_b16) read from LDS feed TWO MFMAs -- the
// throughput shape for a tower of thousands of tokens (SigLIP: 4096 x 16 x 72), where the one-tile kernel spends more issue
// slots on LDS reads than on products (16 reads per 11 MFMAs per chunk at 96-wide images; 16 per 22 with two tiles).
// Measured at SigLIP's 4096 x 16 x 72 (round 5, one box): one tile per wave 212 us; two tiles at the registers the compiler
// wants (146: one workgroup per CU) 210; two tiles compiled for four waves per SIMD (128 VGPRs, 9 dwords spilled: two workgroups
// per CU) 194.  With one tile the loop is bound by the latency of a chunk pair's trip L2 -> registers -> LDS (one pair in
// flight per workgroup, two rounds of workgroups); with two tiles and two workgroups per CU it reaches the issue limit of the
// softmax's VALU work, which at 72-wide heads outweighs the products 1.6 : 1.
#ifndef BMA_CA_FWD_TQ2_WAVES
#define BMA_CA_FWD_TQ2_WAVES 4      // waves per SIMD the two-tile forward is compiled for
#endif
template <int DT, int DH, int DR = DH, int TQ = 1>
__global__ __launch_bounds__(NTHR, (TQ == 2 ? BMA_CA_FWD_TQ2_WAVES : 1)) void causal_fwd_kernel(const CArgs a) {
  constexpr int KS = DH / 32, NT = (DR + 15) / 16, IMG = 32 * (DH + 16);   // (output tiles past the real width are never formed: 5 of 6 at DR = 72)
  constexpr int RWG = 64 * TQ;                                        // query rows of a workgroup
  __shared__ __attribute__((aligned(16))) uint16_t lds[8 * IMG];      // two buffers of (K, V, K, V) images: chunks 2t, 2t+1
  static_assert(4 * 64 * TQ * (4 + 4 * NT) * 4 <= 8 * IMG * 2, "the halves' exchange must fit the chunk images");
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, g = lane >> 4;
  const int half = w >> 2, wq = w & 3;
  const int h = blockIdx.x % a.H, qb = blockIdx.x / a.H;
  const int row0 = RWG * qb + 16 * TQ * wq;                           // first query of this wave
  const float NEG = -__builtin_inff();
  uint4_t qf[TQ][KS];
  f32x4 oacc[TQ][NT];
  float mrun[TQ], lsum[TQ];
#pragma unroll
  for (int t = 0; t < TQ; ++t) {
    const int qrow = row0 + 16 * t + r;                               // this lane's query of tile t
    load_row_frags<KS, DR>(qf[t], a.q + static_cast<int64_t>(h) * a.q_hs, a.q_rs, qrow < a.Lq ? qrow : a.Lq - 1, g);
#pragma unroll
    for (int dt = 0; dt < NT; ++dt) oacc[t][dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    mrun[t] = NEG;
    lsum[t] = 0.0f;
  }
  const uint16_t* kb = a.k + static_cast<int64_t>(h / a.rep) * a.k_hs;     // (grouped queries: `rep` query heads read one k/v head)
  const uint16_t* vb = a.v + static_cast<int64_t>(h / a.rep) * a.v_hs;
  int last = RWG * qb + RWG - 1;
  last = last < a.Lq ? last : a.Lq - 1;
  int chunks = (a.P + last + 1 + 31) >> 5;                         // keys 0 .. P + last
  chunks = chunks < ((a.Lk + 31) >> 5) ? chunks : (a.Lk + 31) >> 5;   // (not causal: P is past every key)
  const int trips = (chunks + 1) >> 1;
  Pair<DR> pr;
  fetch_pair<DH, DR>(pr, kb, a.k_rs, vb, a.v_rs, 0, a.Lk, tid);
  zero_pad_columns<DH, DR>(lds, 8, tid);
  stash_pair<DH, DR>(pr, lds, tid);
#pragma unroll
  for (int t = 0; t < TQ; ++t) landed(qf[t]);
  __syncthreads();
  for (int tr = 0; tr < trips; ++tr) {
    const uint16_t* kl = lds + 4 * IMG * (tr & 1) + 2 * IMG * half;
    const uint16_t* vl = kl + IMG;
    if (tr + 1 < trips) fetch_pair<DH, DR>(pr, kb, a.k_rs, vb, a.v_rs, tr + 1, a.Lk, tid);
    const int c = 2 * tr + half;
    if (c < chunks) {
      f32x4 s[TQ][2];
#pragma unroll
      for (int t = 0; t < TQ; ++t) {
        s[t][0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        s[t][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      }
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const uint4_t kf = row_frag<DH>(kl, kt, ks, r, g);           // one read, TQ products
#pragma unroll
          for (int t = 0; t < TQ; ++t) s[t][kt] = cmfma<DT>(kf, qf[t][ks], s[t][kt]);
        }
      uint4_t pf[TQ];
      bool live[TQ];
#pragma unroll
      for (int t = 0; t < TQ; ++t) {
        const int trow0 = row0 + 16 * t, qrow = trow0 + r;
        float e[2][4];
        const bool edge = 32 * c + 31 > a.P + trow0 || 32 * c + 31 >= a.Lk;   // wave-uniform: some key of the chunk is masked for some query
        float cmax = NEG;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            float v = s[t][kt][rr];
            if (edge && (32 * c + 16 * kt + 4 * g + rr > a.P + qrow || 32 * c + 16 * kt + 4 * g + rr >= a.Lk)) v = NEG;
            e[kt][rr] = v;
            cmax = vmax(cmax, v);
          }
        cmax = rows_max(cmax);
        // (a tile whose 16 queries see none of this chunk's keys -- only in the block's last chunks -- keeps its state)
        live[t] = __builtin_amdgcn_ballot_w64(cmax > NEG) != 0;
        pf[t] = uint4_t{0u, 0u, 0u, 0u};
        if (live[t]) {
          const float mnew = vmax(mrun[t], cmax);
          const float msafe = mnew > NEG ? mnew : 0.0f;               // a query that has seen no key yet: exponents of -inf, no NaN
          const float alpha = __builtin_amdgcn_exp2f((mrun[t] - msafe) * a.scale_log2e);
          const float mneg = -msafe * a.scale_log2e;
          float rs = 0.0f;
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              e[kt][rr] = __builtin_amdgcn_exp2f(__builtin_fmaf(e[kt][rr], a.scale_log2e, mneg));
              rs += e[kt][rr];
            }
          lsum[t] = lsum[t] * alpha + rs;
          mrun[t] = mnew;
          pf[t] = pack_acc<DT>(e);
          if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
            for (int dt = 0; dt < NT; ++dt) {
              oacc[t][dt][0] *= alpha; oacc[t][dt][1] *= alpha; oacc[t][dt][2] *= alpha; oacc[t][dt][3] *= alpha;
            }
          }
        }
      }
      if (TQ == 1) {
        if (live[0]) {
#pragma unroll
          for (int dt = 0; dt < NT; ++dt) oacc[0][dt] = cmfma<DT>(tr_frag<DH>(vl, dt, r, g), pf[0], oacc[0][dt]);
        }
      } else {
        bool any = false;
#pragma unroll
        for (int t = 0; t < TQ; ++t) any = any || live[t];
        if (any) {                                                       // (a dead tile's P is zero: its products add nothing)
#pragma unroll
          for (int dt = 0; dt < NT; ++dt) {
            const uint4_t vf = tr_frag<DH>(vl, dt, r, g);               // one read, TQ products
#pragma unroll
            for (int t = 0; t < TQ; ++t) oacc[t][dt] = cmfma<DT>(vf, pf[t], oacc[t][dt]);
          }
        }
      }
    }
    if (tr + 1 < trips) stash_pair<DH, DR>(pr, lds + 4 * IMG * ((tr + 1) & 1), tid);
    __syncthreads();
  }
  // ---- the odd-chunk half hands (m, l, o) over through LDS; the even half merges and stores -------------------------------
#pragma unroll
  for (int t = 0; t < TQ; ++t) {
    float* xch = reinterpret_cast<float*>(lds) + ((wq * 64 + lane) * TQ + t) * (4 + 4 * NT);     // 16-byte aligned rows: m, l, -, -, o[..]
    const float lfull = rows_sum(lsum[t]);
    lsum[t] = lfull;
    if (half == 1) {
      xch[0] = mrun[t];
      xch[1] = lfull;
#pragma unroll
      for (int dt = 0; dt < NT; ++dt) *reinterpret_cast<f32x4*>(xch + 4 + 4 * dt) = oacc[t][dt];
    }
  }
  __syncthreads();
  if (half == 1) return;
#pragma unroll
  for (int t = 0; t < TQ; ++t) {
    const int qrow = row0 + 16 * t + r;
    if (qrow >= a.Lq) continue;
    const float* xch = reinterpret_cast<const float*>(lds) + ((wq * 64 + lane) * TQ + t) * (4 + 4 * NT);
    const float m2 = xch[0], l2 = xch[1];
    const float m = vmax(mrun[t], m2);                                // finite: chunk 0 belongs to this half and shows key 0
    const float a1 = __builtin_amdgcn_exp2f((mrun[t] - m) * a.scale_log2e), a2 = __builtin_amdgcn_exp2f((m2 - m) * a.scale_log2e);
    const float l = lsum[t] * a1 + l2 * a2;
    const float inv = 1.0f / l;
    if (g == 0) a.lse2[static_cast<int64_t>(h) * a.Lq + qrow] = m * a.scale_log2e + __builtin_amdgcn_logf(l);   // v_log_f32 = log2
    uint16_t* op = a.out + (static_cast<int64_t>(qrow) * a.H + h) * DR + 4 * g;
    const float c1 = a1 * inv, c2 = a2 * inv;
#pragma unroll
    for (int dt = 0; dt < NT; ++dt) {
      if (!dims_real<DR>(dt, g)) continue;
      const f32x4 o2 = *reinterpret_cast<const f32x4*>(xch + 4 + 4 * dt);
      bma::uint2_t ow;
      ow.x = bma::pack16<DT>(oacc[t][dt][0] * c1 + o2[0] * c2, oacc[t][dt][1] * c1 + o2[1] * c2);
      ow.y = bma::pack16<DT>(oacc[t][dt][2] * c1 + o2[2] * c2, oacc[t][dt][3] * c1 + o2[3] * c2);
      *reinterpret_cast<bma::uint2_t*>(op + 16 * dt) = ow;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
template <int DT, int DH, int DR = DH>
__global__ __launch_bounds__(NTHR) void causal_dq_kernel(const CArgs a) {
  constexpr int KS = DH / 32, NT = (DR + 15) / 16, IMG = 32 * (DH + 16);   // (output tiles past the real width are never formed: 5 of 6 at DR = 72)
  __shared__ __attribute__((aligned(16))) uint16_t lds[8 * IMG];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, g = lane >> 4;
  const int half = w >> 2, wq = w & 3;
  const int h = blockIdx.x % a.H, qb = blockIdx.x / a.H;
  const int row0 = 64 * qb + 16 * wq;
  const int qrow = row0 + r;
  const int qc = qrow < a.Lq ? qrow : a.Lq - 1;
  uint4_t qf[KS], dof[KS];
  load_row_frags<KS, DR>(qf, a.q + static_cast<int64_t>(h) * a.q_hs, a.q_rs, qc, g);
  load_row_frags<KS, DR>(dof, a.d_o + static_cast<int64_t>(h) * DR, static_cast<int64_t>(a.H) * DR, qc, g);
  float delta;
  {
    uint4_t of[KS];
    load_row_frags<KS, DR>(of, a.o + static_cast<int64_t>(h) * DR, static_cast<int64_t>(a.H) * DR, qc, g);
    float acc = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const uint32_t dw[4] = {dof[ks].x, dof[ks].y, dof[ks].z, dof[ks].w};
      const uint32_t ow[4] = {of[ks].x, of[ks].y, of[ks].z, of[ks].w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc += bma::unpack16<DT>(dw[j], 0) * bma::unpack16<DT>(ow[j], 0) + bma::unpack16<DT>(dw[j], 1) * bma::unpack16<DT>(ow[j], 1);
    }
    delta = rows_sum(acc);
    if (half == 0 && g == 0 && qrow < a.Lq) a.delta[static_cast<int64_t>(h) * a.Lq + qrow] = delta;
  }
  const float lse2 = a.lse2[static_cast<int64_t>(h) * a.Lq + qc];
  f32x4 dqacc[NT];
#pragma unroll
  for (int dt = 0; dt < NT; ++dt) dqacc[dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  const uint16_t* kb = a.k + static_cast<int64_t>(h / a.rep) * a.k_hs;     // (grouped queries: `rep` query heads read one k/v head)
  const uint16_t* vb = a.v + static_cast<int64_t>(h / a.rep) * a.v_hs;
  int last = 64 * qb + 63;
  last = last < a.Lq ? last : a.Lq - 1;
  int chunks = (a.P + last + 1 + 31) >> 5;
  chunks = chunks < ((a.Lk + 31) >> 5) ? chunks : (a.Lk + 31) >> 5;
  const int trips = (chunks + 1) >> 1;
  Pair<DR> pr;
  fetch_pair<DH, DR>(pr, kb, a.k_rs, vb, a.v_rs, 0, a.Lk, tid);
  zero_pad_columns<DH, DR>(lds, 8, tid);
  stash_pair<DH, DR>(pr, lds, tid);
  landed(qf);
  landed(dof);
  asm volatile("" ::"v"(lse2));
  __syncthreads();
  for (int t = 0; t < trips; ++t) {
    const uint16_t* kl = lds + 4 * IMG * (t & 1) + 2 * IMG * half;
    const uint16_t* vl = kl + IMG;
    if (t + 1 < trips) fetch_pair<DH, DR>(pr, kb, a.k_rs, vb, a.v_rs, t + 1, a.Lk, tid);
    const int c = 2 * t + half;
    if (c < chunks) {
      f32x4 s[2], dp[2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        s[kt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        dp[kt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          s[kt] = cmfma<DT>(row_frag<DH>(kl, kt, ks, r, g), qf[ks], s[kt]);
          dp[kt] = cmfma<DT>(row_frag<DH>(vl, kt, ks, r, g), dof[ks], dp[kt]);
        }
      }
      float e[2][4];
      const bool edge = 32 * c + 31 > a.P + row0 || 32 * c + 31 >= a.Lk;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][rr], a.scale_log2e, -lse2));
          if (edge && (32 * c + 16 * kt + 4 * g + rr > a.P + qrow || 32 * c + 16 * kt + 4 * g + rr >= a.Lk)) p = 0.0f;
          e[kt][rr] = p * (dp[kt][rr] - delta) * a.scale;
        }
      const uint4_t dsf = pack_acc<DT>(e);
#pragma unroll
      for (int dt = 0; dt < NT; ++dt) dqacc[dt] = cmfma<DT>(tr_frag<DH>(kl, dt, r, g), dsf, dqacc[dt]);
    }
    if (t + 1 < trips) stash_pair<DH, DR>(pr, lds + 4 * IMG * ((t + 1) & 1), tid);
    __syncthreads();
  }
  float* xch = reinterpret_cast<float*>(lds) + (wq * 64 + lane) * (4 * NT);
  if (half == 1) {
#pragma unroll
    for (int dt = 0; dt < NT; ++dt) *reinterpret_cast<f32x4*>(xch + 4 * dt) = dqacc[dt];
  }
  __syncthreads();
  if (half == 1 || qrow >= a.Lq) return;
  uint16_t* op = a.dq + static_cast<int64_t>(qrow) * a.d_rs + h * DR + 4 * g;
#pragma unroll
  for (int dt = 0; dt < NT; ++dt) {
    if (!dims_real<DR>(dt, g)) continue;
    const f32x4 o2 = *reinterpret_cast<const f32x4*>(xch + 4 * dt);
    bma::uint2_t ow;
    ow.x = bma::pack16<DT>(dqacc[dt][0] + o2[0], dqacc[dt][1] + o2[1]);
    ow.y = bma::pack16<DT>(dqacc[dt][2] + o2[2], dqacc[dt][3] + o2[3]);
    *reinterpret_cast<bma::uint2_t*>(op + 16 * dt) = ow;
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// waves per SIMD the dk/dv kernel is compiled for at head widths <= 96: a workgroup is 8 waves = 2 per SIMD, so 4 lets two
// workgroups share a CU where 3 (what 152 / 132 VGPRs allow) leaves the second one out.  Round 5, SigLIP's 4096 x 16 x 72
// backward pair: 587-590 us at the default bound, 535-545 at 4 (128 VGPRs, 11 dwords spilled); CLIP's 577-token pair unchanged.
#ifndef BMA_CA_DKV_WAVES
#define BMA_CA_DKV_WAVES 4
#endif
template <int DT, int DH, int DR = DH>
__global__ __launch_bounds__(NTHR, (DH <= 96 ? BMA_CA_DKV_WAVES : 1)) void causal_dkv_kernel(const CArgs a) {
  constexpr int KS = DH / 32, NT = (DR + 15) / 16, IMG = 32 * (DH + 16);   // (output tiles past the real width are never formed: 5 of 6 at DR = 72)
  // Up to 128-wide heads the two wave halves take the even and the odd query chunks and own ALL output dims (their sums meet in
  // LDS at the end).  At 256 two full accumulator pairs are 128 registers on top of 64 of key/value fragments: the halves
  // split the OUTPUT dims instead -- both walk every chunk (S and dP computed twice, at a size where the launch is latency)
  // and each stores its own half of dk / dv.
  constexpr bool SPLIT_D = DH > 128;
  constexpr int NTW = SPLIT_D ? NT / 2 : NT;                          // output tiles a wave owns
  static_assert(!SPLIT_D || (NT % 2 == 0 && DR == DH), "the dims split wants an even tile count and no padded columns");
  __shared__ __attribute__((aligned(16))) uint16_t lds[8 * IMG];      // two buffers of (Q, dO, Q, dO) images: query chunks 2t, 2t+1
  __shared__ __attribute__((aligned(16))) float stat[2][2][64];       // per buffer and chunk: lse2 of its 32 queries, then delta
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, g = lane >> 4;
  const int half = w >> 2, wk = w & 3;
  const int Hkv = a.H / a.rep;
  const int h = blockIdx.x % Hkv, kblk = blockIdx.x / Hkv;        // h: the key/value head; its `rep` query heads follow each other
  const int key0 = 64 * kblk + 16 * wk;                           // first key of this wave
  const int key = key0 + r;                                       // this lane's key (a column of S)
  const int dt0 = SPLIT_D ? half * NTW : 0;
  uint4_t kf[KS], vf[KS];
  load_row_frags<KS, DR>(kf, a.k + static_cast<int64_t>(h) * a.k_hs, a.k_rs, key < a.Lk ? key : a.Lk - 1, g);
  load_row_frags<KS, DR>(vf, a.v + static_cast<int64_t>(h) * a.v_hs, a.v_rs, key < a.Lk ? key : a.Lk - 1, g);
  f32x4 dkacc[NTW], dvacc[NTW];
#pragma unroll
  for (int dt = 0; dt < NTW; ++dt) {
    dkacc[dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    dvacc[dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  }
  const int64_t do_rs = static_cast<int64_t>(a.H) * DR;
  // queries that see at least one key of this block: i >= 64*kblk - P; chunk pairs t0 .. t1-1 of 64 queries, once per query
  // head of the group: `it` runs over (query head of the group, chunk pair)
  int q0 = 64 * kblk - a.P;
  q0 = q0 > 0 ? q0 : 0;
  const int t0 = q0 >> 6, t1 = (a.Lq + 63) >> 6;
  const int nt = t1 - t0, its = nt * a.rep;
  const int c1 = (a.Lq + 31) >> 5;
  Pair<DR> pr;
  float st = 0.0f;
  // the pair to fetch next: chunk pair `ft` of the query head whose bases are fq / fdo / flse / fdelta -- they move on to the
  // group's next head when the pairs of one are through (once per head: the loop itself carries no address arithmetic
  // beyond what the one-head kernel had)
  int ft = t0;
  const uint16_t* fq = a.q + static_cast<int64_t>(h * a.rep) * a.q_hs;
  const uint16_t* fdo = a.d_o + static_cast<int64_t>(h * a.rep) * DR;
  const float* flse = a.lse2 + static_cast<int64_t>(h * a.rep) * a.Lq;
  const float* fdelta = a.delta + static_cast<int64_t>(h * a.rep) * a.Lq;
  auto fetch = [&]() {
    fetch_pair<DH, DR>(pr, fq, a.q_rs, fdo, do_rs, ft, a.Lq, tid);
    if (tid < 128) {                                               // lse2 and delta of the pair's 64 queries
      int qi = 64 * ft + 32 * (tid >> 6) + (tid & 31);
      qi = qi < a.Lq ? qi : a.Lq - 1;
      st = ((tid & 32) ? fdelta : flse)[qi];
    }
    if (++ft == t1) {
      ft = t0;
      fq += a.q_hs;
      fdo += DR;
      flse += a.Lq;
      fdelta += a.Lq;
    }
  };
  auto stash = [&](int buf) {
    stash_pair<DH, DR>(pr, lds + 4 * IMG * buf, tid);
    if (tid < 128) stat[buf][tid >> 6][tid & 63] = st;
  };
  if (its > 0) fetch();
  zero_pad_columns<DH, DR>(lds, 8, tid);
  if (its > 0) stash(0);
  landed(kf);
  landed(vf);
  __syncthreads();
  int t = t0 - 1;
  for (int it = 0; it < its; ++it) {
    const int buf = it & 1;
    if (++t == t1) t = t0;
    if (it + 1 < its) fetch();
#pragma unroll
    for (int cc = 0; cc < (SPLIT_D ? 2 : 1); ++cc) {
      const int hc = SPLIT_D ? cc : half;                          // which chunk of the pair this wave works on now
      const uint16_t* ql = lds + 4 * IMG * buf + 2 * IMG * hc;
      const uint16_t* dl = ql + IMG;
      const int c = 2 * t + hc;
      if (c < c1 && 32 * c + 31 + a.P >= key0) {                   // (a chunk whose queries all sit in front of this wave's keys adds nothing)
        f32x4 s[2], dp[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          s[qt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
          dp[qt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            s[qt] = cmfma<DT>(row_frag<DH>(ql, qt, ks, r, g), kf[ks], s[qt]);
            dp[qt] = cmfma<DT>(row_frag<DH>(dl, qt, ks, r, g), vf[ks], dp[qt]);
          }
        }
        float pe[2][4], de[2][4];
        // masked: the chunk reaches past Lq, or some of its queries do not see some key of this wave
        const bool edge = 32 * c + 32 > a.Lq || key0 + 15 > a.P + 32 * c;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(&stat[buf][hc][16 * qt + 4 * g]);
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(&stat[buf][hc][32 + 16 * qt + 4 * g]);
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int qi = 32 * c + 16 * qt + 4 * g + rr;
            float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qt][rr], a.scale_log2e, -l4[rr]));
            if (edge && (qi >= a.Lq || key > a.P + qi)) p = 0.0f;
            pe[qt][rr] = p;
            de[qt][rr] = p * (dp[qt][rr] - d4[rr]) * a.scale;
          }
        }
        const uint4_t pf = pack_acc<DT>(pe), dsf = pack_acc<DT>(de);
#pragma unroll
        for (int dt = 0; dt < NTW; ++dt) {
          dvacc[dt] = cmfma<DT>(tr_frag<DH>(dl, dt0 + dt, r, g), pf, dvacc[dt]);
          dkacc[dt] = cmfma<DT>(tr_frag<DH>(ql, dt0 + dt, r, g), dsf, dkacc[dt]);
        }
      }
    }
    if (it + 1 < its) stash(buf ^ 1);
    __syncthreads();
  }
  uint16_t* kp = a.dk + static_cast<int64_t>(key) * a.dkv_rs + h * DR + 4 * g;
  uint16_t* vp = a.dv + static_cast<int64_t>(key) * a.dkv_rs + h * DR + 4 * g;
  if (SPLIT_D) {
    if (key >= a.Lk) return;
#pragma unroll
    for (int dt = 0; dt < NTW; ++dt) {
      bma::uint2_t ow;
      ow.x = bma::pack16<DT>(dkacc[dt][0], dkacc[dt][1]);
      ow.y = bma::pack16<DT>(dkacc[dt][2], dkacc[dt][3]);
      *reinterpret_cast<bma::uint2_t*>(kp + 16 * (dt0 + dt)) = ow;
      ow.x = bma::pack16<DT>(dvacc[dt][0], dvacc[dt][1]);
      ow.y = bma::pack16<DT>(dvacc[dt][2], dvacc[dt][3]);
      *reinterpret_cast<bma::uint2_t*>(vp + 16 * (dt0 + dt)) = ow;
    }
    return;
  }
  float* xch = reinterpret_cast<float*>(lds) + (wk * 64 + lane) * (8 * NT);
  if (half == 1) {
#pragma unroll
    for (int dt = 0; dt < NTW; ++dt) {
      *reinterpret_cast<f32x4*>(xch + 8 * dt) = dkacc[dt];
      *reinterpret_cast<f32x4*>(xch + 8 * dt + 4) = dvacc[dt];
    }
  }
  __syncthreads();
  if (half == 1 || key >= a.Lk) return;
#pragma unroll
  for (int dt = 0; dt < NTW; ++dt) {
    if (!dims_real<DR>(dt, g)) continue;
    const f32x4 k2 = *reinterpret_cast<const f32x4*>(xch + 8 * dt), v2 = *reinterpret_cast<const f32x4*>(xch + 8 * dt + 4);
    bma::uint2_t ow;
    ow.x = bma::pack16<DT>(dkacc[dt][0] + k2[0], dkacc[dt][1] + k2[1]);
    ow.y = bma::pack16<DT>(dkacc[dt][2] + k2[2], dkacc[dt][3] + k2[3]);
    *reinterpret_cast<bma::uint2_t*>(kp + 16 * dt) = ow;
    ow.x = bma::pack16<DT>(dvacc[dt][0] + v2[0], dvacc[dt][1] + v2[1]);
    ow.y = bma::pack16<DT>(dvacc[dt][2] + v2[2], dvacc[dt][3] + v2[3]);
    *reinterpret_cast<bma::uint2_t*>(vp + 16 * dt) = ow;
  }
}

int check_common(const void* q, const void* k, const void* v, int64_t Lq, int64_t Lk, int H, int Hkv, int Dh, int dtype,
                 const int64_t* strides, int n_strides) {
  if (Lq < 0 || Lk < Lq || H <= 0 || Hkv <= 0 || H % Hkv) return BMA_EINVAL;
  if (!q || !k || !v) return BMA_EINVAL;
  if (dtype != BMA_BF16 && dtype != BMA_F16) return BMA_EDTYPE;
  if ((Dh != 64 && Dh != 72 && Dh != 128 && Dh != 256) || Lk > (1 << 20) || static_cast<int64_t>(H) * ((Lk + 63) / 64) > 0x7fffffffLL) return BMA_ELIMIT;
  for (int i = 0; i < n_strides; ++i)
    if (strides[i] % 8) return BMA_EALIGN;
  if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v)) % 16) return BMA_EALIGN;
  return BMA_OK;
}

}  // namespace

constexpr int kAllVisible = 1 << 28;      // CArgs::P when the attention is not causal: past every key

// experiment knob (bma_causal_attention_set_plan): 72-wide heads with at least this many queries take the two-tiles-per-wave
// forward; 0 = never
static int64_t g_fwd_two_tiles_min_rows = 1024;

extern "C" void bma_causal_attention_set_plan(int64_t fwd_two_tiles_min_rows) {
  g_fwd_two_tiles_min_rows = fwd_two_tiles_min_rows;
}

// the head widths the kernels are built for: (width in memory) -> (DH, DR)
#define BMA_CAUSAL_WIDTHS(X, DT_)        \
  if (Dh == 128) X(DT_, 128, 128);       \
  else if (Dh == 256) X(DT_, 256, 256);  \
  else if (Dh == 72) X(DT_, 96, 72);     \
  else X(DT_, 64, 64)

extern "C" int bma_causal_attention_gqa(const void* q, int64_t q_rs, int64_t q_hs, const void* k, int64_t k_rs, int64_t k_hs,
                                        const void* v, int64_t v_rs, int64_t v_hs, int64_t Lq, int64_t Lk, int H, int Hkv, int Dh,
                                        int dtype, int causal, float scale, void* out, float* lse2, void* stream) {
  const int64_t strides[] = {q_rs, q_hs, k_rs, k_hs, v_rs, v_hs};
  const int rc = check_common(q, k, v, Lq, Lk, H, Hkv, Dh, dtype, strides, 6);
  if (rc != BMA_OK) return rc;
  if (Lq == 0) return BMA_OK;
  if (!out || !lse2 || reinterpret_cast<uintptr_t>(out) % 16 || reinterpret_cast<uintptr_t>(lse2) % 4) return out && lse2 ? BMA_EALIGN : BMA_EINVAL;
  CArgs a = {};
  a.q = static_cast<const uint16_t*>(q); a.k = static_cast<const uint16_t*>(k); a.v = static_cast<const uint16_t*>(v);
  a.out = static_cast<uint16_t*>(out); a.lse2 = lse2;
  a.q_rs = q_rs; a.q_hs = q_hs; a.k_rs = k_rs; a.k_hs = k_hs; a.v_rs = v_rs; a.v_hs = v_hs;
  a.Lq = static_cast<int>(Lq); a.Lk = static_cast<int>(Lk); a.H = H; a.rep = H / Hkv; a.P = causal ? static_cast<int>(Lk - Lq) : kAllVisible;
  a.scale = scale; a.scale_log2e = scale * 1.4426950408889634f;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(static_cast<unsigned>(H * ((Lq + 63) / 64)));
  BMA_PROF_BEGIN(BMA_K_CAUSAL_ATTN, st, 2.0 * (2.0 * static_cast<double>(Lq) * H + 2.0 * static_cast<double>(Lk) * Hkv) * Dh);
  // a 72-wide tower of thousands of tokens is throughput, not latency: two query tiles per wave (128 rows per workgroup)
  const int64_t tq2_min = g_fwd_two_tiles_min_rows;
  if (Dh == 72 && tq2_min > 0 && Lq >= tq2_min) {
    const dim3 grid2(static_cast<unsigned>(H * ((Lq + 127) / 128)));
    if (dtype == BMA_BF16) hipLaunchKernelGGL((causal_fwd_kernel<BMA_BF16, 96, 72, 2>), grid2, dim3(NTHR), 0, st, a);
    else hipLaunchKernelGGL((causal_fwd_kernel<BMA_F16, 96, 72, 2>), grid2, dim3(NTHR), 0, st, a);
  } else {
#define BMA_CAUSAL_FWD(DT_, DH_, DR_) hipLaunchKernelGGL((causal_fwd_kernel<DT_, DH_, DR_>), grid, dim3(NTHR), 0, st, a)
    if (dtype == BMA_BF16) { BMA_CAUSAL_WIDTHS(BMA_CAUSAL_FWD, BMA_BF16); }
    else { BMA_CAUSAL_WIDTHS(BMA_CAUSAL_FWD, BMA_F16); }
#undef BMA_CAUSAL_FWD
  }
  BMA_PROF_END(BMA_K_CAUSAL_ATTN, st);
  BMA_LAUNCH_CHECK();
  return BMA_OK;
}

extern "C" int bma_causal_attention(const void* q, int64_t q_rs, int64_t q_hs, const void* k, int64_t k_rs, int64_t k_hs,
                                    const void* v, int64_t v_rs, int64_t v_hs, int64_t Lq, int64_t Lk, int H, int Dh, int dtype,
                                    int causal, float scale, void* out, float* lse2, void* stream) {
  return bma_causal_attention_gqa(q, q_rs, q_hs, k, k_rs, k_hs, v, v_rs, v_hs, Lq, Lk, H, H, Dh, dtype, causal, scale, out, lse2, stream);
}

extern "C" int bma_causal_attention_bwd_gqa(const void* q, int64_t q_rs, int64_t q_hs, const void* k, int64_t k_rs, int64_t k_hs,
                                            const void* v, int64_t v_rs, int64_t v_hs, const void* out, const float* lse2,
                                            const void* d_out, int64_t Lq, int64_t Lk, int H, int Hkv, int Dh, int dtype, int causal,
                                            float scale, void* dq, void* dk, void* dv, int64_t dq_row_stride,
                                            int64_t dkv_row_stride, float* delta, void* stream) {
  const int64_t strides[] = {q_rs, q_hs, k_rs, k_hs, v_rs, v_hs};
  const int rc = check_common(q, k, v, Lq, Lk, H, Hkv, Dh, dtype, strides, 6);
  if (rc != BMA_OK) return rc;
  if (!out || !lse2 || !d_out || !dq || !dk || !dv || !delta) return BMA_EINVAL;
  if (dq_row_stride < static_cast<int64_t>(H) * Dh || dkv_row_stride < static_cast<int64_t>(Hkv) * Dh) return BMA_EINVAL;
  if (dq_row_stride % 8 || dkv_row_stride % 8) return BMA_EALIGN;
  if ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(d_out) | reinterpret_cast<uintptr_t>(dq) |
       reinterpret_cast<uintptr_t>(dk) | reinterpret_cast<uintptr_t>(dv)) % 16 ||
      (reinterpret_cast<uintptr_t>(lse2) | reinterpret_cast<uintptr_t>(delta)) % 4)
    return BMA_EALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (Lq == 0) {                                                  // no query: the keys' gradients are zero
    if (Lk > 0) {
      if (hipMemset2DAsync(dk, static_cast<size_t>(dkv_row_stride) * 2, 0, static_cast<size_t>(Hkv) * Dh * 2, static_cast<size_t>(Lk), st) != hipSuccess) return BMA_ELAUNCH;
      if (hipMemset2DAsync(dv, static_cast<size_t>(dkv_row_stride) * 2, 0, static_cast<size_t>(Hkv) * Dh * 2, static_cast<size_t>(Lk), st) != hipSuccess) return BMA_ELAUNCH;
    }
    return BMA_OK;
  }
  CArgs a = {};
  a.q = static_cast<const uint16_t*>(q); a.k = static_cast<const uint16_t*>(k); a.v = static_cast<const uint16_t*>(v);
  a.o = static_cast<const uint16_t*>(out); a.d_o = static_cast<const uint16_t*>(d_out);
  a.dq = static_cast<uint16_t*>(dq); a.dk = static_cast<uint16_t*>(dk); a.dv = static_cast<uint16_t*>(dv);
  a.lse2 = const_cast<float*>(lse2); a.delta = delta;
  a.q_rs = q_rs; a.q_hs = q_hs; a.k_rs = k_rs; a.k_hs = k_hs; a.v_rs = v_rs; a.v_hs = v_hs; a.d_rs = dq_row_stride; a.dkv_rs = dkv_row_stride;
  a.Lq = static_cast<int>(Lq); a.Lk = static_cast<int>(Lk); a.H = H; a.rep = H / Hkv; a.P = causal ? static_cast<int>(Lk - Lq) : kAllVisible;
  a.scale = scale; a.scale_log2e = scale * 1.4426950408889634f;
  const dim3 gq(static_cast<unsigned>(H * ((Lq + 63) / 64))), gk(static_cast<unsigned>(Hkv * ((Lk + 63) / 64)));
  BMA_PROF_BEGIN(BMA_K_CAUSAL_ATTN, st, 2.0 * (4.0 * static_cast<double>(Lq) * H + 4.0 * static_cast<double>(Lk) * Hkv) * Dh);
#define BMA_CAUSAL_BWD(DT_, DH_, DR_)                                                                                          \
  do {                                                                                                                         \
    hipLaunchKernelGGL((causal_dq_kernel<DT_, DH_, DR_>), gq, dim3(NTHR), 0, st, a); /* writes delta for the next launch */   \
    hipLaunchKernelGGL((causal_dkv_kernel<DT_, DH_, DR_>), gk, dim3(NTHR), 0, st, a);                                          \
  } while (0)
  if (dtype == BMA_BF16) { BMA_CAUSAL_WIDTHS(BMA_CAUSAL_BWD, BMA_BF16); }
  else { BMA_CAUSAL_WIDTHS(BMA_CAUSAL_BWD, BMA_F16); }
#undef BMA_CAUSAL_BWD
  BMA_PROF_END(BMA_K_CAUSAL_ATTN, st);
  BMA_LAUNCH_CHECK();
  return BMA_OK;
}

extern "C" int bma_causal_attention_bwd(const void* q, int64_t q_rs, int64_t q_hs, const void* k, int64_t k_rs, int64_t k_hs,
                                        const void* v, int64_t v_rs, int64_t v_hs, const void* out, const float* lse2,
                                        const void* d_out, int64_t Lq, int64_t Lk, int H, int Dh, int dtype, int causal,
                                        float scale, void* dq, void* dk, void* dv, int64_t d_row_stride, float* delta,
                                        void* stream) {
  return bma_causal_attention_bwd_gqa(q, q_rs, q_hs, k, k_rs, k_hs, v, v_rs, v_hs, out, lse2, d_out, Lq, Lk, H, H, Dh, dtype, causal, scale,
                                      dq, dk, dv, d_row_stride, d_row_stride, delta, stream);
}
