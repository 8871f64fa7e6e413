// bma_prefix_attention: every row of a scoring forward against the keys/values of the SHARED prefix.
//
// Joint scoring (reference bimodal_attack.py:1150-1163, :1278-1310): the 599 tokens in front of the suffix --
// prompt head, 576 image tokens, prompt -- are the same in every candidate, so their keys/values exist once
// (hf_adapter.build_prefix_recording) and all N computed rows (17k at search_width 512) attend to them without
// a mask.  The bytes are small (q read once, o written once, the prefix K/V of a head re-read from L2); the
// work is arithmetic, 4*N*P*Dh*H = 168 GFLOP per layer, so this kernel is built like a flash-attention
// forward for the matrix cores, not like the ragged kernel (ragged_attention.hip), whose partial result it
// feeds: out = (o1 [N][H][Dh], lse1 [H][N]) merged there in the epilogue.
//
//   workgroup   4 waves = 128 consecutive rows of one head; a wave owns 32 rows as two 16-row tiles, so every
//               K fragment (ds_read_b128) and V^T fragment (ds_read_b64_tr_b16) read from LDS feeds TWO MFMAs
//   S^T = K Q^T v_mfma_f32_16x16x32: A = 16 keys x 32 dims from an LDS key row, B = Q fragment (registers,
//               loaded once); a query is a lane column, its running max / sum live in the lane
//   O^T += V^T P^T  the S^T accumulator, exponentiated and packed, IS the B operand
//   LDS reads   left to the compiler's schedule (two fragment reads ahead of every pair of MFMAs): issued in groups
//               with hand-counted waits as in the long-block kernel (bma_lds.h: all K fragments of a chunk at once,
//               V^T in groups of four tiles) the launch took 303 us instead of 255 -- with two workgroups per CU
//               and no DMA in flight the fine interleave is the better one
//   staging     32 keys per chunk, two LDS image pairs: chunk c+1 travels L2 -> registers while chunk c is
//               multiplied and is stored to the other pair afterwards; one barrier per chunk
//   placement   blocks b and b+8 share an XCD (round-robin dispatch): block b works on head 4*(b%8) + (b/8)%4
//               when H is a multiple of 8... in general head = (b % 8) * (H/8) + (b/8) % (H/8), so one XCD's
//               L2 holds the prefix of H/8 heads (1.2 MB at 32 heads) instead of all of them (9.8 MB)
//
// Algorithmic work per launch: 4*N*P*Dh*H flops; bytes 2*N*H*Dh*es (+ the prefix once).
#include <type_traits>

#include "bma_common.h"
#include "bma_lds.h"
#include "bma_profile.h"

namespace {

using bma::uint4_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short short4_t __attribute__((ext_vector_type(4)));

struct PArgs {
  const uint16_t *q, *pk, *pv;
  uint16_t* out;
  float* lse;
  int64_t q_rs, q_hs, pk_rs, pk_hs, pv_rs, pv_hs;
  int N, H, Hk, P, row_blocks;
  float scale_log2e;
};

template <int DT>
__device__ __forceinline__ f32x4 pmfma(const uint4_t& a, const uint4_t& b, const f32x4& c) {
  if (DT == BMA_BF16)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

#ifndef BMA_PA_WAVES
#define BMA_PA_WAVES 4
#endif
#ifndef BMA_PA_TILES
#define BMA_PA_TILES 2
#endif
#ifndef BMA_PA_OCC
#define BMA_PA_OCC 2
#endif
// LDS fragment reads issued this many fragments AHEAD of the MFMAs that consume them.  0 = the shipped form: a pair of reads,
// a full wait, two MFMAs -- its ISA exposes one LDS latency per fragment pair, and the SQ counters say waves spend 41 % of
// their cycles in instruction waits with the MFMA pipe 31 % busy (profiles/r5_prefix_attn_pmc.txt).  Read-ahead was the
// obvious cure and is NOT one (round 5, same box, us): 0 -> 286-292, 2 -> 305-310, 4 -> 307-313.  The ring costs 8-12 VGPRs
// (168 -> 176 / 180), which takes the third wave per SIMD away (512 / 3 = 170), and three waves hide more latency than one
// wave's own look-ahead does -- round 4's hand-counted variant (303 against 255) lost to the same arithmetic.
#ifndef BMA_PA_PIPE
#define BMA_PA_PIPE 0
#endif
constexpr int kWaves = BMA_PA_WAVES;        // waves per workgroup
constexpr int kTiles = BMA_PA_TILES;        // 16-row query tiles per wave
constexpr int kRowsPerWg = 16 * kTiles * kWaves;

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float vmax(float a, float b) {      // one v_max_f32, no canonicalising pre-ops
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// all-reduce over the four 16-lane rows of a wave (a query's keys are spread over lanes l, l+16, l+32, l+48) with
// gfx950's row swaps: v_permlane16_swap exchanges the odd 16-lane rows of one register with the even rows of the
// other, so max(x', y') of a self-swap is the xor-16 reduction; v_permlane32_swap does the same for 32-lane halves.
// Two VALU instructions per level instead of a ds_bpermute round trip through the LDS crossbar.
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

template <int DT, int DH>
__global__ __launch_bounds__(64 * kWaves, BMA_PA_OCC) void prefix_attn_kernel(const PArgs a) {
  constexpr int KS = DH / 32;      // k-steps of the QK product
  constexpr int NT = DH / 16;      // 16-dim tiles of the output
  constexpr int PITCH = DH + 16;   // elements per LDS row (row + 32 B: conflict-free transposing reads)
  constexpr int PIECES = DH / 8;   // 16-byte pieces per row
  constexpr int NTHR = 64 * kWaves;
  constexpr int IMG = 32 * PITCH;
  constexpr int ITEMS = (32 * PIECES + NTHR - 1) / NTHR;      // pieces per thread per image (2 at Dh = 128, 4 waves)
  __shared__ __attribute__((aligned(16))) uint16_t lds[2 * 2 * IMG];   // two pairs of (K image, V image)
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  // XCD-aware (head, row block) of this workgroup
  int h, rb;
  {
    const int b = blockIdx.x;
    if (a.H % 8 == 0) {
      const int hx = a.H / 8;
      const int j = b >> 3;
      h = (b & 7) * hx + j % hx;
      rb = j / hx;
    } else {
      h = b % a.H;
      rb = b / a.H;
    }
  }
  if (rb >= a.row_blocks) return;
  const int hk = h / (a.H / a.Hk);
  const int row0 = rb * kRowsPerWg + 16 * kTiles * w;     // first row of this wave
  const float NEG = -__builtin_inff();

  // Q tiles as B operands: lane (query r, dims 8g.. of k-step ks)
  uint4_t qf[kTiles][KS];
#pragma unroll
  for (int t = 0; t < kTiles; ++t) {
    int row = row0 + 16 * t + r;
    row = row < a.N ? row : a.N - 1;
    const uint16_t* qp = a.q + static_cast<int64_t>(row) * a.q_rs + static_cast<int64_t>(h) * a.q_hs + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[t][ks] = *reinterpret_cast<const uint4_t*>(qp + 32 * ks);
  }

  f32x4 oacc[kTiles][NT];
  float mrun[kTiles], lsum[kTiles];
#pragma unroll
  for (int t = 0; t < kTiles; ++t) {
    mrun[t] = NEG;
    lsum[t] = 0.0f;
#pragma unroll
    for (int dt = 0; dt < NT; ++dt) oacc[t][dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  }

  const uint16_t* kbase = a.pk + static_cast<int64_t>(hk) * a.pk_hs;
  const uint16_t* vbase = a.pv + static_cast<int64_t>(hk) * a.pv_hs;
  // per-lane element offsets of its pieces inside a chunk, computed once; a chunk's base is wave-uniform
  // (scalar arithmetic), so the loop body carries no 64-bit vector address arithmetic.  Only the last chunk
  // clamps rows past the prefix.
  int koff[ITEMS], voff[ITEMS];
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int idx = tid + it * NTHR;
    const int key = idx / PIECES, piece = idx % PIECES;
    koff[it] = key * static_cast<int>(a.pk_rs) + 8 * piece;
    voff[it] = key * static_cast<int>(a.pv_rs) + 8 * piece;
  }
  uint4_t kreg[ITEMS], vreg[ITEMS];
  auto fetch = [&](int c) {
    const uint16_t* kc = kbase + static_cast<int64_t>(32 * c) * a.pk_rs;
    const uint16_t* vc = vbase + static_cast<int64_t>(32 * c) * a.pv_rs;
    if (32 * c + 32 <= a.P) {
#pragma unroll
      for (int it = 0; it < ITEMS; ++it) {
        if (32 * PIECES % NTHR && tid + it * NTHR >= 32 * PIECES) continue;
        kreg[it] = *reinterpret_cast<const uint4_t*>(kc + koff[it]);
        vreg[it] = *reinterpret_cast<const uint4_t*>(vc + voff[it]);
      }
    } else {
#pragma unroll
      for (int it = 0; it < ITEMS; ++it) {
        const int idx = tid + it * NTHR;
        if (32 * PIECES % NTHR && idx >= 32 * PIECES) continue;
        int key = idx / PIECES;
        const int piece = idx % PIECES;
        key = 32 * c + key < a.P ? key : a.P - 1 - 32 * c;
        kreg[it] = *reinterpret_cast<const uint4_t*>(kc + static_cast<int64_t>(key) * a.pk_rs + 8 * piece);
        vreg[it] = *reinterpret_cast<const uint4_t*>(vc + static_cast<int64_t>(key) * a.pv_rs + 8 * piece);
      }
    }
  };
  int loff[ITEMS];
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int idx = tid + it * NTHR;
    loff[it] = (idx / PIECES) * PITCH + 8 * (idx % PIECES);
  }
  auto stash = [&](uint16_t* img) {
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      if (32 * PIECES % NTHR && tid + it * NTHR >= 32 * PIECES) continue;
      *reinterpret_cast<uint4_t*>(img + loff[it]) = kreg[it];
      *reinterpret_cast<uint4_t*>(img + IMG + loff[it]) = vreg[it];
    }
  };

  const int chunks = (a.P + 31) >> 5;
  fetch(0);
  stash(lds);
  __syncthreads();
  for (int c = 0; c < chunks; ++c) {
    const uint16_t* kl = lds + 2 * IMG * (c & 1);
    const uint16_t* vl = kl + IMG;
    if (c + 1 < chunks) fetch(c + 1);                       // in flight while chunk c is multiplied

    // ---- S^T = K Q^T for both query tiles: every K fragment feeds two MFMAs ------------------------
    f32x4 s[kTiles][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int t = 0; t < kTiles; ++t) s[t][kt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (BMA_PA_PIPE > 0) {
      // all 2 * KS key fragments of the chunk in flight before the first product (they are dead again before the V^T
      // fragments of the second product are read, so the registers are shared)
      uint4_t kf[2][KS];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          kf[kt][ks] = *reinterpret_cast<const uint4_t*>(kl + (16 * kt + r) * PITCH + 8 * g + 32 * ks);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int t = 0; t < kTiles; ++t) s[t][kt] = pmfma<DT>(kf[kt][ks], qf[t][ks], s[t][kt]);
    } else {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const uint16_t* kr = kl + (16 * kt + r) * PITCH + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const uint4_t kf = *reinterpret_cast<const uint4_t*>(kr + 32 * ks);
#pragma unroll
          for (int t = 0; t < kTiles; ++t) s[t][kt] = pmfma<DT>(kf, qf[t][ks], s[t][kt]);
        }
      }
    }
    // ---- online softmax per tile; the probabilities become the B operand of the second product ----
    // The maximum is taken on the raw scores (the scale is positive) and the scale folded into the exponent's
    // argument: one fma per score.  Only the last chunk has keys to mask: it runs its own copy of the code.
    uint4_t pf[kTiles];
    float alpha[kTiles];
    auto softmax = [&](auto masked) {
#pragma unroll
      for (int t = 0; t < kTiles; ++t) {
        float e[2][4];
        float cmax = NEG;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            float v = s[t][kt][rr];
            if (decltype(masked)::value && 32 * c + 16 * kt + 4 * g + rr >= a.P) v = NEG;
            e[kt][rr] = v;
            cmax = vmax(cmax, v);
          }
        cmax = rows_max(cmax);
        const float mnew = vmax(mrun[t], cmax);               // raw units; finite from the first chunk on (P >= 1)
        alpha[t] = __builtin_amdgcn_exp2f((mrun[t] - mnew) * a.scale_log2e);
        const float mneg = -mnew * a.scale_log2e;
        float rs = 0.0f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            e[kt][rr] = __builtin_amdgcn_exp2f(__builtin_fmaf(e[kt][rr], a.scale_log2e, mneg));
            rs += e[kt][rr];
          }
        lsum[t] = lsum[t] * alpha[t] + rs;
        mrun[t] = mnew;
        pf[t].x = bma::pack16<DT>(e[0][0], e[0][1]);
        pf[t].y = bma::pack16<DT>(e[0][2], e[0][3]);
        pf[t].z = bma::pack16<DT>(e[1][0], e[1][1]);
        pf[t].w = bma::pack16<DT>(e[1][2], e[1][3]);
      }
    };
    if (32 * c + 32 > a.P) softmax(std::true_type{});
    else softmax(std::false_type{});
    // a running maximum that did not move (the usual case after the first chunks) leaves alpha = 1 in every
    // lane of the wave: the 4*NT*kTiles rescaling multiplies are skipped then (wave-uniform branch)
    bool moved = false;
#pragma unroll
    for (int t = 0; t < kTiles; ++t) moved = moved || (alpha[t] != 1.0f);
    if (__builtin_amdgcn_ballot_w64(moved) != 0) {
#pragma unroll
      for (int t = 0; t < kTiles; ++t)
#pragma unroll
        for (int dt = 0; dt < NT; ++dt) {
          oacc[t][dt][0] *= alpha[t]; oacc[t][dt][1] *= alpha[t]; oacc[t][dt][2] *= alpha[t]; oacc[t][dt][3] *= alpha[t];
        }
    }
    // ---- O^T += V^T P^T: every V^T fragment feeds kTiles MFMAs, accumulating in place --------------------
    const int q4 = r >> 2, p4 = r & 3;
    const uint16_t* rd = vl + (4 * g + q4) * PITCH + 4 * p4;
    auto vt_frag = [&](int dt) {
      const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(rd + 16 * dt));
      const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) short4_t*)(rd + 16 * PITCH + 16 * dt));
      const bma::uint2_t l2 = __builtin_bit_cast(bma::uint2_t, lo), h2 = __builtin_bit_cast(bma::uint2_t, hi);
      uint4_t vf;
      vf.x = l2.x; vf.y = l2.y; vf.z = h2.x; vf.w = h2.y;
      return vf;
    };
    if (BMA_PA_PIPE > 0) {
      constexpr int AHEAD = BMA_PA_PIPE < NT ? BMA_PA_PIPE : NT;
      uint4_t ring[AHEAD > 0 ? AHEAD : 1];
#pragma unroll
      for (int i = 0; i < AHEAD; ++i) ring[i] = vt_frag(i);
#pragma unroll
      for (int dt = 0; dt < NT; ++dt) {
        const uint4_t vf = ring[dt % AHEAD];
        if (dt + AHEAD < NT) ring[dt % AHEAD] = vt_frag(dt + AHEAD);
#pragma unroll
        for (int t = 0; t < kTiles; ++t) oacc[t][dt] = pmfma<DT>(vf, pf[t], oacc[t][dt]);
      }
    } else {
#pragma unroll
      for (int dt = 0; dt < NT; ++dt) {
        const uint4_t vf = vt_frag(dt);
#pragma unroll
        for (int t = 0; t < kTiles; ++t) oacc[t][dt] = pmfma<DT>(vf, pf[t], oacc[t][dt]);
      }
    }
    // every wave left chunk c-1 (the other image pair) behind at the previous barrier: safe to overwrite it
    if (c + 1 < chunks) stash(lds + 2 * IMG * ((c + 1) & 1));
    __syncthreads();
  }

  // ---- epilogue: normalise, store o1 and the natural-log LSE -------------------------------------------
#pragma unroll
  for (int t = 0; t < kTiles; ++t) {
    const float l = rows_sum(lsum[t]);
    const int row = row0 + 16 * t + r;
    if (row >= a.N) continue;
    const float inv = 1.0f / l;
    if (g == 0) a.lse[static_cast<int64_t>(h) * a.N + row] = (mrun[t] * a.scale_log2e + __builtin_amdgcn_logf(l)) * 0.6931471805599453f;
    uint16_t* op = a.out + (static_cast<int64_t>(row) * a.H + h) * DH + 4 * g;
#pragma unroll
    for (int dt = 0; dt < NT; ++dt) {
      bma::uint2_t ow;
      ow.x = bma::pack16<DT>(oacc[t][dt][0] * inv, oacc[t][dt][1] * inv);
      ow.y = bma::pack16<DT>(oacc[t][dt][2] * inv, oacc[t][dt][3] * inv);
      *reinterpret_cast<bma::uint2_t*>(op + 16 * dt) = ow;
    }
  }
}

// =====================================================================================================================
// prefix_attn32_kernel (round 6): the same product for 128-wide heads, rebuilt around what the SQ
// counters of the kernel above said (profiles/r5_prefix_attn_pmc.txt: MFMA pipe 31 % busy, 41 % of the wave cycles in
// instruction waits, ~190 vector-class instructions per 32 keys and wave).  230 -> 197 us at 17152 x 599 x 32 x 128 on one
// box, 0.29 -> 0.34 of the MFMA peak (profiles/r6_prefix_attn32.txt; the steps that got there and the ones that did not).
//
//   products    v_mfma_f32_32x32x16: S^T[32 keys][32 queries] = K Q^T in 8 instructions per 32-key tile, O^T[128][32] +=
//               V^T P^T in 8 -- half the MFMA count of the 16x16x32 form for the same LDS bytes per flop (a wave owns 32
//               rows either way), and a query's 32 scores sit in TWO lanes (l, l ^ 32): the row maximum is 8 v_max3 + one
//               v_permlane32_swap instead of 7 v_max + two swap levels per 16-row tile
//   P operand   the exponentiated S^T accumulator, packed pairwise, IS the B operand of the second product: registers
//               8s..8s+7 are k-step s, whose slot 8h + j is key 16s + 8(j>>2) + 4h + (j&3) (cdna_hip_programming.md
//               section 3, "An accumulator tile as the next MFMA's operand"); the V^T fragments are read in that same key
//               order (two ds_read_b64_tr_b16 per fragment: keys +4h..+4h+3 and +8+4h..), so nothing moves between lanes
//   pipeline    one step = one tile t: tile t+1's scores are formed on the matrix pipe WHILE tile t's maximum, exponentials
//               and packing run on the vector ALU (independent: the instruction stream alternates one MFMA with ~7 vector
//               instructions), then O^T += V^T P^T of tile t; the two score accumulators swap names every step
//   staging     K and V rings of four 32-key tiles each (64 KB: two workgroups per CU); a tile travels L2 -> LDS by LDS-DMA
//               (buffer_load_dwordx4 ... lds, 1 KiB = 4 key rows per wave instruction; rows past the prefix read as zeros
//               through the descriptor's bounds check, no clamping) two to three steps ahead of its use: no staging
//               registers, no ds_write; ONE barrier per two steps
//   LDS images  16-byte pieces permuted on the DMA's SOURCE side: K piece c of row r at position c ^ (r & 15) (ds_read_b128
//               of 16 rows x one piece per lane group: 16 distinct positions), V piece c at c ^ ((r & 3) << 2) (a
//               transposing read takes 4 rows x 64 B per half wave: four distinct 64-byte quarters); both conflict-free
//               by the bank rule of MI355X_MICROARCH.md (checked with a script before the first run; SQ_LDS_BANK_CONFLICT 0)
//   LDS reads   inline asm with hand-counted lgkmcnt (bma_lds.h: the compiler cannot tell a read from the DMA's writes
//               and would drain the ring in front of each)
//   maximum     the running maximum of a query moves only when a tile's maximum exceeds it by more than 2^BMA_PA32_DEFER
//               (after scaling): probabilities are then at most 2^4 instead of 1 -- the same relative rounding in the 16-bit
//               type -- and the 64 accumulator registers are rescaled only in the steps where some lane's maximum moved
//               (guide T13; measured against float64: 5.3e-3 -> 6.0e-3 of the row scale in bf16, bound 1.2e-2)
//   epilogue    v_permlane32_swap pairs the two half-waves' 8-byte column groups into 16-byte stores (guide T21)
//
// Per launch the same 4*N*P*Dh*H flops; K/V pulled through L2 -> LDS once per 32*NW rows.
typedef float f32x16 __attribute__((ext_vector_type(16)));
using bma::u32x2;
using bma::u32x4;
using bma::tr_read;
using bma::row_read;
using bma::wait_rows;

template <int DT>
__device__ __forceinline__ f32x16 pmfma32(const u32x4& a, const u32x4& b, const f32x16& c) {
  if (DT == BMA_BF16)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait4x2(u32x2 (&f)[4]) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "n"(N));
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// both halves of a lane pair (l, l ^ 32) get max / sum of the pair
__device__ __forceinline__ float pair_max(float x) {
  u32x2 b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return vmax(__uint_as_float(b.x), __uint_as_float(b.y));
}
__device__ __forceinline__ float pair_sum(float x) {
  u32x2 b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(b.x) + __uint_as_float(b.y);
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned char* lds_dst, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
}

#ifndef BMA_PA32_DEFER
#define BMA_PA32_DEFER 4
#endif
// measurement builds only (wrong results): 1 = no maximum / exponentials (the scores are packed as they are), 2 = no LDS-DMA
// inside the loop, 4 = no waits / barriers inside the loop
#ifndef BMA_PA32_ABL
#define BMA_PA32_ABL 0
#endif
template <int DT, int NW>
__global__ __launch_bounds__(64 * NW, 8 / NW) void prefix_attn32_kernel(const PArgs a) {
  constexpr int DH = 128;
  constexpr int KS = DH / 16;          // k-steps of the QK product
  constexpr int OT = DH / 32;          // 32-dim tiles of the output
  constexpr int ROWB = 2 * DH;         // bytes of an LDS row
  constexpr int TILEB = 32 * ROWB;     // one 32-key tile
  constexpr int PPT = 8 / NW;          // 1-KiB pieces per wave and tile
  __shared__ __attribute__((aligned(1024))) unsigned char lds[8 * TILEB];      // K tiles t & 3 | V tiles t & 3
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qc = lane & 31, hi = lane >> 5;
  int h, rb;
  {
    const int b = blockIdx.x;
    if (a.H % 8 == 0) {
      const int hx = a.H / 8;
      const int j = b >> 3;
      h = (b & 7) * hx + j % hx;
      rb = j / hx;
    } else {
      h = b % a.H;
      rb = b / a.H;
    }
  }
  if (rb >= a.row_blocks) return;
  const int hk = h / (a.H / a.Hk);
  const int row0 = rb * (32 * NW) + 32 * wave;            // first row of this wave
  const float NEG = -__builtin_inff();

  // ---- the key/value stream: descriptors over this head's prefix rows, per-lane offsets computed once ----------------
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(a.pk + static_cast<int64_t>(hk) * a.pk_hs), 0,
      static_cast<int>((static_cast<int64_t>(a.P - 1) * a.pk_rs + DH) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(a.pv + static_cast<int64_t>(hk) * a.pv_hs), 0,
      static_cast<int>((static_cast<int64_t>(a.P - 1) * a.pv_rs + DH) * 2), 0x00020000);
  const int krb = static_cast<int>(a.pk_rs) * 2, vrb = static_cast<int>(a.pv_rs) * 2;      // bytes per key row
  const int lrow = 4 * wave + (lane >> 4), pos = lane & 15;     // piece j of this wave: rows 4*NW*j + lrow of the tile
  const int kvo = lrow * krb + 16 * (pos ^ (lrow & 15));
  const int vvo = lrow * vrb + 16 * (pos ^ ((lrow & 3) << 2));
  const int tiles = (a.P + 31) >> 5;
  auto issue_k = [&](int t) {                                   // key tile t into K slot t & 3 (wave-uniform destination)
    if (t >= tiles) return;
#pragma unroll
    for (int j = 0; j < PPT; ++j)
      dma16(rk, lds + (t & 3) * TILEB + (4 * wave + 4 * NW * j) * ROWB, kvo, (32 * t + 4 * NW * j) * krb);
  };
  auto issue_v = [&](int t) {
    if (t >= tiles) return;
#pragma unroll
    for (int j = 0; j < PPT; ++j)
      dma16(rv, lds + (4 + (t & 3)) * TILEB + (4 * wave + 4 * NW * j) * ROWB, vvo, (32 * t + 4 * NW * j) * vrb);
  };
  issue_k(0); issue_v(0); issue_k(1); issue_k(2); issue_v(1); issue_k(3); issue_v(2);

  // Q rows as B operands: lane (query qc, dims 16ks + 8hi ..)
  u32x4 qf[KS];
  {
    int row = row0 + qc;
    row = row < a.N ? row : a.N - 1;
    const uint16_t* qp = a.q + static_cast<int64_t>(row) * a.q_rs + static_cast<int64_t>(h) * a.q_hs + 8 * hi;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const u32x4*>(qp + 16 * ks);
  }

  f32x16 oacc[OT];
#pragma unroll
  for (int ot = 0; ot < OT; ++ot)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[ot][i] = 0.0f;
  float mrun = NEG, lsum = 0.0f;

  // fragment addresses (bytes) in slot 0; k-step offsets are immediates of the reads, the slot advances by one tile per step
  const uint32_t lbase = bma::lds_addr(lds);
  uint32_t ka[KS], va[OT];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) ka[ks] = lbase + qc * ROWB + 16 * ((2 * ks + hi) ^ (qc & 15));
  {
    const int gp = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
      va[ot] = lbase + 4 * TILEB + (4 * hi + q4) * ROWB + 16 * ((4 * ot + 2 * (gp & 1) + (p4 >> 1)) ^ (q4 << 2)) + 8 * (p4 & 1);
  }
  int kslot = 0, vslot = 0;                                      // wave-uniform: slot the addresses point at
  auto next_k = [&]() {
    const int d = kslot == 3 ? -3 * TILEB : TILEB;
    kslot = (kslot + 1) & 3;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) ka[ks] += d;
  };
  auto next_v = [&]() {
    const int d = vslot == 3 ? -3 * TILEB : TILEB;
    vslot = (vslot + 1) & 3;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) va[ot] += d;
  };
  auto kread = [&](u32x4(&buf)[4], int ks0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) buf[j] = row_read<0>(ka[ks0 + j]);
  };

  // One step = one 32-key tile t: its softmax (scores sc, formed in the step before) runs on the vector ALU while the matrix
  // pipe forms tile t+1's scores sn = K Q^T -- the two are independent, so the wave's MFMAs execute in the shadow of its own
  // exponentials instead of waiting in line behind them (the first form of this kernel ran QK, softmax and PV of a 64-key
  // chunk one after the other: profiles/r6_prefix_attn32_pmc_first_form.txt, 38 % of the wave cycles stalled at issue, 24 %
  // in waits) -- then O^T += V^T P^T of tile t.
  auto step = [&](auto next_c, int t, f32x16& sc, f32x16& sn) {
    constexpr bool NEXT = decltype(next_c)::value;
    u32x4 kf[2][4];
    u32x2 vf[4][4];
    if constexpr (NEXT) {
      kread(kf[0], 0);
      kread(kf[1], 4);
#pragma unroll
      for (int i = 0; i < 16; ++i) sn[i] = 0.0f;
    }
    // ---- maximum of tile t's scores: a query's 32 scores are in this lane and lane ^ 32 ---------------------------------
    if constexpr (!NEXT) {                                     // only the last tile can have keys to mask
      if (32 * t + 32 > a.P) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (32 * t + 8 * (i >> 2) + 4 * hi + (i & 3) >= a.P) sc[i] = NEG;
      }
    }
    float cmax = vmax3(sc[0], sc[1], sc[2]);
    if (!(BMA_PA32_ABL & 1)) {
#pragma unroll
      for (int i = 3; i < 15; i += 2) cmax = vmax3(cmax, sc[i], sc[i + 1]);
      cmax = vmax(cmax, sc[15]);
    }
    if constexpr (NEXT) {
      wait_rows<4, 4>(kf[0]);
#pragma unroll
      for (int j = 0; j < 4; ++j) sn = pmfma32<DT>(kf[0][j], qf[j], sn);
    }
    if (!(BMA_PA32_ABL & 1)) cmax = pair_max(cmax);
    // the running maximum moves only when a score exceeds it by more than BMA_PA32_DEFER (log2 units after scaling): the
    // probabilities are then at most 2^DEFER instead of 1 -- the same relative rounding in the 16-bit type -- and the 64
    // accumulator registers are rescaled only in those steps (0: the textbook update)
    const float mtop = vmax(mrun, cmax);
    const float mnew = BMA_PA32_DEFER == 0 ? mtop : ((cmax - mrun) * a.scale_log2e > static_cast<float>(BMA_PA32_DEFER) ? mtop : mrun);
    const float alpha = __builtin_amdgcn_exp2f((mrun - mnew) * a.scale_log2e);
    const float mneg = -mnew * a.scale_log2e;
    float rs = 0.0f;
    u32x4 pf[2];
    auto exps = [&](int s2) {                                  // eight scores -> one k-step of P
#pragma unroll
      for (int i = 8 * s2; i < 8 * s2 + 8; ++i) {
        if (!(BMA_PA32_ABL & 1)) sc[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[i], a.scale_log2e, mneg));
        if (!(BMA_PA32_ABL & 1) || i == 8 * s2) rs += sc[i];
      }
      pf[s2].x = bma::pack16<DT>(sc[8 * s2 + 0], sc[8 * s2 + 1]);
      pf[s2].y = bma::pack16<DT>(sc[8 * s2 + 2], sc[8 * s2 + 3]);
      pf[s2].z = bma::pack16<DT>(sc[8 * s2 + 4], sc[8 * s2 + 5]);
      pf[s2].w = bma::pack16<DT>(sc[8 * s2 + 6], sc[8 * s2 + 7]);
    };
    exps(0);
    if constexpr (NEXT) {
      wait_rows<0, 4>(kf[1]);
#pragma unroll
      for (int j = 0; j < 4; ++j) sn = pmfma32<DT>(kf[1][j], qf[4 + j], sn);
    }
    exps(1);
    // V^T fragments of tile t: group ot = the two k-steps of output tile ot (four transposing reads)
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
      vf[ot][0] = tr_read<0>(va[ot]);
      vf[ot][1] = tr_read<8 * ROWB>(va[ot]);
      vf[ot][2] = tr_read<16 * ROWB>(va[ot]);
      vf[ot][3] = tr_read<24 * ROWB>(va[ot]);
    }
    lsum = lsum * alpha + rs;
    mrun = mnew;
    // (the probabilities are "used" here so that the compiler cannot sink the exponentials below the branch, behind the
    // products they are meant to run beside)
    asm volatile("" ::"v"(pf[0].x), "v"(pf[0].y), "v"(pf[0].z), "v"(pf[0].w), "v"(pf[1].x), "v"(pf[1].y), "v"(pf[1].z), "v"(pf[1].w), "v"(lsum));
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {     // a maximum moved somewhere in the wave
#pragma unroll
      for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[ot][i] *= alpha;
    }
    // ---- O^T += V^T P^T ---------------------------------------------------------------------------------------------------
    auto pv = [&](auto oo) {
      constexpr int ot = decltype(oo)::value;
      wait4x2<4 * (OT - 1 - ot)>(vf[ot]);
      u32x4 v0, v1;
      v0.x = vf[ot][0].x; v0.y = vf[ot][0].y; v0.z = vf[ot][1].x; v0.w = vf[ot][1].y;
      v1.x = vf[ot][2].x; v1.y = vf[ot][2].y; v1.z = vf[ot][3].x; v1.w = vf[ot][3].y;
      oacc[ot] = pmfma32<DT>(v0, pf[0], oacc[ot]);
      oacc[ot] = pmfma32<DT>(v1, pf[1], oacc[ot]);
    };
    pv(std::integral_constant<int, 0>{}); pv(std::integral_constant<int, 1>{}); pv(std::integral_constant<int, 2>{});
    pv(std::integral_constant<int, 3>{});
    if constexpr (NEXT) next_k();
    next_v();
  };

  // The Q rows are waited for HERE, where the compiler can see it: its own wait-count bookkeeping does not know the
  // waits of the loop (inline asm), and a register load still pending at a loop header in its books costs an
  // s_waitcnt vmcnt(0) -- the ring drained -- in front of every step (ragged_attn_long_kernel learnt this).
  wait_vm<0>();
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) asm volatile("" ::"v"(qf[ks].x), "v"(qf[ks].y), "v"(qf[ks].z), "v"(qf[ks].w));
  __builtin_amdgcn_s_barrier();                            // K tiles 0-3 and V tiles 0-2 landed
  f32x16 sa, sb;
  {                                                        // tile 0's scores, nothing to overlap them with
    u32x4 kf[2][4];
#pragma unroll
    for (int i = 0; i < 16; ++i) sa[i] = 0.0f;
    kread(kf[0], 0);
    kread(kf[1], 4);
    wait_rows<4, 4>(kf[0]);
#pragma unroll
    for (int j = 0; j < 4; ++j) sa = pmfma32<DT>(kf[0][j], qf[j], sa);
    wait_rows<0, 4>(kf[1]);
#pragma unroll
    for (int j = 0; j < 4; ++j) sa = pmfma32<DT>(kf[1][j], qf[4 + j], sa);
    next_k();
  }
  // Barriers sit in front of the odd steps.  The one in front of step t (t odd): everybody is through step t-1, so K tiles
  // <= t and V tiles <= t-1 are multiplied and their slots free: K tiles t+3, t+4 and V tiles t+2, t+3 can be requested from
  // there on; steps t and t+1 read K tiles t+1, t+2 and V tiles t, t+1 -- requested behind the barrier before, landed at this
  // one (every wave drains its own requests in front of a barrier).
  auto sync = [&](int t) {                                 // in front of odd step t
    if (!(BMA_PA32_ABL & 4)) {
      wait_vm<0>();
      __builtin_amdgcn_s_barrier();
    }
    if (!(BMA_PA32_ABL & 2)) { issue_k(t + 3); issue_v(t + 2); }
  };
  // the other half of a barrier's requests, made a step later (they are first read two steps on, behind the NEXT barrier,
  // whose wait covers them): four DMA instructions per wave at a time instead of eight
  auto late = [&](int t) {                                 // in front of even step t >= 2
    if (!(BMA_PA32_ABL & 2)) { issue_k(t + 3); issue_v(t + 2); }
  };
  int t = 0;
  for (; t + 2 < tiles; t += 2) {                          // both steps have a tile behind them; the scores end up in sa again
    if (t) late(t);
    step(std::true_type{}, t, sa, sb);
    sync(t + 1);
    step(std::true_type{}, t + 1, sb, sa);
  }
  if (t + 2 == tiles) {
    if (t) late(t);
    step(std::true_type{}, t, sa, sb);
    sync(t + 1);
    step(std::false_type{}, t + 1, sb, sa);
  } else {
    step(std::false_type{}, t, sa, sb);
  }

  // ---- epilogue: normalise, store o1 (16 bytes per lane) and the natural-log LSE ---------------------------------------
  const float l = pair_sum(lsum);
  const int row = row0 + qc;
  const float inv = 1.0f / l;
  if (row < a.N && hi == 0)
    a.lse[static_cast<int64_t>(h) * a.N + row] = (mrun * a.scale_log2e + __builtin_amdgcn_logf(l)) * 0.6931471805599453f;
  uint16_t* op = a.out + (static_cast<int64_t>(row) * a.H + h) * DH + 8 * hi;
#pragma unroll
  for (int ot = 0; ot < OT; ++ot)
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      // column groups k = 2pr (registers 8pr..8pr+3) and k+1 (8pr+4..8pr+7): lane half 0 holds dims 8k..8k+3, half 1 dims 8k+4..8k+7
      uint32_t ax = bma::pack16<DT>(oacc[ot][8 * pr + 0] * inv, oacc[ot][8 * pr + 1] * inv);
      uint32_t ay = bma::pack16<DT>(oacc[ot][8 * pr + 2] * inv, oacc[ot][8 * pr + 3] * inv);
      uint32_t bx = bma::pack16<DT>(oacc[ot][8 * pr + 4] * inv, oacc[ot][8 * pr + 5] * inv);
      uint32_t by = bma::pack16<DT>(oacc[ot][8 * pr + 6] * inv, oacc[ot][8 * pr + 7] * inv);
      const u32x2 sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
      const u32x2 sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
      u32x4 ow;
      ow.x = sx.x; ow.y = sy.x; ow.z = sx.y; ow.w = sy.y;
      if (row < a.N) *reinterpret_cast<u32x4*>(op + 32 * ot + 16 * pr) = ow;
    }
}

// which kernel takes a launch: 0 = by shape (128-wide heads and a prefix of two chunks or more: the 32x32x16 kernel on
// four waves), 1 = the 16x16x32 kernel always, 4 / 8 = the 32x32x16 kernel on that many waves (measurement, tests)
int g_plan = 0;

template <int DT>
int launch(PArgs& a, int Dh, hipStream_t st) {
  int nw = 0;
  if (Dh == 128 && g_plan != 1) nw = g_plan == 8 ? 8 : (g_plan == 4 || a.P > 64) ? 4 : 0;
  if (nw) {
    // the descriptors of the chunk stream carry 32-bit byte counts and offsets
    if ((static_cast<int64_t>(a.P) * a.pk_rs + Dh) * 2 > 0x7fffffffLL || (static_cast<int64_t>(a.P) * a.pv_rs + Dh) * 2 > 0x7fffffffLL)
      nw = 0;
  }
  const int rows_per_wg = nw ? 32 * nw : kRowsPerWg;
  a.row_blocks = (a.N + rows_per_wg - 1) / rows_per_wg;
  const int64_t blocks = static_cast<int64_t>(a.row_blocks) * a.H;
  if (blocks > 0x7fffffffLL) return BMA_ELIMIT;
  const dim3 grid(static_cast<unsigned>(blocks));
  if (nw == 8) hipLaunchKernelGGL((prefix_attn32_kernel<DT, 8>), grid, dim3(512), 0, st, a);
  else if (nw == 4) hipLaunchKernelGGL((prefix_attn32_kernel<DT, 4>), grid, dim3(256), 0, st, a);
  else if (Dh == 64) hipLaunchKernelGGL((prefix_attn_kernel<DT, 64>), grid, dim3(64 * kWaves), 0, st, a);
  else hipLaunchKernelGGL((prefix_attn_kernel<DT, 128>), grid, dim3(64 * kWaves), 0, st, a);
  return BMA_OK;
}

}  // namespace

extern "C" int bma_prefix_attention(const void* q, int64_t q_rs, int64_t q_hs, const void* pk, int64_t pk_rs,
                                    int64_t pk_hs, const void* pv, int64_t pv_rs, int64_t pv_hs, int P, int64_t N,
                                    int H, int Hk, int Dh, int dtype, float scale, void* out, float* lse,
                                    void* stream) {
  if (N < 0 || H <= 0 || Hk <= 0 || P <= 0) return BMA_EINVAL;
  if (N == 0) return BMA_OK;
  if (!q || !pk || !pv || !out || !lse) return BMA_EINVAL;
  if (dtype != BMA_BF16 && dtype != BMA_F16) return BMA_EDTYPE;
  if ((Dh != 64 && Dh != 128) || H % Hk || N > 0x7fffffffLL) return BMA_ELIMIT;
  const int64_t strides[] = {q_rs, q_hs, pk_rs, pk_hs, pv_rs, pv_hs};
  for (int64_t s : strides)
    if (s % 8) return BMA_EALIGN;
  if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(pk) | reinterpret_cast<uintptr_t>(pv) |
       reinterpret_cast<uintptr_t>(out)) % 16 || reinterpret_cast<uintptr_t>(lse) % 4)
    return BMA_EALIGN;
  PArgs a;
  a.q = static_cast<const uint16_t*>(q); a.pk = static_cast<const uint16_t*>(pk); a.pv = static_cast<const uint16_t*>(pv);
  a.out = static_cast<uint16_t*>(out); a.lse = lse;
  a.q_rs = q_rs; a.q_hs = q_hs; a.pk_rs = pk_rs; a.pk_hs = pk_hs; a.pv_rs = pv_rs; a.pv_hs = pv_hs;
  a.N = static_cast<int>(N); a.H = H; a.Hk = Hk; a.P = P;
  a.row_blocks = 0;                                            // set by launch() for the kernel it picks
  a.scale_log2e = scale * 1.4426950408889634f;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // "bytes" of the profiler slot: q read + o written (the work itself is 4*N*P*Dh*H flops)
  BMA_PROF_BEGIN(BMA_K_PREFIX_ATTN, st, 2.0 * static_cast<double>(N) * H * Dh * 2);
  const int rc = dtype == BMA_BF16 ? launch<BMA_BF16>(a, Dh, st) : launch<BMA_F16>(a, Dh, st);
  BMA_PROF_END(BMA_K_PREFIX_ATTN, st);
  if (rc != BMA_OK) return rc;
  BMA_LAUNCH_CHECK();
  return BMA_OK;
}

extern "C" void bma_prefix_attention_set_plan(int kernel) {
  g_plan = (kernel == 1 || kernel == 4 || kernel == 8) ? kernel : 0;
}
