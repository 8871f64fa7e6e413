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

template <int DT>
int launch(const PArgs& a, int Dh, hipStream_t st) {
  const int hx = a.H % 8 == 0 ? a.H / 8 : 0;
  // with the XCD mapping the grid is padded so that every (b & 7, j % hx) pair has row_blocks entries
  const int64_t blocks = static_cast<int64_t>(a.row_blocks) * a.H;
  (void)hx;
  const dim3 grid(static_cast<unsigned>(blocks));
  if (Dh == 64) hipLaunchKernelGGL((prefix_attn_kernel<DT, 64>), grid, dim3(64 * kWaves), 0, st, a);
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
  a.row_blocks = static_cast<int>((N + kRowsPerWg - 1) / kRowsPerWg);
  a.scale_log2e = scale * 1.4426950408889634f;
  if (static_cast<int64_t>(a.row_blocks) * H > 0x7fffffffLL) return BMA_ELIMIT;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // "bytes" of the profiler slot: q read + o written (the work itself is 4*N*P*Dh*H flops)
  BMA_PROF_BEGIN(BMA_K_PREFIX_ATTN, st, 2.0 * static_cast<double>(N) * H * Dh * 2);
  const int rc = dtype == BMA_BF16 ? launch<BMA_BF16>(a, Dh, st) : launch<BMA_F16>(a, Dh, st);
  BMA_PROF_END(BMA_K_PREFIX_ATTN, st);
  if (rc != BMA_OK) return rc;
  BMA_LAUNCH_CHECK();
  return BMA_OK;
}
