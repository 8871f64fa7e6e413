// bma_ragged_attention: attention of the ragged scoring row list in ONE launch.
//
// Candidate scoring (reference bimodal_attack.py:1278-1310) with ragged rows (layout.ragged_plan):
// candidate i owns rows start[i] .. start[i]+len[i]-1 of the row list -- its tokens at positions
// first[i] .. first[i]+len[i]-1 behind the shared prefix.  A query at position j attends to
//   * the P prefix keys shared by every candidate,
//   * positions t < first[i]: its PARENT's keys/values, rows t of the row list,
//   * positions first[i] <= t <= j: its own rows.
// The library route needs five launches for that (prefix flash attention over all rows, three
// row gathers into a padded block, biased attention on the block, LSE merge) and moves every
// q/k/v/o row through HBM three times.  Here one workgroup owns one (candidate, head) -- and, for blocks
// longer than 64 tokens (a padded Gemma-3 candidate is the trivially ragged block first = 0, len = L), one
// stretch of 64 queries of it -- with one wave per 16 queries; the key/value rows it may see go through LDS
// once, 32 keys at a time:
//
//   S^T = K Q^T   v_mfma_f32_16x16x32: A = 16 keys x 32 dims (ds_read_b128 of an LDS row), B = 32 dims
//                 x 16 queries (loaded once from global memory in operand layout, 16 B per lane)
//   softmax       online over 32-key chunks; a query is a lane column, so its running max/sum
//                 live in the lane (two xor-shuffles fold the four 16-lane groups)
//   O^T = V^T P^T the S^T accumulator tile IS the B operand (keys on registers, query on the
//                 lane): no shuffles; V^T comes from the LDS image of the chunk's value rows
//                 (coalesced 16-byte loads in, ds_read_b64_tr_b16 transposing reads out)
//   epilogue      normalise, optionally merge with a prefix partial (o1, lse1) computed elsewhere
//                 (long image prefixes keep the library flash kernel), store 8 B per lane
//
// Algorithmic bytes per launch: q + k + v read once, o written once = 4 * N * H * Dh * es
// (prefix and parent rows are re-read from L2).
#include <cstdio>
#ifdef BMA_LONG_STAMPS
#include <cstdlib>
#endif
#include <type_traits>
#include <vector>

#include "bma_common.h"
#include "bma_lds.h"
#include "bma_profile.h"

namespace {

using bma::uint4_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short short4_t __attribute__((ext_vector_type(4)));

struct Args {
  const uint16_t *q, *k, *v, *pk, *pv, *o1;
  const float* lse1;
  uint16_t* out;
  const int *start, *first, *len;
  int64_t q_rs, q_hs, k_rs, k_hs, v_rs, v_hs, pk_rs, pk_hs, pv_rs, pv_hs;
  int B2, H, Hk, P, N, nz;
  float scale_log2e;
  unsigned long long* stamps;   // diagnostic builds only (-DBMA_LONG_STAMPS): 12 clock stamps per wave
};

template <int DT>
__device__ __forceinline__ f32x4 mfma(const uint4_t& a, const uint4_t& b, const f32x4& c) {
  if (DT == BMA_BF16)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int DT>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) { return bma::pack16<DT>(lo, hi); }

// One workgroup per (candidate, head, block of 16*QT queries); wave w owns query tile w of the block.  The
// keys/values the block may see go through LDS once, 32 keys at a time, shared by the waves; a wave skips
// chunks that lie entirely behind its last query (causal).  Staging: ONE image pair, chunk c+1 in flight in
// registers while chunk c is multiplied, two barriers per chunk -- the least LDS (18 KB at 128-wide heads, 35 KB
// at 256), hence the most workgroups per CU.  Measured alternatives on MI355X (tools/kernel_bench.py, C3 row
// list / Gemma-3 blocks / C4 blocks): two image pairs with one barrier per chunk 164 / 363 / 176 us, every chunk
// resident in LDS with a single barrier 284 / 367 / 196 us, against 157 / 344 / 166 us for this scheme: the
// kernel is bound by how many workgroups a CU holds, not by its barriers.
// (waves per SIMD asked of the register allocator at 256-wide heads: 164 registers instead of 240 is a third wave per
// SIMD, no spills, and the long Gemma-3 blocks -- a chain of one memory latency per 32-key chunk -- ran 11 % faster,
// 318 -> 282 us.  At 128 the same request, 119 registers instead of 144 and a fourth wave, measured 3-10 % SLOWER on
// the C3 row list and the C4 blocks, so nothing is asked there.)
template <int DT, int QT, int DH>
__global__ __launch_bounds__(64 * QT) __attribute__((amdgpu_waves_per_eu(DH == 256 ? 2 : 1)))
void ragged_attn_kernel(const Args a) {
  constexpr int KS = DH / 32;      // k-steps of the QK product
  constexpr int NT = DH / 16;      // 16-dim tiles of the output
  constexpr int PITCH = DH + 16;   // elements per LDS row (row + 32 B: conflict-free transposing reads)
  constexpr int PIECES = DH / 8;   // 16-byte pieces per row
  constexpr int NTHR = 64 * QT;
  constexpr int IMG = 32 * PITCH;  // elements of one 32-row image
  __shared__ __attribute__((aligned(16))) uint16_t lds[2 * IMG];   // K image, V image
  const int tid = threadIdx.x;
  const int lane = tid & 63, qt = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  // Workgroup -> (candidate i, head h, stretch z).  The workgroups that read the same key/value rows from HBM --
  // the H/Hk query heads on one key/value head, times the nz 64-query stretches of a long block -- get linear ids
  // 8 apart, back to back: same XCD (workgroups go round the eight XCDs by linear id), hence the same L2, and
  // dispatched within a few dozen ids of each other, so the rows are still there.  With (i, h, z) on the grid axes
  // the sharers were a whole launch apart and on two XCDs: Gemma-3 blocks (303 tokens, 8 heads on 4) moved 2.85x
  // their algorithmic bytes through HBM.  One stretch and H = Hk (the LLaVA row list) is the old order: i fastest.
  const int rep = a.H / a.Hk;
  const int S = rep * a.nz;
  const int lin = blockIdx.x, t = lin >> 3;
  const int grp = (t / S) * 8 + (lin & 7);                // (candidate, key/value head), candidate fastest
  if (grp >= a.B2 * a.Hk) return;                         // uniform over the workgroup
  const int sh = t % S;
  const int i = grp % a.B2, hk = grp / a.B2;
  const int h = hk * rep + sh % rep;
  const int q0 = 16 * QT * (a.nz - 1 - sh / rep);         // longest stretch first
  const int st = a.start[i], p0 = a.first[i], ln = a.len[i];
  if (q0 >= ln) return;                                   // uniform over the workgroup
  const int P = a.P;
  const int nkeys = P + p0 + ln;
  const int qend = q0 + 16 * QT < ln ? q0 + 16 * QT : ln; // one past the last query of this workgroup
  const int kend = P + p0 + qend;                         // keys any of its queries may see
  const float NEG = -__builtin_inff();

  // Q tile as the B operand: lane (query r, dims 8g.. of k-step ks)
  uint4_t qf[KS];
  {
    int qi = q0 + 16 * qt + r;
    qi = qi < ln ? qi : ln - 1;
    const uint16_t* qp = a.q + static_cast<int64_t>(st + qi) * a.q_rs + static_cast<int64_t>(h) * a.q_hs + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const uint4_t*>(qp + 32 * ks);
  }
  const int jq = p0 + q0 + 16 * qt + r;                   // this lane's query position behind the prefix
  const int last_key = P + p0 + q0 + 16 * qt + 15;        // last key index any query of the tile may see
  const bool tile_live = q0 + 16 * qt < ln;

  f32x4 oacc[NT];
#pragma unroll
  for (int dt = 0; dt < NT; ++dt) oacc[dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  float mrun = NEG, lsum = 0.0f;

  constexpr int ITEMS = (32 * PIECES + NTHR - 1) / NTHR;
  // the 16-byte piece `idx` of chunk c: key row (prefix, parent's or own) and where it lands in an image
  auto src = [&](int c, int idx, const uint16_t*& kp, const uint16_t*& vp) {
    const int key = idx / PIECES, piece = idx % PIECES;
    int t = 32 * c + key;
    t = t < nkeys ? t : nkeys - 1;
    if (t < P) {
      kp = a.pk + static_cast<int64_t>(t) * a.pk_rs + static_cast<int64_t>(hk) * a.pk_hs + 8 * piece;
      vp = a.pv + static_cast<int64_t>(t) * a.pv_rs + static_cast<int64_t>(hk) * a.pv_hs + 8 * piece;
    } else {
      const int tt = t - P;
      const int64_t row = tt < p0 ? tt : st + (tt - p0);   // the parent's row, or this candidate's own
      kp = a.k + row * a.k_rs + static_cast<int64_t>(hk) * a.k_hs + 8 * piece;
      vp = a.v + row * a.v_rs + static_cast<int64_t>(hk) * a.v_hs + 8 * piece;
    }
  };
  const int chunks = (kend + 31) >> 5;

  // ---- one chunk out of LDS images kl / vl ------------------------------------------------------
  auto compute = [&](int c, const uint16_t* kl, const uint16_t* vl) {
    // S^T = K Q^T: A = 16 keys x 32 dims from LDS rows, B = the Q fragments
    f32x4 s[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const uint16_t* kr = kl + (16 * kt + r) * PITCH + 8 * g;
      f32x4 acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc = mfma<DT>(*reinterpret_cast<const uint4_t*>(kr + 32 * ks), qf[ks], acc);
      s[kt] = acc;
    }
    // online softmax; the probabilities become the B operand of the second product
    float e[2][4];
    float cmax = NEG;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int te = 32 * c + 16 * kt + 4 * g + rr;
        const bool ok = te < nkeys && (te < P || te - P <= jq);
        const float v = ok ? s[kt][rr] * a.scale_log2e : NEG;
        e[kt][rr] = v;
        cmax = fmaxf(cmax, v);
      }
    cmax = fmaxf(cmax, __shfl_xor(cmax, 16, BMA_WAVE));
    cmax = fmaxf(cmax, __shfl_xor(cmax, 32, BMA_WAVE));
    const float mnew = fmaxf(mrun, cmax);
    const float mm = mnew == NEG ? 0.0f : mnew;             // nothing visible yet: keep everything at zero
    const float alpha = __builtin_amdgcn_exp2f(mrun - mm);
    float rs = 0.0f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        e[kt][rr] = __builtin_amdgcn_exp2f(e[kt][rr] - mm);
        rs += e[kt][rr];
      }
    lsum = lsum * alpha + rs;
    mrun = mnew;
    uint4_t pf;
    pf.x = pack2<DT>(e[0][0], e[0][1]);
    pf.y = pack2<DT>(e[0][2], e[0][3]);
    pf.z = pack2<DT>(e[1][0], e[1][1]);
    pf.w = pack2<DT>(e[1][2], e[1][3]);
    // O^T = alpha O^T + V^T P^T: element j of lane group g is key 32c + (j<4 ? 4g+j : 16+4g+j-4);
    // ds_read_b64_tr_b16 hands each 16-lane group a 4-key x 16-dim block column-major, which is the
    // A-operand fragment (dim on the lane, 4 keys in the elements)
    const int q4 = r >> 2, p4 = r & 3;
    const uint16_t* rd = vl + (4 * g + q4) * PITCH + 4 * p4;
#pragma unroll
    for (int dt = 0; dt < NT; ++dt) {
      const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(rd + 16 * dt));
      const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) short4_t*)(rd + 16 * PITCH + 16 * dt));
      const bma::uint2_t l2 = __builtin_bit_cast(bma::uint2_t, lo), h2 = __builtin_bit_cast(bma::uint2_t, hi);
      uint4_t vf;
      vf.x = l2.x; vf.y = l2.y; vf.z = h2.x; vf.w = h2.y;
      f32x4 o = oacc[dt];
      o[0] *= alpha; o[1] *= alpha; o[2] *= alpha; o[3] *= alpha;
      oacc[dt] = mfma<DT>(vf, pf, o);
    }
  };

  uint4_t kreg[ITEMS], vreg[ITEMS];
  auto fetch = [&](int c) {
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int idx = tid + it * NTHR;
      if (idx < 32 * PIECES) {
        const uint16_t *kp, *vp;
        src(c, idx, kp, vp);
        kreg[it] = *reinterpret_cast<const uint4_t*>(kp);
        vreg[it] = *reinterpret_cast<const uint4_t*>(vp);
      }
    }
  };
  fetch(0);
  // The Q fragments must have LANDED before the loop: behind the lane-masked stores below hipcc's wait insertion still
  // counts them as pending and puts `s_waitcnt vmcnt(3..0)` in front of the first S products of EVERY chunk -- waits which
  // drain the next chunk's fetch, issued a few instructions earlier, before the current one is multiplied (round 4, seen in
  // the ISA of csrc/causal_attention.hip first).  A use in an empty asm statement makes it wait here, once.
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) asm volatile("" ::"v"(qf[ks]));
  for (int c = 0; c < chunks; ++c) {
    if (c) __syncthreads();                               // the previous chunk's readers are done
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int idx = tid + it * NTHR;
      if (idx < 32 * PIECES) {
        const int off = (idx / PIECES) * PITCH + 8 * (idx % PIECES);
        *reinterpret_cast<uint4_t*>(lds + off) = kreg[it];
        *reinterpret_cast<uint4_t*>(lds + IMG + off) = vreg[it];
      }
    }
    __syncthreads();
    if (c + 1 < chunks) fetch(c + 1);                     // in flight while chunk c is multiplied
    if (tile_live && 32 * c <= last_key) compute(c, lds, lds + IMG);
  }

  // ---- epilogue -----------------------------------------------------------------------------
  // The accumulator layout (lane = query, 4 dims of every 16-dim tile) stored as it is makes 32-byte pieces
  // scattered over 16 rows, eight stores per wave.  Up to 128-wide heads the wave's 16 finished rows go through its
  // own quarter of the key/value images instead and leave as whole rows, 16 bytes per lane: C3 row list 161.7 ->
  // 158.0 us, C4 blocks 167.6 -> 160.9 us (same box, bit-identical output).  At 256 the barrier this needs costs
  // more than the stores (278 -> 293 us on Gemma-3 blocks: the wave with the last queries has the most chunks and
  // everybody waits for it), so those store straight from the registers.
  float l = lsum;
  l += __shfl_xor(l, 16, BMA_WAVE);
  l += __shfl_xor(l, 32, BMA_WAVE);
  const int qi = q0 + 16 * qt + r;
  constexpr bool STAGED = DH <= 128;
  if (!STAGED && qi >= ln) return;
  const int64_t o_rs = static_cast<int64_t>(a.H) * DH;
  const int64_t row = st + (qi < ln ? qi : ln - 1);
  const float inv = l > 0.0f ? 1.0f / l : 0.0f;
  float w1 = 0.0f;
  if (a.o1) {
    // natural-log LSE of this part; weight of the prefix partial = 1 / (1 + exp(lse2 - lse1))
    const float lse2 = l > 0.0f ? (mrun + __builtin_amdgcn_logf(l)) * 0.6931471805599453f : NEG;
    const float l1 = a.lse1[static_cast<int64_t>(h) * a.N + row];
    w1 = 1.0f / (1.0f + expf(lse2 - l1));
  }
  if (STAGED) __syncthreads();                            // every wave is done reading the images
  uint16_t* ot = lds + 16 * qt * PITCH;                   // rows 16qt .. 16qt+15 of the 64 the two images hold
  uint16_t* op = a.out + row * o_rs + static_cast<int64_t>(h) * DH + 4 * g;
  const uint16_t* o1p = a.o1 ? a.o1 + row * o_rs + static_cast<int64_t>(h) * DH + 4 * g : nullptr;
#pragma unroll
  for (int dt = 0; dt < NT; ++dt) {
    float o[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) o[rr] = oacc[dt][rr] * inv;
    if (o1p) {
      const bma::uint2_t pw = *reinterpret_cast<const bma::uint2_t*>(o1p + 16 * dt);
      const float p1[4] = {bma::unpack16<DT>(pw.x, 0), bma::unpack16<DT>(pw.x, 1), bma::unpack16<DT>(pw.y, 0),
                           bma::unpack16<DT>(pw.y, 1)};
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) o[rr] = o[rr] + w1 * (p1[rr] - o[rr]);
    }
    bma::uint2_t ow;
    ow.x = pack2<DT>(o[0], o[1]);
    ow.y = pack2<DT>(o[2], o[3]);
    if (STAGED)
      *reinterpret_cast<bma::uint2_t*>(ot + r * PITCH + 16 * dt + 4 * g) = ow;
    else
      *reinterpret_cast<bma::uint2_t*>(op + 16 * dt) = ow;
  }
  if (STAGED) {
    // the wave reads back what the wave wrote: LDS operations of one wave complete in order, no barrier
    constexpr int RPP = 64 / PIECES > 16 ? 16 : 64 / PIECES;   // rows per pass (PIECES lanes cover one row)
#pragma unroll
    for (int ps = 0; ps < 16 / RPP; ++ps) {
      const int rl = ps * RPP + lane / PIECES, piece = lane % PIECES;
      const int qrow = q0 + 16 * qt + rl;
      if (rl < 16 && qrow < ln) {
        const uint4_t val = *reinterpret_cast<const uint4_t*>(ot + rl * PITCH + 8 * piece);
        *reinterpret_cast<uint4_t*>(a.out + (st + qrow) * o_rs + static_cast<int64_t>(h) * DH + 8 * piece) = val;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Long blocks (Gemma-3 joint scoring: padded candidates of 303 tokens, 256-wide heads, two query heads on one
// key/value head).  The kernel above gives such a block one workgroup per 64 queries of ONE head, one 32-key chunk in
// flight through staging registers, two barriers per chunk: 246 us per launch on the Gemma-3 blocks (rocprofv3), 0.27
// of the HBM roofline, 0.09 of MFMA peak, with the waves waiting and every pipe idle.  This one is built like a
// flash-attention forward instead -- 203 us on the same input:
//
//   workgroup   persistent, one per CU: 4*REP waves walk a list of items = 64 queries of REP query heads that share one
//               key/value head (a wave owns one 16-query tile); a key/value row fetched into LDS serves 64*REP rows.
//               The end of an item -- last stages, normalisation, stores -- overlaps the start of the next, whose first
//               keys/values and Q rows are requested while the current item's last pair is multiplied
//   staging     32 keys per stage, a ring of two PAIRS of stages filled by LDS-DMA (global_load_lds_dwordx4: no
//               staging registers, no ds_write); one pair is in flight while the other is multiplied; per pair one
//               s_waitcnt vmcnt(0) + one raw s_barrier.  The DMA instructions are issued from inside the stage code
//               (behind the QK products), not together behind the barrier
//   LDS image   rows of 2*DH bytes without padding (the DMA writes 1 KiB = 64 lanes x 16 B contiguously); bank
//               conflicts are avoided by permuting the 16-byte pieces inside each 256-byte half row ON THE SOURCE
//               SIDE (the lane that fills LDS piece `pos` of key row `row` loads piece pos ^ f(row)) and applying the
//               same XOR to the reads:  K image f = row & 15 (ds_read_b128 by 16 rows x 4 pieces: 16 distinct slots
//               per 16-lane group), V image f = (row & 7) << 1 (transposing reads by 8 rows x 32 bytes per half wave)
//   LDS reads   inline asm with hand-counted s_waitcnt lgkmcnt: the compiler cannot tell an LDS read from the DMA
//               writes in flight and drains the ring (vmcnt(0)) in front of every read it schedules itself
//   softmax     as in prefix_attention.hip: maximum on raw scores, the scale folded into the exponent's fma, row
//               reductions by v_permlane16/32_swap, rescaling skipped (wave-uniform) when no maximum moved; only the
//               stages that touch the diagonal or the end of the keys run the masked copy of the code
//   epilogue    the pair of the ring multiplied last is free: every wave parks its 16 finished rows in its own 1/8
//               of it and stores whole rows, 16 bytes per lane
//
// The REP heads' waves take the 16-query tiles in opposite order (tile j and tile 3-j land on the same SIMD), so in
// every pair of stages each SIMD has the same number of causal stages to multiply.
// Where the time goes (tools/stamps_report.py on a -DBMA_LONG_STAMPS build; PMC: tools/pmc_kernel.py): the two waves of
// a SIMD issue 18k VALU + 11k SALU + 2.4k MFMA + 4k LDS instructions each per launch, which fills the SIMD's issue
// slots (400k cycles per wave, 443k per launch): the kernel is instruction-issue bound, not latency bound -- a third
// wave per SIMD (12 waves, 168 registers) made every stage 1.7x longer.
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

using bma::u32x2;
using bma::u32x4;
using bma::tr_read;
using bma::row_read;
using bma::wait_rows;
using bma::wait_lgkm;

__device__ __forceinline__ float vmax1(float x, float y) {      // one v_max_f32, no canonicalising pre-ops
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
// all-reduce over the four 16-lane rows of a wave (a query's keys are spread over lanes l, l+16, l+32, l+48)
__device__ __forceinline__ float rows_max(float x) {
  u32x2 p = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const float m = vmax1(__uint_as_float(p.x), __uint_as_float(p.y));
  u32x2 q = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  return vmax1(__uint_as_float(q.x), __uint_as_float(q.y));
}
__device__ __forceinline__ float rows_sum(float x) {
  u32x2 p = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const float m = __uint_as_float(p.x) + __uint_as_float(p.y);
  u32x2 q = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  return __uint_as_float(q.x) + __uint_as_float(q.y);
}

constexpr int kLongMin = 96;     // shortest max_len that takes this kernel
constexpr int kRing = 4;         // stages of 32 keys in the LDS ring: two pairs

#ifndef BMA_LONG_SCHED
#define BMA_LONG_SCHED 7
#endif
#define BMA_LONG_SB0 do { if (BMA_LONG_SCHED & 1) __builtin_amdgcn_sched_barrier(0); } while (0)
#define BMA_LONG_SB1 do { if (BMA_LONG_SCHED & 2) __builtin_amdgcn_sched_barrier(0); } while (0)
#define BMA_LONG_SB2 do { if (BMA_LONG_SCHED & 4) __builtin_amdgcn_sched_barrier(0); } while (0)

struct LongItem {       // one (candidate, key/value head, head group, stretch): wave-uniform
  int i, hk, hsel, q0, st, p0, ln, nkeys, chunks;
};

template <int DT, int DH, int REP, int QT, int TW>
__global__ __launch_bounds__(64 * TW * REP) void ragged_attn_long_kernel(const Args a) {
  constexpr int NW = TW * REP;              // waves
  constexpr int QW = 16 * QT;               // queries per wave
  constexpr int QB = TW * QW;               // queries per workgroup and head
  constexpr int KS = DH / 32, NT = DH / 16;
  constexpr int ROWB = 2 * DH;              // bytes of an LDS row
  constexpr int IMGB = 32 * ROWB;           // one 32-key image
  constexpr int STAGEB = 2 * IMGB;          // K image + V image
  constexpr int LPR = DH / 8;               // lanes (16-byte pieces) per row
  constexpr int RPI = 64 / LPR;             // rows per DMA instruction (1 KiB)
  constexpr int NPI = 32 / RPI;             // DMA instructions per 32-key image
  constexpr int IPW = (NPI + NW - 1) / NW;  // ... per wave (piece j = wave + it * NW, when j < NPI)
  static_assert(NW * QW * ROWB <= 2 * STAGEB, "the epilogue parks a wave's rows in ONE pair of the ring");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[kRing * STAGEB];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int rep = a.H / a.Hk;
  const int hg = rep / REP;
  const int S = hg * a.nz;                                 // workgroup items that share one (candidate, key/value head)
  const int P = a.P;
  const float NEG = -__builtin_inff();
  // ---- the items of this workgroup -------------------------------------------------------------------------------
  // The kernel is persistent: one workgroup per CU (the ring is 128 KB) walks a list of items, and the end of one item
  // -- its last stages, the normalisation, the stores -- overlaps the start of the next, whose first key/value pair
  // and Q rows are requested while the last pair of the current item is multiplied.  With one workgroup per item the
  // 5-7k cycles in front of the first product (64 KB of Q + 64 KB of K/V at the CU's fill rate), the 5k behind the last
  // and the hand-over between two workgroups (the second cannot start before the first has freed the LDS) were a
  // third of a workgroup's 36k cycles, with nothing else resident on the CU to hide them.
  // Workgroup b runs on XCD b & 7 (round-robin dispatch).  The sets of sharers -- the S items on one key/value head
  // of one candidate -- go round the XCDs; inside an XCD its list of items, set after set, goes round the XCD's
  // workgroups: the S items of a set run on S workgroups of ONE XCD at about the same time, so the rows one of them
  // pulls from HBM are in that L2 for the others.
  const int xcd = blockIdx.x & 7, wgx = gridDim.x >> 3;
  const int groups = a.B2 * a.Hk;
  const int n_items = groups > xcd ? ((groups - xcd + 7) >> 3) * S : 0;
  auto decode = [&](int j, LongItem& it) -> int {          // the first item with live queries at j, j + wgx, ...; n_items if none
    for (; j < n_items; j += wgx) {
      const int sh = j % S, grp = xcd + 8 * (j / S);
      const int i = grp % a.B2;
      const int ln = a.len[i];
      const int q0 = QB * (a.nz - 1 - sh / hg);              // longest stretch of a set first
      if (q0 < ln) {
        it.i = i; it.hk = grp / a.B2; it.hsel = sh % hg; it.q0 = q0; it.st = a.start[i]; it.p0 = a.first[i]; it.ln = ln;
        it.nkeys = P + it.p0 + ln;
        it.chunks = (P + it.p0 + (q0 + QB < ln ? q0 + QB : ln) + 31) >> 5;     // stages the workgroup walks
        return j;
      }
    }
    return n_items;
  };
  LongItem cur, nxt;
  int j = decode(static_cast<int>(blockIdx.x >> 3), cur);
  if (j >= n_items) return;                               // uniform over the workgroup
#ifdef BMA_LONG_STAMPS
  const unsigned long long ts0 = __builtin_amdgcn_s_memtime(), tr0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long ts_wait = 0, ts_qk = 0, ts_sm = 0, ts_pv = 0, ts_epi = 0, n_it = 0, n_mine = 0, n_walk = 0;
#endif
  // (s_setprio 1 on the second head's waves -- the younger ones on their SIMDs, which reached every barrier last -- only
  // swapped who waits: the SIMDs' issue slots are full, 413k cycles per wave against 397k.)
  const int hs = wave / TW;                               // which of the REP heads
  const int pw = hs & 1 ? TW - 1 - wave % TW : wave % TW; // which 16*QT queries of the stretch

  uint4_t qf[QT][KS];                                     // Q tiles as B operands: lane (query r, dims 8g.. of k-step ks)
  auto load_q = [&](const LongItem& it) {
    const int h = it.hk * rep + it.hsel * REP + hs;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      int qi = it.q0 + QW * pw + 16 * t + r;
      qi = qi < it.ln ? qi : it.ln - 1;
      const uint16_t* qp = a.q + static_cast<int64_t>(it.st + qi) * a.q_rs + static_cast<int64_t>(h) * a.q_hs + 8 * g;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) qf[t][ks] = *reinterpret_cast<const uint4_t*>(qp + 32 * ks);
    }
  };

  // ---- LDS-DMA: this lane's piece of every row it fetches ----------------------------------------------------------
  const int lrow = lane / LPR, wp = lane % LPR;
  const int half = wp >> 4, pos = wp & 15;
  // Stages that lie entirely inside the candidate's own rows (all but the first one or two and possibly the last) take
  // their addresses from per-lane pointers computed once per item: one 64-bit multiply-add per piece instead of the
  // prefix / parent / own case analysis (31 vector instructions per K/V pair of pieces; the loop is VALU-bound).
  // (when a wave's pieces are a multiple of 16 rows apart they all have the same swizzle: one offset register per
  // image, the piece's distance is a scalar)
  constexpr bool SAME = (NW * RPI) % 16 == 0;
  constexpr int NL = SAME ? 1 : IPW;
  uint32_t klane[NL], vlane[NL];                            // this lane's byte offset inside a stage of own rows
#pragma unroll
  for (int q = 0; q < NL; ++q) {
    const int row = (wave + q * NW) * RPI + lrow;
    klane[q] = 2 * (row * static_cast<uint32_t>(a.k_rs) + 128 * half + 8 * (pos ^ (row & 15)));
    vlane[q] = 2 * (row * static_cast<uint32_t>(a.v_rs) + 128 * half + 8 * (pos ^ ((row & 7) << 1)));
  }
  // own row of key 0 of an item (wave-uniform; may lie in front of the tensor)
  auto own_k = [&](const LongItem& it) {
    return reinterpret_cast<const unsigned char*>(a.k) +
           2 * ((static_cast<int64_t>(it.st) - (P + it.p0)) * a.k_rs + static_cast<int64_t>(it.hk) * a.k_hs);
  };
  auto own_v = [&](const LongItem& it) {
    return reinterpret_cast<const unsigned char*>(a.v) +
           2 * ((static_cast<int64_t>(it.st) - (P + it.p0)) * a.v_rs + static_cast<int64_t>(it.hk) * a.v_hs);
  };
  const int64_t kstep = 64 * a.k_rs, vstep = 64 * a.v_rs;   // bytes per stage of 32 rows
  // kc / vc: the item's own row of key 32c (wave-uniform; kept as running pointers by the caller)
  auto issue = [&](const LongItem& it, int c, int slot, const unsigned char* kc, const unsigned char* vc) {
    if (32 * c >= P + it.p0 && 32 * c + 32 <= it.nkeys) {                    // wave-uniform
#pragma unroll
      for (int q = 0; q < IPW; ++q) {
        if (NPI % NW && wave + q * NW >= NPI) break;         // wave-uniform
        unsigned char* kd = lds + slot * STAGEB + (wave + q * NW) * RPI * ROWB;
        const unsigned char* kq = kc + (SAME ? 2 * q * NW * RPI * a.k_rs : 0);     // wave-uniform
        const unsigned char* vq = vc + (SAME ? 2 * q * NW * RPI * a.v_rs : 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kq + klane[SAME ? 0 : q]),
                                         (__attribute__((address_space(3))) void*)kd, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vq + vlane[SAME ? 0 : q]),
                                         (__attribute__((address_space(3))) void*)(kd + IMGB), 16, 0, 0);
      }
      return;
    }
    // a stage with prefix keys, the parent's rows or the end of the block in it: per-lane row lookup, written with
    // selects (as branches the three cases cost two exec-mask regions per piece)
    const unsigned char* pkb = reinterpret_cast<const unsigned char*>(a.pk) + 2 * static_cast<int64_t>(it.hk) * a.pk_hs;
    const unsigned char* pvb = reinterpret_cast<const unsigned char*>(a.pv) + 2 * static_cast<int64_t>(it.hk) * a.pv_hs;
    const unsigned char* kb = reinterpret_cast<const unsigned char*>(a.k) + 2 * static_cast<int64_t>(it.hk) * a.k_hs;
    const unsigned char* vb = reinterpret_cast<const unsigned char*>(a.v) + 2 * static_cast<int64_t>(it.hk) * a.v_hs;
    const uint32_t pk_rb = 2 * static_cast<uint32_t>(a.pk_rs), pv_rb = 2 * static_cast<uint32_t>(a.pv_rs);
    const uint32_t k_rb = 2 * static_cast<uint32_t>(a.k_rs), v_rb = 2 * static_cast<uint32_t>(a.v_rs);
#pragma unroll
    for (int q = 0; q < IPW; ++q) {
      if (NPI % NW && wave + q * NW >= NPI) break;           // wave-uniform
      const int row = (wave + q * NW) * RPI + lrow;          // key row inside the stage
      int t = 32 * c + row;
      t = t < it.nkeys ? t : it.nkeys - 1;                   // past the end: the last key again (masked; finite)
      const bool inpre = t < P;
      const int tt = t - P;
      const uint32_t grow = inpre ? t : (tt < it.p0 ? tt : it.st + (tt - it.p0));   // prefix row / the parent's row / own row
      const unsigned char* kp = (inpre ? pkb : kb) + static_cast<uint64_t>(grow) * (inpre ? pk_rb : k_rb) +
                                2 * (128 * half + 8 * (pos ^ (row & 15)));
      const unsigned char* vp = (inpre ? pvb : vb) + static_cast<uint64_t>(grow) * (inpre ? pv_rb : v_rb) +
                                2 * (128 * half + 8 * (pos ^ ((row & 7) << 1)));
      unsigned char* kd = lds + slot * STAGEB + (wave + q * NW) * RPI * ROWB;      // wave-uniform; the DMA adds lane*16
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)kp,
                                       (__attribute__((address_space(3))) void*)kd, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)vp,
                                       (__attribute__((address_space(3))) void*)(kd + IMGB), 16, 0, 0);
    }
  };

  f32x4 oacc[QT][NT];
  float mrun[QT], lsum[QT];
  int jq[QT];
  int qb = 0;                                              // first query of this wave in the current item

  // fragment offsets inside a stage (bytes)
  const int q4 = r >> 2, p4 = r & 3;
  const int vrow = 4 * g + q4;                             // key row of this lane's transposing reads (and + 16)
  const int kbase = r * ROWB;
  const int vbase = IMGB + vrow * ROWB + 8 * p4;
  const int v7 = vrow & 7;

  // The pair after the one being multiplied -- the item's next two stages, or the NEXT item's first two -- is planned
  // at the pair's barrier and requested from INSIDE the stage code, one stage per stage multiplied, behind the QK
  // products: eight waves issuing their 8 DMA instructions together right behind the barrier queued on the CU's one
  // address path for ~2k cycles per pair (stamps: more time than the products).  Whatever a wave has not requested
  // when it reaches the next barrier (it skipped a stage, or multiplies none) it requests there.
  // (Splitting a stage's four DMA instructions between the K reads' latency and the end of the QK products: 244 us
  // against 231; with the level-2 stamps the DMA block costs ~550 cycles per stage where it stands.)
  int pf_kind = 0, pf_c0 = 0, pf_n = 0, pf_done = 0, pf_slot = 0;   // wave-uniform: 0 nothing / 1 current item / 2 next item
  const unsigned char *pf_k = nullptr, *pf_v = nullptr;     // own row of key 32 * pf_c0 of the planned item
  const unsigned char *kfetch = nullptr, *vfetch = nullptr; // ... of the current item's next stage to plan (running)
  auto pf_issue = [&](int k) {
    if (k < pf_n && !(pf_done & (1 << k))) {
      pf_done |= 1 << k;
      if (pf_kind == 1) issue(cur, pf_c0 + k, pf_slot + k, pf_k + k * kstep, pf_v + k * vstep);
      else issue(nxt, pf_c0 + k, pf_slot + k, pf_k + k * kstep, pf_v + k * vstep);
    }
  };

  auto compute = [&](int c, const unsigned char* sb) {
#if defined(BMA_LONG_STAMPS) && BMA_LONG_STAMPS >= 2
    const unsigned long long tc0 = __builtin_amdgcn_s_memtime();
#endif
    // ---- S^T = K Q^T for both query tiles -----------------------------------------------------------------------
    // The K fragments are read by hand as well: all of a stage's rows are requested at once (the compiler kept five
    // reads in flight and waited for lgkmcnt(0) six times per stage), the second half lands while the first is multiplied.
    f32x4 s[QT][2];
    u32x4 kf[2][KS];
    const uint32_t ka = bma::lds_addr(sb) + kbase;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const uint32_t ad = ka + 16 * ((4 * (ks & 3) + g) ^ r);
        if (kt == 0) kf[kt][ks] = ks >> 2 ? row_read<256>(ad) : row_read<0>(ad);
        else kf[kt][ks] = ks >> 2 ? row_read<16 * ROWB + 256>(ad) : row_read<16 * ROWB>(ad);
      }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      if (kt == 0) wait_rows<KS, KS>(kf[0]);
      else wait_rows<0, KS>(kf[1]);
#pragma unroll
      for (int t = 0; t < QT; ++t) s[t][kt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const uint4_t kfr = __builtin_bit_cast(uint4_t, kf[kt][ks]);
#pragma unroll
        for (int t = 0; t < QT; ++t) s[t][kt] = mfma<DT>(kfr, qf[t][ks], s[t][kt]);
      }
      BMA_LONG_SB0;
    }
    pf_issue(c & 1);
#if defined(BMA_LONG_STAMPS) && BMA_LONG_STAMPS >= 2
    asm volatile("" ::"v"(s[0][0][0]), "v"(s[0][1][0]));       // the QK products are done
    const unsigned long long tc1 = __builtin_amdgcn_s_memtime();
#endif
    // ---- online softmax per tile ------------------------------------------------------------------------------------
    uint4_t pf[QT];
    float alpha[QT];
    auto softmax = [&](auto masked) {
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        float e[2][4];
        float cmax = NEG;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            float v = s[t][kt][rr];
            if (decltype(masked)::value) {
              const int te = 32 * c + 16 * kt + 4 * g + rr;
              const bool ok = te < cur.nkeys && (te < P || te - P <= jq[t]);
              v = ok ? v : NEG;
            }
            e[kt][rr] = v;
            cmax = vmax1(cmax, v);
          }
        cmax = rows_max(cmax);
        // key 0 (prefix, parent or the block's first token) is visible to every query and stage 0 comes first, so the
        // running maximum is finite from the first stage on and NEG - NEG never happens
        const float mnew = vmax1(mrun[t], cmax);
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
        pf[t].x = pack2<DT>(e[0][0], e[0][1]);
        pf[t].y = pack2<DT>(e[0][2], e[0][3]);
        pf[t].z = pack2<DT>(e[1][0], e[1][1]);
        pf[t].w = pack2<DT>(e[1][2], e[1][3]);
      }
    };
    if (32 * c + 32 > cur.nkeys || 32 * c + 31 - P > cur.p0 + qb) softmax(std::true_type{});
    else softmax(std::false_type{});
    bool moved = false;
#pragma unroll
    for (int t = 0; t < QT; ++t) moved = moved || (alpha[t] != 1.0f);
    if (__builtin_amdgcn_ballot_w64(moved) != 0) {
#pragma unroll
      for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int dt = 0; dt < NT; ++dt) {
          oacc[t][dt][0] *= alpha[t]; oacc[t][dt][1] *= alpha[t]; oacc[t][dt][2] *= alpha[t]; oacc[t][dt][3] *= alpha[t];
        }
    }
    // ---- O^T += V^T P^T: every V^T fragment feeds all the wave's tiles ------------------------------------------------
    // The transposing reads are inline asm: issued through the builtin, the compiler puts an s_waitcnt vmcnt(0) in
    // front of them (an LDS read it cannot tell apart from the LDS-DMA writes in flight), which drains the ring every
    // stage.  So their completion is counted by hand too: groups of four 16-dim tiles, group n+1 in flight while group
    // n is multiplied (lgkmcnt counts in order for LDS; anything else in flight only makes the wait conservative).
#if defined(BMA_LONG_STAMPS) && BMA_LONG_STAMPS >= 2
    asm volatile("" ::"v"(pf[0].x), "v"(pf[0].w));
    const unsigned long long tc2 = __builtin_amdgcn_s_memtime();
#endif
    BMA_LONG_SB1;
    constexpr int NG = NT / 4;
    const uint32_t va = bma::lds_addr(sb) + vbase;
    u32x2 lo[2][4], hi[2][4];
    auto read_group = [&](int gi, u32x2(&L)[4], u32x2(&H)[4]) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int dt = 4 * gi + j;
        const uint32_t ad = va + 32 * ((dt & 7) ^ v7);
        if (dt >> 3) {
          L[j] = tr_read<256>(ad);
          H[j] = tr_read<256 + 16 * ROWB>(ad);
        } else {
          L[j] = tr_read<0>(ad);
          H[j] = tr_read<16 * ROWB>(ad);
        }
      }
    };
    read_group(0, lo[0], hi[0]);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      if (gi + 1 < NG) {
        read_group(gi + 1, lo[(gi + 1) & 1], hi[(gi + 1) & 1]);
        wait_lgkm<8>(lo[gi & 1], hi[gi & 1]);
      } else {
        wait_lgkm<0>(lo[gi & 1], hi[gi & 1]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint4_t vf;
        vf.x = lo[gi & 1][j].x; vf.y = lo[gi & 1][j].y; vf.z = hi[gi & 1][j].x; vf.w = hi[gi & 1][j].y;
#pragma unroll
        for (int t = 0; t < QT; ++t) oacc[t][4 * gi + j] = mfma<DT>(vf, pf[t], oacc[t][4 * gi + j]);
      }
      BMA_LONG_SB2;
    }
#if defined(BMA_LONG_STAMPS) && BMA_LONG_STAMPS >= 2
    asm volatile("" ::"v"(oacc[0][NT - 1][0]));
    const unsigned long long tc3 = __builtin_amdgcn_s_memtime();
    ts_qk += tc1 - tc0; ts_sm += tc2 - tc1; ts_pv += tc3 - tc2;
#endif
  };

  // ---- the ring: two PAIRS of 32-key stages; one pair in flight while the other is multiplied -----------------------
  // One barrier per 64 keys: with one per 32 the waves of a workgroup (different causal masks, different SIMD partners)
  // lost a quarter of every stage waiting for the slowest of the eight.
  int sb = 0;                                              // first slot of the pair being multiplied
  issue(cur, 0, 0, own_k(cur), own_v(cur));
  if (cur.chunks > 1) issue(cur, 1, 1, own_k(cur) + kstep, own_v(cur) + vstep);
  load_q(cur);
  const int64_t o_rs = static_cast<int64_t>(a.H) * DH;
  for (;;) {
    const int jn = decode(j + wgx, nxt);
    const bool has_next = jn < n_items;
    const int h = cur.hk * rep + cur.hsel * REP + hs;
    qb = cur.q0 + QW * pw;
    const bool wave_live = qb < cur.ln;
    const int last_key = P + cur.p0 + (qb + QW - 1 < cur.ln ? qb + QW - 1 : cur.ln - 1);
    const int mine = wave_live ? (last_key >> 5) + 1 : 0;  // stages 0 .. mine-1 are multiplied by this wave (<= chunks)
    const int chunks = cur.chunks;
    kfetch = own_k(cur) + 2 * kstep;
    vfetch = own_v(cur) + 2 * vstep;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      mrun[t] = NEG;
      lsum[t] = 0.0f;
      jq[t] = cur.p0 + qb + 16 * t + r;                      // this lane's query position behind the prefix
#pragma unroll
      for (int dt = 0; dt < NT; ++dt) oacc[t][dt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    auto plan = [&](int c) {                                // c = first stage of the pair about to be multiplied
      pf_done = 0;
      pf_slot = sb ^ 2;
      if (c + 2 < chunks) {
        pf_kind = 1; pf_c0 = c + 2; pf_n = c + 3 < chunks ? 2 : 1;
        pf_k = kfetch; pf_v = vfetch;
        kfetch += 2 * kstep; vfetch += 2 * vstep;
      } else if (has_next) {
        pf_kind = 2; pf_c0 = 0; pf_n = nxt.chunks > 1 ? 2 : 1;
        pf_k = own_k(nxt); pf_v = own_v(nxt);
      } else {
        pf_kind = 0; pf_n = 0;
      }
    };
    auto advance = [&](int c) {
      pf_issue(0);
      pf_issue(1);
#ifdef BMA_LONG_STAMPS
      const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
#endif
      wait_vm<0>();
      __builtin_amdgcn_s_barrier();                       // everybody's pieces of the pair landed; the other pair is free
#ifdef BMA_LONG_STAMPS
      ts_wait += __builtin_amdgcn_s_memtime() - tw0;
#endif
      plan(c);
    };
    // ---- start of the item: its first pair and its Q rows were requested a pair ago (or in front of the loop) ---------
    // The Q rows are waited for HERE, where the compiler can see it: its own wait-count bookkeeping does not know the
    // waits of the loop (inline asm), and a register load still pending at a loop header in its books costs an
    // s_waitcnt vmcnt(0) -- the ring drained -- in front of every stage.  At this point nothing else is in flight.
    {
#ifdef BMA_LONG_STAMPS
      const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
#endif
      wait_vm<0>();
#pragma unroll
      for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          asm volatile("" ::"v"(qf[t][ks].x), "v"(qf[t][ks].y), "v"(qf[t][ks].z), "v"(qf[t][ks].w));
      __builtin_amdgcn_s_barrier();
#ifdef BMA_LONG_STAMPS
      ts_wait += __builtin_amdgcn_s_memtime() - tw0;
#endif
      plan(0);
    }
    // The stages this wave multiplies come first, in a loop of their own: with the causal skip as a branch inside one
    // loop the accumulators are merged values of that branch and the compiler keeps two copies of them, moving all 64
    // registers twice per stage.  The second loop only keeps the ring and the barriers going for the other waves.
    int c = 0;
    for (; c < mine; ++c) {                                // one call site of the stage code: fewer live registers
      if (c && !(c & 1)) {
        sb ^= 2;
        advance(c);
      }
      compute(c, lds + (sb + (c & 1)) * STAGEB);
    }
    c = (c + 1) & ~1;
    for (c = c < 2 ? 2 : c; c < chunks; c += 2) {
      sb ^= 2;
      advance(c);
    }
    pf_issue(0);                                           // (what of the next item's first pair is not requested yet)
    pf_issue(1);
    if (has_next) load_q(nxt);                             // the Q fragments are dead: the next item's, behind its first pair

    // ---- epilogue of the item -----------------------------------------------------------------------------------------
#ifdef BMA_LONG_STAMPS
    const unsigned long long te0 = __builtin_amdgcn_s_memtime();
#endif
    // (a raw barrier: __syncthreads() is a fence and would drain the next item's DMA and Q loads, which are in flight)
    __builtin_amdgcn_s_barrier();                         // every wave is done reading the pair the rows are parked in
    unsigned char* park = lds + sb * STAGEB + wave * QW * ROWB;   // this wave's rows
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      const float l = rows_sum(lsum[t]);
      const int qi = qb + 16 * t + r;
      const int64_t row = cur.st + (qi < cur.ln ? qi : cur.ln - 1);
      const float inv = l > 0.0f ? 1.0f / l : 0.0f;
      float w1 = 0.0f;
      const uint16_t* o1p = nullptr;
      if (a.o1) {
        // natural-log LSE of this part; weight of the prefix partial = 1 / (1 + exp(lse2 - lse1))
        const float lse2 = l > 0.0f ? (mrun[t] * a.scale_log2e + __builtin_amdgcn_logf(l)) * 0.6931471805599453f : NEG;
        const float l1 = a.lse1[static_cast<int64_t>(h) * a.N + row];
        w1 = 1.0f / (1.0f + expf(lse2 - l1));
        o1p = a.o1 + row * o_rs + static_cast<int64_t>(h) * DH + 4 * g;
      }
#pragma unroll
      for (int dt = 0; dt < NT; ++dt) {
        float o[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) o[rr] = oacc[t][dt][rr] * inv;
        if (o1p) {
          const bma::uint2_t pw2 = *reinterpret_cast<const bma::uint2_t*>(o1p + 16 * dt);
          const float p1[4] = {bma::unpack16<DT>(pw2.x, 0), bma::unpack16<DT>(pw2.x, 1), bma::unpack16<DT>(pw2.y, 0),
                               bma::unpack16<DT>(pw2.y, 1)};
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) o[rr] = o[rr] + w1 * (p1[rr] - o[rr]);
        }
        bma::uint2_t ow;
        ow.x = pack2<DT>(o[0], o[1]);
        ow.y = pack2<DT>(o[2], o[3]);
        // dims 16dt + 4g .. +3 of row 16t + r: 16-byte piece 2*(dt & 7) + (g >> 1) of half dt >> 3, pieces XORed with r
        *reinterpret_cast<bma::uint2_t*>(park + (16 * t + r) * ROWB + 256 * (dt >> 3) +
                                         16 * ((2 * (dt & 7) + (g >> 1)) ^ r) + 8 * (g & 1)) = ow;
      }
    }
    // the wave reads back what the wave wrote: LDS operations of one wave complete in order, no barrier
    // (read by hand like the fragments: a plain LDS load here waits, in the compiler's books, for the DMA in flight)
    {
      constexpr int NPS = QW / RPI;                          // row groups of this wave
      constexpr int NB = NPS >= 8 ? 8 : 4;                   // ... read back NB at a time
      static_assert(NPS % NB == 0, "row groups do not divide into batches");
      const uint32_t pa = bma::lds_addr(park);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the parking stores above
#pragma unroll
      for (int pb = 0; pb < NPS; pb += NB) {
        u32x4 rows[NB];
#pragma unroll
        for (int ps = 0; ps < NB; ++ps) {
          const int rl = (pb + ps) * RPI + lrow;
          rows[ps] = row_read<0>(pa + rl * ROWB + 256 * half + 16 * (pos ^ (rl & 15)));
        }
        wait_rows<0, NB>(rows);
#pragma unroll
        for (int ps = 0; ps < NB; ++ps) {
          const int qrow = qb + (pb + ps) * RPI + lrow;
          if (qrow < cur.ln)
            *reinterpret_cast<uint4_t*>(a.out + (cur.st + qrow) * o_rs + static_cast<int64_t>(h) * DH + 128 * half + 8 * pos) =
                __builtin_bit_cast(uint4_t, rows[ps]);
        }
      }
    }
#ifdef BMA_LONG_STAMPS
    ts_epi += __builtin_amdgcn_s_memtime() - te0;
    n_it += 1; n_mine += mine; n_walk += chunks;
#endif
    if (!has_next) break;
    cur = nxt;
    j = jn;
    sb ^= 2;                                               // the next item's first pair went to the other half of the ring
  }
#ifdef BMA_LONG_STAMPS
  if (a.stamps && lane == 0) {
    unsigned long long* o = a.stamps + (static_cast<int64_t>(blockIdx.x) * NW + wave) * 12;
    o[0] = ts0; o[1] = __builtin_amdgcn_s_memtime(); o[2] = ts_wait; o[3] = ts_qk; o[4] = ts_sm; o[5] = ts_pv;
    o[6] = tr0; o[7] = __builtin_amdgcn_s_memrealtime(); o[8] = ts_epi; o[9] = n_it; o[10] = n_mine; o[11] = n_walk;
  }
#endif
}

// Which blocks take the long-block kernel: set by bma_ragged_attention_set_long (measurement only; the library reads no
// environment).  mode 0 = never, 1 = when the blocks are long enough (default); min_len = shortest max_len that takes it
// (with 16 every ragged test passes through it, and the LLaVA row list / padded blocks take 335 / 261 us instead of
// 156 / 167 -- items of three stages are all start and end).
int g_long_mode = 1, g_long_min = kLongMin;
int long_blocks_mode() { return g_long_mode; }
int long_min() { return g_long_min; }

template <int DT, int DH, int REP, int QT, int TW>
int launch_long_cfg(Args b, int max_len, hipStream_t st) {
  constexpr int QB = 16 * QT * TW;
  b.nz = (max_len + QB - 1) / QB;
  const int rep = b.H / b.Hk;
  const int64_t groups = static_cast<int64_t>(b.B2) * b.Hk, sharers = static_cast<int64_t>(rep / REP) * b.nz;
  if ((groups + 7) / 8 * sharers > 0x7fffffffLL) return BMA_ELIMIT;    // items per XCD list fit an int
  // persistent: one workgroup per CU (the LDS ring admits no second), a multiple of 8 so that every XCD has as many
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return (n + 7) / 8 * 8;
  }();
  const int64_t blocks = cus;
  const dim3 grid(static_cast<unsigned>(blocks));
#ifdef BMA_LONG_STAMPS
  // diagnostic build: BMA_RAGGED_STAMPS=<file> gets 12 x u64 per wave of every launch (overwritten per launch)
  const char* dump = getenv("BMA_RAGGED_STAMPS");
  if (dump && *dump) {
    (void)hipMalloc(reinterpret_cast<void**>(&b.stamps), blocks * TW * REP * 96);
    (void)hipMemsetAsync(b.stamps, 0, blocks * TW * REP * 96, st);
  }
#endif
  hipLaunchKernelGGL((ragged_attn_long_kernel<DT, DH, REP, QT, TW>), grid, dim3(64 * TW * REP), 0, st, b);
#ifdef BMA_LONG_STAMPS
  if (b.stamps) {
    std::vector<unsigned long long> host(blocks * TW * REP * 12);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(host.data(), b.stamps, blocks * TW * REP * 96, hipMemcpyDeviceToHost);
    (void)hipFree(b.stamps);
    if (FILE* f = fopen(dump, "wb")) {
      fwrite(host.data(), 8, host.size(), f);
      fclose(f);
    }
  }
#endif
  return BMA_OK;
}

template <int DT, int DH>
int launch_long(const Args& a, int max_len, hipStream_t st) {
  // 256-wide heads: one 16-query tile per wave (two would need 192 registers of accumulators and Q fragments alone,
  // and two waves per SIMD have 256 each); 128-wide heads: two tiles per wave.  Four tile-waves per head: with six
  // (12 waves, three per SIMD, 168 registers) a stage took 5.5k cycles instead of 3.3k -- the SIMDs' issue slots, not
  // latency, bound the stage -- and Gemma-3 blocks ran 348 us instead of 255.
  // The opposite corner -- ONE wave per SIMD with two tiles each (4 waves: 64 queries of two heads, or 128 of one; the
  // structure of the guide's fastest forward) -- compiles to 498 registers without scratch and runs 1288 / 1744 us:
  // past 256 the compiler parks vector registers in accumulator registers and copies them back around every use.
  // That design needs its register allocation done by hand.
  constexpr int QT = 1;
  const int rep = a.H / a.Hk;
  if (rep % 2 == 0) {
    return launch_long_cfg<DT, DH, 2, QT, 4>(a, max_len, st);
  }
  return launch_long_cfg<DT, DH, 1, QT, 4>(a, max_len, st);
}

template <int DT, int DH>
int launch_dh(const Args& a, int max_len, hipStream_t st) {
  if constexpr (DH == 128 || DH == 256) {
    // blocks of at least 96 tokens fill three quarters of a 128-query workgroup: the long-block kernel
    // (its row lookups multiply a row index by a 32-bit row stride in bytes)
    const bool strides32 = a.k_rs < (1 << 30) && a.v_rs < (1 << 30) && a.pk_rs < (1 << 30) && a.pv_rs < (1 << 30) &&
                           a.k_rs >= 0 && a.v_rs >= 0 && a.pk_rs >= 0 && a.pv_rs >= 0;
    if (max_len >= long_min() && a.scale_log2e > 0.0f && strides32 && long_blocks_mode() != 0) return launch_long<DT, DH>(a, max_len, st);
  }
  const int qt = max_len >= 64 ? 4 : (max_len + 15) / 16;
  Args b = a;
  b.nz = (max_len + 16 * qt - 1) / (16 * qt);
  const int64_t groups = static_cast<int64_t>(a.B2) * a.Hk, sharers = static_cast<int64_t>(a.H / a.Hk) * b.nz;
  const int64_t blocks = (groups + 7) / 8 * sharers * 8;
  if (blocks > 0x7fffffffLL) return BMA_ELIMIT;
  const dim3 grid(static_cast<unsigned>(blocks));
  switch (qt) {
    case 1: hipLaunchKernelGGL((ragged_attn_kernel<DT, 1, DH>), grid, dim3(64), 0, st, b); break;
    case 2: hipLaunchKernelGGL((ragged_attn_kernel<DT, 2, DH>), grid, dim3(128), 0, st, b); break;
    case 3: hipLaunchKernelGGL((ragged_attn_kernel<DT, 3, DH>), grid, dim3(192), 0, st, b); break;
    default: hipLaunchKernelGGL((ragged_attn_kernel<DT, 4, DH>), grid, dim3(256), 0, st, b); break;
  }
  return BMA_OK;
}

template <int DT>
int launch_dt(const Args& a, int max_len, int Dh, hipStream_t st) {
  if (Dh == 32) return launch_dh<DT, 32>(a, max_len, st);
  if (Dh == 64) return launch_dh<DT, 64>(a, max_len, st);
  if (Dh == 128) return launch_dh<DT, 128>(a, max_len, st);
  return launch_dh<DT, 256>(a, max_len, st);
}

}  // namespace

extern "C" void bma_ragged_attention_set_long(int mode, int min_len) {
  g_long_mode = mode != 0 ? 1 : 0;
  g_long_min = min_len > 0 ? min_len : kLongMin;
}

extern "C" int bma_ragged_attention(const void* q, int64_t q_rs, int64_t q_hs, const void* k, int64_t k_rs,
                                    int64_t k_hs, const void* v, int64_t v_rs, int64_t v_hs, const void* pk,
                                    int64_t pk_rs, int64_t pk_hs, const void* pv, int64_t pv_rs, int64_t pv_hs,
                                    int P, const int* start, const int* first, const int* len, int B2,
                                    int max_len, int64_t N, int H, int Hk, int Dh, int dtype, float scale,
                                    const void* o1, const float* lse1, void* out, void* stream) {
  if (B2 < 0 || N < 0 || H <= 0 || Hk <= 0 || P < 0 || max_len <= 0) return BMA_EINVAL;
  if (B2 == 0 || N == 0) return BMA_OK;
  if (!q || !k || !v || !start || !first || !len || !out) return BMA_EINVAL;
  if (P > 0 && (!pk || !pv)) return BMA_EINVAL;
  if ((o1 == nullptr) != (lse1 == nullptr)) return BMA_EINVAL;
  if (dtype != BMA_BF16 && dtype != BMA_F16) return BMA_EDTYPE;
  if ((Dh != 32 && Dh != 64 && Dh != 128 && Dh != 256) || H % Hk || max_len > 4096 || N > 0x7fffffffLL) return BMA_ELIMIT;
  // 16-byte operand loads: every row/head stride a multiple of 8 elements, bases 16-byte aligned
  const int64_t strides[] = {q_rs, q_hs, k_rs, k_hs, v_rs, v_hs, pk_rs, pk_hs, pv_rs, pv_hs};
  for (int64_t s : strides)
    if (s % 8) return BMA_EALIGN;
  if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) |
       reinterpret_cast<uintptr_t>(pk) | reinterpret_cast<uintptr_t>(pv) | reinterpret_cast<uintptr_t>(o1) |
       reinterpret_cast<uintptr_t>(out)) % 16)
    return BMA_EALIGN;
  Args a;
  a.q = static_cast<const uint16_t*>(q); a.k = static_cast<const uint16_t*>(k); a.v = static_cast<const uint16_t*>(v);
  a.pk = static_cast<const uint16_t*>(pk); a.pv = static_cast<const uint16_t*>(pv);
  a.o1 = static_cast<const uint16_t*>(o1); a.lse1 = lse1; a.out = static_cast<uint16_t*>(out);
  a.start = start; a.first = first; a.len = len;
  a.q_rs = q_rs; a.q_hs = q_hs; a.k_rs = k_rs; a.k_hs = k_hs; a.v_rs = v_rs; a.v_hs = v_hs;
  a.pk_rs = pk_rs; a.pk_hs = pk_hs; a.pv_rs = pv_rs; a.pv_hs = pv_hs;
  a.B2 = B2; a.H = H; a.Hk = Hk; a.P = P; a.N = static_cast<int>(N);
  a.scale_log2e = scale * 1.4426950408889634f;
  a.stamps = nullptr;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int es = 2;
  BMA_PROF_BEGIN(BMA_K_RAGGED_ATTN, st, (2.0 * H + 2.0 * Hk) * static_cast<double>(N) * Dh * es);
  const int rc = dtype == BMA_BF16 ? launch_dt<BMA_BF16>(a, max_len, Dh, st) : launch_dt<BMA_F16>(a, max_len, Dh, st);
  BMA_PROF_END(BMA_K_RAGGED_ATTN, st);
  if (rc != BMA_OK) return rc;
  BMA_LAUNCH_CHECK();
  return BMA_OK;
}
