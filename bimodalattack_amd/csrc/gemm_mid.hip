// a1 (gradient pass) -- the products of the batch-1 pass when the image sits in the prompt:  y[M][N] = x[M][K] . W[N][K]^T
// for a few hundred rows.
//
// With PGD on, the reference's compute_gradient (bimodal_attack.py:953-1028) runs the language model over 576 image rows
// plus the prompt: 599-644 rows at batch 1.  Every linear layer is then a product of ~640 activation rows with a
// 34-180 MB weight -- too tall for the weight-streaming kernel of gemm_nt.hip (one 96-row tile), too short for the
// library's 256 x 256 tiles: 644 rows are 2.5 of them, N = 4096 gives 16 columns of tiles, and the tuned hipBLASLt
// kernels run these shapes at 0.20-0.36 of the dense MFMA peak (tools/gemm_bench.py --mid).
//
// Design (gfx950 only):
//   * tiles of 224 rows (MF = 7 fragments of 16 per row half: 644 rows = three row tiles, 96 % full) by 64*NF columns,
//     NF = 3 or 4; one workgroup per CU (the operand rings fill the LDS), 8 waves as 2 (row halves) x 4 (column
//     quarters), two per SIMD, each with a 112 x 16 NF accumulator tile;
//   * both operands go L2 -> LDS by `buffer_load_dwordx4 ... offen lds` in 1-KiB pieces of 8 rows x 128 B (whole cache
//     lines), per operand its own ring of K = 64 units: two slots for x (L2-resident: every column tile re-reads it),
//     three for w (HBM).  The descriptor holds the tile's first row, the per-lane offset never changes, the unit of K is
//     the scalar offset: a piece costs its wave one M0 write and the load (round 6; `global_load_lds` with 64-bit per-lane
//     addresses before that: k loop 2720 -> 2430 cycles per unit);
//   * LDS image: 128-byte rows, the 16-byte chunk c of row r stored at position c ^ (r & 7) -- applied to the DMA's SOURCE
//     offset and to the fragment read -- so every `ds_read_b128` of an MFMA fragment is conflict-free (SQ_LDS_BANK_CONFLICT 0);
//   * the k loop (round 6): every wave runs the whole unit on its own -- four sub-steps of 16 / 12 MFMAs, the fragments
//     of sub-step s+1 travelling LDS -> registers while sub-step s multiplies -- with ONE workgroup barrier per unit; the
//     two waves of a SIMD keep its MFMA pipe fed between them.  (Rounds 4-5 alternated a load phase and an MFMA phase
//     between the row halves, two barriers per phase; round 6 first cut those phases four times finer.  All three
//     schedules ran the same product in the same time -- what they shared was the cost of a DMA piece and the chip's
//     clock under LDS reads: DESIGN.md 5d.)
//   * workgroup ids are remapped so that an XCD owns a contiguous run of (column tile, row tile) pairs, row tile
//     fastest: the three row tiles that read the same weight rows and the ~10 column tiles that read the same
//     activation rows share the XCD's L2, and the weight leaves HBM once;
//   * the grid is made to fit the 256 CUs: K split S ways for every tile when there are few (N = 4096: 48 tiles x 5), or
//     -- when the tiles just overflow one round (N = 22016: 258) -- the last column of tiles alone split 8 ways, run in
//     the shadow of the first round.  Split tiles write fp32 partials in accumulator order (1 KiB per wave
//     instruction); a second launch sums them in split order and converts -- no tickets, no spinning, bitwise
//     reproducible.
//
// Algorithmic bytes per launch: (M*K + N*K + M*N) * es; FLOPs 2*M*N*K.
//
// Diagnostic builds (never shipped): -DBMA_MID_STAMPS (tools/mid_stamps.py: s_memtime / s_memrealtime stamps of workgroups
// 0, 100, 200 and the last one, kept in 8 KiB of LDS behind the rings and copied out at the end) and -DBMA_MID_ABLATE=<bits>
// (tools/mid_ablate.sh: the loop without its DMA pieces / fragment reads / MFMAs -- WRONG results, timing only).

#include <type_traits>
#ifdef BMA_MID_STAMPS
#include <cstdio>
#include <cstdlib>
#include <vector>
#endif

#include "bma_common.h"
#include "bma_profile.h"

namespace {

using bma::uint4_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kNW = 8;            // waves per workgroup: 2 (row halves) x 4 (column quarters), two per SIMD
constexpr int kBK = 64;           // k per ring unit: 128-byte rows
constexpr int kRowB = kBK * 2;    // bytes per LDS row (16-bit types only)
constexpr int kLds = 160 * 1024;
constexpr int kNA = 2;            // ring slots of x
constexpr int kNB = 3;            // ring slots of w
#ifdef BMA_MID_STAMPS
constexpr int kStampN = 128;      // stamps per wave
#endif
#ifndef BMA_MID_ABLATE
#define BMA_MID_ABLATE 0          // 1 = no DMA pieces inside the loop, 2 = no fragment reads inside the loop, 4 = no MFMAs
#endif
constexpr int kAblate = BMA_MID_ABLATE;

struct MidArgs {
  const char* x;
  const char* w;
  char* y;
  float* ws;
  int64_t ldx, ldw, ldy;   // elements
  int M, N, K, m_tiles, n_tiles;
  int t_full;              // tiles [0, t_full) run over all of K; the others are split S ways
  int S;
  int xcd;                 // 1: remap workgroup ids so that an XCD owns a contiguous run of tiles
  int stamp_units;         // diagnostic builds: 0 = only the loop's start / end stamps (the clock), 1 = per-unit stamps too
  unsigned long long* stamps;
};

template <int DT>
__device__ __forceinline__ f32x4 mfma16(const uint4_t& a, const uint4_t& b, const f32x4& c) {
  if (DT == BMA_BF16)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// one LDS-DMA piece: 64 lanes x 16 B from descriptor base + per-lane offset + scalar offset to lds_dst + lane * 16.  (A device
// function of its own: inside the kernel TEMPLATE the host pass drops the whole instantiation -- stub included -- over this builtin.)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned char* lds_dst, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N <= 63, "vmcnt is six bits");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// MF / NF: 16-row fragments of x / of w per wave (tile = 32 MF rows of x by 64 NF rows of w)
template <int DT, int MF, int NF>
__global__ __launch_bounds__(kNW * 64) void gemm_mid_kernel(MidArgs a) {
  constexpr int BM = 32 * MF, BN = 64 * NF;
  constexpr int ASZ = BM * kRowB, BSZ = BN * kRowB;              // bytes of a unit of x / of w
  constexpr int PA = BM / 8, PB = BN / 8;                        // 1-KiB pieces (8 rows x 128 B) per unit
  constexpr int CA = (PA + kNW - 1) / kNW, CB = PB / kNW;        // pieces per wave and unit: x 4 (waves 4-7: 3), w 4 or 3
  constexpr int H0 = (MF + 1) / 2, H1 = MF - H0;                 // row fragments of the two halves of a k step
  static_assert(PB % kNW == 0 && CA == 4 && CB >= 2 && CB <= 4, "piece schedule");
  static_assert(kNA * ASZ + kNB * BSZ <= kLds, "rings beyond the LDS");
#ifdef BMA_MID_STAMPS
  static_assert(kNA * ASZ + kNB * BSZ + 8192 <= kLds, "no room for stamps");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[kNA * ASZ + kNB * BSZ + 8192];
  unsigned long long* const stamp_base = reinterpret_cast<unsigned long long*>(lds + kNA * ASZ + kNB * BSZ);
  int stamp_i = 0;
  const int stamp_wg = blockIdx.x == gridDim.x - 1 ? 3 : (blockIdx.x == 0 ? 0 : (blockIdx.x == 100 ? 1 : (blockIdx.x == 200 ? 2 : -1)));
  if (a.stamps && stamp_wg >= 0 && (threadIdx.x & 63) == 0) stamp_base[(threadIdx.x >> 6) * kStampN + 122] = __builtin_amdgcn_s_memrealtime();   // kernel entry
  // lane 0 writes with ds_write (counted by lgkmcnt, so the DMA's vmcnt arithmetic is untouched)
#define BMA_MID_STAMP()                                                                                         \
  do {                                                                                                          \
    if (stamp_on) {                                                                                             \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                               \
      if ((threadIdx.x & 63) == 0 && stamp_i < 120) stamp_base[(threadIdx.x >> 6) * kStampN + stamp_i] = t_;    \
      ++stamp_i;                                                                                                \
    }                                                                                                           \
  } while (0)
#define BMA_MID_COPY_STAMPS()                                                                                   \
  do {                                                                                                          \
    if (a.stamps && stamp_wg >= 0) {                                                                            \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* the stores have left */                              \
      if ((threadIdx.x & 63) == 0) stamp_base[(threadIdx.x >> 6) * kStampN + 123] = __builtin_amdgcn_s_memrealtime(); \
      __syncthreads();                                                                                          \
      for (int i_ = threadIdx.x; i_ < kNW * kStampN; i_ += kNW * 64) a.stamps[stamp_wg * kNW * kStampN + i_] = stamp_base[i_]; \
    }                                                                                                           \
  } while (0)
#else
  __shared__ __attribute__((aligned(1024))) unsigned char lds[kNA * ASZ + kNB * BSZ];
#define BMA_MID_STAMP() do {} while (0)
#define BMA_MID_COPY_STAMPS() do {} while (0)
#endif
  unsigned char* const lds_b = lds + kNA * ASZ;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  // Work items: [0, t_full) whole tiles, then the split pieces.  When the grid overflows one round of 256 CUs, the blocks
  // past the first round (dispatched as CUs free up) take split pieces -- short ones -- and never a whole tile.
  const int nwg = static_cast<int>(gridDim.x);
  const int late = nwg > 256 && a.t_full <= 256 ? nwg - 256 : 0;   // blocks of the second round
  int v = blockIdx.x;
  if (v >= nwg - late) {
    v = a.t_full + (v - (nwg - late));                           // split pieces [0, late)
  } else {
    const int first = nwg - late;
    if (a.xcd) {                                                 // bijective for any count (cdna_hip_programming.md 5, "XCD swizzle")
      const int q = first >> 3, r = first & 7, xcd = v & 7;
      v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
    }
    if (v >= a.t_full) v += late;                                // split pieces [late, ...)
  }
  const int U = a.K / kBK;
  int tile, split, u0 = 0, u1 = U;
  if (v < a.t_full) {
    tile = v;
    split = -1;
  } else {
    const int t_tail = a.m_tiles * a.n_tiles - a.t_full;
    tile = a.t_full + (v - a.t_full) % t_tail;
    split = (v - a.t_full) / t_tail;
    u0 = static_cast<int>(static_cast<int64_t>(U) * split / a.S);
    u1 = static_cast<int>(static_cast<int64_t>(U) * (split + 1) / a.S);
    if (a.S == 1) split = -1;
  }
  const int m_tile = tile % a.m_tiles, n_tile = tile / a.m_tiles;
  const int m0 = m_tile * BM, n0 = n_tile * BN;

  // ---- DMA pieces: piece p of an operand = rows 8p..8p+7 of the tile, dealt to wave p % 8: every wave issues for both ----
  const int prow = lane >> 3;                                    // row inside the piece == (tile row & 7)
  const int pchunk = (lane & 7) ^ prow;                          // source chunk that lands at LDS position lane & 7
  int vox[CA], vow[CB];
#pragma unroll
  for (int i = 0; i < CA; ++i) {
    int m = m0 + (wave + kNW * i) * 8 + prow;
    m = m < a.M ? m : a.M - 1;                                   // rows past M repeat the last one (never stored)
    vox[i] = static_cast<int>((m - m0) * a.ldx * 2) + pchunk * 16;
  }
#pragma unroll
  for (int i = 0; i < CB; ++i) {
    int n = n0 + (wave + kNW * i) * 8 + prow;
    n = n < a.N ? n : a.N - 1;
    vow[i] = static_cast<int>((n - n0) * a.ldw * 2) + pchunk * 16;
  }
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(a.x) + static_cast<int64_t>(m0) * a.ldx * 2, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(a.w) + static_cast<int64_t>(n0) * a.ldw * 2, 0, 0x7fffffff, 0x00020000);
  bool in_loop = false;
  // piece i of this wave, unit u -> ring slot; wave-uniform destination, the DMA adds lane*16 (the last x piece exists for
  // waves 0-3 only: 28 pieces over 8 waves)
#define BMA_MID_X(i_, u_, slot_)                                                                                \
  do {                                                                                                          \
    if ((kNW * (i_) + kNW <= PA || wave + kNW * (i_) < PA) && !(in_loop && (kAblate & 1)))                      \
      dma16(rx, lds + (slot_) * ASZ + (wave + kNW * (i_)) * 1024, vox[i_], (u_) * kRowB);                       \
  } while (0)
#define BMA_MID_W(i_, u_, slot_)                                                                                \
  do {                                                                                                          \
    if ((i_) < CB && !(in_loop && (kAblate & 1)))                                                               \
      dma16(rw, lds_b + (slot_) * BSZ + (wave + kNW * (i_)) * 1024, vow[(i_) < CB ? (i_) : 0], (u_) * kRowB);   \
  } while (0)

  f32x4 acc[NF][MF];
#pragma unroll
  for (int j = 0; j < NF; ++j)
#pragma unroll
    for (int i = 0; i < MF; ++i) acc[j][i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  // fragment read: row (lane & 15) of a 16-row fragment (2 KiB), chunk (4*kk + (lane >> 4)) ^ (row & 7)
  const int frow = lane & 15, fg = lane >> 4;
  const int foff0 = frow * kRowB + ((fg) ^ (frow & 7)) * 16;          // kk = 0
  const int foff1 = frow * kRowB + ((4 + fg) ^ (frow & 7)) * 16;      // kk = 1
  const int offa = (wr * MF) * 2048, offb = (wc * NF) * 2048;
  const int n_units = u1 - u0;

  // ---- the k loop: every wave on its own, MFMAs of sub-step s while the fragments of sub-step s+1 travel LDS -> registers ----
  // A unit of K is four sub-steps: (k 0..31, row fragments [0,H0)), (k 0..31, [H0,MF)), (k 32..63, [0,H0)), (k 32..63, [H0,MF)).
  // ONE barrier per unit, B_k, behind sub-step 2: in front of it every read of unit k has landed in registers (the
  // fragments of sub-step 3 were asked for a sub-step earlier) and this wave's pieces of unit k+1 have landed in the LDS,
  // so behind it (a) unit k's slots are free: x of unit k+2 goes out in sub-step 3, w of unit k+2 -- into the slot unit
  // k-1 left at B_{k-1} -- went out in sub-steps 0 and 1; (b) sub-step 3 reads the first fragments of unit k+1.
  // vmcnt at B_k: issue order is w(k+1) [unit k-1], x(k+1) [unit k-1, sub-step 3], w(k+2) [unit k]: all but the CB
  // youngest.  Measured (tools/mid_stamps.py, profiles/r6_gemm_mid_stamps.txt): a unit takes 2430 cycles against the 1792
  // its 2 x 56 MFMAs occupy the SIMD's pipe -- the same 74 % the library's 256 x 256 kernel reaches on a 17k-row product;
  // vmcnt and lgkmcnt waits are nil; the older wave of a SIMD reaches B_k ~700 cycles before the younger whichever half is
  // given priority or the DMA issue (the younger one's finish IS the unit).
  {
    uint4_t wf0[NF], wf1[NF], xfA[H0], xfB[H0];
#define BMA_MID_RDF(dst_, base_, n_, i0_, foff_)                                                    \
  if (!(kAblate & 2) || !in_loop) _Pragma("unroll") for (int i = 0; i < (n_); ++i)                  \
    dst_[i] = *reinterpret_cast<const uint4_t*>((base_) + ((i0_) + i) * 2048 + (foff_))
#define BMA_MID_MM(wf_, xf_, i0_, n_)                                                               \
  _Pragma("unroll") for (int i = 0; i < (n_); ++i)                                                  \
    _Pragma("unroll") for (int j = 0; j < NF; ++j)                                                  \
      if (kAblate & 4) asm volatile("" ::"v"(wf_[j]), "v"(xf_[i]));                                 \
      else acc[j][(i0_) + i] = mfma16<DT>(wf_[j], xf_[i], acc[j][(i0_) + i])
    // prologue: x and w of unit 0 land; w and x of unit 1 stay in flight
    BMA_MID_X(0, u0, 0); BMA_MID_X(1, u0, 0); BMA_MID_X(2, u0, 0); BMA_MID_X(3, u0, 0);
    BMA_MID_W(0, u0, 0); BMA_MID_W(1, u0, 0); BMA_MID_W(2, u0, 0); BMA_MID_W(3, u0, 0);
    if (n_units > 1) {
      BMA_MID_W(0, u0 + 1, 1); BMA_MID_W(1, u0 + 1, 1); BMA_MID_W(2, u0 + 1, 1); BMA_MID_W(3, u0 + 1, 1);
      BMA_MID_X(0, u0 + 1, 1); BMA_MID_X(1, u0 + 1, 1); BMA_MID_X(2, u0 + 1, 1); BMA_MID_X(3, u0 + 1, 1);
      if (wave + kNW * 3 < PA) wait_vm<CB + 4>();                // (waves 0-3 issue four x pieces, waves 4-7 three)
      else wait_vm<CB + 3>();
    } else {
      wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
    BMA_MID_RDF(wf0, lds_b + offb, NF, 0, foff0);
    BMA_MID_RDF(xfA, lds + offa, H0, 0, foff0);
    int sa = 0, sb = 0;
    in_loop = true;
    auto one_unit = [&](const bool more1, const bool more2, const int k) __attribute__((always_inline)) {
      const int u2 = u0 + k + 2;
      const int sb1 = sb + 1 == kNB ? 0 : sb + 1;                // w slot of unit k+1
      const int sw = sb == 0 ? kNB - 1 : sb - 1;                 // w slot of unit k-1 == of unit k+2
      const unsigned char* pa = lds + sa * ASZ + offa;
      const unsigned char* pb = lds_b + sb * BSZ + offb;
      // sub-step 0
      BMA_MID_RDF(xfB, pa, H1, H0, foff0);
      BMA_MID_RDF(wf1, pb, NF, 0, foff1);
      if (more2) { BMA_MID_W(0, u2, sw); BMA_MID_W(1, u2, sw); }
      BMA_MID_MM(wf0, xfA, 0, H0);
      // sub-step 1
      BMA_MID_RDF(xfA, pa, H0, 0, foff1);
      if (more2) { BMA_MID_W(2, u2, sw); BMA_MID_W(3, u2, sw); }
      BMA_MID_MM(wf0, xfB, H0, H1);
      // sub-step 2, then B_k
      BMA_MID_RDF(xfB, pa, H1, H0, foff1);
      BMA_MID_MM(wf1, xfA, 0, H0);
#ifdef BMA_MID_STAMPS
      const bool stamp_on = a.stamps && a.stamp_units && k >= 16 && k < 16 + 120 / 4 && stamp_wg >= 0;
      __builtin_amdgcn_sched_barrier(0);
#endif
      BMA_MID_STAMP();                                           // sub-steps 0-2 issued
      if (more2) wait_vm<CB>();
      else wait_vm<0>();
      BMA_MID_STAMP();                                           // pieces of unit k+1 landed
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      BMA_MID_STAMP();                                           // every fragment read landed
      __builtin_amdgcn_s_barrier();
      BMA_MID_STAMP();                                           // through B_k
      // sub-step 3
      if (more2) { BMA_MID_X(0, u2, sa); BMA_MID_X(1, u2, sa); BMA_MID_X(2, u2, sa); BMA_MID_X(3, u2, sa); }
      if (more1) {
        BMA_MID_RDF(wf0, lds_b + sb1 * BSZ + offb, NF, 0, foff0);
        BMA_MID_RDF(xfA, lds + (sa ^ 1) * ASZ + offa, H0, 0, foff0);
      }
      BMA_MID_MM(wf1, xfB, H0, H1);
      sa ^= 1;
      sb = sb1;
    };
#ifdef BMA_MID_STAMPS
    if (a.stamps && lane == 0) {                                 // the clock the loop runs at: cycles against the 100 MHz counter
      stamp_base[wave * kStampN + 124] = __builtin_amdgcn_s_memtime();
      stamp_base[wave * kStampN + 125] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    int k = 0;
    for (; k + 2 < n_units; ++k) one_unit(true, true, k);        // the steady state carries no tail conditions
    for (; k < n_units; ++k) one_unit(k + 1 < n_units, false, k);
#ifdef BMA_MID_STAMPS
    if (a.stamps && lane == 0) {
      stamp_base[wave * kStampN + 126] = __builtin_amdgcn_s_memtime();
      stamp_base[wave * kStampN + 127] = __builtin_amdgcn_s_memrealtime();
    }
#endif
#undef BMA_MID_RDF
#undef BMA_MID_MM
  }
#undef BMA_MID_X
#undef BMA_MID_W
#undef BMA_MID_STAMP

  // ---- split tiles: the partial leaves in register order; bma_gemm_mid's second launch sums the splits ---------------------
  // (Round 6 built the alternative -- the S workgroups of a tile reduce-scattering it among themselves inside the one launch,
  // sc1 stores, a ticket per tile, bit-equal to this form -- and measured it 5-8 us SLOWER per split product: one workgroup
  // per CU has eight waves to push 224 KB write-through and pull 224 KB back behind a wait for the slowest piece, while the
  // second launch streams the same bytes from every CU at full occupancy: profiles/r6_gemm_mid_fused_reduce_ab.txt.)
  if (split >= 0) {
    f32x4* out = reinterpret_cast<f32x4*>(a.ws) +
                 ((static_cast<int64_t>(tile - a.t_full) * a.S + split) * kNW + wave) * (NF * MF * 64) + lane;
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
      for (int i = 0; i < MF; ++i) out[(j * MF + i) * 64] = acc[j][i];
    BMA_MID_COPY_STAMPS();
    return;
  }

  // ---- epilogue: a lane holds y[m][n .. n+3] of each of its tiles ---------------------------------------------------------
#pragma unroll
  for (int j = 0; j < NF; ++j) {
    const int n = n0 + (wc * NF + j) * 16 + fg * 4;
#pragma unroll
    for (int i = 0; i < MF; ++i) {
      const int row = m0 + (wr * MF + i) * 16 + frow;
      if (row < a.M && n < a.N) {
        char* dst = a.y + (static_cast<int64_t>(row) * a.ldy + n) * 2;
        const f32x4 t = acc[j][i];
        if (n + 3 < a.N) {
          bma::uint2_t o;
          o.x = bma::pack16<DT>(t.x, t.y);
          o.y = bma::pack16<DT>(t.z, t.w);
          *reinterpret_cast<bma::uint2_t*>(dst) = o;
        } else {
          const float e[4] = {t.x, t.y, t.z, t.w};
          for (int r = 0; r < 4 && n + r < a.N; ++r)
            reinterpret_cast<uint16_t*>(dst)[r] = static_cast<uint16_t>(bma::pack16<DT>(e[r], 0.0f) & 0xffffu);
        }
      }
    }
  }
  BMA_MID_COPY_STAMPS();
}
#undef BMA_MID_COPY_STAMPS

// second launch of a split product: one thread per float4 of a split tile, the S partials summed in split order
template <int DT>
__global__ __launch_bounds__(256) void gemm_mid_reduce_kernel(MidArgs a, int MF, int NF) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const int frags = MF * NF;
  const int64_t per = static_cast<int64_t>(kNW) * frags * 64;     // float4 per (tile, split)
  const int64_t tt = idx / per;                                   // index among the split tiles
  if (tt >= static_cast<int64_t>(a.m_tiles) * a.n_tiles - a.t_full) return;
  const int in = static_cast<int>(idx - tt * per);
  const int lane = in & 63, frag = (in >> 6) % frags, wave = (in >> 6) / frags;
  const int j = frag / MF, i = frag % MF;
  const f32x4* src = reinterpret_cast<const f32x4*>(a.ws) + tt * a.S * per + in;
  f32x4 t = src[0];
  for (int s = 1; s < a.S; ++s) t += src[s * per];
  const int tile = static_cast<int>(tt) + a.t_full;
  const int m_tile = tile % a.m_tiles, n_tile = tile / a.m_tiles;
  const int row = m_tile * 32 * MF + ((wave >> 2) * MF + i) * 16 + (lane & 15);
  const int n = n_tile * 64 * NF + ((wave & 3) * NF + j) * 16 + (lane >> 4) * 4;
  if (row >= a.M || n >= a.N) return;
  char* dst = a.y + (static_cast<int64_t>(row) * a.ldy + n) * 2;
  if (n + 3 < a.N) {
    bma::uint2_t o;
    o.x = bma::pack16<DT>(t.x, t.y);
    o.y = bma::pack16<DT>(t.z, t.w);
    *reinterpret_cast<bma::uint2_t*>(dst) = o;
  } else {
    const float e[4] = {t.x, t.y, t.z, t.w};
    for (int r = 0; r < 4 && n + r < a.N; ++r)
      reinterpret_cast<uint16_t*>(dst)[r] = static_cast<uint16_t>(bma::pack16<DT>(e[r], 0.0f) & 0xffffu);
  }
}

constexpr int kCUs = 256;
constexpr int kMaxSplit = 8;
constexpr int kMF = 7;
struct MidPlan {
  int mf, nf, m_tiles, n_tiles, t_full, S, xcd;
};

int g_nf = 0, g_S = 0, g_tail = -1, g_flags = -1;   // tuning override (bma_gemm_mid_set_plan): 0 / -1 = the planner's choice

// The decomposition, by tile count (measured on the seven products of a LLaVA-7B layer at 644 rows, tools/gemm_bench.py
// --mid --sweep; profiles/r4_gemm_mid_sweep.txt): the plan that moves the fewest bytes in ONE round of 256 workgroups wins:
//   * few tiles (N = 4096: 48 of 224 x 256): K split floor(256 / tiles) ways for every tile;
//   * up to one round of the wide tile: unsplit, 192-wide tiles when those still fit one round (N = 12288: 192 workgroups
//     against 144; N = 11008: 174 against 129), the 256-wide tile otherwise;
//   * just over a round (N = 22016: 258): as many whole columns of tiles as fit 256 run unsplit, the rest split 8 ways in
//     their shadow (141 us against 174 for two rounds).
bool make_plan(int M, int N, int K, MidPlan& p) {
  if (M <= 0 || N <= 0 || K <= 0 || K % kBK) return false;
  p.mf = kMF;
  p.m_tiles = (M + 32 * p.mf - 1) / (32 * p.mf);
  const int U = K / kBK;
  const int T4 = p.m_tiles * ((N + 255) / 256), T3 = p.m_tiles * ((N + 191) / 192);
  int nf = 4, S = 1, tail = 0;
  if (T4 * 2 <= kCUs) {
    S = kCUs / T4;
  } else if (T4 <= kCUs) {
    nf = T3 <= kCUs ? 3 : 4;
  } else {
    const int cols = (N + 255) / 256, full_cols = kCUs / p.m_tiles;
    tail = cols - full_cols;
    S = 8;
  }
  if (g_nf) nf = g_nf;
  if (g_S) S = g_S;
  if (g_tail >= 0) tail = g_tail;
  if (nf != 3 && nf != 4) return false;
  S = S > kMaxSplit ? kMaxSplit : S;
  S = S > U ? U : S;
  S = S < 1 ? 1 : S;
  p.nf = nf;
  p.S = S;
  p.n_tiles = (N + 64 * nf - 1) / (64 * nf);
  tail = tail > p.n_tiles ? p.n_tiles : tail;
  const int T = p.m_tiles * p.n_tiles;
  p.t_full = S == 1 ? T : (tail > 0 ? T - tail * p.m_tiles : 0);
  const int flags = g_flags >= 0 ? g_flags : 1;
  p.xcd = flags & 1;
  return true;
}

}  // namespace

extern "C" void bma_gemm_mid_set_plan(int w_frags_per_wave, int splits, int tail_columns, int flags) {
  g_nf = w_frags_per_wave; g_S = splits; g_tail = tail_columns; g_flags = flags;
}

extern "C" int bma_gemm_mid_plan(int M, int N, int K, int* out8) {
  MidPlan p;
  if (!make_plan(M, N, K, p)) return BMA_EINVAL;
  if (out8) {
    const int T = p.m_tiles * p.n_tiles;
    out8[0] = p.mf; out8[1] = p.m_tiles; out8[2] = p.nf; out8[3] = p.n_tiles; out8[4] = p.S; out8[5] = p.xcd;
    out8[6] = p.t_full + (T - p.t_full) * p.S; out8[7] = p.t_full;
  }
  return BMA_OK;
}

extern "C" size_t bma_gemm_mid_ws_bytes(int M, int N, int K) {
  MidPlan p;
  if (!make_plan(M, N, K, p) || p.S == 1) return 0;
  return static_cast<size_t>(p.m_tiles * p.n_tiles - p.t_full) * p.S * (32 * p.mf) * (64 * p.nf) * sizeof(float);
}

extern "C" int bma_gemm_mid(const void* x, int64_t ldx, const void* w, int64_t ldw, void* y, int64_t ldy, int M, int N,
                            int K, int dtype, void* ws, size_t ws_bytes, void* stream) {
  if (M < 0 || N < 0 || K <= 0 || ldx < K || ldw < K || ldy < N) return BMA_EINVAL;
  if (M == 0 || N == 0) return BMA_OK;
  if (!x || !w || !y) return BMA_EINVAL;
  if (dtype != BMA_BF16 && dtype != BMA_F16) return BMA_EDTYPE;
  if (K % kBK) return BMA_ELIMIT;
  if ((ldx > ldw ? ldx : ldw) * 2 * 256 + static_cast<int64_t>(K) * 2 >= (int64_t{1} << 31)) return BMA_ELIMIT;   // (32-bit offsets inside a tile)
  if ((ldx * 2) % 16 || (ldw * 2) % 16 || (ldy * 2) % 8) return BMA_EALIGN;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) % 16 || reinterpret_cast<uintptr_t>(y) % 8 ||
      reinterpret_cast<uintptr_t>(ws) % 16)
    return BMA_EALIGN;
  MidPlan p;
  if (!make_plan(M, N, K, p)) return BMA_EINVAL;
  if (p.S > 1 && (!ws || ws_bytes < bma_gemm_mid_ws_bytes(M, N, K))) return BMA_EINVAL;
  MidArgs a;
  a.x = static_cast<const char*>(x);
  a.w = static_cast<const char*>(w);
  a.y = static_cast<char*>(y);
  a.ws = static_cast<float*>(ws);
  a.ldx = ldx; a.ldw = ldw; a.ldy = ldy;
  a.M = M; a.N = N; a.K = K; a.S = p.S; a.t_full = p.t_full; a.m_tiles = p.m_tiles; a.n_tiles = p.n_tiles; a.xcd = p.xcd;
  hipStream_t st = static_cast<hipStream_t>(stream);
  a.stamps = nullptr;
  a.stamp_units = 0;
#ifdef BMA_MID_STAMPS
  // BMA_MID_STAMPS_FILE=<file> gets kStampN x u64 per wave of workgroups 0, 100, 200 and the last one of every launch
  const char* dump = getenv("BMA_MID_STAMPS_FILE");
  a.stamp_units = getenv("BMA_MID_STAMPS_UNITS") ? atoi(getenv("BMA_MID_STAMPS_UNITS")) : 1;
  if (dump && *dump) {
    (void)hipMalloc(reinterpret_cast<void**>(&a.stamps), 4 * kNW * kStampN * 8);
    (void)hipMemsetAsync(a.stamps, 0, 4 * kNW * kStampN * 8, st);
  }
#endif
  const int n_split_tiles = p.m_tiles * p.n_tiles - p.t_full;
  const dim3 grid(static_cast<unsigned>(p.t_full + n_split_tiles * p.S)), block(kNW * 64), rblock(256);
  BMA_PROF_BEGIN(BMA_K_GEMM_MID, st, 2.0 * (static_cast<double>(M) * K + static_cast<double>(N) * K + static_cast<double>(M) * N));
#define BMA_MID_GO(DT_)                                                                         \
  do {                                                                                          \
    if (p.nf == 3) hipLaunchKernelGGL((gemm_mid_kernel<DT_, kMF, 3>), grid, block, 0, st, a);   \
    else hipLaunchKernelGGL((gemm_mid_kernel<DT_, kMF, 4>), grid, block, 0, st, a);             \
  } while (0)
  if (dtype == BMA_BF16) BMA_MID_GO(BMA_BF16);
  else BMA_MID_GO(BMA_F16);
#undef BMA_MID_GO
  BMA_LAUNCH_CHECK();
#ifdef BMA_MID_STAMPS
  if (a.stamps) {
    std::vector<unsigned long long> host(4 * kNW * kStampN);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(host.data(), a.stamps, host.size() * 8, hipMemcpyDeviceToHost);
    (void)hipFree(a.stamps);
    if (FILE* f = fopen(dump, "wb")) {
      fwrite(host.data(), 8, host.size(), f);
      fclose(f);
    }
  }
#endif
  if (p.S > 1) {
    const int64_t n4 = static_cast<int64_t>(n_split_tiles) * kNW * p.mf * p.nf * 64;
    const dim3 rgrid(static_cast<unsigned>((n4 + 255) / 256));
    if (dtype == BMA_BF16) hipLaunchKernelGGL((gemm_mid_reduce_kernel<BMA_BF16>), rgrid, rblock, 0, st, a, p.mf, p.nf);
    else hipLaunchKernelGGL((gemm_mid_reduce_kernel<BMA_F16>), rgrid, rblock, 0, st, a, p.mf, p.nf);
    BMA_LAUNCH_CHECK();
  }
  BMA_PROF_END(BMA_K_GEMM_MID, st);                             // both launches of a split product
  return BMA_OK;
}
