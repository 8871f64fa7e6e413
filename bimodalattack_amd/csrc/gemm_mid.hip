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
//   * both operands go L2 -> LDS by `global_load_lds_dwordx4` in 1-KiB pieces of 8 rows x 128 B (whole cache lines: with
//     64-byte rows a CU took in 19 B/clk, half of what it does with full lines), per operand its own ring of K = 64
//     units: two slots for x (L2-resident: every column tile re-reads it), three or four for w (HBM);
//   * LDS image: 128-byte rows, the 16-byte chunk c of row r stored at position c ^ (r & 7) -- applied to the DMA's SOURCE
//     address and to the fragment read -- so every `ds_read_b128` of an MFMA fragment is conflict-free (as gemm_nt.hip);
//   * the k loop runs in PHASES separated by workgroup barriers: a wave alternates a memory phase (fragments of unit u
//     LDS -> registers, its DMA pieces of a later unit, the counted `s_waitcnt vmcnt` that lands the next one) with a
//     compute phase (2 MF NF MFMAs, registers only); the two row halves run one phase apart, so on every SIMD one wave
//     issues MFMAs while the other sits in its loads -- an LDS-DMA piece holds its wave's issue for 60-100 cycles
//     (MI355X_MICROARCH.md, price list), which at one wave per SIMD came straight out of the MFMA time.  The upper half
//     issues every x piece, the lower half every w piece: a unit of x is issued two phases, a unit of w three or five
//     phases before its first read;
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

#include <type_traits>

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
// Timing experiments only (the loop with its MFMAs or its DMA compiled out): build with -DBMA_MID_DEBUG and
// bma_gemm_mid_set_plan's flags bits 2-4 switch them on -- WRONG results.  The shipped library has no such switch.
#ifdef BMA_MID_DEBUG
constexpr bool kDbg = true;
#else
constexpr bool kDbg = false;
#endif

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
  int dbg;                 // BMA_MID_DEBUG builds only (flags >> 2): bit 0 = no DMA after the prologue, bit 1 = no MFMA, bit 2 = every workgroup reads the same w rows
};

template <int DT>
__device__ __forceinline__ f32x4 mfma16(const uint4_t& a, const uint4_t& b, const f32x4& c) {
  if (DT == BMA_BF16)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N <= 63, "vmcnt is six bits");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// MF / NF: 16-row fragments of x / of w per wave (tile = 32 MF rows of x by 64 NF rows of w); NB: ring slots of w
template <int DT, int MF, int NF, int NB>
__global__ __launch_bounds__(kNW * 64) void gemm_mid_kernel(MidArgs a) {
  constexpr int BM = 32 * MF, BN = 64 * NF;
  constexpr int ASZ = BM * kRowB, BSZ = BN * kRowB;              // bytes of a unit of x / of w
  constexpr int CA = MF, CB = 2 * NF;                            // 1-KiB pieces (8 rows x 128 B) per wave of the issuing half
  constexpr int CMAX = CA > CB ? CA : CB;
  static_assert(kNA * ASZ + NB * BSZ <= kLds, "rings beyond the LDS");
  static_assert(NB >= 3 && NB <= 4 && CB * (NB - 2) <= 63, "w ring");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[kNA * ASZ + NB * BSZ];
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

  // ---- DMA pieces: the upper row half issues x (piece p = rows 8p..8p+7 of the tile, dealt to wave p % 4), the lower w ----
  const int prow = lane >> 3;                                    // row inside the piece == (tile row & 7)
  const int pchunk = (lane & 7) ^ prow;                          // source chunk that lands at LDS position lane & 7
  const char* src[CMAX];
  if (wr == 0) {
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      int m = m0 + (wc + 4 * i) * 8 + prow;
      m = m < a.M ? m : a.M - 1;                                 // rows past M repeat the last one (never stored)
      src[i] = a.x + (static_cast<int64_t>(m) * a.ldx) * 2 + pchunk * 16;
    }
  } else {
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      int n = ((kDbg && (a.dbg & 4)) ? 0 : n0) + (wc + 4 * i) * 8 + prow;
      n = n < a.N ? n : a.N - 1;
      src[i] = a.w + (static_cast<int64_t>(n) * a.ldw) * 2 + pchunk * 16;
    }
  }
  auto issue_a = [&](int u, int slot) {                          // wave-uniform destination; the DMA adds lane*16
    unsigned char* base = lds + slot * ASZ + wc * 1024;
    const int64_t ko = static_cast<int64_t>(u) * kRowB;
#pragma unroll
    for (int i = 0; i < CA; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + ko),
                                       (__attribute__((address_space(3))) void*)(base + i * 4096), 16, 0, 0);
  };
  auto issue_b = [&](int u, int slot) {
    unsigned char* base = lds_b + slot * BSZ + wc * 1024;
    const int64_t ko = static_cast<int64_t>(u) * kRowB;
#pragma unroll
    for (int i = 0; i < CB; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + ko),
                                       (__attribute__((address_space(3))) void*)(base + i * 4096), 16, 0, 0);
  };

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

  uint4_t wf[2][NF], xf[2][MF];
  auto read_frags = [&](int sa, int sb) {
    const unsigned char* pa = lds + sa * ASZ + offa;
    const unsigned char* pb = lds_b + sb * BSZ + offb;
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      wf[0][j] = *reinterpret_cast<const uint4_t*>(pb + j * 2048 + foff0);
      wf[1][j] = *reinterpret_cast<const uint4_t*>(pb + j * 2048 + foff1);
    }
#pragma unroll
    for (int i = 0; i < MF; ++i) {
      xf[0][i] = *reinterpret_cast<const uint4_t*>(pa + i * 2048 + foff0);
      xf[1][i] = *reinterpret_cast<const uint4_t*>(pa + i * 2048 + foff1);
    }
  };
  auto compute = [&]() {
    if (kDbg && (a.dbg & 2)) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int j = 0; j < NF; ++j) asm volatile("" ::"v"(wf[kk][j]));
#pragma unroll
        for (int i = 0; i < MF; ++i) asm volatile("" ::"v"(xf[kk][i]));
      }
      return;
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[j][i] = mfma16<DT>(wf[kk][j], xf[kk][i], acc[j][i]);
    __builtin_amdgcn_s_setprio(0);
  };

  const int n_units = u1 - u0;
  if (wr == 0) {
    // ---- upper row half: issues x.  Unit u+1 goes out in the memory phase of unit u and is waited for behind the MFMAs ----
    issue_a(u0, 0);
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    int sa = 0, sb = 0;
    for (int k = 0; k < n_units; ++k) {
      read_frags(sa, sb);
      if (k + 1 < n_units && !(kDbg && (a.dbg & 1))) issue_a(u0 + k + 1, sa ^ 1);   // slot of unit k-1: both halves read it a barrier ago
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // fragments in registers: slots may be refilled behind the barrier
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      compute();
      wait_vm<0>();                                              // x of unit k+1 landed
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      sa ^= 1;
      sb = sb + 1 == NB ? 0 : sb + 1;
    }
    __builtin_amdgcn_s_barrier();                                // sits out the lower half's last phase
  } else {
    // ---- lower row half, one phase behind: issues w.  Unit u+NB-1 goes out in the memory phase of unit u --------------------
#pragma unroll
    for (int s_ = 0; s_ < NB - 1; ++s_)
      if (s_ < n_units) issue_b(u0 + s_, s_);
    {
      const int behind = (n_units < NB - 1 ? n_units : NB - 1) - 1;
      if (behind >= 2 && NB > 3) wait_vm<CB * 2>();
      else if (behind >= 1) wait_vm<CB>();
      else wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
    int sa = 0, sb = 0;
    for (int k = 0; k < n_units; ++k) {
      read_frags(sa, sb);
      if (k + NB - 1 < n_units && !(kDbg && (a.dbg & 1))) {                 // into the slot of unit k-1
        int ns = sb + NB - 1;
        ns = ns >= NB ? ns - NB : ns;
        issue_b(u0 + k + NB - 1, ns);
      }
      {                                                          // w of unit k+1 landed: the units issued behind it may stay in flight
        const int behind = n_units - 2 - k;                      // min(behind, NB-2)
        if (kDbg && (a.dbg & 1)) wait_vm<0>();
        else if (behind >= NB - 2) wait_vm<CB * (NB - 2)>();
        else if (NB > 3 && behind == 1) wait_vm<CB>();
        else wait_vm<0>();
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      compute();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      sa ^= 1;
      sb = sb + 1 == NB ? 0 : sb + 1;
    }
  }

  // ---- split tiles: the partial leaves in register order; bma_gemm_mid's second launch sums the splits ---------------------
  if (split >= 0) {
    f32x4* out = reinterpret_cast<f32x4*>(a.ws) +
                 ((static_cast<int64_t>(tile - a.t_full) * a.S + split) * kNW + wave) * (NF * MF * 64) + lane;
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
      for (int i = 0; i < MF; ++i) out[(j * MF + i) * 64] = acc[j][i];
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
}

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
  int mf, nf, m_tiles, n_tiles, t_full, S, xcd, dbg;
};

int g_nf = 0, g_S = 0, g_tail = -1, g_flags = -1;   // tuning override (bma_gemm_mid_set_plan): 0 / -1 = the planner's choice

// The decomposition, by tile count (measured on the seven products of a LLaVA-7B layer at 644 rows, tools/gemm_bench.py
// --mid --sweep; profiles/r4_gemm_mid_sweep.txt).  With every CU pulling operands the kernel runs at what L2 delivers into
// the LDS (~8.5 TB/s chip-wide, measured with the MFMAs compiled out and every line an L2 hit), so the plan that moves the
// fewest bytes in ONE round of 256 workgroups wins:
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
  p.dbg = kDbg ? (flags >> 2) & 7 : 0;
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
  a.M = M; a.N = N; a.K = K; a.S = p.S; a.t_full = p.t_full; a.m_tiles = p.m_tiles; a.n_tiles = p.n_tiles; a.xcd = p.xcd; a.dbg = p.dbg;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int n_split_tiles = p.m_tiles * p.n_tiles - p.t_full;
  const dim3 grid(static_cast<unsigned>(p.t_full + n_split_tiles * p.S)), block(kNW * 64), rblock(256);
  BMA_PROF_BEGIN(BMA_K_GEMM_MID, st, 2.0 * (static_cast<double>(M) * K + static_cast<double>(N) * K + static_cast<double>(M) * N));
#define BMA_MID_GO(DT_)                                                                           \
  do {                                                                                            \
    if (p.nf == 3) hipLaunchKernelGGL((gemm_mid_kernel<DT_, kMF, 3, 4>), grid, block, 0, st, a);   \
    else hipLaunchKernelGGL((gemm_mid_kernel<DT_, kMF, 4, 3>), grid, block, 0, st, a);             \
  } while (0)
  if (dtype == BMA_BF16) BMA_MID_GO(BMA_BF16);
  else BMA_MID_GO(BMA_F16);
#undef BMA_MID_GO
  BMA_LAUNCH_CHECK();
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
