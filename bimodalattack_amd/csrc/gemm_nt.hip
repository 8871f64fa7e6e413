// a1 (gradient pass) -- the skinny products of the batch-1 forward/backward:  y[M][N] = x[M][K] . W[N][K]^T
//
// The reference's compute_gradient (bimodal_attack.py:953-1028) runs the language model at batch 1: every linear layer
// is a product of a handful of activation rows (65 at S = 66, 44 behind a reused prefix) with a 34-180 MB weight that is
// streamed from HBM exactly once.  Such a product is bound by how many bytes of W each CU keeps in flight, not by the
// matrix cores (M = 96 padded rows cost 6 us of MFMA time against a 22 us HBM floor for the 180 MB gate/up weight);
// the library's kernels reach 0.27-0.49 of 8 TB/s on these shapes inside the pass (bench.py, gradient_pass.gemms).
//
// Design (gfx950 only):
//   * one workgroup = 4 waves owns a 128-row slab of W (BN) x a K range x all M rows (M <= 16*MT); grid = slabs x splits,
//     with the split count chosen on the host so that the workgroups fill the 256 CUs in whole rounds;
//   * W and x tiles of BK = 64 (128-byte rows) go HBM/L2 -> LDS by `global_load_lds_dwordx4` (1 KiB = 8 rows x 128 B per
//     wave instruction, every byte of every line used), a ring of ST stages with ST-1 in flight (>= 48 KiB of W per CU),
//     counted `s_waitcnt vmcnt(N)` + ONE raw `s_barrier` per stage, a single LDS array (cdna_hip_programming.md 5:
//     "Pipelining across barriers", "Projection GEMM at M = 256" items 3-4);
//   * LDS image: linear rows, 16-byte chunk c of row r stored at chunk c ^ (r & 7) -- applied to the SOURCE address of
//     the DMA and to the fragment read (rule 21) -- so the `ds_read_b128` of an MFMA fragment (16 rows x 16 B per
//     k-group) is conflict-free;
//   * `v_mfma_f32_16x16x32`: A = 16 rows of W, B = 16 rows of x (both K-contiguous, the same read shape), so a lane ends
//     up with 4 consecutive n of one m: 8-byte output stores;
//   * split-K: partials as fp32 in the accumulator's own register order (1 KiB per wave instruction), an agent-scope
//     release + ticket per tile, and the last arriver sums the S partials IN SPLIT ORDER -- bitwise reproducible
//     whichever workgroup arrives last -- then resets the ticket for the next launch.
//
// Measured against the tuned library kernels, each shape over 32 different weights from one hipGraph (tools/
// gemm_bench.py; profiles/archive/r3_gemm_bench.txt): it wins where the library has to split K itself -- the input-gradient
// products through the transposed copies, N = 4096 with K = 12288 / 22016: 1.09-1.27x at 65 rows -- ties on down_proj
// and loses 7-27 % on the wide forward products (whose 172 / 96 / 86 slabs do not fill 256 CUs in whole rounds),
// so ops.gemm_nt_ok routes only K >= 3N to it.  What was tried on top and measured WORSE or no better: a stream-K
// decomposition (every CU the same number of k-steps, partial tiles summed by the last arriver: the 2 x 48-98 KB of
// partials per workgroup cost more than the balance gains, 0.65-0.94x), 3 or 5 LDS stages instead of 4 (+-1 %: the
// pipeline is not latency-bound), W addressed as if pre-tiled per stage (contiguous 16 KiB per stage: +3-14 %), and the
// same kernel at 599-643 rows, with 96 x 128 tiles (0.5-0.8x of the library) and with two row tiles of 320 / 352 rows x
// 192- or 128-row slabs in two 64-KB LDS stages (0.4-0.9x): one stage in flight per CU is ~21 GB/s at the 2.5-3 us a
// fill takes under load, a third of what the MFMA work of such a tile needs -- the LDS cannot hold the bytes in flight
// that would hide that latency, so a tall tile does not pay without operands that hit in the XCD's own L2.
//
// Algorithmic bytes per launch: (M*K + N*K + M*N) * es.

#include "bma_common.h"
#include "bma_profile.h"

namespace {

using bma::uint4_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kNW = 4;            // waves per workgroup
constexpr int kBK = 64;           // k per stage
constexpr int kRowB = kBK * 2;    // bytes per LDS row (16-bit types only)

struct GemmArgs {
  const char* x;
  const char* w;
  char* y;
  float* ws;
  int* cnt;
  int64_t ldx, ldw, ldy;   // elements
  int M, N, K, S, m_tiles;
  int R;                   // rows of w per slab (<= 64 * NTW, a multiple of 4)
  int xcd;                 // 1: workgroup ids remapped so that the splits of a tile share an XCD (grid a multiple of 8)
  int dbg;                 // -DBMA_NT_DEBUG builds only (bma_gemm_nt_set_plan flags bit 2): 1 = stop after publishing the partial (wrong result)
  int fenced;              // bma_gemm_nt_set_plan flags bit 3: the split-K hand-off with an agent-scope release / acquire fence pair as well
  // cross-product prefetch (bma_gemm_nt_next): the weight of the NEXT product of the chain and the decomposition its launch
  // will use -- workgroups that leave early (every split of a tile but the last arriver) pull the first stages of the next
  // launch's workgroups on their own XCD into the L2 while the reducers finish; nw == nullptr: none
  const char* nw;
  int64_t nldw;
  int nN, nT, nS, nR, n_mtiles, nGrid, nXcd;
  int main_grid;           // workgroups that compute; an unsplit launch that leaves CUs idle appends prefetch-only workgroups behind them
};

constexpr int kPrefetchStages = 4;      // the ring depth of the consumer: what its prologue and first step ask for

template <int DT>
__device__ __forceinline__ f32x4 mfma16(const uint4_t& a, const uint4_t& b, const f32x4& c) {
  if (DT == BMA_BF16)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// The first kPrefetchStages stages (128-byte lines) of the weight rows of every workgroup b' = blockIdx.x, blockIdx.x + grid,
// ... of the NEXT launch: b' is congruent to this workgroup's id modulo 8, i.e. it will run on this XCD (workgroups are dealt
// to the 8 XCDs round-robin), so the lines land in the L2 that will be asked for them; whatever that L2 drops before the
// next launch starts is still in the Infinity Cache.  One 4-byte load per line, results discarded (the wave's end waits
// for them: this workgroup has nothing else to do; the reducers it leaves behind are the launch's critical path).
__device__ __forceinline__ void prefetch_next(const GemmArgs& a, int first, int stride) {
  const int tid = threadIdx.x;
  // ONE destination register for every load, threaded through the asm statements as an in/out operand and "used" by the
  // final wait: the compiler must not hand a register with a load still in flight to anything else (a plain output
  // operand is free for reuse the moment the statement is over -- the returning load then overwrote the NEXT address:
  // a memory access fault on the first run of this function).
  unsigned int sink = 0;
  for (int b2 = first; b2 < a.nGrid; b2 += stride) {
    int bid = b2;
    if (a.nXcd) bid = (b2 & 7) * (a.nGrid >> 3) + (b2 >> 3);
    const int split = bid % a.nS, tile = bid / a.nS;
    const int n0 = (tile / a.n_mtiles) * a.nR;
    int rows = a.nN - n0;
    rows = rows < a.nR ? rows : a.nR;
    const int t0 = static_cast<int>(static_cast<int64_t>(a.nT) * split / a.nS);
    const int t1 = static_cast<int>(static_cast<int64_t>(a.nT) * (split + 1) / a.nS);
    int st = t1 - t0;
    st = st < kPrefetchStages ? st : kPrefetchStages;
    if (rows <= 0 || st <= 0 || (tile % a.n_mtiles) != 0) continue;      // (a second row tile re-reads the first one's lines)
    const int lines = rows * st;
    for (int i = tid; i < lines; i += kNW * 64) {
      const int r = i / st, k = i - r * st;
      const char* p = a.nw + (static_cast<int64_t>(n0 + r) * a.nldw) * 2 + static_cast<int64_t>(t0 + k) * kRowB;
      asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(p) : "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink) : : "memory");
}

// MT row tiles of x (64 or 96 rows), NTW 16-row tiles of w per wave (slabs of up to 128 or 192 rows), ST ring stages,
// NTL: the weight stream with the non-temporal policy (read once; x, which every workgroup re-reads, keeps the default)
template <int DT, int MT, int NTW, int ST, bool NTL>
__global__ __launch_bounds__(kNW * 64) void gemm_nt_kernel(GemmArgs a) {
  constexpr int BM = 16 * MT;
  constexpr int BN = 16 * NTW * kNW;
  constexpr int PXW = BM / 8 / kNW;                 // 1 KiB pieces of x issued by each wave per stage
  constexpr int PWW = BN / 8 / kNW;                 // ... of w
  constexpr int PW = PXW + PWW;
  constexpr int STAGE = (BM + BN) * kRowB;
  static_assert(BM % (8 * kNW) == 0, "x tile does not divide over the waves");
  static_assert(PW * (ST - 2) <= 63, "vmcnt is six bits");
  static_assert(ST * STAGE <= 160 * 1024, "ring beyond the LDS");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[ST * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  if (bid >= a.main_grid) {
    // a prefetch-only workgroup on a CU the unsplit product leaves idle: the e-th of them sits on XCD (main_grid + e) % 8
    // and takes every (extra / 8)-th of the next launch's workgroups on that XCD
    const int e = bid - a.main_grid, per_xcd = (static_cast<int>(gridDim.x) - a.main_grid) >> 3;
    prefetch_next(a, (bid & 7) + 8 * (e >> 3), 8 * per_xcd);
    return;
  }
  if (a.xcd) bid = (bid & 7) * (static_cast<int>(gridDim.x) >> 3) + (bid >> 3);   // blocks b, b+8, ... share an XCD: one tile's splits
  const int split = bid % a.S;
  const int tile = bid / a.S;
  const int m0 = (tile % a.m_tiles) * BM;
  const int n0 = (tile / a.m_tiles) * a.R;
  const int T = a.K / kBK;
  const int t0 = static_cast<int>(static_cast<int64_t>(T) * split / a.S);
  const int t1 = static_cast<int>(static_cast<int64_t>(T) * (split + 1) / a.S);
  int rows_here = a.N - n0;                                     // w rows this slab really has
  rows_here = rows_here < a.R ? rows_here : a.R;

  // ---- per-wave DMA pieces: a piece is 8 tile rows x 128 B; x rows first in the stage image, then w rows ------------
  // (Every piece of the image is fetched every stage, padding rows as re-reads of a real one.  Fetching only the pieces
  // that carry real rows -- 9 of x for 65 rows, 11 of w for an 86-row slab, dealt round-robin so that a wave issues C or
  // C-1 per stage, the loop instantiated per C -- was built and measured in round 4: no faster at equal slab height
  // (46.0 against 42.3 us on the gate/up shape) and slower wherever K is split, so the step time is not set by the
  // piece count.)
  const int prow = lane >> 3;                                  // row inside the piece == (tile row & 7)
  const int pchunk = (lane & 7) ^ prow;                        // source chunk that lands at LDS position lane & 7
  const char* srcx[PXW];
  const char* srcw[PWW];
#pragma unroll
  for (int i = 0; i < PXW; ++i) {
    int m = m0 + (wave * PXW + i) * 8 + prow;
    m = m < a.M ? m : a.M - 1;                                 // rows past M repeat the last one (never stored)
    srcx[i] = a.x + (static_cast<int64_t>(m) * a.ldx) * 2 + pchunk * 16;
  }
#pragma unroll
  for (int i = 0; i < PWW; ++i) {
    int r = (wave * PWW + i) * 8 + prow;
    r = r < rows_here ? r : rows_here - 1;                      // rows past the slab repeat its last one: cache hits, never stored
    srcw[i] = a.w + (static_cast<int64_t>(n0 + r) * a.ldw) * 2 + pchunk * 16;
  }
  auto issue = [&](int t, int slot) {
    unsigned char* base = lds + slot * STAGE;
#pragma unroll
    for (int i = 0; i < PXW; ++i)                               // wave-uniform destination; the DMA adds lane*16
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcx[i] + static_cast<int64_t>(t) * kRowB),
                                       (__attribute__((address_space(3))) void*)(base + (wave * PXW + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < PWW; ++i) {
      unsigned char* dst = base + BM * kRowB + (wave * PWW + i) * 1024;
      if (NTL)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcw[i] + static_cast<int64_t>(t) * kRowB),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 2);
      else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcw[i] + static_cast<int64_t>(t) * kRowB),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };

  f32x4 acc[NTW][MT];
#pragma unroll
  for (int j = 0; j < NTW; ++j)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[j][m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  // fragment read offsets inside a stage: row (lane & 15) of a 16-row tile, chunk (4*kk + (lane >> 4)) ^ (row & 7)
  const int frow = lane & 15, fg = lane >> 4;
  const int foff0 = frow * kRowB + ((fg) ^ (frow & 7)) * 16;          // kk = 0
  const int foff1 = frow * kRowB + ((4 + fg) ^ (frow & 7)) * 16;      // kk = 1

  // ---- prologue: ST-1 stages in flight ---------------------------------------------------------------------------
#pragma unroll
  for (int s = 0; s < ST - 1; ++s)
    if (t0 + s < t1) issue(t0 + s, s);

  int slot = 0;
  for (int t = t0; t < t1; ++t) {
    const int rem = t1 - 1 - t;                                  // stages behind t already issued: min(rem, ST-2)
    if (rem >= ST - 2) wait_vm<PW * (ST - 2)>();
    else if (ST > 3 && rem == 2) wait_vm<PW * 2>();
    else if (ST > 2 && rem == 1) wait_vm<PW>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();                                // everybody's pieces of stage t landed; slot t-1 is free
    if (t + ST - 1 < t1) {
      int ns = slot + ST - 1;
      ns = ns >= ST ? ns - ST : ns;
      issue(t + ST - 1, ns);
    }
    const unsigned char* xs = lds + slot * STAGE;
    const unsigned char* wsm = xs + BM * kRowB + wave * (NTW * 16) * kRowB;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int fo = kk ? foff1 : foff0;
      uint4_t wf[NTW];
#pragma unroll
      for (int j = 0; j < NTW; ++j) wf[j] = *reinterpret_cast<const uint4_t*>(wsm + j * 16 * kRowB + fo);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const uint4_t xf = *reinterpret_cast<const uint4_t*>(xs + m * 16 * kRowB + fo);
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[j][m] = mfma16<DT>(wf[j], xf, acc[j][m]);
      }
    }
    slot = slot + 1 == ST ? 0 : slot + 1;
  }

  // ---- split-K: publish the partial, the last arriver of a tile sums them in split order ---------------------------
  // (Round 6 built the alternative VERDICT r5 asked for -- a REDUCE-SCATTER among the tile's S resident workgroups, everybody
  // waiting on the tile's ticket and then summing every S-th fragment -- bit-equal to this form, and measured it 4-13 us SLOWER
  // on every split product of the pass (profiles/r6_gemm_nt_reduce_scatter_ab.txt: qkv dX 35.7 against 27.5 us at 65 rows):
  // waiting costs every workgroup an agent-scope poll that has to go past its XCD's L2 -- ~2 us a round trip under load -- where
  // the last arriver pays one ticket add; reading 1/S of the tile each does not win that back.  Reverted.)
  if (a.S > 1) {
    f32x4* wsv = reinterpret_cast<f32x4*>(a.ws);
    const int64_t per = static_cast<int64_t>(kNW) * NTW * MT * 64;        // float4 per (tile, split)
    f32x4* mine = wsv + (static_cast<int64_t>(tile) * a.S + split) * per + static_cast<int64_t>(wave) * NTW * MT * 64 + lane;
    // The partial leaves WRITE-THROUGH (agent-scope relaxed stores = global_store ... sc1, 8 bytes each): no release
    // fence is needed behind them -- a fence would write back this XCD's L2 with 48 KB freshly dirtied per workgroup,
    // most of the 10-12 us a split product spent behind its k-loop (cdna_hip_programming.md 6 Guideline 16, R1;
    // MI355X_MICROARCH.md "publish-large": 8.2 vs 3.0 us) -- only every storing wave's drain, the barrier, the ticket.
    // (16 bytes per lane through a raw buffer store with aux = sc1; as 8-byte atomic stores the publish was 2 us slower)
    typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
    const int64_t mine_off = ((static_cast<int64_t>(tile) * a.S + split) * per + static_cast<int64_t>(wave) * NTW * MT * 64 + lane) * 16;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(a.ws) + mine_off - static_cast<int64_t>(lane) * 16, 0, NTW * MT * 1024, 0x00020000);
    (void)mine;
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int m = 0; m < MT; ++m)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, acc[j][m]), rsrc, lane * 16 + (j * MT + m) * 1024, 0, 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // every storing wave drains its own stores
#ifdef BMA_NT_DEBUG
    if (a.dbg) return;
#endif
    // (the fenced variant -- flags bit 3 -- adds what the HIP memory model asks for on top: a release fence behind the
    // stores and an acquire in front of the reducer's loads.  The default relies on gfx942 / gfx950 cache behaviour --
    // sc1 stores are written through to the device-coherent level, sc1 loads bypass the CU's L1 -- and is held to the
    // fenced one by tests/test_kernels_gpu.py::test_gemm_nt_split_k_handoff_under_alternating_operands.)
    if (a.fenced) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();                                            // (also: every wave is done with the stage ring)
    int* flag = reinterpret_cast<int*>(lds);
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(a.ws) + (static_cast<int64_t>(tile) * a.S * per + static_cast<int64_t>(wave) * NTW * MT * 64) * 16, 0,
        static_cast<int>(a.S * per * 16), 0x00020000);
    {
      if (tid == 0) {
        const int ticket = __hip_atomic_fetch_add(a.cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int last = ticket == a.S - 1;
        if (last) __hip_atomic_store(a.cnt + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        *flag = last;
      }
      __syncthreads();
      if (*flag == 0) {
        if (a.nw) prefetch_next(a, blockIdx.x, gridDim.x);
        return;
      }
      if (a.fenced) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      // The last arriver reads every partial -- its own too -- back in split order, by sc1 loads (buffer_load_dwordx4 ... sc1:
      // past this CU's L1, which may hold stale lines of the workspace from an earlier launch) INSTEAD of an agent-scope
      // acquire in front of plain loads: valid because every byte was stored sc1, every storing wave drained before its
      // workgroup's barrier and ticket add, the reducer learnt it is last from the value its own add returned, and its
      // other waves load behind the barrier above (MI355X_MICROARCH.md, visibility: the hand-off table's third row).
#pragma unroll
      for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[j][m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      for (int s = 0; s < a.S; ++s) {
        f32x4 v[NTW][MT];
        const int so = static_cast<int>(static_cast<int64_t>(s) * per * 16);
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
          for (int m = 0; m < MT; ++m)
            v[j][m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, lane * 16 + (j * MT + m) * 1024, so, 16));
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[j][m] += v[j][m];
      }
    }
  }

  // ---- epilogue: a lane holds y[m][n .. n+3] for each of its tiles -------------------------------------------------
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    const int rs = (wave * NTW + j) * 16 + fg * 4;             // row inside the slab
    const int n = n0 + rs;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int row = m0 + m * 16 + frow;
      if (row < a.M && rs < rows_here) {
        char* dst = a.y + (static_cast<int64_t>(row) * a.ldy + n) * 2;
        const f32x4 v = acc[j][m];
        if (rs + 3 < rows_here) {
          bma::uint2_t o;
          o.x = bma::pack16<DT>(v.x, v.y);
          o.y = bma::pack16<DT>(v.z, v.w);
          *reinterpret_cast<bma::uint2_t*>(dst) = o;
        } else {                                                 // the slab (or N) ends inside these four: element by element
          const float e[4] = {v.x, v.y, v.z, v.w};
          for (int r = 0; r < 4 && rs + r < rows_here; ++r)
            reinterpret_cast<uint16_t*>(dst)[r] = static_cast<uint16_t>(bma::pack16<DT>(e[r], 0.0f) & 0xffffu);
        }
      }
    }
  }
}

constexpr int kMaxSplit = 16;

// compute units of the device CURRENT on the calling thread (256 on an MI355X; the planner fills whole rounds of them), asked
// once per device id (one function-local static served every device and thread -- a partitioned node's 32-CU
// devices would have been planned for 256).  Plain ints written once with the same value: a race here is harmless.
int cu_count() {
  static int cached[64] = {0};
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (dev >= 0 && dev < 64 && cached[dev]) return cached[dev];
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
  if (dev >= 0 && dev < 64) cached[dev] = v;
  return v;
}

struct Plan {
  int mt, m_tiles, ntw, R, slabs, S, xcd, ntl, dbg, fenced;
};

// tuning override (bma_gemm_nt_set_plan): 0 = the planner's choice
int g_ntw = 0, g_R = 0, g_S = 0, g_flags = -1;

// What one decomposition costs, in units of one stage of a 96 + 128-row image: the workgroups run in rounds of 256 (one
// per CU, the ring fills the LDS); a round lasts (k-steps + pipeline fill) stages, a stage costs in proportion to the
// rows of its image, and a split of K costs the partial-sum round trip
// plus the reducer's tail -- a lot: calibrated on the round-4 sweep (profiles/r4_gemm_sweep_flags_before_writethrough.txt), where two splits of
// 172-row slabs on all 256 CUs (52 us) lost to one pass over 128-row slabs on 172 of them (42 us) for N = 22016, while
// N = 4096 wants its eight splits.
double plan_cost(int bm, int bn, int R, int wgs, int T, int S) {
  const int kCUs = cu_count();
  const int rounds = (wgs + kCUs - 1) / kCUs;
  const double steps = static_cast<double>((T + S - 1) / S);
  (void)R;
  const double rows = bm + bn;                                 // the whole image is fetched every stage, padding rows too
  return rounds * (steps + 6.0) * rows / 224.0 + (S > 1 ? 25.0 + 1.0 * S : 0.0);
}

bool make_plan(int M, int N, int K, Plan& p) {
  if (M <= 0 || N <= 0 || K <= 0 || K % kBK) return false;
  p.mt = M <= 64 ? 4 : 6;                      // 64- or 96-row tiles
  const int bm = 16 * p.mt;
  p.m_tiles = (M + bm - 1) / bm;
  const int T = K / kBK;
  double best = 1e300;
  p.ntw = 2; p.R = 128; p.S = 1;
  for (int ntw = 2; ntw <= 3; ++ntw) {
    if (g_ntw && ntw != g_ntw) continue;
    const int bn = 64 * ntw;
    // candidate slab heights: the full tile, and every height that makes slabs x splits x row tiles a whole number of
    // rounds (172 rows x 2 splits for N = 22016, 192 x 4 for N = 12288, 172 x 4 for N = 11008, 128 x 8 for N = 4096)
    for (int S = 1; S <= kMaxSplit && S <= T; ++S) {
      if (g_S && S != g_S) continue;
      for (int rounds = 1; rounds <= 3; ++rounds) {
        const int want = rounds * cu_count() / (S * p.m_tiles); // slabs that fill `rounds` rounds exactly
        for (int pass = 0; pass < 2; ++pass) {
          // (a slab shorter than the tile only to spread ONE pass over more CUs does not pay -- 251 slabs of 88 rows
          // measured slower than 172 of 128: the padding rows are fetched all the same -- so balanced heights only
          // where K is split anyway)
          int R = pass == 0 ? bn : ((want > 0 && S > 1) ? ((N + want - 1) / want + 3) / 4 * 4 : 0);
          if (g_R) R = g_R;
          if (R < 16 || R > bn) continue;
          const int slabs = (N + R - 1) / R;
          const double c = plan_cost(bm, bn, R, slabs * p.m_tiles * S, T, S);
          if (c < best - 1e-9) { best = c; p.ntw = ntw; p.R = R; p.S = S; }
        }
      }
    }
  }
  if (best >= 1e299) return false;
  p.slabs = (N + p.R - 1) / p.R;
  const int grid = p.slabs * p.m_tiles * p.S;
  const int flags = g_flags >= 0 ? g_flags : 3;
  p.xcd = (flags & 1) && p.S > 1 && grid % 8 == 0;
  p.ntl = (flags & 2) ? 1 : 0;
  p.dbg = (flags & 4) ? 1 : 0;
  p.fenced = (flags & 8) ? 1 : 0;
  return true;
}

}  // namespace

extern "C" void bma_gemm_nt_set_plan(int ntw, int rows_per_slab, int splits, int flags) {
  g_ntw = ntw; g_R = rows_per_slab; g_S = splits; g_flags = flags;
}

extern "C" int bma_gemm_nt_plan(int M, int N, int K, int* out8) {
  Plan p;
  if (!make_plan(M, N, K, p)) return BMA_EINVAL;
  if (out8) {
    out8[0] = p.mt; out8[1] = p.m_tiles; out8[2] = p.ntw; out8[3] = p.R; out8[4] = p.slabs; out8[5] = p.S; out8[6] = p.xcd; out8[7] = p.ntl;
  }
  return BMA_OK;
}

extern "C" size_t bma_gemm_nt_ws_bytes(int M, int N, int K) {
  Plan p;
  if (!make_plan(M, N, K, p) || p.S == 1) return 0;
  return static_cast<size_t>(p.slabs) * p.m_tiles * p.S * (16 * p.mt) * (64 * p.ntw) * sizeof(float);
}

extern "C" int bma_gemm_nt_tiles(int M, int N, int K) {
  Plan p;
  if (!make_plan(M, N, K, p)) return 0;
  return p.slabs * p.m_tiles;
}

namespace {

int launch_gemm_nt(const void* x, int64_t ldx, const void* w, int64_t ldw, void* y, int64_t ldy, int M, int N, int K, int dtype,
                   void* ws, size_t ws_bytes, int* counters, int n_counters, const void* next_w, int64_t next_ldw, int next_N,
                   int next_K, void* stream) {
  if (M < 0 || N < 0 || K <= 0 || ldx < K || ldw < K || ldy < N) return BMA_EINVAL;
  if (M == 0 || N == 0) return BMA_OK;
  if (!x || !w || !y) return BMA_EINVAL;
  if (dtype != BMA_BF16 && dtype != BMA_F16) return BMA_EDTYPE;
  if (K % kBK) return BMA_ELIMIT;
  if ((ldx * 2) % 16 || (ldw * 2) % 16 || (ldy * 2) % 8) return BMA_EALIGN;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) % 16 || reinterpret_cast<uintptr_t>(y) % 8 ||
      reinterpret_cast<uintptr_t>(ws) % 16)
    return BMA_EALIGN;
  Plan p;
  if (!make_plan(M, N, K, p)) return BMA_EINVAL;
  const int tiles = p.slabs * p.m_tiles;
  if (p.S > 1) {
    if (!ws || !counters || ws_bytes < bma_gemm_nt_ws_bytes(M, N, K) || n_counters < tiles) return BMA_EINVAL;
  }
  int extra = 0;
  GemmArgs a = {};
  a.x = static_cast<const char*>(x);
  a.w = static_cast<const char*>(w);
  a.y = static_cast<char*>(y);
  a.ws = static_cast<float*>(ws);
  a.cnt = counters;
  a.ldx = ldx; a.ldw = ldw; a.ldy = ldy;
  a.M = M; a.N = N; a.K = K; a.S = p.S; a.m_tiles = p.m_tiles; a.R = p.R; a.xcd = p.xcd; a.dbg = p.dbg; a.fenced = p.fenced;
  if (next_w) {
    // the decomposition the NEXT launch will use (the same rows M: one pass, one row count)
    if (next_N <= 0 || next_K <= 0 || next_K % kBK || next_ldw < next_K || (next_ldw * 2) % 16 ||
        reinterpret_cast<uintptr_t>(next_w) % 16)
      return BMA_EINVAL;
    Plan q;
    if (!make_plan(M, next_N, next_K, q)) return BMA_EINVAL;
    // who prefetches: the workgroups a split launch lets go early (every split of a tile but its last arriver), or -- an
    // unsplit launch on fewer workgroups than CUs (gate/up: 172 of 256) -- prefetch-only workgroups on the idle CUs
    extra = p.S == 1 ? (cu_count() - tiles) / 8 * 8 : 0;
    if (extra < 0) extra = 0;
    if (p.S > 1 || extra > 0) {
      a.nw = static_cast<const char*>(next_w);
      a.nldw = next_ldw; a.nN = next_N; a.nT = next_K / kBK; a.nS = q.S; a.nR = q.R; a.n_mtiles = q.m_tiles;
      a.nGrid = q.slabs * q.m_tiles * q.S; a.nXcd = q.xcd;
    }
  }
  a.main_grid = tiles * p.S;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(static_cast<unsigned>(tiles * p.S + extra)), block(256);
  BMA_PROF_BEGIN(BMA_K_GEMM_NT, st, 2.0 * (static_cast<double>(M) * K + static_cast<double>(N) * K + static_cast<double>(M) * N));
#define BMA_GEMM_GO3(DT_, MT_, NTW_)                                                                    \
  do {                                                                                                  \
    if (p.ntl) hipLaunchKernelGGL((gemm_nt_kernel<DT_, MT_, NTW_, 4, true>), grid, block, 0, st, a);    \
    else hipLaunchKernelGGL((gemm_nt_kernel<DT_, MT_, NTW_, 4, false>), grid, block, 0, st, a);         \
  } while (0)
#define BMA_GEMM_GO(DT_)                                     \
  do {                                                       \
    if (p.mt == 4 && p.ntw == 2) BMA_GEMM_GO3(DT_, 4, 2);    \
    else if (p.mt == 4) BMA_GEMM_GO3(DT_, 4, 3);             \
    else if (p.ntw == 2) BMA_GEMM_GO3(DT_, 6, 2);            \
    else BMA_GEMM_GO3(DT_, 6, 3);                            \
  } while (0)
  if (dtype == BMA_BF16) BMA_GEMM_GO(BMA_BF16);
  else BMA_GEMM_GO(BMA_F16);
#undef BMA_GEMM_GO
#undef BMA_GEMM_GO3
  BMA_PROF_END(BMA_K_GEMM_NT, st);
  BMA_LAUNCH_CHECK();
  return BMA_OK;
}

}  // namespace

extern "C" int bma_gemm_nt(const void* x, int64_t ldx, const void* w, int64_t ldw, void* y, int64_t ldy, int M, int N,
                           int K, int dtype, void* ws, size_t ws_bytes, int* counters, int n_counters, void* stream) {
  return launch_gemm_nt(x, ldx, w, ldw, y, ldy, M, N, K, dtype, ws, ws_bytes, counters, n_counters, nullptr, 0, 0, 0, stream);
}

extern "C" int bma_gemm_nt_next(const void* x, int64_t ldx, const void* w, int64_t ldw, void* y, int64_t ldy, int M, int N,
                                int K, int dtype, void* ws, size_t ws_bytes, int* counters, int n_counters, const void* next_w,
                                int64_t next_ldw, int next_N, int next_K, void* stream) {
  return launch_gemm_nt(x, ldx, w, ldw, y, ldy, M, N, K, dtype, ws, ws_bytes, counters, n_counters, next_w, next_ldw, next_N,
                        next_K, stream);
}
