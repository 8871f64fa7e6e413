// a7 -- candidate-batch embedding gather/splice
//   reference bimodal_attack.py:1112-1225: emb(sampled_ids), .repeat(sw,1,1) of every shared
//   segment, torch.cat along the sequence axis.
//
// out[B][S][D] is written exactly once; shared rows are read once per workgroup
// and kept in registers while they are stored to many candidates, gathered rows are
// read once per candidate.  HBM-bound on the write stream: algorithmic bytes =
// B*S*D*es written + B*n_opt*D*es gathered (2.78 GB at B=512, S=644, D=4096, bf16).
// The reference's repeat+cat moves every byte at least twice.
//
// Decomposition: one 256-thread workgroup owns one sequence row s and a slab of
// candidates [b0, b0+CB); a lane owns 16-byte column chunks (D*es/16 per row, 512 for
// D=4096 bf16 -> two chunks per lane).  Within a wave the stores of one instruction
// cover 1 KiB of one row: full-line, coalesced.  grid = S x ceil(B/CB) workgroups
// (thousands), so every CU streams.

#include "bma_common.h"
#include "bma_profile.h"

namespace {

using bma::uint4_t;

struct SpliceArgs {
  const void* ptr[BMA_MAX_SEGS];
  int32_t start[BMA_MAX_SEGS + 1];  // first sequence row of segment i; start[n] = S
  int32_t kind[BMA_MAX_SEGS];
  int32_t n;
};

template <int DT>
__device__ __forceinline__ uint4_t scale_chunk(uint4_t w, float s) {
  if (DT == BMA_F32) {
    w.x = __float_as_uint(__uint_as_float(w.x) * s); w.y = __float_as_uint(__uint_as_float(w.y) * s);
    w.z = __float_as_uint(__uint_as_float(w.z) * s); w.w = __float_as_uint(__uint_as_float(w.w) * s);
    return w;
  }
  uint4_t r;
  r.x = bma::pack16<DT>(bma::unpack16<DT>(w.x, 0) * s, bma::unpack16<DT>(w.x, 1) * s);
  r.y = bma::pack16<DT>(bma::unpack16<DT>(w.y, 0) * s, bma::unpack16<DT>(w.y, 1) * s);
  r.z = bma::pack16<DT>(bma::unpack16<DT>(w.z, 0) * s, bma::unpack16<DT>(w.z, 1) * s);
  r.w = bma::pack16<DT>(bma::unpack16<DT>(w.w, 0) * s, bma::unpack16<DT>(w.w, 1) * s);
  return r;
}

// cpr = 16-byte chunks per row (D*es/16).  Each lane handles chunks tid, tid+256, ...
template <int DT>
__global__ __launch_bounds__(256) void splice_kernel(SpliceArgs a, const void* __restrict__ emb, int V,
                                                     const int64_t* __restrict__ ids, int B, int n_opt, int S,
                                                     int cpr, int cand_per_wg, float emb_scale,
                                                     uint4_t* __restrict__ out) {
  const int s = blockIdx.x;
  const int b0 = blockIdx.y * cand_per_wg;
  const int b1 = min(B, b0 + cand_per_wg);
  int seg = 0;
#pragma unroll
  for (int i = 1; i < BMA_MAX_SEGS; ++i)
    if (i < a.n && s >= a.start[i]) seg = i;
  const int r = s - a.start[seg];           // row inside the segment
  const int len = a.start[seg + 1] - a.start[seg];
  const int kind = a.kind[seg];
  const int tid = threadIdx.x;
  const int64_t row_stride = static_cast<int64_t>(S) * cpr;  // chunks between candidates in `out`

  if (kind == BMA_SEG_SHARED) {
    const uint4_t* src = static_cast<const uint4_t*>(a.ptr[seg]) + static_cast<int64_t>(r) * cpr;
    for (int c = tid; c < cpr; c += 256) {
      const uint4_t w = src[c];
      uint4_t* dst = out + static_cast<int64_t>(b0) * row_stride + static_cast<int64_t>(s) * cpr + c;
      for (int b = b0; b < b1; ++b, dst += row_stride) *dst = w;
    }
  } else if (kind == BMA_SEG_PERCAND) {
    const uint4_t* src = static_cast<const uint4_t*>(a.ptr[seg]);
    for (int b = b0; b < b1; ++b) {
      const uint4_t* sp = src + (static_cast<int64_t>(b) * len + r) * cpr;
      uint4_t* dst = out + static_cast<int64_t>(b) * row_stride + static_cast<int64_t>(s) * cpr;
      for (int c = tid; c < cpr; c += 256) dst[c] = sp[c];
    }
  } else {  // BMA_SEG_GATHER
    const uint4_t* table = static_cast<const uint4_t*>(emb);
    for (int b = b0; b < b1; ++b) {
      int64_t id = ids[static_cast<int64_t>(b) * n_opt + r];
      id = id < 0 ? 0 : (id >= V ? V - 1 : id);  // never read outside the table
      const uint4_t* sp = table + id * cpr;
      uint4_t* dst = out + static_cast<int64_t>(b) * row_stride + static_cast<int64_t>(s) * cpr;
      if (emb_scale == 1.0f) {
        for (int c = tid; c < cpr; c += 256) dst[c] = sp[c];
      } else {
        for (int c = tid; c < cpr; c += 256) dst[c] = scale_chunk<DT>(sp[c], emb_scale);
      }
    }
  }
}

// Row-list form (ragged scoring, layout.ragged_plan): output row n is row slot[n] = b*S + s of the padded [B][S][D]
// block bma_splice would build -- written directly, one workgroup per output row, so the block itself (of which ragged
// scoring needs ~3/4 of the rows) is never materialised and never read back by a gather.
template <int DT>
__global__ __launch_bounds__(256) void splice_rows_kernel(SpliceArgs a, const void* __restrict__ emb, int V,
                                                          const int64_t* __restrict__ ids, int B, int n_opt, int S,
                                                          int cpr, float emb_scale, const int* __restrict__ slot,
                                                          uint4_t* __restrict__ out) {
  const int64_t n = blockIdx.x;
  int64_t sl = slot[n];
  const int64_t total = static_cast<int64_t>(B) * S;
  sl = sl < 0 ? 0 : (sl >= total ? total - 1 : sl);        // never read outside the sources
  const int b = static_cast<int>(sl / S);
  const int s = static_cast<int>(sl - static_cast<int64_t>(b) * S);
  int seg = 0;
#pragma unroll
  for (int i = 1; i < BMA_MAX_SEGS; ++i)
    if (i < a.n && s >= a.start[i]) seg = i;
  const int r = s - a.start[seg];
  const int len = a.start[seg + 1] - a.start[seg];
  const int kind = a.kind[seg];
  const uint4_t* sp;
  bool scale = false;
  if (kind == BMA_SEG_SHARED) {
    sp = static_cast<const uint4_t*>(a.ptr[seg]) + static_cast<int64_t>(r) * cpr;
  } else if (kind == BMA_SEG_PERCAND) {
    sp = static_cast<const uint4_t*>(a.ptr[seg]) + (static_cast<int64_t>(b) * len + r) * cpr;
  } else {
    int64_t id = ids[static_cast<int64_t>(b) * n_opt + r];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    sp = static_cast<const uint4_t*>(emb) + id * cpr;
    scale = emb_scale != 1.0f;
  }
  uint4_t* dst = out + n * cpr;
  if (scale) {
    for (int c = threadIdx.x; c < cpr; c += 256) dst[c] = scale_chunk<DT>(sp[c], emb_scale);
  } else {
    for (int c = threadIdx.x; c < cpr; c += 256) dst[c] = sp[c];
  }
}

int fill_args(const bma_segment* segs_host, int n_segs, const void* emb, int V, const int64_t* ids, int B, int n_opt,
              SpliceArgs& a, int& S) {
  S = 0;
  for (int i = 0; i < n_segs; ++i) {
    const bma_segment& sg = segs_host[i];
    if (sg.len < 0) return BMA_EINVAL;
    a.start[i] = S;
    a.kind[i] = sg.kind;
    a.ptr[i] = sg.ptr;
    if (sg.kind == BMA_SEG_GATHER) {
      if (sg.len != n_opt || n_opt <= 0 || V <= 0) return BMA_EINVAL;
      if (B > 0 && (!emb || !ids)) return BMA_EINVAL;
      if (reinterpret_cast<uintptr_t>(emb) % 16) return BMA_EALIGN;
    } else if (sg.kind == BMA_SEG_SHARED || sg.kind == BMA_SEG_PERCAND) {
      if (sg.len > 0 && B > 0 && !sg.ptr) return BMA_EINVAL;
      if (reinterpret_cast<uintptr_t>(sg.ptr) % 16) return BMA_EALIGN;
    } else {
      return BMA_EINVAL;
    }
    S += sg.len;
  }
  for (int i = n_segs; i <= BMA_MAX_SEGS; ++i) a.start[i] = S;
  for (int i = n_segs; i < BMA_MAX_SEGS; ++i) { a.kind[i] = BMA_SEG_SHARED; a.ptr[i] = nullptr; }
  a.n = n_segs;
  return BMA_OK;
}

}  // namespace

extern "C" int bma_splice_rows(const bma_segment* segs_host, int n_segs, const void* emb, int V, const int64_t* ids,
                               int B, int n_opt, int D, int dtype, float emb_scale, const int* slot, int64_t n_rows,
                               void* out, void* stream) {
  if (!segs_host || n_segs <= 0 || n_segs > BMA_MAX_SEGS || B <= 0 || D <= 0 || n_rows < 0) return BMA_EINVAL;
  if (dtype != BMA_F32 && dtype != BMA_BF16 && dtype != BMA_F16) return BMA_EDTYPE;
  const int es = dtype == BMA_F32 ? 4 : 2;
  if ((static_cast<int64_t>(D) * es) % 16 != 0) return BMA_EALIGN;
  SpliceArgs a;
  int S = 0;
  const int rc = fill_args(segs_host, n_segs, emb, V, ids, B, n_opt, a, S);
  if (rc != BMA_OK) return rc;
  if (n_rows == 0) return BMA_OK;
  if (S == 0 || !slot || !out) return BMA_EINVAL;
  if (reinterpret_cast<uintptr_t>(out) % 16) return BMA_EALIGN;
  if (n_rows > 0x7fffffffLL) return BMA_ELIMIT;
  const int cpr = static_cast<int>(static_cast<int64_t>(D) * es / 16);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(static_cast<unsigned>(n_rows)), block(256);
  uint4_t* o = static_cast<uint4_t*>(out);
  // the row list is WRITTEN once; its sources -- at most n_opt*topk + n_opt table rows and the shared segments' rows,
  // each read by many output rows -- stay in cache, so the write is the kernel's HBM traffic
  BMA_PROF_BEGIN(BMA_K_SPLICE, st, static_cast<double>(n_rows) * D * es);
  switch (dtype) {
    case BMA_F32:
      hipLaunchKernelGGL((splice_rows_kernel<BMA_F32>), grid, block, 0, st, a, emb, V, ids, B, n_opt, S, cpr, emb_scale, slot, o);
      break;
    case BMA_BF16:
      hipLaunchKernelGGL((splice_rows_kernel<BMA_BF16>), grid, block, 0, st, a, emb, V, ids, B, n_opt, S, cpr, emb_scale, slot, o);
      break;
    default:
      hipLaunchKernelGGL((splice_rows_kernel<BMA_F16>), grid, block, 0, st, a, emb, V, ids, B, n_opt, S, cpr, emb_scale, slot, o);
      break;
  }
  BMA_PROF_END(BMA_K_SPLICE, st);
  BMA_LAUNCH_CHECK();
  return BMA_OK;
}

extern "C" int bma_splice(const bma_segment* segs_host, int n_segs, const void* emb, int V, const int64_t* ids,
                          int B, int n_opt, int D, int dtype, float emb_scale, void* out, void* stream) {
  if (!segs_host || n_segs <= 0 || n_segs > BMA_MAX_SEGS || B < 0 || D <= 0) return BMA_EINVAL;
  if (dtype != BMA_F32 && dtype != BMA_BF16 && dtype != BMA_F16) return BMA_EDTYPE;
  const int es = dtype == BMA_F32 ? 4 : 2;
  if ((static_cast<int64_t>(D) * es) % 16 != 0) return BMA_EALIGN;
  SpliceArgs a;
  int S = 0;
  for (int i = 0; i < n_segs; ++i) {
    const bma_segment& sg = segs_host[i];
    if (sg.len < 0) return BMA_EINVAL;
    a.start[i] = S;
    a.kind[i] = sg.kind;
    a.ptr[i] = sg.ptr;
    if (sg.kind == BMA_SEG_GATHER) {
      if (sg.len != n_opt || n_opt <= 0 || V <= 0) return BMA_EINVAL;
      if (B > 0 && (!emb || !ids)) return BMA_EINVAL;
      if (reinterpret_cast<uintptr_t>(emb) % 16) return BMA_EALIGN;
    } else if (sg.kind == BMA_SEG_SHARED || sg.kind == BMA_SEG_PERCAND) {
      if (sg.len > 0 && B > 0 && !sg.ptr) return BMA_EINVAL;
      if (reinterpret_cast<uintptr_t>(sg.ptr) % 16) return BMA_EALIGN;
    } else {
      return BMA_EINVAL;
    }
    S += sg.len;
  }
  for (int i = n_segs; i <= BMA_MAX_SEGS; ++i) a.start[i] = S;
  for (int i = n_segs; i < BMA_MAX_SEGS; ++i) { a.kind[i] = BMA_SEG_SHARED; a.ptr[i] = nullptr; }
  a.n = n_segs;
  if (B == 0 || S == 0) return BMA_OK;
  if (!out) return BMA_EINVAL;
  if (reinterpret_cast<uintptr_t>(out) % 16) return BMA_EALIGN;
  const int cpr = static_cast<int>(static_cast<int64_t>(D) * es / 16);
  // candidates per workgroup: enough workgroups to fill 256 CUs several times over,
  // few enough that a shared row is fetched rarely
  int cb = 16;
  while (cb > 1 && static_cast<int64_t>(S) * ((B + cb - 1) / cb) < 2048) cb >>= 1;
  const unsigned gy = static_cast<unsigned>((B + cb - 1) / cb);
  if (gy > 65535u) return BMA_ELIMIT;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(static_cast<unsigned>(S), gy), block(256);
  uint4_t* o = static_cast<uint4_t*>(out);
  {
    double gathered = 0.0, shared = 0.0, percand = 0.0;
    for (int i = 0; i < n_segs; ++i) {
      const double rows = static_cast<double>(segs_host[i].len);
      if (segs_host[i].kind == BMA_SEG_GATHER) gathered += rows * B;
      else if (segs_host[i].kind == BMA_SEG_PERCAND) percand += rows * B;
      else shared += rows;
    }
    // written once + every source row read once
    BMA_PROF_BEGIN(BMA_K_SPLICE, st, (static_cast<double>(B) * S + gathered + percand + shared) * D * es);
  }
  switch (dtype) {
    case BMA_F32:
      hipLaunchKernelGGL((splice_kernel<BMA_F32>), grid, block, 0, st, a, emb, V, ids, B, n_opt, S, cpr, cb,
                         emb_scale, o);
      break;
    case BMA_BF16:
      hipLaunchKernelGGL((splice_kernel<BMA_BF16>), grid, block, 0, st, a, emb, V, ids, B, n_opt, S, cpr, cb,
                         emb_scale, o);
      break;
    default:
      hipLaunchKernelGGL((splice_kernel<BMA_F16>), grid, block, 0, st, a, emb, V, ids, B, n_opt, S, cpr, cb,
                         emb_scale, o);
      break;
  }
  BMA_PROF_END(BMA_K_SPLICE, st);
  BMA_LAUNCH_CHECK();
  return BMA_OK;
}
