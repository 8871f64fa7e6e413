// Version, error strings and the opt-in event profiler of the C ABI (include/bma.h).
#include <mutex>
#include <vector>

#include "bma_common.h"
#include "bma_profile.h"

extern "C" int bma_version(void) { return BMA_VERSION; }

extern "C" const char* bma_strerror(int code) {
  switch (code) {
    case BMA_OK: return "ok";
    case BMA_EINVAL: return "invalid argument";
    case BMA_EDTYPE: return "unsupported dtype";
    case BMA_EALIGN: return "pointer or stride not aligned";
    case BMA_ELAUNCH: return "kernel launch failed";
    case BMA_ELIMIT: return "size beyond kernel limit";
    case BMA_ECOLL: return "collective library missing or failed";
    default: return "unknown error";
  }
}

// ---------------------------------------------------------------------------
// Profiler: when enabled, the dominant kernel of each entry point is bracketed by a
// pair of HIP events recorded on the launch stream, and the algorithmic bytes of the
// launch are tallied on the host.  bma_profile_read() resolves the events (it
// synchronises on them) and returns launches, summed device time and summed bytes.
// Off by default: nothing is recorded, so calls stay capturable into a hipGraph.
// ---------------------------------------------------------------------------
namespace bma_prof {

struct Slot {
  std::vector<hipEvent_t> begin, end;
  double bytes = 0.0;
  double resolved_ms = 0.0;
  int64_t resolved_n = 0;
};

static std::mutex g_mu;
static bool g_on = false;
static Slot g_slot[BMA_K_COUNT];

bool enabled() { return g_on; }

// Events recorded while a stream is being captured into a graph would become graph nodes that
// never fire outside a replay; skip them (the launch is then simply not tallied).
static bool capturing(hipStream_t st) {
  hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(st, &status) == hipSuccess && status != hipStreamCaptureStatusNone;
}

void begin(int k, hipStream_t st, double bytes) {
  if (capturing(st)) return;
  std::lock_guard<std::mutex> lk(g_mu);
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return;
  g_slot[k].begin.push_back(e0);
  g_slot[k].end.push_back(e1);
  g_slot[k].bytes += bytes;
  (void)hipEventRecord(e0, st);
}

void end(int k, hipStream_t st) {
  if (capturing(st)) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_slot[k].end.empty()) return;
  (void)hipEventRecord(g_slot[k].end.back(), st);
}

static void resolve(Slot& s) {
  for (size_t i = 0; i < s.begin.size(); ++i) {
    float ms = 0.0f;
    if (hipEventSynchronize(s.end[i]) == hipSuccess && hipEventElapsedTime(&ms, s.begin[i], s.end[i]) == hipSuccess) {
      s.resolved_ms += ms;
      s.resolved_n += 1;
    }
    (void)hipEventDestroy(s.begin[i]);
    (void)hipEventDestroy(s.end[i]);
  }
  s.begin.clear();
  s.end.clear();
}

}  // namespace bma_prof

extern "C" int bma_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(bma_prof::g_mu);
  for (int k = 0; k < BMA_K_COUNT; ++k) {
    bma_prof::resolve(bma_prof::g_slot[k]);
    bma_prof::g_slot[k].bytes = 0.0;
    bma_prof::g_slot[k].resolved_ms = 0.0;
    bma_prof::g_slot[k].resolved_n = 0;
  }
  bma_prof::g_on = on != 0;
  return BMA_OK;
}

extern "C" int bma_profile_read(int kernel, int64_t* launches, double* total_ms, double* total_bytes) {
  if (kernel < 0 || kernel >= BMA_K_COUNT) return BMA_EINVAL;
  std::lock_guard<std::mutex> lk(bma_prof::g_mu);
  bma_prof::Slot& s = bma_prof::g_slot[kernel];
  bma_prof::resolve(s);
  if (launches) *launches = s.resolved_n;
  if (total_ms) *total_ms = s.resolved_ms;
  if (total_bytes) *total_bytes = s.bytes;
  return BMA_OK;
}

extern "C" const char* bma_profile_kernel_name(int kernel) {
  switch (kernel) {
    case BMA_K_LINF: return "linf_step_vec4";
    case BMA_K_CE_ROWS: return "ce_rows_kernel";
    case BMA_K_CE_ROWS_B1: return "ce_rows_kernel";
    case BMA_K_CE_DLOGITS: return "ce_dlogits_kernel";
    case BMA_K_TOPK: return "mask_topk_kernel";
    case BMA_K_SCATTER: return "sample_scatter_kernel";
    case BMA_K_SPLICE: return "splice_kernel";
    case BMA_K_RMSNORM: return "rmsnorm_kernel";
    case BMA_K_SWIGLU: return "swiglu_kernel";
    case BMA_K_ROPE: return "rope_kernel";
    case BMA_K_ATTN_MERGE: return "attn_merge_kernel";
    case BMA_K_GATHER_ROWS: return "gather_rows_kernel";
    case BMA_K_RAGGED_ATTN: return "ragged_attn_kernel";
    case BMA_K_PREFIX_ATTN: return "prefix_attn_kernel";
    case BMA_K_ADD_RMSNORM: return "add_rmsnorm_kernel";
    case BMA_K_GEMM_NT: return "gemm_nt_kernel";
    case BMA_K_B1_ATTN: return "b1_attn_kernel";
    case BMA_K_GEMM_MID: return "gemm_mid_kernel";
    case BMA_K_CAUSAL_ATTN: return "causal_attn_kernel";
    default: return "?";
  }
}
