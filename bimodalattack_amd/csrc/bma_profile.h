// Internal hooks of the opt-in event profiler (implemented in bma_api.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace bma_prof {
bool enabled();
void begin(int kernel, hipStream_t st, double algorithmic_bytes);
void end(int kernel, hipStream_t st);
}  // namespace bma_prof

#define BMA_PROF_BEGIN(k, st, bytes) \
  do { if (bma_prof::enabled()) bma_prof::begin((k), (st), (bytes)); } while (0)
#define BMA_PROF_END(k, st) \
  do { if (bma_prof::enabled()) bma_prof::end((k), (st)); } while (0)
