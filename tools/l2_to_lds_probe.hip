// How fast does the chip move L2-resident data into the CUs' LDS -- by LDS-DMA (global_load_lds_dwordx4: what bma_gemm_mid
// and bma_gemm_nt stage with) and by plain vector loads followed by ds_write_b128 (what the library's GEMMs do)?
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/l2_to_lds_probe.hip -o /tmp/l2_to_lds_probe && /tmp/l2_to_lds_probe
//
// Every workgroup (512 threads = 8 waves, one per CU: 128 KB of LDS) streams 64 MiB, 128 KB per step -- every byte of a step in
// flight before the first is waited for, the same for both routes -- into the LDS, nothing computed; a checksum of one LDS word
// per step keeps the loads alive.  Three sources: a 1 MiB window per XCD that all its workgroups read (L2 hits), a 32 MiB window
// per XCD (Infinity Cache), 64 MiB of its own per workgroup (HBM).  Printed: chip-wide TB/s at 64 .. 256 workgroups, both routes.  DESIGN.md 5d measured the DMA route at 8-9 TB/s inside bma_gemm_mid with its
// MFMAs compiled out; the library's 256 x 256 x 64 kernel at 0.6 of MFMA peak needs ~11.4 TB/s of operands.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kThreads = 512;
constexpr int kStage = 128 * 1024;          // bytes per step: all of them in flight at once (16 x 16 B per thread)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// window: bytes a workgroup cycles through; shared = 1: the 8 windows belong to the XCDs (every workgroup of an XCD reads the same
// lines), 0: every workgroup has its own
template <bool DMA>
__global__ __launch_bounds__(kThreads) void stream_kernel(const unsigned char* src, size_t window, int shared, int steps, unsigned int* out) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[kStage];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const unsigned char* win = src + static_cast<size_t>(shared ? (blockIdx.x & 7) : blockIdx.x) * window;     // (workgroups are dealt round-robin to the XCDs)
  const int per = static_cast<int>(window / kStage);
  const int skew = shared ? 0 : 0;
  unsigned int sum = 0;
  for (int s = 0; s < steps; ++s) {
    const unsigned char* st = win + static_cast<size_t>((s + skew) % per) * kStage;
    unsigned char* dst = lds;
    if (DMA) {
      // a wave moves 16 KiB per step: 16 instructions of 1 KiB (64 lanes x 16 B), destination wave-uniform + lane * 16
#pragma unroll
      for (int i = 0; i < 16; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(st + (wave * 16 + i) * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(dst + (wave * 16 + i) * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      u32x4 r[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) r[i] = *reinterpret_cast<const u32x4*>(st + (wave * 16 + i) * 1024 + lane * 16);
#pragma unroll
      for (int i = 0; i < 16; ++i) *reinterpret_cast<u32x4*>(dst + (wave * 16 + i) * 1024 + lane * 16) = r[i];
    }
    __syncthreads();
    sum += *reinterpret_cast<const unsigned int*>(dst + ((tid * 68 + s * 4) & (kStage - 4)));
    __syncthreads();                                            // (the next step overwrites what was just read)
  }
  if (sum == 0x12345678u) out[blockIdx.x] = sum;          // (never true for the data below; keeps everything alive)
}

#define CK(x)                                                         \
  do {                                                                \
    hipError_t e_ = (x);                                              \
    if (e_ != hipSuccess) {                                           \
      std::printf("%s: %s\n", #x, hipGetErrorString(e_));            \
      return 1;                                                       \
    }                                                                 \
  } while (0)

int main() {
  unsigned char* src;
  unsigned int* out;
  const size_t total = static_cast<size_t>(16) << 30;                 // 16 GiB: 256 private windows of 64 MiB
  CK(hipMalloc(&src, total));
  CK(hipMalloc(&out, 4096));
  CK(hipMemset(src, 0x5a, total));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  struct Case {
    const char* what;
    size_t window;
    int shared;
  };
  const Case cases[] = {
      {"1 MiB per XCD, shared by its workgroups (L2 hits)                ", static_cast<size_t>(1) << 20, 1},
      {"32 MiB per XCD, shared (256 MiB in all: past the L2, Infinity Cache)", static_cast<size_t>(32) << 20, 1},
      {"64 MiB per WORKGROUP, private (16 GiB in all: HBM)                ", static_cast<size_t>(64) << 20, 0},
  };
  for (const Case& c : cases) {
    std::printf("== %s\n", c.what);
    const int steps = 512;                                            // 64 MiB per workgroup
    for (int route = 0; route < 2; ++route)
      for (int wgs : {64, 128, 240, 256}) {
        for (int it = 0; it < 2; ++it) {                              // the first launch warms the caches and the code
          CK(hipEventRecord(e0));
          if (route == 0) hipLaunchKernelGGL(stream_kernel<true>, dim3(wgs), dim3(kThreads), 0, 0, src, c.window, c.shared, steps, out);
          else hipLaunchKernelGGL(stream_kernel<false>, dim3(wgs), dim3(kThreads), 0, 0, src, c.window, c.shared, steps, out);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms = 0.0f;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (it == 1) {
            const double bytes = static_cast<double>(wgs) * steps * kStage;
            std::printf("   %s  %3d workgroups: %8.1f us  %6.2f TB/s chip-wide  %6.1f GB/s per workgroup\n",
                        route == 0 ? "LDS-DMA (global_load_lds_dwordx4)" : "vector loads + ds_write_b128     ", wgs, 1e3 * ms,
                        bytes / (1e-3 * ms) / 1e12, bytes / wgs / (1e-3 * ms) / 1e9);
          }
        }
      }
  }
  return 0;
}
