// Does the ADDRESS PATTERN of bma_gemm_mid's staging -- 1-KiB LDS-DMA pieces of 8 rows x 128 B, the rows a leading dimension
// (8 KiB at K = 4096) apart -- bound what the L2 delivers into the LDS?  tools/l2_to_lds_probe.hip measured 28 TB/s chip-wide
// for CONTIGUOUS 1-KiB pieces out of an L2-resident window, while three different schedules of the GEMM's k loop all run at
// the same 7-8 TB/s (profiles/r6_gemm_mid_*).  Here the same loop shape (8 waves per workgroup, one workgroup per CU, every
// piece of a step in flight before the one wait, nothing computed) reads an L2-resident window per XCD in four patterns:
//   contiguous      piece p of step s = 1 KiB at (s * P + p) KiB                                  (the earlier probe)
//   rows            piece p = rows 8p..8p+7, 128 B each at column block s, row stride RS          (bma_gemm_mid)
//   rows, swizzled  the same with the 16-byte chunks of a row permuted by (chunk ^ row & 7)        (bma_gemm_mid exactly)
//   pre-tiled       piece p of step s contiguous at ((p * steps) + s) KiB: what a [rows/8][K/64][8][64] copy would give
// for row strides 8 KiB (K = 4096), 8 KiB + 128 B (padded), 44032 B (K = 22016).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/l2_stride_probe.hip -o scratch/l2_stride_probe && scratch/l2_stride_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

constexpr int kThreads = 512;
constexpr int kPieces = 32;                  // 1-KiB pieces per step and workgroup (4 per wave; bma_gemm_mid moves 60): 2 MiB per XCD window
constexpr int kRows = kPieces * 8;           // 448 rows

// mode 0 contiguous, 1 rows, 2 rows swizzled, 3 pre-tiled
__global__ __launch_bounds__(kThreads) void probe(const unsigned char* src, size_t window, long long rs, int cols, int mode, int rounds,
                                                  unsigned int* out) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[kPieces * 1024];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const unsigned char* win = src + static_cast<size_t>(blockIdx.x & 7) * window;
  const int prow = lane >> 3, pchunk = mode == 2 ? ((lane & 7) ^ prow) : (lane & 7);
  unsigned int sum = 0;
  for (int r = 0; r < rounds; ++r)
    for (int s = 0; s < cols; ++s) {
#pragma unroll
      for (int i = 0; i < kPieces / 8; ++i) {
        const int p = wave + 8 * i;
        const unsigned char* g;
        if (mode == 0) g = win + (static_cast<size_t>(s) * kPieces + p) * 1024 + lane * 16;
        else if (mode == 3) g = win + (static_cast<size_t>(p) * cols + s) * 1024 + lane * 16;
        else g = win + static_cast<size_t>(p * 8 + prow) * rs + static_cast<size_t>(s) * 128 + pchunk * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(lds + p * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      sum += *reinterpret_cast<const unsigned int*>(lds + ((tid * 68 + s * 4) & (kPieces * 1024 - 4)));
      __syncthreads();
    }
  if (sum == 0x12345678u) out[blockIdx.x] = sum;
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
  const size_t window = static_cast<size_t>(32) << 20;                 // room for the widest row stride; what is touched is 256 rows x 8 KiB
  CK(hipMalloc(&src, 8 * window));
  CK(hipMalloc(&out, 4096));
  CK(hipMemset(src, 0x5a, 8 * window));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const char* names[] = {"contiguous 1-KiB pieces          ", "8 rows x 128 B per piece         ", "8 rows x 128 B, chunks swizzled  ",
                         "pre-tiled (piece-major, contiguous)"};
  const long long strides[] = {8192, 8192 + 128, 44032};
  const int cols = 64, rounds = 16;                                    // 64 column blocks of 128 B = one 8-KiB row; 1024 steps
  for (long long rs : strides) {
    std::printf("== row stride %lld B; per XCD window: %d rows (%.1f MiB touched), shared by its workgroups\n", rs, kRows,
                kRows * 8192.0 / (1 << 20));
    for (int mode = 0; mode < 4; ++mode) {
      if (mode == 0 && rs != 8192) continue;
      if (mode == 3 && rs != 8192) continue;
      for (int wgs : {64, 256}) {
        float ms = 0.0f;
        for (int it = 0; it < 2; ++it) {
          CK(hipEventRecord(e0));
          hipLaunchKernelGGL(probe, dim3(wgs), dim3(kThreads), 0, 0, src, window, rs, cols, mode, rounds, out);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          CK(hipEventElapsedTime(&ms, e0, e1));
        }
        const double bytes = static_cast<double>(wgs) * rounds * cols * kPieces * 1024;
        std::printf("   %s %3d workgroups: %8.1f us  %6.2f TB/s chip-wide  %6.1f GB/s per workgroup\n", names[mode], wgs, 1e3 * ms,
                    bytes / (1e-3 * ms) / 1e12, bytes / wgs / (1e-3 * ms) / 1e9);
      }
    }
  }
  return 0;
}
