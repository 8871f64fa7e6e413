// LDS reads whose completion is counted by hand (attention kernels).
//
// Why not plain loads / the builtins: (1) with LDS-DMA (global_load_lds) in flight the compiler cannot tell an LDS read
// from the DMA's LDS writes and puts an s_waitcnt vmcnt(0) in front of every read it schedules itself, draining the
// ring; (2) left to itself it keeps two to five fragment reads in flight and waits for lgkmcnt(0) in front of every
// second MFMA, so a 16x16x32 product chain pays the LDS latency eight times per 32 keys.  Here a group of reads is
// issued at once, and a wait ties the group's registers to an s_waitcnt lgkmcnt(N) -- N = the reads issued AFTER the
// group that may still be in flight -- so that nothing consuming them is scheduled above the wait.  lgkmcnt counts LDS
// operations in order; anything else the compiler has in flight (its own LDS stores, scalar loads) only makes a wait
// conservative.
#pragma once
#include <cstdint>

namespace bma {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// LDS byte address of a generic pointer into __shared__ memory
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) unsigned char*)p));
}

// ds_read_b64_tr_b16: a 4-row x 16-column block of 16-bit elements per 16-lane group, delivered column-major
template <int OFF>
__device__ __forceinline__ u32x2 tr_read(uint32_t addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ u32x4 row_read(uint32_t addr) {       // ds_read_b128
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N, int K>
__device__ __forceinline__ void wait_rows(u32x4 (&f)[K]) {
  static_assert(K == 2 || K == 4 || K == 8, "one wait ties 2, 4 or 8 fragments");
  if constexpr (K == 2)
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f[0]), "+v"(f[1]) : "n"(N));
  else if constexpr (K == 8)
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7])
                 : "n"(N));
  else
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "n"(N));
}
template <int N>
__device__ __forceinline__ void wait_lgkm(u32x2 (&l)[4], u32x2 (&h)[4]) {
  asm volatile("s_waitcnt lgkmcnt(%8)"
               : "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3]), "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3])
               : "n"(N));
}

}  // namespace bma
