// bma_allgather_f32: the one collective of the sharded chunk loop (reference bimodal_attack.py:1282-1299 has no sharding;
// SURVEY.md 8b lists this entry point): every rank's n_local fp32 values -- its candidates' losses, padded with +inf --
// gathered into out[world][n_local] on every rank.
//
// The communicator is the HOST's (an ncclComm_t of the RCCL instance the host process already uses): this library
// must call into that same instance, so it does not link RCCL -- it looks `ncclAllGather` up in the RCCL already
// loaded into the process (dlopen with RTLD_NOLOAD) and never loads one itself (see resolve()).  The Python host
// of this repo does not come through here (torch.distributed owns its communicator and does not hand it out:
// bimodalattack_amd/dist.py); a C, C++ or Go host that created its communicator with ncclCommInitRank does.
#include <dlfcn.h>

#include <mutex>

#include "bma_common.h"

namespace {

// rccl.h's prototype, spelled out so that the header need not be on the include path: ncclResult_t is an enum
// (ncclSuccess = 0), ncclDataType_t an enum (ncclFloat32 = 7), ncclComm_t and hipStream_t pointers
typedef int (*all_gather_fn)(const void* sendbuff, void* recvbuff, size_t sendcount, int datatype, void* comm, void* stream);
constexpr int kNcclFloat32 = 7;

// Only an RCCL that is ALREADY in the process will do: the communicator belongs to the instance that made it, and a second
// instance loaded here (another soname, another copy) would be handed a pointer it never created -- undefined behaviour,
// not an error code.  So: the loaded library by its usual sonames (RTLD_NOLOAD), else whatever the process exports under
// the symbol's name (a host that links RCCL statically); else no collective (BMA_ECOLL).  A miss is not remembered: the
// host may load RCCL later.
all_gather_fn resolve() {
  static std::mutex mu;
  static all_gather_fn fn = nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (fn) return fn;
  const char* names[] = {"librccl.so.1", "librccl.so", "libnccl.so.2"};
  void* h = nullptr;
  for (const char* n : names)
    if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
  void* sym = h ? dlsym(h, "ncclAllGather") : nullptr;
  if (!sym) sym = dlsym(RTLD_DEFAULT, "ncclAllGather");
  fn = reinterpret_cast<all_gather_fn>(sym);
  return fn;
}

}  // namespace

extern "C" int bma_allgather_f32(const float* local, int64_t n_local, float* out, int rank, int world, void* comm,
                                 void* stream) {
  if (n_local < 0 || world < 1 || rank < 0 || rank >= world) return BMA_EINVAL;
  if (n_local == 0) return BMA_OK;
  if (!local || !out) return BMA_EINVAL;
  if ((reinterpret_cast<uintptr_t>(local) | reinterpret_cast<uintptr_t>(out)) % 4) return BMA_EALIGN;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!comm) {
    // no communicator: only a world of one can do without (out = local)
    if (world != 1) return BMA_EINVAL;
    if (local != out && hipMemcpyAsync(out, local, static_cast<size_t>(n_local) * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
      return BMA_ELAUNCH;
    return BMA_OK;
  }
  all_gather_fn fn = resolve();
  if (!fn) return BMA_ECOLL;
  return fn(local, out, static_cast<size_t>(n_local), kNcclFloat32, comm, stream) == 0 ? BMA_OK : BMA_ECOLL;
}
