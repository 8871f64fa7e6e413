// Version and error strings of the C ABI (include/bma.h).
#include "bma_common.h"

extern "C" int bma_version(void) { return BMA_VERSION; }

extern "C" const char* bma_strerror(int code) {
  switch (code) {
    case BMA_OK: return "ok";
    case BMA_EINVAL: return "invalid argument";
    case BMA_EDTYPE: return "unsupported dtype";
    case BMA_EALIGN: return "pointer or stride not aligned";
    case BMA_ELAUNCH: return "kernel launch failed";
    case BMA_ELIMIT: return "size beyond kernel limit";
    default: return "unknown error";
  }
}
