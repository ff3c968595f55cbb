// Version / error-string entry points of the C-ABI (include/mvhip.h).
#include "common.h"

extern "C" int mv_version(void) { return MV_ABI_VERSION; }

extern "C" const char *mv_strerror(int code) {
  switch (code) {
    case MV_OK: return "ok";
    case MV_E_NULLPTR: return "a required pointer argument is NULL";
    case MV_E_SHAPE: return "unsupported or inconsistent shape";
    case MV_E_DTYPE: return "unknown dtype";
    case MV_E_WORKSPACE: return "workspace too small or misaligned";
    case MV_E_ALIGN: return "pointer or stride not 16-byte aligned";
    case MV_E_RANGE: return "scalar argument out of range";
    default: break;
  }
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "unknown mvhip error";
}
