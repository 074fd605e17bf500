// api.hip -- version / error-string entry points of libtamd.so.
#include "common.h"

extern "C" {

int tamd_abi_version(void) { return TAMD_ABI_VERSION; }

const char* tamd_error_string(int code) {
  switch (code) {
    case TAMD_OK: return "ok";
    case TAMD_E_DTYPE: return "unsupported dtype";
    case TAMD_E_SHAPE: return "unsupported or inconsistent shape";
    case TAMD_E_ALIGN: return "pointer or stride not 16-byte aligned";
    case TAMD_E_NULL: return "required pointer is NULL";
    case TAMD_E_WORKSPACE: return "workspace too small";
    case TAMD_E_ARG: return "invalid argument";
    default: return code > 0 ? "HIP launch error" : "unknown error";
  }
}

}  // extern "C"
