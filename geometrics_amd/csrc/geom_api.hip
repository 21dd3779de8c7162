// ABI bookkeeping for libgeom_hip.so (version + error strings).
#include "geom_common.h"

extern "C" int geom_abi_version(void) { return GEOM_ABI_VERSION; }

extern "C" const char *geom_strerror(int code)
{
    if (code == 0) return "success";
    if (code == GEOM_EINVAL) return "geom: invalid argument (negative size, null pointer or empty target set)";
    if (code == GEOM_ETOOBIG) return "geom: dimension exceeds the supported range";
    if (code == GEOM_EUNSUPPORTED) return "geom: shape outside this fast path (use the general entry point)";
    if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
    return "geom: unknown error code";
}
