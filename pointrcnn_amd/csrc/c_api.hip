// Status strings / version of libpointrcnn_hip.so.  The entry points themselves live
// beside their kernels (fps.hip, grouping.hip, interp.hip, nms_rotate.hip, mlp.hip).
#include "common.h"

extern "C" const char *prcnn_status_string(int status)
{
    switch (status) {
        case PRCNN_OK: return "ok";
        case PRCNN_BAD_ARG: return "invalid argument (npoint, radius and nsample must be positive)";
        case PRCNN_BAD_SHAPE: return "invalid shape or null tensor";
        case PRCNN_HIP_ERROR: return "HIP runtime / kernel launch error";
        case PRCNN_WORKSPACE: return "workspace missing, misaligned or too small";
        case PRCNN_UNSUPPORTED: return "size not supported by the gfx950 kernels";
        default: return "unknown status";
    }
}

extern "C" int prcnn_version(void) { return 100; }

static thread_local int g_last_hip_error = 0;
void prcnn_note_hip_error(int hip_status) { g_last_hip_error = hip_status; }

extern "C" const char *prcnn_last_hip_error_string(void)
{
    return hipGetErrorString((hipError_t)g_last_hip_error);
}
