// Version and error strings of libmi355attn.
#include "../../include/mi355attn.h"

extern "C" int m355_version(void) { return M355_VERSION; }

extern "C" const char* m355_strerror(int code) {
    switch (code) {
        case M355_OK: return "ok";
        case M355_ERR_INVALID_ARG: return "invalid argument (null/misaligned pointer, bad size or stride)";
        case M355_ERR_UNSUPPORTED: return "unsupported head_dim, group size or dtype";
        case M355_ERR_WORKSPACE: return "workspace missing, misaligned or too small";
        case M355_ERR_LAUNCH: return "kernel launch failed (hipGetLastError)";
        case M355_ERR_SCRATCH_LIMIT: return "batch needs more scratch than the allocation-free entry point holds (M355_STATIC_SCRATCH_BYTES): use the workspace form";
        default: return "unknown error code";
    }
}
