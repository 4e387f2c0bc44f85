// Library identity, error strings, device check.
#include <string.h>
#include "yv_common.h"

extern "C" int yv_version(void) { return 100; }

extern "C" const char* yv_error_string(int code) {
    switch (code) {
        case YV_OK: return "ok";
        case YV_ERR_ARG: return "bad argument (size / null pointer / unsupported shape)";
        case YV_ERR_LIMIT: return "exceeds a documented kernel limit";
        case YV_ERR_WORKSPACE: return "workspace too small";
        case YV_ERR_LAUNCH: return "HIP launch failed";
        default: return "unknown error";
    }
}

extern "C" int yv_device_is_gfx950(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return -1;
    return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}
