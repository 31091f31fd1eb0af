// ABI bookkeeping for libr3d_hip.so
#include "common.h"
#include "../../include/r3d_hip.h"
#include <string.h>

R3D_EXPORT int r3d_abi_version(void) { return R3D_ABI_VERSION; }

R3D_EXPORT int r3d_build_info(char* buf, int cap) {
    static const char info[] = "gfx950;f32-mfma;" __DATE__ " " __TIME__;
    if (!buf || cap <= 0) return R3D_EINVAL;
    const int n = (int)sizeof(info) < cap ? (int)sizeof(info) : cap;
    memcpy(buf, info, (size_t)n);
    buf[cap - 1 < n ? cap - 1 : n - 1] = '\0';
    return R3D_OK;
}
