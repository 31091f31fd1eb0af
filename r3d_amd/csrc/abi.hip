// ABI bookkeeping for libr3d_hip.so
#include "common.h"
#include "../../include/r3d_hip.h"
#include <string.h>
#include <dlfcn.h>

R3D_EXPORT int r3d_abi_version(void) { return R3D_ABI_VERSION; }

R3D_EXPORT int r3d_build_info(char* buf, int cap) {
    static const char info[] = "gfx950;f32-mfma;" __DATE__ " " __TIME__;
    if (!buf || cap <= 0) return R3D_EINVAL;
    const int n = (int)sizeof(info) < cap ? (int)sizeof(info) : cap;
    memcpy(buf, info, (size_t)n);
    buf[cap - 1 < n ? cap - 1 : n - 1] = '\0';
    return R3D_OK;
}

// ---- gradient all-reduce over xGMI: ncclAllReduce(sum, fp32, in place) on the caller's stream.
// The library does not link RCCL: the process has exactly one RCCL loaded (PyTorch's, beside libtorch), and a second copy
// would not share its communicators -- the symbol is looked up in the copy that is already mapped (RTLD_NOLOAD), once.
typedef int (*r3d_nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, void*);
static r3d_nccl_allreduce_fn r3d_find_allreduce() {
    static const char* names[] = {"librccl.so", "librccl.so.1", "libnccl.so", "libnccl.so.2"};
    for (const char* n : names) {
        void* h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (h) {
            void* f = dlsym(h, "ncclAllReduce");
            if (f) return reinterpret_cast<r3d_nccl_allreduce_fn>(f);
        }
    }
    return reinterpret_cast<r3d_nccl_allreduce_fn>(dlsym(RTLD_DEFAULT, "ncclAllReduce"));
}

R3D_EXPORT int r3d_allreduce_flat(float* buf, int64_t count, void* comm, void* stream) {
    R3D_REQUIRE(buf && comm && count >= 0);
    static const r3d_nccl_allreduce_fn fn = r3d_find_allreduce();
    if (!fn) return R3D_ENORCCL;
    if (count == 0) return R3D_OK;
    const int rc = fn(buf, buf, (size_t)count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, comm, stream);
    return rc == 0 ? R3D_OK : R3D_ERCCL_BASE - rc;          // ncclResult_t r -> -(100 + r)
}
