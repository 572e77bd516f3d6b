// Library-level entry points of the C ABI (include/vqvae_hip.h).
#include "common.h"

extern "C" {

int vqvae_abi_version(void) { return VQVAE_HIP_ABI_VERSION; }

const char *vqvae_strerror(int code) {
    switch (code) {
        case VQVAE_OK: return "ok";
        case VQVAE_ERR_NULL: return "vqvae: a required pointer argument is NULL";
        case VQVAE_ERR_SHAPE: return "vqvae: non-positive or inconsistent dimension";
        case VQVAE_ERR_UNSUPPORTED: return "vqvae: shape not supported by the gfx950 kernels";
        case VQVAE_ERR_WORKSPACE: return "vqvae: workspace missing or too small";
        case VQVAE_ERR_OVERFLOW: return "vqvae: element count overflows the index type";
        default: break;
    }
    if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
    return "vqvae: unknown error code";
}

}  // extern "C"
