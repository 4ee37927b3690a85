// Error channel, lazy device init and ABI version of libglfusion_hip.so.
#include "glf_common.h"
#include <atomic>
#include <mutex>

namespace glf {

int init_gemm_attrs();   // gemm_f32.hip

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

static std::atomic<int> g_cus{0};
static std::atomic<int> g_precision{0};

int precision() { return g_precision.load(std::memory_order_relaxed); }
static std::once_flag g_once;
static int g_init_rc = GLF_OK;

int num_cus() {
    int v = g_cus.load(std::memory_order_relaxed);
    return v > 0 ? v : 256;
}

int ensure_init() {
    std::call_once(g_once, [] {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) { g_init_rc = fail(GLF_ERR_LAUNCH, "hipGetDevice: %s", hipGetErrorString(e)); return; }
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, dev);
        if (e != hipSuccess) { g_init_rc = fail(GLF_ERR_LAUNCH, "hipGetDeviceProperties: %s", hipGetErrorString(e)); return; }
        g_cus.store(prop.multiProcessorCount);
        g_init_rc = init_gemm_attrs();
    });
    return g_init_rc;
}

}  // namespace glf

extern "C" const char* glf_last_error(void) { return glf::err_buf(); }
extern "C" int glf_abi_version(void) { return 2; }
extern "C" int glf_init(void) { return glf::ensure_init(); }
extern "C" size_t glf_sizeof_gemm_params(void) { return sizeof(glf_gemm_params); }
extern "C" int glf_set_precision(int mode) {
    if (mode != 0 && mode != 1) return glf::fail(GLF_ERR_UNSUPPORTED, "glf_set_precision: mode must be 0 (fp32 MFMA) or 1 (split-bf16 x6)");
    glf::g_precision.store(mode);
    return GLF_OK;
}
extern "C" int glf_get_precision(void) { return glf::precision(); }
