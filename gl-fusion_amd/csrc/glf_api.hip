// Error channel, lazy device init and ABI version of libglfusion_hip.so.
#include "gemm_common.h"
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

// ring of device floats for operand maxima the library measures itself (f16x3 with amax_a/amax_b == NULL).
// A slot is written (memset + amax kernel) and read (GEMM) on one stream in order; it is reused RING calls later.
static float* g_ring = nullptr;
static float* g_zeros = nullptr;
const float* zero_page() { return g_zeros; }
static std::atomic<unsigned> g_ring_pos{0};
constexpr unsigned RING = 1u << 14;
float* amax_scratch(int n) {
    if (!g_ring) return nullptr;
    const unsigned p = g_ring_pos.fetch_add((unsigned)n, std::memory_order_relaxed);
    unsigned i = p % RING;
    if (i + (unsigned)n > RING) i = 0;
    return g_ring + i;
}

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
        if (g_init_rc == GLF_OK) {
            e = hipMalloc(reinterpret_cast<void**>(&g_ring), RING * sizeof(float));
            if (e != hipSuccess) { g_ring = nullptr; g_init_rc = fail(GLF_ERR_WORKSPACE, "hipMalloc(amax ring): %s", hipGetErrorString(e)); return; }
            e = hipMalloc(reinterpret_cast<void**>(&g_zeros), ZERO_PAGE_FLOATS * sizeof(float));
            if (e == hipSuccess) e = hipMemset(g_zeros, 0, ZERO_PAGE_FLOATS * sizeof(float));
            if (e != hipSuccess) { g_zeros = nullptr; g_init_rc = fail(GLF_ERR_WORKSPACE, "hipMalloc(zero page): %s", hipGetErrorString(e)); }
        }
    });
    return g_init_rc;
}

}  // namespace glf

extern "C" const char* glf_last_error(void) { return glf::err_buf(); }
extern "C" int glf_abi_version(void) { return 3; }
extern "C" int glf_init(void) { return glf::ensure_init(); }
extern "C" size_t glf_sizeof_gemm_params(void) { return sizeof(glf_gemm_params); }
extern "C" int glf_set_precision(int mode) {
    if (mode < 0 || mode > 2) return glf::fail(GLF_ERR_UNSUPPORTED, "glf_set_precision: mode must be 0 (fp32 MFMA), 1 (split-bf16 x6) or 2 (split-fp16 x3)");
    glf::g_precision.store(mode);
    return GLF_OK;
}
extern "C" int glf_get_precision(void) { return glf::precision(); }
