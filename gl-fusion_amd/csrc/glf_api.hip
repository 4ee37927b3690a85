// Error channel, lazy device init and ABI version of libglfusion_hip.so.
#include "gemm_common.h"
#include <atomic>
#include <mutex>
#include <unordered_map>

namespace glf {

int init_gemm_attrs();   // gemm_f32.hip
int init_attn_attrs();   // attn_softmax.hip
int init_gemm_s16_attrs();   // gemm_s16.hip

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

static std::atomic<int> g_precision{0};

int precision() { return g_precision.load(std::memory_order_relaxed); }

// Library-owned device memory (the ONLY allocations the library makes; include/glfusion.h documents them):
//   * one zero page of ZERO_PAGE_FLOATS floats per device (f16x3 kernels load padding / overhang rows from it),
//   * one ring of RING floats per (device, stream) for operand maxima the library measures itself (f16x3 with
//     amax_a / amax_b == NULL).  A slot is written (memset + amax kernel) and read (GEMM) on that one stream, in
//     order, so reuse after the ring wraps is ordered by the stream itself.
// Both are created lazily on first use on a device / stream and live until the process exits.
constexpr unsigned RING = 1u << 12;
struct StreamRing { float* base = nullptr; unsigned pos = 0; };
struct DeviceState {
    int cus = 0;
    float* zeros = nullptr;
    std::unordered_map<hipStream_t, StreamRing> rings;
};
static std::mutex g_mu;
static std::unordered_map<int, DeviceState> g_dev;
static thread_local int t_dev = -1;            // device the calling thread last initialised (fast path)
static thread_local DeviceState* t_state = nullptr;

static DeviceState* state() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (dev == t_dev && t_state) return t_state;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_dev.find(dev);
    if (it == g_dev.end()) return nullptr;
    t_dev = dev; t_state = &it->second;        // unordered_map nodes are address-stable
    return t_state;
}

const float* zero_page() {
    DeviceState* st = state();
    return st ? st->zeros : nullptr;
}

float* amax_scratch(int n, hipStream_t s) {
    DeviceState* st = state();
    if (!st) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    StreamRing& r = st->rings[s];
    if (!r.base) {
        if (hipMalloc(reinterpret_cast<void**>(&r.base), RING * sizeof(float)) != hipSuccess) { r.base = nullptr; return nullptr; }
    }
    if (r.pos + (unsigned)n > RING) r.pos = 0;
    float* out = r.base + r.pos;
    r.pos += (unsigned)n;
    return out;
}

int num_cus() {
    DeviceState* st = state();
    return (st && st->cus > 0) ? st->cus : 256;
}

int ensure_init() {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(GLF_ERR_LAUNCH, "hipGetDevice: %s", hipGetErrorString(e));
    if (dev == t_dev && t_state) return GLF_OK;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_dev.find(dev);
    if (it == g_dev.end()) {
        DeviceState st;
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, dev);
        if (e != hipSuccess) return fail(GLF_ERR_LAUNCH, "hipGetDeviceProperties: %s", hipGetErrorString(e));
        st.cus = prop.multiProcessorCount;
        if (int rc = init_gemm_attrs()) return rc;           // hipFuncSetAttribute is per device
        if (int rc = init_attn_attrs()) return rc;
        if (int rc = init_gemm_s16_attrs()) return rc;
        e = hipMalloc(reinterpret_cast<void**>(&st.zeros), ZERO_PAGE_FLOATS * sizeof(float));
        if (e == hipSuccess) e = hipMemset(st.zeros, 0, ZERO_PAGE_FLOATS * sizeof(float));
        if (e != hipSuccess) return fail(GLF_ERR_WORKSPACE, "hipMalloc(zero page): %s", hipGetErrorString(e));
        it = g_dev.emplace(dev, std::move(st)).first;
    }
    t_dev = dev; t_state = &it->second;
    return GLF_OK;
}

}  // namespace glf

extern "C" const char* glf_last_error(void) { return glf::err_buf(); }
extern "C" int glf_abi_version(void) { return 7; }
extern "C" int glf_init(void) { return glf::ensure_init(); }
extern "C" size_t glf_sizeof_gemm_params(void) { return sizeof(glf_gemm_params); }
extern "C" int glf_set_precision(int mode) {
    if (mode < 0 || mode > 3) return glf::fail(GLF_ERR_UNSUPPORTED, "glf_set_precision: mode must be 0 (fp32 MFMA), 1 (split-bf16 x6), 2 (split-fp16 x3) or 3 (fp16 x1)");
    glf::g_precision.store(mode);
    return GLF_OK;
}
extern "C" int glf_get_precision(void) { return glf::precision(); }
