// Host-side plumbing shared by the translation units of libzkmi355x.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string>
#include <vector>

#include "../../include/zkmi355x.h"
#include "zk_err.h"

namespace zk {

struct KernelTimer {      // HIP-event timing of one named kernel family, live inside the library
    std::string name;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> spans;
};

// One context per VIRTUAL device of the library's device list (zk_set_devices / zk_set_device_list; a process that never calls them has the
// one-entry list zk_init made).  A virtual device is an index into that list; the list may name a HIP device more than once (several shards of a
// multi-device key on one card: how a one-GPU box exercises the multi-device path).  Streams, twiddles, generator tables, event pools and timers are per
// context; keys are bound to the context they were built under.
struct Ctx {
    bool inited = false;
    int vdev = 0;                      // index into the device list
    int device = -1;                   // HIP device of this context
    hipStream_t stream = nullptr;      // main stream (G1 work, Fr stage)
    hipStream_t stream2 = nullptr;     // G2 MSM
    hipStream_t stream3 = nullptr;     // second G1 MSM: the shallow single-wave tails of one product overlap the bulk of another
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join3 = nullptr;
    // twiddles: heap layout, level k (NTT size 2^k) at offset 2^(k-1); fwd = w^j, inv = w^-j
    void* tw_fwd = nullptr;
    void* tw_inv = nullptr;
    uint32_t tw_log = 0;
    struct CtxBufs* bufs = nullptr;    // device tables owned by this context (ntt.hip twiddles, msm.hip generator tables): CtxBufs below
    int profiling = 0;                 // 0 off, 1 = the dominant (accumulate) kernels only, 2 = every family (the same level on every context)
    std::vector<hipEvent_t> event_pool; // recycled events: hipEventCreate costs tens of microseconds
    std::vector<KernelTimer> timers;
    std::vector<std::pair<std::string, uint64_t>> counters;   // work counters collected with profiling level 2 (zk_profile_counter)
};
Ctx& ctx();                              // the context of the calling thread's CURRENT virtual device (0 unless a DeviceScope is open)
int ctx_count();                         // entries of the device list (0 before zk_init / zk_set_devices)
Ctx& ctx_at(int vdev);
// Makes virtual device `vdev` current for the calling thread (hipSetDevice + the library's own index) until the scope closes.
struct DeviceScope {
    int prev_vdev, prev_dev;
    explicit DeviceScope(int vdev);
    ~DeviceScope();
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};
int ensure_init();
uint64_t live_key_handles();             // Groth16 + Pinocchio key handles alive (groth16.hip / pinocchio.hip): the device list may only change at 0
std::vector<void (*)()>& cleanup_hooks();   // run by zk_shutdown once per context (that context current), before its streams die
struct CleanupRegistrar { explicit CleanupRegistrar(void (*f)()) { cleanup_hooks().push_back(f); } };
// device -> device copy between two virtual devices on `s` -- a stream of the CALLING thread's current device, which may be the source's or the
// destination's (HIP takes a stream of either side for a peer copy; groth16_multi.hip pushes partial sums on the source shard's stream and pulls
// scalar slices on the destination's): hipMemcpyPeerAsync over xGMI when the HIP devices differ, a plain device copy when both shards sit on one card
int copy_between(void* dst, int dst_vdev, const void* src, int src_vdev, size_t bytes, hipStream_t s);

#define HIPCHK(expr)                                                                      \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) return ::zk::set_error(ZK_ERR_HIP, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// RAII device buffer
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    int alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) { p = nullptr; return set_error(ZK_ERR_HIP, hipGetErrorString(e), __FILE__, __LINE__); }
        bytes = n;
        return ZK_OK;
    }
    void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// per-context device tables (released by the owning translation unit's cleanup hook)
struct CtxBufs {
    DevBuf tw_fwd, tw_inv, inv_pow2;      // ntt.hip
    DevBuf pow2[2];                       // msm.hip: 2^k * generator, both curves (fixed_base_mul)
};

// Scoped kernel-family timer (no-op unless profiling is on)
struct ScopedTimer {
    int idx = -1;
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t s;
    ScopedTimer(const char* name, hipStream_t stream, int level = 2);
    ~ScopedTimer();
};

void profile_count(const char* name, uint64_t add);      // no-op unless profiling level 2 is on

// A tuning knob by its name ("ZK_MSM_WINDOW"): the value given through zk_set_option (the C-ABI's own configuration call: a host that cannot
// set the process environment reliably after dlopen -- an OCaml program -- uses that), else the environment variable of the same name, else nullptr.
// The returned pointer stays valid for the life of the process (option values are never freed or overwritten in place).  zk_api.hip.
const char* opt(const char* name);
// Knobs are read ONCE per process (tuning switches; a lookup per launch is a lock + string scan on the proof's hot path): a C++11
// function-local static per call site, initialised thread-safely by the language.  zk_set_option therefore acts on what has not run yet: set
// options before the first key is uploaded.
#define ZK_ENV(name) ([]() -> const char* { static const char* const v = ::zk::opt(name); return v; }())
// The kernel-FORM switches of a proof (msm.hip msm_reduce_mixed: ZK_TAIL_SLOTS, ZK_TAIL_FIXUP_SLOTS, ZK_FIXUP_BY_CHUNK, ZK_DS_WIDE_GROUP; msm_sort_accumulate_many: ZK_SORT_FINE_STAGED, ZK_SORT_COARSE_STAGED; msm_red.hip: ZK_RED_WAVES; msm_acc_g1/g2.hip:
// ZK_ACC_G1_GLDS, ZK_ACC_G1_MMADD, ZK_ACC_G2_INLINE; groth16.hip: ZK_GRAPH) are cached like every other knob -- a proof costs no getenv at all --
// unless the process was started with ZK_TEST_FORMS=1 (tests/conftest.py sets it): then they are read per call, so that the GPU suite can hold every
// form to the oracle inside one process.  forms_live() is that one cached test-mode flag.
static inline bool forms_live() {
    static const bool live = [] { const char* e = getenv("ZK_TEST_FORMS"); return e && atoi(e) != 0; }();
    return live;
}
#define ZK_FORM_ENV(name) (::zk::forms_live() ? ::zk::opt(name) : ZK_ENV(name))

static inline uint32_t ceil_log2(uint64_t x) {
    uint32_t l = 0;
    while (((uint64_t)1 << l) < x) l++;
    return l;
}

// ---- ntt.hip
int ntt_ensure_twiddles(uint32_t log_n);
// In-place radix-2 stages over `total` Montgomery Fr elements viewed as independent contiguous
// segments of 2^log_len: forward = DIF (natural -> bit-reversed), inverse = DIT (bit-reversed ->
// natural), inverse scaled by 2^-log_len when `scale` is set.
int ntt_forward(void* d_data, uint64_t total, uint32_t log_len, hipStream_t s);
int ntt_inverse(void* d_data, uint64_t total, uint32_t log_len, bool scale, hipStream_t s);
int ntt_mul_table(void* work, uint64_t total, uint32_t log_len, const void* tab, uint64_t tab_mask, bool scale,
                  const void* pad_src, void* add_dst, hipStream_t s, bool tab_is_data = false);
int fr_to_factor(void* d_buf, uint64_t n, hipStream_t s);   // transform output -> pointwise-table form (x 32, see fr29.cuh)
int fr_bitrev_permute(void* d_data, uint64_t total, uint32_t log_len, hipStream_t s);
int fr_to_mont(void* d_dst, const void* d_src, uint64_t n, int* d_flag_noncanonical, hipStream_t s);
int fr_from_mont(void* d_dst, const void* d_src, uint64_t n, hipStream_t s);
int fr_pointwise_mul(void* d_out, const void* d_a, const void* d_b, uint64_t n, hipStream_t s);

}  // namespace zk
