// Context, errors, profiling hooks and byte-level helpers of libzkmi355x.so.
#include "ff.cuh"
#include "fp_inv.cuh"
#include "zk_common.h"

#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>
#include <mutex>

#include <stdlib.h>

namespace zk {

// Runs when the shared object is loaded, i.e. before the first HIP call of a process that has
// not touched the GPU yet: the pipelined prover keeps ~3 streams per proof in flight and the ROCm
// default of 4 hardware queues serialises them (measured per 2^16 proof: 8.6 ms with 4, 7.0 ms with 16, 5.7-6.1 ms with 32).
// ZK_TRACE=1 (diagnostics): a native backtrace of the thread that raises SIGABRT / SIGSEGV -- the runtime's own threads abort without a word when a
// queue dies, and Python's faulthandler only shows the main thread's Python frames.
static void zk_fatal_signal(int sig) {
    static const char msg[] = "[zk trace] fatal signal, native backtrace of the raising thread:\n";
    signal(SIGABRT, SIG_DFL);
    signal(SIGSEGV, SIG_DFL);
    (void)!write(2, msg, sizeof msg - 1);
    void* frames[64];
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    raise(sig);
}
__attribute__((constructor)) static void zk_request_hw_queues() {
    setenv("GPU_MAX_HW_QUEUES", "32", 0);
    if (getenv("ZK_TRACE")) {
        void* warm[4];
        (void)backtrace(warm, 4);          // loads the unwinder now: no allocation inside the handler
        signal(SIGABRT, zk_fatal_signal);
        signal(SIGSEGV, zk_fatal_signal);
    }
}

// ---- options: zk_set_option(name, value) first, the environment second (zk_common.h: opt)
// The public knobs -- the ones that are not test-only.  A name is accepted with or without its "ZK_" prefix, in either case.
static const char* const PUBLIC_OPTIONS[] = {
    "ZK_MSM_WINDOW",            // Pippenger window bits of keys uploaded afterwards (default: per key size, msm_auto_window / groth16.hip)
    "ZK_KEY_SUBGROUP_CHECK",    // 0: skip the [r] P = O test of key points at upload (keys checked before); folded 17-bit windows are then not used
    "ZK_SLOT_STREAMS",          // 3: three streams for every proof slot, 1: one
    "ZK_SERIAL_STREAMS",        // 1: a lone proof stays on one stream
    "ZK_GRAPH",                 // 1: capture each slot's proof into a hipGraph and replay it
    "ZK_MSM_CHUNK_MIN", "ZK_MSM_TARGET_THREADS",      // accumulate chunking
    "ZK_SORT_TWO_LEVEL", "ZK_SORT_TWO_LEVEL_MIN", "ZK_SORT_MIN_WGS", "ZK_SORT_SCALAR_MAJOR", "ZK_SORT_FINE_STAGED", "ZK_SORT_COARSE_STAGED",
    "ZK_TAIL_SLOTS", "ZK_TAIL_FIXUP_SLOTS", "ZK_FIXUP_BY_CHUNK", "ZK_DS_WIDE_GROUP", "ZK_RED_WAVES",
    "ZK_ACC_G1_GLDS", "ZK_ACC_G1_MMADD", "ZK_ACC_G2_INLINE", "ZK_MSM_BA_CURVES", "ZK_MSM_BA_ROUNDS", "ZK_MSM_API_PRECOMP",
    "ZK_DERIVE_SIDE_BY_SIDE", "ZK_FR_RNS",
    "ZK_PIN_SHARED_SORT",       // 0: every Pinocchio product sorts its own scalar vector (default: vv / vav, yy / yay, ww / waw share one sort each)
    "ZK_PIN_COMPACT_H",         // 0: Pinocchio keys uploaded afterwards keep v_all | w_all in the h pool (pinocchio.hip: the compact h pool is the default)
};
struct OptionTable {
    std::mutex mu;
    std::vector<std::pair<std::string, const char*>> set;      // name -> value (nullptr: explicitly unset, i.e. back to the environment)
    std::vector<std::string*> arena;                           // values live for ever: ZK_ENV call sites keep the pointer
};
static OptionTable& options() {
    static OptionTable* t = new OptionTable;
    return *t;
}
const char* opt(const char* name) {
    OptionTable& t = options();
    {
        std::lock_guard<std::mutex> g(t.mu);
        for (auto& kv : t.set)
            if (kv.first == name) {
                if (kv.second) return kv.second;
                break;
            }
    }
    return getenv(name);
}
static int option_set(const char* name, const char* value) {
    if (!name || !*name) ZK_FAIL(ZK_ERR_ARG, "zk_set_option: no name");
    std::string up;
    for (const char* q = name; *q; q++) up.push_back((char)((*q >= 'a' && *q <= 'z') ? *q - 32 : *q));
    if (up.compare(0, 3, "ZK_") != 0) up = "ZK_" + up;
    bool known = false;
    for (const char* k : PUBLIC_OPTIONS) known = known || up == k;
    if (!known) ZK_FAIL(ZK_ERR_ARG, "zk_set_option: unknown option name");
    OptionTable& t = options();
    std::lock_guard<std::mutex> g(t.mu);
    const char* stored = nullptr;
    if (value) {
        t.arena.push_back(new std::string(value));
        stored = t.arena.back()->c_str();
    }
    for (auto& kv : t.set)
        if (kv.first == up) { kv.second = stored; return ZK_OK; }
    t.set.emplace_back(up, stored);
    return ZK_OK;
}

// The device list and its contexts.  Contexts are heap-allocated and never destroyed (event / stream handles must not be touched at exit).
static std::vector<Ctx*>& ctxs() {
    static std::vector<Ctx*>* v = new std::vector<Ctx*>;
    return *v;
}
static thread_local int tl_vdev = 0;
Ctx& ctx() {
    std::vector<Ctx*>& v = ctxs();
    if (v.empty()) {
        static Ctx* none = new Ctx;        // before zk_init: an un-initialised placeholder (profiling level, nothing else)
        return *none;
    }
    return *v[(size_t)tl_vdev < v.size() ? tl_vdev : 0];
}
int ctx_count() { return (int)ctxs().size(); }
Ctx& ctx_at(int vdev) { return *ctxs()[vdev]; }
DeviceScope::DeviceScope(int vdev) : prev_vdev(tl_vdev), prev_dev(-1) {
    (void)hipGetDevice(&prev_dev);
    tl_vdev = vdev;
    if (vdev < ctx_count() && ctx_at(vdev).device != prev_dev) (void)hipSetDevice(ctx_at(vdev).device);
}
DeviceScope::~DeviceScope() {
    int cur = -1;
    (void)hipGetDevice(&cur);
    tl_vdev = prev_vdev;
    if (prev_dev >= 0 && cur != prev_dev) (void)hipSetDevice(prev_dev);
}
int copy_between(void* dst, int dst_vdev, const void* src, int src_vdev, size_t bytes, hipStream_t s) {
    if (!bytes) return ZK_OK;
    const int dd = ctx_at(dst_vdev).device, sd = ctx_at(src_vdev).device;
    if (dd == sd) HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
    else HIPCHK(hipMemcpyPeerAsync(dst, dd, src, sd, bytes, s));
    return ZK_OK;
}
// the most recent failure of the process (any thread: the multi-device derivation runs one host thread per device)
static std::mutex& err_mutex() { static std::mutex* m = new std::mutex; return *m; }
static std::string& err_text() { static std::string* t = new std::string; return *t; }
int set_error(int code, const char* what, const char* file, int line) {
    char buf[512];
    const char* base = strrchr(file, '/');
    snprintf(buf, sizeof buf, "%s (%s:%d)", what ? what : "", base ? base + 1 : file, line);
    std::lock_guard<std::mutex> g(err_mutex());
    err_text() = buf;
    return code;
}
std::vector<void (*)()>& cleanup_hooks() {
    static std::vector<void (*)()> h;
    return h;
}
int ensure_init() {
    if (ctx_count() > 0) return ZK_OK;
    return zk_init(0);
}

static hipEvent_t pool_event() {
    Ctx& c = ctx();
    if (!c.event_pool.empty()) { hipEvent_t e = c.event_pool.back(); c.event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
ScopedTimer::ScopedTimer(const char* name, hipStream_t stream, int level) : s(stream) {
    Ctx& c = ctx();
    if (c.profiling < level) return;
    for (size_t i = 0; i < c.timers.size(); i++)
        if (c.timers[i].name == name) idx = (int)i;
    if (idx < 0) {
        c.timers.push_back(KernelTimer{name, {}});
        idx = (int)c.timers.size() - 1;
    }
    a = pool_event();
    b = pool_event();
    (void)hipEventRecord(a, s);
}
ScopedTimer::~ScopedTimer() {
    if (idx < 0) return;
    (void)hipEventRecord(b, s);
    ctx().timers[idx].spans.push_back({a, b});
}

void profile_count(const char* name, uint64_t add) {
    Ctx& c = ctx();
    if (c.profiling < 2) return;
    for (auto& kv : c.counters)
        if (kv.first == name) { kv.second += add; return; }
    c.counters.push_back({name, add});
}

// one multiplication chain per lane; all 256 CUs x 8 waves/SIMD busy
__global__ void k_bench_mul_fr(uint32_t* out, uint32_t iters) {
    Fr x, y;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        x.v[i] = FR_R1[i] ^ (threadIdx.x * 2654435761u >> (i & 7));
        y.v[i] = FR_R2[i] + blockIdx.x;
    }
    x.v[7] &= 0x0fffffffu;
    y.v[7] &= 0x0fffffffu;
    for (uint32_t i = 0; i < iters; i++) {
        x = fe_mul_inline(x, y);
        y = fe_mul_inline(y, x);
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) acc ^= x.v[i] ^ y.v[i];
    if (acc == 0x12345678u) out[0] = acc;   // keep the chain live
}
__global__ void k_bench_mul_fp(uint32_t* out, uint32_t iters) {
    FpB<2> x, y;
#pragma unroll
    for (int i = 0; i < FPL; i++) {
        x.v[i] = (FP29_R1[i] ^ (threadIdx.x * 2654435761u >> (i & 7))) & FP29_MASK;
        y.v[i] = (FP29_R2[i] + blockIdx.x) & FP29_MASK;
    }
    x.v[FPL - 1] &= 7;
    y.v[FPL - 1] &= 7;
    for (uint32_t i = 0; i < iters; i++) {
        x = fe_mul_inline(x, y);
        y = fe_mul_inline(y, x);
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < FPL; i++) acc ^= x.v[i] ^ y.v[i];
    if (acc == 0x12345678u) out[0] = acc;
}

// Field-layer self-test (tests/test_gpu_field.py): every lane evaluates a battery of base-field expressions on its
// operand pair through the SAME lazy-reduction code paths the group law uses (bounded adds / subs, products,
// squares, fused double products, the zero test on unreduced values, inversion, full reduction, lane-pair Fp2)
// and stores fully reduced plain integers (dense words) for comparison with host big-integer arithmetic.
static constexpr int FP_SELFTEST_OUTS = 23;
__global__ void k_fp_selftest(uint32_t* __restrict__ out, const uint32_t* __restrict__ a_words, const uint32_t* __restrict__ b_words, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;                      // n is even and pairs (2k, 2k+1) stay together: the lane-pair ops below are uniform
    FpWords aw, bw;
#pragma unroll
    for (int l = 0; l < 12; l++) { aw.w[l] = a_words[12 * i + l]; bw.w[l] = b_words[12 * i + l]; }
    const FpB<2> a = fp_to_mont(aw), b = fp_to_mont(bw);
    uint32_t* o = out + (uint64_t)FP_SELFTEST_OUTS * 12 * i;
    auto put = [&](int k, const FpWords& w) {
#pragma unroll
        for (int l = 0; l < 12; l++) o[12 * k + l] = w.w[l];
    };
    put(0, fp_from_mont(fe_add(a, b)));
    put(1, fp_from_mont(fe_sub(a, b)));
    put(2, fp_from_mont(fe_mul(a, b)));
    put(3, fp_from_mont(fe_sqr(a)));
    put(4, fp_from_mont(fe_neg(a)));
    put(5, fp_from_mont(fe_inv(a)));
    put(6, fp_from_mont(fe_dbl(fe_dbl(fe_dbl(a)))));                                   // 8a through three lazy doublings
    const auto d1 = fe_sub(fe_sub(fe_sub(a, b), b), fe_add(a, a));                      // -a - 2b through a chain of growing bounds
    put(7, fp_from_mont(d1));
    put(8, fp_from_mont(fe_mul(d1, fe_sub(b, a))));                                     // product of two lazily reduced values
    put(9, fp_from_mont(fe_mul_sub(a, b, fe_add(a, b), fe_sub(a, b))));                 // a b - (a + b)(a - b), fused form
    FpWords flags;
#pragma unroll
    for (int l = 0; l < 12; l++) flags.w[l] = 0;
    const auto z1 = fe_sub(fe_add(a, b), fe_add(b, a));                                 // a multiple of p that is not all-zero limbs
    const auto z2 = fe_sub(fe_mul(a, b), fe_mul(b, a));
    flags.w[0] = (fe_is_zero(z1) ? 1u : 0u) | (fe_is_zero(z2) ? 2u : 0u) | (fe_is_zero(a) ? 4u : 0u) | (fe_eq(fe_add(a, b), fe_add(b, a)) ? 8u : 0u) |
                 (fe_eq(a, b) ? 16u : 0u) | (fe_is_zero(fe_sub(a, a)) ? 32u : 0u);
    put(10, flags);
    put(11, fp_pack(fp_canon(fe_add(fe_neg(a), a))));                                   // 0 as a reduced multiple of p
    // lane pair (2k, 2k+1) = one Fp2 value x = a_even + a_odd u, y = b_even + b_odd u
    const Fp2HB<2> x{a}, y{b};
    put(12, fp_from_mont(fe_mul(x, y).v));
    put(13, fp_from_mont(fe_sqr(x).v));
    put(14, fp_from_mont(fe_mul(fe_sub(x, y), fe_add(x, y)).v));
    put(15, fp_from_mont(fe_mul_inline(a, b)));
    put(16, fp_from_mont(fe_inv_fast(fe_add(fe_dbl(a), fe_neg(a)))));                  // lockstep inversion (fp_inv.cuh) of a lazily reduced a
    put(17, fp_from_mont(fe_inv_fast(x).v));                                            // ... and of the lane pair's Fp2 value
    // round 3: a - b - 2 c in one carry pass (X3 of every group addition), on products and on lazily reduced operands at the at-rest bound
    put(18, fp_from_mont(fe_sub_sub_dbl(fe_sqr(a), fe_mul(a, b), fe_mul(b, fe_add(a, b)))));                    // a^2 - a b - 2 b (a + b)
    const Fp big = Fp(fe_sub(fe_dbl(fe_dbl(fe_dbl(fe_dbl(a)))), b));                                             // 16 a - b as a value < 64 p with unreduced limbs
    put(19, fp_from_mont(fe_sub_sub_dbl(a, big, big)));                                                          // a - 3 (16 a - b)
    // the fused double product with its lazily negated factor (fe_neg_lazy: no carry pass) at the at-rest bound: a b - (16 a - b) b
    put(20, fp_from_mont(fe_mul_sub(a, b, big, b)));
    // lane-pair Fp2 product whose left operand is lazily reduced at the at-rest bound: (16 x - y) * y
    put(21, fp_from_mont(fe_mul(Fp2H(big), y).v));
    put(22, fp_from_mont(fe_sqr(Fp2H(big)).v));                                                                 // lane-pair square with its lazy difference, at the at-rest bound
}

}  // namespace zk

using namespace zk;

extern "C" {

const char* zk_strerror(int code) {
    switch (code) {
        case ZK_OK: return "ok";
        case ZK_ERR_ARG: return "invalid argument";
        case ZK_ERR_NOT_ON_CURVE: return "point not on curve";
        case ZK_ERR_SCALAR_RANGE: return "field element not canonical (>= modulus)";
        case ZK_ERR_REMAINDER: return "witness does not satisfy the circuit: p mod Z != 0 (QAP.ml:134)";
        case ZK_ERR_HIP: return "HIP runtime error";
        case ZK_ERR_APPLY_POWERS: return "apply_powers: fewer points than coefficients (curve.ml:116)";
        case ZK_ERR_HANDLE: return "unknown handle";
        case ZK_ERR_DOMAIN: return "domain mismatch (curve.ml:96-100)";
        default: return "unknown error";
    }
}
const char* zk_last_error(void) {
    static thread_local std::string copy;
    std::lock_guard<std::mutex> g(err_mutex());
    copy = err_text();
    return copy.c_str();
}

int zk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// The set-up half of zk_init / zk_set_device_list: one context per list entry.
static int contexts_create(const int32_t* devices, uint32_t count) {
    int n = zk_device_count();
    if (n <= 0) ZK_FAIL(ZK_ERR_HIP, "no HIP device visible: the MI355X path has no CPU fallback");
    if (!devices || count == 0 || count > 64) ZK_FAIL(ZK_ERR_ARG, "device list: 1 .. 64 entries");
    for (uint32_t i = 0; i < count; i++)
        if (devices[i] < 0 || devices[i] >= n) ZK_FAIL(ZK_ERR_ARG, "device index out of range");
    int prev = -1;
    (void)hipGetDevice(&prev);
    std::vector<Ctx*> made;
    auto fail = [&](int rc) {
        for (Ctx* c : made) {          // nothing of a half-built list survives: its streams and events go with it
            if (c->device >= 0 && hipSetDevice(c->device) == hipSuccess) {
                if (c->stream) (void)hipStreamDestroy(c->stream);
                if (c->stream2) (void)hipStreamDestroy(c->stream2);
                if (c->stream3) (void)hipStreamDestroy(c->stream3);
                if (c->ev_join3) (void)hipEventDestroy(c->ev_join3);
                if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
                if (c->ev_join) (void)hipEventDestroy(c->ev_join);
            }
            delete c->bufs;
            delete c;
        }
        if (prev >= 0) (void)hipSetDevice(prev);
        return rc;
    };
    for (uint32_t i = 0; i < count; i++) {
        Ctx* c = new Ctx;
        made.push_back(c);
        c->vdev = (int)i;
        c->device = devices[i];
        c->bufs = new CtxBufs;
        if (hipSetDevice(devices[i]) != hipSuccess) return fail(set_error(ZK_ERR_HIP, "hipSetDevice failed", __FILE__, __LINE__));
        bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) == hipSuccess &&
                  hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&c->ev_join3, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) == hipSuccess;
        if (!ok) return fail(set_error(ZK_ERR_HIP, "creating the context's streams failed", __FILE__, __LINE__));
        c->profiling = ctx().profiling;
        c->inited = true;
    }
    // peer access between every pair of distinct HIP devices of the list: the slices of a proof's scalars and its partial sums then travel device to
    // device over xGMI (hipMemcpyPeerAsync falls back to staging through the host where a pair has no peer link)
    for (uint32_t i = 0; i < count; i++)
        for (uint32_t j = 0; j < count; j++) {
            if (devices[i] == devices[j]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[i], devices[j]) == hipSuccess && can) {
                (void)hipSetDevice(devices[i]);
                const hipError_t e = hipDeviceEnablePeerAccess(devices[j], 0);
                if (e != hipSuccess) (void)hipGetLastError();          // already enabled: fine
            }
        }
    (void)hipSetDevice(devices[0]);          // the list's first device is the process's current device from here on (legacy single-device calls)
    ctxs() = made;
    tl_vdev = 0;
    return ZK_OK;
}

int zk_init(int device) {
    if (ctx_count() > 0) {
        if (ctx_at(0).device == device) return ZK_OK;
        ZK_FAIL(ZK_ERR_ARG, "zk_init: already bound to another device (zk_set_devices / zk_set_device_list change the list)");
    }
    const int32_t d = device;
    return contexts_create(&d, 1);
}

int zk_shutdown(void) {
    if (ctx_count() == 0) return ZK_OK;
    for (int v = 0; v < ctx_count(); v++) {
        DeviceScope ds(v);
        (void)hipDeviceSynchronize();
    }
    zk_profile_reset();
    for (int v = ctx_count() - 1; v >= 0; v--) {          // hooks see every context once, as the current one (keys are released by the first call)
        DeviceScope ds(v);
        for (auto f : zk::cleanup_hooks()) f();
        Ctx& c = ctx_at(v);
        for (hipEvent_t e : c.event_pool) (void)hipEventDestroy(e);
        c.event_pool.clear();
        (void)hipEventDestroy(c.ev_fork);
        (void)hipEventDestroy(c.ev_join);
        (void)hipStreamDestroy(c.stream);
        (void)hipStreamDestroy(c.stream2);
        (void)hipStreamDestroy(c.stream3);
        (void)hipEventDestroy(c.ev_join3);
        delete c.bufs;
        c.bufs = nullptr;
        c.inited = false;
    }
    const int level = ctx_at(0).profiling;
    for (Ctx* c : ctxs()) delete c;
    ctxs().clear();
    tl_vdev = 0;
    ctx().profiling = level;
    return ZK_OK;
}

int zk_set_device_list(const int32_t* devices, uint32_t count) {
    if (!devices || count == 0) ZK_FAIL(ZK_ERR_ARG, "zk_set_device_list: empty list");
    if (ctx_count() == (int)count) {
        bool same = true;
        for (uint32_t i = 0; i < count; i++) same = same && ctx_at((int)i).device == devices[i];
        if (same) return ZK_OK;
    }
    if (zk::live_key_handles() != 0) ZK_FAIL(ZK_ERR_ARG, "zk_set_device_list: free every key handle before changing the device list");
    // a list that cannot be built must leave the current one in place: check it before the old contexts go
    const int n = zk_device_count();
    if (n <= 0) ZK_FAIL(ZK_ERR_HIP, "no HIP device visible: the MI355X path has no CPU fallback");
    if (count > 64) ZK_FAIL(ZK_ERR_ARG, "device list: 1 .. 64 entries");
    for (uint32_t i = 0; i < count; i++)
        if (devices[i] < 0 || devices[i] >= n) ZK_FAIL(ZK_ERR_ARG, "device index out of range");
    // The contexts own per-device tables the cleanup hooks release (twiddles, generator tables), so the old list has to go before the new one is
    // built; should building it fail all the same (a HIP error in hipSetDevice or a stream creation), the previous list is put back -- a later call
    // must not find the library unbound and silently bind device 0.
    std::vector<int32_t> before;
    for (int v = 0; v < ctx_count(); v++) before.push_back(ctx_at(v).device);
    ZKCHK(zk_shutdown());
    const int rc = contexts_create(devices, count);
    if (rc != ZK_OK && !before.empty()) {
        const std::string why = zk_last_error();
        if (contexts_create(before.data(), (uint32_t)before.size()) == ZK_OK) return set_error(rc, ("the new device list could not be built; the previous one is back in place: " + why).c_str(), __FILE__, __LINE__);
    }
    return rc;
}
int zk_set_devices(uint64_t mask) {
    int32_t list[64];
    uint32_t n = 0;
    for (int d = 0; d < 64; d++)
        if (mask >> d & 1) list[n++] = d;
    if (n == 0) ZK_FAIL(ZK_ERR_ARG, "zk_set_devices: empty mask");
    return zk_set_device_list(list, n);
}
int zk_get_device_list(int32_t* devices, uint32_t capacity, uint32_t* count) {
    if (count) *count = (uint32_t)ctx_count();
    if (!devices) return ZK_OK;
    if (capacity < (uint32_t)ctx_count()) ZK_FAIL(ZK_ERR_ARG, "zk_get_device_list: buffer too small");
    for (int v = 0; v < ctx_count(); v++) devices[v] = ctx_at(v).device;
    return ZK_OK;
}

int zk_set_option(const char* name, const char* value) { return option_set(name, value); }
int zk_sync(void) {
    // every stream of the process on every device of the list: the context streams AND the per-slot streams of every key (proofs in flight run on
    // hipStreamNonBlocking slot streams; zk_profile_get reads events recorded there)
    for (int v = 0; v < ctx_count(); v++) {
        bool seen = false;
        for (int u = 0; u < v; u++) seen = seen || ctx_at(u).device == ctx_at(v).device;
        if (seen) continue;
        DeviceScope ds(v);
        HIPCHK(hipDeviceSynchronize());
    }
    return ZK_OK;
}

int zk_profile_enable(int on) {
    const int level = on < 0 ? 0 : (on > 2 ? 2 : on);
    ctx().profiling = level;
    for (int v = 0; v < ctx_count(); v++) ctx_at(v).profiling = level;
    return ZK_OK;
}
int zk_profile_reset(void) {
    for (int v = 0; v < ctx_count(); v++) {
        Ctx& c = ctx_at(v);
        for (auto& t : c.timers)
            for (auto& sp : t.spans) {
                c.event_pool.push_back(sp.first);
                c.event_pool.push_back(sp.second);
            }
        c.timers.clear();
        c.counters.clear();
    }
    return ZK_OK;
}
int zk_profile_counter(const char* name, uint64_t* value) {
    if (!name || !value) ZK_FAIL(ZK_ERR_ARG, "zk_profile_counter: null");
    *value = 0;
    for (int v = 0; v < ctx_count(); v++)
        for (auto& kv : ctx_at(v).counters)
            if (kv.first == name) *value += kv.second;
    return ZK_OK;
}
int zk_profile_get(const char* family, double* total_ms, uint64_t* launches) {
    if (!family) ZK_FAIL(ZK_ERR_ARG, "zk_profile_get: null");
    ZKCHK(zk_sync());
    double tot = 0;
    uint64_t cnt = 0;
    for (int v = 0; v < ctx_count(); v++) {
        DeviceScope ds(v);
        for (auto& t : ctx_at(v).timers) {
            if (t.name != family) continue;
            for (auto& sp : t.spans) {
                float ms = 0;
                HIPCHK(hipEventElapsedTime(&ms, sp.first, sp.second));
                tot += ms;
                cnt++;
            }
        }
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = cnt;
    return ZK_OK;
}
int zk_profile_names(char* buf, size_t buflen) {
    std::string s;
    for (int v = 0; v < ctx_count(); v++)
        for (auto& t : ctx_at(v).timers) {
            if (("," + s + ",").find("," + t.name + ",") != std::string::npos) continue;
            if (!s.empty()) s += ",";
            s += t.name;
        }
    if (!buf || buflen <= s.size()) ZK_FAIL(ZK_ERR_ARG, "zk_profile_names: buffer too small");
    memcpy(buf, s.c_str(), s.size() + 1);
    return ZK_OK;
}

int zk_selftest_fp(const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out) {
    if (!a || !b || !out || !n || (n & 1)) ZK_FAIL(ZK_ERR_ARG, "zk_selftest_fp: need an even number of operand pairs");
    ZKCHK(ensure_init());
    Ctx& c = ctx();
    DevBuf da, db, dout;
    ZKCHK(da.alloc(48 * n));
    ZKCHK(db.alloc(48 * n));
    ZKCHK(dout.alloc((size_t)FP_SELFTEST_OUTS * 48 * n));
    HIPCHK(hipMemcpyAsync(da.p, a, 48 * n, hipMemcpyHostToDevice, c.stream));
    HIPCHK(hipMemcpyAsync(db.p, b, 48 * n, hipMemcpyHostToDevice, c.stream));
    hipLaunchKernelGGL(k_fp_selftest, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c.stream, dout.as<uint32_t>(), (const uint32_t*)da.as<uint32_t>(),
                       (const uint32_t*)db.as<uint32_t>(), (uint32_t)n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dout.p, (size_t)FP_SELFTEST_OUTS * 48 * n, hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    return ZK_OK;
}
int zk_bench_field_mul(int kind, uint32_t iters, double* gmul_per_s) {
    ZKCHK(ensure_init());
    Ctx& c = ctx();
    DevBuf out;
    ZKCHK(out.alloc(64));
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    // kind bit 2 set: ONE wave on the whole chip -> dependent-chain latency instead of throughput.
    // kind bit 3 set: the best throughput over 2 / 4 / 6 / 8 waves per SIMD (256-thread blocks = 4 waves, 256 CUs x 4 SIMDs): the chip's
    // multiplier ceiling whatever occupancy reaches it (the dependent chain tops out at 6 waves per SIMD: scripts/proto/fp28_proto.hip).
    const bool one_wave = (kind & 4) != 0, sweep = (kind & 8) != 0 && !one_wave;
    kind &= 3;
    double best = 0;
    const unsigned wps_list[4] = {8, 2, 4, 6};
    for (int cfg = 0; cfg < (sweep ? 4 : 1); cfg++) {
        const unsigned blocks = one_wave ? 1 : 256 * wps_list[cfg], threads = one_wave ? 64 : 256;
        for (int rep = 0; rep < 2; rep++) {
            HIPCHK(hipEventRecord(a, c.stream));
            if (kind == 0) hipLaunchKernelGGL(k_bench_mul_fr, dim3(blocks), dim3(threads), 0, c.stream, out.as<uint32_t>(), iters);
            else hipLaunchKernelGGL(k_bench_mul_fp, dim3(blocks), dim3(threads), 0, c.stream, out.as<uint32_t>(), iters);
            HIPCHK(hipEventRecord(b, c.stream));
            HIPCHK(hipStreamSynchronize(c.stream));
        }
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, a, b));
        const double rate = 2.0 * iters * blocks * threads / (ms * 1e-3) / 1e9;
        if (rate > best) best = rate;
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    if (gmul_per_s) *gmul_per_s = best;
    return ZK_OK;
}

// ---- to_compressed_bytes (curve.ml:199,208) from the uncompressed canonical encoding: pure byte logic
static const uint8_t P_MINUS1_HALF_BE[48] = {
    0x0d, 0x00, 0x88, 0xf5, 0x1c, 0xbf, 0xf3, 0x4d, 0x25, 0x8d, 0xd3, 0xdb, 0x21, 0xa5, 0xd6, 0x6b,
    0xb2, 0x3b, 0xa5, 0xc2, 0x79, 0xc2, 0x89, 0x5f, 0xb3, 0x98, 0x69, 0x50, 0x7b, 0x58, 0x7b, 0x12,
    0x0f, 0x55, 0xff, 0xff, 0x58, 0xa9, 0xff, 0xff, 0xdc, 0xff, 0x7f, 0xff, 0xff, 0xff, 0xd5, 0x55};
static int be48_gt_half(const uint8_t* y) { return memcmp(y, P_MINUS1_HALF_BE, 48) > 0; }
static int be48_is_zero(const uint8_t* y) {
    uint8_t o = 0;
    for (int i = 0; i < 48; i++) o |= y[i];
    return o == 0;
}
int zk_g1_compress(const uint8_t in[96], uint8_t out[48]) {
    if (!in || !out) ZK_FAIL(ZK_ERR_ARG, "zk_g1_compress: null");
    if (in[0] & 0x40) { memset(out, 0, 48); out[0] = 0xC0; return ZK_OK; }
    memcpy(out, in, 48);
    out[0] |= 0x80;
    if (be48_gt_half(in + 48)) out[0] |= 0x20;
    return ZK_OK;
}
int zk_g2_compress(const uint8_t in[192], uint8_t out[96]) {
    if (!in || !out) ZK_FAIL(ZK_ERR_ARG, "zk_g2_compress: null");
    if (in[0] & 0x40) { memset(out, 0, 96); out[0] = 0xC0; return ZK_OK; }
    memcpy(out, in, 96);
    out[0] |= 0x80;
    const uint8_t* y1 = in + 96;
    const uint8_t* y0 = in + 144;
    int larger = be48_is_zero(y1) ? be48_gt_half(y0) : be48_gt_half(y1);
    if (larger) out[0] |= 0x20;
    return ZK_OK;
}

}  // extern "C"
