// Radix-2 NTT over BLS12-381 Fr for gfx950.
//
// Value-identical to the reference's FFT.Make(F).gen_fft (src/lib/zk/FFT.ml:29-67) with
// zeta_N = w^(2^32/N), w = 5^((r-1)/2^32) (FFT.ml:208-232): out[k] = sum_j a_j zeta_N^(jk).
//
// MI355X design: the log N butterfly stages are grouped into passes; each pass pulls tiles of
// 1024 elements into LDS as 9 x 29-bit limbs (fr29.cuh; 36 KiB) in limb-major (SoA) layout -- lanes walk consecutive dwords, so
// ds_read_b32 / ds_write_b32 are conflict-free for every butterfly distance >= 32 and 2-way at
// worst below -- runs up to 10 stages there, and writes back, so HBM sees ceil(log N / 8..10)
// round trips instead of log N.  Strided passes fetch rows of C >= 4 consecutive elements
// (>= 128 contiguous bytes).  Forward = DIF (natural -> bit-reversed), inverse = DIT
// (bit-reversed -> natural): convolutions never pay a permutation; only the public zk_fr_ntt
// (natural in/out like FFT.ml) adds one.  Twiddles come from per-level tables (level k holds
// zeta_{2^k}^j contiguously), which the late, small-stride stages hit in L2.
#include "ntt_lds.cuh"
#include "zk_common.h"

namespace zk {


__device__ static const uint32_t OMEGA_MONT[8] = {0x0c17f47cu, 0x9cab6d5cu, 0xfd4b71e5u, 0x1ce1e93du,
                                                  0x471dd505u, 0x0d6db230u, 0x743a3b6au, 0x3f0ee990u};
__device__ static const uint32_t OMEGA_INV_MONT[8] = {0xb3082d19u, 0x55a9e082u, 0xc7dc4a13u, 0x082f90b2u,
                                                      0xc76b052cu, 0x76ce3accu, 0x6e54185du, 0x15c39d95u};

FF_INLINE Fr fr_c32() {                            // 32 in Montgomery form: x_R * 32 = x * 2^261
    Fr c;
#pragma unroll
    for (int l = 0; l < 8; l++) c.v[l] = FR29_C32[l];
    return c;
}
// tw[2^(k-1) + j] = w_{2^k}^(+-j), j < 2^(k-1), k = 1..K.  Entry 0 unused.
__global__ void k_gen_twiddles(uint32_t* tw, uint32_t log_k, int inverse) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t total = (uint64_t)1 << log_k;
    if (i >= total || i == 0) return;
    uint32_t k = 64 - __builtin_clzll(i);          // level: 2^(k-1) <= i < 2^k
    uint32_t j = (uint32_t)(i - ((uint64_t)1 << (k - 1)));
    // exponent of the 2^32-th root: j * 2^(32-k) < 2^31
    uint32_t e = j << (32 - k);
    Fr base, acc = fe_one<FrParams>();
#pragma unroll
    for (int l = 0; l < 8; l++) base.v[l] = inverse ? OMEGA_INV_MONT[l] : OMEGA_MONT[l];
    for (int b = 0; b < 32; b++) {
        if ((e >> b) & 1) acc = fe_mul(acc, base);
        base = fe_sqr(base);
    }
    fe_store<FrParams>(tw + 8 * i, fe_mul(acc, fr_c32()));                // w * 2^261: the factor form of fr29.cuh
}

struct PassArgs {
    uint32_t log_hi;     // log2 of the largest butterfly span (2 * half) handled by this pass
    uint32_t s;          // stages in this pass
    uint32_t log_T;      // log2 tile elements
    uint32_t log_RS;     // log2 of the LDS distance of one row step
    uint32_t strided;    // rows come from stride-L global addresses
};

// Optional fused edges of a pass (the basis-conversion levels of the Fr stage):
//   pad_half : element i of a 2^log_len node is read as src[i + len/2] for i < len/2 and as 0 above
//              (the zero-padded upper half of the node, taken straight from the coefficient array)
//   add_low  : the stored value gets lo[i] added for i < len/2 (the node's lower half) and goes to dst
struct PassIO {
    const uint32_t* src;     // nullptr: read `data`
    uint32_t* dst;           // nullptr: write `data`
    const uint32_t* lo;      // add_low source
    uint32_t log_len;
    uint32_t pad_half, add_low;
};
template <bool INVERSE>
__global__ __launch_bounds__(NTT_THREADS) void k_ntt_pass(uint32_t* __restrict__ data, const uint32_t* __restrict__ tw,
                                                         PassArgs a, const uint32_t* __restrict__ scale, PassIO io) {
    __shared__ NttTile lds;
    const uint32_t T = 1u << a.log_T;
    const uint32_t log_L = a.log_hi - a.s;
    const uint32_t RSm = (1u << a.log_RS) - 1;
    const uint64_t tile = blockIdx.x;
    uint64_t base;      // global index of tile element 0
    uint32_t c0 = 0;
    if (a.strided) {
        uint32_t log_tiles_per_span = log_L - a.log_RS;                 // L / C
        uint64_t q = tile >> log_tiles_per_span;
        c0 = (uint32_t)(tile & ((1u << log_tiles_per_span) - 1)) << a.log_RS;
        base = (q << a.log_hi) + c0;
    } else {
        base = tile << a.log_T;
    }
    auto gidx = [&](uint32_t e) -> uint64_t {
        return a.strided ? base + ((uint64_t)(e >> a.log_RS) << log_L) + (e & RSm) : base + e;
    };
    const uint64_t half = io.log_len ? (uint64_t)1 << (io.log_len - 1) : 0, lenm = (half << 1) - 1;
    const uint32_t* src = io.src ? io.src : data;
    uint32_t* dst = io.dst ? io.dst : data;
    for (uint32_t e = threadIdx.x; e < T; e += NTT_THREADS) {
        const uint64_t g = gidx(e);
        Fr9 x;
        if (io.pad_half) x = (g & lenm) < half ? fr9_load(src + 8 * (g + half)) : fr9_zero();
        else x = fr9_load(src + 8 * g);
        fr9_lds_put(lds, e, x);
    }
    __syncthreads();
    lds_ntt_stages<INVERSE>(lds, tw, T, a.s, a.log_RS, log_L, c0, a.strided != 0);
    Fr9 sc;
    if (scale) sc = fr9_load(scale);
    for (uint32_t e = threadIdx.x; e < T; e += NTT_THREADS) {
        Fr9 x = fr9_lds_get(lds, e);                     // < 56 r
        if (scale) x = fr9_mul(x, sc);
        const uint64_t g = gidx(e);
        if (io.add_low && (g & lenm) < half) x = fr9_add(x, fr9_load(io.lo + 8 * g));
        fr9_store(dst + 8 * g, x);
    }
}
// The middle of a convolution in ONE kernel: the last (contiguous) forward pass, the pointwise product
// with a table that is already in the transform domain, and the first (contiguous) inverse pass all
// work on the same 1024-element tile, so the data makes one HBM round trip instead of three.
__global__ __launch_bounds__(NTT_THREADS) void k_ntt_mid(uint32_t* __restrict__ data, const uint32_t* __restrict__ tw_fwd,
                                                        const uint32_t* __restrict__ tw_inv, uint32_t log_T, uint32_t s,
                                                        const uint32_t* __restrict__ tab, uint64_t tab_mask,
                                                        const uint32_t* __restrict__ scale) {
    __shared__ NttTile lds;
    const uint32_t T = 1u << log_T;
    const uint64_t base = (uint64_t)blockIdx.x << log_T;
    for (uint32_t e = threadIdx.x; e < T; e += NTT_THREADS) fr9_lds_put(lds, e, fr9_load(data + 8 * (base + e)));
    __syncthreads();
    lds_ntt_stages<false>(lds, tw_fwd, T, s, 0, 0, 0, false);
    for (uint32_t e = threadIdx.x; e < T; e += NTT_THREADS)      // own elements only: no barrier needed before
        fr9_lds_put(lds, e, fr9_mul(fr9_lds_get(lds, e), fr9_load(tab + 8 * ((base + e) & tab_mask))));
    __syncthreads();
    lds_ntt_stages<true>(lds, tw_inv, T, s, 0, 0, 0, false);
    Fr9 sc;
    if (scale) sc = fr9_load(scale);
    for (uint32_t e = threadIdx.x; e < T; e += NTT_THREADS) {
        Fr9 x = fr9_lds_get(lds, e);
        if (scale) x = fr9_mul(x, sc);
        fr9_store(data + 8 * (base + e), x);
    }
}

__global__ void k_fr_bitrev(uint32_t* data, uint64_t total, uint32_t log_len) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total || log_len == 0) return;
    uint64_t seg = i >> log_len, k = i & (((uint64_t)1 << log_len) - 1);
    uint64_t r = __brevll(k) >> (64 - log_len);
    if (k < r) {
        uint32_t* p = data + 8 * i;
        uint32_t* q = data + 8 * ((seg << log_len) + r);
        Fr a = fe_load<FrParams>(p), b = fe_load<FrParams>(q);
        fe_store<FrParams>(p, b);
        fe_store<FrParams>(q, a);
    }
}
__global__ void k_fr_to_mont(uint32_t* dst, const uint32_t* src, uint64_t n, int* flag) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr a = fe_load<FrParams>(src + 8 * i);
    if (flag && !fe_is_canonical(a)) *flag = 1;
    fe_store<FrParams>(dst + 8 * i, fe_to_mont(a));
}
__global__ void k_fr_from_mont(uint32_t* dst, const uint32_t* src, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe_store<FrParams>(dst + 8 * i, fe_from_mont(fe_load<FrParams>(src + 8 * i)));
}
__global__ void k_fr_pointwise_mul(uint32_t* out, const uint32_t* a, const uint32_t* b, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe_store<FrParams>(out + 8 * i, fe_mul(fe_load<FrParams>(a + 8 * i), fe_load<FrParams>(b + 8 * i)));
}
// Scale factors of the inverse transforms, filled once, in the factor form of fr29.cuh (f * 2^261):
// out[k] = 2^-k for k = 0..32, and out[33 + k] = 32 * 2^-k -- the scale to use when the pointwise table of a
// convolution was in the DATA form (a transform output, x * 2^256): the product then lacks a factor 32.
__global__ void k_gen_inv_pow2(uint32_t* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Fr half = fe_inv(fe_from_u32<FrParams>(2));
    Fr acc = fe_one<FrParams>();
    const Fr c32 = fr_c32();
    for (int k = 0; k <= 32; k++) {
        const Fr f = fe_mul(acc, c32);
        fe_store<FrParams>(out + 8 * k, f);
        fe_store<FrParams>(out + 8 * (33 + k), fe_mul(f, c32));
        acc = fe_mul(acc, half);
    }
}
// buf <- 32 * buf: a transform output (data form) becomes a pointwise table (factor form)
__global__ void k_fr_to_factor(uint32_t* buf, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe_store<FrParams>(buf + 8 * i, fe_mul(fe_load<FrParams>(buf + 8 * i), fr_c32()));
}

// The twiddle tables live in the CONTEXT of the device they were generated on (zk_common.h: CtxBufs; contexts are heap-allocated and never destroyed
// by static destructors, which must not call into a HIP runtime that is already gone).
int ntt_ensure_twiddles(uint32_t log_n) {
    Ctx& c = ctx();
    if (log_n > 30) ZK_FAIL(ZK_ERR_ARG, "NTT size above 2^30 is not supported");
    if (!c.bufs) ZK_FAIL(ZK_ERR_HIP, "ntt: no device context (zk_init)");
    DevBuf &g_tw_fwd = c.bufs->tw_fwd, &g_tw_inv = c.bufs->tw_inv, &g_inv_pow2 = c.bufs->inv_pow2;
    if (!g_inv_pow2.p) {
        ZKCHK(g_inv_pow2.alloc(66 * 32));
        hipLaunchKernelGGL(k_gen_inv_pow2, dim3(1), dim3(64), 0, c.stream, g_inv_pow2.as<uint32_t>());
        HIPCHK(hipGetLastError());
    }
    if (c.tw_log >= log_n && c.tw_fwd) return ZK_OK;
    uint32_t k = log_n < 12 ? 12 : log_n;
    HIPCHK(hipStreamSynchronize(c.stream));
    ZKCHK(g_tw_fwd.alloc(((size_t)32) << k));
    ZKCHK(g_tw_inv.alloc(((size_t)32) << k));
    uint64_t total = (uint64_t)1 << k;
    dim3 grid((unsigned)((total + 255) / 256));
    hipLaunchKernelGGL(k_gen_twiddles, grid, dim3(256), 0, c.stream, g_tw_fwd.as<uint32_t>(), k, 0);
    hipLaunchKernelGGL(k_gen_twiddles, grid, dim3(256), 0, c.stream, g_tw_inv.as<uint32_t>(), k, 1);
    HIPCHK(hipGetLastError());
    c.tw_fwd = g_tw_fwd.p;
    c.tw_inv = g_tw_inv.p;
    c.tw_log = k;
    return ZK_OK;
}
static void ntt_release() {          // zk_shutdown, once per context (current): the buffers themselves die with the context's CtxBufs
    ctx().tw_fwd = ctx().tw_inv = nullptr; ctx().tw_log = 0;
}
static CleanupRegistrar g_ntt_cleanup(ntt_release);

// Pass plan: the contiguous pass takes the s_last smallest strides, strided passes the rest.
static void plan(uint32_t log_len, uint32_t log_T, std::vector<PassArgs>& strided, PassArgs& last) {
    uint32_t s_last = log_len < log_T ? log_len : log_T;
    last = PassArgs{s_last, s_last, log_T, 0, 0};
    uint32_t rem = log_len - s_last;
    if (rem == 0) return;
    const uint32_t smax = log_T - 2;                 // rows of C >= 4 elements
    uint32_t np = (rem + smax - 1) / smax;
    uint32_t hi = log_len;
    for (uint32_t p = 0; p < np; p++) {
        uint32_t s = (rem + (np - p) - 1) / (np - p);
        strided.push_back(PassArgs{hi, s, log_T, log_T - s, 1});
        hi -= s;
        rem -= s;
    }
}

static int run_ntt(void* d, uint64_t total, uint32_t log_len, bool inverse, bool scale, hipStream_t s) {
    if (log_len == 0) return ZK_OK;
    if (total == 0 || (total & (total - 1)) || ((uint64_t)1 << log_len) > total) ZK_FAIL(ZK_ERR_ARG, "ntt: bad sizes");
    ZKCHK(ntt_ensure_twiddles(log_len));
    Ctx& c = ctx();
    uint32_t log_total = ceil_log2(total);
    uint32_t log_T = log_total < (uint32_t)NTT_LOG_T ? log_total : NTT_LOG_T;
    std::vector<PassArgs> st;
    PassArgs last;
    plan(log_len, log_T, st, last);
    dim3 grid((unsigned)(total >> log_T));
    const uint32_t* sc = scale ? c.bufs->inv_pow2.as<uint32_t>() + 8 * log_len : nullptr;
    ScopedTimer t(inverse ? "ntt_inverse" : "ntt_forward", s);
    if (!inverse) {
        for (auto& p : st)
            hipLaunchKernelGGL(k_ntt_pass<false>, grid, dim3(NTT_THREADS), 0, s, (uint32_t*)d, (const uint32_t*)c.tw_fwd, p, (const uint32_t*)nullptr, PassIO{});
        hipLaunchKernelGGL(k_ntt_pass<false>, grid, dim3(NTT_THREADS), 0, s, (uint32_t*)d, (const uint32_t*)c.tw_fwd, last, (const uint32_t*)nullptr, PassIO{});
    } else {
        hipLaunchKernelGGL(k_ntt_pass<true>, grid, dim3(NTT_THREADS), 0, s, (uint32_t*)d, (const uint32_t*)c.tw_inv, last, st.empty() ? sc : (const uint32_t*)nullptr, PassIO{});
        for (size_t i = st.size(); i-- > 0;)
            hipLaunchKernelGGL(k_ntt_pass<true>, grid, dim3(NTT_THREADS), 0, s, (uint32_t*)d, (const uint32_t*)c.tw_inv, st[i], i == 0 ? sc : (const uint32_t*)nullptr, PassIO{});
    }
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
// work <- iNTT( NTT(x) * tab ), per 2^log_len segment, everything fused around the tile passes.
//   pad_src != nullptr : x = upper half of every node of pad_src, zero padded (tree level); else x = work
//   add_dst != nullptr : the result plus the node's lower half goes to add_dst (tree level); else to work
// scale: multiply by 2^-log_len (when the table does not already carry it).
// The table is in the factor form (fr_to_factor of a transform output, scaled or not); a table that is a plain
// transform output is announced with tab_is_data and needs scale = true.
int ntt_mul_table(void* work, uint64_t total, uint32_t log_len, const void* tab, uint64_t tab_mask, bool scale,
                  const void* pad_src, void* add_dst, hipStream_t s, bool tab_is_data) {
    if (tab_is_data && !scale) ZK_FAIL(ZK_ERR_ARG, "ntt_mul_table: a data-form table needs the scaled form");
    if (total == 0 || (total & (total - 1)) || ((uint64_t)1 << log_len) > total || log_len == 0) ZK_FAIL(ZK_ERR_ARG, "ntt_mul_table: bad sizes");
    ZKCHK(ntt_ensure_twiddles(log_len));
    Ctx& c = ctx();
    uint32_t log_total = ceil_log2(total);
    uint32_t log_T = log_total < (uint32_t)NTT_LOG_T ? log_total : NTT_LOG_T;
    std::vector<PassArgs> st;
    PassArgs last;
    plan(log_len, log_T, st, last);
    if (st.empty() && (pad_src || add_dst)) ZK_FAIL(ZK_ERR_ARG, "ntt_mul_table: fused edges need a node larger than one tile");
    dim3 grid((unsigned)(total >> log_T));
    const uint32_t* sc = scale ? c.bufs->inv_pow2.as<uint32_t>() + 8 * (log_len + (tab_is_data ? 33 : 0)) : nullptr;
    ScopedTimer t("ntt_mul_table", s);
    for (size_t i = 0; i < st.size(); i++) {
        PassIO io{};
        if (i == 0 && pad_src) { io.src = (const uint32_t*)pad_src; io.pad_half = 1; io.log_len = log_len; }
        hipLaunchKernelGGL(k_ntt_pass<false>, grid, dim3(NTT_THREADS), 0, s, (uint32_t*)work, (const uint32_t*)c.tw_fwd, st[i], (const uint32_t*)nullptr, io);
    }
    hipLaunchKernelGGL(k_ntt_mid, grid, dim3(NTT_THREADS), 0, s, (uint32_t*)work, (const uint32_t*)c.tw_fwd, (const uint32_t*)c.tw_inv, log_T, last.s,
                       (const uint32_t*)tab, tab_mask, st.empty() ? sc : (const uint32_t*)nullptr);
    for (size_t i = st.size(); i-- > 0;) {
        PassIO io{};
        if (i == 0 && add_dst) { io.dst = (uint32_t*)add_dst; io.lo = (const uint32_t*)add_dst; io.add_low = 1; io.log_len = log_len; }
        hipLaunchKernelGGL(k_ntt_pass<true>, grid, dim3(NTT_THREADS), 0, s, (uint32_t*)work, (const uint32_t*)c.tw_inv, st[i], i == 0 ? sc : (const uint32_t*)nullptr, io);
    }
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
int ntt_forward(void* d, uint64_t total, uint32_t log_len, hipStream_t s) { return run_ntt(d, total, log_len, false, false, s); }
int ntt_inverse(void* d, uint64_t total, uint32_t log_len, bool scale, hipStream_t s) { return run_ntt(d, total, log_len, true, scale, s); }

static inline dim3 grid1d(uint64_t n) { return dim3((unsigned)((n + 255) / 256)); }
int fr_bitrev_permute(void* d, uint64_t total, uint32_t log_len, hipStream_t s) {
    hipLaunchKernelGGL(k_fr_bitrev, grid1d(total), dim3(256), 0, s, (uint32_t*)d, total, log_len);
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
int fr_to_mont(void* dst, const void* src, uint64_t n, int* flag, hipStream_t s) {
    if (!n) return ZK_OK;
    hipLaunchKernelGGL(k_fr_to_mont, grid1d(n), dim3(256), 0, s, (uint32_t*)dst, (const uint32_t*)src, n, flag);
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
int fr_from_mont(void* dst, const void* src, uint64_t n, hipStream_t s) {
    if (!n) return ZK_OK;
    hipLaunchKernelGGL(k_fr_from_mont, grid1d(n), dim3(256), 0, s, (uint32_t*)dst, (const uint32_t*)src, n);
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
int fr_to_factor(void* buf, uint64_t n, hipStream_t s) {
    if (!n) return ZK_OK;
    hipLaunchKernelGGL(k_fr_to_factor, grid1d(n), dim3(256), 0, s, (uint32_t*)buf, n);
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
int fr_pointwise_mul(void* out, const void* a, const void* b, uint64_t n, hipStream_t s) {
    if (!n) return ZK_OK;
    hipLaunchKernelGGL(k_fr_pointwise_mul, grid1d(n), dim3(256), 0, s, (uint32_t*)out, (const uint32_t*)a, (const uint32_t*)b, n);
    HIPCHK(hipGetLastError());
    return ZK_OK;
}

}  // namespace zk

using namespace zk;

extern "C" int zk_fr_ntt(uint8_t* inout, uint32_t log_n, int inverse) {
    if (!inout) ZK_FAIL(ZK_ERR_ARG, "zk_fr_ntt: null buffer");
    if (log_n > 28) ZK_FAIL(ZK_ERR_ARG, "zk_fr_ntt: log_n > 28");   // FFT.ml:230 raises above 2^32
    ZKCHK(ensure_init());
    Ctx& c = ctx();
    uint64_t n = (uint64_t)1 << log_n;
    DevBuf d, flag;
    ZKCHK(d.alloc(n * 32));
    ZKCHK(flag.alloc(4));
    HIPCHK(hipMemsetAsync(flag.p, 0, 4, c.stream));
    HIPCHK(hipMemcpyAsync(d.p, inout, n * 32, hipMemcpyHostToDevice, c.stream));
    ZKCHK(fr_to_mont(d.p, d.p, n, flag.as<int>(), c.stream));
    if (!inverse) {
        ZKCHK(ntt_forward(d.p, n, log_n, c.stream));
        ZKCHK(fr_bitrev_permute(d.p, n, log_n, c.stream));
    } else {
        ZKCHK(fr_bitrev_permute(d.p, n, log_n, c.stream));
        ZKCHK(ntt_inverse(d.p, n, log_n, true, c.stream));
    }
    ZKCHK(fr_from_mont(d.p, d.p, n, c.stream));
    int h_flag = 0;
    HIPCHK(hipMemcpyAsync(&h_flag, flag.p, 4, hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    if (h_flag) ZK_FAIL(ZK_ERR_SCALAR_RANGE, "zk_fr_ntt: input element >= r");
    HIPCHK(hipMemcpy(inout, d.p, n * 32, hipMemcpyDeviceToHost));
    return ZK_OK;
}
