// Convolutions of Fr vectors through a residue number system: 18 primes p = k 2^23 + 1 < 2^31, 32-bit Montgomery arithmetic, CRT back to Fr.
// See rns_ntt.cuh for why; scripts/gen_rns_consts.py for the constants and their bounds.
//
// Data flow of one convolution with a fixed table (the shape of every product of the Fr stage):
//   k_rns_in      Fr (8 words, value x < 2^256) -> 18 residues, one array of `total` words per prime            32 B read, 72 B written per element
//   k_rns_strided the outer radix-2 stages of transforms longer than one LDS tile (DIF forward), rows of >= 32 consecutive words
//   k_rns_mid     one tile of 2^13 words in LDS: the inner 13 forward stages, the pointwise product with the table, the inner 13 inverse stages
//   k_rns_strided the outer inverse stages (DIT)
//   k_rns_out     18 residues -> CRT -> X R^-1 mod r -> Fr (optionally added to the lower half of a tree node)     72 B read, 32 B written
// Forward = DIF (natural -> bit-reversed), inverse = DIT (bit-reversed -> natural), twiddle heap tw[h + j] = w_2h^j per prime; the inverse twiddle
// w_2h^-j is -w_2h^(h-j), read from the same heap.  The scale 2^-log_len lives in the table (or in the constant of a data x data product).
// All residues are kept in [0, p): p < 2^31, so a + b and a - b + p fit 32 bits and one unsigned min() is the conditional subtraction.
#include "rns_ntt.cuh"

#include "fr29.cuh"
#include "rns_consts.cuh"

#include <stdlib.h>

namespace zk {

static_assert(RNS_NP == (int)RNS_PRIMES && RNS_LOG_MAX == (int)RNS_MAX_LOG, "rns_consts.cuh does not match rns_ntt.cuh");
static constexpr uint32_t RNS_LOG_T = 13, RNS_T = 1u << RNS_LOG_T, RNS_THREADS = 512;
static constexpr bool RNS_DEFAULT_ON = false;          // until measured
static constexpr uint32_t RNS_STRIDED_MAX = 8;          // stages per strided pass: rows of 2^(13 - 8) = 32 words = 128 bytes

FF_INLINE uint32_t rns_mred(uint64_t t, uint32_t p, uint32_t pinv) {          // t < p 2^32  ->  t 2^-32 mod p in [0, p)
    const uint32_t m = (uint32_t)t * pinv;
    const uint32_t r = (uint32_t)((t + (uint64_t)m * p) >> 32);               // < 2p < 2^32; the sum is < 2^64
    return min(r, r - p);
}
FF_INLINE uint32_t rns_mul(uint32_t a, uint32_t b, uint32_t p, uint32_t pinv) { return rns_mred((uint64_t)a * b, p, pinv); }
FF_INLINE uint32_t rns_add(uint32_t a, uint32_t b, uint32_t p) {
    const uint32_t s = a + b;
    return min(s, s - p);
}
FF_INLINE uint32_t rns_sub(uint32_t a, uint32_t b, uint32_t p) {
    const uint32_t d = a - b;
    return min(d, d + p);
}

// ---- twiddles: tw[prime][h + j] = w_2h^j (Montgomery form), h = 1, 2, .. 2^(log_cap - 1)
__global__ void k_rns_gen_tw(uint32_t* __restrict__ tw, uint32_t log_cap) {
    const uint32_t i = blockIdx.y, p = RNS_P[i], pinv = RNS_PINV[i];
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, cap = (uint64_t)1 << log_cap;
    if (t >= cap) return;
    uint32_t acc = RNS_ONE[i];
    if (t) {
        const uint32_t k = 64 - __builtin_clzll(t);                            // level: h = 2^(k-1)
        const uint32_t j = (uint32_t)(t - ((uint64_t)1 << (k - 1)));
        uint32_t e = j << (RNS_LOG_MAX - k), base = RNS_ROOT[i];             // w_2h = root^(2^(23-k))
        for (uint32_t b = 0; b < RNS_LOG_MAX; b++) {
            if ((e >> b) & 1u) acc = rns_mul(acc, base, p, pinv);
            base = rns_mul(base, base, p, pinv);
        }
    }
    tw[(uint64_t)i * cap + t] = acc;
}
struct RnsTw {
    DevBuf buf;
    uint32_t log_cap = 0;
};
static RnsTw* g_rns_tw[64];          // per virtual device (contexts are never destroyed by static destructors; released in zk_shutdown)
static void rns_release() {
    const int v = ctx().vdev;
    if (v >= 0 && v < 64 && g_rns_tw[v]) {
        delete g_rns_tw[v];
        g_rns_tw[v] = nullptr;
    }
}
static CleanupRegistrar g_rns_cleanup(rns_release);
int rns_ensure_twiddles(uint32_t log_len) {
    if (log_len > RNS_MAX_LOG) ZK_FAIL(ZK_ERR_ARG, "rns: transform longer than 2^23");
    Ctx& c = ctx();
    if (c.vdev < 0 || c.vdev >= 64) ZK_FAIL(ZK_ERR_ARG, "rns: no device context");
    if (!g_rns_tw[c.vdev]) g_rns_tw[c.vdev] = new RnsTw;
    RnsTw& t = *g_rns_tw[c.vdev];
    if (t.log_cap >= log_len && t.buf.p) return ZK_OK;
    const uint32_t k = log_len < 14 ? 14 : log_len;
    HIPCHK(hipDeviceSynchronize());          // a proof in flight may still read the old heap
    ZKCHK(t.buf.alloc(4 * (size_t)RNS_PRIMES << k));
    hipLaunchKernelGGL(k_rns_gen_tw, dim3((unsigned)(((uint64_t)1 << k) / 256), RNS_PRIMES), dim3(256), 0, c.stream, t.buf.as<uint32_t>(), k);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c.stream));
    t.log_cap = k;
    return ZK_OK;
}
static const uint32_t* rns_tw(uint32_t* log_cap) {
    RnsTw& t = *g_rns_tw[ctx().vdev];
    *log_cap = t.log_cap;
    return t.buf.as<uint32_t>();
}

// ---- Fr -> residues
__global__ void k_rns_in(uint32_t* __restrict__ res, const uint32_t* __restrict__ src, uint64_t total, uint32_t log_len, int mode) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    bool zero = false;
    uint64_t from = e;
    if (mode == RNS_IN_TREE_HI) {
        const uint64_t half = (uint64_t)1 << (log_len - 1), j = e & (2 * half - 1);
        zero = j >= half;
        from = e + half;
    }
    if (zero) {
#pragma unroll
        for (int i = 0; i < RNS_NP; i++) res[(uint64_t)i * total + e] = 0;
        return;
    }
    const Fr9 x = fr9_load(src + 8 * from);                 // exact 29-bit limbs (limb 8: the top 24 bits), value < 2^256
#pragma unroll
    for (int i = 0; i < RNS_NP; i++) {
        const uint32_t p = RNS_P[i], pinv = RNS_PINV[i];
        uint64_t a0 = 0, a1 = 0;                            // 5 and 4 terms < 2^29 p each: < p 2^32, the reduction's range
#pragma unroll
        for (int k = 0; k < 5; k++) a0 += (uint64_t)x.v[k] * RNS_IN_C[i][k];
#pragma unroll
        for (int k = 5; k < 9; k++) a1 += (uint64_t)x.v[k] * RNS_IN_C[i][k];
        res[(uint64_t)i * total + e] = rns_add(rns_mred(a0, p, pinv), rns_mred(a1, p, pinv), p);
    }
}

// ---- residues -> Fr: X R_fr^-1 mod r, Montgomery-reduced by 2^261 from the CRT sum (constants: gen_rns_consts.py)
FF_INLINE Fr9 rns_crt(const uint32_t* __restrict__ res, uint64_t total, uint64_t e) {
    uint32_t g[RNS_NP];
    uint64_t qs = (uint64_t)1 << 44;
#pragma unroll
    for (int i = 0; i < RNS_NP; i++) {
        g[i] = rns_mul(res[(uint64_t)i * total + e], RNS_CRT_INV[i], RNS_P[i], RNS_PINV[i]);
        qs += (uint64_t)g[i] * RNS_CRT_Q[i];
    }
    const uint32_t alpha = (uint32_t)(qs >> 56);            // <= 18
    // columns of sum_i g_i K_i + T[alpha]: two accumulators of nine 60-bit terms each, then 29-bit limbs with a running carry
    uint32_t t[11];
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        uint64_t a = RNS_CRT_T[alpha][k], b = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) a += (uint64_t)g[i] * RNS_CRT_K[i][k];
#pragma unroll
        for (int i = 9; i < RNS_NP; i++) b += (uint64_t)g[i] * RNS_CRT_K[i][k];
        const uint64_t lo = (a & FR29_MASK) + (b & FR29_MASK) + (carry & FR29_MASK);
        t[k] = (uint32_t)lo & FR29_MASK;
        carry = (a >> FR29_W) + (b >> FR29_W) + (carry >> FR29_W) + (lo >> FR29_W);
    }
    t[9] = (uint32_t)carry & FR29_MASK;
    t[10] = (uint32_t)(carry >> FR29_W);                    // the sum is < 2^36 r < 2^291
    // Montgomery reduction by 2^261 (r = 1 mod 2^29: the quotient digit is a negation, as in fr9_mul)
    uint64_t acc = 0;
    uint32_t m[FR29_L];
    Fr9 r;
#pragma unroll
    for (int k = 0; k < FR29_L; k++) {
        acc += t[k];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * FR29_MOD[k - i];
        m[k] = (0u - (uint32_t)acc) & FR29_MASK;
        acc += m[k];
        acc >>= FR29_W;
    }
#pragma unroll
    for (int k = FR29_L; k < 2 * FR29_L - 1; k++) {
        if (k < 11) acc += t[k];
#pragma unroll
        for (int i = k - FR29_L + 1; i < FR29_L; i++) acc += (uint64_t)m[i] * FR29_MOD[k - i];
        r.v[k - FR29_L] = (uint32_t)acc & FR29_MASK;
        acc >>= FR29_W;
    }
    r.v[FR29_L - 1] = (uint32_t)acc;
    return r;                                               // < 2 r
}
__global__ void k_rns_out(uint32_t* __restrict__ dst, const uint32_t* __restrict__ res, uint64_t total, uint32_t log_len, int mode, uint64_t out_lo, uint64_t out_cnt,
                          uint64_t dst_stride) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (mode == RNS_OUT_TREE_ADD) {
        if (g >= total) return;
        const uint64_t half = (uint64_t)1 << (log_len - 1), j = g & (2 * half - 1);
        Fr9 x = rns_crt(res, total, g);
        if (j < half) x = fr9_add(x, fr9_load(dst + 8 * g));
        fr9_store(dst + 8 * g, x);
        return;
    }
    const uint64_t batch = total >> log_len;
    if (g >= batch * out_cnt) return;
    const uint64_t b = g / out_cnt, k = g % out_cnt;
    fr9_store(dst + 8 * (b * dst_stride + k), rns_crt(res, total, (b << log_len) + out_lo + k));
}

// ---- one LDS tile: inner forward stages, pointwise product, inner inverse stages
// mul: 0 none, 1 table (Montgomery multipliers, already scaled by 2^-log_len), 2 data (plain residues of another transform: x y 2^-log_len)
__global__ __launch_bounds__(RNS_THREADS) void k_rns_mid(uint32_t* __restrict__ res, const uint32_t* __restrict__ tw, uint32_t tw_log, uint64_t total, uint32_t log_len,
                                                         uint32_t log_t, int do_fwd, int mul, const uint32_t* __restrict__ tab, uint64_t tab_len, uint64_t tab_mask, int do_inv) {
    __shared__ uint32_t lds[RNS_T];
    const uint32_t i = blockIdx.y, p = RNS_P[i], pinv = RNS_PINV[i], T = 1u << log_t;
    const uint64_t base = (uint64_t)blockIdx.x << log_t;
    uint32_t* __restrict__ a = res + (uint64_t)i * total + base;
    const uint32_t* __restrict__ w = tw + ((uint64_t)i << tw_log);
    const uint32_t inner = log_len < log_t ? log_len : log_t;
    for (uint32_t e = threadIdx.x; e < T; e += RNS_THREADS) lds[e] = a[e];
    __syncthreads();
    if (do_fwd) {
        for (uint32_t b = inner; b-- > 0;) {
            const uint32_t h = 1u << b;
            for (uint32_t q = threadIdx.x; q < T / 2; q += RNS_THREADS) {
                const uint32_t j = q & (h - 1), e0 = ((q >> b) << (b + 1)) | j;
                const uint32_t u = lds[e0], v = lds[e0 + h];
                lds[e0] = rns_add(u, v, p);
                lds[e0 + h] = rns_mul(rns_sub(u, v, p), w[h + j], p, pinv);
            }
            __syncthreads();
        }
    }
    if (mul) {
        const uint32_t* __restrict__ tb = tab + (uint64_t)i * tab_len;
        const uint32_t sc = RNS_SCALE[i][log_len];
        for (uint32_t e = threadIdx.x; e < T; e += RNS_THREADS) {
            uint32_t x = rns_mul(lds[e], tb[(base + e) & tab_mask], p, pinv);
            if (mul == 2) x = rns_mul(x, sc, p, pinv);
            lds[e] = x;
        }
        __syncthreads();
    }
    if (do_inv) {
        for (uint32_t b = 0; b < inner; b++) {
            const uint32_t h = 1u << b;
            for (uint32_t q = threadIdx.x; q < T / 2; q += RNS_THREADS) {
                const uint32_t j = q & (h - 1), e0 = ((q >> b) << (b + 1)) | j;
                const uint32_t wi = j ? p - w[2 * h - j] : w[h];            // w_2h^-j = -w_2h^(h-j)
                const uint32_t u = lds[e0], v = rns_mul(lds[e0 + h], wi, p, pinv);
                lds[e0] = rns_add(u, v, p);
                lds[e0 + h] = rns_sub(u, v, p);
            }
            __syncthreads();
        }
    }
    for (uint32_t e = threadIdx.x; e < T; e += RNS_THREADS) a[e] = lds[e];
}

// ---- the outer stages b in [lo, hi) of transforms longer than a tile: element (upper << hi) | (q << lo) | low, a tile = all q x C consecutive low
template <bool INVERSE>
__global__ __launch_bounds__(RNS_THREADS) void k_rns_strided(uint32_t* __restrict__ res, const uint32_t* __restrict__ tw, uint32_t tw_log, uint64_t total, uint32_t hi, uint32_t lo) {
    __shared__ uint32_t lds[RNS_T];
    const uint32_t i = blockIdx.y, p = RNS_P[i], pinv = RNS_PINV[i];
    const uint32_t s = hi - lo, log_c = RNS_LOG_T - s, C = 1u << log_c;
    const uint64_t per_upper = (uint64_t)1 << (lo - log_c);                   // tiles per value of the upper bits
    const uint64_t upper = blockIdx.x / per_upper, lc = blockIdx.x % per_upper;
    uint32_t* __restrict__ a = res + (uint64_t)i * total + (upper << hi) + (lc << log_c);
    const uint32_t* __restrict__ w = tw + ((uint64_t)i << tw_log);
    const uint32_t low0 = (uint32_t)(lc << log_c);                            // low part of the element index of column 0 (lo <= 23 bits)
    for (uint32_t t = threadIdx.x; t < RNS_T; t += RNS_THREADS) lds[t] = a[((uint64_t)(t >> log_c) << lo) + (t & (C - 1))];
    __syncthreads();
    for (uint32_t st = 0; st < s; st++) {
        const uint32_t bl = INVERSE ? st : s - 1 - st, b = lo + bl;           // local and global bit of this stage
        const uint32_t hl = 1u << bl;
        const uint64_t h = (uint64_t)1 << b;
        for (uint32_t q = threadIdx.x; q < RNS_T / 2; q += RNS_THREADS) {
            const uint32_t r = q & (C - 1), qq = q >> log_c;
            const uint32_t jl = qq & (hl - 1), q0 = ((qq >> bl) << (bl + 1)) | jl;
            const uint32_t t0 = (q0 << log_c) | r, t1 = t0 + (hl << log_c);
            const uint64_t j = ((uint64_t)jl << lo) + low0 + r;               // e0 mod 2^b
            const uint32_t u = lds[t0];
            if (INVERSE) {
                const uint32_t wi = j ? p - w[2 * h - j] : w[h];
                const uint32_t v = rns_mul(lds[t1], wi, p, pinv);
                lds[t0] = rns_add(u, v, p);
                lds[t1] = rns_sub(u, v, p);
            } else {
                const uint32_t v = lds[t1];
                lds[t0] = rns_add(u, v, p);
                lds[t1] = rns_mul(rns_sub(u, v, p), w[h + j], p, pinv);
            }
        }
        __syncthreads();
    }
    for (uint32_t t = threadIdx.x; t < RNS_T; t += RNS_THREADS) a[((uint64_t)(t >> log_c) << lo) + (t & (C - 1))] = lds[t];
}
// table finish: x -> x 2^-log_len as a Montgomery multiplier
__global__ void k_rns_table_scale(uint32_t* __restrict__ res, uint64_t total, uint32_t log_len) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = blockIdx.y;
    if (e >= total) return;
    res[(uint64_t)i * total + e] = rns_mul(res[(uint64_t)i * total + e], RNS_SCALE[i][log_len], RNS_P[i], RNS_PINV[i]);
}

// ------------------------------------------------------------------ host side
bool rns_enabled() {
    const char* e = ZK_FORM_ENV("ZK_FR_RNS");
    return e ? atoi(e) != 0 : RNS_DEFAULT_ON;
}
static inline dim3 g1d(uint64_t n, unsigned t = 256) { return dim3((unsigned)((n + t - 1) / t)); }
static int check_shape(uint64_t total, uint32_t log_len) {
    if (log_len == 0 || log_len > RNS_MAX_LOG || total == 0 || (total & (total - 1)) || ((uint64_t)1 << log_len) > total) ZK_FAIL(ZK_ERR_ARG, "rns: bad transform shape");
    return ZK_OK;
}
// outer stages of a forward (top down) or inverse (bottom up) transform, in passes of at most RNS_STRIDED_MAX stages
static void strided_passes(uint32_t* res, const uint32_t* tw, uint32_t tw_log, uint64_t total, uint32_t log_len, bool inverse, hipStream_t s) {
    if (log_len <= RNS_LOG_T) return;
    const uint32_t outer = log_len - RNS_LOG_T, np = (outer + RNS_STRIDED_MAX - 1) / RNS_STRIDED_MAX;
    uint32_t cuts[8];
    cuts[0] = RNS_LOG_T;
    for (uint32_t k = 0; k < np; k++) cuts[k + 1] = RNS_LOG_T + (uint32_t)(((uint64_t)outer * (k + 1)) / np);
    const dim3 grid((unsigned)(total >> RNS_LOG_T), RNS_PRIMES);
    if (!inverse) {
        for (uint32_t k = np; k-- > 0;) hipLaunchKernelGGL(k_rns_strided<false>, grid, dim3(RNS_THREADS), 0, s, res, tw, tw_log, total, cuts[k + 1], cuts[k]);
    } else {
        for (uint32_t k = 0; k < np; k++) hipLaunchKernelGGL(k_rns_strided<true>, grid, dim3(RNS_THREADS), 0, s, res, tw, tw_log, total, cuts[k + 1], cuts[k]);
    }
}
static uint32_t tile_log(uint64_t total) {
    uint32_t l = 0;
    while (((uint64_t)1 << l) < total) l++;
    return l < RNS_LOG_T ? l : RNS_LOG_T;
}
int rns_table_build(DevBuf& table, const void* d_fr, uint64_t total, uint32_t log_len, hipStream_t s) {
    ZKCHK(check_shape(total, log_len));
    ZKCHK(rns_ensure_twiddles(log_len));
    ZKCHK(table.alloc(4 * (size_t)RNS_PRIMES * total));
    uint32_t tw_log;
    const uint32_t* tw = rns_tw(&tw_log);
    uint32_t* res = table.as<uint32_t>();
    const uint32_t lt = tile_log(total);
    hipLaunchKernelGGL(k_rns_in, g1d(total), dim3(256), 0, s, res, (const uint32_t*)d_fr, total, log_len, (int)RNS_IN_PLAIN);
    strided_passes(res, tw, tw_log, total, log_len, false, s);
    hipLaunchKernelGGL(k_rns_mid, dim3((unsigned)(total >> lt), RNS_PRIMES), dim3(RNS_THREADS), 0, s, res, tw, tw_log, total, log_len, lt, 1, 0, (const uint32_t*)nullptr, (uint64_t)0,
                       (uint64_t)0, 0);
    hipLaunchKernelGGL(k_rns_table_scale, dim3((unsigned)((total + 255) / 256), RNS_PRIMES), dim3(256), 0, s, res, total, log_len);
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
int rns_conv_table(RnsWork& w, const void* d_src_fr, uint64_t total, uint32_t log_len, RnsIn in_mode, const DevBuf& table, uint64_t tab_mask, void* d_dst_fr, RnsOut out_mode,
                   uint64_t out_lo, uint64_t out_cnt, uint64_t dst_stride, hipStream_t s) {
    ZKCHK(check_shape(total, log_len));
    const uint64_t tab_len = tab_mask + 1;
    if ((tab_len & tab_mask) || tab_len < ((uint64_t)1 << log_len) || table.bytes < 4 * (size_t)RNS_PRIMES * tab_len) ZK_FAIL(ZK_ERR_ARG, "rns_conv_table: table does not match the transform");
    if (out_mode == RNS_OUT_RANGE && out_lo + out_cnt > ((uint64_t)1 << log_len)) ZK_FAIL(ZK_ERR_ARG, "rns_conv_table: output range outside the transform");
    ZKCHK(rns_ensure_twiddles(log_len));
    ZKCHK(w.ensure(total));
    uint32_t tw_log;
    const uint32_t* tw = rns_tw(&tw_log);
    uint32_t* res = w.res.as<uint32_t>();
    const uint32_t lt = tile_log(total);
    ScopedTimer t("rns_conv", s);
    hipLaunchKernelGGL(k_rns_in, g1d(total), dim3(256), 0, s, res, (const uint32_t*)d_src_fr, total, log_len, (int)in_mode);
    strided_passes(res, tw, tw_log, total, log_len, false, s);
    hipLaunchKernelGGL(k_rns_mid, dim3((unsigned)(total >> lt), RNS_PRIMES), dim3(RNS_THREADS), 0, s, res, tw, tw_log, total, log_len, lt, 1, 1, (const uint32_t*)table.as<uint32_t>(), tab_len,
                       tab_mask, 1);
    strided_passes(res, tw, tw_log, total, log_len, true, s);
    const uint64_t outs = out_mode == RNS_OUT_TREE_ADD ? total : (total >> log_len) * out_cnt;
    if (outs) hipLaunchKernelGGL(k_rns_out, g1d(outs), dim3(256), 0, s, (uint32_t*)d_dst_fr, (const uint32_t*)res, total, log_len, (int)out_mode, out_lo, out_cnt, dst_stride);
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
int rns_conv_data(RnsWork& wa, RnsWork& wb, const void* d_a_fr, const void* d_b_fr, uint32_t log_len, void* d_dst_fr, uint64_t out_lo, uint64_t out_cnt, hipStream_t s) {
    const uint64_t total = (uint64_t)1 << log_len;
    ZKCHK(check_shape(total, log_len));
    if (out_lo + out_cnt > total) ZK_FAIL(ZK_ERR_ARG, "rns_conv_data: output range outside the transform");
    ZKCHK(rns_ensure_twiddles(log_len));
    ZKCHK(wa.ensure(total));
    ZKCHK(wb.ensure(total));
    uint32_t tw_log;
    const uint32_t* tw = rns_tw(&tw_log);
    uint32_t *ra = wa.res.as<uint32_t>(), *rb = wb.res.as<uint32_t>();
    const uint32_t lt = tile_log(total);
    const dim3 gm((unsigned)(total >> lt), RNS_PRIMES);
    ScopedTimer t("rns_conv", s);
    hipLaunchKernelGGL(k_rns_in, g1d(total), dim3(256), 0, s, rb, (const uint32_t*)d_b_fr, total, log_len, (int)RNS_IN_PLAIN);
    strided_passes(rb, tw, tw_log, total, log_len, false, s);
    hipLaunchKernelGGL(k_rns_mid, gm, dim3(RNS_THREADS), 0, s, rb, tw, tw_log, total, log_len, lt, 1, 0, (const uint32_t*)nullptr, (uint64_t)0, (uint64_t)0, 0);
    hipLaunchKernelGGL(k_rns_in, g1d(total), dim3(256), 0, s, ra, (const uint32_t*)d_a_fr, total, log_len, (int)RNS_IN_PLAIN);
    strided_passes(ra, tw, tw_log, total, log_len, false, s);
    hipLaunchKernelGGL(k_rns_mid, gm, dim3(RNS_THREADS), 0, s, ra, tw, tw_log, total, log_len, lt, 1, 2, (const uint32_t*)rb, total, total - 1, 1);
    strided_passes(ra, tw, tw_log, total, log_len, true, s);
    if (out_cnt) hipLaunchKernelGGL(k_rns_out, g1d(out_cnt), dim3(256), 0, s, (uint32_t*)d_dst_fr, (const uint32_t*)ra, total, log_len, (int)RNS_OUT_RANGE, out_lo, out_cnt, (uint64_t)0);
    HIPCHK(hipGetLastError());
    return ZK_OK;
}

}  // namespace zk
