// Fr stage of the prover on gfx950: the GPU counterpart of QAP.eval (src/lib/zk/QAP.ml:120-135)
// for circuits given as sparse R1CS rows.
//
// The reference interpolates over the INTEGER points X = 0..n-1 (QAP.ml:84,92), so the
// coefficient vectors it feeds to apply_powers are those of the unique polynomials with
//   v(i) = (L w)_i,  w(i) = (R w)_i,  y(i) = (O w)_i,   h = (v*w - y) / Z,  Z = prod (X - i).
// An NTT over roots of unity is not the interpolation map of that domain; here it is only the
// multiplication engine (as in the reference's own FFT.polynomial_mul, FFT.ml:98-105):
//   1 spmv      a = L w, b = R w, c = O w; remainder = 0 <=> a_i*b_i = c_i for all i (deg rem < n)
//   2 newton    values -> Newton (falling-factorial) coefficients by ONE convolution:
//               d_k = sum_{i<=k} (y_i / i!) * ((-1)^(k-i) / (k-i)!)
//   3 tree      Newton -> monomial, bottom-up over the subproduct tree of prod (X - i):
//               F(node) = F(left) + P_left(X) * F(right); P_left is precomputed per n in the NTT
//               domain, so a level is {pad, NTT, pointwise, iNTT, add} -- fused into one LDS-resident
//               kernel for node sizes <= 1024
//   4 product   top half of v*w (y has degree < n: it never reaches h)
//   5 divide    rev(h) = rev(v*w) * (rev Z)^-1 mod X^(n-1); the power-series inverse is per-n
// All exact arithmetic in Fr: the resulting v, w, h equal the reference's coefficient lists.
#include "frstage.cuh"
#include "ntt_lds.cuh"

namespace zk {

static inline dim3 g1d(uint64_t n, unsigned t = 256) { return dim3((unsigned)((n + t - 1) / t)); }
#define FRP(buf) ((buf).template as<uint32_t>())

// ------------------------------------------------------------------ kernels
__global__ void k_spmv(const uint32_t* __restrict__ ptr, const uint32_t* __restrict__ col, const uint32_t* __restrict__ val,
                       const uint32_t* __restrict__ w, uint32_t* __restrict__ out, uint32_t n) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    Fr acc = fe_zero<FrParams>();
    for (uint32_t e = ptr[g]; e < ptr[g + 1]; e++)
        acc = fe_add(acc, fe_mul(fe_load<FrParams>(val + 8 * (uint64_t)e), fe_load<FrParams>(w + 8 * (uint64_t)col[e])));
    fe_store<FrParams>(out + 8 * (uint64_t)g, acc);
}
__global__ void k_check_r1cs(const uint32_t* a, const uint32_t* b, const uint32_t* c, uint32_t n, int* flag) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    Fr p = fe_mul(fe_load<FrParams>(a + 8 * (uint64_t)g), fe_load<FrParams>(b + 8 * (uint64_t)g));
    if (!fe_eq(p, fe_load<FrParams>(c + 8 * (uint64_t)g))) atomicOr(flag, 1);
}
// witness -> Montgomery; bit 1 of *flag reports a value >= r
__global__ void k_fr_to_mont_flag2(uint32_t* dst, const uint32_t* src, uint64_t n, int* flag) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr a = fe_load<FrParams>(src + 8 * i);
    if (!fe_is_canonical(a)) atomicOr(flag, 2);
    fe_store<FrParams>(dst + 8 * i, fe_to_mont(a));
}
// out[i] = i < n_in ? in[i] * (tab ? tab[i] : 1) : 0      for i < total
__global__ void k_scale_pad(uint32_t* out, const uint32_t* in, const uint32_t* tab, uint64_t n_in, uint64_t total) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    Fr x = fe_zero<FrParams>();
    if (i < n_in) {
        x = fe_load<FrParams>(in + 8 * i);
        if (tab) x = fe_mul(x, fe_load<FrParams>(tab + 8 * i));
    }
    fe_store<FrParams>(out + 8 * i, x);
}
// batched form: vector k = i / out_stride, element j = i % out_stride:  out[i] = j < n_in ? in[k * in_stride + j] * (tab ? tab[j] : 1) : 0
__global__ void k_scale_pad2(uint32_t* out, const uint32_t* in, const uint32_t* tab, uint64_t n_in, uint64_t in_stride, uint64_t out_stride, uint64_t total) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint64_t k = i / out_stride, j = i % out_stride;
    Fr x = fe_zero<FrParams>();
    if (j < n_in) {
        x = fe_load<FrParams>(in + 8 * (k * in_stride + j));
        if (tab) x = fe_mul(x, fe_load<FrParams>(tab + 8 * j));
    }
    fe_store<FrParams>(out + 8 * i, x);
}
// Tree levels 1..levels (node length <= tile) in ONE kernel: the tile stays in LDS across the levels,
// per level  node <- lo + P_left * hi  = {hi zero-padded -> NTT -> * table -> iNTT -> + lo}.
// HBM sees the coefficients once in and once out (plus the per-level tables) instead of once per level.
__global__ __launch_bounds__(NTT_THREADS) void k_tree_levels_fused(uint32_t* __restrict__ d, const uint32_t* __restrict__ pntt,
                                                                  const uint32_t* __restrict__ tw_fwd, const uint32_t* __restrict__ tw_inv,
                                                                  uint32_t levels, uint32_t log_T, uint64_t n2) {
    __shared__ NttTile lds;       // transform workspace
    __shared__ NttTile cur;       // the tile's coefficients between levels, each < 2r
    const uint32_t T = 1u << log_T;
    const uint64_t base = (uint64_t)blockIdx.x << log_T;
    for (uint32_t e = threadIdx.x; e < T; e += NTT_THREADS) fr9_lds_put(cur, e, fr9_load(d + 8 * (base + e)));
    __syncthreads();
    for (uint32_t lv = 1; lv <= levels; lv++) {
        const uint32_t half = 1u << (lv - 1), lm = (half << 1) - 1;
        const uint32_t* tab = pntt + 8 * (uint64_t)(lv - 1) * n2;
        for (uint32_t e = threadIdx.x; e < T; e += NTT_THREADS) {
            const bool low = (e & lm) < half;
#pragma unroll
            for (int l = 0; l < FR29_L; l++) lds[l][e] = low ? cur[l][e + half] : 0u;
        }
        __syncthreads();
        lds_ntt_stages<false>(lds, tw_fwd, T, lv, 0, 0, 0, false);
        for (uint32_t e = threadIdx.x; e < T; e += NTT_THREADS)
            fr9_lds_put(lds, e, fr9_mul(fr9_lds_get(lds, e), fr9_load(tab + 8 * ((base + e) & (n2 - 1)))));
        __syncthreads();
        lds_ntt_stages<true>(lds, tw_inv, T, lv, 0, 0, 0, false);
        for (uint32_t e = threadIdx.x; e < T; e += NTT_THREADS) {
            Fr9 x = fr9_lds_get(lds, e);                                  // < 42 r
            if ((e & lm) < half) x = fr9_add(x, fr9_lds_get(cur, e));
            fr9_lds_put(cur, e, fr9_reduce_weak(x));
        }
        __syncthreads();
    }
    for (uint32_t e = threadIdx.x; e < T; e += NTT_THREADS) fr9_store(d + 8 * (base + e), fr9_lds_get(cur, e));
}
// out[k] = k < cnt ? in[top - k] : 0    (reversal of a coefficient window, zero padded)
__global__ void k_reverse_pad(uint32_t* out, const uint32_t* in, uint64_t top, uint64_t cnt, uint64_t total) {
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= total) return;
    Fr x = fe_zero<FrParams>();
    if (k < cnt) x = fe_load<FrParams>(in + 8 * (top - k));
    fe_store<FrParams>(out + 8 * k, x);
}

// ---- per-n tables
// chunk products for the factorial scan: prod[c] = product of max(i,1) over chunk c
static constexpr uint32_t FCH = 256;
__global__ void k_fact_chunk_prod(uint32_t* prod, uint32_t n2) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c * FCH >= n2) return;
    Fr acc = fe_one<FrParams>();
    for (uint32_t i = c * FCH; i < min((c + 1) * FCH, n2); i++)
        if (i) acc = fe_mul(acc, fe_from_u32<FrParams>(i));
    fe_store<FrParams>(prod + 8 * c, acc);
}
__global__ void k_fact_chunk_scan(uint32_t* prod, uint32_t nchunks) {   // exclusive prefix products, one lane
    if (blockIdx.x || threadIdx.x) return;
    Fr run = fe_one<FrParams>();
    for (uint32_t c = 0; c < nchunks; c++) {
        Fr p = fe_load<FrParams>(prod + 8 * c);
        fe_store<FrParams>(prod + 8 * c, run);
        run = fe_mul(run, p);
    }
}
// invfact[i] = 1/i!; alt[i] = (-1)^i / i!
__global__ void k_invfact(uint32_t* invfact, uint32_t* alt, const uint32_t* prefix, uint32_t n2) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c * FCH >= n2) return;
    uint32_t lo = c * FCH, hi = min((c + 1) * FCH, n2);
    Fr f = fe_load<FrParams>(prefix + 8 * c);             // (lo-1)!  (1 for c = 0)
    for (uint32_t i = lo; i < hi; i++)
        if (i) f = fe_mul(f, fe_from_u32<FrParams>(i));
    Fr inv = fe_inv(f);                                    // 1/(hi-1)!
    for (uint32_t i = hi; i-- > lo;) {
        fe_store<FrParams>(invfact + 8 * i, inv);
        fe_store<FrParams>(alt + 8 * i, (i & 1) ? fe_neg(inv) : inv);
        if (i) inv = fe_mul(inv, fe_from_u32<FrParams>(i));
    }
}
// leaves of the subproduct tree: q0[s] = -s   (Q_{0,s} = X - s, leading 1 implicit)
__global__ void k_tree_leaves(uint32_t* q, uint32_t n2, uint32_t offset) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n2) return;
    fe_store<FrParams>(q + 8 * s, fe_neg(fe_from_u32<FrParams>(s + offset)));
}
// scratch[node*2len + j] = j < len ? q[node*len + j] : (j == len ? 1 : 0)    (monic, padded to 2 len)
__global__ void k_tree_expand(uint32_t* scratch, const uint32_t* q, uint32_t log_len, uint64_t total2) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total2) return;
    uint64_t len = (uint64_t)1 << log_len, node = i >> (log_len + 1), j = i & ((len << 1) - 1);
    Fr x = fe_zero<FrParams>();
    if (j < len) x = fe_load<FrParams>(q + 8 * (node * len + j));
    else if (j == len) x = fe_one<FrParams>();
    fe_store<FrParams>(scratch + 8 * i, x);
}
// from N = NTT_{2len}(Q nodes): pntt_next[parent*2len + j] = N[left][j] * scale;  prod[parent*2len + j] = N[left][j]*N[right][j]
__global__ void k_tree_pair(uint32_t* pntt_next, uint32_t* prod, const uint32_t* N, const uint32_t* scale, uint32_t log_len2, uint64_t total) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    uint64_t len2 = (uint64_t)1 << log_len2, parent = i >> log_len2, j = i & (len2 - 1);
    Fr l = fe_load<FrParams>(N + 8 * ((2 * parent) * len2 + j));
    Fr r = fe_load<FrParams>(N + 8 * ((2 * parent + 1) * len2 + j));
    fe_store<FrParams>(pntt_next + 8 * i, fe_mul(l, fe_load<FrParams>(scale)));
    fe_store<FrParams>(prod + 8 * i, fe_mul(l, r));
}
// the LEFT children of a level, expanded like k_tree_expand: out[parent * 2 len + j] = j < len ? q[(2 parent) * len + j] : (j == len ? 1 : 0)
__global__ void k_tree_expand_left(uint32_t* out, const uint32_t* q, uint32_t log_len, uint64_t total) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    uint64_t len = (uint64_t)1 << log_len, parent = i >> (log_len + 1), j = i & ((len << 1) - 1);
    Fr x = fe_zero<FrParams>();
    if (j < len) x = fe_load<FrParams>(q + 8 * ((2 * parent) * len + j));
    else if (j == len) x = fe_one<FrParams>();
    fe_store<FrParams>(out + 8 * i, x);
}
// cyclic wrap of the monic leading term: q[parent*len2 + 0] -= 1
__global__ void k_tree_unwrap(uint32_t* q, uint32_t log_len2, uint64_t nodes) {
    uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nodes) return;
    uint32_t* a = q + 8 * (p << log_len2);
    fe_store<FrParams>(a, fe_sub(fe_load<FrParams>(a), fe_one<FrParams>()));
}
__global__ void k_set_one(uint32_t* p) {
    if (blockIdx.x || threadIdx.x) return;
    fe_store<FrParams>(p, fe_one<FrParams>());
}
// t = 2 - t  (power-series Newton step), first cnt entries
__global__ void k_two_minus(uint32_t* t, uint64_t cnt) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    Fr x = fe_neg(fe_load<FrParams>(t + 8 * i));
    if (i == 0) x = fe_add(x, fe_from_u32<FrParams>(2));
    fe_store<FrParams>(t + 8 * i, x);
}
__global__ void k_inv_pow2_of(uint32_t* out, uint32_t k) {   // 2^-k
    if (blockIdx.x || threadIdx.x) return;
    Fr half = fe_inv(fe_from_u32<FrParams>(2)), acc = fe_one<FrParams>();
    for (uint32_t i = 0; i < k; i++) acc = fe_mul(acc, half);
    fe_store<FrParams>(out, acc);
}

// ------------------------------------------------------------------ host helpers
// ------------------------------------------------------------------ Lagrange-form keys (scope row f4)
// With bases [l_i(tau)] instead of [tau^k] the prover needs v, w only through their VALUES a_i, b_i, and h through
// its values on n - 1 further points j = n + t: for equally spaced points
//   v(j) = Z(j) * sum_i (a_i c_i) / (j - i),   c_i = (-1)^(n-1-i) / (i! (n-1-i)!),   Z(j) = j! / (j-n)!
// i.e. ONE cyclic convolution of size S = 2 n2 with the fixed kernel g[e] = 1/e per polynomial, and
//   h(j) = (v(j) w(j) - y(j)) / Z(j) = Z(j) Sa_j Sb_j - Sc_j .
// 3 convolutions per proof instead of the ~25 of the basis conversion.
__global__ void k_inv_range(uint32_t* __restrict__ out, uint32_t total) {          // out[e] = 1/e, out[0] = 0; chunks of 256 per lane
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lo = c * FCH, hi = min((c + 1) * FCH, total);
    if (lo >= total) return;
    Fr run = fe_one<FrParams>();
    for (uint32_t e = lo; e < hi; e++) {                  // prefix products into out
        fe_store<FrParams>(out + 8 * (uint64_t)e, run);
        if (e) run = fe_mul(run, fe_from_u32<FrParams>(e));
    }
    Fr inv = fe_inv(run);                                  // 1 / prod_{e in chunk, e > 0} e
    for (uint32_t e = hi; e-- > lo;) {
        const Fr pre = fe_load<FrParams>(out + 8 * (uint64_t)e);
        if (e) {
            fe_store<FrParams>(out + 8 * (uint64_t)e, fe_mul(inv, pre));
            inv = fe_mul(inv, fe_from_u32<FrParams>(e));
        } else fe_store<FrParams>(out + 8 * (uint64_t)e, fe_zero<FrParams>());
    }
}
// zt[t] = (n+t)! / t! for t < cnt: prefix[c] = (256 c - 1)! over chunks of e; invfact[t] = 1/t!
__global__ void k_zt(uint32_t* __restrict__ zt, const uint32_t* __restrict__ prefix, const uint32_t* __restrict__ invfact, uint32_t n, uint32_t cnt) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lo = c * FCH, hi = min((c + 1) * FCH, cnt);
    if (lo >= cnt) return;
    const uint32_t j0 = n + lo, ch = j0 / FCH;
    Fr f = fe_load<FrParams>(prefix + 8 * (uint64_t)ch);   // (256 ch - 1)!  (1 for ch = 0)
    for (uint32_t e = ch * FCH; e <= j0; e++)
        if (e) f = fe_mul(f, fe_from_u32<FrParams>(e));     // j0!
    for (uint32_t t = lo; t < hi; t++) {
        fe_store<FrParams>(zt + 8 * (uint64_t)t, fe_mul(f, fe_load<FrParams>(invfact + 8 * (uint64_t)t)));
        f = fe_mul(f, fe_from_u32<FrParams>(n + t + 1));
    }
}
// out[i] = i < n ? x_i * c_i : 0   for i < total
__global__ void k_lag_scale(uint32_t* __restrict__ out, const uint32_t* __restrict__ x, const uint32_t* __restrict__ invfact, uint32_t n, uint64_t total) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    Fr r = fe_zero<FrParams>();
    if (i < n) {
        r = fe_mul(fe_load<FrParams>(x + 8 * i), fe_mul(fe_load<FrParams>(invfact + 8 * i), fe_load<FrParams>(invfact + 8 * (uint64_t)(n - 1 - i))));
        if ((n - 1 - i) & 1) r = fe_neg(r);
    }
    fe_store<FrParams>(out + 8 * i, r);
}
__global__ void k_lag_h(uint32_t* __restrict__ h, const uint32_t* __restrict__ sa, const uint32_t* __restrict__ sb, const uint32_t* __restrict__ sc,
                        const uint32_t* __restrict__ zt, uint32_t n) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t + 1 >= n) return;
    const uint64_t j = (uint64_t)n + t;
    const Fr v = fe_mul(fe_mul(fe_load<FrParams>(zt + 8 * (uint64_t)t), fe_load<FrParams>(sa + 8 * j)), fe_load<FrParams>(sb + 8 * j));
    fe_store<FrParams>(h + 8 * (uint64_t)t, fe_sub(v, fe_load<FrParams>(sc + 8 * j)));
}

int dev_poly_mul(const void* d_a, uint64_t na, const void* d_b, uint64_t nb, void* d_out, hipStream_t s) {
    if (!na || !nb) return ZK_OK;
    uint64_t nout = na + nb - 1;
    uint32_t lg = ceil_log2(nout);
    uint64_t S = (uint64_t)1 << lg;
    DevBuf A, B;
    ZKCHK(A.alloc(32 * S));
    ZKCHK(B.alloc(32 * S));
    hipLaunchKernelGGL(k_scale_pad, g1d(S), dim3(256), 0, s, FRP(A), (const uint32_t*)d_a, (const uint32_t*)nullptr, na, S);
    hipLaunchKernelGGL(k_scale_pad, g1d(S), dim3(256), 0, s, FRP(B), (const uint32_t*)d_b, (const uint32_t*)nullptr, nb, S);
    if (lg == 0) ZKCHK(fr_pointwise_mul(A.p, A.p, B.p, S, s));
    else if (lg <= RNS_MAX_LOG && rns_enabled()) {
        RnsWork wa, wb;
        ZKCHK(rns_conv_data(wa, wb, A.p, B.p, lg, A.p, 0, nout, s));
        HIPCHK(hipStreamSynchronize(s));          // wa, wb die here
    } else {
        ZKCHK(ntt_forward(B.p, S, lg, s));
        ZKCHK(ntt_mul_table(A.p, S, lg, B.p, S - 1, true, nullptr, nullptr, s, true));
    }
    HIPCHK(hipMemcpyAsync(d_out, A.p, 32 * nout, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipStreamSynchronize(s));   // A, B are freed on return
    return ZK_OK;
}

static int upload_csr(CsrDev& d, const zk_csr* h, uint32_t n, uint32_t m, hipStream_t s) {
    if (!h || !h->row_ptr) ZK_FAIL(ZK_ERR_ARG, "R1CS matrix: null row_ptr");
    uint64_t nnz = h->row_ptr[n];
    for (uint32_t g = 0; g < n; g++)
        if (h->row_ptr[g] > h->row_ptr[g + 1]) ZK_FAIL(ZK_ERR_ARG, "R1CS matrix: row_ptr not monotone");
    if (nnz && (!h->col || !h->val)) ZK_FAIL(ZK_ERR_ARG, "R1CS matrix: null col/val");
    for (uint64_t e = 0; e < nnz; e++)
        if (h->col[e] >= m) ZK_FAIL(ZK_ERR_ARG, "R1CS matrix: column index out of range");
    d.nnz = nnz;
    ZKCHK(d.ptr.alloc(4 * (size_t)(n + 1)));
    ZKCHK(d.col.alloc(4 * (size_t)(nnz ? nnz : 1)));
    ZKCHK(d.val.alloc(32 * (size_t)(nnz ? nnz : 1)));
    HIPCHK(hipMemcpyAsync(d.ptr.p, h->row_ptr, 4 * (size_t)(n + 1), hipMemcpyHostToDevice, s));
    if (nnz) {
        HIPCHK(hipMemcpyAsync(d.col.p, h->col, 4 * nnz, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(d.val.p, h->val, 32 * nnz, hipMemcpyHostToDevice, s));
        DevBuf flag;
        ZKCHK(flag.alloc(4));
        HIPCHK(hipMemsetAsync(flag.p, 0, 4, s));
        ZKCHK(fr_to_mont(d.val.p, d.val.p, nnz, flag.as<int>(), s));
        int hf = 0;
        HIPCHK(hipMemcpyAsync(&hf, flag.p, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (hf) ZK_FAIL(ZK_ERR_SCALAR_RANGE, "R1CS coefficient >= r");
    }
    return ZK_OK;
}

// Newton -> monomial for `batch` vectors of n2 coefficients stored back to back in d (in place).
static int tree_convert(const FrStage& f, void* d, uint32_t batch, uint32_t levels, void* tmp, hipStream_t s, RnsWork* rw = nullptr) {
    Ctx& c = ctx();
    const uint64_t total = (uint64_t)batch * f.n2;
    const uint32_t log_total = ceil_log2(total);
    const uint32_t log_T = log_total < (uint32_t)NTT_LOG_T ? log_total : NTT_LOG_T;
    ZKCHK(ntt_ensure_twiddles(levels));
    ScopedTimer t("fr_tree", s);
    const uint32_t fused = levels < log_T ? levels : log_T;
    hipLaunchKernelGGL(k_tree_levels_fused, dim3((unsigned)(total >> log_T)), dim3(NTT_THREADS), 0, s, (uint32_t*)d, (const uint32_t*)FRP(f.pntt),
                       (const uint32_t*)c.tw_fwd, (const uint32_t*)c.tw_inv, fused, log_T, (uint64_t)f.n2);
    for (uint32_t l = fused + 1; l <= levels; l++) {
        const uint32_t* tab = FRP(f.pntt) + 8 * (uint64_t)(l - 1) * f.n2;
        if (rw && f.rns_ok && l >= f.rns_first_level && l < f.p_rns.size() && f.p_rns[l].p) {
            // the same node product as an integer convolution modulo 18 small primes (rns_ntt.cuh): upper halves in, sums with the lower halves out
            ZKCHK(rns_conv_table(*rw, d, total, l, RNS_IN_TREE_HI, f.p_rns[l], (uint64_t)f.n2 - 1, d, RNS_OUT_TREE_ADD, 0, 0, 0, s));
            continue;
        }
        // upper halves (zero padded) -> NTT -> * P_left -> iNTT -> + lower halves, with the padding folded
        // into the first pass, the product into the middle kernel and the addition into the last pass
        ZKCHK(ntt_mul_table(tmp, total, l, tab, (uint64_t)f.n2 - 1, false, d, d, s));
    }
    HIPCHK(hipGetLastError());
    return ZK_OK;
}

// Tables of the subproduct tree over the points offset .. offset + n2 - 1: level l (0-based) holds, for every node of 2^(l+1)
// points, NTT_{2^(l+1)} of the (monic, padded) subproduct of its LEFT half, times 2^-(l+1) -- plain Montgomery form, transform-domain
// (bit-reversed) order, n2 entries per level at pntt + l * n2.  `q` returns the root product (low n2 coefficients, monic).
// offset 0: the prover's basis conversion; offset n: the h bases of a derived Lagrange-form key (lagrange_derive.hip).
int frstage_tree_tables(uint32_t n2, uint32_t log_n2, uint32_t offset, void* pntt, DevBuf& q, hipStream_t s, std::vector<DevBuf>* p_rns, uint32_t rns_first_level) {
    DevBuf N, prod, scale;
    ZKCHK(ntt_ensure_twiddles(log_n2 + 1));
    ZKCHK(q.alloc(32 * (size_t)n2));
    ZKCHK(N.alloc(32 * (size_t)2 * n2));
    ZKCHK(prod.alloc(32 * (size_t)n2));
    ZKCHK(scale.alloc(32));
    hipLaunchKernelGGL(k_tree_leaves, g1d(n2), dim3(256), 0, s, FRP(q), n2, offset);
    for (uint32_t l = 0; l < log_n2; l++) {
        // N = NTT_{2^(l+1)} of every level-l node (monic, padded): 2*n2 entries
        if (p_rns && l + 1 >= rns_first_level) {
            // the same table for the convolutions through the residue number system: the transform (there) of the left children's coefficient form
            hipLaunchKernelGGL(k_tree_expand_left, g1d(n2), dim3(256), 0, s, FRP(prod), (const uint32_t*)FRP(q), l, (uint64_t)n2);
            ZKCHK(rns_table_build((*p_rns)[l + 1], prod.p, n2, l + 1, s));
        }
        hipLaunchKernelGGL(k_tree_expand, g1d(2 * (uint64_t)n2), dim3(256), 0, s, FRP(N), (const uint32_t*)FRP(q), l, 2 * (uint64_t)n2);
        ZKCHK(ntt_forward(N.p, 2 * (uint64_t)n2, l + 1, s));
        hipLaunchKernelGGL(k_inv_pow2_of, dim3(1), dim3(64), 0, s, FRP(scale), l + 1);
        hipLaunchKernelGGL(k_tree_pair, g1d(n2), dim3(256), 0, s, (uint32_t*)pntt + 8 * (uint64_t)l * n2, FRP(prod), (const uint32_t*)FRP(N),
                           (const uint32_t*)FRP(scale), l + 1, (uint64_t)n2);
        ZKCHK(ntt_inverse(prod.p, n2, l + 1, true, s));
        hipLaunchKernelGGL(k_tree_unwrap, g1d(n2 >> (l + 1)), dim3(256), 0, s, FRP(prod), l + 1, (uint64_t)(n2 >> (l + 1)));
        HIPCHK(hipMemcpyAsync(q.p, prod.p, 32 * (size_t)n2, hipMemcpyDeviceToDevice, s));
    }
    HIPCHK(hipStreamSynchronize(s));       // N, prod, scale die here
    return ZK_OK;
}

int frstage_init(FrStage& f, uint32_t n, uint32_t m, const zk_csr* L, const zk_csr* R, const zk_csr* O, hipStream_t s) {
    if (n < 1 || n > (1u << 24)) ZK_FAIL(ZK_ERR_ARG, "constraint count must be in [1, 2^24]");
    if (m == 0) ZK_FAIL(ZK_ERR_ARG, "no variables");
    f.n = n; f.m = m;
    f.log_n2 = ceil_log2(n); f.n2 = 1u << f.log_n2;
    f.log_S = f.log_n2 + 1; f.S = 1u << f.log_S;
    ZKCHK(upload_csr(f.L, L, n, m, s));
    ZKCHK(upload_csr(f.R, R, n, m, s));
    ZKCHK(upload_csr(f.O, O, n, m, s));
    if (n == 1) {
        // ONE gate: v, w, y are the constants (L w)_0, (R w)_0, (O w)_0, Z = X, h = 0 (no coefficient at all: QAP.ml:132-135 divides a
        // polynomial that must vanish) -- the values ARE the coefficients, no table is needed, only Z for the callers that blind with it
        ZKCHK(f.z.alloc(64));
        HIPCHK(hipMemsetAsync(f.z.p, 0, 64, s));
        hipLaunchKernelGGL(k_set_one, dim3(1), dim3(64), 0, s, FRP(f.z) + 8);
        HIPCHK(hipStreamSynchronize(s));
        return ZK_OK;
    }
    ZKCHK(ntt_ensure_twiddles(f.log_S));
    const uint32_t n2 = f.n2, S = f.S;
    // round 4: convolutions through the residue number system where the transforms fit it (rns_ntt.cuh) -- an OPTION (ZK_FR_RNS=1 when the key is
    // uploaded: the tables and the residue arrays of the slots cost 72 bytes per element; measured slower than the Fr transforms in its present form,
    // DESIGN.md A.2), kept under parity by the GPU suite.  The levels of the fused LDS kernel keep the Fr transform.
    f.rns_ok = f.log_S <= RNS_MAX_LOG && rns_enabled();
    f.rns_first_level = (f.log_n2 < (uint32_t)NTT_LOG_T ? f.log_n2 : (uint32_t)NTT_LOG_T) + 1;
    if (f.rns_ok) {
        f.p_rns.clear();
        f.p_rns.resize(f.log_n2 + 1);
    }
    ZKCHK(f.invfact.alloc(32 * (size_t)n2));
    ZKCHK(f.e_ntt.alloc(32 * (size_t)S));
    ZKCHK(f.pntt.alloc(32 * (size_t)n2 * (f.log_n2 ? f.log_n2 : 1)));
    ZKCHK(f.iz_ntt.alloc(32 * (size_t)S));
    ZKCHK(f.z.alloc(32 * (size_t)(n + 1)));
    FrScratch sc0;                       // only used to derive Z for n that is not a power of two
    ZKCHK(frstage_scratch_alloc(f, sc0));

    // ---- 1/i! and the alternating kernel of the Newton convolution
    {
        uint32_t nch = (n2 + FCH - 1) / FCH;
        DevBuf prod, alt;
        ZKCHK(prod.alloc(32 * (size_t)nch));
        ZKCHK(alt.alloc(32 * (size_t)n2));
        hipLaunchKernelGGL(k_fact_chunk_prod, g1d(nch, 64), dim3(64), 0, s, FRP(prod), n2);
        hipLaunchKernelGGL(k_fact_chunk_scan, dim3(1), dim3(64), 0, s, FRP(prod), nch);
        hipLaunchKernelGGL(k_invfact, g1d(nch, 64), dim3(64), 0, s, FRP(f.invfact), FRP(alt), (const uint32_t*)FRP(prod), n2);
        hipLaunchKernelGGL(k_scale_pad, g1d(S), dim3(256), 0, s, FRP(f.e_ntt), (const uint32_t*)FRP(alt), (const uint32_t*)nullptr, (uint64_t)n2, (uint64_t)S);
        if (f.rns_ok) ZKCHK(rns_table_build(f.e_rns, f.e_ntt.p, S, f.log_S, s));          // from the coefficient form, before the Fr transform overwrites it
        ZKCHK(ntt_forward(f.e_ntt.p, S, f.log_S, s));
        ZKCHK(fr_to_factor(f.e_ntt.p, S, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    // ---- subproduct tree, bottom-up.  q holds Q_{l,node} (low 2^l coefficients, monic), n2 entries per level.
    {
        DevBuf q;
        ZKCHK(frstage_tree_tables(n2, f.log_n2, 0, f.pntt.p, q, s, f.rns_ok ? &f.p_rns : nullptr, f.rns_first_level));
        ZKCHK(fr_to_factor(f.pntt.p, (uint64_t)n2 * f.log_n2, s));
        // ---- Z = prod_{i<n} (X - i)
        if (n == n2) {
            HIPCHK(hipMemcpyAsync(f.z.p, q.p, 32 * (size_t)n, hipMemcpyDeviceToDevice, s));   // root product, monic
            hipLaunchKernelGGL(k_set_one, dim3(1), dim3(64), 0, s, FRP(f.z) + 8 * (uint64_t)n);
        } else {
            // Z = X^(n falling) : Newton coordinates e_n, converted through the tree (n < n2)
            HIPCHK(hipMemsetAsync(sc0.d.p, 0, 32 * (size_t)n2, s));
            hipLaunchKernelGGL(k_set_one, dim3(1), dim3(64), 0, s, FRP(sc0.d) + 8 * (uint64_t)n);
            ZKCHK(tree_convert(f, sc0.d.p, 1, f.log_n2, sc0.tmp.p, s));
            HIPCHK(hipMemcpyAsync(f.z.p, sc0.d.p, 32 * (size_t)(n + 1), hipMemcpyDeviceToDevice, s));
        }
        HIPCHK(hipStreamSynchronize(s));
    }
    // ---- (rev Z)^-1 mod X^(n-1) by Newton iteration g <- g (2 - f g), then its NTT_S
    {
        const uint64_t need = n - 1;
        DevBuf fz, g, t1, t2;
        ZKCHK(fz.alloc(32 * (size_t)(n + 1)));
        ZKCHK(g.alloc(32 * (size_t)(2 * need + 2)));
        ZKCHK(t1.alloc(32 * (size_t)(4 * need + 4)));
        ZKCHK(t2.alloc(32 * (size_t)(4 * need + 4)));
        hipLaunchKernelGGL(k_reverse_pad, g1d(n + 1), dim3(256), 0, s, FRP(fz), (const uint32_t*)FRP(f.z), (uint64_t)n, (uint64_t)n + 1, (uint64_t)n + 1);
        hipLaunchKernelGGL(k_set_one, dim3(1), dim3(64), 0, s, FRP(g));   // rev Z has constant term 1
        uint64_t prec = 1;
        while (prec < need) {
            uint64_t p2 = prec * 2 < need ? prec * 2 : need;
            uint64_t fl = p2 < (uint64_t)n + 1 ? p2 : (uint64_t)n + 1;
            ZKCHK(dev_poly_mul(fz.p, fl, g.p, prec, t1.p, s));                 // f*g, keep p2 terms
            uint64_t got = fl + prec - 1;
            if (got < p2) HIPCHK(hipMemsetAsync((char*)t1.p + 32 * got, 0, 32 * (p2 - got), s));
            hipLaunchKernelGGL(k_two_minus, g1d(p2), dim3(256), 0, s, FRP(t1), p2);
            ZKCHK(dev_poly_mul(g.p, prec, t1.p, p2, t2.p, s));                 // g*(2 - f g) mod X^p2
            HIPCHK(hipMemcpyAsync(g.p, t2.p, 32 * p2, hipMemcpyDeviceToDevice, s));
            prec = p2;
        }
        hipLaunchKernelGGL(k_scale_pad, g1d(S), dim3(256), 0, s, FRP(f.iz_ntt), (const uint32_t*)FRP(g), (const uint32_t*)nullptr, need, (uint64_t)S);
        if (f.rns_ok) ZKCHK(rns_table_build(f.iz_rns, f.iz_ntt.p, S, f.log_S, s));
        ZKCHK(ntt_forward(f.iz_ntt.p, S, f.log_S, s));
        ZKCHK(fr_to_factor(f.iz_ntt.p, S, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    HIPCHK(hipGetLastError());
    return ZK_OK;
}

int frstage_scratch_alloc(const FrStage& f, FrScratch& sc) {
    ZKCHK(sc.wit.alloc(32 * (size_t)f.m));
    ZKCHK(sc.abc.alloc(32 * (size_t)3 * f.n));
    ZKCHK(sc.d.alloc(32 * (size_t)2 * f.n2));
    ZKCHK(sc.tmp.alloc(32 * (size_t)2 * f.n2));
    ZKCHK(sc.bufA.alloc(32 * (size_t)2 * f.S));          // two convolution buffers back to back: A = [0, S), B = [S, 2S)
    ZKCHK(sc.h.alloc(32 * (size_t)f.n));
    ZKCHK(sc.flag.alloc(4));
    if (f.rns_ok && f.n > 1) {          // residue arrays of the convolutions, allocated HERE: a proof may be enqueued under stream capture (ZK_GRAPH), where nothing can be allocated
        ZKCHK(sc.rns_a.ensure(f.lagrange ? (uint64_t)f.S : 2 * (uint64_t)f.S));          // the Newton step of the tau-power path transforms both vectors in one batch
        if (!f.lagrange) ZKCHK(sc.rns_b.ensure((uint64_t)f.S));
    }
    return ZK_OK;
}

int frstage_eval(const FrStage& f, FrScratch& sc, const void* d_wit_canon, hipStream_t s) {
    HIPCHK(hipGetLastError());           // a failed launch of an earlier call must not be blamed on this one
    const uint32_t n = f.n, n2 = f.n2, S = f.S;
    const bool rns = f.rns_ok && n > 1 && rns_enabled();
    uint32_t* a = FRP(sc.abc);
    uint32_t* b = a + 8 * (uint64_t)n;
    uint32_t* cc = b + 8 * (uint64_t)n;
    HIPCHK(hipMemsetAsync(sc.flag.p, 0, 4, s));
    {
        ScopedTimer t("fr_spmv", s);
        // witness -> Montgomery; a non-canonical value sets bit 0 of a scratch flag which we fold into bit 1
        DevBuf& w = sc.wit;
        hipLaunchKernelGGL(k_fr_to_mont_flag2, g1d(f.m), dim3(256), 0, s, FRP(w), (const uint32_t*)d_wit_canon, (uint64_t)f.m, sc.flag.as<int>());
        hipLaunchKernelGGL(k_spmv, g1d(n), dim3(256), 0, s, (const uint32_t*)FRP(f.L.ptr), (const uint32_t*)FRP(f.L.col), (const uint32_t*)FRP(f.L.val), (const uint32_t*)FRP(w), a, n);
        hipLaunchKernelGGL(k_spmv, g1d(n), dim3(256), 0, s, (const uint32_t*)FRP(f.R.ptr), (const uint32_t*)FRP(f.R.col), (const uint32_t*)FRP(f.R.val), (const uint32_t*)FRP(w), b, n);
        hipLaunchKernelGGL(k_spmv, g1d(n), dim3(256), 0, s, (const uint32_t*)FRP(f.O.ptr), (const uint32_t*)FRP(f.O.col), (const uint32_t*)FRP(f.O.val), (const uint32_t*)FRP(w), cc, n);
        hipLaunchKernelGGL(k_check_r1cs, g1d(n), dim3(256), 0, s, (const uint32_t*)a, (const uint32_t*)b, (const uint32_t*)cc, n, sc.flag.as<int>());
        HIPCHK(hipGetLastError());
    }
    if (n == 1) {      // one gate: the values are the (constant) polynomials; h is empty
        HIPCHK(hipMemcpyAsync(sc.d.p, a, 32, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(FRP(sc.d) + 8 * (uint64_t)n2, b, 32, hipMemcpyDeviceToDevice, s));
        return ZK_OK;
    }
    // ---- values -> Newton coefficients (both vectors), into d[0..n2) and d[n2..2 n2)
    {
        ScopedTimer t("fr_newton", s);
        // both vectors as two nodes of one batched convolution
        hipLaunchKernelGGL(k_scale_pad2, g1d(2 * (uint64_t)S), dim3(256), 0, s, FRP(sc.bufA), (const uint32_t*)a, (const uint32_t*)FRP(f.invfact), (uint64_t)n, (uint64_t)n,
                           (uint64_t)S, 2 * (uint64_t)S);
        if (rns) ZKCHK(rns_conv_table(sc.rns_a, sc.bufA.p, 2 * (uint64_t)S, f.log_S, RNS_IN_PLAIN, f.e_rns, (uint64_t)S - 1, sc.bufA.p, RNS_OUT_RANGE, 0, n, (uint64_t)S, s));
        else ZKCHK(ntt_mul_table(sc.bufA.p, 2 * (uint64_t)S, f.log_S, f.e_ntt.p, (uint64_t)S - 1, true, nullptr, nullptr, s));
        hipLaunchKernelGGL(k_scale_pad2, g1d(2 * (uint64_t)n2), dim3(256), 0, s, FRP(sc.d), (const uint32_t*)FRP(sc.bufA), (const uint32_t*)nullptr, (uint64_t)n, (uint64_t)S,
                           (uint64_t)n2, 2 * (uint64_t)n2);
    }
    // ---- Newton -> monomial
    ZKCHK(tree_convert(f, sc.d.p, 2, f.log_n2, sc.tmp.p, s, rns ? &sc.rns_a : nullptr));
    // ---- h
    {
        ScopedTimer t("fr_quotient", s);
        hipLaunchKernelGGL(k_scale_pad, g1d(S), dim3(256), 0, s, FRP(sc.bufA), (const uint32_t*)FRP(sc.d), (const uint32_t*)nullptr, (uint64_t)n, (uint64_t)S);
        hipLaunchKernelGGL(k_scale_pad, g1d(S), dim3(256), 0, s, (FRP(sc.bufA) + 8 * (uint64_t)S), (const uint32_t*)(FRP(sc.d) + 8 * (uint64_t)n2), (const uint32_t*)nullptr, (uint64_t)n, (uint64_t)S);
        if (rns) {          // v w: the coefficients n-1 .. 2n-2 are all the quotient needs
            ZKCHK(rns_conv_data(sc.rns_a, sc.rns_b, sc.bufA.p, (void*)(FRP(sc.bufA) + 8 * (uint64_t)S), f.log_S, (void*)(FRP(sc.bufA) + 8 * (uint64_t)(n - 1)), (uint64_t)n - 1, (uint64_t)n, s));
        } else {
            ZKCHK(ntt_forward((void*)(FRP(sc.bufA) + 8 * (uint64_t)S), S, f.log_S, s));
            ZKCHK(ntt_mul_table(sc.bufA.p, S, f.log_S, (void*)(FRP(sc.bufA) + 8 * (uint64_t)S), (uint64_t)S - 1, true, nullptr, nullptr, s, true));   // v*w, coefficients 0..2n-2
        }
        // t[k] = (v w)[2n-2-k], k < n-1
        hipLaunchKernelGGL(k_reverse_pad, g1d(S), dim3(256), 0, s, (FRP(sc.bufA) + 8 * (uint64_t)S), (const uint32_t*)FRP(sc.bufA), (uint64_t)(2 * (uint64_t)n - 2), (uint64_t)n - 1, (uint64_t)S);
        if (rns) ZKCHK(rns_conv_table(sc.rns_a, (void*)(FRP(sc.bufA) + 8 * (uint64_t)S), S, f.log_S, RNS_IN_PLAIN, f.iz_rns, (uint64_t)S - 1, (void*)(FRP(sc.bufA) + 8 * (uint64_t)S), RNS_OUT_RANGE, 0,
                                     (uint64_t)n - 1, 0, s));
        else ZKCHK(ntt_mul_table((void*)(FRP(sc.bufA) + 8 * (uint64_t)S), S, f.log_S, f.iz_ntt.p, (uint64_t)S - 1, true, nullptr, nullptr, s));
        // h[j] = hh[n-2-j], j < n-1
        hipLaunchKernelGGL(k_reverse_pad, g1d(n - 1), dim3(256), 0, s, FRP(sc.h), (const uint32_t*)(FRP(sc.bufA) + 8 * (uint64_t)S), (uint64_t)n - 2, (uint64_t)n - 1, (uint64_t)n - 1);
    }
    HIPCHK(hipGetLastError());
    return ZK_OK;
}

int frstage_init_lagrange(FrStage& f, hipStream_t s) {
    const uint32_t n = f.n, S = f.S;
    if (n == 1) { f.lagrange = true; return ZK_OK; }      // one gate: l_0 = 1, nothing to extrapolate
    ZKCHK(f.g_ntt.alloc(32 * (size_t)S));
    ZKCHK(f.zt.alloc(32 * (size_t)(n > 1 ? n - 1 : 1)));
    hipLaunchKernelGGL(k_inv_range, g1d((S + FCH - 1) / FCH, 64), dim3(64), 0, s, FRP(f.g_ntt), S);
    if (f.rns_ok) ZKCHK(rns_table_build(f.g_rns, f.g_ntt.p, S, f.log_S, s));
    ZKCHK(ntt_forward(f.g_ntt.p, S, f.log_S, s));
    ZKCHK(fr_to_factor(f.g_ntt.p, S, s));
    const uint32_t nch = (S + FCH - 1) / FCH;              // factorials up to 2 n <= S
    DevBuf prod;
    ZKCHK(prod.alloc(32 * (size_t)nch));
    hipLaunchKernelGGL(k_fact_chunk_prod, g1d(nch, 64), dim3(64), 0, s, FRP(prod), S);
    hipLaunchKernelGGL(k_fact_chunk_scan, dim3(1), dim3(64), 0, s, FRP(prod), nch);
    hipLaunchKernelGGL(k_zt, g1d((n - 1 + FCH - 1) / FCH, 64), dim3(64), 0, s, FRP(f.zt), (const uint32_t*)FRP(prod), (const uint32_t*)FRP(f.invfact), n, n - 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    f.lagrange = true;
    return ZK_OK;
}
int frstage_eval_lagrange(const FrStage& f, FrScratch& sc, const void* d_wit_canon, hipStream_t s) {
    HIPCHK(hipGetLastError());
    if (!f.lagrange) ZK_FAIL(ZK_ERR_ARG, "frstage_eval_lagrange: tables not initialised");
    const uint32_t n = f.n, S = f.S;
    uint32_t* a = FRP(sc.abc);
    uint32_t* b = a + 8 * (uint64_t)n;
    uint32_t* cc = b + 8 * (uint64_t)n;
    HIPCHK(hipMemsetAsync(sc.flag.p, 0, 4, s));
    {
        ScopedTimer t("fr_spmv", s);
        DevBuf& w = sc.wit;
        hipLaunchKernelGGL(k_fr_to_mont_flag2, g1d(f.m), dim3(256), 0, s, FRP(w), (const uint32_t*)d_wit_canon, (uint64_t)f.m, sc.flag.as<int>());
        hipLaunchKernelGGL(k_spmv, g1d(n), dim3(256), 0, s, (const uint32_t*)FRP(f.L.ptr), (const uint32_t*)FRP(f.L.col), (const uint32_t*)FRP(f.L.val), (const uint32_t*)FRP(w), a, n);
        hipLaunchKernelGGL(k_spmv, g1d(n), dim3(256), 0, s, (const uint32_t*)FRP(f.R.ptr), (const uint32_t*)FRP(f.R.col), (const uint32_t*)FRP(f.R.val), (const uint32_t*)FRP(w), b, n);
        hipLaunchKernelGGL(k_spmv, g1d(n), dim3(256), 0, s, (const uint32_t*)FRP(f.O.ptr), (const uint32_t*)FRP(f.O.col), (const uint32_t*)FRP(f.O.val), (const uint32_t*)FRP(w), cc, n);
        hipLaunchKernelGGL(k_check_r1cs, g1d(n), dim3(256), 0, s, (const uint32_t*)a, (const uint32_t*)b, (const uint32_t*)cc, n, sc.flag.as<int>());
        HIPCHK(hipGetLastError());
    }
    if (n > 1) {
        ScopedTimer t("fr_extrapolate", s);
        const uint32_t* src[3] = {a, b, cc};
        void* dst[3] = {sc.bufA.p, (void*)(FRP(sc.bufA) + 8 * (uint64_t)S), sc.tmp.p};
        for (int k = 0; k < 3; k++) {
            hipLaunchKernelGGL(k_lag_scale, g1d(S), dim3(256), 0, s, (uint32_t*)dst[k], src[k], (const uint32_t*)FRP(f.invfact), n, (uint64_t)S);
            if (f.rns_ok && f.g_rns.p && rns_enabled())          // only the values at n .. 2n-2 are read (k_lag_h)
                ZKCHK(rns_conv_table(sc.rns_a, dst[k], S, f.log_S, RNS_IN_PLAIN, f.g_rns, (uint64_t)S - 1, (void*)((uint32_t*)dst[k] + 8 * (uint64_t)n), RNS_OUT_RANGE, (uint64_t)n, (uint64_t)n - 1, 0, s));
            else ZKCHK(ntt_mul_table(dst[k], S, f.log_S, f.g_ntt.p, (uint64_t)S - 1, true, nullptr, nullptr, s));
        }
        hipLaunchKernelGGL(k_lag_h, g1d(n), dim3(256), 0, s, FRP(sc.h), (const uint32_t*)FRP(sc.bufA), (const uint32_t*)(FRP(sc.bufA) + 8 * (uint64_t)S),
                           (const uint32_t*)FRP(sc.tmp), (const uint32_t*)FRP(f.zt), n);
    }
    HIPCHK(hipGetLastError());
    return ZK_OK;
}

// ---- leading coefficients of the interpolants of a and b (frstage.cuh): block-wise partial sums of x_i c_i, then one small launch adds them
static constexpr uint32_t LEAD_THREADS = 256;
__global__ __launch_bounds__(LEAD_THREADS) void k_lead_partial(uint32_t* __restrict__ part, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                               const uint32_t* __restrict__ invfact, uint32_t n) {
    __shared__ __attribute__((aligned(32))) uint32_t sh[2][LEAD_THREADS][8];
    Fr sa = fe_zero<FrParams>(), sb = fe_zero<FrParams>();
    for (uint64_t i = (uint64_t)blockIdx.x * LEAD_THREADS + threadIdx.x; i < n; i += (uint64_t)gridDim.x * LEAD_THREADS) {
        Fr c = fe_mul(fe_load<FrParams>(invfact + 8 * i), fe_load<FrParams>(invfact + 8 * (uint64_t)(n - 1 - i)));
        if ((n - 1 - i) & 1) c = fe_neg(c);
        sa = fe_add(sa, fe_mul(fe_load<FrParams>(a + 8 * i), c));
        sb = fe_add(sb, fe_mul(fe_load<FrParams>(b + 8 * i), c));
    }
    fe_store<FrParams>(sh[0][threadIdx.x], sa);
    fe_store<FrParams>(sh[1][threadIdx.x], sb);
    __syncthreads();
    for (uint32_t w = LEAD_THREADS / 2; w; w >>= 1) {
        if (threadIdx.x < w) {
            fe_store<FrParams>(sh[0][threadIdx.x], fe_add(fe_load<FrParams>(sh[0][threadIdx.x]), fe_load<FrParams>(sh[0][threadIdx.x + w])));
            fe_store<FrParams>(sh[1][threadIdx.x], fe_add(fe_load<FrParams>(sh[1][threadIdx.x]), fe_load<FrParams>(sh[1][threadIdx.x + w])));
        }
        __syncthreads();
    }
    if (threadIdx.x < 2) fe_store<FrParams>(part + 8 * (2 * (uint64_t)blockIdx.x + threadIdx.x), fe_load<FrParams>(sh[threadIdx.x][0]));
}
__global__ void k_lead_final(uint32_t* __restrict__ out, const uint32_t* __restrict__ part, uint32_t nblocks) {
    if (threadIdx.x >= 2) return;
    Fr s = fe_zero<FrParams>();
    for (uint32_t k = 0; k < nblocks; k++) s = fe_add(s, fe_load<FrParams>(part + 8 * (2 * (uint64_t)k + threadIdx.x)));
    fe_store<FrParams>(out + 8 * threadIdx.x, s);
}
int frstage_leading_coeffs(const FrStage& f, const FrScratch& sc, void* d_kappa, void* d_partials, hipStream_t s) {
    const uint32_t n = f.n;
    uint32_t blocks = (n + LEAD_THREADS - 1) / LEAD_THREADS;
    if (blocks > LEAD_BLOCKS) blocks = LEAD_BLOCKS;
    const uint32_t* a = FRP(sc.abc);
    if (n == 1) {          // one gate: the interpolants are the constants a_0, b_0 (frstage_init builds no table for it)
        HIPCHK(hipMemcpyAsync(d_kappa, a, 32, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync((uint8_t*)d_kappa + 32, a + 8, 32, hipMemcpyDeviceToDevice, s));
        return ZK_OK;
    }
    hipLaunchKernelGGL(k_lead_partial, dim3(blocks), dim3(LEAD_THREADS), 0, s, (uint32_t*)d_partials, a, a + 8 * (uint64_t)n, (const uint32_t*)FRP(f.invfact), n);
    hipLaunchKernelGGL(k_lead_final, dim3(1), dim3(64), 0, s, (uint32_t*)d_kappa, (const uint32_t*)d_partials, blocks);
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
__global__ void k_shifted_powers(uint32_t* __restrict__ out, uint32_t n, uint32_t e, uint32_t cnt) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= cnt) return;
    Fr base = fe_from_u32<FrParams>(n + t), r = fe_one<FrParams>();
    for (uint32_t k = e; k; k >>= 1) {
        if (k & 1) r = fe_mul(r, base);
        base = fe_mul(base, base);
    }
    fe_store<FrParams>(out + 8 * (uint64_t)t, r);
}
int frstage_shifted_powers(void* d_out, uint32_t n, uint32_t e, uint32_t cnt, hipStream_t s) {
    if (!cnt) return ZK_OK;
    hipLaunchKernelGGL(k_shifted_powers, g1d(cnt), dim3(256), 0, s, (uint32_t*)d_out, n, e, cnt);
    HIPCHK(hipGetLastError());
    return ZK_OK;
}

}  // namespace zk

using namespace zk;

// y = M x over Fr for one sparse matrix: the R1CS product at the heart of QAP.eval (rows = gates: the values at X = g of sum_k sol_k poly_k,
// QAP.ml:121-131) and, with M TRANSPOSED, of keygen (u_k(tau) = sum_g M[g][k] l_g(tau): groth16.ml:59-68, pinocchio.ml:104-109 through the Lagrange basis).
extern "C" int zk_fr_spmv(uint32_t rows, uint32_t cols, const zk_csr* M, const uint8_t* x, uint8_t* y) {
    if (!M || !x || !y || rows == 0 || cols == 0) ZK_FAIL(ZK_ERR_ARG, "zk_fr_spmv: null or empty argument");
    ZKCHK(ensure_init());
    Ctx& c = ctx();
    CsrDev d;
    ZKCHK(upload_csr(d, M, rows, cols, c.stream));          // validates row_ptr / col, ZK_ERR_SCALAR_RANGE for a coefficient >= r
    DevBuf dx, dy, flag;
    ZKCHK(dx.alloc(32 * (size_t)cols));
    ZKCHK(dy.alloc(32 * (size_t)rows));
    ZKCHK(flag.alloc(4));
    HIPCHK(hipMemsetAsync(flag.p, 0, 4, c.stream));
    HIPCHK(hipMemcpyAsync(dx.p, x, 32 * (size_t)cols, hipMemcpyHostToDevice, c.stream));
    ZKCHK(fr_to_mont(dx.p, dx.p, cols, flag.as<int>(), c.stream));
    hipLaunchKernelGGL(k_spmv, g1d(rows), dim3(256), 0, c.stream, (const uint32_t*)FRP(d.ptr), (const uint32_t*)FRP(d.col), (const uint32_t*)FRP(d.val), (const uint32_t*)FRP(dx), FRP(dy), rows);
    HIPCHK(hipGetLastError());
    ZKCHK(fr_from_mont(dy.p, dy.p, rows, c.stream));
    int hf = 0;
    HIPCHK(hipMemcpyAsync(&hf, flag.p, 4, hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipMemcpyAsync(y, dy.p, 32 * (size_t)rows, hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    if (hf) ZK_FAIL(ZK_ERR_SCALAR_RANGE, "zk_fr_spmv: vector element >= r");
    return ZK_OK;
}

extern "C" int zk_fr_poly_mul(const uint8_t* a, size_t na, const uint8_t* b, size_t nb, uint8_t* out, size_t* nout) {
    if (!nout) ZK_FAIL(ZK_ERR_ARG, "zk_fr_poly_mul: null nout");
    *nout = 0;
    if (na == 0 || nb == 0) return ZK_OK;                         // mul_scalar / sum of nothing = [] (polynomial.ml:117-131)
    if (!a || !b || !out) ZK_FAIL(ZK_ERR_ARG, "zk_fr_poly_mul: null buffer");
    ZKCHK(ensure_init());
    Ctx& c = ctx();
    size_t n = na + nb - 1;
    DevBuf A, B, Oo, flag;
    ZKCHK(A.alloc(32 * na));
    ZKCHK(B.alloc(32 * nb));
    ZKCHK(Oo.alloc(32 * n));
    ZKCHK(flag.alloc(4));
    HIPCHK(hipMemsetAsync(flag.p, 0, 4, c.stream));
    HIPCHK(hipMemcpyAsync(A.p, a, 32 * na, hipMemcpyHostToDevice, c.stream));
    HIPCHK(hipMemcpyAsync(B.p, b, 32 * nb, hipMemcpyHostToDevice, c.stream));
    ZKCHK(fr_to_mont(A.p, A.p, na, flag.as<int>(), c.stream));
    ZKCHK(fr_to_mont(B.p, B.p, nb, flag.as<int>(), c.stream));
    ZKCHK(dev_poly_mul(A.p, na, B.p, nb, Oo.p, c.stream));
    ZKCHK(fr_from_mont(Oo.p, Oo.p, n, c.stream));
    int hf = 0;
    HIPCHK(hipMemcpyAsync(&hf, flag.p, 4, hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipMemcpyAsync(out, Oo.p, 32 * n, hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    if (hf) ZK_FAIL(ZK_ERR_SCALAR_RANGE, "zk_fr_poly_mul: coefficient >= r");
    // Polynomial.normalize (polynomial.ml:100-107): strip trailing zeros
    while (n) {
        const uint8_t* p = out + 32 * (n - 1);
        bool z = true;
        for (int i = 0; i < 32; i++) z = z && p[i] == 0;
        if (!z) break;
        n--;
    }
    *nout = n;
    return ZK_OK;
}
