// Groth16.Make(C).prove (src/groth16/groth16.ml:123-161,235-237) on gfx950.
//
// With the coefficient vectors v, w, h of QAP.eval the reference's proof is
//   A = alpha + sum_i v_i [tau^i]_1 + r delta                                    (groth16.ml:128-134)
//   B = beta  + sum_i w_i [tau^i]_2 + s delta                                    (groth16.ml:135-141)
//   C = sum_mid w_k [L_k(tau)/delta]_1 + sum_i h_i [tau^i Z(tau)/delta]_1 + s A + r B1 - r s delta   (:151-159)
// (the reference reaches the same sums through m*n single scalar multiplications,
// groth16.ml:116-121).  Single-point scalar multiples are latency poison on a GPU (255 dependent
// doublings on one lane), so the blinding terms are folded INTO the multi-scalar products:
//   s A + r B1 - r s delta = s alpha + r beta_1 + r s delta + sum_i (s v_i + r w_i) [tau^i]_1
// and the whole proof is three MSMs over the key exactly as it was uploaded:
//   G1 pool = a | d1 | b1 | ti1[n+2] | tiztd[n-1] | ltd_mid[n_mid],   G2 pool = b2 | d2 | ti2[n+2]
//   A: scalars (1, r, 0, v, 0, 0)          B: scalars (1, s, w)          C: scalars (s, rs, r, s v + r w, h, w|mid)
// The G2 product runs on a second HIP stream beside the two G1 products.
#include "ec.cuh"
#include "frstage.cuh"
#include "msm.cuh"

#include <map>
#include <memory>
#include <string.h>

namespace zk {

struct Groth16Key {
    uint32_t n = 0, m = 0, n_mid = 0;
    uint32_t rank = 0, world = 1;
    uint64_t p1 = 0, p2 = 0;            // full pool sizes (points)
    uint64_t lo1 = 0, hi1 = 0;          // this rank's slice of the G1 pool
    uint64_t lo2 = 0, hi2 = 0;
    FrStage fr;
    MsmBases g1, g2;
    MsmWorkspace ws1, ws2;
    DevBuf mid_idx;                     // variable index of the j-th mid variable
    DevBuf scalA, scalC, scalB;         // canonical scalars, full pool length
    DevBuf wit_raw, rs, results;        // results: A, C (G1 XYZZ) then B (G2 XYZZ)
    bool have_witness = false;
};

static std::map<uint64_t, std::unique_ptr<Groth16Key>>& g_keys = *new std::map<uint64_t, std::unique_ptr<Groth16Key>>;   // never destroyed (see ntt.hip)
static uint64_t g_next_handle = 1;
static void handles_release() { g_keys.clear(); }
static CleanupRegistrar g_key_cleanup(handles_release);

static inline dim3 g1d(uint64_t n, unsigned t = 256) { return dim3((unsigned)((n + t - 1) / t)); }

// scalar vectors for the three MSMs (canonical form).  rs = r | s (canonical, device).
__global__ void k_groth16_scalars(uint32_t* __restrict__ scalA, uint32_t* __restrict__ scalC, uint32_t* __restrict__ scalB,
                                  const uint32_t* __restrict__ v, const uint32_t* __restrict__ w, const uint32_t* __restrict__ h,
                                  const uint32_t* __restrict__ wit_mont, const uint32_t* __restrict__ mid_idx,
                                  const uint32_t* __restrict__ rs, uint32_t n, uint32_t n_mid, int with_blinding) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t p1 = 3 + (uint64_t)(n + 2) + (n - 1) + n_mid;
    if (i >= p1) return;
    Fr r = fe_zero<FrParams>(), s = r;
    if (with_blinding) { r = fe_to_mont(fe_load<FrParams>(rs)); s = fe_to_mont(fe_load<FrParams>(rs + 8)); }
    const Fr zero = fe_zero<FrParams>(), one = fe_one<FrParams>();
    Fr a = zero, c = zero;
    const uint64_t ti = 3, tz = ti + n + 2, lt = tz + (n - 1);
    if (i == 0) { a = with_blinding ? one : zero; c = s; }                       // alpha
    else if (i == 1) { a = r; c = fe_mul(r, s); }                                 // delta
    else if (i == 2) { c = r; }                                                   // beta_1
    else if (i < tz) {
        uint64_t k = i - ti;
        if (k < n) {
            Fr vk = fe_load<FrParams>(v + 8 * k), wk = fe_load<FrParams>(w + 8 * k);
            a = vk;
            c = fe_add(fe_mul(s, vk), fe_mul(r, wk));
        }
    } else if (i < lt) c = fe_load<FrParams>(h + 8 * (i - tz));
    else c = fe_load<FrParams>(wit_mont + 8 * (uint64_t)mid_idx[i - lt]);
    fe_store<FrParams>(scalA + 8 * i, fe_from_mont(a));
    fe_store<FrParams>(scalC + 8 * i, fe_from_mont(c));
    const uint64_t p2 = 2 + (uint64_t)(n + 2);
    if (i < p2) {
        Fr b = zero;
        if (i == 0) b = with_blinding ? one : zero;                               // beta_2
        else if (i == 1) b = s;                                                   // delta_2
        else if (i - 2 < n) b = fe_load<FrParams>(w + 8 * (i - 2));
        fe_store<FrParams>(scalB + 8 * i, fe_from_mont(b));
    }
}

static int key_lookup(uint64_t handle, Groth16Key** out) {
    auto it = g_keys.find(handle);
    if (it == g_keys.end()) ZK_FAIL(ZK_ERR_HANDLE, "unknown Groth16 key handle");
    *out = it->second.get();
    return ZK_OK;
}

static int upload(uint32_t n, uint32_t m, const zk_csr* L, const zk_csr* R, const zk_csr* O, const uint8_t* mid,
                  const uint8_t* pk_g1, size_t pk_g1_points, const uint8_t* pk_g2, size_t pk_g2_points, uint32_t rank,
                  uint32_t world, uint64_t* handle) {
    if (!handle || !mid || !pk_g1 || !pk_g2) ZK_FAIL(ZK_ERR_ARG, "pk_upload: null argument");
    if (world == 0 || rank >= world) ZK_FAIL(ZK_ERR_ARG, "pk_upload: bad rank / world");
    ZKCHK(ensure_init());
    Ctx& c = ctx();
    auto key = std::make_unique<Groth16Key>();
    Groth16Key& k = *key;
    k.n = n; k.m = m; k.rank = rank; k.world = world;
    std::vector<uint32_t> mids;
    for (uint32_t i = 0; i < m; i++)
        if (mid[i]) mids.push_back(i);
    k.n_mid = (uint32_t)mids.size();
    if (n < 2) ZK_FAIL(ZK_ERR_ARG, "pk_upload: need at least 2 constraints");
    k.p1 = 3 + (uint64_t)(n + 2) + (n - 1) + k.n_mid;
    k.p2 = 2 + (uint64_t)(n + 2);
    if (pk_g1_points != k.p1) ZK_FAIL(ZK_ERR_DOMAIN, "pk_upload: G1 key length != 3 + (n+2) + (n-1) + |mids|");
    if (pk_g2_points != k.p2) ZK_FAIL(ZK_ERR_DOMAIN, "pk_upload: G2 key length != 2 + (n+2)");
    ZKCHK(frstage_init(k.fr, n, m, L, R, O, c.stream));
    // contiguous pool slices per rank
    k.lo1 = k.p1 * rank / world; k.hi1 = k.p1 * (rank + 1) / world;
    k.lo2 = k.p2 * rank / world; k.hi2 = k.p2 * (rank + 1) / world;
    if (k.hi1 == k.lo1 || k.hi2 == k.lo2) ZK_FAIL(ZK_ERR_ARG, "pk_upload: more ranks than key points");
    ZKCHK(msm_bases_from_bytes(k.g1, CURVE_G1, pk_g1 + 96 * k.lo1, k.hi1 - k.lo1, 0, true, c.stream));
    ZKCHK(msm_bases_from_bytes(k.g2, CURVE_G2, pk_g2 + 192 * k.lo2, k.hi2 - k.lo2, 0, true, c.stream));
    ZKCHK(msm_workspace_alloc(k.ws1, k.g1));
    ZKCHK(msm_workspace_alloc(k.ws2, k.g2));
    ZKCHK(k.mid_idx.alloc(4 * (size_t)(k.n_mid ? k.n_mid : 1)));
    if (k.n_mid) HIPCHK(hipMemcpyAsync(k.mid_idx.p, mids.data(), 4 * (size_t)k.n_mid, hipMemcpyHostToDevice, c.stream));
    ZKCHK(k.scalA.alloc(32 * k.p1));
    ZKCHK(k.scalC.alloc(32 * k.p1));
    ZKCHK(k.scalB.alloc(32 * k.p2));
    ZKCHK(k.wit_raw.alloc(32 * (size_t)m));
    ZKCHK(k.rs.alloc(64));
    ZKCHK(k.results.alloc(2 * xyzz_bytes(CURVE_G1) + xyzz_bytes(CURVE_G2)));
    HIPCHK(hipStreamSynchronize(c.stream));
    *handle = g_next_handle++;
    g_keys[*handle] = std::move(key);
    return ZK_OK;
}

// Fr stage + the three MSMs over this rank's slice; results left in k.results (A, C, B as XYZZ)
static int prove_core(Groth16Key& k, const uint8_t* sol, const uint8_t* r, const uint8_t* s, int with_blinding) {
    Ctx& c = ctx();
    if (sol) { HIPCHK(hipMemcpyAsync(k.wit_raw.p, sol, 32 * (size_t)k.m, hipMemcpyHostToDevice, c.stream)); k.have_witness = true; }
    else if (!k.have_witness) ZK_FAIL(ZK_ERR_ARG, "no witness: pass sol or call zk_groth16_set_witness first");
    if (with_blinding) {
        HIPCHK(hipMemcpyAsync(k.rs.p, r, 32, hipMemcpyHostToDevice, c.stream));
        HIPCHK(hipMemcpyAsync((char*)k.rs.p + 32, s, 32, hipMemcpyHostToDevice, c.stream));
    }
    ZKCHK(frstage_eval(k.fr, k.wit_raw.p, c.stream));
    const uint32_t* v = k.fr.d.as<uint32_t>();
    const uint32_t* w = v + 8 * (uint64_t)k.fr.n2;
    hipLaunchKernelGGL(k_groth16_scalars, g1d(k.p1), dim3(256), 0, c.stream, k.scalA.as<uint32_t>(), k.scalC.as<uint32_t>(), k.scalB.as<uint32_t>(), v, w,
                       (const uint32_t*)k.fr.h.as<uint32_t>(), (const uint32_t*)k.fr.wit.as<uint32_t>(), (const uint32_t*)k.mid_idx.as<uint32_t>(),
                       (const uint32_t*)k.rs.as<uint32_t>(), k.n, k.n_mid, with_blinding);
    HIPCHK(hipGetLastError());
    char* res = k.results.as<char>();
    // fork: G2 on stream2
    HIPCHK(hipEventRecord(c.ev_fork, c.stream));
    HIPCHK(hipStreamWaitEvent(c.stream2, c.ev_fork, 0));
    ZKCHK(msm_run(k.g2, k.ws2, k.scalB.as<char>() + 32 * k.lo2, res + 2 * xyzz_bytes(CURVE_G1), c.stream2));
    HIPCHK(hipEventRecord(c.ev_join, c.stream2));
    ZKCHK(msm_run(k.g1, k.ws1, k.scalA.as<char>() + 32 * k.lo1, res, c.stream));
    ZKCHK(msm_run(k.g1, k.ws1, k.scalC.as<char>() + 32 * k.lo1, res + xyzz_bytes(CURVE_G1), c.stream));
    HIPCHK(hipStreamWaitEvent(c.stream, c.ev_join, 0));
    return ZK_OK;
}
static int check_flag(Groth16Key& k) {
    Ctx& c = ctx();
    int hf = 0;
    HIPCHK(hipMemcpyAsync(&hf, k.fr.flag.p, 4, hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    if (hf & 2) ZK_FAIL(ZK_ERR_SCALAR_RANGE, "witness value >= r");
    if (hf & 1) ZK_FAIL(ZK_ERR_REMAINDER, "p mod Z != 0");
    return ZK_OK;
}
// proof bytes: a | b | c  from results laid out A, C, B
static int emit_proof(const void* d_results, uint8_t* proof) {
    Ctx& c = ctx();
    uint8_t g1pts[192];
    ZKCHK(points_xyzz_to_bytes(CURVE_G1, d_results, 2, g1pts, c.stream));
    ZKCHK(points_xyzz_to_bytes(CURVE_G2, (const char*)d_results + 2 * xyzz_bytes(CURVE_G1), 1, proof + 96, c.stream));
    memcpy(proof, g1pts, 96);
    memcpy(proof + 288, g1pts + 96, 96);
    return ZK_OK;
}

}  // namespace zk

using namespace zk;
extern "C" {

int zk_groth16_pk_upload(uint32_t n, uint32_t m, const zk_csr* L, const zk_csr* R, const zk_csr* O, const uint8_t* mid,
                         const uint8_t* pk_g1, size_t pk_g1_points, const uint8_t* pk_g2, size_t pk_g2_points, uint64_t* handle) {
    return upload(n, m, L, R, O, mid, pk_g1, pk_g1_points, pk_g2, pk_g2_points, 0, 1, handle);
}
int zk_groth16_pk_upload_sharded(uint32_t n, uint32_t m, const zk_csr* L, const zk_csr* R, const zk_csr* O, const uint8_t* mid,
                                 const uint8_t* pk_g1, size_t pk_g1_points, const uint8_t* pk_g2, size_t pk_g2_points,
                                 uint32_t rank, uint32_t world, uint64_t* handle) {
    return upload(n, m, L, R, O, mid, pk_g1, pk_g1_points, pk_g2, pk_g2_points, rank, world, handle);
}
int zk_groth16_pk_free(uint64_t handle) {
    auto it = g_keys.find(handle);
    if (it == g_keys.end()) ZK_FAIL(ZK_ERR_HANDLE, "unknown Groth16 key handle");
    (void)zk_sync();
    g_keys.erase(it);
    return ZK_OK;
}
int zk_groth16_prove(uint64_t handle, const uint8_t* sol, const uint8_t r[32], const uint8_t s[32], uint8_t proof[384]) {
    Groth16Key* k;
    ZKCHK(key_lookup(handle, &k));
    if (!r || !s || !proof) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_prove: null argument");
    if (k->world != 1) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_prove: key is sharded; use prove_partial + combine");
    ZKCHK(prove_core(*k, sol, r, s, 1));
    ZKCHK(check_flag(*k));
    return emit_proof(k->results.p, proof);
}
int zk_groth16_set_witness(uint64_t handle, const uint8_t* sol) {
    Groth16Key* k;
    ZKCHK(key_lookup(handle, &k));
    if (!sol) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_set_witness: null");
    Ctx& c = ctx();
    HIPCHK(hipMemcpyAsync(k->wit_raw.p, sol, 32 * (size_t)k->m, hipMemcpyHostToDevice, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    k->have_witness = true;
    return ZK_OK;
}
int zk_groth16_prove_partial(uint64_t handle, const uint8_t* sol, const uint8_t r[32], const uint8_t s[32],
                             uint8_t partial[ZK_GROTH16_PARTIAL_BYTES]) {
    Groth16Key* k;
    ZKCHK(key_lookup(handle, &k));
    if (!r || !s || !partial) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_prove_partial: null argument");
    ZKCHK(prove_core(*k, sol, r, s, 1));
    ZKCHK(check_flag(*k));
    Ctx& c = ctx();
    HIPCHK(hipMemcpyAsync(partial, k->results.p, ZK_GROTH16_PARTIAL_BYTES, hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    return ZK_OK;
}
int zk_groth16_combine(const uint8_t* partials, uint32_t world, uint8_t proof[384]) {
    if (!partials || !proof || world == 0) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_combine: bad argument");
    ZKCHK(ensure_init());
    Ctx& c = ctx();
    const size_t g1b = xyzz_bytes(CURVE_G1), g2b = xyzz_bytes(CURVE_G2), blk = 2 * g1b + g2b;
    DevBuf parts, g1p, g2p, sum;
    ZKCHK(parts.alloc(blk * world));
    ZKCHK(g1p.alloc(2 * g1b * world));
    ZKCHK(g2p.alloc(g2b * world));
    ZKCHK(sum.alloc(blk));
    HIPCHK(hipMemcpyAsync(parts.p, partials, blk * world, hipMemcpyHostToDevice, c.stream));
    for (uint32_t j = 0; j < world; j++) {     // rank-major blocks -> [rank][A, C] and [rank][B]
        HIPCHK(hipMemcpyAsync(g1p.as<char>() + 2 * g1b * j, parts.as<char>() + blk * j, 2 * g1b, hipMemcpyDeviceToDevice, c.stream));
        HIPCHK(hipMemcpyAsync(g2p.as<char>() + g2b * j, parts.as<char>() + blk * j + 2 * g1b, g2b, hipMemcpyDeviceToDevice, c.stream));
    }
    ZKCHK(xyzz_sum_columns(CURVE_G1, sum.p, g1p.p, world, 2, c.stream));
    ZKCHK(xyzz_sum_columns(CURVE_G2, sum.as<char>() + 2 * g1b, g2p.p, world, 1, c.stream));
    return emit_proof(sum.p, proof);
}
int zk_groth16_qap_eval(uint64_t handle, const uint8_t* sol, uint8_t* v_out, uint8_t* w_out, uint8_t* h_out) {
    Groth16Key* k;
    ZKCHK(key_lookup(handle, &k));
    Ctx& c = ctx();
    if (sol) { HIPCHK(hipMemcpyAsync(k->wit_raw.p, sol, 32 * (size_t)k->m, hipMemcpyHostToDevice, c.stream)); k->have_witness = true; }
    else if (!k->have_witness) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_qap_eval: no witness");
    ZKCHK(frstage_eval(k->fr, k->wit_raw.p, c.stream));
    ZKCHK(check_flag(*k));
    DevBuf tmp;
    ZKCHK(tmp.alloc(32 * (size_t)k->n));
    const char* v = k->fr.d.as<char>();
    const void* srcs[3] = {v, v + 32 * (size_t)k->fr.n2, k->fr.h.p};
    uint8_t* outs[3] = {v_out, w_out, h_out};
    size_t cnt[3] = {k->n, k->n, (size_t)k->n - 1};
    for (int i = 0; i < 3; i++) {
        if (!outs[i]) continue;
        ZKCHK(fr_from_mont(tmp.p, srcs[i], cnt[i], c.stream));
        HIPCHK(hipMemcpyAsync(outs[i], tmp.p, 32 * cnt[i], hipMemcpyDeviceToHost, c.stream));
        HIPCHK(hipStreamSynchronize(c.stream));
    }
    return ZK_OK;
}
}
