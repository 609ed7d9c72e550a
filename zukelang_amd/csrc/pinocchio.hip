// Pinocchio Protocol 2 prove on gfx950: Compute.f / ZKCompute.f (src/pinocchio/pinocchio.ml:210-248,
// 427-514) on the kernels of the Groth16 path (second protocol of the scope table, row a9).
//
// With c = the witness, c_mid its restriction to the mids, h = QAP.eval's quotient and the blinding
// scalars dv, dw, dy (drawn in that order, :428-430) the eight proof elements are multi-scalar
// products over the evaluation key (:37-60); every blinding term is folded into the product whose base
// it multiplies, and  - one * dy  rides on si[0] = g^(s^0) = G1.one:
//   vv'   = <vv   | vt,              c_mid | dv>                      (:438-439)
//   ww'   = <ww   | wt,              c_mid | dw>            in G2     (:442-443)
//   yy'   = <yy   | yt,              c_mid | dy>                      (:446-447)
//   h'    = <si | v_all | w_all,     h + dv dw Z - dy e_0 | dw c | dv c>   (:450,481-486; t = apply_powers Z si, :431)
//   vavv' = <vav  | vavt,            c_mid | dv>                      (:489-490)
//   waww' = <waw  | wawt,            c_mid | dw>            in G2     (:493-494)
//   yayy' = <yay  | yayt,            c_mid | dy>                      (:497-498)
//   bvwy' = <bvwy | vbt | wbt | ybt, c_mid | dv | dw | dy>            (:500-505)
// NonZK.prove (Compute.f) is the same with dv = dw = dy = 0.
#include "ec.cuh"
#include "frstage.cuh"
#include "msm.cuh"

#include <map>
#include <memory>
#include <string.h>

namespace zk {

static constexpr int PIN_G1 = 6, PIN_G2 = 2;

struct PinKey {
    uint32_t n = 0, m = 0, n_mid = 0;
    FrStage fr;
    FrScratch fs;
    MsmBases g1[PIN_G1], g2[PIN_G2];
    MsmWorkspace ws1[PIN_G1], ws2[PIN_G2];
    DevBuf scal1[PIN_G1], scal2[PIN_G2];
    DevBuf mid_idx, wit_raw, deltas, results, out_dev, flag_dev;
    hipStream_t st[3] = {nullptr, nullptr, nullptr};
    hipEvent_t fork = nullptr, join[2] = {nullptr, nullptr};
    ~PinKey() {
        for (auto s : st) if (s) (void)hipStreamDestroy(s);
        if (fork) { (void)hipEventDestroy(fork); (void)hipEventDestroy(join[0]); (void)hipEventDestroy(join[1]); }
    }
};
static std::map<uint64_t, std::unique_ptr<PinKey>>& g_pin = *new std::map<uint64_t, std::unique_ptr<PinKey>>;   // never destroyed (see ntt.hip)
static uint64_t g_pin_next = 0x5000000001ull;
static void pin_release() { g_pin.clear(); }
static CleanupRegistrar g_pin_cleanup(pin_release);

static inline dim3 g1d(uint64_t n, unsigned t = 256) { return dim3((unsigned)((n + t - 1) / t)); }

struct PinScalPtrs {
    uint32_t* s1[PIN_G1];
    uint32_t* s2[PIN_G2];
};
// one lane per entry of the longest vector (the h pool: n + 1 + 2 m)
__global__ void k_pinocchio_scalars(PinScalPtrs out, const uint32_t* __restrict__ h, const uint32_t* __restrict__ z,
                                    const uint32_t* __restrict__ wit_mont, const uint32_t* __restrict__ mid_idx,
                                    const uint32_t* __restrict__ deltas, uint32_t n, uint32_t m, uint32_t n_mid) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t ph = (uint64_t)n + 1 + 2 * (uint64_t)m;
    if (i >= ph) return;
    const Fr dv = fe_to_mont(fe_load<FrParams>(deltas)), dw = fe_to_mont(fe_load<FrParams>(deltas + 8)),
             dy = fe_to_mont(fe_load<FrParams>(deltas + 16));
    // pool 5: h'
    Fr x;
    if (i <= n) {
        x = fe_mul(fe_mul(dv, dw), fe_load<FrParams>(z + 8 * i));            // dv dw Z_i
        if (i + 1 < n) x = fe_add(x, fe_load<FrParams>(h + 8 * i));          // h has n - 1 coefficients
        if (i == 0) x = fe_sub(x, dy);
    } else if (i <= (uint64_t)n + m) x = fe_mul(dw, fe_load<FrParams>(wit_mont + 8 * (i - n - 1)));
    else x = fe_mul(dv, fe_load<FrParams>(wit_mont + 8 * (i - n - 1 - m)));
    fe_store<FrParams>(out.s1[5] + 8 * i, fe_from_mont(x));
    // pools over the mids
    if (i < (uint64_t)n_mid + 3) {
        Fr c = fe_zero<FrParams>();
        const bool is_mid = i < n_mid;
        if (is_mid) c = fe_load<FrParams>(wit_mont + 8 * (uint64_t)mid_idx[i]);
        const uint64_t e = i - n_mid;      // index among the appended single points
        auto put = [&](uint32_t* dst, uint32_t extras, const Fr& e0, const Fr& e1, const Fr& e2) {
            if (is_mid) fe_store<FrParams>(dst + 8 * i, fe_from_mont(c));
            else if (e < extras) fe_store<FrParams>(dst + 8 * i, fe_from_mont(e == 0 ? e0 : (e == 1 ? e1 : e2)));
        };
        put(out.s1[0], 1, dv, dv, dv);       // vv | vt
        put(out.s1[1], 1, dy, dy, dy);       // yy | yt
        put(out.s1[2], 1, dv, dv, dv);       // vav | vavt
        put(out.s1[3], 1, dy, dy, dy);       // yay | yayt
        put(out.s1[4], 3, dv, dw, dy);       // bvwy | vbt | wbt | ybt
        put(out.s2[0], 1, dw, dw, dw);       // ww | wt
        put(out.s2[1], 1, dw, dw, dw);       // waw | wawt
    }
}

static int pin_lookup(uint64_t handle, PinKey** out) {
    auto it = g_pin.find(handle);
    if (it == g_pin.end()) ZK_FAIL(ZK_ERR_HANDLE, "unknown Pinocchio key handle");
    *out = it->second.get();
    return ZK_OK;
}

}  // namespace zk

using namespace zk;
extern "C" {

int zk_pinocchio_pk_upload(uint32_t n, uint32_t m, const zk_csr* L, const zk_csr* R, const zk_csr* O, const uint8_t* mid,
                           const uint8_t* pk_g1, size_t pk_g1_points, const uint8_t* pk_g2, size_t pk_g2_points, uint64_t* handle) {
    if (!handle || !mid || !pk_g1 || !pk_g2) ZK_FAIL(ZK_ERR_ARG, "pinocchio pk_upload: null argument");
    ZKCHK(ensure_init());
    Ctx& c = ctx();
    auto key = std::make_unique<PinKey>();
    PinKey& k = *key;
    k.n = n; k.m = m;
    std::vector<uint32_t> mids;
    for (uint32_t i = 0; i < m; i++)
        if (mid[i]) mids.push_back(i);
    const uint64_t nm = k.n_mid = (uint32_t)mids.size();
    if (n < 2 || nm == 0) ZK_FAIL(ZK_ERR_ARG, "pinocchio pk_upload: need >= 2 constraints and a non-empty I_mid");
    if (pk_g1_points != 5 * nm + (n + 1) + 2 * (uint64_t)m + 7) ZK_FAIL(ZK_ERR_DOMAIN, "pinocchio pk_upload: G1 key length");
    if (pk_g2_points != 2 * nm + (n + 1) + 2) ZK_FAIL(ZK_ERR_DOMAIN, "pinocchio pk_upload: G2 key length");
    ZKCHK(frstage_init(k.fr, n, m, L, R, O, c.stream));
    ZKCHK(frstage_scratch_alloc(k.fr, k.fs));
    // slices of the flattened key (pinocchio.ml:37-60; layout in include/zkmi355x.h)
    const uint8_t *VV = pk_g1, *YY = VV + 96 * nm, *VAV = YY + 96 * nm, *YAY = VAV + 96 * nm, *BV = YAY + 96 * nm,
                  *SI = BV + 96 * nm, *VALL = SI + 96 * (uint64_t)(n + 1), *WALL = VALL + 96 * (uint64_t)m, *ONES = WALL + 96 * (uint64_t)m;
    const uint8_t *WW = pk_g2, *WAW = WW + 192 * nm, *ONES2 = WAW + 192 * nm + 192 * (uint64_t)(n + 1);
    std::vector<uint8_t> buf;
    auto pool1 = [&](int idx, const uint8_t* base, uint64_t cnt, std::initializer_list<const uint8_t*> extras) -> int {
        buf.assign(base, base + 96 * cnt);
        for (auto e : extras) buf.insert(buf.end(), e, e + 96);
        return msm_bases_from_bytes(k.g1[idx], CURVE_G1, buf.data(), buf.size() / 96, 0, true, c.stream);
    };
    ZKCHK(pool1(0, VV, nm, {ONES + 96 * 0}));
    ZKCHK(pool1(1, YY, nm, {ONES + 96 * 1}));
    ZKCHK(pool1(2, VAV, nm, {ONES + 96 * 2}));
    ZKCHK(pool1(3, YAY, nm, {ONES + 96 * 3}));
    ZKCHK(pool1(4, BV, nm, {ONES + 96 * 4, ONES + 96 * 5, ONES + 96 * 6}));
    ZKCHK(pool1(5, SI, (uint64_t)n + 1 + 2 * (uint64_t)m, {}));          // si | v_all | w_all are contiguous in the key
    auto pool2 = [&](int idx, const uint8_t* base, const uint8_t* extra) -> int {
        buf.assign(base, base + 192 * nm);
        buf.insert(buf.end(), extra, extra + 192);
        return msm_bases_from_bytes(k.g2[idx], CURVE_G2, buf.data(), nm + 1, 0, true, c.stream);
    };
    ZKCHK(pool2(0, WW, ONES2));
    ZKCHK(pool2(1, WAW, ONES2 + 192));
    for (int i = 0; i < PIN_G1; i++) { ZKCHK(msm_workspace_alloc(k.ws1[i], k.g1[i])); ZKCHK(k.scal1[i].alloc(32 * k.g1[i].n)); }
    for (int i = 0; i < PIN_G2; i++) { ZKCHK(msm_workspace_alloc(k.ws2[i], k.g2[i])); ZKCHK(k.scal2[i].alloc(32 * k.g2[i].n)); }
    ZKCHK(k.mid_idx.alloc(4 * nm));
    HIPCHK(hipMemcpyAsync(k.mid_idx.p, mids.data(), 4 * nm, hipMemcpyHostToDevice, c.stream));
    ZKCHK(k.wit_raw.alloc(32 * (size_t)m));
    ZKCHK(k.deltas.alloc(96));
    ZKCHK(k.results.alloc(PIN_G1 * xyzz_bytes(CURVE_G1) + PIN_G2 * xyzz_bytes(CURVE_G2)));
    ZKCHK(k.out_dev.alloc(960));
    for (auto& s : k.st) HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&k.fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&k.join[0], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&k.join[1], hipEventDisableTiming));
    HIPCHK(hipStreamSynchronize(c.stream));
    *handle = g_pin_next++;
    g_pin[*handle] = std::move(key);
    return ZK_OK;
}
int zk_pinocchio_pk_free(uint64_t handle) {
    auto it = g_pin.find(handle);
    if (it == g_pin.end()) ZK_FAIL(ZK_ERR_HANDLE, "unknown Pinocchio key handle");
    (void)hipDeviceSynchronize();
    g_pin.erase(it);
    return ZK_OK;
}
int zk_pinocchio_prove(uint64_t handle, const uint8_t* sol, const uint8_t dv[32], const uint8_t dw[32], const uint8_t dy[32],
                       uint8_t proof[960]) {
    PinKey* kp;
    ZKCHK(pin_lookup(handle, &kp));
    PinKey& k = *kp;
    if (!sol || !dv || !dw || !dy || !proof) ZK_FAIL(ZK_ERR_ARG, "zk_pinocchio_prove: null argument");
    hipStream_t s0 = k.st[0];
    uint8_t d3[96];
    memcpy(d3, dv, 32); memcpy(d3 + 32, dw, 32); memcpy(d3 + 64, dy, 32);
    HIPCHK(hipMemcpyAsync(k.wit_raw.p, sol, 32 * (size_t)k.m, hipMemcpyHostToDevice, s0));
    HIPCHK(hipMemcpyAsync(k.deltas.p, d3, 96, hipMemcpyHostToDevice, s0));
    HIPCHK(hipStreamSynchronize(s0));                 // d3 is a stack buffer
    ZKCHK(frstage_eval(k.fr, k.fs, k.wit_raw.p, s0));
    PinScalPtrs ptrs;
    for (int i = 0; i < PIN_G1; i++) ptrs.s1[i] = k.scal1[i].as<uint32_t>();
    for (int i = 0; i < PIN_G2; i++) ptrs.s2[i] = k.scal2[i].as<uint32_t>();
    const uint64_t ph = (uint64_t)k.n + 1 + 2 * (uint64_t)k.m;
    hipLaunchKernelGGL(k_pinocchio_scalars, g1d(ph), dim3(256), 0, s0, ptrs, (const uint32_t*)k.fs.h.as<uint32_t>(),
                       (const uint32_t*)k.fr.z.as<uint32_t>(), (const uint32_t*)k.fs.wit.as<uint32_t>(),
                       (const uint32_t*)k.mid_idx.as<uint32_t>(), (const uint32_t*)k.deltas.as<uint32_t>(), k.n, k.m, k.n_mid);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(k.fork, s0));
    HIPCHK(hipStreamWaitEvent(k.st[1], k.fork, 0));
    HIPCHK(hipStreamWaitEvent(k.st[2], k.fork, 0));
    char* res = k.results.as<char>();
    char* out = k.out_dev.as<char>();
    const size_t x1 = xyzz_bytes(CURVE_G1), x2 = xyzz_bytes(CURVE_G2);
    // proof byte offsets: vv 0 | ww 96 | yy 288 | h 384 | vavv 480 | waww 576 | yayy 768 | bvwy 864
    const size_t off1[PIN_G1] = {0, 288, 480, 768, 864, 384};
    const size_t off2[PIN_G2] = {96, 576};
    // G2 products on stream 1, the big h' product on stream 0, the five mid-sized G1 products on stream 2
    for (int i = 0; i < PIN_G2; i++) {
        ZKCHK(msm_run(k.g2[i], k.ws2[i], k.scal2[i].p, res + PIN_G1 * x1 + i * x2, k.st[1]));
        ZKCHK(points_xyzz_to_bytes_dev(CURVE_G2, res + PIN_G1 * x1 + i * x2, 1, out + off2[i], k.st[1]));
    }
    HIPCHK(hipEventRecord(k.join[0], k.st[1]));
    ZKCHK(msm_run(k.g1[5], k.ws1[5], k.scal1[5].p, res + 5 * x1, s0));
    ZKCHK(points_xyzz_to_bytes_dev(CURVE_G1, res + 5 * x1, 1, out + off1[5], s0));
    for (int i = 0; i < 5; i++) {
        ZKCHK(msm_run(k.g1[i], k.ws1[i], k.scal1[i].p, res + i * x1, k.st[2]));
        ZKCHK(points_xyzz_to_bytes_dev(CURVE_G1, res + i * x1, 1, out + off1[i], k.st[2]));
    }
    HIPCHK(hipEventRecord(k.join[1], k.st[2]));
    HIPCHK(hipStreamWaitEvent(s0, k.join[0], 0));
    HIPCHK(hipStreamWaitEvent(s0, k.join[1], 0));
    int hf = 0;
    HIPCHK(hipMemcpyAsync(proof, k.out_dev.p, 960, hipMemcpyDeviceToHost, s0));
    HIPCHK(hipMemcpyAsync(&hf, k.fs.flag.p, 4, hipMemcpyDeviceToHost, s0));
    HIPCHK(hipStreamSynchronize(s0));
    if (hf & 2) ZK_FAIL(ZK_ERR_SCALAR_RANGE, "witness value >= r");
    if (hf & 1) ZK_FAIL(ZK_ERR_REMAINDER, "p mod Z != 0");
    return ZK_OK;
}
}
