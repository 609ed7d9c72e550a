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
//
// The COMPACT h pool (round 5; default, ZK_PIN_COMPACT_H=0 / a key that fails the check below keeps the layout above).  v_all[k] = [v_k(s)] and
// w_all[k] = [w_k(s)] are 2 m of the h product's n + 1 + 2 m points (2 n + 4 of 3 n + 5 in the iterated-cubic benchmark) and serve the blinding only:
//   sum_k (dw c_k) [v_k(s)] = dw [v(s)],   v = sum_k c_k v_k  -- the very polynomial QAP.eval builds,
// so with the coefficients of v, w the prover already holds (the tau-power Fr stage) the two terms ride on si:
//   h' = <si, h + dv dw Z - dy e_0 + dw v + dv w>                                    n + 1 points,
// and with the derived key, where v, w exist as VALUES (v(n + t) for t < n - 1 from the extrapolation, a_i on 0 .. n-1),
//   v = kappa_v X^(n-1) + (a polynomial of degree <= n - 2),   kappa_v = sum_i a_i c_i  (leading coefficient of the interpolant),
//   h' = <[lambda_t(s)] | [Z(s)] | [1] | [s^(n-1)],  h(n+t) + dw (v(n+t) - kappa_v (n+t)^(n-1)) + dv (w(n+t) - kappa_w (n+t)^(n-1)) | dv dw | -dy | dw kappa_v + dv kappa_w>
// n + 2 points.  Same group element, same bytes -- PROVIDED the key's v_all / w_all are what KeyGen.generate makes them (pinocchio.ml:104-109,
// 140-147): that is checked at upload on the key's own points (pin_compact_check: <v_all, rho> = <si, coefficients of sum_k rho_k v_k> for a
// pseudo-random rho, likewise w_all); a key that fails keeps the full pool and is used point by point exactly as the reference uses it.
#include "ec.cuh"
#include "frstage.cuh"
#include "msm.cuh"

#include <map>
#include <memory>
#include <string.h>

namespace zk {

static constexpr int PIN_G1 = 6, PIN_G2 = 2;

// Everything one proof in flight needs: the pipelined form keeps several of them busy on one key
// (one stream each -- see groth16.hip for why not three).
struct PinSlot {
    FrScratch fs;
    MsmWorkspace ws1[PIN_G1], ws2[PIN_G2];
    DevBuf scal1[PIN_G1], scal2[PIN_G2];
    DevBuf wit_raw, deltas, results, out_dev;
    DevBuf kappa;                     // compact derived pool: kappa_v | kappa_w | the partial sums of frstage_leading_coeffs
    hipStream_t st = nullptr;
    hipStream_t s1 = nullptr, s2 = nullptr;      // slot 0 only: the G2 pair and the h pool run beside the five I_mid pools while the proof is alone
    hipEvent_t done = nullptr, fork = nullptr, join1 = nullptr, join2 = nullptr;
    uint8_t* host = nullptr;          // pinned: proof 960 B | flags 4 B | deltas 96 B
    bool busy = false;
    ~PinSlot() {
        if (st) (void)hipStreamDestroy(st);
        if (s1) (void)hipStreamDestroy(s1);
        if (s2) (void)hipStreamDestroy(s2);
        if (fork) { (void)hipEventDestroy(fork); (void)hipEventDestroy(join1); (void)hipEventDestroy(join2); }
        if (done) (void)hipEventDestroy(done);
        if (host) (void)hipHostFree(host);
    }
};
static constexpr uint32_t PIN_MAX_SLOTS = 15;
// key points are checked like the reference checks them on the way in (of_bytes_exn, curve.ml:199-212: encoding, curve, prime-order subgroup);
// ZK_KEY_SUBGROUP_CHECK=0 skips the subgroup part (see groth16.hip)
static bool key_subgroup_check() {          // read per key (set-up path); without the check a pool never gets folded windows (msm.cuh)
    const char* e = ::zk::opt("ZK_KEY_SUBGROUP_CHECK");
    return !(e && atoi(e) == 0);
}
struct PinKey {
    uint32_t n = 0, m = 0, n_mid = 0;
    // A whole key is shard 0 of 1.  A shard of a multi-device key (PinGroup below) holds the points [lo, hi) of every pool -- the pools are cut in `world`
    // equal slices -- with its own window tables; full = the pool's length.  An EMPTY slice (a pool shorter than the device list) has no base set at all.
    uint32_t rank = 0, world = 1;
    uint64_t full1[PIN_G1] = {}, lo1[PIN_G1] = {}, hi1[PIN_G1] = {};
    uint64_t full2[PIN_G2] = {}, lo2[PIN_G2] = {}, hi2[PIN_G2] = {};
    FrStage fr;
    MsmBases g1[PIN_G1], g2[PIN_G2];
    DevBuf mid_idx, wit_resident;
    bool have_witness = false;
    bool lagrange = false;              // pool 5 holds [lambda_t(s)] (n-1) | [Z(s)] | [1] | v_all | w_all instead of si (n+1) | v_all | w_all
    // One sort per distinct scalar vector (round 5): vv|vt and vav|vavt carry (c_mid | dv), yy|yt and yay|yayt (c_mid | dy), ww|wt and waw|wawt (c_mid | dw)
    // (pinocchio.ml:438-447,489-498).  share1[i] / share2[i] = the pool whose sorted references pool i's bucket accumulation reads, -1 = its own: set at
    // upload when the two base sets have the same geometry AND the same identity flags (msm_bases_same_geometry; v_k = 0 makes both of a pair the identity).
    int share1[PIN_G1] = {-1, -1, -1, -1, -1, -1}, share2[PIN_G2] = {-1, -1};
    bool compact = false;               // pool 5 without v_all | w_all: si (n+1), or derived [lambda_t(s)] (n-1) | [Z(s)] | [1] | [s^(n-1)]  (header comment)
    DevBuf pw;                          // compact + derived: (n + t)^(n-1), t < n - 1
    std::unique_ptr<PinSlot> slots[PIN_MAX_SLOTS];
};
int derive_shifted_bases_g1(const FrStage& f, const uint8_t* d_si, uint8_t* d_out, hipStream_t s);   // lagrange_derive.hip
static std::map<uint64_t, std::unique_ptr<PinKey>>& g_pin = *new std::map<uint64_t, std::unique_ptr<PinKey>>;   // never destroyed (see ntt.hip)
static uint64_t g_pin_next = 0x5000000001ull;

static inline dim3 g1d(uint64_t n, unsigned t = 256) { return dim3((unsigned)((n + t - 1) / t)); }

struct PinScalPtrs {
    uint32_t* s1[PIN_G1];
    uint32_t* s2[PIN_G2];
};
// the inputs of the h product's scalar vector, by key form (PinKey::lagrange, PinKey::compact)
struct PinHArgs {
    const uint32_t* h;          // tau-power: n - 1 coefficients; derived: h(n + t), t < n - 1
    const uint32_t* z;          // Z's n + 1 coefficients
    const uint32_t* vco;        // compact, tau-power: the coefficients of v and w (n each)
    const uint32_t* wco;
    const uint32_t* sa;         // compact, derived: the convolution outputs Sa_j, Sb_j at j = n + t (v(j) = zt[t] Sa_j), zt, (n + t)^(n-1), kappa_v | kappa_w
    const uint32_t* sb;
    const uint32_t* zt;
    const uint32_t* pw;
    const uint32_t* kappa;
    uint32_t lagrange, compact;
};
static inline uint64_t pin_h_points(uint32_t n, uint32_t m, bool lagrange, bool compact) {
    return compact ? (uint64_t)n + (lagrange ? 2 : 1) : (uint64_t)n + 1 + 2 * (uint64_t)m;
}
// one lane per entry of the longest vector (the h pool, or the pools over the mids where the compact h pool is shorter)
__global__ void k_pinocchio_scalars(PinScalPtrs out, PinHArgs ha, const uint32_t* __restrict__ wit_mont, const uint32_t* __restrict__ mid_idx,
                                    const uint32_t* __restrict__ deltas, uint32_t n, uint32_t m, uint32_t n_mid) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lagrange = ha.lagrange;
    const uint64_t ph = ha.compact ? (uint64_t)n + (lagrange ? 2 : 1) : (uint64_t)n + 1 + 2 * (uint64_t)m;
    if (i >= ph && i >= (uint64_t)n_mid + 3) return;
    const Fr dv = fe_to_mont(fe_load<FrParams>(deltas)), dw = fe_to_mont(fe_load<FrParams>(deltas + 8)),
             dy = fe_to_mont(fe_load<FrParams>(deltas + 16));
    const uint32_t* h = ha.h;
    const uint32_t* z = ha.z;
    // pool 5: h'
    if (i < ph) {
        Fr x;
        if (ha.compact && lagrange) {
            const Fr kv = fe_load<FrParams>(ha.kappa), kw = fe_load<FrParams>(ha.kappa + 8);
            if (i + 1 < n) {
                const Fr p = fe_load<FrParams>(ha.pw + 8 * i), zt = fe_load<FrParams>(ha.zt + 8 * i);
                const Fr ve = fe_sub(fe_mul(zt, fe_load<FrParams>(ha.sa + 8 * ((uint64_t)n + i))), fe_mul(kv, p));      // (v - kappa_v X^(n-1))(n + t)
                const Fr we = fe_sub(fe_mul(zt, fe_load<FrParams>(ha.sb + 8 * ((uint64_t)n + i))), fe_mul(kw, p));
                x = fe_add(fe_load<FrParams>(h + 8 * i), fe_add(fe_mul(dw, ve), fe_mul(dv, we)));
            } else if (i + 1 == n) x = fe_mul(dv, dw);                                   // [Z(s)]
            else if (i == n) x = fe_neg(dy);                                             // [1]
            else x = fe_add(fe_mul(dw, kv), fe_mul(dv, kw));                             // [s^(n-1)]
        } else if (ha.compact) {
            x = fe_mul(fe_mul(dv, dw), fe_load<FrParams>(z + 8 * i));                    // dv dw Z_i, i <= n
            if (i + 1 < n) x = fe_add(x, fe_load<FrParams>(h + 8 * i));
            if (i < n) x = fe_add(x, fe_add(fe_mul(dw, fe_load<FrParams>(ha.vco + 8 * i)), fe_mul(dv, fe_load<FrParams>(ha.wco + 8 * i))));
            if (i == 0) x = fe_sub(x, dy);
        } else if (i <= n && lagrange) {
            // derived key: h through its VALUES on n .. 2n-2 against [lambda_t(s)], then dv dw on the single point [Z(s)], -dy on [1]
            if (i + 1 < n) x = fe_load<FrParams>(h + 8 * i);
            else if (i + 1 == n) x = fe_mul(dv, dw);
            else x = fe_neg(dy);
        } else if (i <= n) {
            x = fe_mul(fe_mul(dv, dw), fe_load<FrParams>(z + 8 * i));            // dv dw Z_i
            if (i + 1 < n) x = fe_add(x, fe_load<FrParams>(h + 8 * i));          // h has n - 1 coefficients
            if (i == 0) x = fe_sub(x, dy);
        } else if (i <= (uint64_t)n + m) x = fe_mul(dw, fe_load<FrParams>(wit_mont + 8 * (i - n - 1)));
        else x = fe_mul(dv, fe_load<FrParams>(wit_mont + 8 * (i - n - 1 - m)));
        fe_store<FrParams>(out.s1[5] + 8 * i, fe_from_mont(x));
    }
    // pools over the mids
    if (i < (uint64_t)n_mid + 3) {
        Fr c = fe_zero<FrParams>();
        const bool is_mid = i < n_mid;
        if (is_mid) c = fe_load<FrParams>(wit_mont + 8 * (uint64_t)mid_idx[i]);
        const uint64_t e = i - n_mid;      // index among the appended single points
        auto put = [&](uint32_t* dst, uint32_t extras, const Fr& e0, const Fr& e1, const Fr& e2) {
            if (!dst) return;                                    // a pool that reads another pool's sort (PinKey::share1 / share2)
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

// ---- the compact h pool's precondition: v_all / w_all are the images of the key's own powers si (header comment)
static bool pin_compact_wanted() {          // read per key (set-up path)
    const char* e = ::zk::opt("ZK_PIN_COMPACT_H");
    return !(e && atoi(e) == 0);
}
// rho[k]: pseudo-random canonical scalars below 2^254 (splitmix64 of the index: a consistency check of a key against itself, not a secret)
__global__ void k_pin_rho(uint32_t* __restrict__ rho, uint64_t m, uint64_t seed) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    Fr r;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint64_t x = seed + (4 * k + q + 1) * 0x9E3779B97F4A7C15ull;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        x ^= x >> 31;
        r.v[2 * q] = (uint32_t)x;
        r.v[2 * q + 1] = (uint32_t)(x >> 32);
    }
    r.v[7] &= 0x3fffffffu;
    fe_store<FrParams>(rho + 8 * k, r);
}
// four scalar vectors over si | v_all | w_all (canonical):  0: coefficients of v_rho on si,  1: rho on v_all,  2: coefficients of w_rho on si,  3: rho on w_all
__global__ void k_pin_check_scalars(uint32_t* __restrict__ out, const uint32_t* __restrict__ vco, const uint32_t* __restrict__ wco,
                                    const uint32_t* __restrict__ rho, uint32_t n, uint32_t m) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t ph = (uint64_t)n + 1 + 2 * (uint64_t)m;
    if (i >= ph) return;
    const Fr zero = fe_zero<FrParams>();
    Fr x[4] = {zero, zero, zero, zero};
    if (i < n) { x[0] = fe_from_mont(fe_load<FrParams>(vco + 8 * i)); x[2] = fe_from_mont(fe_load<FrParams>(wco + 8 * i)); }
    else if (i > n && i <= (uint64_t)n + m) x[1] = fe_load<FrParams>(rho + 8 * (i - n - 1));
    else if (i > (uint64_t)n + m) x[3] = fe_load<FrParams>(rho + 8 * (i - n - 1 - m));
#pragma unroll
    for (int q = 0; q < 4; q++) fe_store<FrParams>(out + 8 * (q * ph + i), x[q]);
}
// *ok = the key's v_all / w_all equal sum_j (v_k)_j si[j] / sum_j (w_k)_j si[j] for every variable k (up to the 2^-254 of a random combination).
// `full`: si | v_all | w_all as a plain (one-window) base set.
static int pin_compact_check(const FrStage& fr, const MsmBases& full, uint32_t n, uint32_t m, bool* ok, hipStream_t s) {
    *ok = false;
    const uint64_t ph = (uint64_t)n + 1 + 2 * (uint64_t)m;
    if (full.n != ph) ZK_FAIL(ZK_ERR_ARG, "pin_compact_check: unexpected pool length");
    FrScratch fs;
    MsmWorkspace w;
    DevBuf rho, scal, res;
    ZKCHK(frstage_scratch_alloc(fr, fs));
    ZKCHK(msm_workspace_alloc(w, full));
    ZKCHK(rho.alloc(32 * (size_t)m));
    ZKCHK(scal.alloc(32 * 4 * (size_t)ph));
    ZKCHK(res.alloc(4 * xyzz_bytes(CURVE_G1)));
    hipLaunchKernelGGL(k_pin_rho, g1d(m), dim3(256), 0, s, rho.as<uint32_t>(), (uint64_t)m, 0x5EEDC0DE2026ull ^ ((uint64_t)n << 32) ^ m);
    HIPCHK(hipGetLastError());
    ZKCHK(frstage_eval(fr, fs, rho.p, s));          // v_rho = sum_k rho_k v_k and w_rho as coefficient vectors (rho satisfies no gate: flags and h are not read)
    hipLaunchKernelGGL(k_pin_check_scalars, g1d(ph), dim3(256), 0, s, scal.as<uint32_t>(), (const uint32_t*)fs.d.as<uint32_t>(),
                       (const uint32_t*)(fs.d.as<uint32_t>() + 8 * (uint64_t)fr.n2), (const uint32_t*)rho.as<uint32_t>(), n, m);
    HIPCHK(hipGetLastError());
    for (int q = 0; q < 4; q++) ZKCHK(msm_run(full, w, scal.as<uint8_t>() + 32 * q * ph, res.as<uint8_t>() + q * xyzz_bytes(CURVE_G1), s));
    uint8_t pts[4 * 96];
    ZKCHK(points_xyzz_to_bytes(CURVE_G1, res.p, 4, pts, s));          // synchronises
    *ok = memcmp(pts, pts + 96, 96) == 0 && memcmp(pts + 192, pts + 288, 96) == 0;
    return ZK_OK;
}

static int pin_lookup(uint64_t handle, PinKey** out) {
    auto it = g_pin.find(handle);
    if (it == g_pin.end()) ZK_FAIL(ZK_ERR_HANDLE, "unknown Pinocchio key handle");
    *out = it->second.get();
    return ZK_OK;
}
static inline void pin_slice(uint64_t full, uint32_t rank, uint32_t world, uint64_t* lo, uint64_t* hi) {
    *lo = full * rank / world;
    *hi = full * (rank + 1) / world;
}

// Builds a key -- whole (rank 0 of world 1) or rank's shard -- on the CURRENT virtual device; nothing is registered under a handle.
// compact_in: -1 = decide here (the upload's consistency check of v_all / w_all against si, header comment: needs the WHOLE h pool, so rank 0 of a
// group runs it and hands its verdict to the other shards), 0 / 1 = the verdict.
static int pin_key_build(std::unique_ptr<PinKey>& out, uint32_t n, uint32_t m, const zk_csr* L, const zk_csr* R, const zk_csr* O, const uint8_t* mid,
                         const uint8_t* pk_g1, size_t pk_g1_points, const uint8_t* pk_g2, size_t pk_g2_points, uint32_t rank, uint32_t world, int compact_in) {
    if (!mid || !pk_g1 || !pk_g2) ZK_FAIL(ZK_ERR_ARG, "pinocchio pk_upload: null argument");
    Ctx& c = ctx();
    auto key = std::make_unique<PinKey>();
    PinKey& k = *key;
    k.n = n; k.m = m; k.rank = rank; k.world = world;
    std::vector<uint32_t> mids;
    for (uint32_t i = 0; i < m; i++)
        if (mid[i]) mids.push_back(i);
    const uint64_t nm = k.n_mid = (uint32_t)mids.size();
    if (n < 1) ZK_FAIL(ZK_ERR_ARG, "pinocchio pk_upload: need at least 1 constraint");      // an EMPTY I_mid is fine: the pools keep their appended single points
    if (pk_g1_points != 5 * nm + (n + 1) + 2 * (uint64_t)m + 7) ZK_FAIL(ZK_ERR_DOMAIN, "pinocchio pk_upload: G1 key length");
    if (pk_g2_points != 2 * nm + (n + 1) + 2) ZK_FAIL(ZK_ERR_DOMAIN, "pinocchio pk_upload: G2 key length");
    ZKCHK(frstage_init(k.fr, n, m, L, R, O, c.stream));
    // slices of the flattened key (pinocchio.ml:37-60; layout in include/zkmi355x.h)
    const uint8_t *VV = pk_g1, *YY = VV + 96 * nm, *VAV = YY + 96 * nm, *YAY = VAV + 96 * nm, *BV = YAY + 96 * nm,
                  *SI = BV + 96 * nm, *VALL = SI + 96 * (uint64_t)(n + 1), *WALL = VALL + 96 * (uint64_t)m, *ONES = WALL + 96 * (uint64_t)m;
    const uint8_t *WW = pk_g2, *WAW = WW + 192 * nm, *ONES2 = WAW + 192 * nm + 192 * (uint64_t)(n + 1);
    std::vector<uint8_t> buf;
    // pool idx = base[0 .. cnt) | extras, of which this shard keeps [lo, hi)
    auto pool1 = [&](int idx, const uint8_t* base, uint64_t cnt, std::initializer_list<const uint8_t*> extras) -> int {
        const uint64_t full = cnt + extras.size();
        k.full1[idx] = full;
        pin_slice(full, rank, world, &k.lo1[idx], &k.hi1[idx]);
        const uint64_t lo = k.lo1[idx], hi = k.hi1[idx];
        if (hi == lo) return ZK_OK;
        buf.clear();
        if (lo < cnt) buf.assign(base + 96 * lo, base + 96 * (hi < cnt ? hi : cnt));
        uint64_t e = cnt;
        for (auto x : extras) {
            if (e >= lo && e < hi) buf.insert(buf.end(), x, x + 96);
            e++;
        }
        return msm_bases_from_bytes(k.g1[idx], CURVE_G1, buf.data(), buf.size() / 96, 0, true, c.stream, key_subgroup_check());
    };
    ZKCHK(pool1(0, VV, nm, {ONES + 96 * 0}));
    ZKCHK(pool1(1, YY, nm, {ONES + 96 * 1}));
    ZKCHK(pool1(2, VAV, nm, {ONES + 96 * 2}));
    ZKCHK(pool1(3, YAY, nm, {ONES + 96 * 3}));
    ZKCHK(pool1(4, BV, nm, {ONES + 96 * 4, ONES + 96 * 5, ONES + 96 * 6}));
    const uint64_t ph_full = (uint64_t)n + 1 + 2 * (uint64_t)m;          // si | v_all | w_all are contiguous in the key
    int compact = compact_in;
    if (compact < 0) compact = 0;
    if (compact_in < 0 && pin_compact_wanted()) {
        // every point is decoded and checked as before (of_bytes_exn); the pool then keeps v_all | w_all only if they fail the check of the header comment
        MsmBases full;
        ZKCHK(msm_bases_from_bytes(full, CURVE_G1, SI, ph_full, 0, false, c.stream, key_subgroup_check()));
        bool ok = false;
        ZKCHK(pin_compact_check(k.fr, full, n, m, &ok, c.stream));
        compact = ok ? 1 : 0;
    }
    k.compact = compact != 0;
    ZKCHK(pool1(5, SI, k.compact ? (uint64_t)n + 1 : ph_full, {}));          // this shard's slice decodes and checks its own points (rank 0 has seen all of them when it ran the check)
    auto pool2 = [&](int idx, const uint8_t* base, const uint8_t* extra) -> int {
        k.full2[idx] = nm + 1;
        pin_slice(nm + 1, rank, world, &k.lo2[idx], &k.hi2[idx]);
        const uint64_t lo = k.lo2[idx], hi = k.hi2[idx];
        if (hi == lo) return ZK_OK;
        buf.clear();
        if (lo < nm) buf.assign(base + 192 * lo, base + 192 * (hi < nm ? hi : nm));
        if (hi == nm + 1) buf.insert(buf.end(), extra, extra + 192);
        return msm_bases_from_bytes(k.g2[idx], CURVE_G2, buf.data(), buf.size() / 192, 0, true, c.stream, key_subgroup_check());
    };
    ZKCHK(pool2(0, WW, ONES2));
    ZKCHK(pool2(1, WAW, ONES2 + 192));
    {
        const char* e = ::zk::opt("ZK_PIN_SHARED_SORT");          // 0: every product sorts for itself (A/B, tests)
        if (!(e && atoi(e) == 0)) {
            auto try_share = [&](const MsmBases& a, const MsmBases& b, int* slot, int from) -> int {
                if (!a.n || !b.n) return ZK_OK;
                bool same = false;
                ZKCHK(msm_bases_same_geometry(a, b, &same, c.stream));
                if (same) *slot = from;
                return ZK_OK;
            };
            ZKCHK(try_share(k.g1[0], k.g1[2], &k.share1[2], 0));
            ZKCHK(try_share(k.g1[1], k.g1[3], &k.share1[3], 1));
            ZKCHK(try_share(k.g2[0], k.g2[1], &k.share2[1], 0));
        }
    }
    ZKCHK(k.mid_idx.alloc(4 * (nm ? nm : 1)));
    if (nm) HIPCHK(hipMemcpyAsync(k.mid_idx.p, mids.data(), 4 * nm, hipMemcpyHostToDevice, c.stream));
    ZKCHK(k.wit_resident.alloc(32 * (size_t)m));
    HIPCHK(hipStreamSynchronize(c.stream));
    out = std::move(key);
    return ZK_OK;
}

// ---- the derivation of the h pool (zk_pinocchio_pk_derive_lagrange), in two halves so that a multi-device key can run the first on one device
// d_old: the WHOLE h pool as uploaded, dense affine (si | v_all | w_all, or si alone when compact), on the current device.
// pool_out: the whole derived pool, dense affine:  [lambda_t(s)] (n-1) | [Z(s)] | [1] | [s^(n-1)]   resp.   ... | [1] | v_all | w_all
static int pin_derive_pool(const PinKey& k, const uint8_t* d_old, DevBuf& pool_out, hipStream_t s) {
    const uint32_t n = k.n;
    const uint64_t ph = pin_h_points(n, k.m, true, k.compact);
    ZKCHK(pool_out.alloc(96 * ph));
    uint8_t* pool = pool_out.as<uint8_t>();
    const uint8_t* si = d_old;
    // [lambda_t(s)], t < n - 1
    ZKCHK(derive_shifted_bases_g1(k.fr, si, pool, s));
    // [Z(s)] = sum_i Z_i [s^i]: one MSM over the n + 1 powers with the canonical coefficients of Z
    {
        MsmBases b;
        MsmWorkspace w;
        DevBuf zc, res, bytes, flag;
        ZKCHK(msm_bases_from_device_affine(b, CURVE_G1, si, (uint64_t)n + 1, 0, false, s));
        ZKCHK(msm_workspace_alloc(w, b));
        ZKCHK(zc.alloc(32 * ((size_t)n + 1)));
        ZKCHK(res.alloc(xyzz_bytes(CURVE_G1)));
        ZKCHK(bytes.alloc(96));
        ZKCHK(flag.alloc(4));
        ZKCHK(fr_from_mont(zc.p, k.fr.z.p, (uint64_t)n + 1, s));
        ZKCHK(msm_run(b, w, zc.p, res.p, s));
        ZKCHK(points_xyzz_to_bytes_dev(CURVE_G1, res.p, 1, bytes.p, s));
        HIPCHK(hipMemsetAsync(flag.p, 0, 4, s));
        ZKCHK(points_bytes_to_affine(CURVE_G1, pool + 96 * (uint64_t)(n - 1), bytes.p, 1, flag.as<int>(), s));
        HIPCHK(hipStreamSynchronize(s));
    }
    HIPCHK(hipMemcpyAsync(pool + 96 * (uint64_t)n, si, 96, hipMemcpyDeviceToDevice, s));                                       // [1] = si[0]
    if (k.compact) HIPCHK(hipMemcpyAsync(pool + 96 * ((uint64_t)n + 1), si + 96 * (uint64_t)(n - 1), 96, hipMemcpyDeviceToDevice, s));    // [s^(n-1)]
    else HIPCHK(hipMemcpyAsync(pool + 96 * ((uint64_t)n + 1), si + 96 * ((uint64_t)n + 1), 96 * 2 * (uint64_t)k.m, hipMemcpyDeviceToDevice, s));   // v_all | w_all
    HIPCHK(hipStreamSynchronize(s));
    return ZK_OK;
}
// d_pool: the whole derived pool on the current device; the key takes its slice of it (own window tables), the Fr stage flips to the values of h
static int pin_install_derived(PinKey& k, const uint8_t* d_pool, hipStream_t s) {
    const uint32_t n = k.n;
    const uint64_t ph = pin_h_points(n, k.m, true, k.compact);
    const MsmBases& old = k.g1[5];
    uint64_t lo, hi;
    pin_slice(ph, k.rank, k.world, &lo, &hi);
    MsmBases nb;
    if (hi > lo) ZKCHK(msm_bases_from_device_affine(nb, CURVE_G1, d_pool + 96 * lo, hi - lo, old.n ? old.c : 0, true, s, old.n ? old.in_subgroup : key_subgroup_check()));   // derived from checked points: msm.cuh, msm_fold
    if (k.compact) {
        ZKCHK(k.pw.alloc(32 * (size_t)(n > 1 ? n - 1 : 1)));
        ZKCHK(frstage_shifted_powers(k.pw.p, n, n - 1, n - 1, s));
    }
    ZKCHK(frstage_init_lagrange(k.fr, s));
    HIPCHK(hipStreamSynchronize(s));
    k.g1[5] = std::move(nb);
    k.full1[5] = ph; k.lo1[5] = lo; k.hi1[5] = hi;
    k.lagrange = true;
    for (uint32_t i = 0; i < PIN_MAX_SLOTS; i++) k.slots[i].reset();          // the h pool changed its length (compact) and the Fr scratch its form: slots are rebuilt on demand
    return ZK_OK;
}
static int pin_check_idle(PinKey& k, const char* who) {
    for (uint32_t i = 0; i < PIN_MAX_SLOTS; i++)
        if (k.slots[i] && k.slots[i]->busy) ZK_FAIL(ZK_ERR_ARG, who);
    return ZK_OK;
}
// this shard's slice of pool `pool` (0..5 G1, 6..7 G2) as uncompressed points at out (host)
static int pin_slice_points(PinKey& k, int pool, uint8_t* out) {
    const MsmBases& b = pool < PIN_G1 ? k.g1[pool] : k.g2[pool - PIN_G1];
    if (!b.n) return ZK_OK;
    Ctx& c = ctx();
    DevBuf bytes, dense;
    ZKCHK(bytes.alloc(aff_bytes(b.curve) * b.n));
    ZKCHK(dense.alloc(aff_bytes(b.curve) * b.n));
    ZKCHK(msm_bases_dense(b, 0, b.n, dense.p, c.stream));
    ZKCHK(points_affine_to_bytes(b.curve, bytes.p, dense.p, b.n, c.stream));
    HIPCHK(hipMemcpyAsync(out, bytes.p, aff_bytes(b.curve) * b.n, hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    return ZK_OK;
}

static int pin_slot_get(PinKey& k, uint32_t idx, PinSlot** out) {
    if (idx >= PIN_MAX_SLOTS) ZK_FAIL(ZK_ERR_ARG, "pinocchio: slot index out of range (max 15 proofs in flight)");
    if (!k.slots[idx]) {
        auto sl = std::make_unique<PinSlot>();
        ZKCHK(frstage_scratch_alloc(k.fr, sl->fs));
        // scalar vectors over the FULL pools (any shard may own a proof's Fr stage); workspaces for the shard's own base sets
        for (int i = 0; i < PIN_G1; i++) { if (k.g1[i].n) ZKCHK(msm_workspace_alloc(sl->ws1[i], k.g1[i])); ZKCHK(sl->scal1[i].alloc(32 * k.full1[i])); }
        for (int i = 0; i < PIN_G2; i++) { if (k.g2[i].n) ZKCHK(msm_workspace_alloc(sl->ws2[i], k.g2[i])); ZKCHK(sl->scal2[i].alloc(32 * k.full2[i])); }
        ZKCHK(sl->wit_raw.alloc(32 * (size_t)k.m));
        ZKCHK(sl->deltas.alloc(96));
        ZKCHK(sl->kappa.alloc(32 * (2 + 2 * (size_t)LEAD_BLOCKS)));
        ZKCHK(sl->results.alloc(PIN_G1 * xyzz_bytes(CURVE_G1) + PIN_G2 * xyzz_bytes(CURVE_G2)));
        ZKCHK(sl->out_dev.alloc(960));
        HIPCHK(hipStreamCreateWithFlags(&sl->st, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&sl->done, hipEventDisableTiming));
        if (idx == 0 && k.world == 1 && !::zk::opt("ZK_SERIAL_STREAMS")) {
            // the slot of the synchronous zk_pinocchio_prove: two more streams, used only while no other proof is in flight (groth16.hip, slot 0)
            HIPCHK(hipStreamCreateWithFlags(&sl->s1, hipStreamNonBlocking));
            HIPCHK(hipStreamCreateWithFlags(&sl->s2, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&sl->fork, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&sl->join1, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&sl->join2, hipEventDisableTiming));
        }
        HIPCHK(hipHostMalloc((void**)&sl->host, 1088, hipHostMallocDefault));
        k.slots[idx] = std::move(sl);
    }
    *out = k.slots[idx].get();
    return ZK_OK;
}

// First half of a proof, on the slot's stream: the Fr stage, then the eight scalar vectors over the FULL pools.  all_vectors: also the vectors of
// pools that read another pool's sort on THIS key (a multi-device key's other shards decide on their own slices).
static int pin_scalars_enqueue(PinKey& k, PinSlot& sl, const uint8_t* sol, const uint8_t dv[32], const uint8_t dw[32], const uint8_t dy[32], bool all_vectors) {
    hipStream_t s0 = sl.st;
    const void* wit = k.wit_resident.p;
    if (sol) {
        HIPCHK(hipMemcpyAsync(sl.wit_raw.p, sol, 32 * (size_t)k.m, hipMemcpyHostToDevice, s0));
        wit = sl.wit_raw.p;
    } else if (!k.have_witness) ZK_FAIL(ZK_ERR_ARG, "no witness: pass sol or call zk_pinocchio_set_witness first");
    memcpy(sl.host + 992, dv, 32); memcpy(sl.host + 1024, dw, 32); memcpy(sl.host + 1056, dy, 32);
    HIPCHK(hipMemcpyAsync(sl.deltas.p, sl.host + 992, 96, hipMemcpyHostToDevice, s0));
    if (k.lagrange) ZKCHK(frstage_eval_lagrange(k.fr, sl.fs, wit, s0));
    else ZKCHK(frstage_eval(k.fr, sl.fs, wit, s0));
    PinScalPtrs ptrs;
    for (int i = 0; i < PIN_G1; i++) ptrs.s1[i] = all_vectors || k.share1[i] < 0 ? sl.scal1[i].as<uint32_t>() : nullptr;
    for (int i = 0; i < PIN_G2; i++) ptrs.s2[i] = all_vectors || k.share2[i] < 0 ? sl.scal2[i].as<uint32_t>() : nullptr;
    const uint64_t ph = pin_h_points(k.n, k.m, k.lagrange, k.compact);
    PinHArgs ha{};
    ha.h = sl.fs.h.as<uint32_t>(); ha.z = k.fr.z.as<uint32_t>();
    ha.lagrange = k.lagrange ? 1u : 0u; ha.compact = k.compact ? 1u : 0u;
    if (k.compact && k.lagrange) {
        uint32_t* kap = sl.kappa.as<uint32_t>();
        ZKCHK(frstage_leading_coeffs(k.fr, sl.fs, kap, kap + 16, s0));
        ha.sa = sl.fs.bufA.as<uint32_t>(); ha.sb = sl.fs.bufA.as<uint32_t>() + 8 * (uint64_t)k.fr.S;
        ha.zt = k.fr.zt.as<uint32_t>(); ha.pw = k.pw.as<uint32_t>(); ha.kappa = kap;
    } else if (k.compact) {
        ha.vco = sl.fs.d.as<uint32_t>(); ha.wco = sl.fs.d.as<uint32_t>() + 8 * (uint64_t)k.fr.n2;
    }
    const uint64_t lanes = ph > (uint64_t)k.n_mid + 3 ? ph : (uint64_t)k.n_mid + 3;
    hipLaunchKernelGGL(k_pinocchio_scalars, g1d(lanes), dim3(256), 0, s0, ptrs, ha, (const uint32_t*)sl.fs.wit.as<uint32_t>(),
                       (const uint32_t*)k.mid_idx.as<uint32_t>(), (const uint32_t*)sl.deltas.as<uint32_t>(), k.n, k.m, k.n_mid);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(sl.host + 960, sl.fs.flag.p, 4, hipMemcpyDeviceToHost, s0));
    return ZK_OK;
}
// proof byte offsets: vv 0 | ww 96 | yy 288 | h 384 | vavv 480 | waww 576 | yayy 768 | bvwy 864
static const uint32_t PIN_OFF1[PIN_G1] = {0, 288, 480, 768, 864, 384};
static const uint32_t PIN_OFF2[PIN_G2] = {96, 576};
// Second half: the eight products over this key's slices of the pools, scalars from sl.scal*[i] + lo.  raw: the XYZZ (partial) sums stay in sl.results --
// six G1 then two G2 points, the identity for an empty slice -- for the first device of a multi-device key to add up; else affine bytes in sl.out_dev.
static int pin_msms_enqueue(PinKey& k, PinSlot& sl, bool raw) {
    hipStream_t s0 = sl.st;
    char* res = sl.results.as<char>();
    char* out = sl.out_dev.as<char>();
    const size_t x1 = xyzz_bytes(CURVE_G1), x2 = xyzz_bytes(CURVE_G2);
    auto sc1 = [&](int i) { return (const void*)(sl.scal1[i].as<char>() + 32 * k.lo1[i]); };
    auto sc2 = [&](int i) { return (const void*)(sl.scal2[i].as<char>() + 32 * k.lo2[i]); };
    // One group of products on one stream: sort + bucket accumulation per MSM; the reductions of every product whose pool got the same
    // window plan go out as ONE mixed G1 / G2 chain of launches (from 2^16 constraints up that is the whole group); one to-bytes launch.
    // G1 products [lo1, hi1), G2 products [lo2, hi2).
    auto group = [&](int lo1, int hi1, int lo2, int hi2, hipStream_t st) -> int {
        // a pool whose scalar vector another pool of the group has sorted reads that sort (the source has the lower index: it ran first on this stream)
        for (int i = lo2; i < hi2; i++) {
            if (!k.g2[i].n) { HIPCHK(hipMemsetAsync(res + PIN_G1 * x1 + i * x2, 0, x2, st)); continue; }          // empty slice: the identity
            const int f = k.share2[i];
            if (f >= lo2 && f < i) { MsmWorkspace *w1[1] = {&sl.ws2[i]}, *f1[1] = {&sl.ws2[f]}; ZKCHK(msm_accumulate_sorted(k.g2[i], w1, f1, 1, st)); }
            else ZKCHK(msm_sort_accumulate(k.g2[i], sl.ws2[i], f < 0 ? sc2(i) : sc2(f), st));
        }
        for (int i = lo1; i < hi1; i++) {
            if (!k.g1[i].n) { HIPCHK(hipMemsetAsync(res + i * x1, 0, x1, st)); continue; }
            const int f = k.share1[i];
            if (f >= lo1 && f < i) { MsmWorkspace *w1[1] = {&sl.ws1[i]}, *f1[1] = {&sl.ws1[f]}; ZKCHK(msm_accumulate_sorted(k.g1[i], w1, f1, 1, st)); }
            else ZKCHK(msm_sort_accumulate(k.g1[i], sl.ws1[i], f < 0 ? sc1(i) : sc1(f), st));
        }
        bool done1[PIN_G1] = {}, done2[PIN_G2] = {};
        for (int i = lo1; i < hi1; i++) done1[i] = !k.g1[i].n;
        for (int i = lo2; i < hi2; i++) done2[i] = !k.g2[i].n;
        for (;;) {
            int lead_c = -1, lead_nw = -1;
            for (int i = lo1; i < hi1 && lead_c < 0; i++) if (!done1[i]) { lead_c = (int)k.g1[i].c; lead_nw = (int)k.g1[i].nw; }
            for (int i = lo2; i < hi2 && lead_c < 0; i++) if (!done2[i]) { lead_c = (int)k.g2[i].c; lead_nw = (int)k.g2[i].nw; }
            if (lead_c < 0) break;
            MsmWorkspace *w1[PIN_G1], *w2[PIN_G2];
            void *o1[PIN_G1], *o2[PIN_G2];
            const MsmBases *b1 = nullptr, *b2 = nullptr;
            uint32_t n1 = 0, n2 = 0;
            for (int i = lo1; i < hi1; i++)
                if (!done1[i] && (int)k.g1[i].c == lead_c && (int)k.g1[i].nw == lead_nw) { w1[n1] = &sl.ws1[i]; o1[n1] = res + i * x1; n1++; done1[i] = true; b1 = &k.g1[i]; }
            for (int i = lo2; i < hi2; i++)
                if (!done2[i] && (int)k.g2[i].c == lead_c && (int)k.g2[i].nw == lead_nw) { w2[n2] = &sl.ws2[i]; o2[n2] = res + PIN_G1 * x1 + i * x2; n2++; done2[i] = true; b2 = &k.g2[i]; }
            ZKCHK(msm_reduce_mixed(b1, w1, o1, n1, b2, w2, o2, n2, st));
        }
        if (raw) return ZK_OK;
        uint32_t o1[PIN_G1], o2[PIN_G2];
        for (int i = lo1; i < hi1; i++) o1[i - lo1] = PIN_OFF1[i];
        for (int i = lo2; i < hi2; i++) o2[i - lo2] = PIN_OFF2[i];
        return proof_points_to_bytes_dev(res + lo1 * x1, hi1 - lo1, o1, res + PIN_G1 * x1 + lo2 * x2, hi2 - lo2, o2, out, st);
    };
    // fork only while this is the one proof in flight on the key (single-proof latency); with others in flight every slot keeps to one stream
    bool forked = sl.s1 != nullptr && ctx().profiling < 2 && !raw;
    for (uint32_t i = 0; forked && i < PIN_MAX_SLOTS; i++)
        if (k.slots[i] && k.slots[i].get() != &sl && k.slots[i]->busy) forked = false;
    if (forked) {
        HIPCHK(hipEventRecord(sl.fork, s0));
        HIPCHK(hipStreamWaitEvent(sl.s1, sl.fork, 0));
        HIPCHK(hipStreamWaitEvent(sl.s2, sl.fork, 0));
        ZKCHK(group(0, 0, 0, PIN_G2, sl.s1));                  // the G2 pair: the longest reduction chain
        HIPCHK(hipEventRecord(sl.join1, sl.s1));
        ZKCHK(group(5, 6, 0, 0, sl.s2));                       // the h pool
        HIPCHK(hipEventRecord(sl.join2, sl.s2));
        ZKCHK(group(0, 5, 0, 0, s0));                          // the five pools over I_mid
        HIPCHK(hipStreamWaitEvent(s0, sl.join1, 0));
        HIPCHK(hipStreamWaitEvent(s0, sl.join2, 0));
    } else {
        ZKCHK(group(0, PIN_G1, 0, PIN_G2, s0));                // all eight products: one chain of reductions, one to-bytes launch
    }
    return ZK_OK;
}

// ================================================================== multi-device keys: N GPUs behind ONE handle (round 5; groth16_multi.hip is the model)
// With a device list of N entries (zk_set_devices / zk_set_device_list) zk_pinocchio_pk_upload builds one SHARD per entry -- 1/N of the points of each
// of the eight pools, with its own window tables, slots and streams on its device -- and returns one handle.  A proof on slot t runs the Fr stage and
// the eight scalar vectors ONCE, on the slot's owner device (t mod N); every device copies its slices of the vectors out of the owner's memory
// (hipMemcpyPeerAsync over xGMI; a plain device copy where two shards share a card) behind an event of the owner's stream, runs the eight products over
// its slices (pinocchio.ml:438-505: each is a sum over key points, so it splits by points) and sends 1 920 bytes of XYZZ partial sums to the first
// device, which adds the N blocks per product (exact group additions: the bytes do not depend on N), converts and lands the proof in pinned memory.
// The consistency check behind the compact h pool runs on the first device (it needs the whole h pool), and so does the derivation of the h bases; the
// derived pool then travels to every device, which installs its slice.
struct PinGroupSlot {
    DevBuf parts, g1p, g2p, sum, out;          // on the first device: [device][1920] landing area, the combine's scratch
    uint8_t* host = nullptr;                   // pinned: the proof (960 B)
    hipEvent_t ev_scal = nullptr;              // the owner's scalar vectors are complete
    std::vector<hipEvent_t> ev_part;           // per device: its block has landed on the first device
    hipEvent_t done = nullptr;
    bool busy = false;
    int owner = 0;
    ~PinGroupSlot() {
        if (ev_scal) (void)hipEventDestroy(ev_scal);
        for (hipEvent_t e : ev_part)
            if (e) (void)hipEventDestroy(e);
        if (done) (void)hipEventDestroy(done);
        if (host) (void)hipHostFree(host);
    }
};
struct PinGroup {
    uint32_t n = 0, m = 0;
    bool broken = false;
    std::vector<std::unique_ptr<PinKey>> sub;
    std::unique_ptr<PinGroupSlot> slots[PIN_MAX_SLOTS];
};
static constexpr size_t PIN_PARTIAL_BYTES = PIN_G1 * 192 + PIN_G2 * 384;
static std::map<uint64_t, std::unique_ptr<PinGroup>>& g_pin_groups = *new std::map<uint64_t, std::unique_ptr<PinGroup>>;
static uint64_t g_pin_group_next = 0x7000000001ull;
static PinGroup* pin_group_lookup(uint64_t handle) {
    auto it = g_pin_groups.find(handle);
    return it == g_pin_groups.end() ? nullptr : it->second.get();
}
static void pin_group_destroy(PinGroup& g) {
    {
        DeviceScope ds(0);
        for (auto& sl : g.slots) sl.reset();
    }
    for (size_t v = 0; v < g.sub.size(); v++) {
        DeviceScope ds((int)v);
        g.sub[v].reset();
    }
}
static void pin_release() {
    g_pin.clear();
    for (auto& kv : g_pin_groups) pin_group_destroy(*kv.second);
    g_pin_groups.clear();
}
uint64_t pinocchio_live_handles() { return g_pin.size() + g_pin_groups.size(); }
static CleanupRegistrar g_pin_cleanup(pin_release);
static int pin_group_sync(PinGroup& g) {
    for (size_t v = 0; v < g.sub.size(); v++) {
        DeviceScope ds((int)v);
        HIPCHK(hipDeviceSynchronize());
    }
    return ZK_OK;
}
static int pin_group_check_idle(PinGroup& g, const char* who) {
    if (g.broken) ZK_FAIL(ZK_ERR_HIP, "multi-device Pinocchio key is inconsistent after a failed derivation: free it");
    for (auto& sl : g.slots)
        if (sl && sl->busy) ZK_FAIL(ZK_ERR_ARG, who);
    return ZK_OK;
}
static int pin_group_upload(uint32_t n, uint32_t m, const zk_csr* L, const zk_csr* R, const zk_csr* O, const uint8_t* mid, const uint8_t* pk_g1, size_t pk_g1_points,
                            const uint8_t* pk_g2, size_t pk_g2_points, uint64_t* handle) {
    const int N = ctx_count();
    auto grp = std::make_unique<PinGroup>();
    PinGroup& g = *grp;
    g.n = n; g.m = m;
    g.sub.resize(N);
    int rc = ZK_OK, compact = -1;
    // the shards one after another (set-up work; the kernels of one card's shards would take turns anyway, groth16_multi.hip): the first one decides the
    // form of the h pool for all
    for (int v = 0; v < N && rc == ZK_OK; v++) {
        DeviceScope ds(v);
        try {
            rc = pin_key_build(g.sub[v], n, m, L, R, O, mid, pk_g1, pk_g1_points, pk_g2, pk_g2_points, (uint32_t)v, (uint32_t)N, compact);
        } catch (const std::exception& e) {
            rc = set_error(ZK_ERR_HIP, e.what(), __FILE__, __LINE__);
        }
        if (rc == ZK_OK && v == 0) compact = g.sub[0]->compact ? 1 : 0;
    }
    if (rc != ZK_OK) {
        pin_group_destroy(g);
        return rc;
    }
    *handle = g_pin_group_next++;
    g_pin_groups[*handle] = std::move(grp);
    return ZK_OK;
}
static int pin_group_slot_get(PinGroup& g, uint32_t idx, PinGroupSlot** out) {
    if (idx >= PIN_MAX_SLOTS) ZK_FAIL(ZK_ERR_ARG, "pinocchio: slot index out of range (max 15 proofs in flight)");
    const size_t N = g.sub.size();
    if (!g.slots[idx]) {
        for (size_t v = 0; v < N; v++) {
            DeviceScope ds((int)v);
            PinSlot* sl;
            ZKCHK(pin_slot_get(*g.sub[v], idx, &sl));
        }
        DeviceScope ds(0);
        auto gs = std::make_unique<PinGroupSlot>();
        ZKCHK(gs->parts.alloc(PIN_PARTIAL_BYTES * N));
        ZKCHK(gs->g1p.alloc(PIN_G1 * 192 * N));
        ZKCHK(gs->g2p.alloc(PIN_G2 * 384 * N));
        ZKCHK(gs->sum.alloc(PIN_PARTIAL_BYTES));
        ZKCHK(gs->out.alloc(960));
        HIPCHK(hipHostMalloc((void**)&gs->host, 960, hipHostMallocDefault));
        HIPCHK(hipEventCreateWithFlags(&gs->done, hipEventDisableTiming));
        gs->ev_part.assign(N, nullptr);
        gs->owner = (int)(idx % N);
        {
            DeviceScope dso(gs->owner);
            HIPCHK(hipEventCreateWithFlags(&gs->ev_scal, hipEventDisableTiming));
        }
        for (size_t v = 0; v < N; v++) {
            DeviceScope dsv((int)v);
            HIPCHK(hipEventCreateWithFlags(&gs->ev_part[v], hipEventDisableTiming));
        }
        g.slots[idx] = std::move(gs);
    }
    *out = g.slots[idx].get();
    return ZK_OK;
}
static int pin_group_prove_async(PinGroup& g, const uint8_t* sol, const uint8_t dv[32], const uint8_t dw[32], const uint8_t dy[32], uint32_t slot) {
    if (g.broken) ZK_FAIL(ZK_ERR_HIP, "multi-device Pinocchio key is inconsistent after a failed derivation: free it");
    PinGroupSlot* gsp;
    ZKCHK(pin_group_slot_get(g, slot, &gsp));
    PinGroupSlot& gs = *gsp;
    if (gs.busy) ZK_FAIL(ZK_ERR_ARG, "slot still has a proof in flight: call zk_pinocchio_prove_wait first");
    const int N = (int)g.sub.size(), owner = gs.owner;
    PinSlot* so = g.sub[owner]->slots[slot].get();
    {   // ---- the Fr stage and the eight scalar vectors over the FULL pools, on the owner
        DeviceScope ds(owner);
        ZKCHK(pin_scalars_enqueue(*g.sub[owner], *so, sol, dv, dw, dy, true));
        HIPCHK(hipEventRecord(gs.ev_scal, so->st));
    }
    gs.busy = true;          // from here on the slot is in flight whatever happens: _wait (or the failure path below) drains it
    int rc = ZK_OK;
    for (int v = 0; v < N && rc == ZK_OK; v++) {
        DeviceScope ds(v);
        PinKey& k = *g.sub[v];
        PinSlot& sv = *k.slots[slot];
        auto body = [&]() -> int {
            if (v != owner) {
                HIPCHK(hipStreamWaitEvent(sv.st, gs.ev_scal, 0));
                // this device's slices of the vectors it sorts itself (a pool that reads another pool's sort on THIS shard needs none)
                for (int i = 0; i < PIN_G1; i++)
                    if (k.g1[i].n && k.share1[i] < 0)
                        ZKCHK(copy_between(sv.scal1[i].as<char>() + 32 * k.lo1[i], v, so->scal1[i].as<char>() + 32 * k.lo1[i], owner, 32 * (k.hi1[i] - k.lo1[i]), sv.st));
                for (int i = 0; i < PIN_G2; i++)
                    if (k.g2[i].n && k.share2[i] < 0)
                        ZKCHK(copy_between(sv.scal2[i].as<char>() + 32 * k.lo2[i], v, so->scal2[i].as<char>() + 32 * k.lo2[i], owner, 32 * (k.hi2[i] - k.lo2[i]), sv.st));
            }
            ZKCHK(pin_msms_enqueue(k, sv, true));
            ZKCHK(copy_between(gs.parts.as<char>() + PIN_PARTIAL_BYTES * v, 0, sv.results.p, v, PIN_PARTIAL_BYTES, sv.st));
            HIPCHK(hipEventRecord(gs.ev_part[v], sv.st));
            return ZK_OK;
        };
        rc = body();
    }
    {   // ---- first device: add the N blocks per product, convert, land the proof (one stream for all slots: groth16_multi.hip says why)
        DeviceScope ds(0);
        hipStream_t cs = ctx().stream2;
        auto body = [&]() -> int {
            for (int v = 0; v < N; v++) HIPCHK(hipStreamWaitEvent(cs, gs.ev_part[v], 0));
            if (rc != ZK_OK) return rc;
            const size_t b1 = PIN_G1 * 192, b2 = PIN_G2 * 384;
            HIPCHK(hipMemcpy2DAsync(gs.g1p.p, b1, gs.parts.p, PIN_PARTIAL_BYTES, b1, N, hipMemcpyDeviceToDevice, cs));                      // [device][six G1 sums]
            HIPCHK(hipMemcpy2DAsync(gs.g2p.p, b2, gs.parts.as<char>() + b1, PIN_PARTIAL_BYTES, b2, N, hipMemcpyDeviceToDevice, cs));        // [device][two G2 sums]
            ZKCHK(xyzz_sum_columns(CURVE_G1, gs.sum.p, gs.g1p.p, N, PIN_G1, cs));
            ZKCHK(xyzz_sum_columns(CURVE_G2, gs.sum.as<char>() + b1, gs.g2p.p, N, PIN_G2, cs));
            ZKCHK(proof_points_to_bytes_dev(gs.sum.p, PIN_G1, PIN_OFF1, gs.sum.as<char>() + b1, PIN_G2, PIN_OFF2, gs.out.p, cs));
            HIPCHK(hipMemcpyAsync(gs.host, gs.out.p, 960, hipMemcpyDeviceToHost, cs));
            return ZK_OK;
        };
        const int rc0 = body();
        if (rc == ZK_OK) rc = rc0;
        (void)hipEventRecord(gs.done, cs);
    }
    if (rc != ZK_OK) {
        (void)pin_group_sync(g);
        gs.busy = false;
    }
    return rc;
}
static int pin_group_prove_wait(PinGroup& g, uint32_t slot, uint8_t proof[960]) {
    if (slot >= PIN_MAX_SLOTS || !g.slots[slot]) ZK_FAIL(ZK_ERR_ARG, "zk_pinocchio_prove_wait: slot never used");
    PinGroupSlot& gs = *g.slots[slot];
    if (!gs.busy) ZK_FAIL(ZK_ERR_ARG, "no proof in flight on this slot");
    {
        DeviceScope ds(0);
        HIPCHK(hipEventSynchronize(gs.done));          // behind every device's block, which is behind the owner's Fr stage and its flag copy
    }
    gs.busy = false;
    int hf;
    memcpy(&hf, g.sub[gs.owner]->slots[slot]->host + 960, 4);
    if (hf & 2) ZK_FAIL(ZK_ERR_SCALAR_RANGE, "witness value >= r");
    if (hf & 1) ZK_FAIL(ZK_ERR_REMAINDER, "p mod Z != 0");
    memcpy(proof, gs.host, 960);
    return ZK_OK;
}
static int pin_group_derive(PinGroup& g) {
    if (g.sub[0]->lagrange) return ZK_OK;
    ZKCHK(pin_group_check_idle(g, "zk_pinocchio_pk_derive_lagrange: a proof is in flight on this key"));
    ZKCHK(pin_group_sync(g));
    const int N = (int)g.sub.size();
    PinKey& k0 = *g.sub[0];
    const uint64_t ph_old = k0.full1[5], ph = pin_h_points(k0.n, k0.m, true, k0.compact);
    // ---- gather the h pool as uploaded on the first device (window 0 of a shard's tables IS its slice in pool order)
    DevBuf old_dense;
    {
        DeviceScope ds(0);
        ZKCHK(old_dense.alloc(96 * ph_old));
    }
    for (int v = 0; v < N; v++) {
        DeviceScope ds(v);
        PinKey& k = *g.sub[v];
        if (!k.g1[5].n) continue;
        Ctx& c = ctx();
        DevBuf t;
        ZKCHK(t.alloc(96 * k.g1[5].n));
        ZKCHK(msm_bases_dense(k.g1[5], 0, k.g1[5].n, t.p, c.stream));
        ZKCHK(copy_between(old_dense.as<char>() + 96 * k.lo1[5], 0, t.p, v, 96 * k.g1[5].n, c.stream));
        HIPCHK(hipStreamSynchronize(c.stream));
    }
    // ---- derive on the first device, send the whole derived pool to every device
    std::vector<DevBuf> pool(N);
    {
        DeviceScope ds(0);
        ZKCHK(pin_derive_pool(k0, old_dense.as<uint8_t>(), pool[0], ctx().stream));          // nothing of the key has changed yet
    }
    for (int v = 1; v < N; v++) {
        DeviceScope ds(v);
        ZKCHK(pool[v].alloc(96 * ph));
        ZKCHK(copy_between(pool[v].p, v, pool[0].p, 0, 96 * ph, ctx().stream));
        HIPCHK(hipStreamSynchronize(ctx().stream));
    }
    // ---- install: from the first commit on the key is only consistent once every device has succeeded
    {
        DeviceScope ds(0);
        for (auto& sl : g.slots) sl.reset();          // group slots refer to the shards' slots, which the install replaces
    }
    for (int v = 0; v < N; v++) {
        DeviceScope ds(v);
        const int rc = pin_install_derived(*g.sub[v], pool[v].as<uint8_t>(), ctx().stream);
        if (rc != ZK_OK) {
            g.broken = true;
            return rc;
        }
    }
    return ZK_OK;
}

}  // namespace zk

using namespace zk;
extern "C" {

int zk_pinocchio_pk_upload(uint32_t n, uint32_t m, const zk_csr* L, const zk_csr* R, const zk_csr* O, const uint8_t* mid,
                           const uint8_t* pk_g1, size_t pk_g1_points, const uint8_t* pk_g2, size_t pk_g2_points, uint64_t* handle) {
    if (!handle || !mid || !pk_g1 || !pk_g2) ZK_FAIL(ZK_ERR_ARG, "pinocchio pk_upload: null argument");
    ZKCHK(ensure_init());
    if (ctx_count() > 1) return pin_group_upload(n, m, L, R, O, mid, pk_g1, pk_g1_points, pk_g2, pk_g2_points, handle);
    std::unique_ptr<PinKey> key;
    ZKCHK(pin_key_build(key, n, m, L, R, O, mid, pk_g1, pk_g1_points, pk_g2, pk_g2_points, 0, 1, -1));
    *handle = g_pin_next++;
    g_pin[*handle] = std::move(key);
    return ZK_OK;
}
// The h pool of an uploaded key (si | v_all | w_all, or si alone: compact) rewritten for the VALUES of h: [lambda_t(s)]_1 derived from the powers si in the
// exponent (lagrange_derive.hip: the transposed interpolation over the points n .. 2n-2), [Z(s)]_1 = <si, Z> once, [1] = si[0], and for the compact form
// [s^(n-1)] = si[n-1].  Same proofs; the per-proof basis conversion disappears (only h ever needed coefficients: v(s), w(s) come from the per-variable pools).
int zk_pinocchio_pk_derive_lagrange(uint64_t handle) {
    if (PinGroup* g = pin_group_lookup(handle)) return pin_group_derive(*g);
    PinKey* kp;
    ZKCHK(pin_lookup(handle, &kp));
    PinKey& k = *kp;
    if (k.lagrange) return ZK_OK;
    ZKCHK(pin_check_idle(k, "zk_pinocchio_pk_derive_lagrange: a proof is in flight on this key"));
    HIPCHK(hipDeviceSynchronize());
    Ctx& c = ctx();
    const uint64_t ph_old = pin_h_points(k.n, k.m, false, k.compact);
    if (k.g1[5].n != ph_old) ZK_FAIL(ZK_ERR_ARG, "zk_pinocchio_pk_derive_lagrange: unexpected pool length");
    DevBuf pool, old_dense;
    ZKCHK(old_dense.alloc(96 * ph_old));
    ZKCHK(msm_bases_dense(k.g1[5], 0, ph_old, old_dense.p, c.stream));            // window 0 = the pool as uploaded, back in the dense affine format
    ZKCHK(pin_derive_pool(k, old_dense.as<uint8_t>(), pool, c.stream));
    return pin_install_derived(k, pool.as<uint8_t>(), c.stream);
}
int zk_pinocchio_pool_points(uint64_t handle, int pool, uint8_t* out, size_t capacity_points, size_t* count) {
    if (pool < 0 || pool >= PIN_G1 + PIN_G2) ZK_FAIL(ZK_ERR_ARG, "zk_pinocchio_pool_points: pool 0..5 (G1) or 6..7 (G2)");
    if (PinGroup* g = pin_group_lookup(handle)) {          // the handle holds the whole pools (its devices' slices are an internal matter)
        PinKey& k0 = *g->sub[0];
        const uint64_t total = pool < PIN_G1 ? k0.full1[pool] : k0.full2[pool - PIN_G1];
        if (count) *count = total;
        if (!out) return ZK_OK;
        if (capacity_points < total) ZK_FAIL(ZK_ERR_ARG, "zk_pinocchio_pool_points: buffer too small");
        ZKCHK(pin_group_check_idle(*g, "zk_pinocchio_pool_points: a proof is in flight on this key"));
        const size_t pb = pool < PIN_G1 ? 96 : 192;
        for (size_t v = 0; v < g->sub.size(); v++) {
            DeviceScope ds((int)v);
            PinKey& k = *g->sub[v];
            ZKCHK(pin_slice_points(k, pool, out + pb * (pool < PIN_G1 ? k.lo1[pool] : k.lo2[pool - PIN_G1])));
        }
        return ZK_OK;
    }
    PinKey* kp;
    ZKCHK(pin_lookup(handle, &kp));
    const MsmBases& b = pool < PIN_G1 ? kp->g1[pool] : kp->g2[pool - PIN_G1];
    if (count) *count = b.n;
    if (!out) return ZK_OK;
    if (capacity_points < b.n) ZK_FAIL(ZK_ERR_ARG, "zk_pinocchio_pool_points: buffer too small");
    return pin_slice_points(*kp, pool, out);
}
int zk_pinocchio_pk_free(uint64_t handle) {
    auto ig = g_pin_groups.find(handle);
    if (ig != g_pin_groups.end()) {
        (void)pin_group_sync(*ig->second);
        pin_group_destroy(*ig->second);
        g_pin_groups.erase(ig);
        return ZK_OK;
    }
    auto it = g_pin.find(handle);
    if (it == g_pin.end()) ZK_FAIL(ZK_ERR_HANDLE, "unknown Pinocchio key handle");
    (void)hipDeviceSynchronize();
    g_pin.erase(it);
    return ZK_OK;
}
int zk_pinocchio_reserve_slots(uint64_t handle, uint32_t count) {
    if (count > PIN_MAX_SLOTS) ZK_FAIL(ZK_ERR_ARG, "zk_pinocchio_reserve_slots: at most 15 slots");
    if (PinGroup* g = pin_group_lookup(handle)) {
        for (uint32_t i = 0; i < count; i++) {
            PinGroupSlot* gs;
            ZKCHK(pin_group_slot_get(*g, i, &gs));
        }
        return ZK_OK;
    }
    PinKey* kp;
    ZKCHK(pin_lookup(handle, &kp));
    for (uint32_t i = 0; i < count; i++) {
        PinSlot* sl;
        ZKCHK(pin_slot_get(*kp, i, &sl));
    }
    return ZK_OK;
}
int zk_pinocchio_set_witness(uint64_t handle, const uint8_t* sol) {
    if (!sol) ZK_FAIL(ZK_ERR_ARG, "zk_pinocchio_set_witness: null");
    if (PinGroup* g = pin_group_lookup(handle)) {
        ZKCHK(pin_group_check_idle(*g, "zk_pinocchio_set_witness: a proof is in flight on this key"));
        for (size_t v = 0; v < g->sub.size(); v++) {          // any device may own a proof's Fr stage
            DeviceScope ds((int)v);
            PinKey& k = *g->sub[v];
            HIPCHK(hipMemcpyAsync(k.wit_resident.p, sol, 32 * (size_t)k.m, hipMemcpyHostToDevice, ctx().stream));
            HIPCHK(hipStreamSynchronize(ctx().stream));
            k.have_witness = true;
        }
        return ZK_OK;
    }
    PinKey* kp;
    ZKCHK(pin_lookup(handle, &kp));
    Ctx& c = ctx();
    (void)hipDeviceSynchronize();      // no proof may still be reading the previous witness
    HIPCHK(hipMemcpyAsync(kp->wit_resident.p, sol, 32 * (size_t)kp->m, hipMemcpyHostToDevice, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    kp->have_witness = true;
    return ZK_OK;
}
// Enqueues one proof on the slot's stream and returns without waiting:
// Fr stage -> the eight scalar vectors -> eight MSMs -> affine bytes -> pinned host buffer.
int zk_pinocchio_prove_async(uint64_t handle, const uint8_t* sol, const uint8_t dv[32], const uint8_t dw[32], const uint8_t dy[32], uint32_t slot) {
    if (!dv || !dw || !dy) ZK_FAIL(ZK_ERR_ARG, "zk_pinocchio_prove_async: null argument");
    if (PinGroup* g = pin_group_lookup(handle)) return pin_group_prove_async(*g, sol, dv, dw, dy, slot);
    PinKey* kp;
    ZKCHK(pin_lookup(handle, &kp));
    PinKey& k = *kp;
    PinSlot* slp;
    ZKCHK(pin_slot_get(k, slot, &slp));
    PinSlot& sl = *slp;
    if (sl.busy) ZK_FAIL(ZK_ERR_ARG, "slot still has a proof in flight: call zk_pinocchio_prove_wait first");
    ZKCHK(pin_scalars_enqueue(k, sl, sol, dv, dw, dy, false));
    ZKCHK(pin_msms_enqueue(k, sl, false));
    HIPCHK(hipMemcpyAsync(sl.host, sl.out_dev.p, 960, hipMemcpyDeviceToHost, sl.st));
    HIPCHK(hipEventRecord(sl.done, sl.st));
    sl.busy = true;
    return ZK_OK;
}
int zk_pinocchio_prove_wait(uint64_t handle, uint32_t slot, uint8_t proof[960]) {
    if (!proof) ZK_FAIL(ZK_ERR_ARG, "zk_pinocchio_prove_wait: null proof");
    if (PinGroup* g = pin_group_lookup(handle)) return pin_group_prove_wait(*g, slot, proof);
    PinKey* kp;
    ZKCHK(pin_lookup(handle, &kp));
    if (slot >= PIN_MAX_SLOTS || !kp->slots[slot]) ZK_FAIL(ZK_ERR_ARG, "zk_pinocchio_prove_wait: slot never used");
    PinSlot& sl = *kp->slots[slot];
    if (!sl.busy) ZK_FAIL(ZK_ERR_ARG, "no proof in flight on this slot");
    HIPCHK(hipEventSynchronize(sl.done));
    sl.busy = false;
    int hf;
    memcpy(&hf, sl.host + 960, 4);
    if (hf & 2) ZK_FAIL(ZK_ERR_SCALAR_RANGE, "witness value >= r");
    if (hf & 1) ZK_FAIL(ZK_ERR_REMAINDER, "p mod Z != 0");
    memcpy(proof, sl.host, 960);
    return ZK_OK;
}
int zk_pinocchio_prove(uint64_t handle, const uint8_t* sol, const uint8_t dv[32], const uint8_t dw[32], const uint8_t dy[32],
                       uint8_t proof[960]) {
    if (!sol) ZK_FAIL(ZK_ERR_ARG, "zk_pinocchio_prove: null argument");
    ZKCHK(zk_pinocchio_prove_async(handle, sol, dv, dw, dy, 0));
    return zk_pinocchio_prove_wait(handle, 0, proof);
}
}
