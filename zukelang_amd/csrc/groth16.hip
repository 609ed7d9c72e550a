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
#include "groth16_key.cuh"

#include <map>
#include <memory>
#include <stdlib.h>
#include <string.h>

namespace zk {

static std::map<uint64_t, std::unique_ptr<Groth16Key>>& g_keys = *new std::map<uint64_t, std::unique_ptr<Groth16Key>>;   // never destroyed (see ntt.hip)
static uint64_t g_next_handle = 1;
static void handles_release() { group_release_all(); g_keys.clear(); }
static CleanupRegistrar g_key_cleanup(handles_release);

static inline dim3 g1d(uint64_t n, unsigned t = 256) { return dim3((unsigned)((n + t - 1) / t)); }

// Caller-owned scalar vectors (zk_groth16_msm_partial_async): bit 1 of *flag reports a value >= r.  The digits of a product are only right for
// canonical scalars (the top window absorbs one carry of a value < 2^255; folded widths take r - s), and nothing upstream has looked at these.
__global__ void k_check_canonical3(const uint32_t* __restrict__ a, const uint32_t* __restrict__ c, uint64_t n1, const uint32_t* __restrict__ b, uint64_t n2, int* flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool bad = false;
    if (i < n1) bad = !fe_is_canonical(fe_load<FrParams>(a + 8 * i)) || !fe_is_canonical(fe_load<FrParams>(c + 8 * i));
    if (i < n2) bad = bad || !fe_is_canonical(fe_load<FrParams>(b + 8 * i));
    if (bad) atomicOr(flag, 2);
}
// scalar vectors for the three MSMs (canonical form).  rs = r | s (canonical, device).
__global__ void k_groth16_scalars(uint32_t* __restrict__ scalA, uint32_t* __restrict__ scalC, uint32_t* __restrict__ scalB,
                                  const uint32_t* __restrict__ v, const uint32_t* __restrict__ w, const uint32_t* __restrict__ h,
                                  const uint32_t* __restrict__ wit_mont, const uint32_t* __restrict__ mid_idx,
                                  const uint32_t* __restrict__ rs, uint32_t n, uint32_t n_mid, uint32_t nt) {
    // nt = points of the tau basis: n + 2 powers tau^0..tau^(n+1) (groth16.ml:73) or n Lagrange points (row f4)
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t p1 = 3 + (uint64_t)nt + (n - 1) + n_mid;
    if (i >= p1) return;
    const Fr r = fe_to_mont(fe_load<FrParams>(rs)), s = fe_to_mont(fe_load<FrParams>(rs + 8));
    const Fr zero = fe_zero<FrParams>(), one = fe_one<FrParams>();
    Fr a = zero, c = zero;
    const uint64_t ti = 3, tz = ti + nt, lt = tz + (n - 1);
    if (i == 0) { a = one; c = s; }                                               // alpha
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
    const uint64_t p2 = 2 + (uint64_t)nt;
    if (i < p2) {
        Fr b = zero;
        if (i == 0) b = one;                                                      // beta_2
        else if (i == 1) b = s;                                                   // delta_2
        else if (i - 2 < n) b = fe_load<FrParams>(w + 8 * (i - 2));
        fe_store<FrParams>(scalB + 8 * i, fe_from_mont(b));
    }
}


// Contiguous slice [lo, hi) of a pool of `points` base points for `rank` of `world`, cut for equal WORK: the first `heavy` points of the G1
// pool (a | d1 | b1 | the tau basis) carry two products of a proof, A and C (groth16.ml:128-134 vs :147-160), every other point one, so
// the cuts sit at equal shares of points + heavy (with uniform cuts the first ranks of an 8-way split would do twice the G1 work of the
// others).  heavy = 0: uniform.  The rule the host side mirrors (zukelang_amd/groth16.py: shard_bounds).
void groth16_shard_range(uint64_t points, uint64_t heavy, uint32_t rank, uint32_t world, uint64_t* lo, uint64_t* hi) {
    if (heavy > points) heavy = points;
    const uint64_t total = points + heavy;
    auto cut = [&](uint64_t g) -> uint64_t {
        const uint64_t t = (uint64_t)((unsigned __int128)total * g / world);
        return t <= 2 * heavy ? t / 2 : t - heavy;
    };
    *lo = cut(rank);
    *hi = cut((uint64_t)rank + 1);
}

static int key_lookup(uint64_t handle, Groth16Key** out) {
    auto it = g_keys.find(handle);
    if (it == g_keys.end()) {
        if (group_lookup(handle)) ZK_FAIL(ZK_ERR_ARG, "multi-device key: this entry point serves the shards of the one-process-per-GPU path");
        ZK_FAIL(ZK_ERR_HANDLE, "unknown Groth16 key handle");
    }
    *out = it->second.get();
    return ZK_OK;
}

int groth16_slot_get(Groth16Key& k, uint32_t idx, Slot** out) {
    if (idx >= MAX_SLOTS) ZK_FAIL(ZK_ERR_ARG, "slot index out of range (max 15 proofs in flight)");
    if (!k.slots[idx]) {
        auto sl = std::make_unique<Slot>();
        ZKCHK(frstage_scratch_alloc(k.fr, sl->fs));
        ZKCHK(sl->scalA.alloc(32 * k.p1));
        ZKCHK(sl->scalC.alloc(32 * k.p1));
        ZKCHK(sl->scalB.alloc(32 * k.p2));
        ZKCHK(sl->wit_raw.alloc(32 * (size_t)k.m));
        ZKCHK(sl->rs.alloc(64));
        ZKCHK(sl->results.alloc(2 * xyzz_bytes(CURVE_G1) + xyzz_bytes(CURVE_G2)));
        ZKCHK(sl->out_dev.alloc(384));
        // A and C go out as ONE accumulate launch over the G1 pool: A carries the non-zero prefix a | d1 | b1 | tau basis only (this rank's part of it),
        // C the whole slice; every scalar has a digit in (nearly) every window
        const uint64_t pre = k.p2 + 1, a_pts = k.lo1 >= pre ? 0 : (k.hi1 < pre ? k.hi1 : pre) - k.lo1;
        const uint64_t e1 = (a_pts + k.g1.n) * k.g1.nw;
        ZKCHK(msm_workspace_alloc(sl->wsA, k.g1, 0, e1));
        ZKCHK(msm_workspace_alloc(sl->wsC, k.g1, 0, e1));
        ZKCHK(msm_workspace_alloc(sl->wsB, k.g2, 0, k.g2.n * k.g2.nw));
        HIPCHK(hipStreamCreateWithFlags(&sl->s0, hipStreamNonBlocking));
        // One stream per proof by default: the chip runs at most 16 hardware queues side by side and
        // falls off a cliff beyond (24 streams x 1 ms of 1-workgroup kernels: 72 ms, scripts/proto/concurrency.hip),
        // so queues are better spent on MORE PROOFS in flight than on the three MSMs of one proof
        // (2^16: 3.9 ms/proof with 8 x 3 streams, 2.9 ms with 15 x 1).  ZK_SLOT_STREAMS=3 restores the fork.
        // Slot 0 -- the slot of the synchronous zk_groth16_prove -- keeps two more streams and forks onto them while it is the ONLY
        // proof in flight: the G2 product's latency-bound reduction chain then runs beside the G1 products instead of in front of
        // them (single-proof latency; with other proofs in flight it stays on one stream like every other slot).
        const char* ss = ZK_ENV("ZK_SLOT_STREAMS");
        const bool want3 = !k.one_stream_slots && (ss ? atoi(ss) >= 3 : idx == 0);
        if (ZK_ENV("ZK_SERIAL_STREAMS") || !want3) {
            sl->s1 = sl->s2 = sl->s0;
            sl->serial = true;
        } else {
            HIPCHK(hipStreamCreateWithFlags(&sl->s1, hipStreamNonBlocking));
            HIPCHK(hipStreamCreateWithFlags(&sl->s2, hipStreamNonBlocking));
        }
        HIPCHK(hipEventCreateWithFlags(&sl->fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&sl->join1, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&sl->join2, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&sl->done, hipEventDisableTiming));
        HIPCHK(hipHostMalloc((void**)&sl->host, 512 + ZK_GROTH16_PARTIAL_BYTES, hipHostMallocDefault));
        sl->host_partial = sl->host + 512;
        k.slots[idx] = std::move(sl);
    }
    *out = k.slots[idx].get();
    return ZK_OK;
}

// The key's window width (see upload): one per key, 20 bits from 2^21 points in the G1 pool; 17 bits from 2^20 points (2^19 gates on one GPU, the
// shards of config 4's 2^22 key on eight): 17 divides 255, and with the digits taken from min(s, r - s) (msm.cuh: msm_windows) that is 15 windows
// instead of 16 for twice the buckets -- measured +3.3 % at 2^19 (where 20 bits lose 2 %); at 2^18 +0.8 ... 1.8 % for 5 % more latency of a lone
// proof, at 2^16 -6 %: both stay at 16 (profiles/r04_window_17_ab.txt).
// Without the subgroup check there are no folded digits (msm.cuh: msm_fold), and 17 bits without them are 16 windows over twice the buckets of 16 bits.
static uint32_t key_window(uint64_t g1_points, bool in_subgroup) {
    if (::zk::opt("ZK_MSM_WINDOW")) return 0;          // bases_setup reads it
    if (g1_points >= ((uint64_t)1 << 21)) return 20;
    if (g1_points >= ((uint64_t)1 << 20)) return in_subgroup ? 17 : 16;
    return msm_auto_window(g1_points, true);
}
int groth16_key_build(std::unique_ptr<Groth16Key>& out, uint32_t n, uint32_t m, const zk_csr* L, const zk_csr* R, const zk_csr* O, const uint8_t* mid,
                      const uint8_t* pk_g1, size_t pk_g1_points, const uint8_t* pk_g2, size_t pk_g2_points, uint32_t rank, uint32_t world, bool lagrange, bool shard_of_group) {
    if (!mid || !pk_g1 || !pk_g2 || !L || !R || !O) ZK_FAIL(ZK_ERR_ARG, "pk_upload: null argument");
    if (world == 0 || rank >= world) ZK_FAIL(ZK_ERR_ARG, "pk_upload: bad rank / world");
    ZKCHK(ensure_init());
    Ctx& c = ctx();
    auto key = std::make_unique<Groth16Key>();
    Groth16Key& k = *key;
    k.vdev = c.vdev;
    k.one_stream_slots = shard_of_group;
    k.n = n; k.m = m; k.rank = rank; k.world = world;
    std::vector<uint32_t> mids;
    for (uint32_t i = 0; i < m; i++)
        if (mid[i]) mids.push_back(i);
    k.n_mid = (uint32_t)mids.size();
    if (n < 1) ZK_FAIL(ZK_ERR_ARG, "pk_upload: need at least 1 constraint");
    k.lagrange = lagrange;
    const uint64_t nt = lagrange ? n : (uint64_t)n + 2;
    k.p1 = 3 + nt + (n - 1) + k.n_mid;
    k.p2 = 2 + nt;
    if (pk_g1_points != k.p1) ZK_FAIL(ZK_ERR_DOMAIN, "pk_upload: G1 key length != 3 + (n+2 | n) + (n-1) + |mids|");
    if (pk_g2_points != k.p2) ZK_FAIL(ZK_ERR_DOMAIN, "pk_upload: G2 key length != 2 + (n+2 | n)");
    ZKCHK(frstage_init(k.fr, n, m, L, R, O, c.stream));
    if (lagrange) ZKCHK(frstage_init_lagrange(k.fr, c.stream));
    // contiguous pool slices per rank
    groth16_shard_range(k.p1, 3 + nt, rank, world, &k.lo1, &k.hi1);
    groth16_shard_range(k.p2, 0, rank, world, &k.lo2, &k.hi2);
    if (k.hi1 == k.lo1 || k.hi2 == k.lo2) ZK_FAIL(ZK_ERR_ARG, "pk_upload: more ranks than key points");
    // ONE window width per key (both pools: their bucket reductions then go out as one chain of launches).  From 2^21 points
    // in this rank's G1 pool (n >= ~2^19.6 on one GPU) 20-bit windows: 13 instead of 16 digits per scalar pay for the 16x
    // bucket count and the two-level sort (measured, profiles/r02_window_sweep_*.json: +7.5 % at 2^20, +12 % at 2^22, -2 % at
    // 2^19, -16 % at 2^18); round 4: 17-bit windows, fifteen of them, from 2^20 points (key_window above).  ZK_MSM_WINDOW overrides (config 3's sweep).
    // key points are checked like the reference checks them on the way in (of_bytes_exn: encoding, curve, prime-order subgroup); ZK_KEY_SUBGROUP_CHECK=0
    // (zk_set_option "key_subgroup_check") skips the subgroup part for keys that were checked before (it is [r] P = O per point: 0.3 s at 2^20
    // constraints, 1.4 s at 2^22).  Read per key (set-up path).  A key uploaded without the check never gets folded windows: msm.cuh.
    const char* e_chk = ::zk::opt("ZK_KEY_SUBGROUP_CHECK");
    const bool chk = !(e_chk && atoi(e_chk) == 0);
    const uint32_t cw = key_window(k.hi1 - k.lo1, chk);
    ZKCHK(msm_bases_from_bytes(k.g1, CURVE_G1, pk_g1 + 96 * k.lo1, k.hi1 - k.lo1, cw, true, c.stream, chk));
    ZKCHK(msm_bases_from_bytes(k.g2, CURVE_G2, pk_g2 + 192 * k.lo2, k.hi2 - k.lo2, cw, true, c.stream, chk));
    ZKCHK(k.mid_idx.alloc(4 * (size_t)(k.n_mid ? k.n_mid : 1)));
    if (k.n_mid) HIPCHK(hipMemcpyAsync(k.mid_idx.p, mids.data(), 4 * (size_t)k.n_mid, hipMemcpyHostToDevice, c.stream));
    ZKCHK(k.wit_resident.alloc(32 * (size_t)m));
    HIPCHK(hipStreamSynchronize(c.stream));
    Slot* sl;
    ZKCHK(groth16_slot_get(k, 0, &sl));
    out = std::move(key);
    return ZK_OK;
}
// A key uploaded WHOLE through zk_groth16_pk_upload[_lagrange] while the device list has several entries becomes a multi-device key: one shard per
// entry behind one handle (groth16_multi.hip).  The explicit shard upload (rank of world: one process per GPU) always builds on the list's first device.
static int upload(uint32_t n, uint32_t m, const zk_csr* L, const zk_csr* R, const zk_csr* O, const uint8_t* mid,
                  const uint8_t* pk_g1, size_t pk_g1_points, const uint8_t* pk_g2, size_t pk_g2_points, uint32_t rank,
                  uint32_t world, uint64_t* handle, bool lagrange = false, bool whole = false) {
    if (!handle) ZK_FAIL(ZK_ERR_ARG, "pk_upload: null argument");
    ZKCHK(ensure_init());
    if (whole && ctx_count() > 1) return group_upload(n, m, L, R, O, mid, pk_g1, pk_g1_points, pk_g2, pk_g2_points, lagrange, handle);
    std::unique_ptr<Groth16Key> key;
    ZKCHK(groth16_key_build(key, n, m, L, R, O, mid, pk_g1, pk_g1_points, pk_g2, pk_g2_points, rank, world, lagrange));
    *handle = g_next_handle++;
    g_keys[*handle] = std::move(key);
    return ZK_OK;
}
uint64_t pinocchio_live_handles();          // pinocchio.hip
uint64_t live_key_handles() { return g_keys.size() + group_live_handles() + pinocchio_live_handles(); }

// First half of a proof: Fr stage -> the three scalar vectors (canonical Fr, FULL pool lengths p1, p1, p2)
// written to dA / dC / dB (device memory; the slot's own buffers in the single-call path).
// The host half of scalars_enqueue: witness (when handed over as a host buffer) and r | s into the slot's pinned staging memory.
int groth16_stage_inputs(Groth16Key& k, Slot& sl, const uint8_t* sol, const uint8_t* r, const uint8_t* s) {
    if (sl.busy) ZK_FAIL(ZK_ERR_ARG, "slot still has a proof in flight: call the matching _wait first");
    if (sol) {
        if (!sl.wit_pinned) HIPCHK(hipHostMalloc((void**)&sl.wit_pinned, 32 * (size_t)k.m, hipHostMallocDefault));
        memcpy(sl.wit_pinned, sol, 32 * (size_t)k.m);
    } else if (!k.have_witness) ZK_FAIL(ZK_ERR_ARG, "no witness: pass sol or call zk_groth16_set_witness first");
    memcpy(sl.host + 392, r, 32);
    memcpy(sl.host + 424, s, 32);
    return ZK_OK;
}
// staged = true: stage_inputs has run (graph capture / replay: the stream operations below read the pinned staging memory when they EXECUTE)
int groth16_scalars_enqueue(Groth16Key& k, Slot& sl, const uint8_t* sol, const uint8_t* r, const uint8_t* s, void* dA, void* dC, void* dB, bool staged) {
    if (!staged) ZKCHK(groth16_stage_inputs(k, sl, sol, r, s));
    const void* wit = k.wit_resident.p;
    if (sol) {
        HIPCHK(hipMemcpyAsync(sl.wit_raw.p, sl.wit_pinned, 32 * (size_t)k.m, hipMemcpyHostToDevice, sl.s0));
        wit = sl.wit_raw.p;
    }
    HIPCHK(hipMemcpyAsync(sl.rs.p, sl.host + 392, 64, hipMemcpyHostToDevice, sl.s0));
    const uint32_t *v, *w;
    if (k.lagrange) {          // values a = L w, b = R w and h on the shifted points: no basis conversion
        ZKCHK(frstage_eval_lagrange(k.fr, sl.fs, wit, sl.s0));
        v = sl.fs.abc.as<uint32_t>();
        w = v + 8 * (uint64_t)k.n;
    } else {
        ZKCHK(frstage_eval(k.fr, sl.fs, wit, sl.s0));
        v = sl.fs.d.as<uint32_t>();
        w = v + 8 * (uint64_t)k.fr.n2;
    }
    hipLaunchKernelGGL(k_groth16_scalars, g1d(k.p1), dim3(256), 0, sl.s0, (uint32_t*)dA, (uint32_t*)dC, (uint32_t*)dB, v, w,
                       (const uint32_t*)sl.fs.h.as<uint32_t>(), (const uint32_t*)sl.fs.wit.as<uint32_t>(), (const uint32_t*)k.mid_idx.as<uint32_t>(),
                       (const uint32_t*)sl.rs.as<uint32_t>(), k.n, k.n_mid, k.lagrange ? k.n : k.n + 2);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(sl.host + 384, sl.fs.flag.p, 4, hipMemcpyDeviceToHost, sl.s0));
    return ZK_OK;
}
// Second half: the three MSMs over this rank's slice of the pools.  dA / dC / dB point at the scalars of
// that slice ((hi1-lo1), (hi1-lo1), (hi2-lo2) elements).  {C on s0, B on s1, A on s2} -> affine bytes (or raw
// XYZZ partial sums) -> pinned host buffer.
// fork onto the slot's extra streams only while no other proof is in flight on this key (and never when ZK_SLOT_STREAMS forces it)
static bool runs_serial(Groth16Key& k, Slot& sl) {
    bool serial = sl.serial;
    if (ctx().profiling >= 2) serial = true;      // the per-family event timers want un-overlapped launches (bench.py's one-proof-in-flight pass)
    if (!serial && !ZK_ENV("ZK_SLOT_STREAMS"))
        for (uint32_t i = 0; i < MAX_SLOTS; i++)
            if (k.slots[i] && k.slots[i].get() != &sl && k.slots[i]->busy) serial = true;
    return serial;
}
int groth16_msms_enqueue(Groth16Key& k, Slot& sl, const void* dA, const void* dC, const void* dB, bool raw, int force_serial) {
    const bool serial = force_serial < 0 ? runs_serial(k, sl) : force_serial != 0;
    char* res = sl.results.as<char>();
    char* out = sl.out_dev.as<char>();
    const size_t g1b = xyzz_bytes(CURVE_G1);
    if (!serial) {
        HIPCHK(hipEventRecord(sl.fork, sl.s0));
        HIPCHK(hipStreamWaitEvent(sl.s1, sl.fork, 0));
        HIPCHK(hipStreamWaitEvent(sl.s2, sl.fork, 0));
    }
    if (serial) {
        // one stream: sort + accumulate (A and C together: same bases), then ALL reductions as one chain of launches
        ZKCHK(msm_sort_accumulate(k.g2, sl.wsB, dB, sl.s0));
        MsmWorkspace* ws[2] = {&sl.wsA, &sl.wsC};
        const void* scal[2] = {dA, dC};
        void* outs[2] = {res, res + g1b};
        ZKCHK(msm_sort_accumulate_many(k.g1, ws, scal, 2, sl.s0));
        MsmWorkspace* ws2[1] = {&sl.wsB};
        void* outs2[1] = {res + 2 * g1b};
        if (k.g1.c == k.g2.c && k.g1.nw == k.g2.nw) {
            ZKCHK(msm_reduce_mixed(&k.g1, ws, outs, 2, &k.g2, ws2, outs2, 1, sl.s0));      // all three reductions: one chain of launches
        } else {                                                                            // tiny keys: the pools got different windows
            ZKCHK(msm_reduce(k.g2, ws2, outs2, 1, sl.s0));
            ZKCHK(msm_reduce(k.g1, ws, outs, 2, sl.s0));
        }
        if (!raw) {
            const uint32_t o1[2] = {0, 288}, o2[1] = {96};          // results: A | C | B;  proof: a | b | c
            ZKCHK(proof_points_to_bytes_dev(res, 2, o1, res + 2 * g1b, 1, o2, out, sl.s0));
        }
    } else {
        // B (G2, the longest chain) first
        ZKCHK(msm_run(k.g2, sl.wsB, dB, res + 2 * g1b, sl.s1));
        if (!raw) ZKCHK(points_xyzz_to_bytes_dev(CURVE_G2, res + 2 * g1b, 1, out + 96, sl.s1));
        HIPCHK(hipEventRecord(sl.join1, sl.s1));
        ZKCHK(msm_run(k.g1, sl.wsC, dC, res + g1b, sl.s0));
        if (!raw) ZKCHK(points_xyzz_to_bytes_dev(CURVE_G1, res + g1b, 1, out + 288, sl.s0));
        ZKCHK(msm_run(k.g1, sl.wsA, dA, res, sl.s2));
        if (!raw) ZKCHK(points_xyzz_to_bytes_dev(CURVE_G1, res, 1, out, sl.s2));
        HIPCHK(hipEventRecord(sl.join2, sl.s2));
        HIPCHK(hipStreamWaitEvent(sl.s0, sl.join1, 0));
        HIPCHK(hipStreamWaitEvent(sl.s0, sl.join2, 0));
    }
    if (!raw) HIPCHK(hipMemcpyAsync(sl.host, sl.out_dev.p, 384, hipMemcpyDeviceToHost, sl.s0));
    return ZK_OK;
}
// Enqueues one whole proof on the slot's streams and returns without waiting.
static int prove_enqueue(Groth16Key& k, Slot& sl, const uint8_t* sol, const uint8_t* r, const uint8_t* s, bool raw) {
    const char* eg = ZK_FORM_ENV("ZK_GRAPH");              // cached; per proof only under ZK_TEST_FORMS=1 (tests switch it inside one process)
    const bool graphs = eg && atoi(eg) != 0;
    if (graphs && !ctx().profiling) {
        // The captured launches freeze the kernel forms that were selected at capture time.  Outside test mode the form switches cannot change (they are
        // cached); under ZK_TEST_FORMS=1 a slot's graphs are dropped whenever the switches differ from the ones they were captured under.
        if (forms_live()) {
            uint64_t h = 1469598103934665603ull;
            for (const char* name : {"ZK_TAIL_SLOTS", "ZK_TAIL_FIXUP_SLOTS", "ZK_FIXUP_BY_CHUNK", "ZK_ACC_G1_GLDS", "ZK_ACC_G1_MMADD", "ZK_ACC_G2_INLINE", "ZK_RED_WAVES", "ZK_SORT_FINE_STAGED", "ZK_SORT_COARSE_STAGED", "ZK_DS_WIDE_GROUP", "ZK_FR_RNS"}) {
                const char* v = ::zk::opt(name);
                for (const char* q = v ? v : "\x01"; *q; q++) h = (h ^ (uint8_t)*q) * 1099511628211ull;
                h = (h ^ 0xff) * 1099511628211ull;
            }
            if (h != sl.graph_forms) {
                for (hipGraphExec_t& g : sl.graph)
                    if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
                sl.graph_forms = h;
            }
        }
        // One hipGraph per slot and proof shape, captured from the very calls below the first time and replayed afterwards: a proof is ~60 stream
        // operations whose arguments never change (the slot owns every buffer; witness, r, s travel through its pinned staging memory).
        ZKCHK(groth16_stage_inputs(k, sl, sol, r, s));
        const bool serial = runs_serial(k, sl);
        hipGraphExec_t& ge = sl.graph[(serial ? 1 : 0) | (raw ? 2 : 0) | (sol ? 4 : 0)];
        if (!ge) {
            HIPCHK(hipStreamBeginCapture(sl.s0, hipStreamCaptureModeThreadLocal));
            int rc = groth16_scalars_enqueue(k, sl, sol, r, s, sl.scalA.p, sl.scalC.p, sl.scalB.p, true);
            if (rc == ZK_OK)
                rc = groth16_msms_enqueue(k, sl, sl.scalA.as<char>() + 32 * k.lo1, sl.scalC.as<char>() + 32 * k.lo1, sl.scalB.as<char>() + 32 * k.lo2, raw, serial ? 1 : 0);
            hipGraph_t g = nullptr;
            const hipError_t e = hipStreamEndCapture(sl.s0, &g);
            if (rc != ZK_OK) {
                if (g) (void)hipGraphDestroy(g);
                return rc;
            }
            HIPCHK(e);
            const hipError_t ei = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            HIPCHK(ei);
        }
        HIPCHK(hipGraphLaunch(ge, sl.s0));
        HIPCHK(hipEventRecord(sl.done, sl.s0));
        sl.busy = true;
        return ZK_OK;
    }
    ZKCHK(groth16_scalars_enqueue(k, sl, sol, r, s, sl.scalA.p, sl.scalC.p, sl.scalB.p));
    ZKCHK(groth16_msms_enqueue(k, sl, sl.scalA.as<char>() + 32 * k.lo1, sl.scalC.as<char>() + 32 * k.lo1, sl.scalB.as<char>() + 32 * k.lo2, raw));
    HIPCHK(hipEventRecord(sl.done, sl.s0));
    sl.busy = true;
    return ZK_OK;
}
int groth16_prove_finish(Slot& sl) {
    if (!sl.busy) ZK_FAIL(ZK_ERR_ARG, "no proof in flight on this slot");
    HIPCHK(hipEventSynchronize(sl.done));
    sl.busy = false;
    int hf;
    memcpy(&hf, sl.host + 384, 4);
    if (hf & 2) ZK_FAIL(ZK_ERR_SCALAR_RANGE, "an Fr input (witness value, or a scalar of a caller-owned vector) is >= r");
    if (hf & 1) ZK_FAIL(ZK_ERR_REMAINDER, "p mod Z != 0");
    return ZK_OK;
}

}  // namespace zk

using namespace zk;
extern "C" {

int zk_groth16_pk_upload(uint32_t n, uint32_t m, const zk_csr* L, const zk_csr* R, const zk_csr* O, const uint8_t* mid,
                         const uint8_t* pk_g1, size_t pk_g1_points, const uint8_t* pk_g2, size_t pk_g2_points, uint64_t* handle) {
    return upload(n, m, L, R, O, mid, pk_g1, pk_g1_points, pk_g2, pk_g2_points, 0, 1, handle, false, true);
}
int zk_groth16_pk_upload_sharded(uint32_t n, uint32_t m, const zk_csr* L, const zk_csr* R, const zk_csr* O, const uint8_t* mid,
                                 const uint8_t* pk_g1, size_t pk_g1_points, const uint8_t* pk_g2, size_t pk_g2_points,
                                 uint32_t rank, uint32_t world, uint64_t* handle) {
    return upload(n, m, L, R, O, mid, pk_g1, pk_g1_points, pk_g2, pk_g2_points, rank, world, handle);
}
int zk_groth16_pk_upload_lagrange(uint32_t n, uint32_t m, const zk_csr* L, const zk_csr* R, const zk_csr* O, const uint8_t* mid,
                                  const uint8_t* pk_g1, size_t pk_g1_points, const uint8_t* pk_g2, size_t pk_g2_points, uint64_t* handle) {
    return upload(n, m, L, R, O, mid, pk_g1, pk_g1_points, pk_g2, pk_g2_points, 0, 1, handle, true, true);
}
// ---- derivation in two halves, so that the ranks of a node can share it (one set per rank, broadcast, install) -- and the single-call form on top
static int derive_precheck(Groth16Key& k, const char* who) {
    if (k.world != 1) ZK_FAIL(ZK_ERR_ARG, "derive: sharded keys hold only a slice of the powers");
    for (uint32_t i = 0; i < MAX_SLOTS; i++)
        if (k.slots[i] && k.slots[i]->busy) ZK_FAIL(ZK_ERR_ARG, who);
    HIPCHK(hipDeviceSynchronize());
    return ZK_OK;
}
// the Lagrange-form pool sizes of a key with n constraints: g1' = 3 + n + (n - 1) + n_mid points, g2' = 2 + n points
int zk_groth16_lagrange_pool_sizes(uint64_t handle, uint64_t* g1_points, uint64_t* g2_points) {
    if (GroupKey* g = group_lookup(handle)) return group_lagrange_pool_sizes(*g, g1_points, g2_points);
    Groth16Key* k;
    ZKCHK(key_lookup(handle, &k));
    if (g1_points) *g1_points = 3 + (uint64_t)k->n + (k->n - 1) + k->n_mid;
    if (g2_points) *g2_points = 2 + (uint64_t)k->n;
    return ZK_OK;
}
int zk_groth16_pk_derive_lagrange_sets(uint64_t handle, uint32_t sets, void* d_g1_out, void* d_g2_out) {
    Groth16Key* kp;
    ZKCHK(key_lookup(handle, &kp));
    Groth16Key& k = *kp;
    if (!d_g1_out || !d_g2_out || (sets & ~7u)) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_pk_derive_lagrange_sets: bad argument");
    if (k.lagrange) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_pk_derive_lagrange_sets: the key already holds Lagrange-form pools");
    ZKCHK(derive_precheck(k, "zk_groth16_pk_derive_lagrange_sets: a proof is in flight on this key"));
    Ctx& c = ctx();
    DevBuf d1, d2;
    // window 0 of the resident tables IS the key as uploaded (pool order): back into the dense affine format the derivation reads
    ZKCHK(d1.alloc(96 * k.g1.n));
    ZKCHK(d2.alloc(192 * k.g2.n));
    ZKCHK(msm_bases_dense(k.g1, 0, k.g1.n, d1.p, c.stream));
    ZKCHK(msm_bases_dense(k.g2, 0, k.g2.n, d2.p, c.stream));
    ZKCHK(groth16_derive_lagrange_pools(k.fr, d1.as<uint8_t>(), k.n_mid, d2.as<uint8_t>(), (uint8_t*)d_g1_out, (uint8_t*)d_g2_out, sets, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    return ZK_OK;
}
int zk_groth16_pk_install_lagrange(uint64_t handle, const void* d_g1, const void* d_g2, uint32_t rank, uint32_t world) {
    if (group_lookup(handle)) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_pk_install_lagrange: multi-device keys derive and install in one call (zk_groth16_pk_derive_lagrange)");
    Groth16Key* kp;
    ZKCHK(key_lookup(handle, &kp));
    Groth16Key& k = *kp;
    if (!d_g1 || !d_g2 || world == 0 || rank >= world) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_pk_install_lagrange: bad argument");
    if (k.lagrange) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_pk_install_lagrange: the key already holds Lagrange-form pools");
    ZKCHK(derive_precheck(k, "zk_groth16_pk_install_lagrange: a proof is in flight on this key"));
    return groth16_install_lagrange(k, d_g1, d_g2, rank, world);
}
}  // extern "C"
namespace zk {
int groth16_install_lagrange(Groth16Key& k, const void* d_g1, const void* d_g2, uint32_t rank, uint32_t world) {
    Ctx& c = ctx();
    // Everything is built into TEMPORARIES first; the handle changes only after every allocation and launch has succeeded (an OOM while the new
    // window tables are built -- they are 13-16x the pools -- leaves the key exactly as it was: tau-power pools, tau-power Fr stage, lagrange = false).
    MsmBases g1, g2;
    const uint64_t p1n = 3 + (uint64_t)k.n + (k.n - 1) + k.n_mid, p2n = 2 + (uint64_t)k.n;
    uint64_t lo1, hi1, lo2, hi2;
    groth16_shard_range(p1n, p2n + 1, rank, world, &lo1, &hi1);        // p2n + 1 = 3 + the n Lagrange points: the A prefix counts twice
    groth16_shard_range(p2n, 0, rank, world, &lo2, &hi2);
    if (hi1 == lo1 || hi2 == lo2) ZK_FAIL(ZK_ERR_ARG, "install_lagrange: more ranks than key points");
    // the derived pools are linear combinations of the key's own points: in the subgroup exactly when those were checked (msm.cuh: msm_fold)
    const uint32_t cw = key_window(hi1 - lo1, k.g1.in_subgroup);
    ZKCHK(msm_bases_from_device_affine(g1, CURVE_G1, (const uint8_t*)d_g1 + 96 * lo1, hi1 - lo1, cw, true, c.stream, k.g1.in_subgroup));
    ZKCHK(msm_bases_from_device_affine(g2, CURVE_G2, (const uint8_t*)d_g2 + 192 * lo2, hi2 - lo2, cw, true, c.stream, k.g2.in_subgroup));
    ZKCHK(frstage_init_lagrange(k.fr, c.stream));        // ADDS the Lagrange tables to the Fr stage; the tau-power path keeps working until `lagrange` flips
    HIPCHK(hipStreamSynchronize(c.stream));              // the caller's pools have been read
    HIPCHK(hipGetLastError());
    // ---- commit (nothing below can fail before the key is consistent again)
    for (uint32_t i = 0; i < MAX_SLOTS; i++) k.slots[i].reset();          // workspaces are sized for the old pools
    k.g1 = std::move(g1);
    k.g2 = std::move(g2);
    k.lagrange = true;
    k.rank = rank; k.world = world;
    k.p1 = p1n; k.p2 = p2n; k.lo1 = lo1; k.hi1 = hi1; k.lo2 = lo2; k.hi2 = hi2;
    Slot* sl;
    ZKCHK(groth16_slot_get(k, 0, &sl));          // on failure the key is valid in Lagrange form; the slot is created at the next use
    return ZK_OK;
}
}  // namespace zk
extern "C" {
int zk_groth16_pk_derive_lagrange(uint64_t handle) {
    if (GroupKey* g = group_lookup(handle)) return group_derive_lagrange(*g);
    Groth16Key* kp;
    ZKCHK(key_lookup(handle, &kp));
    Groth16Key& k = *kp;
    if (k.lagrange) return ZK_OK;
    DevBuf n1, n2;
    ZKCHK(n1.alloc(96 * (3 + (uint64_t)k.n + (k.n - 1) + k.n_mid)));
    ZKCHK(n2.alloc(192 * (2 + (uint64_t)k.n)));
    ZKCHK(zk_groth16_pk_derive_lagrange_sets(handle, 7, n1.p, n2.p));
    return zk_groth16_pk_install_lagrange(handle, n1.p, n2.p, 0, 1);
}
int zk_groth16_pk_shard(uint64_t handle, uint32_t rank, uint32_t world) {
    Groth16Key* kp;
    ZKCHK(key_lookup(handle, &kp));
    Groth16Key& k = *kp;
    if (k.world != 1) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_pk_shard: the key is sharded already");
    if (world == 0 || rank >= world) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_pk_shard: bad rank / world");
    if (world == 1) return ZK_OK;
    for (uint32_t i = 0; i < MAX_SLOTS; i++)
        if (k.slots[i] && k.slots[i]->busy) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_pk_shard: a proof is in flight on this key");
    HIPCHK(hipDeviceSynchronize());
    Ctx& c = ctx();
    uint64_t lo1, hi1, lo2, hi2;
    groth16_shard_range(k.p1, k.p2 + 1, rank, world, &lo1, &hi1);        // p2 + 1 = 3 + the tau basis
    groth16_shard_range(k.p2, 0, rank, world, &lo2, &hi2);
    if (hi1 == lo1 || hi2 == lo2) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_pk_shard: more ranks than key points");
    for (uint32_t i = 0; i < MAX_SLOTS; i++) k.slots[i].reset();
    // window 0 of the resident tables is the pool in order: the rank keeps its contiguous slice and builds its own window tables
    MsmBases g1, g2;
    const uint32_t cw = key_window(hi1 - lo1, k.g1.in_subgroup);
    DevBuf d1, d2;
    ZKCHK(d1.alloc(96 * (hi1 - lo1)));
    ZKCHK(d2.alloc(192 * (hi2 - lo2)));
    ZKCHK(msm_bases_dense(k.g1, lo1, hi1 - lo1, d1.p, c.stream));
    ZKCHK(msm_bases_dense(k.g2, lo2, hi2 - lo2, d2.p, c.stream));
    ZKCHK(msm_bases_from_device_affine(g1, CURVE_G1, d1.p, hi1 - lo1, cw, true, c.stream, k.g1.in_subgroup));      // a slice of checked points is checked
    ZKCHK(msm_bases_from_device_affine(g2, CURVE_G2, d2.p, hi2 - lo2, cw, true, c.stream, k.g2.in_subgroup));
    HIPCHK(hipStreamSynchronize(c.stream));
    k.g1 = std::move(g1);
    k.g2 = std::move(g2);
    k.rank = rank; k.world = world;
    k.lo1 = lo1; k.hi1 = hi1; k.lo2 = lo2; k.hi2 = hi2;
    Slot* sl;
    ZKCHK(groth16_slot_get(k, 0, &sl));
    return ZK_OK;
}
int zk_groth16_pool_points(uint64_t handle, int group, uint8_t* out, size_t capacity_points, size_t* count) {
    if (GroupKey* g = group_lookup(handle)) return group_pool_points(*g, group, out, capacity_points, count);
    Groth16Key* k;
    ZKCHK(key_lookup(handle, &k));
    return single_pool_points(*k, group, out, capacity_points, count);
}
}  // extern "C"
namespace zk {
// the key's resident pool slice (all of it for a key held whole) as uncompressed bytes, in pool order
int single_pool_points(Groth16Key& key, int group, uint8_t* out, size_t capacity_points, size_t* count) {
    Groth16Key* k = &key;
    if (group != 1 && group != 2) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_pool_points: group must be 1 or 2");
    const MsmBases& b = group == 1 ? k->g1 : k->g2;
    if (count) *count = b.n;
    if (!out) return ZK_OK;
    if (capacity_points < b.n) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_pool_points: buffer too small");
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
}  // namespace zk
extern "C" {
int zk_groth16_pk_free(uint64_t handle) {
    if (group_lookup(handle)) return group_free(handle);
    auto it = g_keys.find(handle);
    if (it == g_keys.end()) ZK_FAIL(ZK_ERR_HANDLE, "unknown Groth16 key handle");
    (void)hipDeviceSynchronize();
    g_keys.erase(it);
    return ZK_OK;
}
int zk_groth16_reserve_slots(uint64_t handle, uint32_t count) {
    if (GroupKey* g = group_lookup(handle)) return group_reserve_slots(*g, count);
    Groth16Key* k;
    ZKCHK(key_lookup(handle, &k));
    if (count > MAX_SLOTS) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_reserve_slots: at most 15 slots");
    for (uint32_t i = 0; i < count; i++) {
        Slot* sl;
        ZKCHK(groth16_slot_get(*k, i, &sl));
    }
    return ZK_OK;
}
int zk_groth16_prove_async(uint64_t handle, const uint8_t* sol, const uint8_t r[32], const uint8_t s[32], uint32_t slot) {
    if (GroupKey* g = group_lookup(handle)) {
        if (!r || !s) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_prove_async: null argument");
        return group_prove_async(*g, sol, r, s, slot);
    }
    Groth16Key* k;
    ZKCHK(key_lookup(handle, &k));
    if (!r || !s) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_prove_async: null argument");
    if (k->world != 1) ZK_FAIL(ZK_ERR_ARG, "key is sharded; use prove_partial + combine");
    Slot* sl;
    ZKCHK(groth16_slot_get(*k, slot, &sl));
    return prove_enqueue(*k, *sl, sol, r, s, false);
}
int zk_groth16_prove_wait(uint64_t handle, uint32_t slot, uint8_t proof[384]) {
    if (GroupKey* g = group_lookup(handle)) {
        if (!proof) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_prove_wait: null proof");
        return group_prove_wait(*g, slot, proof);
    }
    Groth16Key* k;
    ZKCHK(key_lookup(handle, &k));
    if (!proof) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_prove_wait: null proof");
    if (slot >= MAX_SLOTS || !k->slots[slot]) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_prove_wait: slot never used");
    Slot& sl = *k->slots[slot];
    ZKCHK(groth16_prove_finish(sl));
    memcpy(proof, sl.host, 384);
    return ZK_OK;
}
int zk_groth16_prove(uint64_t handle, const uint8_t* sol, const uint8_t r[32], const uint8_t s[32], uint8_t proof[384]) {
    ZKCHK(zk_groth16_prove_async(handle, sol, r, s, 0));
    return zk_groth16_prove_wait(handle, 0, proof);
}
int zk_groth16_set_witness(uint64_t handle, const uint8_t* sol) {
    if (GroupKey* g = group_lookup(handle)) return group_set_witness(*g, sol);
    Groth16Key* k;
    ZKCHK(key_lookup(handle, &k));
    if (!sol) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_set_witness: null");
    Ctx& c = ctx();
    (void)hipDeviceSynchronize();      // no proof may still be reading the previous witness
    HIPCHK(hipMemcpyAsync(k->wit_resident.p, sol, 32 * (size_t)k->m, hipMemcpyHostToDevice, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    k->have_witness = true;
    return ZK_OK;
}
int zk_groth16_prove_partial_async(uint64_t handle, const uint8_t* sol, const uint8_t r[32], const uint8_t s[32], uint32_t slot) {
    Groth16Key* k;
    ZKCHK(key_lookup(handle, &k));
    if (!r || !s) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_prove_partial_async: null argument");
    Slot* sl;
    ZKCHK(groth16_slot_get(*k, slot, &sl));
    ZKCHK(prove_enqueue(*k, *sl, sol, r, s, true));      // raw XYZZ partial sums: no affine conversion
    // the 768-byte partial block follows the proof on the slot's stream into pinned memory
    HIPCHK(hipMemcpyAsync(sl->host_partial, sl->results.p, ZK_GROTH16_PARTIAL_BYTES, hipMemcpyDeviceToHost, sl->s0));
    HIPCHK(hipEventRecord(sl->done, sl->s0));
    return ZK_OK;
}
int zk_groth16_prove_partial_wait(uint64_t handle, uint32_t slot, uint8_t partial[ZK_GROTH16_PARTIAL_BYTES]) {
    Groth16Key* k;
    ZKCHK(key_lookup(handle, &k));
    if (!partial) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_prove_partial_wait: null");
    if (slot >= MAX_SLOTS || !k->slots[slot]) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_prove_partial_wait: slot never used");
    Slot& sl = *k->slots[slot];
    ZKCHK(groth16_prove_finish(sl));
    memcpy(partial, sl.host_partial, ZK_GROTH16_PARTIAL_BYTES);
    return ZK_OK;
}
int zk_groth16_prove_partial_wait_device(uint64_t handle, uint32_t slot, void* d_partial) {
    Groth16Key* k;
    ZKCHK(key_lookup(handle, &k));
    if (!d_partial) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_prove_partial_wait_device: null");
    if (slot >= MAX_SLOTS || !k->slots[slot]) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_prove_partial_wait_device: slot never used");
    Slot& sl = *k->slots[slot];
    const int rc = groth16_prove_finish(sl);                     // the slot is free again whatever the Fr stage's flags say
    // results = A | C | B raw XYZZ partial sums, device to device: the block never visits the host
    HIPCHK(hipMemcpyAsync(d_partial, sl.results.p, ZK_GROTH16_PARTIAL_BYTES, hipMemcpyDeviceToDevice, sl.s0));
    HIPCHK(hipStreamSynchronize(sl.s0));
    return rc;
}
int zk_groth16_prove_partial(uint64_t handle, const uint8_t* sol, const uint8_t r[32], const uint8_t s[32],
                             uint8_t partial[ZK_GROTH16_PARTIAL_BYTES]) {
    ZKCHK(zk_groth16_prove_partial_async(handle, sol, r, s, 0));
    return zk_groth16_prove_partial_wait(handle, 0, partial);
}
// ---- distributed Fr stage: the two halves of a proof as separate calls on caller-owned device buffers
int zk_groth16_shard_range(uint64_t points, uint64_t heavy_prefix, uint32_t rank, uint32_t world, uint64_t* lo, uint64_t* hi) {
    if (!lo || !hi || world == 0 || rank >= world) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_shard_range: bad argument");
    groth16_shard_range(points, heavy_prefix, rank, world, lo, hi);
    return ZK_OK;
}
int zk_groth16_pool_layout(uint64_t handle, uint64_t* p1, uint64_t* p2, uint64_t* lo1, uint64_t* hi1, uint64_t* lo2, uint64_t* hi2) {
    if (GroupKey* g = group_lookup(handle)) {          // the handle holds the whole pools (its devices' slices are an internal matter)
        uint64_t a = 0, b = 0;
        ZKCHK(group_pool_layout(*g, &a, &b));
        if (p1) *p1 = a;
        if (p2) *p2 = b;
        if (lo1) *lo1 = 0;
        if (hi1) *hi1 = a;
        if (lo2) *lo2 = 0;
        if (hi2) *hi2 = b;
        return ZK_OK;
    }
    Groth16Key* k;
    ZKCHK(key_lookup(handle, &k));
    if (p1) *p1 = k->p1;
    if (p2) *p2 = k->p2;
    if (lo1) *lo1 = k->lo1;
    if (hi1) *hi1 = k->hi1;
    if (lo2) *lo2 = k->lo2;
    if (hi2) *hi2 = k->hi2;
    return ZK_OK;
}
int zk_groth16_scalars_async(uint64_t handle, const uint8_t* sol, const uint8_t r[32], const uint8_t s[32], uint32_t slot,
                             void* d_scal_a, void* d_scal_c, void* d_scal_b) {
    Groth16Key* k;
    ZKCHK(key_lookup(handle, &k));
    if (!r || !s || !d_scal_a || !d_scal_c || !d_scal_b) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_scalars_async: null argument");
    Slot* sl;
    ZKCHK(groth16_slot_get(*k, slot, &sl));
    ZKCHK(groth16_scalars_enqueue(*k, *sl, sol, r, s, d_scal_a, d_scal_c, d_scal_b));
    HIPCHK(hipEventRecord(sl->done, sl->s0));
    sl->busy = true;
    return ZK_OK;
}
int zk_groth16_scalars_wait(uint64_t handle, uint32_t slot) {
    Groth16Key* k;
    ZKCHK(key_lookup(handle, &k));
    if (slot >= MAX_SLOTS || !k->slots[slot]) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_scalars_wait: slot never used");
    return groth16_prove_finish(*k->slots[slot]);
}
int zk_groth16_msm_partial_async(uint64_t handle, uint32_t slot, const void* d_scal_a_slice, const void* d_scal_c_slice, const void* d_scal_b_slice) {
    Groth16Key* k;
    ZKCHK(key_lookup(handle, &k));
    if (!d_scal_a_slice || !d_scal_c_slice || !d_scal_b_slice) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_msm_partial_async: null argument");
    Slot* sl;
    ZKCHK(groth16_slot_get(*k, slot, &sl));
    if (sl->busy) ZK_FAIL(ZK_ERR_ARG, "slot still has work in flight: call the matching _wait first");
    // the remainder flag belongs to the Fr stage, which ran elsewhere (zk_groth16_scalars_wait); the RANGE of these vectors is checked here: they are
    // caller-owned device memory nobody has looked at (ZK_ERR_SCALAR_RANGE from the matching _wait)
    memset(sl->host + 384, 0, 4);
    {
        const uint64_t n1 = k->hi1 - k->lo1, n2 = k->hi2 - k->lo2;
        HIPCHK(hipMemsetAsync(sl->fs.flag.p, 0, 4, sl->s0));
        hipLaunchKernelGGL(k_check_canonical3, g1d(n1 > n2 ? n1 : n2), dim3(256), 0, sl->s0, (const uint32_t*)d_scal_a_slice, (const uint32_t*)d_scal_c_slice, n1,
                           (const uint32_t*)d_scal_b_slice, n2, sl->fs.flag.as<int>());
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(sl->host + 384, sl->fs.flag.p, 4, hipMemcpyDeviceToHost, sl->s0));
    }
    ZKCHK(groth16_msms_enqueue(*k, *sl, d_scal_a_slice, d_scal_c_slice, d_scal_b_slice, true));
    HIPCHK(hipMemcpyAsync(sl->host_partial, sl->results.p, ZK_GROTH16_PARTIAL_BYTES, hipMemcpyDeviceToHost, sl->s0));
    HIPCHK(hipEventRecord(sl->done, sl->s0));
    sl->busy = true;
    return ZK_OK;
}
int zk_device_malloc(size_t bytes, void** dptr) {
    ZKCHK(ensure_init());
    if (!dptr) ZK_FAIL(ZK_ERR_ARG, "zk_device_malloc: null");
    HIPCHK(hipMalloc(dptr, bytes ? bytes : 1));
    return ZK_OK;
}
int zk_device_free(void* dptr) {
    if (dptr) HIPCHK(hipFree(dptr));
    return ZK_OK;
}
int zk_device_memcpy(void* dst, const void* src, size_t bytes) {
    ZKCHK(ensure_init());
    HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDefault));
    return ZK_OK;
}
// Persistent buffers of zk_groth16_combine: hipMalloc / hipFree per call would synchronise the whole device
// (hipFree does) and stall every proof in flight on the other streams.
struct CombineBufs {
    DevBuf parts, g1p, g2p, sum, out;
    uint8_t* host = nullptr;      // pinned, 384 B
    uint32_t world = 0;
};
static CombineBufs& g_combine = *new CombineBufs;          // never destroyed (see ntt.hip); released by zk_shutdown
static void combine_release() {
    g_combine.parts.release(); g_combine.g1p.release(); g_combine.g2p.release(); g_combine.sum.release(); g_combine.out.release();
    if (g_combine.host) { (void)hipHostFree(g_combine.host); g_combine.host = nullptr; }
    g_combine.world = 0;
}
static CleanupRegistrar g_combine_cleanup(combine_release);
// partials: `world` blocks of 768 B, block j at partials + j * stride, in HOST memory (from_device = false) or DEVICE memory
static int combine_impl(const uint8_t* partials, size_t stride, uint32_t world, uint8_t proof[384], bool from_device) {
    if (!partials || !proof || world == 0 || stride < ZK_GROTH16_PARTIAL_BYTES) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_combine: bad argument");
    ZKCHK(ensure_init());
    Ctx& c = ctx();
    const size_t g1b = xyzz_bytes(CURVE_G1), g2b = xyzz_bytes(CURVE_G2), blk = 2 * g1b + g2b;
    CombineBufs& B = g_combine;
    if (B.world < world) {
        ZKCHK(B.parts.alloc(blk * world));
        ZKCHK(B.g1p.alloc(2 * g1b * world));
        ZKCHK(B.g2p.alloc(g2b * world));
        ZKCHK(B.sum.alloc(blk));
        ZKCHK(B.out.alloc(384));
        if (!B.host) HIPCHK(hipHostMalloc((void**)&B.host, 384, hipHostMallocDefault));
        B.world = world;
    }
    const char* src = (const char*)partials;
    if (!from_device) {
        if (stride == blk) HIPCHK(hipMemcpyAsync(B.parts.p, partials, blk * world, hipMemcpyHostToDevice, c.stream));
        else
            for (uint32_t j = 0; j < world; j++) HIPCHK(hipMemcpyAsync(B.parts.as<char>() + blk * j, partials + stride * j, blk, hipMemcpyHostToDevice, c.stream));
        src = B.parts.as<char>();
        stride = blk;
    }
    for (uint32_t j = 0; j < world; j++) {     // rank-major blocks -> [rank][A, C] and [rank][B]
        HIPCHK(hipMemcpyAsync(B.g1p.as<char>() + 2 * g1b * j, src + stride * j, 2 * g1b, hipMemcpyDeviceToDevice, c.stream));
        HIPCHK(hipMemcpyAsync(B.g2p.as<char>() + g2b * j, src + stride * j + 2 * g1b, g2b, hipMemcpyDeviceToDevice, c.stream));
    }
    ZKCHK(xyzz_sum_columns(CURVE_G1, B.sum.p, B.g1p.p, world, 2, c.stream));
    ZKCHK(xyzz_sum_columns(CURVE_G2, B.sum.as<char>() + 2 * g1b, B.g2p.p, world, 1, c.stream));
    // A, C (G1) -> proof[0..96), proof[288..384); B (G2) -> proof[96..288): one launch, the three inversions side by side
    const uint32_t o1[2] = {0, 288}, o2[1] = {96};
    ZKCHK(proof_points_to_bytes_dev(B.sum.p, 2, o1, B.sum.as<char>() + 2 * g1b, 1, o2, B.out.p, c.stream));
    HIPCHK(hipMemcpyAsync(B.host, B.out.p, 384, hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    memcpy(proof, B.host, 384);
    return ZK_OK;
}
int zk_groth16_combine(const uint8_t* partials, uint32_t world, uint8_t proof[384]) {
    return combine_impl(partials, ZK_GROTH16_PARTIAL_BYTES, world, proof, false);
}
int zk_groth16_combine_device(const void* d_partials, size_t stride_bytes, uint32_t world, uint8_t proof[384]) {
    return combine_impl((const uint8_t*)d_partials, stride_bytes, world, proof, true);
}
int zk_groth16_qap_eval(uint64_t handle, const uint8_t* sol, uint8_t* v_out, uint8_t* w_out, uint8_t* h_out) {
    if (GroupKey* g = group_lookup(handle)) return group_qap_eval(*g, sol, v_out, w_out, h_out);
    Groth16Key* k;
    ZKCHK(key_lookup(handle, &k));
    return single_qap_eval(*k, sol, v_out, w_out, h_out);
}
}
namespace zk {
int single_qap_eval(Groth16Key& key, const uint8_t* sol, uint8_t* v_out, uint8_t* w_out, uint8_t* h_out) {
    Groth16Key* k = &key;
    Slot* sl;
    ZKCHK(groth16_slot_get(*k, 0, &sl));
    if (sl->busy) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_qap_eval: slot 0 has a proof in flight");
    const void* wit = k->wit_resident.p;
    if (sol) {
        HIPCHK(hipMemcpyAsync(sl->wit_raw.p, sol, 32 * (size_t)k->m, hipMemcpyHostToDevice, sl->s0));
        wit = sl->wit_raw.p;
    } else if (!k->have_witness) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_qap_eval: no witness");
    ZKCHK(frstage_eval(k->fr, sl->fs, wit, sl->s0));
    int hf = 0;
    HIPCHK(hipMemcpyAsync(&hf, sl->fs.flag.p, 4, hipMemcpyDeviceToHost, sl->s0));
    HIPCHK(hipStreamSynchronize(sl->s0));
    if (hf & 2) ZK_FAIL(ZK_ERR_SCALAR_RANGE, "witness value >= r");
    if (hf & 1) ZK_FAIL(ZK_ERR_REMAINDER, "p mod Z != 0");
    DevBuf tmp;
    ZKCHK(tmp.alloc(32 * (size_t)k->n));
    const char* v = sl->fs.d.as<char>();
    const void* srcs[3] = {v, v + 32 * (size_t)k->fr.n2, sl->fs.h.p};
    uint8_t* outs[3] = {v_out, w_out, h_out};
    size_t cnt[3] = {k->n, k->n, (size_t)k->n - 1};
    for (int i = 0; i < 3; i++) {
        if (!outs[i] || cnt[i] == 0) continue;          // a single gate has no h coefficient
        ZKCHK(fr_from_mont(tmp.p, srcs[i], cnt[i], sl->s0));
        HIPCHK(hipMemcpyAsync(outs[i], tmp.p, 32 * cnt[i], hipMemcpyDeviceToHost, sl->s0));
        HIPCHK(hipStreamSynchronize(sl->s0));
    }
    return ZK_OK;
}
}  // namespace zk
