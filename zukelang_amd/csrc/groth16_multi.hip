// Multi-device Groth16 keys: N GPUs of one node behind ONE handle of ONE process (what an OCaml host reaches through the ctypes
// shim -- `Groth16.Make(C).prove`, src/groth16/groth16.ml:235-237, is one call on one key and knows nothing of ranks).
//
// With a device list of N entries (zk_set_devices / zk_set_device_list), zk_groth16_pk_upload builds one SHARD of the key per entry -- the contiguous
// slices of both base pools that zk_groth16_shard_range cuts for equal work, each with its own window tables, slots and streams on its device -- and
// returns one handle.  A proof on slot t then runs
//   * the Fr stage (QAP.eval, QAP.ml:120-135) ONCE, on the slot's owner device (t mod N: proofs in flight rotate over the devices, as the ranks of
//     groth16.py's GroupProver do), which leaves the three scalar vectors over the full pools in that device's memory;
//   * on every device: a device-to-device copy of ITS slices of the three vectors out of the owner's memory (hipMemcpyPeerAsync over xGMI; a plain
//     device copy where two shards share a card), enqueued on the device's own slot stream behind an event of the owner's stream, then the three
//     multi-scalar products over the slice (groth16.ml:116-161) and a 768-byte copy of the raw XYZZ partial sums to the list's first device;
//   * on the first device: the sum of the N blocks per product (EC addition: exact, so the bytes do not depend on N or on the cuts), the affine
//     conversion and the copy of the proof to pinned host memory.
// Everything is enqueued by the calling thread and nothing synchronises before zk_groth16_prove_wait: the per-device streams are chained by events
// only, so up to 15 proofs stay in flight exactly as on one GPU.  No collective library is involved -- the exchange is N - 1 peer copies per vector,
// the pattern xGMI's point-to-point links serve directly; the one-process-per-GPU path (torch.distributed / RCCL: bench.py --gpus N) is unchanged.
// The derivation of the key's Lagrange form (zk_groth16_pk_derive_lagrange) gathers the tau-power pools on up to three devices, derives one of
// the three independent sets on each (one host thread per device), copies every set to every device and installs the shards.
#include "groth16_key.cuh"

#include <map>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <functional>
#include <thread>
#include <vector>

namespace zk {

struct GroupSlot {
    DevBuf parts, g1p, g2p, sum, out;          // on the list's first device: [device][768] landing area of the partial sums, the combine's scratch
    uint8_t* host = nullptr;                   // pinned: the proof (384 B)
    hipEvent_t ev_scal = nullptr;              // the owner's scalar vectors are complete
    std::vector<hipEvent_t> ev_part;           // per device: its 768-byte block has landed on the first device
    hipEvent_t done = nullptr;
    bool busy = false;
    int owner = 0;
    ~GroupSlot() {
        if (ev_scal) (void)hipEventDestroy(ev_scal);
        for (hipEvent_t e : ev_part)
            if (e) (void)hipEventDestroy(e);
        if (done) (void)hipEventDestroy(done);
        if (host) (void)hipHostFree(host);
    }
};

struct GroupKey {
    uint32_t n = 0, m = 0, n_mid = 0;
    uint64_t p1 = 0, p2 = 0;                   // the whole pools
    bool lagrange = false, broken = false;
    std::vector<std::unique_ptr<Groth16Key>> sub;      // sub[v]: the shard on virtual device v (rank v of world N)
    std::unique_ptr<GroupSlot> slots[MAX_SLOTS];
};

static std::map<uint64_t, std::unique_ptr<GroupKey>>& g_groups = *new std::map<uint64_t, std::unique_ptr<GroupKey>>;   // never destroyed (see ntt.hip)
static uint64_t g_group_next = 0x6000000001ull;

GroupKey* group_lookup(uint64_t handle) {
    auto it = g_groups.find(handle);
    return it == g_groups.end() ? nullptr : it->second.get();
}
uint64_t group_live_handles() { return g_groups.size(); }
static void group_destroy(GroupKey& g) {
    // slots first (events, pinned memory, buffers of the first device), then the shards, each with its own device current
    {
        DeviceScope ds(0);
        for (auto& sl : g.slots) sl.reset();
    }
    for (size_t v = 0; v < g.sub.size(); v++) {
        DeviceScope ds((int)v);
        g.sub[v].reset();
    }
}
void group_release_all() {
    for (auto& kv : g_groups) group_destroy(*kv.second);
    g_groups.clear();
}

// runs f(v) for every virtual device of the key, one host thread each (the set-up paths synchronise their streams internally); the worst code wins
template <class F> static int on_every_device(size_t count, F f) {
    std::vector<int> rc(count, ZK_OK);
    // One host thread per PHYSICAL device; list entries that share a card ("virtual devices") take their turns on that card's thread.  Besides being
    // all the parallelism one card has to give, this bounds what the set-up kernels ask of the card at once: k_subgroup_check and the window-table
    // builders carry 1-2 KiB of private memory per lane, i.e. > 1 GiB of scratch per QUEUE they run on, and a handful of them on different queues of ONE
    // device exhaust its scratch aperture (HSA_STATUS_ERROR_OUT_OF_RESOURCES: the runtime aborts the process).
    std::vector<std::vector<int>> groups;
    for (size_t v = 0; v < count; v++) {
        size_t gi = 0;
        while (gi < groups.size() && ctx_at(groups[gi][0]).device != ctx_at((int)v).device) gi++;
        if (gi == groups.size()) groups.emplace_back();
        groups[gi].push_back((int)v);
    }
    // f allocates (std::vector, make_unique, device buffers): an exception that left a thread function would be std::terminate for the host process --
    // it becomes a return code like every other failure
    auto run_group = [&rc, &f](const std::vector<int>& vs) {
        for (int v : vs) {
            DeviceScope ds(v);
            try {
                rc[v] = f(v);
            } catch (const std::bad_alloc&) {
                rc[v] = set_error(ZK_ERR_HIP, "out of host memory while setting a device of the key up", __FILE__, __LINE__);
            } catch (const std::exception& e) {
                rc[v] = set_error(ZK_ERR_HIP, e.what(), __FILE__, __LINE__);
            } catch (...) {
                rc[v] = set_error(ZK_ERR_HIP, "exception while setting a device of the key up", __FILE__, __LINE__);
            }
        }
    };
    // A host thread that cannot be started (the process is at its thread limit) must not take the process down: std::thread's constructor throws,
    // and a joinable thread destroyed during the unwinding would call std::terminate.  Whatever could not be started runs on the calling thread.
    std::vector<std::thread> th;
    size_t started = 1;          // group 0 runs on the calling thread
    for (size_t gi = 1; gi < groups.size(); gi++) {
        try {
            th.emplace_back(run_group, std::cref(groups[gi]));
            started = gi + 1;
        } catch (...) {
            break;
        }
    }
    run_group(groups[0]);
    for (size_t gi = started; gi < groups.size(); gi++) run_group(groups[gi]);
    for (auto& t : th) t.join();
    int worst = ZK_OK;
    for (int r : rc)
        if (r < worst) worst = r;
    return worst;
}

static void trace(const char* what) {
    static const bool on = getenv("ZK_TRACE") != nullptr;          // diagnostics: environment only
    if (on) {
        fprintf(stderr, "[zk multi] %s\n", what);
        fflush(stderr);
    }
}
static int sync_all(GroupKey& g) {
    for (size_t v = 0; v < g.sub.size(); v++) {
        DeviceScope ds((int)v);
        HIPCHK(hipDeviceSynchronize());
    }
    return ZK_OK;
}
static int check_idle(GroupKey& g, const char* who) {
    if (g.broken) ZK_FAIL(ZK_ERR_HIP, "multi-device key is inconsistent after a failed derivation: free it");
    for (auto& sl : g.slots)
        if (sl && sl->busy) ZK_FAIL(ZK_ERR_ARG, who);
    return ZK_OK;
}

int group_upload(uint32_t n, uint32_t m, const zk_csr* L, const zk_csr* R, const zk_csr* O, const uint8_t* mid, const uint8_t* pk_g1, size_t pk_g1_points,
                 const uint8_t* pk_g2, size_t pk_g2_points, bool lagrange, uint64_t* handle) {
    if (!handle || !mid || !pk_g1 || !pk_g2 || !L || !R || !O) ZK_FAIL(ZK_ERR_ARG, "pk_upload: null argument");
    const size_t N = (size_t)ctx_count();
    auto key = std::make_unique<GroupKey>();
    GroupKey& g = *key;
    g.sub.resize(N);
    // every shard checks its own slice of the key points and builds the Fr-stage tables (any device may own a proof's Fr stage)
    const int rc = on_every_device(N, [&](int v) {
        return groth16_key_build(g.sub[v], n, m, L, R, O, mid, pk_g1, pk_g1_points, pk_g2, pk_g2_points, (uint32_t)v, (uint32_t)N, lagrange, true);
    });
    if (rc != ZK_OK) {
        group_destroy(g);
        return rc;
    }
    g.n = n; g.m = m; g.n_mid = g.sub[0]->n_mid; g.p1 = g.sub[0]->p1; g.p2 = g.sub[0]->p2; g.lagrange = lagrange;
    *handle = g_group_next++;
    g_groups[*handle] = std::move(key);
    return ZK_OK;
}
int group_free(uint64_t handle) {
    auto it = g_groups.find(handle);
    if (it == g_groups.end()) ZK_FAIL(ZK_ERR_HANDLE, "unknown Groth16 key handle");
    (void)sync_all(*it->second);
    group_destroy(*it->second);
    g_groups.erase(it);
    return ZK_OK;
}

static int group_slot_get(GroupKey& g, uint32_t idx, GroupSlot** out) {
    if (idx >= MAX_SLOTS) ZK_FAIL(ZK_ERR_ARG, "slot index out of range (max 15 proofs in flight)");
    const size_t N = g.sub.size();
    if (!g.slots[idx]) {
        for (size_t v = 0; v < N; v++) {          // the slot's share on every device: Fr scratch, scalar vectors, workspaces, one stream
            DeviceScope ds((int)v);
            Slot* sl;
            ZKCHK(groth16_slot_get(*g.sub[v], idx, &sl));
        }
        DeviceScope ds(0);
        auto gs = std::make_unique<GroupSlot>();
        const size_t g1b = xyzz_bytes(CURVE_G1), g2b = xyzz_bytes(CURVE_G2), blk = 2 * g1b + g2b;
        static_assert(ZK_GROTH16_PARTIAL_BYTES == 768, "partial block = A | C | B raw XYZZ");
        ZKCHK(gs->parts.alloc(blk * N));
        ZKCHK(gs->g1p.alloc(2 * g1b * N));
        ZKCHK(gs->g2p.alloc(g2b * N));
        ZKCHK(gs->sum.alloc(blk));
        ZKCHK(gs->out.alloc(384));
        HIPCHK(hipHostMalloc((void**)&gs->host, 384, hipHostMallocDefault));
        HIPCHK(hipEventCreateWithFlags(&gs->done, hipEventDisableTiming));
        gs->ev_part.assign(N, nullptr);
        // the owner records ev_scal, device v records ev_part[v]: events live on the device whose stream records them
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
int group_reserve_slots(GroupKey& g, uint32_t count) {
    if (count > MAX_SLOTS) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_reserve_slots: at most 15 slots");
    if (g.broken) ZK_FAIL(ZK_ERR_HIP, "multi-device key is inconsistent after a failed derivation: free it");
    for (uint32_t i = 0; i < count; i++) {
        GroupSlot* gs;
        ZKCHK(group_slot_get(g, i, &gs));
    }
    return ZK_OK;
}
int group_set_witness(GroupKey& g, const uint8_t* sol) {
    if (!sol) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_set_witness: null");
    ZKCHK(check_idle(g, "zk_groth16_set_witness: a proof is in flight on this key"));
    for (size_t v = 0; v < g.sub.size(); v++) {          // any device may own a proof's Fr stage
        DeviceScope ds((int)v);
        Groth16Key& k = *g.sub[v];
        HIPCHK(hipMemcpyAsync(k.wit_resident.p, sol, 32 * (size_t)k.m, hipMemcpyHostToDevice, ctx().stream));
        HIPCHK(hipStreamSynchronize(ctx().stream));
        k.have_witness = true;
    }
    return ZK_OK;
}

int group_prove_async(GroupKey& g, const uint8_t* sol, const uint8_t* r, const uint8_t* s, uint32_t slot) {
    if (g.broken) ZK_FAIL(ZK_ERR_HIP, "multi-device key is inconsistent after a failed derivation: free it");
    GroupSlot* gsp;
    ZKCHK(group_slot_get(g, slot, &gsp));
    GroupSlot& gs = *gsp;
    if (gs.busy) ZK_FAIL(ZK_ERR_ARG, "slot still has a proof in flight: call the matching _wait first");
    const int N = (int)g.sub.size(), owner = gs.owner;
    Slot* so = g.sub[owner]->slots[slot].get();
    {   // ---- Fr stage on the owner: the three scalar vectors over the FULL pools, in the owner's slot buffers
        DeviceScope ds(owner);
        ZKCHK(groth16_scalars_enqueue(*g.sub[owner], *so, sol, r, s, so->scalA.p, so->scalC.p, so->scalB.p));
        HIPCHK(hipEventRecord(gs.ev_scal, so->s0));
    }
    // from here on the slot is in flight whatever happens: a failed enqueue below leaves work on some streams, and _wait drains it
    gs.busy = true;
    int rc = ZK_OK;
    for (int v = 0; v < N && rc == ZK_OK; v++) {
        DeviceScope ds(v);
        Groth16Key& k = *g.sub[v];
        Slot& sv = *k.slots[slot];
        auto body = [&]() -> int {
            char *a = sv.scalA.as<char>() + 32 * k.lo1, *c = sv.scalC.as<char>() + 32 * k.lo1, *b = sv.scalB.as<char>() + 32 * k.lo2;
            if (v != owner) {
                // this device's slices out of the owner's memory, on this device's stream, behind the owner's Fr stage.  The A vector is zero
                // beyond a | d1 | b1 | the tau basis (groth16.ml:128-134 touches no other key point): only that part of the slice travels, the
                // rest is cleared in place
                HIPCHK(hipStreamWaitEvent(sv.s0, gs.ev_scal, 0));
                const uint64_t nzA = k.p2 + 1, a_hi = k.hi1 < nzA ? k.hi1 : (k.lo1 > nzA ? k.lo1 : nzA);
                ZKCHK(copy_between(a, v, so->scalA.as<char>() + 32 * k.lo1, owner, 32 * (a_hi - k.lo1), sv.s0));
                if (k.hi1 > a_hi) HIPCHK(hipMemsetAsync(a + 32 * (a_hi - k.lo1), 0, 32 * (k.hi1 - a_hi), sv.s0));
                ZKCHK(copy_between(c, v, so->scalC.as<char>() + 32 * k.lo1, owner, 32 * (k.hi1 - k.lo1), sv.s0));
                ZKCHK(copy_between(b, v, so->scalB.as<char>() + 32 * k.lo2, owner, 32 * (k.hi2 - k.lo2), sv.s0));
            }
            ZKCHK(groth16_msms_enqueue(k, sv, a, c, b, true, 1));          // raw XYZZ partial sums A | C | B in sv.results, one stream
            ZKCHK(copy_between(gs.parts.as<char>() + ZK_GROTH16_PARTIAL_BYTES * v, 0, sv.results.p, v, ZK_GROTH16_PARTIAL_BYTES, sv.s0));
            HIPCHK(hipEventRecord(gs.ev_part[v], sv.s0));
            return ZK_OK;
        };
        rc = body();
    }
    {   // ---- first device: add the N blocks per product, convert, land the proof in pinned memory.  On ONE stream for all slots (the context's second
        // stream): the Fp2 column sum carries 3 KiB of private memory per lane, i.e. 1.6 GiB of scratch for every QUEUE it is dispatched on -- on the
        // slots' own streams a handful of proofs in flight exhausted the device's scratch aperture (HSA_STATUS_ERROR_OUT_OF_RESOURCES, the runtime aborts
        // the process).  A combine is ~50 us of work behind its N events; the slots' combines queue up in the order the proofs were enqueued.
        DeviceScope ds(0);
        hipStream_t cs = ctx().stream2;
        auto body = [&]() -> int {
            for (int v = 0; v < N; v++) HIPCHK(hipStreamWaitEvent(cs, gs.ev_part[v], 0));
            if (rc != ZK_OK) return rc;
            const size_t g1b = xyzz_bytes(CURVE_G1), g2b = xyzz_bytes(CURVE_G2), blk = 2 * g1b + g2b;
            HIPCHK(hipMemcpy2DAsync(gs.g1p.p, 2 * g1b, gs.parts.p, blk, 2 * g1b, N, hipMemcpyDeviceToDevice, cs));                      // [device][A, C]
            HIPCHK(hipMemcpy2DAsync(gs.g2p.p, g2b, gs.parts.as<char>() + 2 * g1b, blk, g2b, N, hipMemcpyDeviceToDevice, cs));        // [device][B]
            ZKCHK(xyzz_sum_columns(CURVE_G1, gs.sum.p, gs.g1p.p, N, 2, cs));
            ZKCHK(xyzz_sum_columns(CURVE_G2, gs.sum.as<char>() + 2 * g1b, gs.g2p.p, N, 1, cs));
            const uint32_t o1[2] = {0, 288}, o2[1] = {96};          // sums: A | C | B;  proof: a | b | c
            ZKCHK(proof_points_to_bytes_dev(gs.sum.p, 2, o1, gs.sum.as<char>() + 2 * g1b, 1, o2, gs.out.p, cs));
            HIPCHK(hipMemcpyAsync(gs.host, gs.out.p, 384, hipMemcpyDeviceToHost, cs));
            return ZK_OK;
        };
        const int rc0 = body();
        if (rc == ZK_OK) rc = rc0;
        (void)hipEventRecord(gs.done, cs);
    }
    if (rc != ZK_OK) {          // nothing of a half-enqueued proof may stay in flight behind the caller's back
        (void)sync_all(g);
        gs.busy = false;
    }
    return rc;
}
int group_prove_wait(GroupKey& g, uint32_t slot, uint8_t proof[384]) {
    if (slot >= MAX_SLOTS || !g.slots[slot]) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_prove_wait: slot never used");
    GroupSlot& gs = *g.slots[slot];
    if (!gs.busy) ZK_FAIL(ZK_ERR_ARG, "no proof in flight on this slot");
    {
        DeviceScope ds(0);
        HIPCHK(hipEventSynchronize(gs.done));          // behind every device's block, which is behind the owner's Fr stage and its flag copy
    }
    gs.busy = false;
    int hf;
    memcpy(&hf, g.sub[gs.owner]->slots[slot]->host + 384, 4);
    if (hf & 2) ZK_FAIL(ZK_ERR_SCALAR_RANGE, "witness value >= r");
    if (hf & 1) ZK_FAIL(ZK_ERR_REMAINDER, "p mod Z != 0");
    memcpy(proof, gs.host, 384);
    return ZK_OK;
}

int group_lagrange_pool_sizes(GroupKey& g, uint64_t* g1_points, uint64_t* g2_points) {
    if (g1_points) *g1_points = 3 + (uint64_t)g.n + (g.n - 1) + g.n_mid;
    if (g2_points) *g2_points = 2 + (uint64_t)g.n;
    return ZK_OK;
}
int group_pool_layout(GroupKey& g, uint64_t* p1, uint64_t* p2) {
    *p1 = g.p1;
    *p2 = g.p2;
    return ZK_OK;
}
int group_pool_points(GroupKey& g, int group, uint8_t* out, size_t capacity_points, size_t* count) {
    if (group != 1 && group != 2) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_pool_points: group must be 1 or 2");
    const uint64_t total = group == 1 ? g.p1 : g.p2;
    if (count) *count = total;
    if (!out) return ZK_OK;
    if (capacity_points < total) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_pool_points: buffer too small");
    ZKCHK(check_idle(g, "zk_groth16_pool_points: a proof is in flight on this key"));
    const size_t pb = group == 1 ? 96 : 192;
    for (size_t v = 0; v < g.sub.size(); v++) {          // the shards are the pool in order
        DeviceScope ds((int)v);
        Groth16Key& k = *g.sub[v];
        const uint64_t lo = group == 1 ? k.lo1 : k.lo2, hi = group == 1 ? k.hi1 : k.hi2;
        ZKCHK(single_pool_points(k, group, out + pb * lo, hi - lo, nullptr));
    }
    return ZK_OK;
}
int group_qap_eval(GroupKey& g, const uint8_t* sol, uint8_t* v_out, uint8_t* w_out, uint8_t* h_out) {
    ZKCHK(check_idle(g, "zk_groth16_qap_eval: a proof is in flight on this key"));
    DeviceScope ds(0);
    return single_qap_eval(*g.sub[0], sol, v_out, w_out, h_out);          // every shard holds the whole circuit and the Fr-stage tables
}

// zk_groth16_pk_derive_lagrange on a multi-device key: the three derived sets -- [l_i(tau)]_1, [l_i(tau)]_2, the h bases -- are
// independent, so each is derived on ONE device (devices 0, 1, 2 of the list; 0, 1, 0 on two), from a copy of the whole tau-power pools gathered out
// of the shards; every set then travels to every device and each device installs its shard of the Lagrange-form pools (own window tables).
int group_derive_lagrange(GroupKey& g) {
    if (g.lagrange) return ZK_OK;
    ZKCHK(check_idle(g, "zk_groth16_pk_derive_lagrange: a proof is in flight on this key"));
    trace("derive: sync");
    ZKCHK(sync_all(g));
    trace("derive: synced");
    const int N = (int)g.sub.size();
    const uint64_t n = g.n, p1o = g.p1, p2o = g.p2, p1n = 3 + n + (n - 1) + g.n_mid, p2n = 2 + n;
    int owner[3] = {0, N > 1 ? 1 : 0, N > 2 ? 2 : 0};
    std::vector<uint32_t> sets(N, 0);
    for (int sidx = 0; sidx < 3; sidx++) sets[owner[sidx]] |= 1u << sidx;
    std::vector<DevBuf> in1(N), in2(N), full1(N), full2(N);
    for (int v = 0; v < N; v++) {
        DeviceScope ds(v);
        ZKCHK(full1[v].alloc(96 * p1n));
        ZKCHK(full2[v].alloc(192 * p2n));
        if (sets[v]) {
            ZKCHK(in1[v].alloc(96 * p1o));
            ZKCHK(in2[v].alloc(192 * p2o));
        }
    }
    trace("derive: gather");
    // ---- gather: every shard's slice of the pools, dense affine, to every deriving device (window 0 of a shard's tables IS its slice in pool order)
    for (int v = 0; v < N; v++) {
        DeviceScope ds(v);
        Groth16Key& k = *g.sub[v];
        Ctx& c = ctx();
        DevBuf t1, t2;
        ZKCHK(t1.alloc(96 * (k.hi1 - k.lo1)));
        ZKCHK(t2.alloc(192 * (k.hi2 - k.lo2)));
        ZKCHK(msm_bases_dense(k.g1, 0, k.hi1 - k.lo1, t1.p, c.stream));
        ZKCHK(msm_bases_dense(k.g2, 0, k.hi2 - k.lo2, t2.p, c.stream));
        for (int d = 0; d < N; d++) {
            if (!sets[d]) continue;
            ZKCHK(copy_between(in1[d].as<char>() + 96 * k.lo1, d, t1.p, v, 96 * (k.hi1 - k.lo1), c.stream));
            ZKCHK(copy_between(in2[d].as<char>() + 192 * k.lo2, d, t2.p, v, 192 * (k.hi2 - k.lo2), c.stream));
        }
        HIPCHK(hipStreamSynchronize(c.stream));
    }
    trace("derive: derive sets");
    // ---- derive: one host thread per deriving device (the derivation synchronises its stream between phases)
    int rc = on_every_device((size_t)N, [&](int v) -> int {
        if (!sets[v]) return ZK_OK;
        Ctx& c = ctx();
        ZKCHK(groth16_derive_lagrange_pools(g.sub[v]->fr, in1[v].as<uint8_t>(), g.n_mid, in2[v].as<uint8_t>(), full1[v].as<uint8_t>(), full2[v].as<uint8_t>(), sets[v], c.stream));
        HIPCHK(hipStreamSynchronize(c.stream));
        return ZK_OK;
    });
    if (rc != ZK_OK) return rc;          // nothing of the key has changed yet
    trace("derive: copy sets");
    // ---- every set (and the copied parts a | d1 | b1, ltd_mid, b2 | d2, which every derivation writes) to every other device
    struct Region { int src; int pool; uint64_t lo, hi; };
    const Region regions[6] = {{owner[0], 1, 3, 3 + n}, {owner[1], 2, 2, 2 + n}, {owner[2], 1, 3 + n, 3 + n + (n - 1)},
                               {owner[0], 1, 0, 3}, {owner[0], 1, 3 + n + (n - 1), p1n}, {owner[0], 2, 0, 2}};
    for (int v = 0; v < N; v++) {
        DeviceScope ds(v);
        Ctx& c = ctx();
        for (const Region& rg : regions) {
            if (rg.src == v || rg.hi <= rg.lo) continue;
            const size_t pb = rg.pool == 1 ? 96 : 192;
            char* dst = (rg.pool == 1 ? full1[v] : full2[v]).as<char>();
            const char* src = (rg.pool == 1 ? full1[rg.src] : full2[rg.src]).as<char>();
            ZKCHK(copy_between(dst + pb * rg.lo, v, src + pb * rg.lo, rg.src, pb * (rg.hi - rg.lo), c.stream));
        }
        HIPCHK(hipStreamSynchronize(c.stream));
    }
    for (int v = 0; v < N; v++) { in1[v].release(); in2[v].release(); }
    trace("derive: install");
    // ---- install: each device builds the window tables of ITS slice of the new pools and flips its Fr stage.  From the first commit on the key
    // is only consistent once every device has succeeded.
    for (auto& sl : g.slots) {          // group slots refer to the shards' slots, which the install replaces
        DeviceScope ds(0);
        sl.reset();
    }
    rc = on_every_device((size_t)N, [&](int v) { return groth16_install_lagrange(*g.sub[v], full1[v].p, full2[v].p, (uint32_t)v, (uint32_t)N); });
    if (rc != ZK_OK) {
        g.broken = true;
        return rc;
    }
    trace("derive: done");
    g.lagrange = true;
    g.p1 = p1n;
    g.p2 = p2n;
    return ZK_OK;
}

}  // namespace zk
