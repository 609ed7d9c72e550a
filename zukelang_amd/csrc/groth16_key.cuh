// Internal interface of the Groth16 prover (groth16.hip) that the multi-device form (groth16_multi.hip) builds on: the key and slot records and the
// two halves of a proof -- Fr stage -> scalar vectors, multi-scalar products over a slice of the pools -- as enqueue-only functions.
#pragma once
#include "frstage.cuh"
#include "msm.cuh"

#include <memory>

namespace zk {

// Everything one proof in flight owns: scratch of the Fr stage, the three scalar vectors, one MSM
// workspace and one stream per product, pinned host landing buffers.  Several slots let the shallow
// single-wave tails of one proof (bucket reduction, affine conversion) run under the bulk kernels
// of the next.
struct Slot {
    FrScratch fs;
    DevBuf scalA, scalC, scalB, wit_raw, rs, results, out_dev;
    MsmWorkspace wsA, wsC, wsB;
    hipStream_t s0 = nullptr, s1 = nullptr, s2 = nullptr;     // C + Fr stage | B (G2) | A
    hipEvent_t fork = nullptr, join1 = nullptr, join2 = nullptr, done = nullptr;
    uint8_t* host = nullptr;            // pinned: proof 384 B | flag 4 B | r 32 B | s 32 B
    uint8_t* host_partial = nullptr;    // pinned: 768 B of raw partial sums (sharded mode)
    uint8_t* wit_pinned = nullptr;      // pinned staging copy of a witness handed over as a host buffer (allocated at first use): the caller's
                                        // buffer is read before the call returns, as the header promises, whatever memory it lives in
    bool busy = false, serial = false;
    // ZK_GRAPH=1: the whole proof of this slot captured ONCE per shape (streams forked or not | raw partial sums | witness from the host) and replayed
    hipGraphExec_t graph[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    uint64_t graph_forms = 0;           // hash of the kernel-form switches the graphs were captured under (test mode only)
    ~Slot() {
        for (hipGraphExec_t g : graph)
            if (g) (void)hipGraphExecDestroy(g);
        if (s0) { (void)hipStreamDestroy(s0); if (!serial) { (void)hipStreamDestroy(s1); (void)hipStreamDestroy(s2); } }
        if (fork) { (void)hipEventDestroy(fork); (void)hipEventDestroy(join1); (void)hipEventDestroy(join2); (void)hipEventDestroy(done); }
        if (host) (void)hipHostFree(host);
        if (wit_pinned) (void)hipHostFree(wit_pinned);
    }
};
static constexpr uint32_t MAX_SLOTS = 15;      // + the context stream = the 16 hardware queues the chip runs side by side

struct Groth16Key {
    uint32_t n = 0, m = 0, n_mid = 0;
    uint32_t rank = 0, world = 1;
    uint64_t p1 = 0, p2 = 0;            // full pool sizes (points)
    uint64_t lo1 = 0, hi1 = 0;          // this rank's slice of the G1 pool
    uint64_t lo2 = 0, hi2 = 0;
    FrStage fr;
    MsmBases g1, g2;
    DevBuf mid_idx;                     // variable index of the j-th mid variable
    DevBuf wit_resident;                // zk_groth16_set_witness
    bool have_witness = false;
    bool lagrange = false;              // key holds [l_i(tau)] and the shifted-domain h bases instead of tau powers (row f4)
    std::unique_ptr<Slot> slots[MAX_SLOTS];
    bool one_stream_slots = false;      // shard of a multi-device key: its products always run on the slot's one stream (no extra streams for slot 0)
    int vdev = 0;                       // virtual device (context) the key was built under; every call on it runs with that context current
};


// Builds a key (whole: rank 0 of world 1, or rank's shard) on the CURRENT virtual device; nothing is registered under a handle.
int groth16_key_build(std::unique_ptr<Groth16Key>& out, uint32_t n, uint32_t m, const zk_csr* L, const zk_csr* R, const zk_csr* O, const uint8_t* mid,
                      const uint8_t* pk_g1, size_t pk_g1_points, const uint8_t* pk_g2, size_t pk_g2_points, uint32_t rank, uint32_t world, bool lagrange,
                      bool shard_of_group = false);
int groth16_slot_get(Groth16Key& k, uint32_t idx, Slot** out);
// host half of a proof's inputs (witness handed over as a host buffer, r, s) into the slot's pinned staging memory; fails when the slot is busy
int groth16_stage_inputs(Groth16Key& k, Slot& sl, const uint8_t* sol, const uint8_t* r, const uint8_t* s);
// Fr stage -> the three scalar vectors over the FULL pools (p1, p1, p2 canonical Fr) at dA / dC / dB; enqueue only (slot stream s0)
int groth16_scalars_enqueue(Groth16Key& k, Slot& sl, const uint8_t* sol, const uint8_t* r, const uint8_t* s, void* dA, void* dC, void* dB, bool staged = false);
// the three products over this key's slice; dA / dC / dB point at the scalars OF THAT SLICE; raw: XYZZ partial sums stay in sl.results (A | C | B)
int groth16_msms_enqueue(Groth16Key& k, Slot& sl, const void* dA, const void* dC, const void* dB, bool raw, int force_serial = -1);
int groth16_prove_finish(Slot& sl);          // waits for sl.done, frees the slot, maps the Fr stage's flags to ZK_ERR_REMAINDER / ZK_ERR_SCALAR_RANGE
// replaces the key's pools by rank's slice of the complete Lagrange-form pools at d_g1 / d_g2 (device memory of the key's device) and flips its Fr stage
int groth16_install_lagrange(Groth16Key& k, const void* d_g1, const void* d_g2, uint32_t rank, uint32_t world);
int groth16_derive_lagrange_pools(const FrStage& f, const uint8_t* d_g1, uint64_t n_mid, const uint8_t* d_g2, uint8_t* out_g1, uint8_t* out_g2, uint32_t sets, hipStream_t s);   // lagrange_derive.hip
void groth16_shard_range(uint64_t points, uint64_t heavy, uint32_t rank, uint32_t world, uint64_t* lo, uint64_t* hi);

// ---- multi-device keys (groth16_multi.hip): one shard per entry of the device list behind ONE handle
struct GroupKey;
GroupKey* group_lookup(uint64_t handle);
uint64_t group_live_handles();
void group_release_all();
int group_upload(uint32_t n, uint32_t m, const zk_csr* L, const zk_csr* R, const zk_csr* O, const uint8_t* mid, const uint8_t* pk_g1, size_t pk_g1_points,
                 const uint8_t* pk_g2, size_t pk_g2_points, bool lagrange, uint64_t* handle);
int group_free(uint64_t handle);
int group_reserve_slots(GroupKey& g, uint32_t count);
int group_set_witness(GroupKey& g, const uint8_t* sol);
int group_prove_async(GroupKey& g, const uint8_t* sol, const uint8_t* r, const uint8_t* s, uint32_t slot);
int group_prove_wait(GroupKey& g, uint32_t slot, uint8_t proof[384]);
int group_derive_lagrange(GroupKey& g);
int group_pool_points(GroupKey& g, int group, uint8_t* out, size_t capacity_points, size_t* count);
int group_pool_layout(GroupKey& g, uint64_t* p1, uint64_t* p2);
int group_qap_eval(GroupKey& g, const uint8_t* sol, uint8_t* v_out, uint8_t* w_out, uint8_t* h_out);
int group_lagrange_pool_sizes(GroupKey& g, uint64_t* g1_points, uint64_t* g2_points);
int single_qap_eval(Groth16Key& k, const uint8_t* sol, uint8_t* v_out, uint8_t* w_out, uint8_t* h_out);
int single_pool_points(Groth16Key& k, int group, uint8_t* out, size_t capacity_points, size_t* count);

}  // namespace zk
