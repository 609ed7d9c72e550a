// Batch-affine halving rounds of the bucket accumulation (msm_ba.cuh), both curves, field products out of line.
#include "msm_ba.cuh"

#include <stdlib.h>

namespace zk {

static uint64_t env_u64(const char* name, uint64_t dflt) {
    const char* e = ::zk::opt(name);
    return e ? (uint64_t)atoll(e) : dflt;
}

int msm_batch_affine_rounds(const MsmBases& b, MsmWorkspace* const* ws, uint32_t count, uint32_t rounds, hipStream_t s) {
    if (!rounds) return ZK_OK;
    const uint32_t nb = ws[0]->nbuckets;
    BaPlanJobs pj{};
    uint64_t max_entries = 0;
    for (uint32_t i = 0; i < count; i++) {
        MsmWorkspace& w = *ws[i];
        if (w.ba_rounds < rounds || !w.ba_e[0].p) ZK_FAIL(ZK_ERR_ARG, "msm: workspace has no batch-affine buffers");
        pj.off0[i] = w.offsets.as<uint32_t>();
        pj.offs[i] = w.ba_offs.as<uint32_t>();
        if (w.ba_max_entries > max_entries) max_entries = w.ba_max_entries;
    }
    hipLaunchKernelGGL(k_ba_plan, dim3(rounds, count), dim3(1024), 0, s, pj, nb);
    // lanes per round: L additions per lane share one inversion; few lanes when a single proof owns the chip would leave it
    // empty, many lanes make the inversion expensive per addition -- both knobs are measured defaults
    static const uint64_t target_lanes = env_u64("ZK_BA_TARGET_LANES", 16384);
    static const uint64_t lmin = env_u64("ZK_BA_LMIN", 16), lmax = env_u64("ZK_BA_LMAX", 128);
    for (uint32_t r = 0; r < rounds; r++) {
        BaJobs j{};
        for (uint32_t i = 0; i < count; i++) {
            MsmWorkspace& w = *ws[i];
            j.refs[i] = w.sorted.as<uint32_t>();
            j.src[i] = r ? w.ba_e[(r - 1) & 1].as<uint8_t>() : nullptr;
            j.dst[i] = w.ba_e[r & 1].as<uint8_t>();
            j.off_in[i] = r ? w.ba_offs.as<uint32_t>() + (uint64_t)(r - 1) * (nb + 1) : w.offsets.as<uint32_t>();
            j.off_out[i] = w.ba_offs.as<uint32_t>() + (uint64_t)r * (nb + 1);
            j.cap[i] = w.ba_cap[r & 1];
        }
        const uint64_t items = (max_entries >> (r + 1)) + nb;          // upper bound of the round's output items over the jobs
        uint64_t L = items / target_lanes;
        L = L < lmin ? lmin : (L > lmax ? lmax : L);
        const uint64_t lanes = (items + L - 1) / L;
        const unsigned threads = 128;
        if (b.curve == CURVE_G1) {
            const dim3 g((unsigned)((lanes + threads - 1) / threads), count);
            if (r == 0) hipLaunchKernelGGL((k_ba_round<Fp, true>), g, dim3(threads), 0, s, b.table.as<uint8_t>(), j, nb, (uint32_t)L);
            else hipLaunchKernelGGL((k_ba_round<Fp, false>), g, dim3(threads), 0, s, b.table.as<uint8_t>(), j, nb, (uint32_t)L);
        } else {
            const dim3 g((unsigned)((2 * lanes + threads - 1) / threads), count);
            if (r == 0) hipLaunchKernelGGL((k_ba_round<Fp2H, true>), g, dim3(threads), 0, s, b.table.as<uint8_t>(), j, nb, (uint32_t)L);
            else hipLaunchKernelGGL((k_ba_round<Fp2H, false>), g, dim3(threads), 0, s, b.table.as<uint8_t>(), j, nb, (uint32_t)L);
        }
    }
    HIPCHK(hipGetLastError());
    return ZK_OK;
}

}  // namespace zk
