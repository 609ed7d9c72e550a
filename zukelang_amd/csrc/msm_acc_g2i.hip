// G2 bucket accumulation on lane pairs with the field products EXPANDED IN PLACE (as msm_acc_g1.hip does for G1): no call inside the mixed
// addition, so the record of the next step can travel HBM -> LDS by LDS-DMA while this step computes -- every function entry drains the
// loads in flight (s_waitcnt 0 is part of the calling convention), which left the out-of-line form (msm_acc_g2.hip) waiting for a full
// random-gather round trip at the top of every step.  One copy of the group law (no 6-product second step): ~68 KB of code.
#define ZK_FP_INLINE_MUL 1
#include "msm_acc.cuh"

namespace zk {
int msm_accumulate_launch_g2_inline(uint64_t nthreads, const void* table, const AccJobs& jobs, uint32_t count, uint32_t nb, uint32_t chunk, hipStream_t s) {
    hipLaunchKernelGGL((k_msm_accumulate<Fp2H, false, true, false>), dim3((unsigned)((2 * nthreads + 127) / 128), count), dim3(128), 0, s, (const uint8_t*)table, jobs, nb, chunk);
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
}  // namespace zk
