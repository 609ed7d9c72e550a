// G2 bucket accumulation on lane pairs (Fp2H: one Fp2 component per lane), field products out of line.
#include "msm_acc.cuh"

#include <stdlib.h>

namespace zk {
int msm_accumulate_launch_g1(uint64_t nthreads, const void* table, const AccJobs& jobs, uint32_t count, uint32_t nb, uint32_t chunk, hipStream_t s);
int msm_accumulate_launch_g2_inline(uint64_t nthreads, const void* table, const AccJobs& jobs, uint32_t count, uint32_t nb, uint32_t chunk, hipStream_t s);
int msm_accumulate_launch(Curve curve, uint64_t nthreads, const void* table, const AccJobs& jobs, uint32_t count, uint32_t nb, uint32_t chunk, hipStream_t s, bool raw) {
    if (raw) {
        // the finisher of the batch-affine rounds: a few entries per bucket are left, the products stay out of line for both curves
        if (curve == CURVE_G1) hipLaunchKernelGGL((k_msm_accumulate<Fp, true>), dim3((unsigned)((nthreads + 127) / 128), count), dim3(128), 0, s, (const uint8_t*)table, jobs, nb, chunk);
        else hipLaunchKernelGGL((k_msm_accumulate<Fp2H, true>), dim3((unsigned)((2 * nthreads + 127) / 128), count), dim3(128), 0, s, (const uint8_t*)table, jobs, nb, chunk);
        HIPCHK(hipGetLastError());
        return ZK_OK;
    }
    if (curve == CURVE_G1) return msm_accumulate_launch_g1(nthreads, table, jobs, count, nb, chunk, s);
    const char* ei = ZK_FORM_ENV("ZK_ACC_G2_INLINE");
    const bool g2_inline = !(ei && atoi(ei) == 0);      // A/B switch (cached; per launch only under ZK_TEST_FORMS=1)
    if (g2_inline) return msm_accumulate_launch_g2_inline(nthreads, table, jobs, count, nb, chunk, s);
    hipLaunchKernelGGL((k_msm_accumulate<Fp2H, false>), dim3((unsigned)((2 * nthreads + 127) / 128), count), dim3(128), 0, s, (const uint8_t*)table, jobs, nb, chunk);
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
}  // namespace zk
