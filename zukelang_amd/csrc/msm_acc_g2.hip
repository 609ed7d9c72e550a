// G2 bucket accumulation on lane pairs (Fp2H: one Fp2 component per lane), field products out of line.
#include "msm_acc.cuh"

namespace zk {
int msm_accumulate_launch_g1(uint64_t nthreads, const void* table, const AccJobs& jobs, uint32_t count, uint32_t nb, uint32_t chunk, hipStream_t s);
int msm_accumulate_launch(Curve curve, uint64_t nthreads, const void* table, const AccJobs& jobs, uint32_t count, uint32_t nb, uint32_t chunk, hipStream_t s) {
    if (curve == CURVE_G1) return msm_accumulate_launch_g1(nthreads, table, jobs, count, nb, chunk, s);
    hipLaunchKernelGGL(k_msm_accumulate<Fp2H>, dim3((unsigned)((2 * nthreads + 127) / 128), count), dim3(128), 0, s, (const uint8_t*)table, jobs, nb, chunk);
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
}  // namespace zk
