// G2 bucket accumulation on lane pairs (Fp2H: one Fp2 component per lane), field products out of line.
#include "msm_acc.cuh"

namespace zk {
int msm_accumulate_launch_g1(uint64_t nthreads, const void* table, const uint32_t* offsets, const uint32_t* sorted, uint32_t nb, uint32_t chunk,
                             void* buckets, void* head, void* tail, hipStream_t s);
int msm_accumulate_launch(Curve curve, uint64_t nthreads, const void* table, const uint32_t* offsets, const uint32_t* sorted, uint32_t nb,
                          uint32_t chunk, void* buckets, void* head, void* tail, hipStream_t s) {
    if (curve == CURVE_G1) return msm_accumulate_launch_g1(nthreads, table, offsets, sorted, nb, chunk, buckets, head, tail, s);
    hipLaunchKernelGGL(k_msm_accumulate<Fp2H>, dim3((unsigned)((2 * nthreads + 127) / 128)), dim3(128), 0, s, (const uint8_t*)table, offsets, sorted, nb,
                       chunk, (uint8_t*)buckets, (uint8_t*)head, (uint8_t*)tail);
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
}  // namespace zk
