// G1 bucket accumulation with the field products expanded in place: no call boundary inside the mixed
// addition, so the gather of the next table entry stays in flight across it (the compiler must drain
// outstanding loads at every call).  ~40 KB of straight-line code per mixed addition: it fits the
// 64 KB instruction cache shared by a CU pair.
#define ZK_FP_INLINE_MUL 1
#include "msm_acc.cuh"

#include <stdlib.h>

namespace zk {
int msm_accumulate_launch_g1(uint64_t nthreads, const void* table, const AccJobs& jobs, uint32_t count, uint32_t nb, uint32_t chunk, hipStream_t s) {
    // ZK_ACC_G1_GLDS=0: the register look-ahead (A/B of the LDS-DMA look-ahead).  Cached; per launch only under ZK_TEST_FORMS=1 (zk_common.h).
    const char *eg = ZK_FORM_ENV("ZK_ACC_G1_GLDS"), *em = ZK_FORM_ENV("ZK_ACC_G1_MMADD");
    const bool glds = !(eg && atoi(eg) == 0);
    const bool mm = em && atoi(em) != 0;      // A/B: the 6-product second step (more code)
    if (glds && !mm) hipLaunchKernelGGL((k_msm_accumulate<Fp, false, true, false>), dim3((unsigned)((nthreads + 127) / 128), count), dim3(128), 0, s, (const uint8_t*)table, jobs, nb, chunk);
    else if (glds) hipLaunchKernelGGL((k_msm_accumulate<Fp, false, true>), dim3((unsigned)((nthreads + 127) / 128), count), dim3(128), 0, s, (const uint8_t*)table, jobs, nb, chunk);
    else hipLaunchKernelGGL((k_msm_accumulate<Fp, false, false>), dim3((unsigned)((nthreads + 127) / 128), count), dim3(128), 0, s, (const uint8_t*)table, jobs, nb, chunk);
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
}  // namespace zk
