// Radix-2 butterfly stages over one LDS-resident tile (shared by the NTT passes and the fused
// basis-conversion levels of the Fr stage).  Layout: limb-major lds[8][NTT_T].
#pragma once
#include "ff.cuh"

namespace zk {

static constexpr int NTT_LOG_T = 10;
static constexpr int NTT_T = 1 << NTT_LOG_T;
static constexpr int NTT_THREADS = 256;

// Runs `s` stages on T tile elements.  Row step in LDS = 2^log_RS elements; global column stride
// L = 2^log_L; twiddle heap: tw[h + j] = w_{2h}^(+-j).  Ends with a barrier.
template <bool INVERSE>
FF_INLINE void lds_ntt_stages(uint32_t (*lds)[NTT_T], const uint32_t* __restrict__ tw, uint32_t T, uint32_t s,
                              uint32_t log_RS, uint32_t log_L, uint32_t c0, bool strided) {
    const uint32_t RSm = (1u << log_RS) - 1;
    for (uint32_t st = 0; st < s; st++) {
        const uint32_t log_hl = INVERSE ? st : s - 1 - st;              // local half = 2^log_hl rows
        const uint32_t log_hs = log_hl + log_RS;
        const uint32_t hs = 1u << log_hs;
        const uint32_t hlm = (1u << log_hl) - 1;
        const uint64_t h = (uint64_t)1 << (log_hl + log_L);              // global half-distance
        for (uint32_t b = threadIdx.x; b < T / 2; b += NTT_THREADS) {
            uint32_t e = ((b >> log_hs) << (log_hs + 1)) | (b & (hs - 1));
            uint32_t rho = (e >> log_RS) & hlm;
            uint32_t c = strided ? c0 + (e & RSm) : (e & RSm) & ((1u << log_L) - 1);
            uint64_t j = ((uint64_t)rho << log_L) + c;
            Fr w = fe_load<FrParams>(tw + 8 * (h + j));
            Fr u, v;
#pragma unroll
            for (int l = 0; l < 8; l++) { u.v[l] = lds[l][e]; v.v[l] = lds[l][e + hs]; }
            Fr x, y;
            if (INVERSE) {
                v = fe_mul(v, w);
                x = fe_add(u, v);
                y = fe_sub(u, v);
            } else {
                x = fe_add(u, v);
                y = fe_mul(fe_sub(u, v), w);
            }
#pragma unroll
            for (int l = 0; l < 8; l++) { lds[l][e] = x.v[l]; lds[l][e + hs] = y.v[l]; }
        }
        __syncthreads();
    }
}

}  // namespace zk
