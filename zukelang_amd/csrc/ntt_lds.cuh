// Radix-2 butterfly stages over one LDS-resident tile (shared by the NTT passes and the fused
// basis-conversion levels of the Fr stage).  Layout: limb-major lds[9][NTT_T] of lazily reduced
// 29-bit-limb integers (fr29.cuh); twiddles are read from memory in the 8-word format, as w * 2^261 (a heap of
// unpacked 36-byte entries measured 1-3 % slower: three loads per twiddle instead of two aligned ones).
//
// Bound discipline (value < B r, see fr29.cuh):
//   DIT (inverse): inputs < 2r; x = u + v w < (B + 2) r, y = u - v w + 4r < (B + 4) r: after s <= 10 stages B <= 42.
//   DIF (forward): inputs < 3r; x = u + v doubles B, y = (u - v + 32 r) w < 2r; x is brought back below 2r after
//                  every fourth stage (B = 3, 6, 12, 24 -> 48 -> 2), so u - v + 32 r < 56 r and the outputs are < 48 r
//                  (< 56 r after the product-free last stage of a contiguous pass).
#pragma once
#include "fr29.cuh"

namespace zk {

static constexpr int NTT_LOG_T = 10;
static constexpr int NTT_T = 1 << NTT_LOG_T;
static constexpr int NTT_THREADS = 256;
using NttTile = uint32_t[FR29_L][NTT_T];

// Runs `s` stages on T tile elements.  Row step in LDS = 2^log_RS elements; global column stride
// L = 2^log_L; twiddle heap: tw[h + j] = w_{2h}^(+-j).  Ends with a barrier.
template <bool INVERSE>
FF_INLINE void lds_ntt_stages(NttTile& lds, const uint32_t* __restrict__ tw, uint32_t T, uint32_t s,
                              uint32_t log_RS, uint32_t log_L, uint32_t c0, bool strided) {
    const uint32_t RSm = (1u << log_RS) - 1;
    for (uint32_t st = 0; st < s; st++) {
        const uint32_t log_hl = INVERSE ? st : s - 1 - st;              // local half = 2^log_hl rows
        const uint32_t log_hs = log_hl + log_RS;
        const uint32_t hs = 1u << log_hs;
        const uint32_t hlm = (1u << log_hl) - 1;
        const uint64_t h = (uint64_t)1 << (log_hl + log_L);              // global half-distance
        const bool shrink = !INVERSE && (st & 3) == 3;
        if (log_hl + log_L == 0) {
            // span 2: the twiddle is w_2^0 = 1 and the product is skipped (a fifth of the stages of the fused tree levels).
            // DIT: first stage, inputs < 2r, outputs < 6r like with the product.  DIF: last stage, y = u - v + 32 r stays
            // below 56 r, which the pointwise product or the store that follows accepts.
            for (uint32_t b = threadIdx.x; b < T / 2; b += NTT_THREADS) {
                const uint32_t e = ((b >> log_hs) << (log_hs + 1)) | (b & (hs - 1));
                const Fr9 u = fr9_lds_get(lds, e), v = fr9_lds_get(lds, e + hs);
                Fr9 x = fr9_add(u, v);
                if (shrink) x = fr9_reduce_weak(x);
                fr9_lds_put(lds, e, x);
                fr9_lds_put(lds, e + hs, INVERSE ? fr9_sub<2>(u, v) : fr9_sub<5>(u, v));
            }
            __syncthreads();
            continue;
        }
        for (uint32_t b = threadIdx.x; b < T / 2; b += NTT_THREADS) {
            uint32_t e = ((b >> log_hs) << (log_hs + 1)) | (b & (hs - 1));
            uint32_t rho = (e >> log_RS) & hlm;
            uint32_t c = strided ? c0 + (e & RSm) : (e & RSm) & ((1u << log_L) - 1);
            uint64_t j = ((uint64_t)rho << log_L) + c;
            const Fr9 w = fr9_load(tw + 8 * (h + j));
            const Fr9 u = fr9_lds_get(lds, e), v = fr9_lds_get(lds, e + hs);
            Fr9 x, y;
            if (INVERSE) {
                const Fr9 t = fr9_mul(v, w);
                x = fr9_add(u, t);
                y = fr9_sub<2>(u, t);
            } else {
                x = fr9_add(u, v);
                if (shrink) x = fr9_reduce_weak(x);
                y = fr9_mul(fr9_sub_raw<5>(u, v), w);
            }
            fr9_lds_put(lds, e, x);
            fr9_lds_put(lds, e + hs, y);
        }
        __syncthreads();
    }
}

}  // namespace zk
