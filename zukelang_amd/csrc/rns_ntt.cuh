// Convolutions of Fr vectors through a residue number system (round 4; csrc/rns_ntt.hip) -- internal interface.
//
// The Fr stage multiplies polynomials over Bls12_381.Fr (QAP.eval, src/lib/zk/QAP.ml:120-135; Polynomial.mul, polynomial.ml:124-131 == the reference's own
// FFT.polynomial_mul, FFT.ml:98-105).  On the 9 x 29-bit Montgomery multiplier a radix-2 butterfly is ~380 vector instructions and the NTT kernels are
// bound by their issue rate.  A product of Fr polynomials is also an INTEGER convolution (coefficients < 2^22 r^2 < 2^532) reduced mod r afterwards, and an
// integer convolution can be taken modulo 18 NTT-friendly 31-bit primes, where a butterfly is ~11 instructions: 18 x 11 = 200 per Fr butterfly (measured:
// 4.5 T 31-bit butterflies/s = 250 G Fr-butterfly equivalents/s against ~96 G, profiles/r04_rns_modmul_rate_probe.txt), at the price of moving 72 instead
// of 32 bytes per element -- the prover is bound by its vector ALUs with the HBM four fifths idle, so that is the right trade.  Exact arithmetic: the results
// are the same field elements, hence the same proof bytes.
#pragma once
#include "zk_common.h"

namespace zk {

static constexpr uint32_t RNS_PRIMES = 18;
static constexpr uint32_t RNS_MAX_LOG = 23;          // transform length up to 2^23 (2^22 constraints)

// how the Fr input of a transform is laid over its RNS work array
enum RnsIn {
    RNS_IN_PLAIN = 0,        // work[e] = src[e]
    RNS_IN_TREE_HI = 1,      // nodes of 2^log_len: work[node + j] = j < len / 2 ? src[node + len / 2 + j] : 0   (upper half of every node, zero padded)
};
enum RnsOut {
    RNS_OUT_RANGE = 0,       // dst[b * dst_stride + k] = result[b * len + out_lo + k], k < out_cnt, for every transform b of the batch
    RNS_OUT_TREE_ADD = 1,    // dst[node + j] = result[node + j] + (j < len / 2 ? dst[node + j] : 0)
};

// residues of `total` elements: [RNS_PRIMES][total] 32-bit words
struct RnsWork {
    DevBuf res;
    uint64_t cap = 0;
    int ensure(uint64_t total) {
        if (cap >= total) return ZK_OK;
        ZKCHK(res.alloc(4 * (size_t)RNS_PRIMES * total));
        cap = total;
        return ZK_OK;
    }
};

bool rns_enabled();                                   // ZK_FR_RNS (a kernel-form switch: cached unless ZK_TEST_FORMS=1); default in rns_ntt.hip
int rns_ensure_twiddles(uint32_t log_len);
// A fixed factor of convolutions (subproduct of a tree node, kernel of an extrapolation, power-series inverse): `total` Montgomery-form Fr coefficients
// (R = 2^256, as everywhere in memory), transformed in independent blocks of 2^log_len, scaled by 2^-log_len and stored as Montgomery multipliers.
int rns_table_build(DevBuf& table, const void* d_fr, uint64_t total, uint32_t log_len, hipStream_t s);
// dst <- (src (*) table) for every block of 2^log_len of the `total` elements; table index = element index & tab_mask (tab_mask + 1 a multiple of 2^log_len)
int rns_conv_table(RnsWork& w, const void* d_src_fr, uint64_t total, uint32_t log_len, RnsIn in_mode, const DevBuf& table, uint64_t tab_mask,
                   void* d_dst_fr, RnsOut out_mode, uint64_t out_lo, uint64_t out_cnt, uint64_t dst_stride, hipStream_t s);
// dst <- a (*) b, both data (one transform of 2^log_len each): the product v w of the quotient step, zk_fr_poly_mul
int rns_conv_data(RnsWork& wa, RnsWork& wb, const void* d_a_fr, const void* d_b_fr, uint32_t log_len, void* d_dst_fr, uint64_t out_lo, uint64_t out_cnt, hipStream_t s);

}  // namespace zk
