// Bucket accumulation of the Pippenger MSM (step 4 of msm.hip): shared by msm_acc_g1.hip (field products
// expanded in place) and msm_acc_g2.hip (lane-pair Fp2, products out of line).
#pragma once
#include "ec.cuh"
#include "msm.cuh"

#include <type_traits>

namespace zk {

// one table entry in the dense memory format, still packed: what the loop keeps in flight for the NEXT step
template <class F> struct PackedAff {
    FpWords x, y;
};
template <class F> FF_INLINE PackedAff<F> packed_aff_load(const uint8_t* p) {
    if constexpr (std::is_same<F, Fp2H>::value) {
        const uint32_t c = 48 * pair_comp();
        return {fpw_load(p + c), fpw_load(p + 96 + c)};
    } else {
        return {fpw_load(p), fpw_load(p + 48)};
    }
}
template <class F> FF_INLINE Aff<F> packed_aff_unpack(const PackedAff<F>& a, bool negate) {
    const FpB<1> x = fp_unpack(a.x), y = fp_unpack(a.y);
    Aff<F> r;
    if constexpr (std::is_same<F, Fp2H>::value) {
        r.x = Fp2H(x);
        r.y = negate ? Fp2H(fe_neg(y)) : Fp2H(y);
    } else {
        r.x = x;
        if (negate) r.y = fe_neg(y);      // table entries are fully reduced: -y = 2p - y
        else r.y = y;
    }
    return r;
}

// RAW = false: the sorted entries are table references (index | sign).  RAW = true (the finisher of the batch-affine rounds,
// msm_ba.cuh): entry `pos` is the affine point at pts + pos * 2 * RawLayout<F>::ELEM in the raw limb layout, identity = (0, 0).
template <class F, bool RAW>
__global__ __launch_bounds__(128, 2) void k_msm_accumulate(const uint8_t* __restrict__ table, AccJobs jobs, uint32_t nb, uint32_t chunk) {
    const uint32_t* __restrict__ offsets = jobs.offsets[blockIdx.y];
    const uint32_t* __restrict__ sorted = jobs.sorted[blockIdx.y];
    uint8_t* __restrict__ buckets = jobs.buckets[blockIdx.y];
    uint8_t* __restrict__ head = jobs.head[blockIdx.y];
    uint8_t* __restrict__ tail = jobs.tail[blockIdx.y];
    constexpr int AB = FieldOps<F>::WORDS * 8, XB = RawLayout<F>::XYZZ;       // table: dense; sums: raw layout
    constexpr bool PAIR = std::is_same<F, Fp2H>::value;      // G2: two lanes per chunk, one Fp2 component each
    const uint64_t t = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> (PAIR ? 1 : 0);
    const uint32_t N = offsets[nb];
    uint64_t start64 = t * chunk;
    if (start64 >= N) return;
    uint32_t pos = (uint32_t)start64;
    const uint32_t pos0 = pos;
    const uint32_t end = min(pos + chunk, N);
    // largest kb with offsets[kb] <= pos (then offsets[kb+1] > pos, so the bucket is non-empty)
    uint32_t lo = 0, hi = nb;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (offsets[mid] <= pos) lo = mid; else hi = mid;
    }
    uint32_t kb = lo;
    // Flat loop: one mixed addition per lane per iteration whatever the run boundaries are (a loop
    // nest over runs would make the wave pay the longest run of every lane in turn).  A lane
    // crossing into the next bucket stores its running sum first -- a short divergent epilogue.
    // The reference and the table entry of step i+1 are requested before the mixed addition of step i
    // (24 registers of look-ahead): the gather latency hides behind ~5k ALU instructions.
    uint32_t bstart = offsets[kb], bend = offsets[kb + 1];
    uint32_t seg_start = pos;
    bool first = true;
    Xyzz<F> acc = xyzz_inf<F>();
    // (G2: the products are calls, which drain outstanding loads anyway -- only the reference is fetched ahead there,
    // the 24 look-ahead registers would be spilled)
    constexpr int RB = RawLayout<F>::ELEM;
    const uint8_t* __restrict__ pts = jobs.pts[blockIdx.y];
    uint32_t v_next = 0;
    PackedAff<F> p_next;
    if constexpr (!RAW) {
        v_next = sorted[pos];
        if constexpr (!PAIR) p_next = packed_aff_load<F>(table + (uint64_t)AB * (v_next & 0x7fffffffu));
    }
    for (; pos < end; pos++) {
        const uint32_t v = v_next;
        PackedAff<F> pk;
        if constexpr (!RAW) {
            if constexpr (PAIR) pk = packed_aff_load<F>(table + (uint64_t)AB * (v & 0x7fffffffu));
            else pk = p_next;
            if (pos + 1 < end) {
                v_next = sorted[pos + 1];
                if constexpr (!PAIR) p_next = packed_aff_load<F>(table + (uint64_t)AB * (v_next & 0x7fffffffu));
            }
        }
        if (pos == bend) {                              // run finished inside the chunk
            const bool complete = seg_start == bstart;
            uint8_t* dst = complete ? buckets + (uint64_t)XB * kb : (first ? head + (uint64_t)XB * t : tail + (uint64_t)XB * t);
            xyzz_store_raw<F>(dst, acc);
            first = false;
            acc = xyzz_inf<F>();
            do { kb++; bstart = bend; bend = offsets[kb + 1]; } while (bend == bstart);   // skip empty buckets
            seg_start = pos;
        }
        Aff<F> p;
        if constexpr (RAW) p = {load_raw_f((const F*)nullptr, pts + (uint64_t)2 * RB * pos), load_raw_f((const F*)nullptr, pts + (uint64_t)2 * RB * pos + RB)};
        else p = packed_aff_unpack<F>(pk, (v >> 31) != 0);
        // The mixed addition is inlined so the accumulator lives in VGPRs for the whole chunk.  G2 runs
        // as F = Fp2H, one Fp2 component per lane of a pair, which gives it the register footprint of G1.
        // Second step of the chunk (a wave-uniform test): every lane holds the identity (run border just crossed) or
        // the single affine point of step one, so the 6-product addition of two affine points does.
        if (pos == pos0 + 1) xyzz_mmadd_impl(acc, p);
        else xyzz_madd_impl(acc, p);
    }
    {
        const bool complete = (seg_start == bstart) && (end == bend);
        uint8_t* dst = complete ? buckets + (uint64_t)XB * kb : (first ? head + (uint64_t)XB * t : tail + (uint64_t)XB * t);
        xyzz_store_raw<F>(dst, acc);
    }
}

}  // namespace zk
