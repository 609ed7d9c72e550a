// Steps 1-3 of the Pippenger pipeline of msm.hip: signed c-bit digits of the scalars and the counting sort of the (scalar, window) pairs by
// bucket -- global atomics for small products, LDS-privatised histograms for resident keys, two levels (512 coarse bins, then one workgroup per
// bin) from 2^15 buckets and 2^20 pairs up.  Split off msm.hip in round 5; msm_sort_launch (msm.cuh) is the one entry point.
#include "ec.cuh"
#include "msm.cuh"

#include <stdlib.h>
#include <string.h>

namespace zk {

// ------------------------------------------------------------------ digits
// Signed c-bit digits d_j in [-(2^(c-1) - 1), 2^(c-1)] with sum_j d_j 2^(cj) = s.  Adding the constant
// K = sum_j (2^(c-1) - 1) 2^(cj) turns the recoding into plain base-2^c digit extraction:
// d_j = ((s + K) >> cj & mask) - (2^(c-1) - 1), so every (scalar, window) pair is independent and
// gets its own lane: one atomic per lane in flight instead of nw dependent ones.
struct DigitArgs {
    uint64_t n;
    uint32_t c, nw, precomp, nb_per_window;
    uint32_t K[9];         // the recoding constant, 288 bits
    const uint8_t* ident;  // precomp: 1 = base i is the identity: it never enters a bucket (nullptr: no filter)
    uint32_t coarse_shift;  // two-level sort, level 1: histogram / rank by bucket >> coarse_shift and emit (bucket, reference) records
    uint32_t scalar_major;  // LDS sorts: a workgroup owns a range of SCALARS and files all their digits (each scalar is read once per pass,
                            // not once per window: 13-16x less scalar traffic in the two passes); 0: a range of (scalar, window) pairs, window-major
    uint32_t alias_windows; // EXPERIMENT, compiled in only with -DZK_EXPERIMENTS (scripts/table_alias_ab.sh; results WRONG by design): every window
                            // reads window 0's table entries -- same additions and gathers, 1/16 of the table footprint.  Always 0 in the shipped library.
    uint32_t fold;          // digits of min(s, r - s), sign carried to every digit (msm.cuh: msm_windows): bit 31 of the ninth word of a prepared scalar is the sign
};
// scalar i plus the recoding constant (9 words); false: the scalar is zero or its base is the identity -- no digit of it enters a bucket
FF_INLINE bool digits_prepare(const uint32_t* __restrict__ scalars, uint64_t i, const DigitArgs& a, uint32_t s[9]) {
    const uint32_t* sp = scalars + 8 * i;
    uint4 lo = reinterpret_cast<const uint4*>(sp)[0], hi = reinterpret_cast<const uint4*>(sp)[1];
    s[0] = lo.x; s[1] = lo.y; s[2] = lo.z; s[3] = lo.w; s[4] = hi.x; s[5] = hi.y; s[6] = hi.z; s[7] = hi.w; s[8] = 0;
    if ((s[0] | s[1] | s[2] | s[3] | s[4] | s[5] | s[6] | s[7]) == 0) return false;
    if (a.ident && a.ident[i]) return false;
    uint32_t flip = 0;
    if (a.fold) {                                        // wave-uniform
        uint32_t t[8];
        int64_t bw = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            bw += (int64_t)FR_MOD[k] - (int64_t)s[k];
            t[k] = (uint32_t)bw;
            bw >>= 32;
        }
        bool less = false, decided = false;              // t < s, from the top word down (r is odd: t != s)
#pragma unroll
        for (int k = 7; k >= 0; k--) {
            if (!decided && t[k] != s[k]) { less = t[k] < s[k]; decided = true; }
        }
        if (less) {
            flip = 0x80000000u;
#pragma unroll
            for (int k = 0; k < 8; k++) s[k] = t[k];
        }
    }
    uint64_t cy = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        cy += (uint64_t)s[k] + a.K[k];
        s[k] = (uint32_t)cy;
        cy >>= 32;
    }
    s[8] |= flip;                                        // c nw <= 276 bits: the ninth word uses 20 bits at most
    return true;
}
FF_INLINE bool digit_at(const uint32_t s[9], uint64_t i, uint32_t j, const DigitArgs& a, uint32_t& key, uint32_t& val);
FF_INLINE bool digit_of(const uint32_t* __restrict__ scalars, uint64_t i, uint32_t j, const DigitArgs& a, uint32_t& key, uint32_t& val) {
    uint32_t s[9];
    return digits_prepare(scalars, i, a, s) && digit_at(s, i, j, a, key, val);
}
FF_INLINE bool digit_at(const uint32_t s[9], uint64_t i, uint32_t j, const DigitArgs& a, uint32_t& key, uint32_t& val) {
    const uint32_t off = j * a.c, w = off >> 5, b = off & 31;
    uint32_t x0 = 0, x1 = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {        // static indexing keeps the scalar in registers
        if ((int)w == k) x0 = s[k];
        if ((int)w + 1 == k) x1 = k == 8 ? s[k] & 0x7fffffffu : s[k];
    }
    const uint64_t x = ((uint64_t)x1 << 32) | x0;
    const uint32_t e = (uint32_t)(x >> b) & ((1u << a.c) - 1);
    const uint32_t bias = (1u << (a.c - 1)) - 1;
    if (e == bias) return false;                        // digit 0
    const uint32_t below = e < bias ? 1u : 0u;
    const uint32_t d = below ? bias - e : e - bias;      // the digit's magnitude
    const uint32_t neg = below ^ (s[8] >> 31);           // ... its sign, turned round for a folded scalar
    key = (a.precomp ? 0u : j * a.nb_per_window) + (d - 1);
#ifdef ZK_EXPERIMENTS
    val = (uint32_t)(a.precomp && !a.alias_windows ? (uint64_t)j * a.n + i : i) | (neg << 31);
#else
    val = (uint32_t)(a.precomp ? (uint64_t)j * a.n + i : i) | (neg << 31);
#endif
    return true;
}
// Wave-aggregated atomic increment.  Boolean-heavy witnesses put a large share of the digits into ONE bucket
// (scalar 1 = digit 1 of window 0): same-address atomics serialise and the sort of a 2^16 proof went from 1.1 to
// 5.0 ms.  Up to three rounds peel off the key of the wave's first pending lane when at least 8 lanes share it
// (one atomic for all of them, ranks from the ballot); everything else -- all of a uniform input -- does its own atomic.
// The (scalar, window) pairs are laid out WINDOW-major (pair g = window g / n of scalar g % n), so the lanes of a wave
// hold the same window of 64 consecutive scalars -- that is where equal digits sit side by side.
// Returns the slot of this lane's entry (meaningful for the scatter; the count ignores it).
template <class Counter> FF_INLINE uint32_t wave_aggregated_add(Counter* __restrict__ ctr, bool ok, uint32_t key) {
    const uint32_t lane = __lane_id();
    uint32_t pos = 0;
    uint64_t pending = __ballot(ok);
    for (int round = 0; round < 3 && pending; round++) {
        const int leader = __ffsll((unsigned long long)pending) - 1;
        const uint32_t k0 = (uint32_t)__shfl((int)key, leader);
        const uint64_t same = __ballot(ok && key == k0);
        if (__popcll(same) < 8) break;                                   // wave-uniform
        uint32_t base = 0;
        if (lane == (uint32_t)leader) base = atomicAdd(&ctr[k0], (uint32_t)__popcll(same));
        base = (uint32_t)__shfl((int)base, leader);
        if (ok && key == k0) {
            pos = base + (uint32_t)__popcll(same & (((uint64_t)1 << lane) - 1));
            ok = false;
        }
        pending &= ~same;
    }
    if (ok) pos = atomicAdd(&ctr[key], 1u);
    return pos;
}
// The sort kernels serve up to 4 MSMs over the same bases in one launch (blockIdx.y = job): same digit geometry,
// different scalar vectors and buffers.
struct SortJobs {
    const uint32_t* scalars[MAX_SORT_JOBS];
    uint32_t* counts[MAX_SORT_JOBS];
    uint32_t* offsets[MAX_SORT_JOBS];
    uint32_t* cursor[MAX_SORT_JOBS];
    uint32_t* sorted[MAX_SORT_JOBS];
    uint32_t* wgcount[MAX_SORT_JOBS];
    uint2* sorted2[MAX_SORT_JOBS];         // two-level sort: level-1 records
};
__global__ void k_msm_count(SortJobs jobs, DigitArgs a) {
    const uint32_t* __restrict__ scalars = jobs.scalars[blockIdx.y];
    uint32_t* __restrict__ counts = jobs.counts[blockIdx.y];
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;      // no early return: the ballots need whole waves
    uint32_t key = 0, val = 0;
    const bool ok = g < a.n * a.nw && digit_of(scalars, g % a.n, (uint32_t)(g / a.n), a, key, val);
    (void)wave_aggregated_add(counts, ok, key);
}
__global__ void k_msm_scatter(SortJobs jobs, DigitArgs a) {
    const uint32_t* __restrict__ scalars = jobs.scalars[blockIdx.y];
    uint32_t* __restrict__ cursor = jobs.cursor[blockIdx.y];
    uint32_t* __restrict__ sorted = jobs.sorted[blockIdx.y];
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t key = 0, val = 0;
    const bool ok = g < a.n * a.nw && digit_of(scalars, g % a.n, (uint32_t)(g / a.n), a, key, val);
    const uint32_t pos = wave_aggregated_add(cursor, ok, key);
    if (ok) sorted[pos] = val;
}
// ---- LDS-privatised counting sort (one bucket set of <= 2^15 buckets: the resident-key mode)
// Global atomics saturate at a few G/s chip-wide, which made the sort as expensive as the accumulate
// at 2^20.  Each workgroup instead owns a contiguous range of (scalar, window) pairs and histograms it
// in LDS (2^15 counters = 128 KiB of the 160 KiB), writes its column of the [bucket][workgroup] count
// matrix, one exclusive scan over that matrix gives every workgroup its private cursor per bucket, and
// the scatter pass ranks with LDS atomics again.  No global atomic at all, and the bucket offsets
// fall out of the same scan.
static constexpr uint32_t SORT_THREADS = 1024;
// LDS_BINS: counters the kernel reserves -- SORT_MAX_BUCKETS (128 KiB: one workgroup per CU and hardly any LDS left for the accumulate kernels of the
// other proofs in flight) or SORT_FEW_BINS for the 512 coarse bins of the two-level sort's first level (2 KiB: the sort of a 2^20 proof no longer
// evicts the accumulate workgroups from the compute units it runs on)
// k_sort_scatter_staged (level 1 of the two-level sort with its records staged through LDS) -- measured at 2^20, same box: its writes 1.14 -> 0.74 GB per proof
// (sort kernels 3.2 -> 2.8 GB as 2 FETCH + WRITE), but the launch is 2 % LONGER (the extra LDS pass and five barriers per tile cost more than the stores
// save) and the pipelined prover loses 0.6-0.9 %: off unless ZK_SORT_COARSE_STAGED=1 (profiles/r04_sort_fine_staged.txt)
static constexpr bool SORT_COARSE_STAGED_DEFAULT = false;
template <uint32_t LDS_BINS>
__global__ __launch_bounds__(SORT_THREADS) void k_sort_count_lds(SortJobs jobs, DigitArgs a, uint64_t per_wg, uint32_t nb) {
    const uint32_t* __restrict__ scalars = jobs.scalars[blockIdx.y];
    uint32_t* __restrict__ wgcount = jobs.wgcount[blockIdx.y];
    __shared__ uint32_t hist[LDS_BINS];
    const uint32_t wg = blockIdx.x;
    for (uint32_t b = threadIdx.x; b < nb; b += SORT_THREADS) hist[b] = 0;
    __syncthreads();
    if (a.scalar_major) {
        const uint64_t lo = (uint64_t)wg * per_wg, hi = min(lo + per_wg, a.n);          // per_wg counts scalars here
        for (uint64_t i0 = lo; i0 < hi; i0 += SORT_THREADS) {                           // whole waves: the aggregated add ballots
            const uint64_t i = i0 + threadIdx.x;
            uint32_t sk[9];
            const bool live = i < hi && digits_prepare(scalars, i, a, sk);
            for (uint32_t j = 0; j < a.nw; j++) {
                uint32_t key = 0, val = 0;
                const bool ok = live && digit_at(sk, i, j, a, key, val);
                (void)wave_aggregated_add(hist, ok, key >> a.coarse_shift);
            }
        }
    } else {
        const uint64_t total = a.n * a.nw, lo = (uint64_t)wg * per_wg, hi = min(lo + per_wg, total);
        for (uint64_t g = lo + threadIdx.x; g < hi; g += SORT_THREADS) {
            uint32_t key = 0, val = 0;
            const bool ok = digit_of(scalars, g % a.n, (uint32_t)(g / a.n), a, key, val);
            (void)wave_aggregated_add(hist, ok, key >> a.coarse_shift);
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += SORT_THREADS) wgcount[(uint64_t)wg * nb + b] = hist[b];   // [workgroup][bucket]: coalesced
}
template <uint32_t LDS_BINS>
__global__ __launch_bounds__(SORT_THREADS) void k_sort_scatter_lds(SortJobs jobs, DigitArgs a, uint64_t per_wg, uint32_t nb) {
    const uint32_t* __restrict__ scalars = jobs.scalars[blockIdx.y];
    const uint32_t* __restrict__ base = jobs.wgcount[blockIdx.y];
    const uint32_t* __restrict__ offsets = jobs.offsets[blockIdx.y];
    uint32_t* __restrict__ sorted = jobs.sorted[blockIdx.y];
    __shared__ uint32_t cur[LDS_BINS];
    const uint32_t wg = blockIdx.x;
    for (uint32_t b = threadIdx.x; b < nb; b += SORT_THREADS) cur[b] = offsets[b] + base[(uint64_t)wg * nb + b];
    __syncthreads();
    if (a.scalar_major) {
        const uint64_t lo = (uint64_t)wg * per_wg, hi = min(lo + per_wg, a.n);
        for (uint64_t i0 = lo; i0 < hi; i0 += SORT_THREADS) {
            const uint64_t i = i0 + threadIdx.x;
            uint32_t sk[9];
            const bool live = i < hi && digits_prepare(scalars, i, a, sk);
            for (uint32_t j = 0; j < a.nw; j++) {
                uint32_t key = 0, val = 0;
                const bool ok = live && digit_at(sk, i, j, a, key, val);
                const uint32_t pos = wave_aggregated_add(cur, ok, key >> a.coarse_shift);
                if (ok) {
                    if (a.coarse_shift) jobs.sorted2[blockIdx.y][pos] = make_uint2(key, val);
                    else sorted[pos] = val;
                }
            }
        }
        return;
    }
    const uint64_t total = a.n * a.nw, lo = (uint64_t)wg * per_wg, hi = min(lo + per_wg, total);
    for (uint64_t g = lo + threadIdx.x; g < hi; g += SORT_THREADS) {
        uint32_t key = 0, val = 0;
        const bool ok = digit_of(scalars, g % a.n, (uint32_t)(g / a.n), a, key, val);
        const uint32_t pos = wave_aggregated_add(cur, ok, key >> a.coarse_shift);
        if (ok) {
            if (a.coarse_shift) jobs.sorted2[blockIdx.y][pos] = make_uint2(key, val);
            else sorted[pos] = val;
        }
    }
}
// Level 1 of the two-level sort with its 8-byte records STAGED through LDS (round 4; ZK_SORT_COARSE_STAGED): the plain scatter stores every record on its own
// and the counters see 1.14 GB written per 2^20 proof for 545 MB of records.  Here the workgroup files the digits of COARSE_STAGE_WINDOWS windows of its 1024
// scalars (<= 8 k records) per tile: rank per coarse bin with LDS atomics, scan the 512 tile counts, lay the records out in bin order in LDS and store them
// from there -- consecutive lanes write the consecutive records of a bin (runs of ~16 = 128 bytes) and the next tile continues every run.  Scalar-major only.
static constexpr uint32_t COARSE_STAGE_WINDOWS = 8, COARSE_STAGE_TILE = SORT_THREADS * COARSE_STAGE_WINDOWS;
__global__ __launch_bounds__(SORT_THREADS) void k_sort_scatter_staged(SortJobs jobs, DigitArgs a, uint64_t per_wg, uint32_t nb) {
    const uint32_t* __restrict__ scalars = jobs.scalars[blockIdx.y];
    const uint32_t* __restrict__ base = jobs.wgcount[blockIdx.y];
    const uint32_t* __restrict__ offsets = jobs.offsets[blockIdx.y];
    uint2* __restrict__ out = jobs.sorted2[blockIdx.y];
    __shared__ uint32_t cur[SORT_FEW_BINS];          // the workgroup's cursor in every coarse bin
    __shared__ uint32_t tcnt[SORT_FEW_BINS];         // per tile: counts, then (count << 16 | exclusive offset inside the tile)
    __shared__ uint32_t wtot[SORT_THREADS / 64];
    __shared__ uint32_t tile_n;
    __shared__ uint2 stage[COARSE_STAGE_TILE];
    const uint32_t wg = blockIdx.x, t = threadIdx.x, lane = t & 63u, wv = t >> 6;
    for (uint32_t b = t; b < nb; b += SORT_THREADS) { cur[b] = offsets[b] + base[(uint64_t)wg * nb + b]; tcnt[b] = 0; }
    __syncthreads();
    const uint64_t lo = (uint64_t)wg * per_wg, hi = min(lo + per_wg, a.n);
    for (uint64_t i0 = lo; i0 < hi; i0 += SORT_THREADS) {                           // whole waves: the aggregated add ballots
        const uint64_t i = i0 + t;
        uint32_t sk[9];
        const bool live = i < hi && digits_prepare(scalars, i, a, sk);
        for (uint32_t j0 = 0; j0 < a.nw; j0 += COARSE_STAGE_WINDOWS) {
            uint32_t key[COARSE_STAGE_WINDOWS], val[COARSE_STAGE_WINDOWS], rk[COARSE_STAGE_WINDOWS];
            uint32_t okm = 0;
#pragma unroll
            for (uint32_t k = 0; k < COARSE_STAGE_WINDOWS; k++) {
                key[k] = 0; val[k] = 0;
                const bool ok = live && j0 + k < a.nw && digit_at(sk, i, j0 + k, a, key[k], val[k]);
                rk[k] = wave_aggregated_add(tcnt, ok, key[k] >> a.coarse_shift);
                okm |= (ok ? 1u : 0u) << k;
            }
            __syncthreads();
            {   // exclusive scan of the tile's counts: one thread per bin (nb <= 512 <= SORT_THREADS)
                const uint32_t c = t < nb ? tcnt[t] : 0;
                uint32_t incl = c;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t y = (uint32_t)__shfl_up((int)incl, d);
                    if (lane >= (uint32_t)d) incl += y;
                }
                if (lane == 63) wtot[wv] = incl;
                __syncthreads();
                uint32_t before = 0;
#pragma unroll
                for (uint32_t w = 0; w < SORT_THREADS / 64; w++) before += w < wv ? wtot[w] : 0u;
                if (t < nb) tcnt[t] = (before + incl - c) | (c << 16);          // offset < 8192 (13 bits) | count <= 8192 (14 bits)
                if (t == SORT_THREADS - 1) tile_n = before + incl;
            }
            __syncthreads();
#pragma unroll
            for (uint32_t k = 0; k < COARSE_STAGE_WINDOWS; k++)
                if (okm >> k & 1u) stage[(tcnt[key[k] >> a.coarse_shift] & 0xffffu) + rk[k]] = make_uint2(key[k], val[k]);
            __syncthreads();
            const uint32_t tn = tile_n;
#pragma unroll
            for (uint32_t k = 0; k < COARSE_STAGE_WINDOWS; k++) {          // slot q holds record number (q - offset) of its bin in this tile
                const uint32_t q = k * SORT_THREADS + t;
                if (q < tn) {
                    const uint2 r = stage[q];
                    const uint32_t bin = r.x >> a.coarse_shift;
                    out[cur[bin] + q - (tcnt[bin] & 0xffffu)] = r;
                }
            }
            __syncthreads();
            if (t < nb) { cur[t] += tcnt[t] >> 16; tcnt[t] = 0; }
            __syncthreads();
        }
    }
}
// Column scan of the [workgroup][bucket] count matrix: one lane per bucket walks down the workgroups
// (row-coalesced), turning counts into each workgroup's exclusive rank inside the bucket and leaving
// the bucket totals, which the single-workgroup k_scan below turns into bucket offsets.
__global__ void k_sort_colscan(SortJobs jobs, uint32_t nb, uint32_t nwg) {
    uint32_t* __restrict__ cnt = jobs.wgcount[blockIdx.y];
    uint32_t* __restrict__ totals = jobs.counts[blockIdx.y];
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    uint32_t run = 0;
    for (uint32_t wg0 = 0; wg0 < nwg; wg0 += 8) {          // eight independent loads in flight per lane: the walk is latency, not bandwidth
        uint32_t x[8];
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) x[j] = wg0 + j < nwg ? cnt[(uint64_t)(wg0 + j) * nb + b] : 0u;
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            if (wg0 + j < nwg) cnt[(uint64_t)(wg0 + j) * nb + b] = run;
            run += x[j];
        }
    }
    totals[b] = run;
}

// single workgroup: offsets[k] = sum_{q<k} counts[q], offsets[nb] = total; cursor = offsets.
// The usual geometries (2^12 .. 2^15 counters, a multiple of 4096) run in tiles of 4096: every thread holds one 16-byte vector of each tile
// (up to eight COALESCED loads issued back to back), a tile is scanned with wave shuffles and sixteen wave totals in LDS, and both outputs
// leave as coalesced 16-byte stores.  The general path below walks a contiguous chunk per thread with dependent-latency scalar loads (one
// HBM round trip per counter: 120-150 us per launch on a 2^16 proof's critical path).
__global__ __launch_bounds__(1024) void k_scan(SortJobs jobs, uint32_t nb) {
    const uint32_t* __restrict__ counts = jobs.counts[blockIdx.x];
    uint32_t* __restrict__ offsets = jobs.offsets[blockIdx.x];
    uint32_t* __restrict__ cursor = jobs.cursor[blockIdx.x];
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x;
    if (nb <= 32768 && (nb & 4095) == 0) {
        const uint32_t tiles = nb >> 12, lane = t & 63u, wv = t >> 6;
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(counts);
        uint4* __restrict__ o1 = reinterpret_cast<uint4*>(offsets);
        uint4* __restrict__ o2 = reinterpret_cast<uint4*>(cursor);
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = (uint32_t)j < tiles ? src[t + 1024u * j] : make_uint4(0, 0, 0, 0);
        uint32_t carry = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if ((uint32_t)j >= tiles) break;                       // block-uniform
            const uint32_t s4 = v[j].x + v[j].y + v[j].z + v[j].w;
            uint32_t incl = s4;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = (uint32_t)__shfl_up((int)incl, d);
                if (lane >= (uint32_t)d) incl += y;
            }
            if (lane == 63) part[wv] = incl;
            __syncthreads();
            uint32_t before = 0, tot = 0;
#pragma unroll
            for (uint32_t w = 0; w < 16; w++) {
                const uint32_t y = part[w];
                before += w < wv ? y : 0u;
                tot += y;
            }
            __syncthreads();
            uint32_t run = carry + before + incl - s4;
            uint4 o;
            o.x = run; run += v[j].x;
            o.y = run; run += v[j].y;
            o.z = run; run += v[j].z;
            o.w = run;
            o1[t + 1024u * j] = o;
            o2[t + 1024u * j] = o;
            carry += tot;
        }
        if (t == 0) offsets[nb] = carry;
        return;
    }
    const uint32_t per = (nb + 1023) / 1024;
    const uint32_t lo = t * per, hi = min(lo + per, nb);
    uint32_t s = 0;
    for (uint32_t k = lo; k < hi; k++) s += counts[k];
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - s;
    for (uint32_t k = lo; k < hi; k++) {
        offsets[k] = run;
        cursor[k] = run;
        run += counts[k];
    }
    if (t == 1023) offsets[nb] = part[1023];
}

// ---- two-level sort, level 2: ONE workgroup per coarse bin.  The bin's records are consecutive (level 1); the workgroup counts
// their fine bucket bits in LDS, scans the counts (these ARE the final bucket offsets: bin start + exclusive prefix -- no global
// atomic, no separate scan launch), and scatters the references with LDS cursors.  A bin that swallowed a skewed share of the
// digits (boolean-heavy witnesses) is simply a longer loop for its workgroup: the per-record work is a few instructions.
// (scalar, window) pairs per coarse bin from which the second level stages its scatter through LDS (k_sort_fine_staged): measured on the pipelined
// prover -- 2^16 constraints (2 k pairs per bin, a quarter of a tile) -1.3 %, 2^18 (8 k) -0.6 %, 2^20 (c = 20: 27 k) +0.1 % with the lone proof 0.4 ms shorter and
// the sort's un-overlapped time 1.95 -> 1.50 ms, 2^22 (106 k) +1.1 % and 8.0 -> 6.0 ms
static constexpr uint64_t SORT_FINE_STAGED_MIN = 16384;
__global__ __launch_bounds__(SORT_THREADS) void k_sort_fine(SortJobs jobs, uint32_t fine_bits, uint32_t nbins, uint32_t nb) {
    const uint2* __restrict__ rec = jobs.sorted2[blockIdx.y];
    const uint32_t* __restrict__ coff = jobs.cursor[blockIdx.y];          // coarse offsets (level 1 left them in its cursor array)
    uint32_t* __restrict__ offsets = jobs.offsets[blockIdx.y];
    uint32_t* __restrict__ sorted = jobs.sorted[blockIdx.y];
    __shared__ uint32_t cnt[SORT_MAX_FINE];
    __shared__ uint32_t part[SORT_THREADS];
    const uint32_t bin = blockIdx.x, t = threadIdx.x, nf = 1u << fine_bits, fm = nf - 1;
    const uint32_t lo = coff[bin], hi = coff[bin + 1];
    for (uint32_t f = t; f < nf; f += SORT_THREADS) cnt[f] = 0;
    __syncthreads();
    // (both passes fetch the record of the NEXT round before ranking the current one: a round is otherwise one HBM round trip long)
    uint32_t kn = lo + t < hi ? rec[lo + t].x : 0;
    for (uint32_t base = lo; base < hi; base += SORT_THREADS) {          // whole waves keep the ballots of wave_aggregated_add valid
        const uint32_t i = base + t;
        const bool ok = i < hi;
        const uint32_t key = kn & fm;
        const uint32_t in = i + SORT_THREADS;
        kn = in < hi ? rec[in].x : 0;
        (void)wave_aggregated_add(cnt, ok, ok ? key : 0);
    }
    __syncthreads();
    // exclusive scan of cnt[0..nf): every thread owns nf / SORT_THREADS consecutive counters (1..4)
    const uint32_t per = (nf + SORT_THREADS - 1) / SORT_THREADS;
    uint32_t s = 0;
    for (uint32_t k = 0; k < per; k++) { const uint32_t f = t * per + k; if (f < nf) s += cnt[f]; }
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < SORT_THREADS; d <<= 1) {
        const uint32_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = lo + part[t] - s;
    for (uint32_t k = 0; k < per; k++) {
        const uint32_t f = t * per + k;
        if (f < nf) {
            const uint32_t c = cnt[f];
            offsets[(uint64_t)bin * nf + f] = run;
            cnt[f] = run;                                  // becomes the bucket's cursor
            run += c;
        }
    }
    if (bin == nbins - 1 && t == 0) offsets[nb] = hi;
    __syncthreads();
    uint2 rn = lo + t < hi ? rec[lo + t] : make_uint2(0, 0);
    for (uint32_t base = lo; base < hi; base += SORT_THREADS) {
        const uint32_t i = base + t;
        const bool ok = i < hi;
        const uint2 r = rn;
        const uint32_t in = i + SORT_THREADS;
        rn = in < hi ? rec[in] : make_uint2(0, 0);
        const uint32_t pos = wave_aggregated_add(cnt, ok, r.x & fm);
        if (ok) sorted[pos] = r.y;
    }
}

// The same level-2 pass with its scatter STAGED through LDS (round 4; ZK_SORT_FINE_STAGED): the plain form stores every 4-byte reference on its own,
// scattered over the bin's ~130 KB output range -- the counters see 1.53 GB written per 2^20 proof for 272 MB of references (lines leave the L2 half
// written).  Here a workgroup takes its bin in tiles of 8 k records, ranks the tile's records per fine bucket (LDS atomics), scans the tile's counts,
// lays the references out in bucket order in LDS and writes them from there: consecutive threads store the consecutive references of a bucket (runs of
// ~8 = 32 bytes at 2^20 and 2^22, where c = 20 leaves 1024 fine buckets per bin, and the next tile continues every run).  Counters at 2^20: this kernel's writes 1.53 GB -> 0.51 GB per proof, all
// kernels' 5.50 -> 4.47 GB (profiles/r04_sort_fine_staged.txt).  Counting pass, scan and offsets as in k_sort_fine.
static constexpr uint32_t SORT_TILE_PER_THREAD = 8, SORT_TILE = SORT_THREADS * SORT_TILE_PER_THREAD;
static inline size_t sort_fine_staged_lds(uint32_t fine_bits) { return 4 * ((size_t)2 << fine_bits); }          // cnt + tcnt
__global__ __launch_bounds__(SORT_THREADS) void k_sort_fine_staged(SortJobs jobs, uint32_t fine_bits, uint32_t nbins, uint32_t nb) {
    const uint2* __restrict__ rec = jobs.sorted2[blockIdx.y];
    const uint32_t* __restrict__ coff = jobs.cursor[blockIdx.y];
    uint32_t* __restrict__ offsets = jobs.offsets[blockIdx.y];
    uint32_t* __restrict__ sorted = jobs.sorted[blockIdx.y];
    extern __shared__ uint32_t sort_dyn[];
    const uint32_t bin = blockIdx.x, t = threadIdx.x, nf = 1u << fine_bits, fm = nf - 1, lane = t & 63u, wv = t >> 6;
    uint32_t* cnt = sort_dyn;                        // pass 1: counts; then every bucket's cursor in the output
    uint32_t* tcnt = sort_dyn + nf;                  // per tile: counts, then (count << 16 | exclusive offset inside the tile)
    __shared__ uint32_t part[SORT_THREADS];
    __shared__ uint32_t st_val[SORT_TILE];           // the tile's references in bucket order ...
    __shared__ uint16_t st_key[SORT_TILE];           // ... and the bucket of each
    const uint32_t lo = coff[bin], hi = coff[bin + 1];
    for (uint32_t f = t; f < nf; f += SORT_THREADS) { cnt[f] = 0; tcnt[f] = 0; }
    __syncthreads();
    uint32_t kn = lo + t < hi ? rec[lo + t].x : 0;
    for (uint32_t base = lo; base < hi; base += SORT_THREADS) {
        const uint32_t i = base + t;
        const bool ok = i < hi;
        const uint32_t key = kn & fm;
        const uint32_t in = i + SORT_THREADS;
        kn = in < hi ? rec[in].x : 0;
        (void)wave_aggregated_add(cnt, ok, ok ? key : 0);
    }
    __syncthreads();
    const uint32_t per = (nf + SORT_THREADS - 1) / SORT_THREADS;          // <= 4 (SORT_MAX_FINE)
    {
        uint32_t s = 0;
        for (uint32_t k = 0; k < per; k++) { const uint32_t f = t * per + k; if (f < nf) s += cnt[f]; }
        part[t] = s;
        __syncthreads();
        for (uint32_t d = 1; d < SORT_THREADS; d <<= 1) {
            const uint32_t v = t >= d ? part[t - d] : 0;
            __syncthreads();
            part[t] += v;
            __syncthreads();
        }
        uint32_t run = lo + part[t] - s;
        for (uint32_t k = 0; k < per; k++) {
            const uint32_t f = t * per + k;
            if (f < nf) {
                const uint32_t c = cnt[f];
                offsets[(uint64_t)bin * nf + f] = run;
                cnt[f] = run;
                run += c;
            }
        }
        if (bin == nbins - 1 && t == 0) offsets[nb] = hi;
    }
    __syncthreads();
    for (uint32_t base = lo; base < hi; base += SORT_TILE) {
        const uint32_t tile = hi - base < SORT_TILE ? hi - base : SORT_TILE;
        uint32_t f[SORT_TILE_PER_THREAD], v[SORT_TILE_PER_THREAD], rk[SORT_TILE_PER_THREAD];
#pragma unroll
        for (uint32_t k = 0; k < SORT_TILE_PER_THREAD; k++) {          // whole waves: the aggregated add ballots
            const uint32_t j = k * SORT_THREADS + t;
            const bool ok = j < tile;
            const uint2 r = ok ? rec[base + j] : make_uint2(0, 0);
            f[k] = r.x & fm;
            v[k] = r.y;
            rk[k] = wave_aggregated_add(tcnt, ok, ok ? f[k] : 0);
        }
        __syncthreads();
        {   // exclusive scan of the tile's counts (each thread owns `per` consecutive buckets): wave shuffles + sixteen wave totals
            uint32_t c[4], s = 0;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                const uint32_t ff = t * per + k;
                c[k] = k < per && ff < nf ? tcnt[ff] : 0;
                s += c[k];
            }
            uint32_t incl = s;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = (uint32_t)__shfl_up((int)incl, d);
                if (lane >= (uint32_t)d) incl += y;
            }
            if (lane == 63) part[wv] = incl;
            __syncthreads();
            uint32_t before = 0;
#pragma unroll
            for (uint32_t w = 0; w < SORT_THREADS / 64; w++) before += w < wv ? part[w] : 0u;
            uint32_t run = before + incl - s;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                const uint32_t ff = t * per + k;
                if (k < per && ff < nf) {
                    tcnt[ff] = run | (c[k] << 16);          // offset in the tile (< 8192: 13 bits) | the tile's count of this bucket (<= 8192: 14 bits)
                    run += c[k];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < SORT_TILE_PER_THREAD; k++) {
            const uint32_t j = k * SORT_THREADS + t;
            if (j < tile) {
                const uint32_t slot = (tcnt[f[k]] & 0xffffu) + rk[k];
                st_val[slot] = v[k];
                st_key[slot] = (uint16_t)f[k];
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < SORT_TILE_PER_THREAD; k++) {          // slot j holds reference number (j - offset) of its bucket in this tile
            const uint32_t j = k * SORT_THREADS + t;
            if (j < tile) {
                const uint32_t ff = st_key[j];
                sorted[cnt[ff] + j - (tcnt[ff] & 0xffffu)] = st_val[j];
            }
        }
        __syncthreads();
        for (uint32_t ff = t; ff < nf; ff += SORT_THREADS) {
            cnt[ff] += tcnt[ff] >> 16;
            tcnt[ff] = 0;
        }
        __syncthreads();
    }
}

// ================================================================== host side
// Steps 1-3 for `count` MSMs over the same bases (their workspaces share the sort's geometry): digits, histogram, scan, scatter -- leaves the bucket
// offsets in ws[i]->offsets and the references, bucket by bucket, in ws[i]->sorted.  One chain of launches for all of them (blockIdx.y = job).
int msm_sort_launch(const MsmBases& b, MsmWorkspace* const* ws, const void* const* d_scalars, uint32_t count, hipStream_t s) {
    if (count == 0 || count > MAX_SORT_JOBS) ZK_FAIL(ZK_ERR_ARG, "msm_sort_launch: 1..4 MSMs per batch");
    SortJobs sj{};
    for (uint32_t i = 0; i < count; i++) {
        MsmWorkspace& w = *ws[i];
        if (w.c != b.c || w.precomp != b.precomp || w.curve != b.curve || w.cap_points < b.n || w.nbuckets != ws[0]->nbuckets || w.chunk != ws[0]->chunk ||
            w.sort_wgs != ws[0]->sort_wgs || w.sort_fine_bits != ws[0]->sort_fine_bits)
            ZK_FAIL(ZK_ERR_ARG, "msm: workspace does not match bases");
        sj.scalars[i] = (const uint32_t*)d_scalars[i];
        sj.counts[i] = w.counts.as<uint32_t>(); sj.offsets[i] = w.offsets.as<uint32_t>(); sj.cursor[i] = w.cursor.as<uint32_t>();
        sj.sorted[i] = w.sorted.as<uint32_t>(); sj.wgcount[i] = w.wgcount.as<uint32_t>(); sj.sorted2[i] = w.sorted2.as<uint2>();
    }
    MsmWorkspace& w = *ws[0];
    const uint32_t nbw = 1u << (b.c - 1);
    // scalar-major LDS sorts need ONE bucket set (resident keys: every window files into the same 2^(c-1) buckets); ZK_SORT_SCALAR_MAJOR=0 restores
    // the window-major ranges
    static const bool want_sm = !(::zk::opt("ZK_SORT_SCALAR_MAJOR") && atoi(::zk::opt("ZK_SORT_SCALAR_MAJOR")) == 0);
    const bool sm = want_sm && b.precomp && w.sort_wgs != 0;
#ifdef ZK_EXPERIMENTS
    static const uint32_t alias = (::zk::opt("ZK_EXPERIMENT_TABLE_ALIAS") && atoi(::zk::opt("ZK_EXPERIMENT_TABLE_ALIAS"))) ? 1u : 0u;
#else
    const uint32_t alias = 0u;
#endif
    DigitArgs da{b.n, b.c, b.nw, b.precomp ? 1u : 0u, nbw, {0, 0, 0, 0, 0, 0, 0, 0, 0}, b.ident.as<uint8_t>(),
                 0u, sm ? 1u : 0u, alias, b.fold ? 1u : 0u};
    for (uint32_t j = 0; j < b.nw; j++) {               // K += (2^(c-1) - 1) << (c*j)
        uint64_t v = ((uint64_t)1 << (b.c - 1)) - 1;
        uint32_t off = j * b.c, wd = off >> 5, sh = off & 31;
        unsigned __int128 add = (unsigned __int128)v << sh;
        uint64_t cy = 0;
        for (uint32_t k = wd; k < 9; k++) {
            cy += (uint64_t)da.K[k] + (uint32_t)(add & 0xffffffffu);
            da.K[k] = (uint32_t)cy;
            cy >>= 32;
            add >>= 32;
            if (!add && !cy) break;
        }
    }
    {
        ScopedTimer t("msm_sort", s);
        if (w.sort_fine_bits) {
            // level 1 over the coarse bins (its counts / offsets / cursor live in `coarse`), level 2 writes the real offsets and references
            const uint32_t bins = w.nbuckets >> w.sort_fine_bits;
            if (bins > SORT_FEW_BINS) ZK_FAIL(ZK_ERR_ARG, "two-level sort: more coarse bins than its first level reserves counters for");
            SortJobs l1 = sj;
            for (uint32_t i = 0; i < count; i++) {
                uint32_t* c3 = ws[i]->coarse.as<uint32_t>();
                l1.counts[i] = c3; l1.offsets[i] = c3 + (bins + 1); l1.cursor[i] = c3 + 2 * (bins + 1);
            }
            DigitArgs d1 = da;
            d1.coarse_shift = w.sort_fine_bits;
            const uint64_t total = sm ? b.n : b.n * b.nw, per_wg = (total + w.sort_wgs - 1) / w.sort_wgs;
            hipLaunchKernelGGL(k_sort_count_lds<SORT_FEW_BINS>, dim3(w.sort_wgs, count), dim3(SORT_THREADS), 0, s, l1, d1, per_wg, bins);
            dim3 gc = grid_for(bins, 256);
            gc.y = count;
            hipLaunchKernelGGL(k_sort_colscan, gc, dim3(256), 0, s, l1, bins, w.sort_wgs);
            hipLaunchKernelGGL(k_scan, dim3(count), dim3(1024), 0, s, l1, bins);
            const char* e_cs = ZK_FORM_ENV("ZK_SORT_COARSE_STAGED");          // a kernel-form switch (zk_common.h)
            const bool coarse_staged = sm && (e_cs ? atoi(e_cs) != 0 : SORT_COARSE_STAGED_DEFAULT);
            if (coarse_staged) hipLaunchKernelGGL(k_sort_scatter_staged, dim3(w.sort_wgs, count), dim3(SORT_THREADS), 0, s, l1, d1, per_wg, bins);
            else hipLaunchKernelGGL(k_sort_scatter_lds<SORT_FEW_BINS>, dim3(w.sort_wgs, count), dim3(SORT_THREADS), 0, s, l1, d1, per_wg, bins);
            SortJobs l2 = sj;
            for (uint32_t i = 0; i < count; i++) l2.cursor[i] = l1.offsets[i];          // the coarse offsets (k_scan wrote offsets = cursor; the scatter advanced neither: it ranks in LDS)
            const char* e_st = ZK_FORM_ENV("ZK_SORT_FINE_STAGED");          // a kernel-form switch (zk_common.h)
            bool staged = e_st ? atoi(e_st) != 0 : b.n * b.nw / bins >= SORT_FINE_STAGED_MIN;
            // its static LDS (part + the two staging tiles) plus 8 bytes per fine bucket must fit the device's per-workgroup limit (84 KB at c = 22; the
            // MI355X allows 160 KB): asked once per device, the plain form serves wherever it does not fit
            if (staged) {
                static int lds_limit[64] = {0};
                const int dev = ctx().device >= 0 && ctx().device < 64 ? ctx().device : 0;
                if (!lds_limit[dev]) {
                    int v = 0;
                    lds_limit[dev] = hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, ctx().device) == hipSuccess && v > 0 ? v : 65536;
                }
                const size_t need = sort_fine_staged_lds(w.sort_fine_bits) + 4 * (size_t)SORT_THREADS + 8 * (size_t)SORT_TILE;
                if (need > (size_t)lds_limit[dev]) staged = false;
            }
            if (staged) hipLaunchKernelGGL(k_sort_fine_staged, dim3(bins, count), dim3(SORT_THREADS), sort_fine_staged_lds(w.sort_fine_bits), s, l2, w.sort_fine_bits, bins, w.nbuckets);
            else hipLaunchKernelGGL(k_sort_fine, dim3(bins, count), dim3(SORT_THREADS), 0, s, l2, w.sort_fine_bits, bins, w.nbuckets);
        } else if (w.sort_wgs) {
            const uint64_t total = sm ? b.n : b.n * b.nw, per_wg = (total + w.sort_wgs - 1) / w.sort_wgs;
            hipLaunchKernelGGL(k_sort_count_lds<SORT_MAX_BUCKETS>, dim3(w.sort_wgs, count), dim3(SORT_THREADS), 0, s, sj, da, per_wg, w.nbuckets);
            dim3 gc = grid_for(w.nbuckets, 256);
            gc.y = count;
            hipLaunchKernelGGL(k_sort_colscan, gc, dim3(256), 0, s, sj, w.nbuckets, w.sort_wgs);
            hipLaunchKernelGGL(k_scan, dim3(count), dim3(1024), 0, s, sj, w.nbuckets);
            hipLaunchKernelGGL(k_sort_scatter_lds<SORT_MAX_BUCKETS>, dim3(w.sort_wgs, count), dim3(SORT_THREADS), 0, s, sj, da, per_wg, w.nbuckets);
        } else {
            for (uint32_t i = 0; i < count; i++) HIPCHK(hipMemsetAsync(ws[i]->counts.p, 0, 4 * (size_t)(w.nbuckets + 1), s));
            dim3 g = grid_for(b.n * b.nw, 256);
            g.y = count;
            hipLaunchKernelGGL(k_msm_count, g, dim3(256), 0, s, sj, da);
            hipLaunchKernelGGL(k_scan, dim3(count), dim3(1024), 0, s, sj, w.nbuckets);
            hipLaunchKernelGGL(k_msm_scatter, g, dim3(256), 0, s, sj, da);
        }
    }
    HIPCHK(hipGetLastError());
    return ZK_OK;
}

}  // namespace zk
