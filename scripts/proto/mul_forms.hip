// Probe: which FORM of the 14 x 29-bit Montgomery product keeps the gfx950 multiplier busiest at the 1-3 waves per SIMD the EC kernels run at?
//   V0  product scanning, one 64-bit accumulator per column (the shipped form, ff.cuh fp_mul_limbs): 2(k+1) DEPENDENT multiply-adds per column
//   V1  operand scanning (row-wise, 15 lazily carried 64-bit column accumulators): the 28 multiply-adds of a row are independent
//   V2  product scanning with TWO accumulators per column (a b part | m p part), joined once per column
//   V3  product scanning with ONE chain per column forced by inline v_mad_u64_u32 (no join of the shifted carry: 26 instructions fewer per product)
// Prints G products/s for 1, 2, 3, 4, 6 waves per SIMD and the single-wave latency; checks that the forms agree.
// RESULT (round 3, profiles/r03_mul_forms.txt): as a bare dependent chain V1 / V2 reach 77 G products/s at 2 waves per SIMD where V0 reaches 70
// (78.5 for all three from 3 waves up) -- but swapped into the library (ff.cuh / fr29.cuh, A/B on one box: gpurun_out/r03c) the accumulate
// kernels, the proofs per second and even the library's own multiplier benchmark did not move (G1 accumulate 9.79 vs 9.83 ms at 2^20): inside
// a group addition the multiply-add chains are not what the SIMDs wait for.  The shipped form stays V0.
// V3: the compiler (LLVM's reassociation ranks the shifted carry highest and adds it LAST, hence the separate chain per column and its join) cannot be
// talked out of the join in C++, and every inline-asm block is followed by a wait state (s_nop 0: 759 of them in a loop of two products), so the
// forced chain is slower at every occupancy (45.7 / 70.6 / 76.8 G products/s at 1 / 2 / 6 waves per SIMD against 64.2 / 71.3 / 78.7).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define W 29
#define L 14
#define MASK 0x1fffffffu
// p = BLS12-381 base field modulus in 29-bit limbs, NINV = -p^-1 mod 2^29 (scripts/gen_fp29_consts.py)
__device__ static const uint32_t P29[14] = {0x1fffaaabu, 0x0ff7ffffu, 0x14ffffeeu, 0x17fffd62u, 0x0f6241eau, 0x09507b58u, 0x0afd9cc3u,
                                            0x109e70a2u, 0x1764774bu, 0x121a5d66u, 0x12c6e9edu, 0x12ffcd34u, 0x00111ea3u, 0x0000000du};
#define NINV 0x1ffcfffdu

__device__ __forceinline__ void mul_v0(uint32_t* r, const uint32_t* a, const uint32_t* b) {
    uint64_t acc = 0;
    uint32_t m[L];
#pragma unroll
    for (int k = 0; k < L; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P29[k - i];
        m[k] = ((uint32_t)acc * NINV) & MASK;
        acc += (uint64_t)m[k] * P29[0];
        acc >>= W;
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; k++) {
#pragma unroll
        for (int i = k - L + 1; i < L; i++) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = k - L + 1; i < L; i++) acc += (uint64_t)m[i] * P29[k - i];
        r[k - L] = (uint32_t)acc & MASK;
        acc >>= W;
    }
    r[L - 1] = (uint32_t)acc;
}
__device__ __forceinline__ void mul_v1(uint32_t* r, const uint32_t* a, const uint32_t* b) {
    uint64_t t[L + 1];
#pragma unroll
    for (int j = 0; j <= L; j++) t[j] = 0;
#pragma unroll
    for (int i = 0; i < L; i++) {
#pragma unroll
        for (int j = 0; j < L; j++) t[j] += (uint64_t)a[i] * b[j];
        const uint32_t m = ((uint32_t)t[0] * NINV) & MASK;
#pragma unroll
        for (int j = 0; j < L; j++) t[j] += (uint64_t)m * P29[j];
        t[1] += t[0] >> W;
#pragma unroll
        for (int j = 0; j < L; j++) t[j] = t[j + 1];
        t[L] = 0;
    }
#pragma unroll
    for (int k = 0; k < L - 1; k++) {
        r[k] = (uint32_t)t[k] & MASK;
        t[k + 1] += t[k] >> W;
    }
    r[L - 1] = (uint32_t)t[L - 1];
}
__device__ __forceinline__ void mul_v2(uint32_t* r, const uint32_t* a, const uint32_t* b) {
    uint64_t acc = 0;
    uint32_t m[L];
#pragma unroll
    for (int k = 0; k < L; k++) {
        uint64_t s1 = 0;
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) s1 += (uint64_t)m[i] * P29[k - i];
        acc += s1;
        m[k] = ((uint32_t)acc * NINV) & MASK;
        acc += (uint64_t)m[k] * P29[0];
        acc >>= W;
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; k++) {
        uint64_t s1 = 0;
#pragma unroll
        for (int i = k - L + 1; i < L; i++) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = k - L + 1; i < L; i++) s1 += (uint64_t)m[i] * P29[k - i];
        acc += s1;
        r[k - L] = (uint32_t)acc & MASK;
        acc >>= W;
    }
    r[L - 1] = (uint32_t)acc;
}

// V3  product scanning with ONE chain per column, forced: the multiply-adds are inline v_mad_u64_u32 (opaque to the reassociation that makes the
//     compiler start every column's sum at zero and JOIN it to the shifted carry with a v_lshl_add_u64: 26 instructions per product)
__device__ __forceinline__ uint64_t mad_vv(uint32_t a, uint32_t b, uint64_t c) {
    uint64_t cy;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(c), "=s"(cy) : "v"(a), "v"(b));
    return c;
}
__device__ __forceinline__ uint64_t mad_vs(uint32_t a, uint32_t b, uint64_t c) {     // b: a constant (scalar register)
    uint64_t cy;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(c), "=s"(cy) : "v"(a), "s"(b));
    return c;
}
__device__ __forceinline__ void mul_v3(uint32_t* r, const uint32_t* a, const uint32_t* b) {
    uint64_t acc = 0;
    uint32_t m[L];
#pragma unroll
    for (int k = 0; k < L; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc = mad_vv(a[i], b[k - i], acc);
#pragma unroll
        for (int i = 0; i < k; i++) acc = mad_vs(m[i], P29[k - i], acc);
        m[k] = ((uint32_t)acc * NINV) & MASK;
        acc = mad_vs(m[k], P29[0], acc);
        acc >>= W;
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; k++) {
#pragma unroll
        for (int i = k - L + 1; i < L; i++) acc = mad_vv(a[i], b[k - i], acc);
#pragma unroll
        for (int i = k - L + 1; i < L; i++) acc = mad_vs(m[i], P29[k - i], acc);
        r[k - L] = (uint32_t)acc & MASK;
        acc >>= W;
    }
    r[L - 1] = (uint32_t)acc;
}

template <int V> __device__ __forceinline__ void mul(uint32_t* r, const uint32_t* a, const uint32_t* b) {
    if (V == 0) mul_v0(r, a, b);
    else if (V == 1) mul_v1(r, a, b);
    else if (V == 3) mul_v3(r, a, b);
    else mul_v2(r, a, b);
}
template <int V> __global__ void k_bench(uint32_t* out, uint32_t iters) {
    uint32_t x[L], y[L], t[L];
#pragma unroll
    for (int i = 0; i < L; i++) {
        x[i] = (0x1234567u * (i + 1) ^ (threadIdx.x * 2654435761u >> (i & 7))) & MASK;
        y[i] = (0x7654321u * (i + 3) + blockIdx.x) & MASK;
    }
    x[L - 1] &= 0xffff;
    y[L - 1] &= 0xffff;
    for (uint32_t it = 0; it < iters; it++) {
        mul<V>(t, x, y);
#pragma unroll
        for (int i = 0; i < L; i++) x[i] = t[i];
        mul<V>(t, y, x);
#pragma unroll
        for (int i = 0; i < L; i++) y[i] = t[i];
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < L; i++) acc ^= x[i] ^ y[i];
    if (acc == 0x12345678u) out[0] = acc;
}
// the same dependent chain with the loop body UNROLLED to 2 * U products (U = 10: ~40 KB of straight-line code, the size of an inlined group addition):
// does a long loop body alone slow a wave down (instruction fetch)?
template <int V, int U> __global__ void k_bench_long(uint32_t* out, uint32_t iters) {
    uint32_t x[L], y[L], t[L];
#pragma unroll
    for (int i = 0; i < L; i++) {
        x[i] = (0x1234567u * (i + 1) ^ (threadIdx.x * 2654435761u >> (i & 7))) & MASK;
        y[i] = (0x7654321u * (i + 3) + blockIdx.x) & MASK;
    }
    x[L - 1] &= 0xffff;
    y[L - 1] &= 0xffff;
    for (uint32_t it = 0; it < iters; it += U) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            mul<V>(t, x, y);
#pragma unroll
            for (int i = 0; i < L; i++) x[i] = t[i];
            mul<V>(t, y, x);
#pragma unroll
            for (int i = 0; i < L; i++) y[i] = t[i];
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < L; i++) acc ^= x[i] ^ y[i];
    if (acc == 0x12345678u) out[0] = acc;
}
// two INDEPENDENT chains per lane (what an inlined group addition offers the scheduler: U2 | S2, PPP | Q, ZZ3 | ZZZ3)
template <int V> __global__ void k_bench2(uint32_t* out, uint32_t iters) {
    uint32_t x[L], y[L], u[L], v[L], t[L], s[L];
#pragma unroll
    for (int i = 0; i < L; i++) {
        x[i] = (0x1234567u * (i + 1) ^ (threadIdx.x * 2654435761u >> (i & 7))) & MASK;
        y[i] = (0x7654321u * (i + 3) + blockIdx.x) & MASK;
        u[i] = (x[i] * 3 + 1) & MASK;
        v[i] = (y[i] * 5 + 7) & MASK;
    }
    x[L - 1] &= 0xffff; y[L - 1] &= 0xffff; u[L - 1] &= 0xffff; v[L - 1] &= 0xffff;
    for (uint32_t it = 0; it < iters; it++) {
        mul<V>(t, x, y);
        mul<V>(s, u, v);
#pragma unroll
        for (int i = 0; i < L; i++) { x[i] = t[i]; u[i] = s[i]; }
        mul<V>(t, y, x);
        mul<V>(s, v, u);
#pragma unroll
        for (int i = 0; i < L; i++) { y[i] = t[i]; v[i] = s[i]; }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < L; i++) acc ^= x[i] ^ y[i] ^ u[i] ^ v[i];
    if (acc == 0x12345678u) out[0] = acc;
}
template <int V> __global__ void k_one(uint32_t* out, const uint32_t* a, const uint32_t* b) {
    uint32_t x[L], y[L], t[L];
    for (int i = 0; i < L; i++) { x[i] = a[i]; y[i] = b[i]; }
    mul<V>(t, x, y);
    for (int i = 0; i < L; i++) out[i] = t[i];
}

template <int V> static void run(uint32_t* d, const char* name) {
    const uint32_t iters = 2000;
    printf("%s\n", name);
    for (int wps : {1, 2, 3, 4, 6}) {
        const int blocks = 256 * wps, threads = 256;
        k_bench<V><<<blocks, threads>>>(d, 10);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k_bench<V><<<blocks, threads>>>(d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        hipEventRecord(e0);
        k_bench2<V><<<blocks, threads>>>(d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms2;
        hipEventElapsedTime(&ms2, e0, e1);
        printf("  %d waves/SIMD: one chain %.2f G mul/s | two independent chains per lane %.2f G mul/s\n", wps, 2.0 * iters * blocks * threads / ms / 1e6,
               4.0 * iters * blocks * threads / ms2 / 1e6);
    }
    for (int wps : {1, 2}) {
        const int blocks = 256 * wps, threads = 256;
        hipEvent_t a0, a1;
        hipEventCreate(&a0); hipEventCreate(&a1);
        k_bench_long<V, 10><<<blocks, threads>>>(d, 10);
        hipEventRecord(a0);
        k_bench_long<V, 10><<<blocks, threads>>>(d, iters);
        hipEventRecord(a1);
        hipEventSynchronize(a1);
        float msl;
        hipEventElapsedTime(&msl, a0, a1);
        printf("  %d waves/SIMD, loop body unrolled to 20 products (~40 KB): %.2f G mul/s\n", wps, 2.0 * iters * blocks * threads / msl / 1e6);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k_bench<V><<<1, 64>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("  single wave: %.3f us / mul\n", ms * 1e3 / (2.0 * iters));
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 4096);
    uint32_t ha[L], hb[L], r0[L], r1[L], r2[L], r3[L];
    for (int i = 0; i < L; i++) { ha[i] = (0x9e3779b9u * (i + 1)) & MASK; hb[i] = (0x85ebca6bu * (i + 7)) & MASK; }
    ha[L - 1] &= 0xffff; hb[L - 1] &= 0xffff;
    uint32_t *da = d + 64, *db = d + 128;
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice);
    hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    k_one<0><<<1, 1>>>(d, da, db); hipMemcpy(r0, d, sizeof r0, hipMemcpyDeviceToHost);
    k_one<1><<<1, 1>>>(d, da, db); hipMemcpy(r1, d, sizeof r1, hipMemcpyDeviceToHost);
    k_one<2><<<1, 1>>>(d, da, db); hipMemcpy(r2, d, sizeof r2, hipMemcpyDeviceToHost);
    k_one<3><<<1, 1>>>(d, da, db); hipMemcpy(r3, d, sizeof r3, hipMemcpyDeviceToHost);
    int same = 1;
    for (int i = 0; i < L; i++) same &= (r0[i] == r1[i]) & (r0[i] == r2[i]) & (r0[i] == r3[i]);
    printf("forms agree: %s\n", same ? "yes" : "NO");
    run<0>(d, "V0 product scanning, one accumulator (shipped)");
    run<1>(d, "V1 operand scanning, 15 lazily carried column accumulators");
    run<2>(d, "V2 product scanning, two accumulators per column");
    run<3>(d, "V3 product scanning, ONE chain per column forced with inline v_mad_u64_u32 (no join)");
    return same ? 0 : 1;
}
