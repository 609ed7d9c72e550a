// Prototype: BLS12-381 Fp Montgomery product on 14 x 28-bit limbs (R = 2^392), plain C++:
// column sums of up to 28 products < 2^56 never overflow a 64-bit accumulator, so every partial
// product is ONE v_mad_u64_u32 and there are no carry instructions at all.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define W 28
#define L 14
#define MASK 0x0fffffffu
__device__ static const uint32_t P28[14] = {0xfffaaab, 0xfefffff, 0x3ffffb9, 0xfffeb15, 0x6241eab, 0xa0f6b0f, 0xf6730d2,
                                            0xf38512b, 0x4774b84, 0x4bacd76, 0xba7b643, 0xe69a4b1, 0x1ea397f, 0x1a011};
#define PINV 0xffcfffdu

__device__ __forceinline__ void mont_mul28(uint32_t* r, const uint32_t* a, const uint32_t* b) {
    uint64_t acc = 0;
    uint32_t m[L];
#pragma unroll
    for (int k = 0; k < L; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P28[k - i];
        m[k] = ((uint32_t)acc * PINV) & MASK;
        acc += (uint64_t)m[k] * P28[0];
        acc >>= W;
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; k++) {
#pragma unroll
        for (int i = k - L + 1; i < L; i++) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = k - L + 1; i < L; i++) acc += (uint64_t)m[i] * P28[k - i];
        r[k - L] = (uint32_t)acc & MASK;
        acc >>= W;
    }
    r[L - 1] = (uint32_t)acc;
}


// two independent chains per column: the a*b products and the m*p products
__device__ __forceinline__ void mont_mul28_2c(uint32_t* r, const uint32_t* a, const uint32_t* b) {
    uint64_t carry = 0;
    uint32_t m[L];
#pragma unroll
    for (int k = 0; k < L; k++) {
        uint64_t A = carry, B = 0;
#pragma unroll
        for (int i = 0; i <= k; i++) A += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) B += (uint64_t)m[i] * P28[k - i];
        uint64_t acc = A + B;
        m[k] = ((uint32_t)acc * PINV) & MASK;
        acc += (uint64_t)m[k] * P28[0];
        carry = acc >> W;
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; k++) {
        uint64_t A = carry, B = 0;
#pragma unroll
        for (int i = k - L + 1; i < L; i++) A += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = k - L + 1; i < L; i++) B += (uint64_t)m[i] * P28[k - i];
        uint64_t acc = A + B;
        r[k - L] = (uint32_t)acc & MASK;
        carry = acc >> W;
    }
    r[L - 1] = (uint32_t)carry;
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
        if (V) mont_mul28_2c(t, x, y); else mont_mul28(t, x, y);
#pragma unroll
        for (int i = 0; i < L; i++) x[i] = t[i];
        if (V) mont_mul28_2c(t, y, x); else mont_mul28(t, y, x);
#pragma unroll
        for (int i = 0; i < L; i++) y[i] = t[i];
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < L; i++) acc ^= x[i] ^ y[i];
    if (acc == 0x12345678u) out[0] = acc;
}

// one product for checking: out[14] = mont(a, b)
__global__ void k_one(uint32_t* out, const uint32_t* a, const uint32_t* b) {
    uint32_t x[L], y[L], t[L];
    for (int i = 0; i < L; i++) { x[i] = a[i]; y[i] = b[i]; }
    mont_mul28(t, x, y);
    for (int i = 0; i < L; i++) out[i] = t[i];
}

int main(int argc, char** argv) {
    uint32_t* d;
    hipMalloc(&d, 4096);
    // check vector from stdin-free constants: a = 3, b = 5 (plain limbs) -> 15 / R mod p, printed for the host to verify
    uint32_t ha[L] = {0}, hb[L] = {0}, hr[L];
    for (int i = 0; i < L; i++) { ha[i] = (0x9e3779b9u * (i + 1)) & MASK; hb[i] = (0x85ebca6bu * (i + 7)) & MASK; }
    ha[L - 1] &= 0xffff; hb[L - 1] &= 0xffff;
    uint32_t *da = d + 64, *db = d + 128;
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice);
    hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    k_one<<<1, 1>>>(d, da, db);
    hipMemcpy(hr, d, sizeof hr, hipMemcpyDeviceToHost);
    printf("a="); for (int i = 0; i < L; i++) printf("%x,", ha[i]);
    printf("\nb="); for (int i = 0; i < L; i++) printf("%x,", hb[i]);
    printf("\nr="); for (int i = 0; i < L; i++) printf("%x,", hr[i]);
    printf("\n");
    const uint32_t iters = 2000;
    // throughput of the dependent product chain against waves per SIMD (256 CUs x 4 SIMDs; 256-thread blocks = 4 waves)
    for (int V = 0; V < 2; V++)
    for (int wps : {1, 2, 3, 4, 8}) {
        const int blocks = 256 * wps, threads = 256;
        if (V) k_bench<1><<<blocks, threads>>>(d, 10); else k_bench<0><<<blocks, threads>>>(d, 10);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (V) k_bench<1><<<blocks, threads>>>(d, iters); else k_bench<0><<<blocks, threads>>>(d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double muls = 2.0 * iters * blocks * threads;
        printf("variant %d %2d waves/SIMD: %.3f ms, %.2f G mul/s\n", V, wps, ms, muls / ms / 1e6);
    }
    // single wave latency
    {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k_bench<0><<<1, 64>>>(d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("single wave: %.3f us / mul\n", ms * 1e3 / (2.0 * iters));
    }
    return 0;
}
