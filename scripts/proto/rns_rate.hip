// Probe (round 4): would the Fr-stage NTTs run faster over 31-bit NTT primes (residue number system, CRT back to Fr) than on the 9 x 29-bit Montgomery
// multiplier?  Measures modular products per second for one 31-bit prime in three forms -- Shoup (precomputed quotient of the constant factor: what a
// twiddle or a table entry is), Montgomery with 32-bit words, and a whole radix-2 butterfly (Shoup product + lazy add / sub in [0, 4p)) -- as
// dependent chains on every SIMD at 1..8 waves, plus the issue rates of the instructions involved and of v_fma_f64.
// An Fr product on 29-bit limbs is ~300 vector instructions; 18 primes of 31 bits cover the 532-bit integer convolution coefficients.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP 2048
static constexpr uint32_t P = 2013265921u;          // 15 * 2^27 + 1

__device__ __forceinline__ uint32_t shoup_mul(uint32_t x, uint32_t w, uint32_t wq) {      // x < 4p?  result in [0, 2p)
    const uint32_t q = __umulhi(x, wq);
    return x * w - q * P;
}
__device__ __forceinline__ uint32_t mont_mul(uint32_t a, uint32_t b, uint32_t pinv) {    // a b / 2^32 mod p, result in [0, 2p)
    const uint64_t t = (uint64_t)a * b;
    const uint32_t m = (uint32_t)t * pinv;
    const uint64_t u = t + (uint64_t)m * P;          // low word cancels
    return (uint32_t)(u >> 32);
}

template <int KIND> __global__ void k_chain(uint32_t* out, uint32_t seed) {
    uint32_t x[4], y[4];
    for (int i = 0; i < 4; i++) { x[i] = (threadIdx.x * 2654435761u + seed + i) % P; y[i] = (blockIdx.x * 40503u + 977u * i + 1) % P; }
    const uint32_t w = 1234567891u % P, wq = (uint32_t)(((uint64_t)w << 32) / P), pinv = 2013265919u;      // -p^-1 mod 2^32 for p = 15 * 2^27 + 1
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (KIND == 0) x[i] = shoup_mul(x[i], w, wq);
            else if (KIND == 1) x[i] = mont_mul(x[i], y[i], pinv);
            else if (KIND == 2) {          // DIF butterfly: (u, v) -> (u + v, (u - v) w), values kept in [0, 2p)
                uint32_t u = x[i], v = y[i];
                uint32_t s = u + v;
                s = s >= 2 * P ? s - 2 * P : s;
                const uint32_t d = u - v + 2 * P;
                x[i] = s;
                y[i] = shoup_mul(d, w, wq);
            } else if (KIND == 3) {
                double a = __uint_as_float(x[i]), b = __uint_as_float(y[i]);
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(b));
                x[i] = (uint32_t)__double2loint(a);
            }
        }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 4; i++) acc ^= x[i] ^ y[i];
    if (acc == 0x12345678u) out[0] = acc;
}
template <int KIND> __global__ void k_rate(uint32_t* out, uint32_t seed) {
    uint32_t a = threadIdx.x * 2654435761u + seed, b = blockIdx.x * 40503u + 12345u;
    uint32_t lo[8];
    double f[8];
    for (int i = 0; i < 8; i++) { lo[i] = a ^ i; f[i] = 1.0 + i + a; }
    const double fb = 1.0000001;
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (KIND == 0) asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(lo[i]) : "v"(a));
            else if (KIND == 1) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(lo[i]) : "v"(a));
            else if (KIND == 2) asm volatile("v_fma_f64 %0, %1, %0, %0" : "+v"(f[i]) : "v"(fb));
            else if (KIND == 3) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(lo[i]) : "v"(b));
            else if (KIND == 4) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(lo[i]) : "v"(b));
            else if (KIND == 5) asm volatile("v_min_u32 %0, %0, %1" : "+v"(lo[i]) : "v"(b));
        }
    }
    uint32_t x = 0;
    for (int i = 0; i < 8; i++) x ^= lo[i] ^ (uint32_t)__double2loint(f[i]);
    if (x == 0x12345678u) out[0] = x;
}

template <class K> static float timed(K launch) {
    launch();
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    uint32_t* d;
    hipMalloc(&d, 4096);
    const char* cn[3] = {"Shoup product (31-bit prime)", "Montgomery product (32-bit words)", "DIF butterfly (Shoup + lazy add/sub)"};
    for (int w = 1; w <= 8; w *= 2) {
        const int blocks = 256 * w, threads = 256;
        float ms[3];
        ms[0] = timed([&] { k_chain<0><<<blocks, threads>>>(d, 1); });
        ms[1] = timed([&] { k_chain<1><<<blocks, threads>>>(d, 1); });
        ms[2] = timed([&] { k_chain<2><<<blocks, threads>>>(d, 1); });
        for (int k = 0; k < 3; k++)
            printf("%-40s waves/SIMD=%d  %.3f ms  %.1f G ops/s\n", cn[k], w, ms[k], (double)REP * 4 * blocks * threads / (ms[k] * 1e-3) / 1e9);
    }
    const char* rn[6] = {"v_mul_hi_u32", "v_mul_lo_u32", "v_fma_f64", "v_mul_u32_u24", "v_sub_u32", "v_min_u32"};
    for (int w = 2; w <= 8; w *= 2) {
        const int blocks = 256 * w, threads = 256;
        float ms[6];
        ms[0] = timed([&] { k_rate<0><<<blocks, threads>>>(d, 1); });
        ms[1] = timed([&] { k_rate<1><<<blocks, threads>>>(d, 1); });
        ms[2] = timed([&] { k_rate<2><<<blocks, threads>>>(d, 1); });
        ms[3] = timed([&] { k_rate<3><<<blocks, threads>>>(d, 1); });
        ms[4] = timed([&] { k_rate<4><<<blocks, threads>>>(d, 1); });
        ms[5] = timed([&] { k_rate<5><<<blocks, threads>>>(d, 1); });
        for (int k = 0; k < 6; k++)
            printf("%-40s waves/SIMD=%d  %.3f ms  %.2f cycles/instr @2.4GHz\n", rn[k], w, ms[k], ms[k] * 1e-3 * 2.4e9 / ((double)REP * 8 * w));
    }
    return 0;
}
