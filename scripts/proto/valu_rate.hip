// Issue-rate probe for the integer VALU instructions the field multiplier is built from (gfx950).
// Each kernel runs REP x UNROLL copies of one instruction pattern on 8 independent register sets
// per lane; all 256 CUs, 8 waves per SIMD.  Prints cycles per wave64 instruction per SIMD at the
// clock the run achieved (s_memtime is constant-rate, so the figure is derived from wall time and
// an assumed 2.4 GHz).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP 4096

template <int KIND> __global__ void k_rate(uint32_t* out, uint32_t seed) {
    uint32_t a = threadIdx.x * 2654435761u + seed, b = blockIdx.x * 40503u + 12345u;
    uint64_t acc[8];
    uint32_t lo[8], hi[8];
    for (int i = 0; i < 8; i++) { acc[i] = a + i; lo[i] = a ^ i; hi[i] = b + i; }
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (KIND == 0) {
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
            } else if (KIND == 1) {
                asm volatile("v_add_u32 %0, %1, %0" : "+v"(lo[i]) : "v"(a));
            } else if (KIND == 2) {
                asm volatile("v_add_co_u32 %0, vcc, %2, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo[i]), "+v"(hi[i]) : "v"(a) : "vcc");
            } else if (KIND == 3) {
                asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc[i]), "+v"(hi[i]) : "v"(a), "v"(b) : "vcc");
            } else if (KIND == 4) {
                asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(lo[i]) : "v"(a));
            } else if (KIND == 5) {
                asm volatile("v_lshrrev_b64 %0, 28, %0" : "+v"(acc[i]));
            } else if (KIND == 6) {
                asm volatile("v_alignbit_b32 %0, %1, %0, 28" : "+v"(lo[i]) : "v"(hi[i]));
            } else if (KIND == 7) {
                asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(lo[i]) : "v"(a));
            } else if (KIND == 8) {
                asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(lo[i]) : "v"(a), "v"(b));
            } else if (KIND == 9) {
                asm volatile("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(acc[i]) : "v"(acc[(i + 1) & 7]));
            } else if (KIND == 10) {
                asm volatile("v_and_b32 %0, 0xfffffff, %0" : "+v"(lo[i]));
            } else if (KIND == 11) {
                asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(lo[i]) : "v"(a), "v"(b));
            } else if (KIND == 12) {
                asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
            } else if (KIND == 13) {
                asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(lo[i]) : "v"(a));
            }
        }
    }
    uint32_t x = 0;
    for (int i = 0; i < 8; i++) x ^= (uint32_t)acc[i] ^ (uint32_t)(acc[i] >> 32) ^ lo[i] ^ hi[i];
    if (x == 0x12345678u) out[0] = x;
}

template <int KIND> void run(const char* name, int per_iter, uint32_t* d, int waves_per_simd) {
    const int threads = 256, blocks = 256 * waves_per_simd;   // 4 waves per block = 1 per SIMD
    k_rate<KIND><<<blocks, threads>>>(d, 1);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k_rate<KIND><<<blocks, threads>>>(d, 2);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)REP * 8 * per_iter * waves_per_simd;
    printf("%-34s waves/SIMD=%d  %.3f ms  %.2f cycles/instr @2.4GHz\n", name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 4096);
    for (int w = 1; w <= 8; w *= 2) {
        run<0>("v_mad_u64_u32", 1, d, w);
        run<1>("v_add_u32", 1, d, w);
        run<2>("v_add_co + v_addc_co (pair)", 2, d, w);
        run<3>("v_mad_u64_u32 + v_addc_co (pair)", 2, d, w);
        run<4>("v_mul_lo_u32", 1, d, w);
        run<5>("v_lshrrev_b64", 1, d, w);
        run<6>("v_alignbit_b32", 1, d, w);
        run<7>("v_mul_hi_u32", 1, d, w);
        run<8>("v_mad_u32_u24", 1, d, w);
        run<9>("v_lshl_add_u64", 1, d, w);
        run<10>("v_and_b32 literal", 1, d, w);
        run<11>("v_add3_u32", 1, d, w);
        run<12>("v_mad_i64_i32", 1, d, w);
        run<13>("v_pk_add_u16", 1, d, w);
    }
    return 0;
}
