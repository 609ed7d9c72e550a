// Calibration of rocprofv3's FETCH_SIZE on THIS library's gather pattern (MI355X_MICROARCH.md, HBM: "Other access widths are uncalibrated: calibrate on a
// known byte count in your own access pattern").  Every lane reads the 112 used bytes of ONE 128-byte record (seven 16-byte pieces) at a pseudo-random,
// non-repeating record index of a 4 GiB table (far beyond L2 + Infinity Cache): known traffic = records x 128 B (whole lines) -- or x 2 x 64 B sectors.
//   k_gather_regs   seven global_load_dwordx4 per lane         (round 2's register look-ahead)
//   k_gather_lds    seven global_load_lds_dwordx4 per lane     (round 3's LDS-DMA look-ahead)
//   k_stream        a plain coalesced 16-B-per-lane streaming read of the same number of bytes (the guide's reference case: FETCH_SIZE reads one half)
// Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./gather_calib`; the program prints the known byte counts per kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

static constexpr uint64_t REC = 128, TABLE_RECS = 1ull << 25, GATHERS = 1ull << 24;      // 4 GiB table, 2^24 gathers = 2 GiB of distinct lines

__device__ __forceinline__ uint64_t rec_index(uint64_t i) { return (i * 0x9E3779B97F4A7C15ull >> 17) & (TABLE_RECS - 1); }      // odd multiplier: a permutation of the low bits' range

__global__ void k_gather_regs(const uint8_t* __restrict__ table, uint32_t* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint4* src = reinterpret_cast<const uint4*>(table + REC * rec_index(i));
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < 7; j++) { const uint4 x = src[j]; acc ^= x.x ^ x.y ^ x.z ^ x.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_gather_lds(const uint8_t* __restrict__ table, uint32_t* out) {
    __shared__ uint4 buf[4][7][64];
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint8_t* src = table + REC * rec_index(i);
#pragma unroll
    for (int j = 0; j < 7; j++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 16 * j), (__attribute__((address_space(3))) void*)&buf[wave][j][0], 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < 7; j++) { const uint4 x = buf[wave][j][lane]; acc ^= x.x ^ x.y ^ x.z ^ x.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_stream(const uint8_t* __restrict__ table, uint32_t* out, uint64_t vecs) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < vecs; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 x = reinterpret_cast<const uint4*>(table)[i];
        acc ^= x.x ^ x.y ^ x.z ^ x.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    uint8_t* table;
    uint32_t* out;
    if (hipMalloc(&table, REC * TABLE_RECS) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(table, 1, REC * TABLE_RECS);
    hipDeviceSynchronize();
    k_gather_regs<<<(unsigned)(GATHERS / 256), 256>>>(table, out);
    hipDeviceSynchronize();
    k_gather_lds<<<(unsigned)(GATHERS / 256), 256>>>(table, out);
    hipDeviceSynchronize();
    const uint64_t vecs = GATHERS * 128 / 16;          // the same 2 GiB, streamed
    k_stream<<<4096, 256>>>(table, out, vecs);
    hipDeviceSynchronize();
    printf("known: %llu gathers of one 128-byte record each = %llu bytes of whole lines (%llu bytes used); k_stream reads %llu bytes\n", (unsigned long long)GATHERS,
           (unsigned long long)(GATHERS * 128), (unsigned long long)(GATHERS * 112), (unsigned long long)(vecs * 16));
    return 0;
}
