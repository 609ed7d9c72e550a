// How many kernels from different streams does the chip actually run at once?
// K streams, one tiny (1 workgroup) ~1 ms spinning kernel each: wall time = ceil(K / C) ms for a limit C.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <chrono>
__global__ void spin(unsigned long long ticks, unsigned* out) {
    unsigned long long t0 = wall_clock64();
    unsigned x = 0;
    while (wall_clock64() - t0 < ticks) x++;
    if (x == 0xffffffffu) out[0] = x;
}
int main() {
    unsigned* d; hipMalloc(&d, 64);
    int rate_khz = 0; hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    printf("wall clock %d kHz, GPU_MAX_HW_QUEUES=%s\n", rate_khz, getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(unset)");
    unsigned long long ticks = (unsigned long long)rate_khz;   // 1 ms
    for (int K : {1, 2, 4, 6, 8, 12, 16, 24, 32, 48}) {
        std::vector<hipStream_t> st(K);
        for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        for (auto& s : st) spin<<<1, 64, 0, s>>>(1000, d);
        hipDeviceSynchronize();
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, 0);
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (auto& s : st) spin<<<1, 64, 0, s>>>(ticks, d);
        hipDeviceSynchronize();
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("K=%2d streams x 1 ms kernel: %.2f ms\n", K, ms);
        for (auto& s : st) hipStreamDestroy(s);
    }
    return 0;
}
