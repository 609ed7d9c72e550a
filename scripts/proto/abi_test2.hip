#include <hip/hip_runtime.h>
#include <stdint.h>
struct F14 { uint32_t v[14]; };
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __noinline__ F14 f_vec(u32x8 a0, u32x4 a1, u32x2 a2, u32x8 b0, u32x4 b1, u32x2 b2) {
    uint32_t a[14], b[14];
    for (int i = 0; i < 8; i++) { a[i] = a0[i]; b[i] = b0[i]; }
    for (int i = 0; i < 4; i++) { a[8 + i] = a1[i]; b[8 + i] = b1[i]; }
    for (int i = 0; i < 2; i++) { a[12 + i] = a2[i]; b[12 + i] = b2[i]; }
    F14 r; for (int i = 0; i < 14; i++) r.v[i] = a[i] * b[13 - i] + 1; return r; }
__device__ __forceinline__ F14 f_wrap(const F14& a, const F14& b) {
    u32x8 a0, b0; u32x4 a1, b1; u32x2 a2, b2;
    for (int i = 0; i < 8; i++) { a0[i] = a.v[i]; b0[i] = b.v[i]; }
    for (int i = 0; i < 4; i++) { a1[i] = a.v[8 + i]; b1[i] = b.v[8 + i]; }
    for (int i = 0; i < 2; i++) { a2[i] = a.v[12 + i]; b2[i] = b.v[12 + i]; }
    return f_vec(a0, a1, a2, b0, b1, b2);
}
__global__ void k(uint32_t* o, const uint32_t* in) {
    F14 a, b;
    for (int i = 0; i < 14; i++) { a.v[i] = in[i + threadIdx.x]; b.v[i] = in[i + 100 + threadIdx.x]; }
    F14 r = f_wrap(a, b);
    F14 q = f_wrap(r, b);
    for (int i = 0; i < 14; i++) o[i * 64 + threadIdx.x] = r.v[i] ^ q.v[i];
}
