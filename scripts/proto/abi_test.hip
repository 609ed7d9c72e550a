#include <hip/hip_runtime.h>
#include <stdint.h>
struct F14 { uint32_t v[14]; };
typedef uint32_t u32x14 __attribute__((ext_vector_type(14)));
__device__ __noinline__ F14 f_struct(F14 a, F14 b) { F14 r; for (int i = 0; i < 14; i++) r.v[i] = a.v[i] * b.v[13 - i] + 1; return r; }
__device__ __noinline__ F14 f_ref(const F14& a, const F14& b) { F14 r; for (int i = 0; i < 14; i++) r.v[i] = a.v[i] * b.v[13 - i] + 1; return r; }
__global__ void k(uint32_t* o, const uint32_t* in) {
    F14 a, b;
    for (int i = 0; i < 14; i++) { a.v[i] = in[i + threadIdx.x]; b.v[i] = in[i + 100 + threadIdx.x]; }
    F14 r = f_struct(a, b);
    F14 q = f_ref(r, b);
    for (int i = 0; i < 14; i++) o[i * 64 + threadIdx.x] = r.v[i] ^ q.v[i];
}
