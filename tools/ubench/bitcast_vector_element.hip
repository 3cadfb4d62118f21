// Minimal reproducer of a hipcc (ROCm 7.2, clang 20) miscompile met in round 3: __builtin_bit_cast(unsigned, v[i]) on an
// ELEMENT of an ext_vector_type value is folded to element 0 for every i.  `hipcc -O3 --offload-arch=gfx950 -S
// --cuda-device-only` of kernel `bad` emits ONE compare (of element 0 against dbits); kernel `good` (__float_as_uint)
// emits v_max_u32 / v_max3_u32 over all four.  Host side: runs both on {1, 2, 3, 400} with dbits = bits(100.0f):
// expected 0 (400 is not below 100), `bad` prints 1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void bad(const f32x4 *p, unsigned dbits, int *out)
{
    const f32x4 d = p[threadIdx.x];
    const unsigned b0 = __builtin_bit_cast(unsigned, d[0]), b1 = __builtin_bit_cast(unsigned, d[1]);
    const unsigned b2 = __builtin_bit_cast(unsigned, d[2]), b3 = __builtin_bit_cast(unsigned, d[3]);
    out[threadIdx.x] = max(max(b0, b1), max(b2, b3)) < dbits;
}
__global__ void good(const f32x4 *p, unsigned dbits, int *out)
{
    const f32x4 d = p[threadIdx.x];
    const unsigned b0 = __float_as_uint(d[0]), b1 = __float_as_uint(d[1]), b2 = __float_as_uint(d[2]), b3 = __float_as_uint(d[3]);
    out[threadIdx.x] = max(max(b0, b1), max(b2, b3)) < dbits;
}
int main()
{
    const float h[4] = {1.0f, 2.0f, 3.0f, 400.0f}, lim = 100.0f;
    unsigned dbits; memcpy(&dbits, &lim, 4);
    f32x4 *p; int *o, r[2];
    (void)hipMalloc(&p, 16); (void)hipMalloc(&o, 8); (void)hipMemcpy(p, h, 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(bad, dim3(1), dim3(1), 0, 0, p, dbits, o);
    hipLaunchKernelGGL(good, dim3(1), dim3(1), 0, 0, p, dbits, o + 1);
    (void)hipMemcpy(r, o, 8, hipMemcpyDeviceToHost);
    printf("all four below 100?  __builtin_bit_cast on elements: %d   __float_as_uint: %d   (expected 0)\n", r[0], r[1]);
    return 0;
}
