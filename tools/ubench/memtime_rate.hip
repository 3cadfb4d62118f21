// What does s_memtime count?  One wave per SIMD runs a long chain of v_mfma_f32_16x16x4_f32 (as the kernels that are
// stamped do); s_memtime and s_memrealtime (the constant 100 MHz counter) are read before and after, and the host times
// the launch with events.  Prints ticks per microsecond of both counters and ticks per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k(float *out, unsigned long long *t, int iters)
{
    f32x4 acc[2] = {{0, 1, 2, 3}, {1, 2, 3, 4}};
    const float a = 0.001f * threadIdx.x, b = 1.0f + 0.002f * threadIdx.x;
    unsigned long long m0, m1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(m0), "=s"(r0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 64; ++m) acc[m & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 1], 0, 0, 0);
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]));
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(m1), "=s"(r1)::"memory");
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1];
    if (blockIdx.x == 0 && threadIdx.x == 0) { t[0] = m1 - m0; t[1] = r1 - r0; }
}

int main()
{
    float *out; unsigned long long *t, h[2];
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&t, 16);
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, out, t, iters);
        (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    const double us_real = h[1] / 100.0;      // s_memrealtime: 100 MHz
    printf("launch %.3f ms (events); wave: %.3f ms by s_memrealtime\n", ms, us_real / 1e3);
    printf("s_memtime: %.1f ticks per microsecond; %.2f ticks per MFMA (%d MFMAs); %.2f ns per MFMA\n", h[0] / us_real,
           (double)h[0] / (64.0 * iters), 64 * iters, us_real * 1e3 / (64.0 * iters));
    return 0;
}
