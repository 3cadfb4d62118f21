// Micro-benchmark (gfx950): issue interval of the f16 MFMA shapes, one wave per SIMD, 4 independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
template <int MODE, int NACC>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters)
{
    f32x4 acc[4] = {{0, 0, 0, 0}, {1, 1, 1, 1}, {2, 2, 2, 2}, {3, 3, 3, 3}};
    f16x4 a4 = {(_Float16)1, (_Float16)2, (_Float16)3, (_Float16)4};
    f16x8 a8 = {(_Float16)1, (_Float16)2, (_Float16)3, (_Float16)4, (_Float16)5, (_Float16)6, (_Float16)7, (_Float16)8};
    float af = threadIdx.x;
    unsigned long long t0, t1;
    STAMP(t0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            f32x4 &c = acc[m % NACC];
            if (MODE == 0) c = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, a4, c, 0, 0, 0);
            else if (MODE == 1) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, a8, c, 0, 0, 0);
            else c = __builtin_amdgcn_mfma_f32_16x16x4f32(af, af, c, 0, 0, 0);
        }
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    }
    STAMP(t1);
    out[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
template <int MODE, int NACC> void run(const char *name)
{
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 1024); (void)hipMalloc(&cyc, 32);
    const int it = 1000;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<MODE, NACC>), dim3(1), dim3(256), 0, 0, out, cyc, it);
    (void)hipDeviceSynchronize();
    unsigned long long h[4]; (void)hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
    printf("%-32s %d accumulators: %.1f cycles per MFMA\n", name, NACC, (double)h[0] / (it * 16.0));
}
int main()
{
    run<0, 4>("v_mfma_f32_16x16x16_f16"); run<0, 2>("v_mfma_f32_16x16x16_f16"); run<0, 1>("v_mfma_f32_16x16x16_f16");
    run<1, 4>("v_mfma_f32_16x16x32_f16"); run<1, 2>("v_mfma_f32_16x16x32_f16"); run<1, 1>("v_mfma_f32_16x16x32_f16");
    run<2, 4>("v_mfma_f32_16x16x4_f32"); run<2, 1>("v_mfma_f32_16x16x4_f32");
    return 0;
}
