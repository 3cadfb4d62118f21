// Micro-benchmark: does VALU work co-issue with v_mfma_f32_16x16x4_f32 on gfx950, or do they share
// the SIMD's FP32 datapath?  One wave per SIMD (256-thread block, 1 block), s_memtime around a loop of
// NM MFMAs with K independent VALU ops after each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int K, int MODE>   // MODE 0: v_fma_f32, 1: v_exp_f32, 2: bf16 mfma + v_fma
__global__ __launch_bounds__(256) void kern(float *out, unsigned long long *cyc, int iters)
{
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {1, 1, 1, 1}, acc2 = {2, 2, 2, 2}, acc3 = {3, 3, 3, 3};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    bf16x8 ab = {1, 2, 3, 4, 5, 6, 7, 8};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            f32x4 &acc = (m & 3) == 0 ? acc0 : (m & 3) == 1 ? acc1 : (m & 3) == 2 ? acc2 : acc3;
            if (MODE == 2) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (MODE == 1) v[k] = __builtin_amdgcn_exp2f(v[k]);
                else v[k] = __builtin_fmaf(v[k], b, a);
            }
            asm volatile("" : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3));
            asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[threadIdx.x] = acc0[0] + acc1[1] + acc2[2] + acc3[3] + s;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int K, int MODE> void run(const char *name)
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 4); hipMalloc(&cyc, 4 * 8);
    const int iters = 2000;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((kern<K, MODE>), dim3(1), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[4]; hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
    printf("%-28s K=%d : %.1f cycles per MFMA slot (wave0), %.1f (wave3)\n", name, K, (double)h[0] / (iters * 8.0),
           (double)h[3] / (iters * 8.0));
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<0, 0>("f32 mfma16x16x4 + v_fma"); run<2, 0>("f32 mfma16x16x4 + v_fma"); run<4, 0>("f32 mfma16x16x4 + v_fma");
    run<6, 0>("f32 mfma16x16x4 + v_fma"); run<8, 0>("f32 mfma16x16x4 + v_fma");
    run<2, 1>("f32 mfma16x16x4 + v_exp"); run<4, 1>("f32 mfma16x16x4 + v_exp");
    run<0, 2>("bf16 mfma16x16x32 + v_fma"); run<2, 2>("bf16 mfma16x16x32 + v_fma"); run<4, 2>("bf16 mfma16x16x32 + v_fma");
    return 0;
}
