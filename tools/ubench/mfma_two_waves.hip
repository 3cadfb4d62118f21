// Does the f32 matrix pipe lose issue slots when TWO waves of a SIMD feed it?  48 x v_mfma_f32_16x16x4_f32 per iteration
// on NACC accumulator chains, A and B operands resident in VGPRs, nothing else in the loop; 256-thread workgroups,
// 256 (one wave per SIMD), 512 or 768 (two / three waves per SIMD) of them.  Prints the time per MFMA and SIMD; the
// pipe's own rate is 32 cycles (13.3 ns at 2.4 GHz).  Build with -mllvm -amdgpu-mfma-vgpr-form.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256, 2) void k16(const float *w, float *out, int iters)
{
    float W[48];
#pragma unroll
    for (int i = 0; i < 48; ++i) W[i] = w[i * 64 + (threadIdx.x & 63)];
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){(float)i, 1, 2, 3};
    float b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) b[i] = 1.0f + threadIdx.x * 0.002f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 48; ++m) acc[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[m], b[m % 16], acc[m % NACC], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("" : "+v"(acc[i]));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][i & 3];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename K> void run(const char *name, K kern, int grid, const float *w, float *out)
{
    const int iters = 20000;
    // dynamic LDS sized so that exactly grid / 256 workgroups fit a CU (160 KB): an even spread, one wave of each per SIMD
    const int smem = grid == 256 ? 0 : grid == 512 ? 70 * 1024 : 50 * 1024;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, 0, w, out, iters);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, 0, w, out, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)iters * 48 * (grid / 256);      // MFMAs one SIMD executed
    printf("%-28s %d wave(s) per SIMD: %.3f ns per MFMA and SIMD (%.1f cycles at 2.4 GHz)\n", name, grid / 256, ms * 1e6 / per_simd,
           ms * 1e6 / per_simd * 2.4);
}

int main()
{
    std::vector<float> w(48 * 64, 0.001f);
    float *dw, *out; (void)hipMalloc(&dw, w.size() * 4); (void)hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&out, (size_t)1024 * 256 * 4);
    for (int g : {256, 512, 768}) {
        run("2 accumulator chains", k16<2>, g, dw, out);
        run("3 accumulator chains", k16<3>, g, dw, out);
        run("4 accumulator chains", k16<4>, g, dw, out);
    }
    return 0;
}
