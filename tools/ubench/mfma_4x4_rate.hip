// v_mfma_f32_4x4x1_16B_f32 issue rate and the clock the chip holds under it: 192 MFMAs per iteration on 3 accumulator
// chains (VGPR-form accumulators: build with -mllvm -amdgpu-mfma-vgpr-form), A operands from 192 resident VGPRs, B from a
// VGPR or (BA = 1) from AGPRs; one wave per SIMD on every CU (grid 1024 x 64) or a single wave (grid 1).
// Prints s_memtime cycles per MFMA, wall time per MFMA and the clock implied by the two.
// For comparison the same for 48 x v_mfma_f32_16x16x4_f32 (4 waves per workgroup, 256 workgroups).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int BA>
__global__ __launch_bounds__(64) void k4(const float *w, float *out, unsigned long long *cyc, int iters)
{
    float W[192];
#pragma unroll
    for (int i = 0; i < 192; ++i) W[i] = w[i * 64 + threadIdx.x];
    f32x4 acc[3] = {{0, 0, 0, 0}, {1, 1, 1, 1}, {2, 2, 2, 2}};
    float b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) b[i] = 1.0f + threadIdx.x * 0.002f + i;
    if (BA) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" : "+a"(b[i]));      // park the B operands in AGPRs
    }
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 192; ++m) acc[m % 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(W[m], b[(m / 3) & 15], acc[m % 3], 0, 0, 0);
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]));
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[blockIdx.x * 64 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

__global__ __launch_bounds__(256) void k16(const float *w, float *out, unsigned long long *cyc, int iters)
{
    float W[48];
#pragma unroll
    for (int i = 0; i < 48; ++i) W[i] = w[i * 64 + (threadIdx.x & 63)];
    f32x4 acc[3] = {{0, 0, 0, 0}, {1, 1, 1, 1}, {2, 2, 2, 2}};
    float b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) b[i] = 1.0f + threadIdx.x * 0.002f + i;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 48; ++m) acc[m % 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[m], b[m / 3], acc[m % 3], 0, 0, 0);
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]));
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <typename K> void run(const char *name, K kern, int grid, int block, int per_iter, const float *w)
{
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&cyc, 8);
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, w, out, cyc, iters);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, w, out, cyc, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * per_iter;
    printf("%-44s grid %4d: %.2f s_memtime ticks per MFMA, %.3f ns per MFMA  => %.3f ticks/ns\n", name, grid, h / n, ms * 1e6 / n,
           (h / n) / (ms * 1e6 / n));
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    std::vector<float> w(192 * 64, 0.001f);
    float *dw; (void)hipMalloc(&dw, w.size() * 4); (void)hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    run("4x4x1, B from VGPR", k4<0>, 1, 64, 192, dw);
    run("4x4x1, B from VGPR", k4<0>, 1024, 64, 192, dw);
    run("4x4x1, B from AGPR", k4<1>, 1, 64, 192, dw);
    run("4x4x1, B from AGPR", k4<1>, 1024, 64, 192, dw);
    run("16x16x4 (4 waves per workgroup)", k16, 1, 256, 48, dw);
    run("16x16x4 (4 waves per workgroup)", k16, 256, 256, 48, dw);
    return 0;
}
