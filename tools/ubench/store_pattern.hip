// Write rate of a [rows][32] fp32 array (128-byte rows) under the two thread -> row mappings of tcn_first_d1_kernel,
// no arithmetic: (A) a thread owns 8 consecutive rows (one store instruction of a wave = eight 128-byte pieces 1 KiB
// apart), (B) rows interleaved so that one store instruction of a wave = 1 KiB contiguous.  Also (C): like (A) with
// nontemporal stores.  34.4 GB per launch (4096 x 65536 rows).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, long rows_per_stream)
{
    const int c4 = threadIdx.x & 7;
    const long b = blockIdx.x, base = (long)blockIdx.y * 256;
    const int gA = (threadIdx.x >> 3) * 8;                                       // (A): rows gA + s
    const int gB = 64 * (threadIdx.x >> 6) + ((threadIdx.x >> 3) & 7);           // (B): rows gB + 8 s
    const f32x4 v = {(float)threadIdx.x, 1.0f, 2.0f, (float)blockIdx.y};
    float *ob = out + (b * rows_per_stream + base) * 32 + 4 * c4;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int r = MODE == 1 ? gB + 8 * s : gA + s;
        if (MODE == 2) __builtin_nontemporal_store(v, (f32x4 *)(ob + (long)r * 32));
        else *(f32x4 *)(ob + (long)r * 32) = v;
    }
}

int main()
{
    const long B = 4096, T = 65536;
    float *out; (void)hipMalloc(&out, B * T * 32 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char *names[3] = {"A consecutive rows per thread", "B interleaved (1 KiB per store)", "C = A, nontemporal"};
    for (int m = 0; m < 3; ++m)
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(B, T / 256), dim3(256), 0, 0, out, T);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(B, T / 256), dim3(256), 0, 0, out, T);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(B, T / 256), dim3(256), 0, 0, out, T);
            (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-34s %.2f ms  %.2f TB/s\n", names[m], ms, B * T * 128.0 / ms / 1e9);
        }
    return 0;
}
