// Micro-benchmark (gfx950): issue cost of v_exp_f32 / v_rcp_f32 alone and mixed with plain / packed VALU
// ops of the same wave.  Straight-line bodies of 128 ops (16 independent chains x 8 rounds) so loop
// overhead is negligible; one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
#define PIN16(a) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), \
                                   "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]))
// MODE 0: exp only   1: fma only   2: exp,fma alternating (2 ops per slot)   3: pk_fma only   4: exp, pk_fma alternating
//      5: exp,exp,fma,fma groups  6: rcp only   7: exp then rcp dependent pairs (sigmoid-like: exp,add,rcp)
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters)
{
    float e[16], f[16]; f32x2 p[16];
    for (int i = 0; i < 16; ++i) { e[i] = threadIdx.x * 1e-4f + i * 0.01f; f[i] = e[i] + 1; p[i] = (f32x2){e[i], f[i]}; }
    const float a = 0.999f, b = 0.001f;
    unsigned long long t0, t1;
    STAMP(t0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0 || MODE == 2 || MODE == 4) e[i] = __builtin_amdgcn_exp2f(e[i]);
                if (MODE == 1 || MODE == 2) f[i] = __builtin_fmaf(f[i], a, b);
                if (MODE == 3 || MODE == 4) p[i] = __builtin_elementwise_fma(p[i], (f32x2){a, a}, (f32x2){b, b});
                if (MODE == 5) { if (i & 2) f[i] = __builtin_fmaf(f[i], a, b); else e[i] = __builtin_amdgcn_exp2f(e[i]); }
                if (MODE == 6) e[i] = __builtin_amdgcn_rcpf(e[i]);
                if (MODE == 7) { if ((r % 3) == 0) e[i] = __builtin_amdgcn_exp2f(e[i]); else if ((r % 3) == 1) e[i] = e[i] + 1.0f; else e[i] = __builtin_amdgcn_rcpf(e[i]); }
                if (MODE == 2 || MODE == 4) asm volatile("" : "+v"(e[i]), "+v"(f[i]), "+v"(p[i]));
            }
            PIN16(e); PIN16(f); PIN16(p);
        }
    }
    STAMP(t1);
    float s = 0;
    for (int i = 0; i < 16; ++i) s += e[i] + f[i] + p[i][0] + p[i][1];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
template <int MODE> void run(const char *name, int nops)
{
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 1024); (void)hipMalloc(&cyc, 32);
    const int it = 500;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(256), 0, 0, out, cyc, it);
    (void)hipDeviceSynchronize();
    unsigned long long h[4]; (void)hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
    printf("%-46s %7.1f cycles per 128-slot body = %.2f per slot (%d VALU ops per slot)\n", name, (double)h[0] / it,
           (double)h[0] / it / 128.0, nops);
}
int main()
{
    run<1>("v_fma", 1); run<3>("v_pk_fma", 1); run<0>("v_exp", 1); run<6>("v_rcp", 1);
    run<2>("v_exp + v_fma alternating", 2); run<4>("v_exp + v_pk_fma alternating", 2);
    run<5>("exp,exp,fma,fma groups (1 op per slot)", 1); run<7>("exp round, add round, rcp round", 1);
    return 0;
}
