// Micro-benchmarks (gfx950): issue cost of VALU flavours alone, blocked MFMA/VALU mixes, LDS ops beside
// f32 MFMAs, and whether a VALU-only wave overlaps an MFMA-only wave on the same SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")

// MODE: 0 v_fma  1 v_pk_fma  2 v_exp  3 v_rcp  4 permlane32_swap  5 v_add  6 v_pk_add 7 v_pk_mul
template <int MODE>
__global__ __launch_bounds__(256) void valu_alone(float *out, unsigned long long *cyc, int iters)
{
    float v[8]; f32x2 p[8]; unsigned u[8];
    for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 0.01f + i; p[i] = (f32x2){v[i], v[i] + 1}; u[i] = threadIdx.x + i; }
    const float a = 1.0001f, b = 0.5f;
    unsigned long long t0, t1;
    STAMP(t0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (MODE == 0) v[k] = __builtin_fmaf(v[k], a, b);
            else if (MODE == 1) p[k] = __builtin_elementwise_fma(p[k], (f32x2){a, a}, (f32x2){b, b});
            else if (MODE == 2) v[k] = __builtin_amdgcn_exp2f(v[k]);
            else if (MODE == 3) v[k] = __builtin_amdgcn_rcpf(v[k]);
            else if (MODE == 4) { u32x2 r = __builtin_amdgcn_permlane32_swap(u[k], u[(k + 1) & 7], false, false); u[k] = r[0]; u[(k + 1) & 7] = r[1]; }
            else if (MODE == 5) v[k] = v[k] + a;
            else if (MODE == 6) p[k] = p[k] + (f32x2){a, b};
            else if (MODE == 7) p[k] = p[k] * (f32x2){a, a};
        }
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
        asm volatile("" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]));
        asm volatile("" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]));
    }
    STAMP(t1);
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i] + p[i][0] + p[i][1] + u[i];
    out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

// NM MFMAs back to back, then NV v_fma back to back (blocked mix)
template <int NM, int NV>
__global__ __launch_bounds__(256) void blocked(float *out, unsigned long long *cyc, int iters)
{
    f32x4 acc[4] = {{0, 0, 0, 0}, {1, 1, 1, 1}, {2, 2, 2, 2}, {3, 3, 3, 3}};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f, v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    unsigned long long t0, t1;
    STAMP(t0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 3], 0, 0, 0);
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], b, a);
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
    }
    STAMP(t1);
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + s;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

// 8 MFMAs with one LDS op (MODE 0: ds_read_b128, 1: s_barrier, 2: s_nop 0, 3: nothing) after each
template <int MODE>
__global__ __launch_bounds__(256) void mfma_lds(float *out, unsigned long long *cyc, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
    __syncthreads();
    f32x4 acc[4] = {{0, 0, 0, 0}, {1, 1, 1, 1}, {2, 2, 2, 2}, {3, 3, 3, 3}};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    f32x4 r = {0, 0, 0, 0};
    unsigned long long t0, t1;
    STAMP(t0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 3], 0, 0, 0);
            if (MODE == 0) { f32x4 t = *(volatile f32x4 *)&lds[((threadIdx.x + m * 64) & 1023) * 4]; r += t; }
            else if (MODE == 1) asm volatile("s_barrier" ::: "memory");
            else if (MODE == 2) asm volatile("s_nop 0");
            asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
        }
    }
    STAMP(t1);
    out[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + r[0] + r[1] + r[2] + r[3];
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

// 512 threads: waves 0-3 (one per SIMD) run MFMAs, waves 4-7 run v_fma; ROLE 0 both, 1 only MFMA waves, 2 only VALU
template <int ROLE>
__global__ __launch_bounds__(512) void two_waves(float *out, unsigned long long *cyc, int iters)
{
    const int wv = threadIdx.x >> 6;
    f32x4 acc[4] = {{0, 0, 0, 0}, {1, 1, 1, 1}, {2, 2, 2, 2}, {3, 3, 3, 3}};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f, v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    unsigned long long t0 = 0, t1 = 0;
    __syncthreads();
    STAMP(t0);
    if (wv < 4) {
        if (ROLE != 2)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m & 3], 0, 0, 0);
                asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
            }
    } else {
        if (ROLE != 1)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int k = 0; k < 64; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], b, a);
                asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
            }
    }
    STAMP(t1);
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + s;
    if ((threadIdx.x & 63) == 0) cyc[wv] = t1 - t0;
}

template <typename F> void go(const char *name, F launch, int nthreads, double per)
{
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 512 * 4); (void)hipMalloc(&cyc, 8 * 8);
    (void)hipMemset(cyc, 0, 64);
    for (int r = 0; r < 2; ++r) launch(out, cyc);
    (void)hipDeviceSynchronize();
    unsigned long long h[8]; (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("%-44s", name);
    for (int w = 0; w < nthreads / 64; ++w) printf(" %8.1f", (double)h[w] / per);
    printf("\n");
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    const int it = 2000;
#define VA(M, nm) go(nm " alone: cycles/instr per wave", [&](float *o, unsigned long long *c) { hipLaunchKernelGGL((valu_alone<M>), dim3(1), dim3(256), 0, 0, o, c, it); }, 256, it * 8.0)
    VA(0, "v_fma_f32"); VA(1, "v_pk_fma_f32"); VA(2, "v_exp_f32"); VA(3, "v_rcp_f32"); VA(4, "v_permlane32_swap");
    VA(5, "v_add_f32"); VA(6, "v_pk_add_f32"); VA(7, "v_pk_mul_f32");
#define BL(NM, NV) go("blocked " #NM " MFMA + " #NV " v_fma: cycles/iter", [&](float *o, unsigned long long *c) { hipLaunchKernelGGL((blocked<NM, NV>), dim3(1), dim3(256), 0, 0, o, c, it); }, 256, (double)it)
    BL(8, 0); BL(8, 8); BL(8, 16); BL(8, 32); BL(16, 32); BL(48, 64); BL(0, 32);
#define ML(M, nm) go("8 MFMA each followed by " nm ": cycles/MFMA", [&](float *o, unsigned long long *c) { hipLaunchKernelGGL((mfma_lds<M>), dim3(1), dim3(256), 0, 0, o, c, it); }, 256, it * 8.0)
    ML(3, "nothing"); ML(0, "ds_read_b128"); ML(1, "s_barrier"); ML(2, "s_nop 0");
#define TW(R, nm) go("2 waves/SIMD " nm ": cycles/iter (w0-3 MFMAx8, w4-7 v_fma x64)", [&](float *o, unsigned long long *c) { hipLaunchKernelGGL((two_waves<R>), dim3(1), dim3(512), 0, 0, o, c, it); }, 512, (double)it)
    TW(1, "MFMA only"); TW(2, "VALU only"); TW(0, "both");
    return 0;
}
