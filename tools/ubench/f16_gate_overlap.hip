// Can one wave's gate block (vector work: 24 quarter-rate transcendentals + ~70 packed fp32 ops, the VALU half of a GRU
// step) run UNDER another wave's fp16 MFMA block (27 MFMAs: 18 x 16x16x16 + 9 x 16x16x32, the matrix half of the f16x3
// engine's step) when both waves sit on the same SIMD?  And does it take enforced anti-phase to get there?
//
// One 512-thread workgroup per CU = two 4-wave "stream groups", waves w and w + 4 on the same SIMD.  Modes:
//   0  group 1 alone: [MFMA block ; gate block] per iteration                      (one wave per SIMD: the B = 4096 shape)
//   1  both groups, each [MFMA ; gates], a barrier of its own 4 waves per iteration, free running (two independent
//      workgroups per CU behave like this: the B >= 8192 shape today)
//   2  both groups in ENFORCED anti-phase: two workgroup-wide barriers per iteration,
//      group 1: MFMA | gates,  group 2: gates | MFMA
//   3  MFMA blocks only (both groups)        4  gate blocks only (both groups)     -- the two floors
// Prints cycles per iteration and group-step.  Build: hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct State {
    f32x4 ar, an, az;
    f16x4 A4[6];
    f16x8 A8[3];
    f16x4 b4;
    f16x8 b8;
    f32x2 g[6], h[2];
};

__device__ __forceinline__ void mfma_block(State &s)
{
    // 18 K=16 MFMAs + 9 K=32 MFMAs on three accumulator chains, as the f16x3 engine issues them
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        s.ar = __builtin_amdgcn_mfma_f32_16x16x16f16(s.A4[i], s.b4, s.ar, 0, 0, 0);
        s.an = __builtin_amdgcn_mfma_f32_16x16x16f16(s.A4[(i + 1) % 6], s.b4, s.an, 0, 0, 0);
        s.az = __builtin_amdgcn_mfma_f32_16x16x16f16(s.A4[(i + 2) % 6], s.b4, s.az, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        s.ar = __builtin_amdgcn_mfma_f32_16x16x32_f16(s.A8[i], s.b8, s.ar, 0, 0, 0);
        s.an = __builtin_amdgcn_mfma_f32_16x16x32_f16(s.A8[(i + 1) % 3], s.b8, s.an, 0, 0, 0);
        s.az = __builtin_amdgcn_mfma_f32_16x16x32_f16(s.A8[(i + 2) % 3], s.b8, s.az, 0, 0, 0);
    }
    asm volatile("" : "+v"(s.ar), "+v"(s.an), "+v"(s.az));
}

#ifdef SCALAR_GATES
// the same gate block on plain (non-packed) fp32 ops: twice the instructions; build with -fno-slp-vectorize
__device__ __forceinline__ void gate_block(State &s)
{
    float *g = (float *)s.g, *h = (float *)s.h;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const float er = __builtin_amdgcn_exp2f(s.ar[v]) + 1.0f, ez = __builtin_amdgcn_exp2f(s.az[v]) + 1.0f;
        const float r = __builtin_amdgcn_rcpf(er), z = __builtin_amdgcn_rcpf(ez);
        const float pn = __builtin_fmaf(r, s.an[v], g[v]);
        const float en = __builtin_amdgcn_exp2f(pn) + 1.0f;
        const float n = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(en), 1.0f);
        h[v] = __builtin_fmaf(z, h[v] - n, n);
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) g[i] = __builtin_fmaf(g[i], (i & 1) ? 1.001f : 0.999f, h[(2 * ((i >> 1) & 1)) + (i & 1)]);
    const f32x4 hv = {h[0], h[1], h[2], h[3]};
    const f16x4 hi = __builtin_convertvector(hv, f16x4);
    const f16x4 lo = __builtin_convertvector(hv - __builtin_convertvector(hi, f32x4), f16x4);
    s.b4 = hi + lo * (_Float16)0.001;
    s.b8 = __builtin_shufflevector(hi, lo, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
    for (int v = 0; v < 4; ++v) { s.ar[v] = g[v] * 1e-3f; s.an[v] = g[4 + v] * 1e-3f; s.az[v] = g[8 + v] * 1e-3f; }
}
#else
__device__ __forceinline__ void gate_block(State &s)
{
    // the gate block of gru_mfma2_kernel: 12 v_exp + 12 v_rcp and the packed fp32 ops around them, per lane 4 units
    const f32x2 one = {1.0f, 1.0f};
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        f32x2 ar = {s.ar[2 * p], s.ar[2 * p + 1]}, an = {s.an[2 * p], s.an[2 * p + 1]}, az = {s.az[2 * p], s.az[2 * p + 1]};
        f32x2 er = {__builtin_amdgcn_exp2f(ar[0]), __builtin_amdgcn_exp2f(ar[1])};
        f32x2 ez = {__builtin_amdgcn_exp2f(az[0]), __builtin_amdgcn_exp2f(az[1])};
        er += one; ez += one;
        const f32x2 r = {__builtin_amdgcn_rcpf(er[0]), __builtin_amdgcn_rcpf(er[1])};
        const f32x2 z = {__builtin_amdgcn_rcpf(ez[0]), __builtin_amdgcn_rcpf(ez[1])};
        f32x2 pn = __builtin_elementwise_fma(r, an, s.g[p]);
        f32x2 en = {__builtin_amdgcn_exp2f(pn[0]), __builtin_amdgcn_exp2f(pn[1])};
        en += one;
        const f32x2 rn = {__builtin_amdgcn_rcpf(en[0]), __builtin_amdgcn_rcpf(en[1])};
        const f32x2 n = __builtin_elementwise_fma((f32x2){-2.0f, -2.0f}, rn, one);
        s.h[p] = __builtin_elementwise_fma(z, s.h[p] - n, n);
    }
    // input terms of the next step, head partial, fp16 split of h: ~25 more packed / conversion ops
#pragma unroll
    for (int i = 0; i < 6; ++i) s.g[i] = __builtin_elementwise_fma(s.g[i], (f32x2){0.999f, 1.001f}, s.h[i & 1]);
    const f32x4 hv = {s.h[0][0], s.h[0][1], s.h[1][0], s.h[1][1]};
    const f16x4 hi = __builtin_convertvector(hv, f16x4);
    const f16x4 lo = __builtin_convertvector(hv - __builtin_convertvector(hi, f32x4), f16x4);
    s.b4 = hi + lo * (_Float16)0.001;
    s.b8 = __builtin_shufflevector(hi, lo, 0, 1, 2, 3, 4, 5, 6, 7);
    s.ar = (f32x4){s.g[0][0], s.g[0][1], s.g[1][0], s.g[1][1]} * 1e-3f;
    s.an = (f32x4){s.g[2][0], s.g[2][1], s.g[3][0], s.g[3][1]} * 1e-3f;
    s.az = (f32x4){s.g[4][0], s.g[4][1], s.g[5][0], s.g[5][1]} * 1e-3f;
}
#endif

__global__ __launch_bounds__(512, 1) void k(float *out, int iters, int mode)
{
    const int grp = threadIdx.x >> 8;          // waves 0-3: group 0, waves 4-7: group 1 (wave w and w+4 share a SIMD)
    State s;
    const float t = 0.001f * (threadIdx.x & 63);
    s.ar = s.an = s.az = (f32x4){t, t + 0.1f, t + 0.2f, t + 0.3f};
#pragma unroll
    for (int i = 0; i < 6; ++i) { s.A4[i] = (f16x4){(_Float16)(0.01f * i), (_Float16)t, (_Float16)0.02f, (_Float16)0.03f}; s.g[i] = (f32x2){t, -t}; }
#pragma unroll
    for (int i = 0; i < 3; ++i) s.A8[i] = __builtin_shufflevector(s.A4[i], s.A4[i + 3], 0, 1, 2, 3, 4, 5, 6, 7);
    s.b4 = s.A4[1]; s.b8 = s.A8[1]; s.h[0] = s.h[1] = (f32x2){0.1f, 0.2f};
    if (mode == 0 && grp == 1) return;
    for (int it = 0; it < iters; ++it) {
        if (mode == 0 || mode == 1) {
            mfma_block(s);
            gate_block(s);
            // a barrier among the group's own 4 waves only is not expressible; free running needs none for this question
        } else if (mode == 2) {
            if (grp == 0) mfma_block(s); else gate_block(s);
            __builtin_amdgcn_s_barrier();
            if (grp == 0) gate_block(s); else mfma_block(s);
            __builtin_amdgcn_s_barrier();
        } else if (mode == 3) {
            mfma_block(s);
            s.b4 = __builtin_convertvector(s.ar, f16x4);
        } else {
            gate_block(s);
        }
    }
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s.ar[0] + s.an[1] + s.az[2] + s.h[0][0] + s.g[3][1] + (float)s.b4[0];
}

int main()
{
    float *out; (void)hipMalloc(&out, (size_t)256 * 512 * 4);
    const int iters = 200000;
    const char *names[5] = {"one group alone [MFMA ; gates]", "two groups free running", "two groups, enforced anti-phase",
                            "MFMA blocks only, two groups", "gate blocks only, two groups"};
    for (int mode = 0; mode < 5; ++mode) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, 1000, mode);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, mode);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double ns_it = ms * 1e6 / iters;
        const int groups = mode == 0 ? 1 : 2;
        printf("mode %d %-34s %8.1f ns per iteration = %7.0f cycles at 2.4 GHz; per group-step %7.1f ns (%5.0f cycles)\n", mode, names[mode],
               ns_it, ns_it * 2.4, ns_it / groups, ns_it * 2.4 / groups);
    }
    return 0;
}
