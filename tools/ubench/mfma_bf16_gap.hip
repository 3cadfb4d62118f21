// Micro-benchmark (gfx950): what a vector instruction costs in the gap between two v_mfma_f32_16x16x32_bf16 of ONE wave --
// as a function of whether the second MFMA accumulates onto the first one's result (the GEMV chains of the bf16x3 GRU
// engine, csrc/gru_mfma2.hip step_b) or onto another accumulator.  One wave per SIMD, s_memtime around 1000 x 16 MFMAs,
// everything in one asm statement per iteration so that nothing is scheduled or padded by the compiler.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
#define M(D) "v_mfma_f32_16x16x32_bf16 " D ", %4, %5, " D "\n\t"
#define X16(A) A A A A A A A A A A A A A A A A
#define X8(A) A A A A A A A A
// NACC accumulators used round robin (1: one dependent chain, 2: every MFMA depends on the one before the previous, 4);
// FILL: 0 nothing, 1 v_exp_f32, 2 two (dependent) v_pk_add_f32, 3 v_exp_f32 + v_pk_add_f32, 4 s_nop 0, 5 s_nop 1, 6 one v_pk_add_f32,
// 7 two independent v_add_f32, 8 two independent v_fma_f32, 9 v_exp_f32 + v_add_f32, 10 v_cvt_pk_bf16_f32, 11 v_and_b32 + v_lshlrev_b32,
// 12 four independent v_add_f32, 13 v_rcp_f32 + v_exp_f32
template <int NACC, int FILL>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters)
{
    f32x4 a0 = {0, 0, 0, 0}, a1 = {1, 1, 1, 1}, a2 = {2, 2, 2, 2}, a3 = {3, 3, 3, 3};
    bf16x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (__bf16)(0.001f * (threadIdx.x + i)); B[i] = (__bf16)(0.002f * (threadIdx.x + 2 * i)); }
    float e = 0.001f * threadIdx.x, e2 = e + 1.0f, e3 = e + 2.0f, e4 = e + 3.0f;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 p = {e, e};
    unsigned long long t0, t1;
    STAMP(t0);
    for (int it = 0; it < iters; ++it) {
#define F0 ""
#define F1 "v_exp_f32 %6, %6\n\t"
#define F2 "v_pk_add_f32 %7, %7, 1.0 op_sel_hi:[1,0]\n\tv_pk_add_f32 %7, %7, 1.0 op_sel_hi:[1,0]\n\t"
#define F3 "v_exp_f32 %6, %6\n\tv_pk_add_f32 %7, %7, 1.0 op_sel_hi:[1,0]\n\t"
#define F4 "s_nop 0\n\t"
#define F5 "s_nop 1\n\t"
#define F6 "v_pk_add_f32 %7, %7, 1.0 op_sel_hi:[1,0]\n\t"
#define F7 "v_add_f32 %6, 1.0, %6\n\tv_add_f32 %8, 1.0, %8\n\t"
#define F8 "v_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %9, %8, %8, %6\n\t"
#define F9 "v_exp_f32 %6, %6\n\tv_add_f32 %8, 1.0, %8\n\t"
#define F10 "v_cvt_pk_bf16_f32 %8, %6, %9\n\t"
#define F11 "v_and_b32 %8, 0xffff0000, %6\n\tv_lshlrev_b32 %9, 16, %6\n\t"
#define F12 "v_add_f32 %6, 1.0, %6\n\tv_add_f32 %8, 1.0, %8\n\tv_add_f32 %9, 1.0, %9\n\tv_add_f32 %10, 1.0, %10\n\t"
#define F13 "v_rcp_f32 %6, %6\n\tv_exp_f32 %8, %8\n\t"
#define F14 "v_cvt_pk_bf16_f32 %8, %6, %9\n\tv_cvt_pk_bf16_f32 %10, %9, %6\n\t"
#define F15 "v_sub_f32 %8, %6, %9\n\tv_sub_f32 %10, %9, %6\n\t"
#define BODY(F)                                                                                                          \
        if (NACC == 0) asm volatile(X16(F) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(A), "v"(B), "v"(e), "v"(p), "v"(e2), "v"(e3), "v"(e4));       \
        else if (NACC == 1) asm volatile(X16(M("%0") F) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(A), "v"(B), "v"(e), "v"(p), "v"(e2), "v"(e3), "v"(e4));       \
        else if (NACC == 2) asm volatile(X8(M("%0") F M("%1") F) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(A), "v"(B), "v"(e), "v"(p), "v"(e2), "v"(e3), "v"(e4)); \
        else asm volatile(X8(M("%0") F M("%1") F) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(A), "v"(B), "v"(e), "v"(p), "v"(e2), "v"(e3), "v"(e4));
        if (FILL == 0) { BODY(F0) } else if (FILL == 1) { BODY(F1) } else if (FILL == 2) { BODY(F2) }
        else if (FILL == 3) { BODY(F3) } else if (FILL == 4) { BODY(F4) } else if (FILL == 5) { BODY(F5) } else if (FILL == 6) { BODY(F6) }
        else if (FILL == 7) { BODY(F7) } else if (FILL == 8) { BODY(F8) } else if (FILL == 9) { BODY(F9) } else if (FILL == 10) { BODY(F10) }
        else if (FILL == 11) { BODY(F11) } else if (FILL == 12) { BODY(F12) } else if (FILL == 13) { BODY(F13) } else if (FILL == 14) { BODY(F14) } else { BODY(F15) }
    }
    STAMP(t1);
    out[threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + e + p[0] + e2 + e3 + e4;
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
template <int NACC, int FILL> void run(const char *what)
{
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 1024); (void)hipMalloc(&cyc, 32);
    const int it = 1000;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<NACC, FILL>), dim3(1), dim3(256), 0, 0, out, cyc, it);
    (void)hipDeviceSynchronize();
    unsigned long long h[4]; (void)hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
    printf("%d accumulator(s), gap = %-28s %.1f cycles per MFMA (or per group)\n", NACC, what, (double)h[0] / (it * 16.0));
}
int main()
{
    run<1, 0>("nothing"); run<1, 1>("v_exp_f32"); run<1, 2>("2 x v_pk_add_f32"); run<1, 3>("v_exp_f32 + v_pk_add_f32"); run<1, 4>("s_nop 0"); run<1, 5>("s_nop 1");
    run<2, 0>("nothing"); run<2, 1>("v_exp_f32"); run<2, 2>("2 x v_pk_add_f32"); run<2, 3>("v_exp_f32 + v_pk_add_f32"); run<2, 4>("s_nop 0"); run<2, 5>("s_nop 1");
    run<1, 6>("1 x v_pk_add_f32"); run<1, 7>("2 x v_add_f32"); run<1, 8>("2 x v_fma_f32"); run<1, 9>("v_exp_f32 + v_add_f32");
    run<0, 6>("1 x v_pk_add_f32, NO MFMA"); run<0, 7>("2 x v_add_f32, NO MFMA"); run<0, 1>("v_exp_f32, NO MFMA"); run<0, 10>("v_cvt_pk_bf16_f32, NO MFMA");
    run<1, 14>("2 x v_cvt_pk_bf16_f32"); run<1, 15>("2 x v_sub_f32 (independent)"); run<2, 7>("2 x v_add_f32"); run<2, 11>("v_and_b32 + v_lshlrev_b32"); run<2, 14>("2 x v_cvt_pk_bf16_f32"); run<2, 15>("2 x v_sub_f32 (independent)");
    run<1, 10>("v_cvt_pk_bf16_f32"); run<1, 11>("v_and_b32 + v_lshlrev_b32"); run<1, 12>("4 x v_add_f32"); run<1, 13>("v_rcp_f32 + v_exp_f32");
    return 0;
}
