// Accuracy of the gate arithmetic on the MI355X, measured ("say which op -- v_exp / v_rcp -- it comes from").
// For 2^24 arguments spread over the range the GRU gates see: error of v_exp_f32 (2^x) and v_rcp_f32 alone in units of the last
// place of the exact result, and the error of the composite forms against float64:
//   sigmoid   cur: rcp(1 + exp2(a))                        nr : the same with one Newton step on the reciprocal
//   tanh      cur: 1 - 2 rcp(1 + exp2(p))                  A  : (e - 1) rcp(e + 1)        A_nr: A with the Newton step
// absolute and relative, overall and for |tanh| < 0.1.  One thread per argument, block-reduced, atomics on doubles.
//   hipcc -O3 --offload-arch=gfx950 gate_ulp.hip -o gate_ulp.bin && ./gate_ulp.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>

struct Acc { double max_, sum2; };
enum { EXP_ULP, RCP_ULP, SIG_CUR_REL, SIG_NR_REL, TANH_CUR_ABS, TANH_A_ABS, TANH_ANR_ABS, TANH_CUR_ABS_SMALL, TANH_A_ABS_SMALL,
       TANH_ANR_ABS_SMALL, TANH_CUR_REL_SMALL, TANH_A_REL_SMALL, TANH_ANR_REL_SMALL, NACC };

__device__ double ulp_of(double exact)        // spacing of float32 at |exact|
{
    int e;
    frexp(exact, &e);
    return ldexp(1.0, e - 24);
}
__device__ float rcp_nr(float s)
{
    const float r = __builtin_amdgcn_rcpf(s);
    return __builtin_fmaf(__builtin_fmaf(-s, r, 1.0f), r, r);
}
__global__ void measure(unsigned n, double *acc, unsigned long long *cnt_small)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    double v[NACC];
    for (int k = 0; k < NACC; ++k) v[k] = 0.0;
    bool small = false;
    if (i < n) {
        // argument of the exp2 in log2 units: uniform in [-24, 24] (sigmoid / tanh inputs up to ~ +-16.6 resp. +-8.3)
        const float p = -24.0f + 48.0f * ((float)i + 0.5f) / (float)n;
        const double ed = exp2((double)p);
        const float e = __builtin_amdgcn_exp2f(p);
        v[EXP_ULP] = fabs((double)e - ed) / ulp_of(ed);
        const float s = 1.0f + e;                              // the argument the reciprocal sees
        const float r = __builtin_amdgcn_rcpf(s);
        v[RCP_ULP] = fabs((double)r - 1.0 / (double)s) / ulp_of(1.0 / (double)s);
        const double sig = 1.0 / (1.0 + ed);
        v[SIG_CUR_REL] = fabs((double)r - sig) / sig;
        v[SIG_NR_REL] = fabs((double)rcp_nr(s) - sig) / sig;
        const double th = (ed - 1.0) / (ed + 1.0);             // tanh(p ln2 / 2)
        const float t_cur = __builtin_fmaf(-2.0f, r, 1.0f);
        const float pc = fminf(p, 64.0f);
        const float ec = __builtin_amdgcn_exp2f(pc);
        const float t_a = (ec - 1.0f) * __builtin_amdgcn_rcpf(ec + 1.0f);
        const float t_anr = (ec - 1.0f) * rcp_nr(ec + 1.0f);
        v[TANH_CUR_ABS] = fabs((double)t_cur - th);
        v[TANH_A_ABS] = fabs((double)t_a - th);
        v[TANH_ANR_ABS] = fabs((double)t_anr - th);
        small = fabs(th) < 0.1 && fabs(th) > 1e-3;
        if (small) {
            v[TANH_CUR_ABS_SMALL] = v[TANH_CUR_ABS]; v[TANH_A_ABS_SMALL] = v[TANH_A_ABS]; v[TANH_ANR_ABS_SMALL] = v[TANH_ANR_ABS];
            v[TANH_CUR_REL_SMALL] = v[TANH_CUR_ABS] / fabs(th); v[TANH_A_REL_SMALL] = v[TANH_A_ABS] / fabs(th);
            v[TANH_ANR_REL_SMALL] = v[TANH_ANR_ABS] / fabs(th);
        }
    }
    __shared__ double smax[256], ssum[256];
    for (int k = 0; k < NACC; ++k) {
        smax[threadIdx.x] = v[k]; ssum[threadIdx.x] = v[k] * v[k];
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) {
                smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + st]);
                ssum[threadIdx.x] += ssum[threadIdx.x + st];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            atomicMax((unsigned long long *)&acc[2 * k], (unsigned long long)__double_as_longlong(smax[0]));   // non-negative doubles order as integers
            atomicAdd(&acc[2 * k + 1], ssum[0]);
        }
        __syncthreads();
    }
    if (small) atomicAdd(cnt_small, 1ull);
}

int main()
{
    const unsigned n = 1u << 24;
    double *acc, h[2 * NACC];
    unsigned long long *cnt, hc;
    (void)hipMalloc(&acc, sizeof(h)); (void)hipMemset(acc, 0, sizeof(h));
    (void)hipMalloc(&cnt, 8); (void)hipMemset(cnt, 0, 8);
    hipLaunchKernelGGL(measure, dim3(n / 256), dim3(256), 0, 0, n, acc, cnt);
    (void)hipMemcpy(h, acc, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipMemcpy(&hc, cnt, 8, hipMemcpyDeviceToHost);
    const char *names[NACC] = {"v_exp_f32 (2^x), ulp", "v_rcp_f32 of (1 + e), ulp", "sigmoid cur, rel", "sigmoid + Newton, rel",
                               "tanh cur 1-2rcp(1+e), abs", "tanh A (e-1)rcp(e+1), abs", "tanh A + Newton, abs",
                               "tanh cur, abs, 1e-3<|t|<0.1", "tanh A, abs, small", "tanh A + Newton, abs, small",
                               "tanh cur, REL, small", "tanh A, REL, small", "tanh A + Newton, REL, small"};
    printf("%u arguments, exp2 argument uniform in [-24, 24]; %llu of them with 1e-3 < |tanh| < 0.1\n", n, hc);
    for (int k = 0; k < NACC; ++k) {
        const double cntk = k >= TANH_CUR_ABS_SMALL ? (double)hc : (double)n;
        printf("%-34s max %.3e   rms %.3e\n", names[k], h[2 * k], sqrt(h[2 * k + 1] / cntk));
    }
    return 0;
}
