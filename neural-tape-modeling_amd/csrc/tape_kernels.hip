// N4 ("next" row): Jiles-Atherton magnetisation stage of the reference's white-box tape simulator,
// code/tape.py:516-551 (Tape.H_mag) + :587-635 (Tape._f): per stream a strictly sequential fp64 recurrence
// (trapezoidal dH/dt, 4th-order Runge-Kutta of the hysteresis ODE, clamp to +-Ms), the same archetype as the
// GRU path: streams are the only parallel axis.  One lane per stream; H and M tiles of 64 streams x 64
// samples go through LDS so that global accesses are 512-B row segments.
#include "ntm_common.h"

namespace ntm {

struct JaParams { double Ms, A, alpha, K, c, rA; };   // rA = 1/A

// The recurrence is one wave's dependent fp64 chain (a GPU has far more SIMDs than 4096 streams need waves), so
// what counts is the DEPTH of the per-sample computation, not its instruction count.  The two helpers below
// replace ocml's tanh (165 instructions, mostly serial) and the IEEE division (12).

// 1/x for x != 0 (finite, normal): v_rcp_f64 seed r0 (~2^-23), e = 1 - x r0, result r0 (1 + e)(1 + e^2): error
// e^4, within 1 ulp; dependent depth 4.  Only used where the denominator cannot vanish.
__device__ __forceinline__ double rcp_nr(double x)
{
    const double r0 = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r0, 1.0);
    const double r1 = fma(e, r0, r0), e2 = e * e;
    return fma(r1, e2, r1);
}

// expm1(x) for x <= 0: x = n ln2 + r, |r| <= ln2/2; expm1(r) = r + r^2 q(r) with the degree-11 Taylor q evaluated by
// Estrin's scheme (depth 5); expm1(x) = 2^n expm1(r) + (2^n - 1).  Relative error ~2e-16 where it matters
// (small |x|, n = 0); x very negative gives -1.
__device__ __forceinline__ double expm1_neg(double x)
{
    const double nf = __builtin_rint(x * 1.4426950408889634);
    double r = fma(-nf, 6.93147180369123816490e-01, x);        // ln2 hi / lo (fdlibm split)
    r = fma(-nf, 1.90821492927058770002e-10, r);
    const int n = (int)nf;
    const double s = __builtin_amdgcn_ldexp(1.0, n < -1080 ? -1080 : n);
    const double sm1 = s - 1.0;
    const double r2 = r * r;
    const double a0 = fma(r, 1.0 / 6, 1.0 / 2), a1 = fma(r, 1.0 / 120, 1.0 / 24), a2 = fma(r, 1.0 / 5040, 1.0 / 720);
    const double a3 = fma(r, 1.0 / 362880, 1.0 / 40320), a4 = fma(r, 1.0 / 39916800, 1.0 / 3628800);
    const double a5 = fma(r, 1.0 / 6227020800.0, 1.0 / 479001600);
    const double r4 = r2 * r2;
    const double b0 = fma(a1, r2, a0), b1 = fma(a3, r2, a2), b2 = fma(a5, r2, a4);
    const double r8 = r4 * r4;
    const double q = fma(b2, r8, fma(b1, r4, b0));
    const double pm = fma(r2, q, r);
    return fma(s, pm, sm1);
}

// coth(x) for |x| > 1e-4 from ONE expm1: with em = expm1(-2|x|) in (-1, 0), coth|x| = (2 + em) / (-em).
__device__ __forceinline__ double coth_gt(double x)
{
    const double em = expm1_neg(-2.0 * fabs(x));
    return copysign((2.0 + em) * rcp_nr(-em), x);
}

// L'(x) = 1/x^2 - coth^2 x + 1 for |x| < 1 (its argument here is L(Q), a Langevin value) as the even Taylor series
// sum_k (2k-1) 2^2k B_2k / (2k)! x^(2k-2), 16 terms by Estrin's scheme: depth 6 and 20 instructions instead of a second
// expm1 + two reciprocals (depth 23, 45 instructions) on the critical path of every RK4 stage; truncation 1e-15 at
// |x| = 1 (ratio of successive terms 1/pi^2), and no cancellation where the closed form loses digits (1/x^2 - coth^2 x
// for small x).  Coefficients: exact rationals rounded to double.
__device__ __forceinline__ double langevin_prime_lt1(double x)
{
    const double z = x * x, z2 = z * z, z4 = z2 * z2, z8 = z4 * z4;
    const double a0 = fma(z, -0.06666666666666667, 0.3333333333333333), a1 = fma(z, -0.0014814814814814814, 0.010582010582010581);
    const double a2 = fma(z, -2.380844708887037e-05, 0.0001924001924001924), a3 = fma(z, -3.332191318496952e-07, 2.8503732207435913e-06);
    const double a4 = fma(z, -4.332978728872515e-09, 3.8263339078575285e-08), a5 = fma(z, -5.3846925685597234e-11, 4.852350845790551e-10);
    const double a6 = fma(z, -6.489292139993081e-13, 5.930254350058414e-12), a7 = fma(z, -7.648843294003344e-15, 7.062066668463177e-14);
    const double b0 = fma(a1, z2, a0), b1 = fma(a3, z2, a2), b2 = fma(a5, z2, a4), b3 = fma(a7, z2, a6);
    const double c0 = fma(b1, z4, b0), c1 = fma(b3, z4, b2);
    return fma(c1, z8, c0);
}

__device__ __forceinline__ double ja_f(double Mn, double Hn, double Hp, const JaParams &p)
{
    const double Q = (Hn + p.alpha * Mn) * p.rA;
    const double LQ = fabs(Q) > 1e-4 ? coth_gt(Q) - rcp_nr(Q) : Q * (1.0 / 3.0);
    double LpQ;
    // (the reference evaluates L' on L(Q), not on Q: code/tape.py:598-603 -- reproduced)
    LpQ = fabs(LQ) > 1e-4 ? langevin_prime_lt1(LQ) : 1.0 / 3.0;        // |LQ| < 1 always: LQ is a Langevin value
    const double M_diff = p.Ms * LQ - Mn;
    const double dS = Hp > 0.0 ? 1.0 : -1.0;
    const double sgn = M_diff > 0.0 ? 1.0 : (M_diff < 0.0 ? -1.0 : 0.0);
    const double dM = (dS == sgn) ? 1.0 : 0.0;
    const double t1n = (1.0 - p.c) * dM * M_diff;
    const double t1d = (1.0 - p.c) * dS * p.K - p.alpha * M_diff;
    const double t1 = (t1n / t1d) * Hp;                  // IEEE division: t1d may vanish (inf, as in the reference)
    const double t2 = p.c * (p.Ms / p.A) * Hp * LpQ;
    const double t3 = 1.0 - p.c * p.alpha * (p.Ms / p.A) * LpQ;     // >= 1 - c alpha Ms / (3 A) > 0 for physical parameters
    return (t1 + t2) * rcp_nr(t3);
}

constexpr int JT = 64;   // tile: 64 streams x 64 samples

__global__ __launch_bounds__(64) void tape_hmag_kernel(const double *H, double *M, int64_t B, int64_t N, double *state,
                                                       double Ts, JaParams p)
{
#pragma clang fp contract(off)   // only the explicit fma() calls of the helpers fuse
    __shared__ double tile[JT][JT + 1];
    const int l = threadIdx.x;
    const int64_t s0 = (int64_t)blockIdx.x * JT;
    const int64_t b = s0 + l;
    const bool valid = b < B;
    double Mp = valid ? state[3 * b] : 0.0, Hpv = valid ? state[3 * b + 1] : 0.0, Hpp = valid ? state[3 * b + 2] : 0.0;
    for (int64_t n0 = 0; n0 < N; n0 += JT) {
        const int nt = (int)((N - n0) < JT ? (N - n0) : JT);
        for (int r = 0; r < JT; ++r)
            tile[r][l] = (s0 + r < B && l < nt) ? H[(s0 + r) * N + n0 + l] : 0.0;
        __syncthreads();
        for (int n = 0; n < nt; ++n) {
            const double Hn = tile[l][n];
            const double Hprime = 2.0 * (Hn - Hpv) / Ts - Hpp;
            const double k1 = Ts * ja_f(Mp, Hpv, Hpp, p);
            const double k2 = Ts * ja_f(Mp + k1 / 2.0, (Hn + Hpv) / 2.0, (Hprime + Hpp) / 2.0, p);
            const double k3 = Ts * ja_f(Mp + k2 / 2.0, (Hn + Hpv) / 2.0, (Hprime + Hpp) / 2.0, p);
            const double k4 = Ts * ja_f(Mp + k3, Hn, Hprime, p);
            double m = Mp + k1 * (1.0 / 6.0) + k2 * (1.0 / 3.0) + k3 * (1.0 / 3.0) + k4 * (1.0 / 6.0);
            m = m < -p.Ms ? -p.Ms : (m > p.Ms ? p.Ms : m);
            tile[l][n] = m;
            Hpv = Hn; Hpp = Hprime; Mp = m;
        }
        __syncthreads();
        for (int r = 0; r < JT; ++r)
            if (s0 + r < B && l < nt) M[(s0 + r) * N + n0 + l] = tile[r][l];
        __syncthreads();
    }
    if (valid) { state[3 * b] = Mp; state[3 * b + 1] = Hpv; state[3 * b + 2] = Hpp; }
}

// Record-head field of the reference's chain: I_rec = I_in + bias (code/tape.py:476-510), H = (N E I_rec) / G (:512-514),
// one fp64 pass, same operation order (bit-identical to the reference's torch ops).  bias [N] is shared by the streams.
__global__ __launch_bounds__(256) void tape_record_field_kernel(const double *I, const double *bias, double *H, int64_t B,
                                                                int64_t N, double gain, double gap)
{
#pragma clang fp contract(off)
    const int64_t total = B * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const double irec = bias ? I[i] + bias[i % N] : I[i];
        H[i] = (gain * irec) / gap;
    }
}

hipError_t launch_tape_record_field(const double *I, const double *bias, double *H, int64_t B, int64_t N, double gain,
                                    double gap, hipStream_t stream)
{
    if (B == 0 || N == 0) return hipSuccess;
    const int64_t total = B * N;
    const unsigned grid = (unsigned)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    hipLaunchKernelGGL(tape_record_field_kernel, dim3(grid), dim3(256), 0, stream, I, bias, H, B, N, gain, gap);
    return hipGetLastError();
}

// The two resamplers of Tape.__call__ (code/tape.py:330-332, 471-474, 553-558: torchaudio.transforms.Resample, sinc
// interpolation with a Hann window -- the polyphase kernel table is built on the host, tape.py of this package): one thread
// per output sample,
//   y[i * up + p] = sum_k ker[p][k] * xpad[i * down + k],   xpad[j] = x[j - width]  (zero outside [0, N))
// fp64, taps = 2 width + down (15 when oversampling by 16, 210 when coming back).  HBM / L2 bound streaming.
__global__ __launch_bounds__(256) void resample_fir_kernel(const double *x, double *y, int64_t N, int64_t M, int up, int down,
                                                           int width, const double *ker)
{
    const int64_t b = blockIdx.y;
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= M) return;
    const int64_t i = o / up;
    const int p = (int)(o - i * up);
    const int taps = 2 * width + down;
    const double *xb = x + b * N;
    const double *kp = ker + (size_t)p * taps;
    const int64_t j0 = i * down - width;
    double acc = 0.0;
    for (int k = 0; k < taps; ++k) {
        const int64_t j = j0 + k;
        if (j >= 0 && j < N) acc = __builtin_fma(kp[k], xb[j], acc);
    }
    y[b * M + o] = acc;
}

hipError_t launch_resample_fir(const double *x, double *y, int64_t B, int64_t N, int64_t M, int up, int down, int width,
                               const double *ker, hipStream_t stream)
{
    if (B == 0 || M == 0) return hipSuccess;
    hipLaunchKernelGGL(resample_fir_kernel, dim3((unsigned)((M + 255) / 256), (unsigned)B), dim3(256), 0, stream, x, y, N, M, up,
                       down, width, ker);
    return hipGetLastError();
}

// The playback-loss filter of Tape.H_play (code/tape.py:565-574: torchaudio.functional.lfilter with a = [1, 0, ...],
// i.e. a FIR, output clamped to [-1, 1] like lfilter's default clamp=True; no state carried, as in the reference):
//   y[n] = sum_{k < taps} h[k] x[n - k]
__global__ __launch_bounds__(256) void fir_f64_kernel(const double *x, double *y, int64_t N, const double *h, int taps, int clamp)
{
    const int64_t b = blockIdx.y;
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const double *xb = x + b * N;
    double acc = 0.0;
    for (int k = 0; k < taps && k <= n; ++k) acc = __builtin_fma(h[k], xb[n - k], acc);
    if (clamp) acc = acc < -1.0 ? -1.0 : (acc > 1.0 ? 1.0 : acc);
    y[b * N + n] = acc;
}

hipError_t launch_fir_f64(const double *x, double *y, int64_t B, int64_t N, const double *h, int taps, int clamp, hipStream_t stream)
{
    if (B == 0 || N == 0) return hipSuccess;
    hipLaunchKernelGGL(fir_f64_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)B), dim3(256), 0, stream, x, y, N, h, taps, clamp);
    return hipGetLastError();
}

hipError_t launch_tape_hmag(const double *H, double *M, int64_t B, int64_t N, double *state, double Ts, const double *par,
                            hipStream_t stream)
{
    if (B == 0 || N == 0) return hipSuccess;
    const JaParams p{par[0], par[1], par[2], par[3], par[4], 1.0 / par[1]};
    hipLaunchKernelGGL(tape_hmag_kernel, dim3((unsigned)((B + JT - 1) / JT)), dim3(64), 0, stream, H, M, B, N, state, Ts, p);
    return hipGetLastError();
}

}  // namespace ntm
