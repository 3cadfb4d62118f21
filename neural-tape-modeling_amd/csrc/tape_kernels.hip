// N4 ("next" row): Jiles-Atherton magnetisation stage of the reference's white-box tape simulator,
// code/tape.py:516-551 (Tape.H_mag) + :587-635 (Tape._f): per stream a strictly sequential fp64 recurrence
// (trapezoidal dH/dt, 4th-order Runge-Kutta of the hysteresis ODE, clamp to +-Ms), the same archetype as the
// GRU path: streams are the only parallel axis.  One lane per stream; H and M tiles of 64 streams x 64
// samples go through LDS so that global accesses are 512-B row segments.
#include "ntm_common.h"

namespace ntm {

struct JaParams { double Ms, A, alpha, K, c; };

__device__ __forceinline__ double ja_f(double Mn, double Hn, double Hp, const JaParams &p)
{
    const double Q = (Hn + p.alpha * Mn) / p.A;
    const double LQ = fabs(Q) > 1e-4 ? (1.0 / tanh(Q)) - 1.0 / Q : Q / 3.0;
    double LpQ;
    // (the reference evaluates L' on L(Q), not on Q: code/tape.py:598-603 -- reproduced)
    if (fabs(LQ) > 1e-4) { const double ct = 1.0 / tanh(LQ); LpQ = 1.0 / (LQ * LQ) - ct * ct + 1.0; } else LpQ = 1.0 / 3.0;
    const double M_diff = p.Ms * LQ - Mn;
    const double dS = Hp > 0.0 ? 1.0 : -1.0;
    const double sgn = M_diff > 0.0 ? 1.0 : (M_diff < 0.0 ? -1.0 : 0.0);
    const double dM = (dS == sgn) ? 1.0 : 0.0;
    const double t1n = (1.0 - p.c) * dM * M_diff;
    const double t1d = (1.0 - p.c) * dS * p.K - p.alpha * M_diff;
    const double t1 = (t1n / t1d) * Hp;
    const double t2 = p.c * (p.Ms / p.A) * Hp * LpQ;
    const double t3 = 1.0 - p.c * p.alpha * (p.Ms / p.A) * LpQ;
    return (t1 + t2) / t3;
}

constexpr int JT = 64;   // tile: 64 streams x 64 samples

__global__ __launch_bounds__(64) void tape_hmag_kernel(const double *H, double *M, int64_t B, int64_t N, double *state,
                                                       double Ts, JaParams p)
{
#pragma clang fp contract(off)   // keep the reference's operation-by-operation fp64 rounding
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
            double m = Mp + k1 / 6.0 + k2 / 3.0 + k3 / 3.0 + k4 / 6.0;
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

hipError_t launch_tape_hmag(const double *H, double *M, int64_t B, int64_t N, double *state, double Ts, const double *par,
                            hipStream_t stream)
{
    if (B == 0 || N == 0) return hipSuccess;
    const JaParams p{par[0], par[1], par[2], par[3], par[4]};
    hipLaunchKernelGGL(tape_hmag_kernel, dim3((unsigned)((B + JT - 1) / JT)), dim3(64), 0, stream, H, M, B, N, state, Ts, p);
    return hipGetLastError();
}

}  // namespace ntm
