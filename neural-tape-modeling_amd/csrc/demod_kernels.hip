// "Next" row N3: DelayAnalyzer.demodulate, code/utilities/utilities.py:408-465 -- undo the time-varying delay of a
// (C, N) recording from the pulse indices of its pilot train.  Two piecewise-linear maps, both evaluated with
// the formulas of scipy.interpolate.interp1d(kind='linear') -- fp64, with the audio knot values kept in float32 as
// scipy keeps them -- so that the result matches the reference's numpy/scipy code to the last bit before the final
// fp32 store:
//   y_hat[j] = f(j),  f through the knots (y_idx[i] -> y_idx[0] + i*period), extrapolated          (:441-447)
//   dem[t]   = g(t),  g through the knots (y_hat[j] -> x[:, j]); below the first knot x[:, 0], above the
//              last knot x[:, 1] (the reference passes fill_value=(output[:, 0], output[:, 1]))      (:450-455)
//   out[t]   = dem[t + shift] (zeros in the last `shift` samples) when shift = y_idx[0] - x_idx[0] > 0 (:458-465)
// Data-parallel over samples, HBM/latency-bound; not on the hot path (dataset preparation).
#include "ntm_common.h"

namespace ntm {

#pragma clang fp contract(off)   // slope * dx + y0 must round like numpy does (no FMA)

// numpy.searchsorted(a, v, side='left') -- the first index with a[idx] >= v, a ascending -- found from a GUESS: both maps
// are near-linear (a pulse every `period` samples, a
// time warp of a few samples per thousand), so the answer lies within a few elements of an extrapolated position. Gallop
// from the guess until the bracket a[lo - 1] < v <= a[hi] is established, then bisect inside it -- 3-5 dependent loads
// instead of log2(n) = 25 for the 26 M-sample map (the kernels are bound by exactly that latency chain).
template <typename T>
__device__ __forceinline__ int64_t searchsorted_left_near(const T *a, int64_t n, double v, int64_t guess)
{
    int64_t g = guess < 0 ? 0 : (guess > n - 1 ? n - 1 : guess);
    int64_t lo, hi;              // invariant at the end: every index < lo has a < v, every index >= hi has a >= v
    if ((double)a[g] >= v) {
        hi = g;
        int64_t step = 1;
        lo = g - step;
        while (lo >= 0 && (double)a[lo] >= v) { hi = lo; step <<= 1; lo = hi - step; }
        lo = lo < 0 ? 0 : lo + 1;
    } else {
        lo = g + 1;
        int64_t step = 1;
        hi = g + step;
        while (hi < n && (double)a[hi] < v) { lo = hi + 1; step <<= 1; hi = lo - 1 + step; }
        hi = hi > n ? n : hi;
    }
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((double)a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void demod_yhat_kernel(const int64_t *y_idx, int P, int64_t period, int64_t N, double *y_hat)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    int64_t hi = searchsorted_left_near(y_idx, (int64_t)P, (double)j, (j - y_idx[0]) / (period > 0 ? period : 1));
    hi = hi < 1 ? 1 : (hi > P - 1 ? P - 1 : hi);
    const int64_t lo = hi - 1;
    const double x_lo = (double)y_idx[lo], x_hi = (double)y_idx[hi];
    const double v_lo = (double)(y_idx[0] + lo * period), v_hi = (double)(y_idx[0] + hi * period);
    const double slope = (v_hi - v_lo) / (x_hi - x_lo);
    y_hat[j] = slope * ((double)j - x_lo) + v_lo;
}

__global__ __launch_bounds__(256) void demod_apply_kernel(const float *x, float *out, int C, int64_t N, int64_t shift,
                                                          const double *y_hat)
{
    const int64_t tp = (int64_t)blockIdx.x * 256 + threadIdx.x;      // output position
    if (tp >= N) return;
    const int64_t t = shift > 0 ? tp + shift : tp;                   // position before the roll
    if (t >= N) {
        for (int c = 0; c < C; ++c) out[c * N + tp] = 0.0f;
        return;
    }
    const double tv = (double)t;
    if (tv < y_hat[0] || tv > y_hat[N - 1]) {
        const int64_t src = tv < y_hat[0] ? 0 : 1;
        for (int c = 0; c < C; ++c) out[c * N + tp] = x[c * N + src];
        return;
    }
    // guess: y_hat is j + (a slowly varying offset), so the knot for t sits about that offset before t
    const int64_t off = (int64_t)(y_hat[t] - tv);
    int64_t hi = searchsorted_left_near(y_hat, N, tv, t - off);
    hi = hi < 1 ? 1 : (hi > N - 1 ? N - 1 : hi);
    const int64_t lo = hi - 1;
    const double x_lo = y_hat[lo], dx = y_hat[hi] - x_lo, dt = tv - x_lo;
    for (int c = 0; c < C; ++c) {
        // float32 knot values, as the reference's caller passes them (code/dataset.py:397): scipy's interp1d keeps
        // that dtype, so the difference is rounded to float32 before the float64 slope is formed
        const float v_lo = x[c * N + lo], v_hi = x[c * N + hi];
        const float dv = v_hi - v_lo;
        const double slope = (double)dv / dx;
        out[c * N + tp] = (float)(slope * dt + (double)v_lo);
    }
}

hipError_t launch_demodulate(const float *x, float *out, int C, int64_t N, const int64_t *y_idx, int P, int64_t period,
                             int64_t shift, double *scratch, hipStream_t stream)
{
    const unsigned grid = (unsigned)((N + 255) / 256);
    hipLaunchKernelGGL(demod_yhat_kernel, dim3(grid), dim3(256), 0, stream, y_idx, P, period, N, scratch);
    hipLaunchKernelGGL(demod_apply_kernel, dim3(grid), dim3(256), 0, stream, x, out, C, N, shift, scratch);
    return hipGetLastError();
}

}   // namespace ntm
