// K2 (time-varying fractional delay line) and K3 (ESR partial sums): streaming kernels, coalesced
// 4-16 B/lane global accesses, no MFMA.  (K4, the TCN, lives in tcn_kernels.hip.)
#include "ntm_common.h"

namespace ntm {

// ---------------------------------------------------------------------------------------
// K2: TimeVaryingDelayLine.forward, code/model.py:269-320, in closed form.
//   y[n] = sum_{m in {k+1,k}, 0<=m<=D} relu(1-|m-d[n]|) * xpad[n-m],  k = floor(d[n])
//   xpad[i<0] = buffer[D+i].  Products and the sum are individually rounded (no fma contraction)
//   in the reference's order, so the result is bit-identical to the O(T*D) unfold formulation.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void delay_check_kernel(const float *d, int64_t n, float Dmax, int32_t *flag)
{
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        bad |= (d[i] > Dmax) ? 1 : 0;
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

__global__ __launch_bounds__(256) void delay_apply_kernel(const float *x, const float *d, float *y, int64_t B,
                                                          int64_t T, const float *buf, int D, int warmup,
                                                          const int32_t *flag)
{
#pragma clang fp contract(off)   // products and the sum must round separately (bit-exact parity)
    if (flag && *flag) return;
    const int64_t b = blockIdx.x;
    const float *xb = x + b * T, *db = d + b * T, *bb = buf + b * (int64_t)D;
    float *yb = y + b * T;
    for (int64_t n = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; n < T; n += (int64_t)gridDim.y * blockDim.x) {
        if (warmup) { yb[n] = xb[n]; continue; }
        const float dn = db[n];
        const float kf = floorf(dn);
        float acc = 0.0f;
#pragma unroll
        for (int tap = 1; tap >= 0; --tap) {          // m = k+1 first, then m = k (reference sum order)
            const float mf = kf + (float)tap;
            if (mf < 0.0f || mf > (float)D) continue;
            const float w = 1.0f - fabsf(mf - dn);
            if (!(w > 0.0f)) continue;
            const int64_t src = n - (int64_t)mf;
            const float xv = src >= 0 ? xb[src] : bb[D + src];
            const float prod = w * xv;
            acc = acc + prod;
        }
        yb[n] = acc;
    }
}

// buffer <- cat(buffer[T:], x[-D:])   (code/model.py:314-315).  T >= D: pure copy of x's tail.
// T < D: the surviving D-T samples are staged through `scratch` by the first kernel.
__global__ __launch_bounds__(256) void delay_stage_kernel(const float *buf, float *scratch, int64_t T, int D,
                                                          const int32_t *flag)
{
    if (flag && *flag) return;
    const int64_t b = blockIdx.x;
    const int keep = D - (int)T;
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < keep; i += gridDim.y * blockDim.x)
        scratch[b * keep + i] = buf[b * (int64_t)D + T + i];
}

__global__ __launch_bounds__(256) void delay_update_kernel(const float *x, float *buf, const float *scratch,
                                                           int64_t T, int D, const int32_t *flag)
{
    if (flag && *flag) return;
    const int64_t b = blockIdx.x;
    const int keep = T >= D ? 0 : D - (int)T;
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < D; i += gridDim.y * blockDim.x)
        buf[b * (int64_t)D + i] = i < keep ? scratch[b * keep + i] : x[b * T + (T - (D - keep)) + (i - keep)];
}

hipError_t launch_delay(const float *x, const float *d, float *y, int64_t B, int64_t T, float *dl_state, int D,
                        int warmup, float *scratch, int32_t *err_flag, hipStream_t stream)
{
    if (B == 0 || T == 0) return hipSuccess;
    const unsigned gx = (unsigned)((T + 255) / 256 > 4096 ? 4096 : (T + 255) / 256);
    if (err_flag) {
        hipError_t e = hipMemsetAsync(err_flag, 0, sizeof(int32_t), stream);
        if (e != hipSuccess) return e;
        const int64_t n = B * T;
        const unsigned gc = (unsigned)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
        hipLaunchKernelGGL(delay_check_kernel, dim3(gc), dim3(256), 0, stream, d, n, (float)D, err_flag);
    }
    hipLaunchKernelGGL(delay_apply_kernel, dim3((unsigned)B, gx), dim3(256), 0, stream, x, d, y, B, T, dl_state, D,
                       warmup, err_flag);
    if (D > 0) {
        const unsigned gd = (unsigned)((D + 255) / 256);
        if (T < D)
            hipLaunchKernelGGL(delay_stage_kernel, dim3((unsigned)B, gd), dim3(256), 0, stream, dl_state, scratch, T,
                               D, err_flag);
        hipLaunchKernelGGL(delay_update_kernel, dim3((unsigned)B, gd), dim3(256), 0, stream, x, dl_state, scratch, T,
                           D, err_flag);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// K3: per-stream ESR sums over samples [skip,T): out[2b] += sum (t-y)^2, out[2b+1] += sum t^2.
// grid (splits, B); fp64 accumulation; one fp64 atomic pair per block.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void esr_sums_kernel(const float *y, const float *t, int64_t T, int64_t skip,
                                                       double *out)
{
    const int64_t b = blockIdx.x;
    const float *yb = y + b * T, *tb = t + b * T;
    double se = 0.0, st = 0.0;
    for (int64_t n = skip + (int64_t)blockIdx.y * blockDim.x + threadIdx.x; n < T;
         n += (int64_t)gridDim.y * blockDim.x) {
        const float tv = tb[n], e = tv - yb[n];
        se += (double)e * (double)e;
        st += (double)tv * (double)tv;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        se += __shfl_down(se, off);
        st += __shfl_down(st, off);
    }
    __shared__ double part[2][4];
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { part[0][wv] = se; part[1][wv] = st; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&out[2 * b + 0], (part[0][0] + part[0][1]) + (part[0][2] + part[0][3]));
        atomicAdd(&out[2 * b + 1], (part[1][0] + part[1][1]) + (part[1][2] + part[1][3]));
    }
}

hipError_t launch_esr(const float *y, const float *t, int64_t B, int64_t T, int64_t skip, double *out,
                      hipStream_t stream)
{
    if (B == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(out, 0, sizeof(double) * 2 * (size_t)B, stream);
    if (e != hipSuccess) return e;
    const int64_t n = T - skip;
    if (n <= 0) return hipSuccess;
    int64_t splits = (n + 4095) / 4096;            // >= 16 samples per thread
    const int64_t want = (2048 + B - 1) / B;       // enough blocks to fill 256 CUs
    if (splits > want) splits = want;
    if (splits < 1) splits = 1;
    hipLaunchKernelGGL(esr_sums_kernel, dim3((unsigned)B, (unsigned)splits), dim3(256), 0, stream, y, t, T, skip, out);
    return hipGetLastError();
}

}  // namespace ntm
