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

// one output sample: y[n] = w_b x[n-k-1] + w_a x[n-k] with the reference's tap order and rounding
__device__ __forceinline__ float delay_sample(const float *xb, const float *bb, int D, int64_t n, float dn)
{
#pragma clang fp contract(off)   // products and the sum must round separately (bit-exact parity)
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
    return acc;
}

// thread -> 4 consecutive samples: d is read and y written with 16-byte accesses when the rows allow it
// (HBM-bound pass: 12 B/sample + the two gathered taps, which hit L2)
__global__ __launch_bounds__(256) void delay_apply_kernel(const float *x, const float *d, float *y, int64_t B,
                                                          int64_t T, const float *buf, int D, int warmup,
                                                          const int32_t *flag)
{
    if (flag && *flag) return;
    const int64_t b = blockIdx.x;
    const float *xb = x + b * T, *db = d + b * T, *bb = buf + b * (int64_t)D;
    float *yb = y + b * T;
    const bool vec = ((T & 3) == 0) && (((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(y) |
                                          reinterpret_cast<uintptr_t>(x)) & 15) == 0);
    for (int64_t n0 = 4 * ((int64_t)blockIdx.y * blockDim.x + threadIdx.x); n0 < T; n0 += 4 * (int64_t)gridDim.y * blockDim.x) {
        if (vec) {                                   // n0 + 3 < T because T is a multiple of 4
            if (warmup) { *(f32x4 *)(yb + n0) = *(const f32x4 *)(xb + n0); continue; }
            const f32x4 dv = *(const f32x4 *)(db + n0);
            f32x4 out;
#pragma unroll
            for (int c = 0; c < 4; ++c) out[c] = delay_sample(xb, bb, D, n0 + c, dv[c]);
            *(f32x4 *)(yb + n0) = out;
        } else {
            for (int c = 0; c < 4 && n0 + c < T; ++c)
                yb[n0 + c] = warmup ? xb[n0 + c] : delay_sample(xb, bb, D, n0 + c, db[n0 + c]);
        }
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
    const unsigned gx = (unsigned)((T + 1023) / 1024 > 4096 ? 4096 : (T + 1023) / 1024);     // 4 samples per thread
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


// ---------------------------------------------------------------------------------------
// K3b: DC-pre-emphasised ESR sums: both signals through H(z) = (1 - z^-1)/(1 - R z^-1) (zero state at
// `skip`), then sum f(t-y)^2 and sum f(t)^2 per stream.  The recursion v[n] = R v[n-1] + u[n] is linear, so a
// 256-thread block scans 4096 samples at a time: every thread runs its 16 samples from zero state, the
// (R^16, end value) pairs are combined by a shuffle scan, and the incoming carry is added as R^(k+1) c.
// ---------------------------------------------------------------------------------------
constexpr int DCL = 16;   // samples per thread per chunk

__global__ __launch_bounds__(256) void esr_dcpre_kernel(const float *y, const float *t, int64_t T, int64_t skip, float R,
                                                        double *out)
{
    const int64_t b = blockIdx.x;
    const float *yb = y + b * T, *tb = t + b * T;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float rp[DCL + 1];                       // R^0 .. R^16
    rp[0] = 1.0f;
#pragma unroll
    for (int k = 1; k <= DCL; ++k) rp[k] = rp[k - 1] * R;
    __shared__ float wA[4], wBe[4], wBt[4], carry[2];
    // the chunk is fetched with coalesced loads (lane-contiguous) and handed to the threads through LDS: a thread's
    // 16 consecutive samples sit in a row of 17 floats (the pad keeps the row reads conflict-free).  Reading them
    // straight from global memory (64-byte stride across lanes) cost 1.7 ms at 4096 x 65 536 instead of 0.6.
    __shared__ float s_e[256 * (DCL + 1)], s_t[256 * (DCL + 1)];
    if (tid == 0) { carry[0] = 0.0f; carry[1] = 0.0f; }
    double se = 0.0, st = 0.0;
    for (int64_t c0 = skip; c0 < T; c0 += 256 * DCL) {
        __syncthreads();
        const float cin_e = carry[0], cin_t = carry[1];
#pragma unroll
        for (int i = 0; i < DCL; ++i) {
            const int idx = i * 256 + tid;
            const int64_t n = c0 + idx;
            float tv = 0.0f, yv = 0.0f;
            if (n < T) { tv = tb[n]; yv = yb[n]; }
            s_t[(idx >> 4) * (DCL + 1) + (idx & 15)] = tv;
            s_e[(idx >> 4) * (DCL + 1) + (idx & 15)] = tv - yv;
        }
        __syncthreads();
        const int64_t n0 = c0 + (int64_t)tid * DCL;
        float ue[DCL], ut[DCL];
        float pe = 0.0f, pt = 0.0f;
        if (n0 > skip && n0 - 1 < T) {
            if (tid > 0) { pt = s_t[(tid - 1) * (DCL + 1) + DCL - 1]; pe = s_e[(tid - 1) * (DCL + 1) + DCL - 1]; }
            else { pt = tb[n0 - 1]; pe = pt - yb[n0 - 1]; }
        }
        float fe = 0.0f, ft = 0.0f;
#pragma unroll
        for (int k = 0; k < DCL; ++k) {
            const int64_t n = n0 + k;
            float tv = 0.0f, ev = 0.0f;
            if (n < T) { tv = s_t[tid * (DCL + 1) + k]; ev = s_e[tid * (DCL + 1) + k]; } else { tv = pt; ev = pe; }   // past the end: u = 0
            fe = (ev - pe) + R * fe;
            ft = (tv - pt) + R * ft;
            pe = ev; pt = tv;
            ue[k] = fe; ut[k] = ft;          // response from zero state
        }
        // inclusive scan of (A, Be, Bt) over the wave: (A2,B2) o (A1,B1) = (A1 A2, A2 B1 + B2)
        float A = rp[DCL], Be = fe, Bt = ft;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float Ap = __shfl_up(A, off), Bep = __shfl_up(Be, off), Btp = __shfl_up(Bt, off);
            if (lane >= off) { Be = __builtin_fmaf(A, Bep, Be); Bt = __builtin_fmaf(A, Btp, Bt); A *= Ap; }
        }
        if (lane == 63) { wA[wv] = A; wBe[wv] = Be; wBt[wv] = Bt; }
        // exclusive values for this lane
        float Ax = __shfl_up(A, 1), Bex = __shfl_up(Be, 1), Btx = __shfl_up(Bt, 1);
        if (lane == 0) { Ax = 1.0f; Bex = 0.0f; Btx = 0.0f; }
        __syncthreads();
        float ce = cin_e, ct = cin_t;        // carry entering this wave
        for (int v = 0; v < wv; ++v) { ce = __builtin_fmaf(wA[v], ce, wBe[v]); ct = __builtin_fmaf(wA[v], ct, wBt[v]); }
        const float le = __builtin_fmaf(Ax, ce, Bex), lt = __builtin_fmaf(Ax, ct, Btx);   // carry entering this lane
#pragma unroll
        for (int k = 0; k < DCL; ++k) {
            if (n0 + k < T) {
                const float ve = __builtin_fmaf(rp[k + 1], le, ue[k]), vt = __builtin_fmaf(rp[k + 1], lt, ut[k]);
                se += (double)ve * (double)ve;
                st += (double)vt * (double)vt;
            }
        }
        if (tid == 255) {                    // carry leaving the chunk
            carry[0] = __builtin_fmaf(rp[DCL], le, fe);
            carry[1] = __builtin_fmaf(rp[DCL], lt, ft);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        se += __shfl_down(se, off);
        st += __shfl_down(st, off);
    }
    __shared__ double part[2][4];
    if (lane == 0) { part[0][wv] = se; part[1][wv] = st; }
    __syncthreads();
    if (tid == 0) {
        out[2 * b + 0] = (part[0][0] + part[0][1]) + (part[0][2] + part[0][3]);
        out[2 * b + 1] = (part[1][0] + part[1][1]) + (part[1][2] + part[1][3]);
    }
}

hipError_t launch_esr_dcpre(const float *y, const float *t, int64_t B, int64_t T, int64_t skip, float R, double *out,
                            hipStream_t stream)
{
    if (B == 0) return hipSuccess;
    hipLaunchKernelGGL(esr_dcpre_kernel, dim3((unsigned)B), dim3(256), 0, stream, y, t, T, skip, R, out);
    return hipGetLastError();
}

}  // namespace ntm
