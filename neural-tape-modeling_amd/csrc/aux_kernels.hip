// K2 (time-varying fractional delay line) and K3 (ESR partial sums): streaming kernels, coalesced
// 4-16 B/lane global accesses, no MFMA.  (K4, the TCN, lives in tcn_kernels.hip.)
#include "ntm_common.h"

namespace ntm {

// ---------------------------------------------------------------------------------------
// K2: TimeVaryingDelayLine.forward, code/model.py:269-320, in closed form.
//   y[n] = sum_{m in {k+1,k}, 0<=m<=D} relu(1-|m-d[n]|) * xpad[n-m],  k = floor(d[n])
//   xpad[i<0] = buffer[D+i].  Products and the sum are individually rounded (no fma contraction)
//   in the reference's order, so the result is bit-identical to the O(T*D) unfold formulation.
// Two launches per call, ONE pass over the audio:
//   delay_apply_kernel   reads d and the taps, writes y, and raises the error flag where d > D (the reference's
//                        assert, code/model.py:284) -- d is read once, there is no separate range-check pass;
//   delay_update_kernel  buffer <- cat(buffer[T:], x[-D:]) in place (B x D floats), skipped when the flag is up,
//                        so the carried state stays untouched exactly when the reference would have raised.
// The flag is STICKY and caller-owned: once it is non-zero every later K2 launch on it is a no-op (state frozen at
// the last good call) until the caller clears it -- the host checks it when it wants to (once per predict), not per call.
// ---------------------------------------------------------------------------------------
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte access at 4-byte alignment

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

// the same sample when both taps are known to lie inside x and inside [0, D]: xa = x[n-k], xb1 = x[n-k-1]
__device__ __forceinline__ float delay_sample_fast(float dn, float kf, float xa, float xb1)
{
#pragma clang fp contract(off)
    float acc = 0.0f;
    const float wb = 1.0f - fabsf((kf + 1.0f) - dn);
    if (wb > 0.0f) { const float prod = wb * xb1; acc = acc + prod; }
    const float wa = 1.0f - fabsf(kf - dn);
    if (wa > 0.0f) { const float prod = wa * xa; acc = acc + prod; }
    return acc;
}

// thread -> DV consecutive samples (DV = 8: two 16-byte loads of d, two 16-byte stores of y).  Fast path, taken when
// the DV delays of the thread share one integer part k with 0 <= k < D and the window lies inside x: the DV + 1
// samples x[n0-k-1 .. n0+DV-1-k] the taps need are contiguous -> two 16-byte loads at 4-byte alignment + one dword
// instead of 2 DV scalar gathers.  Everything else (k changes inside the thread's run, history taps, k = D, d < 0,
// NaN) goes through delay_sample().  HBM-bound pass: 12 B/sample (d, x once, y).
constexpr int DV = 8;
__global__ __launch_bounds__(256) void delay_apply_kernel(const float *x, const float *d, float *y, int64_t B,
                                                          int64_t T, const float *buf, int D, int warmup,
                                                          int32_t *flag)
{
    if (flag && *flag) return;                       // sticky: an earlier violation froze this state
    const int64_t tiles = (T + 256 * DV - 1) / (256 * DV);
    const int64_t b = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const float *xb = x + b * T, *db = d + b * T, *bb = buf + b * (int64_t)D;
    float *yb = y + b * T;
    const float Dmax = (float)D;
    const bool vec = ((T & 3) == 0) && (((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(y)) & 15) == 0);
    const int64_t n0 = (tile * 256 + threadIdx.x) * DV;
    int bad = 0;
    if (n0 + DV <= T && vec) {
        const f32x4 d0 = *(const f32x4 *)(db + n0), d1 = *(const f32x4 *)(db + n0 + 4);
        float dv[DV] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
        const float kf = floorf(dv[0]);
        bool same = kf >= 0.0f && kf < Dmax && (n0 - (int64_t)kf - 1) >= 0;
#pragma unroll
        for (int c = 0; c < DV; ++c) {
            bad |= !(dv[c] <= Dmax);                 // also true for NaN, like `max_delay >= max(dt)` failing
            same = same && (floorf(dv[c]) == kf);
        }
        float out[DV];
        if (warmup) {
            const f32x4u x0 = *(const f32x4u *)(xb + n0), x1 = *(const f32x4u *)(xb + n0 + 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) { out[c] = x0[c]; out[4 + c] = x1[c]; }
        } else if (same) {
            const float *w0 = xb + (n0 - (int64_t)kf - 1);
            const f32x4u x0 = *(const f32x4u *)w0, x1 = *(const f32x4u *)(w0 + 4);
            const float win[DV + 1] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3], w0[8]};
#pragma unroll
            for (int c = 0; c < DV; ++c) out[c] = delay_sample_fast(dv[c], kf, win[c + 1], win[c]);
        } else {
#pragma unroll
            for (int c = 0; c < DV; ++c) out[c] = delay_sample(xb, bb, D, n0 + c, dv[c]);
        }
        *(f32x4 *)(yb + n0) = (f32x4){out[0], out[1], out[2], out[3]};
        *(f32x4 *)(yb + n0 + 4) = (f32x4){out[4], out[5], out[6], out[7]};
    } else {
        for (int c = 0; c < DV && n0 + c < T; ++c) {
            const float dn = db[n0 + c];
            bad |= !(dn <= Dmax);
            yb[n0 + c] = warmup ? xb[n0 + c] : delay_sample(xb, bb, D, n0 + c, dn);
        }
    }
    if (flag && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// buffer <- cat(buffer[T:], x[-D:])   (code/model.py:314-315), in place, one workgroup per stream.
// T >= D: the tail of x.  T < D: the surviving D-T samples move down by T -- chunk after chunk in ascending order,
// every chunk read completely (into registers) before the workgroup barrier that precedes its write, so no source
// sample is overwritten before it has been read -- then the T new samples follow.
constexpr int DU_THREADS = 1024;
__global__ __launch_bounds__(DU_THREADS) void delay_update_kernel(const float *x, float *buf, int64_t T, int D,
                                                                 const int32_t *flag)
{
    if (flag && *flag) return;
    const int64_t b = blockIdx.x;
    float *bb = buf + b * (int64_t)D;
    const float *xb = x + b * T;
    const int keep = T >= D ? 0 : D - (int)T;
    for (int c0 = 0; c0 < keep; c0 += DU_THREADS) {
        const int i = c0 + threadIdx.x;
        const float v = i < keep ? bb[i + T] : 0.0f;
        __syncthreads();
        if (i < keep) bb[i] = v;
    }
    const int64_t x0 = T - (D - keep);               // first sample of x that enters the buffer
    for (int i = keep + threadIdx.x; i < D; i += DU_THREADS) bb[i] = xb[x0 + (i - keep)];
}

hipError_t launch_delay(const float *x, const float *d, float *y, int64_t B, int64_t T, float *dl_state, int D,
                        int warmup, int32_t *err_flag, hipStream_t stream)
{
    if (B == 0 || T == 0) return hipSuccess;
    const int64_t tiles = (T + 256 * DV - 1) / (256 * DV);
    if (B * tiles > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(delay_apply_kernel, dim3((unsigned)(B * tiles)), dim3(256), 0, stream, x, d, y, B, T, dl_state, D,
                       warmup, err_flag);
    if (D > 0)
        hipLaunchKernelGGL(delay_update_kernel, dim3((unsigned)B), dim3(DU_THREADS), 0, stream, x, dl_state, T, D, err_flag);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// K3: per-stream ESR sums over samples [skip,T): sum (t-y)^2 and sum t^2 in fp64.
// grid (B, splits): block (b, p) owns every splits-th 256-sample slab of stream b and writes ITS OWN partial row
// out[(b * splits + p) * 2 + 0..1] -- no atomics, so the result is bit-reproducible from run to run for every B; the
// caller adds the `splits` rows of a stream in index order (as ntm_stft_sums' callers do).
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
        double *o = out + (b * gridDim.y + blockIdx.y) * 2;
        o[0] = (part[0][0] + part[0][1]) + (part[0][2] + part[0][3]);
        o[1] = (part[1][0] + part[1][1]) + (part[1][2] + part[1][3]);
    }
}

int esr_default_splits(int64_t B, int64_t T, int64_t skip)
{
    const int64_t n = T - skip;
    if (B <= 0 || n <= 0) return 1;
    int64_t splits = (n + 4095) / 4096;            // >= 16 samples per thread
    const int64_t want = (2048 + B - 1) / B;       // enough blocks to fill 256 CUs
    if (splits > want) splits = want;
    return (int)(splits < 1 ? 1 : splits);
}

hipError_t launch_esr(const float *y, const float *t, int64_t B, int64_t T, int64_t skip, int splits, double *out,
                      hipStream_t stream)
{
    if (B == 0) return hipSuccess;
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
