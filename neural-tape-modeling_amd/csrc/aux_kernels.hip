// K2 (time-varying fractional delay line) and K3 (ESR partial sums): streaming kernels, coalesced
// 4-16 B/lane global accesses, no MFMA.  (K4, the TCN, lives in tcn_kernels.hip.)
#include "ntm_common.h"
#include "delay_math.h"

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
// A 256-thread workgroup covers DG runs of 1024 consecutive samples of one stream; in a run thread i owns the 4
// samples 4i..4i+3, so every 16-byte access of a wavefront (d load, tap window, y store) is lane-contiguous -- 1 KB per
// instruction, fully coalesced; the d loads of all runs are issued before anything else.
// Fast path, taken when the 4 delays of a thread share one integer part k with 0 <= k < D and the window lies inside
// x: the 5 samples x[n0-k-1 .. n0+3-k] the taps need are contiguous -> one 16-byte load at 4-byte alignment + one
// dword instead of 8 scalar gathers.  Everything else (k changes inside the run, history taps, k = D, d < 0, NaN) goes
// through delay_sample().  HBM-bound pass: 12 B/sample (d, x once, y).  Sample indices inside a stream are 32-bit
// (T < 2^31) and the classification avoids float -> int64 conversions: at 4 samples per thread the pass is as much
// instruction-bound as memory-bound.
// History of the shape at 4096 x 65 536, D = 1847 (tools/attic/delay_probe.py; torch's two-in one-out elementwise add, the
// same traffic, takes 0.54 ms): four launches with a separate range-check pass ~1.5 ms | one pass, 8 consecutive
// samples per thread (32-byte lane stride) 1.03 | 4 per thread, two runs per workgroup 0.89 | XCD-aware ids 0.85 |
// a loop of 8 runs per workgroup with the next d prefetched 0.99 (worse: dropped) | 32-bit lean classification with
// unconditional window loads (all four loads of a workgroup's two runs in flight together) 0.81 = 4.1 TB/s.
constexpr int DV = 4;            // samples per thread and run
constexpr int DRUN = 256 * DV;   // samples per run
constexpr int DG = 2;            // runs per workgroup
template <bool NT>               // nontemporal d / y accesses
__global__ __launch_bounds__(256) void delay_apply_kernel(const float *x, const float *d, float *y, int64_t B,
                                                          int T, const float *buf, int D, int warmup,
                                                          int32_t *flag, unsigned tiles_u)
{
    // XCD-aware mapping (speed only): workgroups are dealt round-robin over the 8 XCDs, so ids l and l + 8 share an
    // L2.  All tiles of a stream get the same l % 8 and follow each other on that XCD: the tap window of a tile (it
    // reaches up to D samples back into the previous tile's part of x) is then in THAT L2 instead of being fetched a
    // second time through the fabric by another XCD.
    const unsigned l = blockIdx.x, j = l >> 3;
    const unsigned tile = j % tiles_u;
    const int64_t b = (int64_t)(j / tiles_u) * 8 + (l & 7);
    if (b >= B) return;
    const float *xb = x + b * T, *db = d + b * T, *bb = buf + b * (int64_t)D;
    float *yb = y + b * T;
    const float Dmax = (float)D;
    const bool vec = ((T & 3) == 0) && (((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(y)) & 15) == 0);
    const int base = (int)(tile * (unsigned)(DRUN * DG)) + (int)threadIdx.x * DV;
    int bad = 0;
    if (vec && base + (DG - 1) * DRUN + DV <= T) {
        // ---- hot path: every run of the workgroup is complete ------------------------------------------------
        f32x4 dv[DG];
#pragma unroll
        for (int g = 0; g < DG; ++g) {
            const f32x4 *p = (const f32x4 *)(db + base + g * DRUN);
            dv[g] = NT ? __builtin_nontemporal_load(p) : *p;
        }
        if (flag && *flag) return;                   // sticky: an earlier violation froze this state
        float kf[DG];
        bool same[DG];
        f32x4u xv[DG];
        float x4[DG];
#pragma unroll
        for (int g = 0; g < DG; ++g) {               // classify, issue the tap-window loads of every run
            kf[g] = floorf(dv[g][0]);
            const int w = base + g * DRUN - (int)kf[g] - 1;      // first sample of the tap window
            // one integer part for the 4 delays <=> 0 <= d_c - k < 1 (the difference is exact in that range)
            const float t1 = dv[g][1] - kf[g], t2 = dv[g][2] - kf[g], t3 = dv[g][3] - kf[g];
            same[g] = kf[g] >= 0.0f && kf[g] < Dmax && w >= 0 && t1 >= 0.0f && t1 < 1.0f && t2 >= 0.0f && t2 < 1.0f &&
                      t3 >= 0.0f && t3 < 1.0f;
            bad |= !(dv[g][0] <= Dmax) | !(dv[g][1] <= Dmax) | !(dv[g][2] <= Dmax) | !(dv[g][3] <= Dmax);   // NaN too
            // the loads are unconditional (lanes off the fast path read a harmless in-range window), so that all of
            // them are in flight together; warm-up mode reads the samples themselves
            const int ws = warmup ? base + g * DRUN : (same[g] ? w : 0);
            xv[g] = *(const f32x4u *)(xb + ws);
            x4[g] = xb[ws + 4 < T ? ws + 4 : ws];
        }
#pragma unroll
        for (int g = 0; g < DG; ++g) {
            const float win[DV + 1] = {xv[g][0], xv[g][1], xv[g][2], xv[g][3], x4[g]};
            f32x4 out;
#pragma unroll
            for (int c = 0; c < DV; ++c) out[c] = warmup ? win[c] : delay_sample_fast(dv[g][c], kf[g], win[c + 1], win[c]);
            if (!warmup && !same[g]) {               // rare: the general form, sample by sample
#pragma unroll
                for (int c = 0; c < DV; ++c) out[c] = delay_sample(xb, bb, D, base + g * DRUN + c, dv[g][c]);
            }
            f32x4 *q = (f32x4 *)(yb + base + g * DRUN);
            if constexpr (NT) __builtin_nontemporal_store(out, q);
            else *q = out;
        }
    } else {
        // ---- ragged tail / unaligned rows: sample by sample ---------------------------------------------------
        if (flag && *flag) return;
        for (int g = 0; g < DG; ++g)
            for (int c = 0; c < DV; ++c) {
                const int n = base + g * DRUN + c;
                if (n >= T) break;
                const float dn = db[n];
                bad |= !(dn <= Dmax);
                yb[n] = warmup ? xb[n] : delay_sample(xb, bb, D, n, dn);
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

// the interpolation alone (the carried buffer is read, not moved on)
hipError_t launch_delay_apply(const float *x, const float *d, float *y, int64_t B, int64_t T, const float *dl_state, int D,
                              int warmup, int32_t *err_flag, hipStream_t stream)
{
    if (B == 0 || T == 0) return hipSuccess;
    if (T > 0x7fffffffLL - DRUN * DG) return hipErrorInvalidValue;      // 32-bit sample indices inside a stream
    const int64_t tiles = (T + DRUN * DG - 1) / (DRUN * DG);
    const int64_t nblk = 8 * tiles * ((B + 7) / 8);
    if (nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(delay_apply_kernel<true>, dim3((unsigned)nblk), dim3(256), 0, stream, x, d, y, B, (int)T, dl_state, D,
                       warmup, err_flag, (unsigned)tiles);
    return hipGetLastError();
}

hipError_t launch_delay(const float *x, const float *d, float *y, int64_t B, int64_t T, float *dl_state, int D,
                        int warmup, int32_t *err_flag, hipStream_t stream)
{
    if (B == 0 || T == 0) return hipSuccess;
    hipError_t e = launch_delay_apply(x, d, y, B, T, dl_state, D, warmup, err_flag, stream);
    if (e != hipSuccess) return e;
    if (D > 0)
        hipLaunchKernelGGL(delay_update_kernel, dim3((unsigned)B), dim3(DU_THREADS), 0, stream, x, dl_state, T, D, err_flag);
    return hipGetLastError();
}

// the buffer update alone: behind the fused DiffDelRNN kernel (gru_mfma2.hip, FUSE), which interpolates but leaves the
// carried buffer to this launch -- it has to see the range flag of EVERY workgroup of that kernel
hipError_t launch_delay_update(const float *x, int64_t B, int64_t T, float *dl_state, int D, const int32_t *err_flag,
                               hipStream_t stream)
{
    if (B <= 0 || T <= 0 || D <= 0) return hipSuccess;
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

// K3s: this rank's four loss scalars from its per-stream ESR rows -- what the loss loop of code/test-model.py:386-398
// aggregates (sum over segments of the per-segment loss, segment count, and the raw sums): out4 = [sum_b ESR_b, B,
// sum_b err2_b, sum_b tgt2_b] with ESR_b = (err2_b / n) / (tgt2_b / n + eps).  ONE workgroup, thread i adds rows i, i + 256, ...
// in index order, fixed tree behind it: bit-reproducible, no atomics.  B rows of 16 bytes: nothing to optimise.
__global__ __launch_bounds__(256) void loss_scalars_kernel(const double *rows, int64_t B, double n, double eps, double *out4)
{
    double se = 0.0, s0 = 0.0, s1 = 0.0;
    for (int64_t b = threadIdx.x; b < B; b += 256) {
        const double e = rows[2 * b], t = rows[2 * b + 1];
        se += (e / n) / (t / n + eps);
        s0 += e;
        s1 += t;
    }
    __shared__ double red[3][256];
    red[0][threadIdx.x] = se; red[1][threadIdx.x] = s0; red[2][threadIdx.x] = s1;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st)
            for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out4[0] = red[0][0];
        out4[1] = (double)B;
        out4[2] = red[1][0];
        out4[3] = red[2][0];
    }
}

hipError_t launch_loss_scalars(const double *rows, int64_t B, double n, double eps, double *out4, hipStream_t stream)
{
    hipLaunchKernelGGL(loss_scalars_kernel, dim3(1), dim3(256), 0, stream, rows, B, n, eps, out4);
    return hipGetLastError();
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
// `skip`), then sum f(t-y)^2 and sum f(t)^2 per stream.  The recursion v[n] = R v[n-1] + u[n] is linear.
// One 256-thread block per stream; each of its four waves owns a CONTIGUOUS quarter of [skip, T) and streams through
// it in chunks of 1024 samples WITHOUT any workgroup barrier (round 1 shared every chunk between the four waves:
// two barriers per 4096 samples, 0.94 ms at 4096 x 65 536):
//   * a chunk is fetched with coalesced loads and handed to the lanes through the wave's own LDS rows (lane l gets
//     samples 16 l .. 16 l + 15; rows of 17 floats keep the reads conflict-free);
//   * every lane runs its 16 samples from zero state, a shuffle scan of the (R^16, end value) pairs gives each lane
//     the state entering it, the chunk's last state is carried to the next chunk in a register;
//   * quarters 1..3 do not know the state c entering them, so a wave accumulates for its quarter (started from zero
//     state)  S2 = sum v^2,  S1 = sum R^(k+1) v_k  and its end state E; with c the true sums are
//     S2 + 2 c S1 + c^2 C2  (C2 = sum R^(2(k+1)), a geometric series) and the state leaving is R^len c + E: one thread
//     chains the four quarters at the end.  fp64 accumulators.
// ---------------------------------------------------------------------------------------
constexpr int DCL = 16;   // samples per lane per chunk
constexpr int DCW = 64 * DCL;   // samples per wave and chunk

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
    __shared__ float s_e[4][64 * (DCL + 1)], s_t[4][64 * (DCL + 1)];
    __shared__ double q_sum[4][6];           // per quarter: S2e, S1e, Ee, S2t, S1t, Et
    float *se_ = s_e[wv], *st_ = s_t[wv];

    // this wave's quarter: whole chunks, the last quarter takes the remainder
    const int64_t n_all = T - skip;
    const int64_t chunks = (n_all + DCW - 1) / DCW, cq = (chunks + 3) / 4;
    const int64_t q0 = skip + (int64_t)wv * cq * DCW;
    const int64_t q1 = (skip + (int64_t)(wv + 1) * cq * DCW < T) ? skip + (int64_t)(wv + 1) * cq * DCW : T;

    double S2e = 0.0, S1e = 0.0, S2t = 0.0, S1t = 0.0;
    float ce = 0.0f, ct = 0.0f;              // state entering the chunk (the quarter starts from zero state)
    const float r_chunk = powf(R, (float)DCW);
    float wbase = powf(R, (float)(lane * DCL + 1));      // R^(k+1) of this lane's first sample of the chunk, k within the quarter
    for (int64_t c0 = q0; c0 < q1; c0 += DCW) {
        // all 32 loads of the chunk are issued before the first one is consumed (clamped addresses instead of
        // per-load branches: with a branch and an LDS store behind every load the compiler waited for each load in
        // turn -- 16 dependent HBM round trips per chunk, which is what held round 1's kernel at 0.94 ms)
        float tvv[DCL], yvv[DCL];
        const int64_t last = q1 - 1;
#pragma unroll
        for (int i = 0; i < DCL; ++i) {
            const int64_t n = c0 + i * 64 + lane;
            const int64_t nc = n < q1 ? n : last;
            tvv[i] = tb[nc];
            yvv[i] = yb[nc];
        }
#pragma unroll
        for (int i = 0; i < DCL; ++i) {
            const int idx = i * 64 + lane;
            const bool in = c0 + idx < q1;
            const float tv = in ? tvv[i] : 0.0f, yv = in ? yvv[i] : 0.0f;
            st_[(idx >> 4) * (DCL + 1) + (idx & 15)] = tv;
            se_[(idx >> 4) * (DCL + 1) + (idx & 15)] = tv - yv;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int64_t n0 = c0 + (int64_t)lane * DCL;
        float ue[DCL], ut[DCL];
        float pe = 0.0f, pt = 0.0f;          // the sample before this lane's run (input of the FIR part)
        if (n0 > skip && n0 - 1 < T) {
            if (lane > 0) { pt = st_[(lane - 1) * (DCL + 1) + DCL - 1]; pe = se_[(lane - 1) * (DCL + 1) + DCL - 1]; }
            else { pt = tb[n0 - 1]; pe = pt - yb[n0 - 1]; }
        }
        float fe = 0.0f, ft = 0.0f;
#pragma unroll
        for (int k = 0; k < DCL; ++k) {
            const int64_t n = n0 + k;
            float tv = 0.0f, ev = 0.0f;
            if (n < q1) { tv = st_[lane * (DCL + 1) + k]; ev = se_[lane * (DCL + 1) + k]; } else { tv = pt; ev = pe; }   // past the end: u = 0
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
        // state entering this lane: exclusive scan value applied to the chunk's incoming state
        float Ax = __shfl_up(A, 1), Bex = __shfl_up(Be, 1), Btx = __shfl_up(Bt, 1);
        if (lane == 0) { Ax = 1.0f; Bex = 0.0f; Btx = 0.0f; }
        const float le = __builtin_fmaf(Ax, ce, Bex), lt = __builtin_fmaf(Ax, ct, Btx);
        float wk = wbase;
#pragma unroll
        for (int k = 0; k < DCL; ++k) {
            if (n0 + k < q1) {
                const float ve = __builtin_fmaf(rp[k + 1], le, ue[k]), vt = __builtin_fmaf(rp[k + 1], lt, ut[k]);
                S2e += (double)ve * (double)ve;
                S2t += (double)vt * (double)vt;
                S1e += (double)(wk * ve);
                S1t += (double)(wk * vt);
            }
            wk *= R;
        }
        wbase *= r_chunk;
        // state leaving the chunk = the inclusive value of lane 63 applied to the incoming state
        const float Al = __shfl(A, 63), Bel = __shfl(Be, 63), Btl = __shfl(Bt, 63);
        ce = __builtin_fmaf(Al, ce, Bel);
        ct = __builtin_fmaf(Al, ct, Btl);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        S2e += __shfl_down(S2e, off); S1e += __shfl_down(S1e, off);
        S2t += __shfl_down(S2t, off); S1t += __shfl_down(S1t, off);
    }
    if (lane == 0) {
        q_sum[wv][0] = S2e; q_sum[wv][1] = S1e; q_sum[wv][2] = (double)ce;
        q_sum[wv][3] = S2t; q_sum[wv][4] = S1t; q_sum[wv][5] = (double)ct;
    }
    __syncthreads();
    if (tid == 0) {                          // chain the quarters in order
        const double Rd = (double)R, R2 = Rd * Rd;
        double tot_e = 0.0, tot_t = 0.0, c_e = 0.0, c_t = 0.0;
        for (int w = 0; w < 4; ++w) {
            const int64_t a0 = skip + (int64_t)w * cq * DCW;
            const int64_t a1 = (skip + (int64_t)(w + 1) * cq * DCW < T) ? skip + (int64_t)(w + 1) * cq * DCW : T;
            const double len = a1 > a0 ? (double)(a1 - a0) : 0.0;
            const double Aq = pow(Rd, len);
            const double C2 = R2 < 1.0 ? R2 * (1.0 - Aq * Aq) / (1.0 - R2) : len;      // sum_{k<len} R^(2(k+1))
            tot_e += q_sum[w][0] + 2.0 * c_e * q_sum[w][1] + c_e * c_e * C2;
            tot_t += q_sum[w][3] + 2.0 * c_t * q_sum[w][4] + c_t * c_t * C2;
            c_e = Aq * c_e + q_sum[w][2];
            c_t = Aq * c_t + q_sum[w][5];
        }
        out[2 * b + 0] = tot_e;
        out[2 * b + 1] = tot_t;
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
