// Arithmetic of the time-varying fractional delay line (TimeVaryingDelayLine.forward, code/model.py:269-320, in closed
// form), shared by the streaming pass (aux_kernels.hip, delay_apply_kernel) and the pass fused into the GRU kernel's
// output flush (gru_mfma2.hip, FUSE).  Every product and sum rounds separately in the reference's order (fma
// contraction off), so both users give the bits of the reference's O(T*D) unfold formulation.
#pragma once
#include "ntm_common.h"

namespace ntm {

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

typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));   // 8-byte access at 4-byte alignment

// One output sample from its two adjacent taps t = (x[n-k-1], x[n-k]), valid when 0 <= d < D (so k + 1 <= D) and both
// taps exist: the same roundings as delay_sample_fast without its branches -- here w_b = 1 - |(k+1) - d| = d - k lies in
// [0, 1) and w_a = 1 - |k - d| in (0, 1], so relu() is the identity and the absolute values have known signs
// (|(k+1) - d| = (k+1) - d,  |k - d| = -(k - d): negation is exact); a zero weight contributes w * x = +-0, which the
// leading `0 +` absorbs exactly as skipping the tap does.
__device__ __forceinline__ float delay_pair(float dn, f32x2u t)
{
#pragma clang fp contract(off)
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    const float kf = floorf(dn);
    // both taps side by side (lane 0: m = k + 1, lane 1: m = k), one packed op per line
    const f32x2_ m = (f32x2_){kf, kf} + (f32x2_){1.0f, 0.0f};
    const f32x2_ tt = m - (f32x2_){dn, dn};                          // (k+1) - d  >  0,   k - d  <=  0
    // 1 - |tt| with the known signs: 1 - tt[0], 1 + tt[1].  As ONE packed fma with the factors (-1, +1): the product
    // is exact, so the fma rounds exactly what the subtraction / addition rounds (hipcc builds a mixed-sign packed add
    // out of five instructions)
    const f32x2_ w = __builtin_elementwise_fma(tt, (f32x2_){-1.0f, 1.0f}, (f32x2_){1.0f, 1.0f});
    const f32x2_ pr = w * (f32x2_){t[0], t[1]};
    const float acc = 0.0f + pr[0];
    return acc + pr[1];
}

// The general form for a thread's 4 samples n0 .. n0+3 of one stream, sample by sample, for the lanes delay_pair() does
// not cover; also evaluates the range check for them.  full: all 4 samples exist, their delays are in dn and the results
// go to `out` (the caller stores them); otherwise (ragged tail of a row) the delays are fetched and the outputs stored
// here.  Rare: it sits behind a wave-level branch.
__device__ __forceinline__ void delay_general4(const float *xb, const float *db, float *yb, const float *bb, int D, int n0, int T,
                                               bool full, f32x4 dn, f32x4 &out, int &bad)
{
    const float Dmax = (float)D;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (n0 + c >= T) break;
        const float dc = full ? dn[c] : db[n0 + c];
        bad |= !(dc <= Dmax);                      // NaN too
        const float v = delay_sample(xb, bb, D, n0 + c, dc);
        out[c] = v;                                  // (the caller may still want the value: loss sums on y)
        if (!full) yb[n0 + c] = v;
    }
}

}  // namespace ntm
