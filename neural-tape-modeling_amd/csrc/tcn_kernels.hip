// K4: builder-defined causal dilated-Conv1d TCN (BASELINE configs[3]; the reference has no TCN).
//   block:  out[n][co] = PReLU(b[co] + sum_{ci,k} W[ci][k][co] in[n-(K-1-k)dil][ci]) + sum_ci R[ci][co] in[n][ci]
// Activations between blocks are channels-last [B][T][32] so that a lane fetches 8 consecutive input
// channels with two 16-B loads.  Blocks with 32 input channels run on the matrix pipe (exact fp32
// v_mfma_f32_16x16x4_f32): M = 16 output channels (A = weights, resident in VGPRs: 112 per lane),
// N = 16 samples (B = input), K = 32x13 taps + 32 residual; the 1-channel first block and the 1x1 output
// conv are plain VALU kernels (< 4 % of the flops).
//
// Measured (512 x 65 536, MI355X): 12.7 ms per 32->32 block = 75 TFLOP/s, the matrix pipe 50 % busy, 39 % of
// the wave cycles in s_waitcnt.  Every input row is fetched 13 x (taps) x 2 (the two waves that share a
// sample range) = 26 times into L1; at 112 GB per block that L1-fill traffic, not HBM or the MFMA rate, is
// the bound, for every dilation alike.  Next step: polyphase tile order (16 outputs
// n = p + m d per tile) so that consecutive taps reuse the same rows from L1/LDS.
#include "ntm_common.h"

#include <type_traits>

namespace ntm {

__device__ float tcn_zeros[32];   // zero page for taps that fall before the start of a stream

constexpr int TC = 32;    // channels
constexpr int TK = 13;    // kernel size
constexpr int TCH = 8192; // samples per workgroup (weights are loaded once per workgroup)

// ---- first block: 1 input channel, x [B][T] -> out [B][T][32] -------------------------------------
// thread -> (sample, group of 4 output channels): a wave writes 8 samples x 128 B = 1 KiB contiguous.
__global__ __launch_bounds__(256) void tcn_first_kernel(const float *x, float *out, const float *W, const float *bias,
                                                        const float *alpha, const float *R, int dil, int64_t T)
{
    const int64_t b = blockIdx.x;
    const int c4 = threadIdx.x & 7;
    const int64_t n = (int64_t)blockIdx.y * 32 + (threadIdx.x >> 3);
    if (n >= T) return;
    const float *xb = x + b * T;
    f32x4 acc = *(const f32x4 *)(bias + 4 * c4);
    for (int k = 0; k < TK; ++k) {
        const int64_t src = n - (int64_t)(TK - 1 - k) * dil;
        const float xv = src >= 0 ? xb[src] : 0.0f;
        const f32x4 wv = *(const f32x4 *)(W + k * TC + 4 * c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(wv[e], xv, acc[e]);
    }
    const float x0 = xb[n];
    const f32x4 al = *(const f32x4 *)(alpha + 4 * c4), rv = *(const f32x4 *)(R + 4 * c4);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(rv[e], x0, acc[e] >= 0.0f ? acc[e] : al[e] * acc[e]);
    *(f32x4 *)(out + (b * T + n) * TC + 4 * c4) = v;
}

// ---- 32 -> 32 channel block on the matrix pipe -----------------------------------------------------
// workgroup = (stream b, chunk of TCH samples), 4 waves: wave w computes output channels 16(w&1)..+15 for
// the samples 64(w>>1)..+63 of every 128-sample iteration (4 N-tiles of 16 samples).
// A[i][kslot] (lane i = l&15, kslot = l>>4):  conv K-step (k,s): W[ci = 8 kslot + s][k][co = 16 mt + i]
// B[kslot][j] (lane j = l&15, kslot = l>>4):  in[n0 + j - (12-k) dil][ci = 8 kslot + s]   (8 ci per lane = 2 x 16 B)
// D: lane (q = l>>4, j): sample n0 + j, channels 16 mt + 4 q + v  -> one 16-B store.
__global__ __launch_bounds__(256, 1) void tcn_block_mfma_kernel(const float *in, float *out, const float *W,
                                                                const float *bias, const float *alpha, const float *R,
                                                                int dil, int64_t T)
{
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = w & 1, ng = w >> 1;
    const int q = l >> 4, j = l & 15;
    const int64_t b = blockIdx.x;
    const int64_t c0 = (int64_t)blockIdx.y * TCH;
    const float *ib = in + b * T * TC;
    float *ob = out + b * T * TC;

    float Aw[TK][8], Ar[8];
#pragma unroll
    for (int k = 0; k < TK; ++k)
#pragma unroll
        for (int s = 0; s < 8; ++s) Aw[k][s] = W[((8 * q + s) * TK + k) * TC + 16 * mt + j];
#pragma unroll
    for (int s = 0; s < 8; ++s) Ar[s] = R[(8 * q + s) * TC + 16 * mt + j];
    f32x4 bi, al;
#pragma unroll
    for (int v = 0; v < 4; ++v) { bi[v] = bias[16 * mt + 4 * q + v]; al[v] = alpha[16 * mt + 4 * q + v]; }

    const int64_t cend = (c0 + TCH < T) ? c0 + TCH : T;
    const int lane_off = j * TC + 8 * q;             // floats, loop-invariant
    const int64_t halo = (int64_t)(TK - 1) * dil;
    // one 128-sample iteration; CHECK = false on interior iterations (every tap inside [0,T)): the tap
    // address is a wave-uniform base plus the invariant lane offset, no per-lane arithmetic at all
    auto iteration = [&](const int64_t it0, auto check_c) {
        constexpr bool CHECK = decltype(check_c)::value;
        const int64_t nb = it0 + 64 * ng;            // first sample of this wave's four N-tiles
        f32x4 acc[4], res[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { acc[nt] = bi; res[nt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; }
        auto load_tap = [&](int k, f32x4 (&lo)[4], f32x4 (&hi)[4]) {
            const int64_t shift = (int64_t)(TK - 1 - k) * dil;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const float *base = ib + (nb + 16 * nt - shift) * TC;     // wave-uniform
                const f32x4 *p = (const f32x4 *)(base + lane_off);
                if constexpr (CHECK) {
                    const int64_t src = nb + 16 * nt + j - shift;
                    if (!(src >= 0 && src < T)) p = (const f32x4 *)(tcn_zeros + 8 * q);
                }
                lo[nt] = p[0]; hi[nt] = p[1];
            }
        };
        f32x4 blo[2][4], bhi[2][4];
        load_tap(0, blo[0], bhi[0]);
#pragma unroll
        for (int k = 0; k < TK; ++k) {
            const int cb = k & 1;
            // the loads of tap k+1 are requested before the 32 MFMAs of tap k
            if (k + 1 < TK) load_tap(k + 1, blo[cb ^ 1], bhi[cb ^ 1]);
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const float bv = s < 4 ? blo[cb][nt][s] : bhi[cb][nt][s - 4];
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Aw[k][s], bv, acc[nt], 0, 0, 0);
                    if (k == TK - 1) res[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ar[s], bv, res[nt], 0, 0, 0);
                }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int64_t n = nb + 16 * nt + j;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float u = acc[nt][e];
                v[e] = (u >= 0.0f ? u : al[e] * u) + res[nt][e];
            }
            if (!CHECK || n < T) *(f32x4 *)(ob + n * TC + 16 * mt + 4 * q) = v;
        }
    };
    for (int64_t it0 = c0; it0 < cend; it0 += 128) {
        if (it0 >= halo && it0 + 128 <= T) iteration(it0, std::false_type{});
        else iteration(it0, std::true_type{});
    }
}

// ---- 1x1 output conv: [B][T][32] -> y [B][T] --------------------------------------------------------
__global__ __launch_bounds__(256) void tcn_out_kernel(const float *in, float *y, const float *ow, const float *ob,
                                                      int64_t T)
{
    const int64_t b = blockIdx.x;
    const int64_t n = (int64_t)blockIdx.y * blockDim.x + threadIdx.x;
    if (n >= T) return;
    const f32x4 *p = (const f32x4 *)(in + (b * T + n) * TC);
    float acc = ob[0];
#pragma unroll
    for (int c4 = 0; c4 < TC / 4; ++c4) {
        const f32x4 v = p[c4];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_fmaf(ow[4 * c4 + e], v[e], acc);
    }
    y[b * T + n] = acc;
}

hipError_t launch_tcn(const float *params, int L, int C, int K, const int *dil, const float *x, float *y, int64_t B,
                      int64_t T, float *scratch, hipStream_t stream)
{
    if (B == 0 || T == 0) return hipSuccess;
    if (C != TC || K != TK) return hipErrorInvalidValue;
    float *bufA = scratch, *bufB = scratch + (size_t)B * C * T;
    const float *p = params;
    const dim3 grid1((unsigned)B, (unsigned)((T + 255) / 256));
    const dim3 gridf((unsigned)B, (unsigned)((T + 31) / 32));
    const dim3 gridm((unsigned)B, (unsigned)((T + TCH - 1) / TCH));
    const float *in = x;
    int cin = 1;
    for (int l = 0; l < L; ++l) {
        const float *W = p;      p += (size_t)C * cin * K;
        const float *bias = p;   p += C;
        const float *alpha = p;  p += C;
        const float *R = p;      p += (size_t)C * cin;
        float *out = (l & 1) ? bufB : bufA;
        if (cin == 1) hipLaunchKernelGGL(tcn_first_kernel, gridf, dim3(256), 0, stream, in, out, W, bias, alpha, R, dil[l], T);
        else hipLaunchKernelGGL(tcn_block_mfma_kernel, gridm, dim3(256), 0, stream, in, out, W, bias, alpha, R, dil[l], T);
        in = out;
        cin = C;
    }
    if (cin != C) return hipErrorInvalidValue;   // L == 0
    hipLaunchKernelGGL(tcn_out_kernel, grid1, dim3(256), 0, stream, in, y, p, p + C, T);
    return hipGetLastError();
}

}  // namespace ntm
