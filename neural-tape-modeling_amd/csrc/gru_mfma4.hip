// GRU-HS[64] + head, MFMA variant 4 ("one wavefront per 4 streams"): exact fp32 on v_mfma_f32_4x4x1_16B_f32.
//
// MFMA2 (gru_mfma2.hip) gives 16 streams to a 4-wave workgroup and pays, per step, a workgroup barrier, the wait for
// the own LDS write ahead of it and the drain of 8-pass MFMAs (~165 of its 2151 cycles).  Here a wavefront owns 4
// streams and ALL 192 rows of W_hh, so nothing is exchanged between waves and there is no barrier at all:
//   * v_mfma_f32_4x4x1_16B_f32 = 16 independent 4x4 outer products: block blk of lane l = 4 blk + (l & 3).
//     A: lane 4 blk + i holds A[blk][i];  B: lane 4 blk + j holds B[blk][j];  D: lane 4 blk + j, register i =
//     D[blk][i][j]  (tools/ubench/mfma_4x4.hip).  Block blk accumulates rows 4 blk .. 4 blk + 3 of a gate (64 units =
//     16 blocks x 4) for the wave's 4 streams j: D[blk][i][j] += W[4 blk + i][k] * h[k][j], one k per MFMA, and
//     NOTHING forces the 16 blocks of one instruction to use the same k -- the k order is free per block because the
//     weights are resident, pre-permuted A operands (192 registers per lane, in the accumulator half of the file).
//   * so the new h feeds the next step straight out of the registers it was computed in: lane (blk, j) holds
//     h[4 blk + i][j] in register i, which IS B[blk][j] for k = 4 blk + i; the BLGP modifier (B lanes of 16-lane group
//     g' broadcast to all four groups: blgp 4 + g') routes the registers of the other three lane groups as well.
//     That is 16 of the 64 k of every block = 48 MFMAs = 384 cycles with no data movement at all;
//   * the other 48 k (the three other blocks of each lane group, 4 groups) come back from LDS: one ds_write_b128 of
//     the new h and 12 ds_read_b128 issued right behind it (same wave: LDS ops complete in order, no barrier), in
//     flight while the 48 register-fed MFMAs run;
//   * 2-pass MFMAs drain in a few cycles; the gate math (the same packed-fp32 block as MFMA2, 4 units x 1 stream per
//     lane) follows the 192 MFMAs (8 cycles each = 1536 cycles, the same matrix-pipe time as 48 x 16x16x4).
// One wave per SIMD at B = 4096 (1024 waves); a 64-thread workgroup, ~36 KB of LDS each.
#include "ntm_common.h"

#include <type_traits>

namespace ntm {

typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef NTM4_ABL
#define NTM4_ABL 0    // DIAGNOSTIC builds only (wrong results, timing ablations): 1 no LDS reads of h, 2 no gate math,
#endif                // 4 no head partial, 8 no tile housekeeping, 16 register-fed MFMAs only, 32 no LDS write of h

namespace m4 {
constexpr int SW = 4;             // streams per wavefront
constexpr int TT = 64;            // samples per x / y tile
constexpr int HS = 80;            // floats per stream row of the h exchange (64 + 16: conflict-free b128 reads)
constexpr int XS = TT + 4;        // x tile row
constexpr int YS = TT + 4;        // y partial row (16-B aligned rows)
constexpr int YP = SW * YS;       // floats per partial plane
constexpr int SMEM_FLOATS = SW * HS + 2 * SW * XS + 2 * 16 * YP;
}  // namespace m4

template <int BLGP>
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, BLGP);
}

__global__ __launch_bounds__(64) void gru_mfma4_kernel(GruArgs a)
{
    using namespace m4;
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    float *hs = smem;                    // [4 streams][80]
    float *xt = hs + SW * HS;            // [2][4][68]
    float *yp = xt + 2 * SW * XS;        // [2][16 planes][4][68]

    const int l = threadIdx.x;
    const int blk = l >> 2, j = l & 3, b4 = blk & 3;
    const int64_t s0 = (int64_t)blockIdx.x * SW;
    const int64_t T = a.T;
    const bool valid = (s0 + j) < a.B;
    constexpr float LOG2E = 1.44269504088896340736f;
    constexpr float SRZ = -LOG2E, SN = 2.0f * LOG2E;     // folded into the rows: sigmoid / tanh start at v_exp_f32

    // ---- resident A operands: lane (blk, ia = l & 3) holds row 4 blk + ia of each gate; column order per block:
    //      rounds 0..3  (register-fed, BLGP 4 + g'):  k = 16 g' + 4 b4 + i
    //      rounds 4..15 (LDS-fed, rho = 1..3, g'' = 0..3):  k = 16 g'' + 4 ((b4 + rho) & 3) + i
    float Ar[64], Az[64], An[64];
    {
        const int row = 4 * blk + j;         // the A layout's row-in-block index is the lane's low two bits as well
        const float *pr = a.w_hh + (size_t)(0 * kH + row) * kH;
        const float *pz = a.w_hh + (size_t)(1 * kH + row) * kH;
        const float *pn = a.w_hh + (size_t)(2 * kH + row) * kH;
#pragma unroll
        for (int rd = 0; rd < 16; ++rd) {
            const int k0 = rd < 4 ? 16 * rd + 4 * b4 : 16 * ((rd - 4) & 3) + 4 * ((b4 + 1 + ((rd - 4) >> 2)) & 3);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                Ar[4 * rd + i] = pr[k0 + i] * SRZ;
                Az[4 * rd + i] = pz[k0 + i] * SRZ;
                An[4 * rd + i] = pn[k0 + i] * SN;
            }
        }
    }
    // per-lane gate parameters of units 4 blk + i, as packed pairs (i = 0,1 | 2,3)
    f32x2 wir[2], wiz[2], win[2], br[2], bz[2], bin_[2], bhn[2], wo[2], hold[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int u = 4 * blk + i;
        wir[i >> 1][i & 1] = a.w_ih[u] * SRZ;
        wiz[i >> 1][i & 1] = a.w_ih[kH + u] * SRZ;
        win[i >> 1][i & 1] = a.w_ih[2 * kH + u] * SN;
        br[i >> 1][i & 1] = (a.b_ih[u] + a.b_hh[u]) * SRZ;
        bz[i >> 1][i & 1] = (a.b_ih[kH + u] + a.b_hh[kH + u]) * SRZ;
        bin_[i >> 1][i & 1] = a.b_ih[2 * kH + u] * SN;
        bhn[i >> 1][i & 1] = a.b_hh[2 * kH + u] * SN;
        wo[i >> 1][i & 1] = a.w_o[u];
        hold[i >> 1][i & 1] = (a.h_state && valid) ? a.h_state[(s0 + j) * kH + u] : 0.0f;
    }
    const float bo = a.b_o ? a.b_o[0] : 0.0f;

    // x tile: lane -> stream l >> 4, samples 4 (l & 15) .. + 3
    const int xs_ = l >> 4, xc = 4 * (l & 15);
    const bool x_vec_ok = ((reinterpret_cast<uintptr_t>(a.x) & 15) == 0) && ((a.xs & 3) == 0);
    const bool y_vec_ok = ((reinterpret_cast<uintptr_t>(a.y) & 15) == 0) && ((a.ys & 3) == 0);
    auto load_x_tile = [&](int64_t tile) -> f32x4 {
        const int64_t st = s0 + xs_, tt = tile * TT + xc;
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (st < a.B) {
            const float *p = a.x + st * a.xs + tt;
            if (x_vec_ok && tt + 3 < T) v = *(const f32x4 *)p;
            else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (tt + c < T) v[c] = p[c];
            }
        }
        return v;
    };
    auto store_x_tile = [&](int64_t tile, f32x4 v) { *(f32x4 *)&xt[(tile & 1) * SW * XS + xs_ * XS + xc] = v; };
    auto flush_y_tile = [&](int64_t tile) {
        const float *src = yp + (tile & 1) * 16 * YP + xs_ * YS + xc;
        f32x4 v = {bo, bo, bo, bo};
#pragma unroll
        for (int pl = 0; pl < 16; ++pl) v += *(const f32x4 *)(src + pl * YP);      // fixed order: deterministic
        const int64_t gs = s0 + xs_, gt = tile * TT + xc;
        if (gs < a.B) {
            float *dst = a.y + gs * a.ys + gt;
            if (y_vec_ok && gt + 3 < T) *(f32x4 *)dst = v;
            else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (gt + c < T) dst[c] = v[c];
            }
        }
    };

    f32x4 xr = load_x_tile(0);
    store_x_tile(0, xr);
    int64_t next_flush = 0;

    float hT[4] = {hold[0][0], hold[0][1], hold[1][0], hold[1][1]};
    float *const hwr = hs + j * HS + 4 * blk;
    const float *hrd[3];
#pragma unroll
    for (int rho = 1; rho <= 3; ++rho) hrd[rho - 1] = hs + j * HS + 4 * ((b4 + rho) & 3);
    *(f32x4 *)hwr = (f32x4){hT[0], hT[1], hT[2], hT[3]};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    f32x4 hL[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) hL[q] = *(const f32x4 *)(hrd[q >> 2] + 16 * (q & 3));
    f32x2 cr[2], cz[2], gi[2];
    {
        const float x0 = xt[j * XS];
        const f32x2 xx = {x0, x0};
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            cr[p] = __builtin_elementwise_fma(wir[p], xx, br[p]);
            cz[p] = __builtin_elementwise_fma(wiz[p], xx, bz[p]);
            gi[p] = __builtin_elementwise_fma(win[p], xx, bin_[p]);
        }
    }
    float *const yp_lane = yp + blk * YP + j * YS;

    // hk_c as in gru_mfma2.hip: tile housekeeping at compile-time positions in whole tiles (no phase tests in the step);
    // -1 = test the phase at run time (ragged last tile)
    auto step = [&](const int64_t t, auto hk_c) {
        constexpr int HK = decltype(hk_c)::value;
        const int ph = (int)(t & 63);
        const int64_t tile = t >> 6;
        f32x4 acc_r = {cr[0][0], cr[0][1], cr[1][0], cr[1][1]};
        f32x4 acc_n = {bhn[0][0], bhn[0][1], bhn[1][0], bhn[1][1]};
        f32x4 acc_z = {cz[0][0], cz[0][1], cz[1][0], cz[1][1]};
        // ---- rounds 0..3: B operands are the registers the previous step's gates left, routed by BLGP -----------
#define NTM4_ROUND(G)                                                        \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                      \
            acc_r = mfma4<4 + G>(Ar[4 * G + i], hT[i], acc_r);               \
            acc_n = mfma4<4 + G>(An[4 * G + i], hT[i], acc_n);               \
            acc_z = mfma4<4 + G>(Az[4 * G + i], hT[i], acc_z);               \
        }
        NTM4_ROUND(0) NTM4_ROUND(1) NTM4_ROUND(2) NTM4_ROUND(3)
#undef NTM4_ROUND
        // x of step t+1 (its tile was staged at ph 34 of the previous tile at the latest)
        float xn = xt[(((t + 1) >> 6) & 1) * SW * XS + j * XS + (int)((t + 1) & 63)];
        // ---- rounds 4..15: the other blocks' units, read back from LDS behind the previous step's write ---------
#pragma unroll
        for (int q = 0; q < ((NTM4_ABL & 16) ? 0 : 12); ++q) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc_r = mfma4<0>(Ar[16 + 4 * q + i], hL[q][i], acc_r);
                acc_n = mfma4<0>(An[16 + 4 * q + i], hL[q][i], acc_n);
                acc_z = mfma4<0>(Az[16 + 4 * q + i], hL[q][i], acc_z);
            }
        }
        asm volatile("" : "+v"(acc_r), "+v"(acc_n), "+v"(acc_z), "+v"(xn));

        // tile housekeeping, once per 64 steps each
        if (!(NTM4_ABL & 8)) {
            if (HK == 1 || (HK < 0 && ph == 2)) {
                if ((tile + 1) * TT < T) xr = load_x_tile(tile + 1);
                if (t > 65) { flush_y_tile(next_flush); ++next_flush; }
            } else if (HK == 2 || (HK < 0 && ph == 34)) {
                if ((tile + 1) * TT < T) store_x_tile(tile + 1, xr);
            }
        }

        // ---- the VALU block ------------------------------------------------------------------------------------
        const f32x2 xx = {xn, xn};
        f32x2 ncr[2], ncz[2], ngi[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            ncr[p] = __builtin_elementwise_fma(wir[p], xx, br[p]);
            ncz[p] = __builtin_elementwise_fma(wiz[p], xx, bz[p]);
            ngi[p] = __builtin_elementwise_fma(win[p], xx, bin_[p]);
        }
        const f32x2 one = {1.0f, 1.0f};
        f32x2 hn[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const f32x2 ar = {acc_r[2 * p], acc_r[2 * p + 1]}, an = {acc_n[2 * p], acc_n[2 * p + 1]};
            const f32x2 az = {acc_z[2 * p], acc_z[2 * p + 1]};
            if (NTM4_ABL & 2) { hn[p] = (ar + an) + (az + gi[p]); continue; }
            f32x2 er = {__builtin_amdgcn_exp2f(ar[0]), __builtin_amdgcn_exp2f(ar[1])};
            f32x2 ez = {__builtin_amdgcn_exp2f(az[0]), __builtin_amdgcn_exp2f(az[1])};
            er += one; ez += one;
            const f32x2 r = {__builtin_amdgcn_rcpf(er[0]), __builtin_amdgcn_rcpf(er[1])};
            const f32x2 z = {__builtin_amdgcn_rcpf(ez[0]), __builtin_amdgcn_rcpf(ez[1])};
            const f32x2 pn = __builtin_elementwise_fma(r, an, gi[p]);
            f32x2 en = {__builtin_amdgcn_exp2f(pn[0]), __builtin_amdgcn_exp2f(pn[1])};
            en += one;
            const f32x2 rn = {__builtin_amdgcn_rcpf(en[0]), __builtin_amdgcn_rcpf(en[1])};
            const f32x2 n = __builtin_elementwise_fma((f32x2){-2.0f, -2.0f}, rn, one);   // tanh
            hn[p] = __builtin_elementwise_fma(z, hold[p] - n, n);                         // n + z (h - n)
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) { hold[p] = hn[p]; cr[p] = ncr[p]; cz[p] = ncz[p]; gi[p] = ngi[p]; }
        hT[0] = hn[0][0]; hT[1] = hn[0][1]; hT[2] = hn[1][0]; hT[3] = hn[1][1];
        // publish h_t inside the wave and fetch the other blocks' units for the next step (in-order LDS: the reads
        // see the write; the fences only pin the compiler)
        if (!(NTM4_ABL & 32)) *(f32x4 *)hwr = (f32x4){hT[0], hT[1], hT[2], hT[3]};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (!(NTM4_ABL & 1)) {
#pragma unroll
            for (int q = 0; q < 12; ++q) hL[q] = *(const f32x4 *)(hrd[q >> 2] + 16 * (q & 3));
        } else {
#pragma unroll
            for (int q = 0; q < 12; ++q) hL[q] = (f32x4){hT[q & 3], hT[0], hT[1], hT[2]};
        }
        // head partial of y_t over this lane's four units
        if (!(NTM4_ABL & 4)) {
            const f32x2 pp = __builtin_elementwise_fma(hn[1], wo[1], hn[0] * wo[0]);
            yp_lane[(tile & 1) * 16 * YP + ph] = pp[0] + pp[1];
        }
    };
    {
        using H0 = std::integral_constant<int, 0>;
        using HA = std::integral_constant<int, 1>;
        using HB = std::integral_constant<int, 2>;
        using HR = std::integral_constant<int, -1>;
        const int64_t full = (T / TT) * TT;
        for (int64_t t0 = 0; t0 < full; t0 += TT) {
            step(t0, H0{}); step(t0 + 1, H0{}); step(t0 + 2, HA{});
            for (int p = 3; p < 34; ++p) step(t0 + p, H0{});
            step(t0 + 34, HB{});
            for (int p = 35; p < TT; ++p) step(t0 + p, H0{});
        }
        for (int64_t t = full; t < T; ++t) step(t, HR{});
    }

    // ---- epilogue: remaining y tiles, final state ------------------------------------------------------------------
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    while (next_flush * TT < T) { flush_y_tile(next_flush); ++next_flush; }
    if (a.h_state && valid) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a.h_state[(s0 + j) * kH + 4 * blk + i] = hold[i >> 1][i & 1];
    }
}

hipError_t launch_gru_mfma4(const GruArgs &a, hipStream_t stream)
{
    if (a.B == 0) return hipSuccess;
    const unsigned grid = (unsigned)((a.B + m4::SW - 1) / m4::SW);
    hipLaunchKernelGGL(gru_mfma4_kernel, dim3(grid), dim3(64), 0, stream, a);
    return hipGetLastError();
}

}  // namespace ntm
