// GRU-HS[64] + head, exact fp32, "MFMA + partner VALU" hybrid (NTM_GRU_MFMA3).
//
// Measured on MI355X (tools/ubench/valu_costs.hip): a VALU instruction costs the wave that also issues f32
// MFMAs ~8 cycles of matrix-pipe time, but a VALU-only wave sharing the SIMD with an MFMA-only wave proceeds at
// ~10 cycles per instruction while the MFMA wave loses only ~3 %.  The fp32 matrix pipe and the VALU are
// therefore separate resources ACROSS waves.  This kernel gives every SIMD two waves of the same 16-stream
// group:
//   role A (waves 0-3) is gru_mfma2_kernel's wave: own-quarter-first MFMAs, gates, h publish -- but it only
//          multiplies 3 of the 4 quarters of h (36 MFMAs instead of 48, plus 3 per K-step handed back);
//   role B (waves 4-7, same SIMD as A's wave w = wave-4) multiplies the remaining quarter (w+2)&3 for the same
//          16 units x 16 streams on the VALU (v_pk_fma_f32, KB of its 4 K-steps: 24 KB instructions, weights
//          resident in 48 KB VGPRs) while A's MFMAs run, and hands the 12 partial sums per lane to A through LDS.
//          It also owns everything that is not on the recurrence's critical path: the head partial (of the
//          previous step, from the h values it has just read), x staging, y flush.
// Two workgroup barriers per step: BAR1 = "h_{t-1} is complete in LDS" (A waits behind its first 3 MFMAs),
// BAR2 = "B's partial sums are in LDS" (A waits behind all but its last 6 MFMAs, which cover the read latency).
#include "ntm_common.h"

#include <type_traits>

#ifndef NTM3_KB
#define NTM3_KB 4
#endif

namespace ntm {

namespace m3 {
constexpr int SG = 16, TT = 64;
constexpr int HB_J = 20, HB_K = SG * HB_J, HB = 4 * HB_K;      // h exchange: [4 kslot][16 stream][20]
constexpr int XS = TT + 1;                                      // x tile row
constexpr int YS = TT + 4, YP_Q = SG * YS;                      // y partial plane [16][68]
constexpr int OFF_XB = 2 * HB;                                  // 2560
constexpr int OFF_PB = OFF_XB + 2 * SG * XS;                    // 4640: partial sums [4 w][3 g][64 lanes][4]
constexpr int OFF_YP = OFF_PB + 4 * 3 * 64 * 4;                 // 7712: y partials [2][4 planes][YP_Q]
constexpr int SMEM_FLOATS = OFF_YP + 2 * 4 * YP_Q;              // 16 416 floats = 65 664 B
static_assert(OFF_PB % 4 == 0 && OFF_YP % 4 == 0, "16-B aligned LDS regions");
}  // namespace m3

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KB>
__global__ __launch_bounds__(512, 2) void gru_mfma3_kernel(GruArgs a)
{
    using namespace m3;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *hb = smem, *xb = smem + OFF_XB, *pb = smem + OFF_PB, *yp = smem + OFF_YP;

    const int tid = threadIdx.x, l = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = wv & 3;                  // unit block / partner index
    const bool roleB = wv >= 4;
    const int q = l >> 4, j = l & 15;
    const int64_t s0 = (int64_t)blockIdx.x * SG;
    const int64_t T = a.T;
    const bool valid = (s0 + j) < a.B;
    constexpr float LOG2E = 1.44269504088896340736f;
    constexpr float SRZ = -LOG2E, SN = 2.0f * LOG2E;    // folded into W / biases: gates start with v_exp_f32
    const int qb = (w + 2) & 3;                          // the quarter of h that role B multiplies
    const float *const wr_ = a.w_hh + (size_t)(0 * kH) * kH;
    const float *const wz_ = a.w_hh + (size_t)(1 * kH) * kH;
    const float *const wn_ = a.w_hh + (size_t)(2 * kH) * kH;

    if (!roleB) {
        // =============================== role A: matrix pipe + gates ===============================
        constexpr int NA = 12 + (4 - KB);              // K-steps on the matrix pipe
        // K-step s of quarter Q: units 16Q + 4k + (s&3); order: own quarter, (w+1)&3, (w+3)&3, rest of qb
        float Ar[NA], Az[NA], An[NA];
        {
            const int row = 16 * w + j;
#pragma unroll
            for (int sg = 0; sg < NA; ++sg) {
                const int Q = sg < 4 ? w : sg < 8 ? ((w + 1) & 3) : sg < 12 ? ((w + 3) & 3) : qb;
                const int i = sg < 12 ? (sg & 3) : KB + (sg - 12);
                const int col = 16 * Q + 4 * q + i;
                Ar[sg] = wr_[row * kH + col] * SRZ; Az[sg] = wz_[row * kH + col] * SRZ; An[sg] = wn_[row * kH + col] * SN;
            }
        }
        f32x2 wir[2], wiz[2], win[2], br[2], bz[2], bin_[2], bhn[2], hold[2];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int u = 16 * w + 4 * q + v;
            wir[v >> 1][v & 1] = a.w_ih[u] * SRZ;
            wiz[v >> 1][v & 1] = a.w_ih[kH + u] * SRZ;
            win[v >> 1][v & 1] = a.w_ih[2 * kH + u] * SN;
            br[v >> 1][v & 1] = (a.b_ih[u] + a.b_hh[u]) * SRZ;
            bz[v >> 1][v & 1] = (a.b_ih[kH + u] + a.b_hh[kH + u]) * SRZ;
            bin_[v >> 1][v & 1] = a.b_ih[2 * kH + u] * SN;
            bhn[v >> 1][v & 1] = a.b_hh[2 * kH + u] * SN;
            hold[v >> 1][v & 1] = (a.h_state && valid) ? a.h_state[(s0 + j) * kH + u] : 0.0f;
        }
        float hT[4] = {hold[0][0], hold[0][1], hold[1][0], hold[1][1]};
        float *const hrow = hb + q * HB_K + j * HB_J;
        *(f32x4 *)(hrow + 4 * w) = (f32x4){hT[0], hT[1], hT[2], hT[3]};
        __syncthreads();   // P1: x tile 0 staged by role B, h_0 published
        f32x2 cr[2], cz[2], gi[2];
        {
            const float x0 = xb[j * XS];
            const f32x2 xx = {x0, x0};
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                cr[p] = __builtin_elementwise_fma(wir[p], xx, br[p]);
                cz[p] = __builtin_elementwise_fma(wiz[p], xx, bz[p]);
                gi[p] = __builtin_elementwise_fma(win[p], xx, bin_[p]);
            }
        }
        const float *const prd = pb + (w * 3) * 256 + 4 * l;   // + g*256

        auto step = [&](const int64_t t, auto cur_c) {
            constexpr int cur = decltype(cur_c)::value;
            float hB[NA];
#pragma unroll
            for (int i = 0; i < 4; ++i) hB[i] = hT[i];
            f32x4 acc_r = {cr[0][0], cr[0][1], cr[1][0], cr[1][1]};
            f32x4 acc_n = {bhn[0][0], bhn[0][1], bhn[1][0], bhn[1][1]};
            f32x4 acc_z = {cz[0][0], cz[0][1], cz[1][0], cz[1][1]};
            acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(Ar[0], hB[0], acc_r, 0, 0, 0);
            acc_n = __builtin_amdgcn_mfma_f32_16x16x4f32(An[0], hB[0], acc_n, 0, 0, 0);
            acc_z = __builtin_amdgcn_mfma_f32_16x16x4f32(Az[0], hB[0], acc_z, 0, 0, 0);
            // BAR1: every A wave's ds_write_b128 of h_{t-1} has completed
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+v"(acc_r), "+v"(acc_n), "+v"(acc_z)::"memory");
            {
                const f32x4 v1 = *(const f32x4 *)(hrow + cur * HB + 4 * ((w + 1) & 3));
                const f32x4 v3 = *(const f32x4 *)(hrow + cur * HB + 4 * ((w + 3) & 3));
#pragma unroll
                for (int i = 0; i < 4; ++i) { hB[4 + i] = v1[i]; hB[8 + i] = v3[i]; }
                if constexpr (KB < 4) {
                    const f32x4 v2 = *(const f32x4 *)(hrow + cur * HB + 4 * qb);
#pragma unroll
                    for (int i = KB; i < 4; ++i) hB[12 + i - KB] = v2[i];
                }
            }
            float xn = xb[(((t + 1) >> 6) & 1) * SG * XS + j * XS + (int)((t + 1) & 63)];
#pragma unroll
            for (int sg = 1; sg < NA - 2; ++sg) {
                acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(Ar[sg], hB[sg], acc_r, 0, 0, 0);
                acc_n = __builtin_amdgcn_mfma_f32_16x16x4f32(An[sg], hB[sg], acc_n, 0, 0, 0);
                acc_z = __builtin_amdgcn_mfma_f32_16x16x4f32(Az[sg], hB[sg], acc_z, 0, 0, 0);
                if (sg == 3) asm volatile("" : "+v"(acc_r), "+v"(acc_n), "+v"(acc_z));
            }
            // BAR2: role B's partial sums are in LDS; the last 6 MFMAs cover the read latency
#ifndef NTM3_NOBAR2
            asm volatile("s_barrier" : "+v"(acc_r), "+v"(acc_n), "+v"(acc_z)::"memory");
            f32x4 pr4 = *(const f32x4 *)(prd + 0 * 256);
            f32x4 pn4 = *(const f32x4 *)(prd + 1 * 256);
            f32x4 pz4 = *(const f32x4 *)(prd + 2 * 256);
#else
            f32x4 pr4 = {0, 0, 0, 0}, pn4 = pr4, pz4 = pr4;   // diagnostic build (KB = 0 only): no hand-over
#endif
#pragma unroll
            for (int sg = NA - 2; sg < NA; ++sg) {
                acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(Ar[sg], hB[sg], acc_r, 0, 0, 0);
                acc_n = __builtin_amdgcn_mfma_f32_16x16x4f32(An[sg], hB[sg], acc_n, 0, 0, 0);
                acc_z = __builtin_amdgcn_mfma_f32_16x16x4f32(Az[sg], hB[sg], acc_z, 0, 0, 0);
            }
            asm volatile("" : "+v"(acc_r), "+v"(acc_n), "+v"(acc_z), "+v"(xn), "+v"(pr4), "+v"(pn4), "+v"(pz4));

            // ---- VALU block: input terms of step t+1, partner's partial sums, gates, blend, publish ----
            const f32x2 xx = {xn, xn};
            f32x2 ncr[2], ncz[2], ngi[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                ncr[p] = __builtin_elementwise_fma(wir[p], xx, br[p]);
                ncz[p] = __builtin_elementwise_fma(wiz[p], xx, bz[p]);
                ngi[p] = __builtin_elementwise_fma(win[p], xx, bin_[p]);
            }
            asm volatile("" : "+v"(ncr[0]), "+v"(ncr[1]), "+v"(ncz[0]), "+v"(ncz[1]), "+v"(ngi[0]), "+v"(ngi[1]),
                              "+v"(acc_r), "+v"(acc_n), "+v"(acc_z));
            const f32x2 one = {1.0f, 1.0f};
            f32x2 hn[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const f32x2 ar = (f32x2){acc_r[2 * p], acc_r[2 * p + 1]} + (f32x2){pr4[2 * p], pr4[2 * p + 1]};
                const f32x2 an = (f32x2){acc_n[2 * p], acc_n[2 * p + 1]} + (f32x2){pn4[2 * p], pn4[2 * p + 1]};
                const f32x2 az = (f32x2){acc_z[2 * p], acc_z[2 * p + 1]} + (f32x2){pz4[2 * p], pz4[2 * p + 1]};
                f32x2 er = {__builtin_amdgcn_exp2f(ar[0]), __builtin_amdgcn_exp2f(ar[1])};
                f32x2 ez = {__builtin_amdgcn_exp2f(az[0]), __builtin_amdgcn_exp2f(az[1])};
                er += one; ez += one;
                const f32x2 r = {__builtin_amdgcn_rcpf(er[0]), __builtin_amdgcn_rcpf(er[1])};
                const f32x2 z = {__builtin_amdgcn_rcpf(ez[0]), __builtin_amdgcn_rcpf(ez[1])};
                const f32x2 pn = __builtin_elementwise_fma(r, an, gi[p]);
                f32x2 en = {__builtin_amdgcn_exp2f(pn[0]), __builtin_amdgcn_exp2f(pn[1])};
                en += one;
                const f32x2 rn = {__builtin_amdgcn_rcpf(en[0]), __builtin_amdgcn_rcpf(en[1])};
                const f32x2 n = __builtin_elementwise_fma((f32x2){-2.0f, -2.0f}, rn, one);
                hn[p] = __builtin_elementwise_fma(z, hold[p] - n, n);
            }
#pragma unroll
            for (int p = 0; p < 2; ++p) { hold[p] = hn[p]; cr[p] = ncr[p]; cz[p] = ncz[p]; gi[p] = ngi[p]; }
            hT[0] = hn[0][0]; hT[1] = hn[0][1]; hT[2] = hn[1][0]; hT[3] = hn[1][1];
            *(f32x4 *)(hrow + (cur ^ 1) * HB + 4 * w) = (f32x4){hT[0], hT[1], hT[2], hT[3]};
        };
        for (int64_t t = 0; t < T; t += 2) {
            step(t, std::integral_constant<int, 0>{});
            if (t + 1 < T) step(t + 1, std::integral_constant<int, 1>{});
        }
        // epilogue barriers (role B computes the head of the last step in between)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // E1: h_{T-1} complete
        asm volatile("s_barrier" ::: "memory");                          // E2
        if (a.h_state && valid) {
#pragma unroll
            for (int v = 0; v < 4; ++v) a.h_state[(s0 + j) * kH + 16 * w + 4 * q + v] = hold[v >> 1][v & 1];
        }
    } else {
        // =============================== role B: VALU quarter + head + I/O ==========================
        const int tb = tid - 256;          // 0..255 among the role-B threads
        // weights of quarter qb for units 16w+4q+v: Wb[g][p][k][i] = {W_g[u(2p)][c], W_g[u(2p+1)][c]}, c = 16qb+4k+i
        f32x2 Wb[3][2][4][KB > 0 ? KB : 1];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < KB; ++i) {
                    const int u0 = 16 * w + 4 * q + 2 * p, c = 16 * qb + 4 * k + i;
                    Wb[0][p][k][i] = (f32x2){wr_[u0 * kH + c] * SRZ, wr_[(u0 + 1) * kH + c] * SRZ};
                    Wb[1][p][k][i] = (f32x2){wn_[u0 * kH + c] * SN, wn_[(u0 + 1) * kH + c] * SN};
                    Wb[2][p][k][i] = (f32x2){wz_[u0 * kH + c] * SRZ, wz_[(u0 + 1) * kH + c] * SRZ};
                }
        f32x2 wob[4][2];                   // head weights of quarter qb: units 16qb + 4k + i
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            wob[k][0] = (f32x2){a.w_o[16 * qb + 4 * k + 0], a.w_o[16 * qb + 4 * k + 1]};
            wob[k][1] = (f32x2){a.w_o[16 * qb + 4 * k + 2], a.w_o[16 * qb + 4 * k + 3]};
        }
        const float bo = a.b_o ? a.b_o[0] : 0.0f;

        auto load_x_tile = [&](int64_t tile, float (&xr)[4]) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int e = tb + 256 * c;
                const int64_t st = s0 + (e >> 6), tt = tile * TT + (e & 63);
                xr[c] = (st < a.B && tt < T) ? a.x[st * a.xs + tt] : 0.0f;
            }
        };
        auto store_x_tile = [&](int64_t tile, const float (&xr)[4]) {
            float *dst = xb + (tile & 1) * SG * XS;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int e = tb + 256 * c;
                dst[(e >> 6) * XS + (e & 63)] = xr[c];
            }
        };
        const bool y_vec_ok = ((reinterpret_cast<uintptr_t>(a.y) & 15) == 0) && ((a.ys & 3) == 0);
        auto flush_y_tile = [&](int64_t tile) {
            const float *src = yp + (tile & 1) * 4 * YP_Q + (tb >> 4) * YS + 4 * (tb & 15);
            f32x4 v = {bo, bo, bo, bo};
#pragma unroll
            for (int pl = 0; pl < 4; ++pl) v += *(const f32x4 *)(src + pl * YP_Q);
            const int64_t gs = s0 + (tb >> 4), gt = tile * TT + 4 * (tb & 15);
            if (gs < a.B) {
                float *dst = a.y + gs * a.ys + gt;
                if (y_vec_ok && gt + 3 < T) {
                    *(f32x4 *)dst = v;
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (gt + c < T) dst[c] = v[c];
                }
            }
        };
        float xr[4];
        load_x_tile(0, xr);
        store_x_tile(0, xr);
        int64_t next_flush = 0;
        __syncthreads();   // P1

        const float *const hq = hb + j * HB_J + 4 * qb;        // + cur*HB + k*HB_K
        float *const pwr = pb + (w * 3) * 256 + 4 * l;           // + g*256
        float *const ypl = yp + w * YP_Q + j * YS;               // plane w (all four lane groups store the same value)

        // head partial of step t-1 over quarter qb (from the h values just read) -> y partial plane w
        auto head = [&](const f32x4 (&hv)[4], int64_t tp) {
            f32x2 s2 = {0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s2 = __builtin_elementwise_fma(wob[k][0], (f32x2){hv[k][0], hv[k][1]}, s2);
                s2 = __builtin_elementwise_fma(wob[k][1], (f32x2){hv[k][2], hv[k][3]}, s2);
            }
            ypl[((tp >> 6) & 1) * 4 * YP_Q + (tp & 63)] = s2[0] + s2[1];
        };

        for (int64_t t = 0; t < T; ++t) {
            const int cur = (int)(t & 1);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                    // BAR1
            f32x4 hv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) hv[k] = *(const f32x4 *)(hq + cur * HB + k * HB_K);
            f32x2 acc[3][2];
#pragma unroll
            for (int g = 0; g < 3; ++g) { acc[g][0] = (f32x2){0.0f, 0.0f}; acc[g][1] = acc[g][0]; }
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < KB; ++i) {
                    const f32x2 h2 = {hv[k][i], hv[k][i]};
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        acc[g][0] = __builtin_elementwise_fma(Wb[g][0][k][i], h2, acc[g][0]);
                        acc[g][1] = __builtin_elementwise_fma(Wb[g][1][k][i], h2, acc[g][1]);
                    }
                }
#pragma unroll
            for (int g = 0; g < 3; ++g)
                *(f32x4 *)(pwr + g * 256) = (f32x4){acc[g][0][0], acc[g][0][1], acc[g][1][0], acc[g][1][1]};
            if (t > 0) head(hv, t - 1);
#ifndef NTM3_NOBAR2
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                    // BAR2
#endif
            // off-critical-path work, while role A runs its gate block
            const int ph = (int)(t & 63);
            const int64_t tile = t >> 6;
            if (ph == 0 && t >= 64) { flush_y_tile(next_flush); ++next_flush; }
            if (ph == 2) {
                if ((tile + 1) * TT < T) load_x_tile(tile + 1, xr);
            } else if (ph == 34) {
                if ((tile + 1) * TT < T) store_x_tile(tile + 1, xr);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                        // E1
        {
            f32x4 hv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) hv[k] = *(const f32x4 *)(hq + (int)(T & 1) * HB + k * HB_K);
            head(hv, T - 1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                        // E2
        while (next_flush * TT < T) { flush_y_tile(next_flush); ++next_flush; }
    }
}

hipError_t launch_gru_mfma3(const GruArgs &a, hipStream_t stream)
{
    constexpr size_t smem = 96 * 1024;   // > half a CU's LDS: one workgroup (8 waves, 2 per SIMD) per CU
    static_assert(m3::SMEM_FLOATS * sizeof(float) <= smem, "LDS carve-up");
    hipError_t e = hipFuncSetAttribute((const void *)gru_mfma3_kernel<NTM3_KB>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    const unsigned grid = (unsigned)((a.B + m3::SG - 1) / m3::SG);
    hipLaunchKernelGGL(gru_mfma3_kernel<NTM3_KB>, dim3(grid), dim3(512), smem, stream, a);
    return hipGetLastError();
}

}  // namespace ntm
