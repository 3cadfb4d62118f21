// GRU-HS[64] + affine head, persistent over the whole sequence (K1 in docs/DESIGN_measurement_log_r1_r5.md).
//
// Replaces torch.nn.GRU(1,64,batch_first=True) + torch.nn.Linear(64,1) as the reference calls them
// at code/model.py:81-82 / :412-413.  Per sample and stream (gate row order r,z,n):
//     gi = W_ih x_t + b_ih ; gh = W_hh h + b_hh
//     r = s(gi_r+gh_r)  z = s(gi_z+gh_z)  n = tanh(gi_n + r*gh_n)  h' = n + z*(h-n)  y_t = W_o h' + b_o
//
// Two variants, both exact fp32:
//   gru_mfma_kernel  16 streams per 4-wave workgroup; step t is the [192x64]x[64x16] product
//                    W_hh . H_t on v_mfma_f32_16x16x4_f32 (bitwise an fp32 fma chain), W_hh resident
//                    in VGPRs as MFMA A-operands, h exchanged between the 4 waves through LDS.
//   gru_valu_kernel  NS streams per wavefront; lane j owns hidden unit j with its three W_hh rows in
//                    192 VGPRs, h broadcast through LDS, v_fma_f32 GEMV, DPP wave reduction for the head.
#include "ntm_common.h"

namespace ntm {

// =====================================================================================
// Variant 1: MFMA.  Geometry of one workgroup (256 threads = 4 waves, one per SIMD):
//   wave w owns hidden units [16w,16w+16) for all three gates and 16 streams.
//   lane l: q = l>>4, j = l&15.
//   D = A.B + C per gate with   A[i][k] = W_g[16w+i][u(s,k)]   (i = l&15, k = l>>4)
//                               B[k][c] = h[stream c][u(s,k)]   (k = l>>4, c = l&15)
//   K-step s (16 of them) uses the unit permutation u(s,k) = 16k+s, so lane (q,j) needs
//   h[stream j][16q .. 16q+15]: 16 contiguous floats = 4 ds_read_b128.
//   C/D: lane (q,j) holds stream j, units 16w+4q+v (v=0..3) -> gates are lane-local and the new
//   h leaves as ONE ds_write_b128 into q-block w of the exchange buffer.
// LDS exchange buffer hb[2][4 qblk][16 stream][20]: row stride 20 floats (5 x 16 B) makes both the
//   b128 reads (16-lane groups hold 16 distinct streams) and the b128 writes conflict-free.
// =====================================================================================
constexpr int SG = 16;             // streams per workgroup
constexpr int TT = 64;             // samples per x / y staging tile
constexpr int HB_J = 20;           // floats per (qblk, stream) row (16 + 4 pad)
constexpr int HB_Q = SG * HB_J;    // 320
constexpr int HB = 4 * HB_Q;       // 1280 floats per buffer
constexpr int XS = TT + 1;         // padded tile row
constexpr int YP_Q = SG * XS;      // 1040 (== 16 mod 32: q=0/1 land on disjoint banks)
constexpr int MFMA_SMEM_FLOATS = 2 * HB + 2 * SG * XS + 2 * 4 * YP_Q;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// STAMP = true is a DIAGNOSTIC build (ntm_debug_gru_stamps): s_memtime stamps split every step into
// read | phase A | phase B | tail | write | barrier and the per-wave sums go to a.dbg.  Never timed.
#define NTM_STAMP(k)                                                                        \
    if constexpr (STAMP) {                                                                  \
        unsigned long long now_;                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");        \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        seg[k] += now_ - last_;                                                             \
        last_ = now_;                                                                       \
    }

template <bool STAMP>
__global__ __launch_bounds__(256, 1) void gru_mfma_kernel(GruArgs a)
{
    unsigned long long seg[6] = {0, 0, 0, 0, 0, 0}, last_ = 0;
    (void)seg; (void)last_;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *hb = smem;                   // [2][HB]
    float *xb = hb + 2 * HB;            // [2][SG][XS]
    float *yp = xb + 2 * SG * XS;       // [2][4][SG][XS]

    const int tid = threadIdx.x;
    const int l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = l >> 4, j = l & 15;
    const int64_t s0 = (int64_t)blockIdx.x * SG;
    const int64_t T = a.T;
    const bool valid = (s0 + j) < a.B;

    // ---- resident operands -------------------------------------------------------------
    float Ar[16], Az[16], An[16];
    {
        const int row = 16 * w + j;
        const float *pr = a.w_hh + (size_t)(0 * kH + row) * kH + 16 * q;
        const float *pz = a.w_hh + (size_t)(1 * kH + row) * kH + 16 * q;
        const float *pn = a.w_hh + (size_t)(2 * kH + row) * kH + 16 * q;
#pragma unroll
        for (int s = 0; s < 16; ++s) { Ar[s] = pr[s]; Az[s] = pz[s]; An[s] = pn[s]; }
    }
    float wir[4], wiz[4], win[4], br[4], bz[4], bin_[4], bhn[4], hold[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int u = 16 * w + 4 * q + v;
        wir[v] = a.w_ih[u];
        wiz[v] = a.w_ih[kH + u];
        win[v] = a.w_ih[2 * kH + u];
        br[v] = a.b_ih[u] + a.b_hh[u];
        bz[v] = a.b_ih[kH + u] + a.b_hh[kH + u];
        bin_[v] = a.b_ih[2 * kH + u];
        bhn[v] = a.b_hh[2 * kH + u];
        hold[v] = (a.h_state && valid) ? a.h_state[(s0 + j) * kH + u] : 0.0f;
    }
    float wo[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) wo[s] = a.w_o[16 * q + s];
    const float bo = a.b_o ? a.b_o[0] : 0.0f;

    // h_0 into exchange buffer 0
    *(f32x4 *)&hb[w * HB_Q + j * HB_J + 4 * q] = (f32x4){hold[0], hold[1], hold[2], hold[3]};

    // x tile loader: element e = tid + 256*c -> stream e>>6, sample e&63 (a wave reads 256 B rows)
    auto load_x_tile = [&](int64_t tile, float (&xr)[4]) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int e = tid + 256 * c;
            const int64_t st = s0 + (e >> 6), tt = tile * TT + (e & 63);
            xr[c] = (st < a.B && tt < T) ? a.x[st * a.xs + tt] : 0.0f;
        }
    };
    auto store_x_tile = [&](int64_t tile, const float (&xr)[4]) {
        float *dst = xb + (tile & 1) * SG * XS;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int e = tid + 256 * c;
            dst[(e >> 6) * XS + (e & 63)] = xr[c];
        }
    };
    auto flush_y_tile = [&](int64_t tile) {
        const float *src = yp + (tile & 1) * 4 * YP_Q;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int e = tid + 256 * c;
            const int st = e >> 6, tt = e & 63;
            const float v = ((src[0 * YP_Q + st * XS + tt] + src[1 * YP_Q + st * XS + tt]) +
                             (src[2 * YP_Q + st * XS + tt] + src[3 * YP_Q + st * XS + tt])) + bo;
            const int64_t gs = s0 + st, gt = tile * TT + tt;
            if (gs < a.B && gt < T) a.y[gs * a.ys + gt] = v;
        }
    };

    float xr[4];
    load_x_tile(0, xr);
    store_x_tile(0, xr);
    int64_t next_flush = 0;
    __syncthreads();

    if constexpr (STAMP) {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_)::"memory");
    }
    for (int64_t t = 0; t <= T; ++t) {
        const int cur = (int)(t & 1);
        // (1) h_{t-1} of stream j, units 16q..16q+15: the B operands of all 16 K-steps
        float hB[16];
        {
            const f32x4 *src = (const f32x4 *)&hb[cur * HB + q * HB_Q + j * HB_J];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 v4 = src[c];
                hB[4 * c + 0] = v4.x; hB[4 * c + 1] = v4.y; hB[4 * c + 2] = v4.z; hB[4 * c + 3] = v4.w;
            }
        }
        if (t == T) {
            // last iteration: only the head of step T-1 is left
            if (w == 0) {
                float p = 0.0f;
#pragma unroll
                for (int s = 0; s < 16; ++s) p = __builtin_fmaf(wo[s], hB[s], p);
                const int64_t tp = t - 1;
                yp[((tp >> 6) & 1) * 4 * YP_Q + q * YP_Q + j * XS + (tp & 63)] = p;
            }
            break;
        }

        // (2) tile housekeeping, once per 64 steps each
        const int ph = (int)(t & 63);
        const int64_t tile = t >> 6;
        if (ph == 1) {
            if (t > 64) { flush_y_tile(next_flush); ++next_flush; }
            if ((tile + 1) * TT < T) load_x_tile(tile + 1, xr);
        } else if (ph == 33) {
            if ((tile + 1) * TT < T) store_x_tile(tile + 1, xr);
        }
        NTM_STAMP(0)   // LDS reads of h_{t-1} (+ tile housekeeping)

        // (3) phase A: W_hr.h and W_hn.h as two interleaved MFMA chains (32 MFMAs).  Everything else
        //     that does not depend on them rides in their shadow, at most three VALU ops per MFMA
        //     pair: the r/z/n input terms (C operands), and the head partial of the PREVIOUS step
        //     (y_{t-1} over units 16q..16q+15; the four q-partials are summed at tile flush).
        //     Empty asm statements pin the order: a tied value must be complete before the pin and
        //     may only be consumed after it, so neither MFMAs nor VALU ops drift between gaps.
        const float xt = xb[(tile & 1) * SG * XS + j * XS + ph];
        f32x4 acc_r, acc_za, acc_zb, acc_n, gin;
        float hp = 0.0f;
#pragma unroll
        for (int v = 0; v < 4; ++v) acc_n[v] = bhn[v];
        acc_n = mfma16(An[0], hB[0], acc_n);
#pragma unroll
        for (int v = 0; v < 4; ++v) acc_r[v] = __builtin_fmaf(wir[v], xt, br[v]);
        acc_r = mfma16(Ar[0], hB[0], acc_r);
#pragma unroll
        for (int s = 1; s < 16; ++s) {
            hp = __builtin_fmaf(wo[s - 1], hB[s - 1], hp);
            if (s == 15) hp = __builtin_fmaf(wo[15], hB[15], hp);
            if (s <= 4) acc_za[s - 1] = __builtin_fmaf(wiz[s - 1], xt, bz[s - 1]);
            else if (s <= 8) gin[s - 5] = __builtin_fmaf(win[s - 5], xt, bin_[s - 5]);
            acc_n = mfma16(An[s], hB[s], acc_n);
            acc_r = mfma16(Ar[s], hB[s], acc_r);
            if (s < 4) asm volatile("" : "+v"(acc_r), "+v"(acc_n), "+v"(hp));
            else if (s < 8) asm volatile("" : "+v"(acc_r), "+v"(acc_n), "+v"(hp), "+v"(acc_za));
            else asm volatile("" : "+v"(acc_r), "+v"(acc_n), "+v"(hp), "+v"(acc_za), "+v"(gin));
        }
        if (t > 0 && w == 0) {
            const int64_t tp = t - 1;
            yp[((tp >> 6) & 1) * 4 * YP_Q + q * YP_Q + j * XS + (tp & 63)] = hp;
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) acc_zb[v] = 0.0f;
        NTM_STAMP(1)   // phase A issue
        // (4) phase B: W_hz.h as two chains (16 MFMAs).  Underneath, the r and n gate math as 44
        //     micro-ops, stage-major over the lane's 4 elements so that consecutive ops of one
        //     element are >= one MFMA slot apart (no dependent-VALU stall in front of an MFMA):
        //       e = -log2e*acc_r | 2^e | 1+e | 1/e (= r) | r*acc_n+gi_n | 2log2e*e | 2^e | 1+e | 1/e |
        //       n = 1-2e | dn = h-n
        float e[4], dn[4];
        acc_za = mfma16(Az[0], hB[0], acc_za);
#pragma unroll
        for (int m = 1; m < 16; ++m) {
#pragma unroll
            for (int i = 3 * (m - 1); i < 3 * m && i < 44; ++i) {
                const int st = i >> 2, v = i & 3;
                if (st == 0) e[v] = acc_r[v] * -1.44269504088896340736f;
                else if (st == 1 || st == 6) e[v] = __builtin_amdgcn_exp2f(e[v]);
                else if (st == 2 || st == 7) e[v] = 1.0f + e[v];
                else if (st == 3 || st == 8) e[v] = __builtin_amdgcn_rcpf(e[v]);
                else if (st == 4) e[v] = __builtin_fmaf(e[v], acc_n[v], gin[v]);
                else if (st == 5) e[v] = e[v] * 2.88539008177792681472f;
                else if (st == 9) e[v] = __builtin_fmaf(-2.0f, e[v], 1.0f);
                else dn[v] = hold[v] - e[v];
            }
            if (m & 1) acc_zb = mfma16(Az[m], hB[m], acc_zb);
            else acc_za = mfma16(Az[m], hB[m], acc_za);
            asm volatile("" : "+v"(acc_za), "+v"(acc_zb), "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]));
        }
        asm volatile("" : "+v"(dn[0]), "+v"(dn[1]), "+v"(dn[2]), "+v"(dn[3]));
        NTM_STAMP(2)   // phase B issue
        // (5) phase C (exposed tail): z gate and blend  h' = n + z (h - n)
        f32x4 hn;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float z = sigmoid_f32(acc_za[v] + acc_zb[v]);
            hold[v] = __builtin_fmaf(z, dn[v], e[v]);
            hn[v] = hold[v];
        }
        asm volatile("" : "+v"(hn));
        NTM_STAMP(3)   // tail VALU (includes waiting for the last MFMAs)
        // (7) publish h_t into the other buffer; one barrier per step
        *(f32x4 *)&hb[(cur ^ 1) * HB + w * HB_Q + j * HB_J + 4 * q] = hn;
        NTM_STAMP(4)   // ds_write + its completion
        __syncthreads();
        NTM_STAMP(5)   // barrier
    }
    if constexpr (STAMP) {
        if (a.dbg && l == 0)
            for (int k = 0; k < 6; ++k) a.dbg[((size_t)blockIdx.x * 4 + w) * 12 + k] = seg[k];   // 12 slots per wave: the MFMA2 stamps use all of them
    }

    __syncthreads();
    while (next_flush * TT < T) { flush_y_tile(next_flush); ++next_flush; }

    if (a.h_state && valid) {
#pragma unroll
        for (int v = 0; v < 4; ++v) a.h_state[(s0 + j) * kH + 16 * w + 4 * q + v] = hold[v];
    }
}

// =====================================================================================
// Variant 2: VALU, NS streams per wavefront (workgroup = one wave).
// =====================================================================================
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_shift_add(float v)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
    return v + __builtin_bit_cast(float, moved);
}

// Sum over the 64 lanes; the total is valid in lane 63 (gfx9 row_shr / row_bcast scan).
__device__ __forceinline__ float wave_sum_lane63(float v)
{
    v = dpp_shift_add<0x111, 0xf>(v);  // row_shr:1
    v = dpp_shift_add<0x112, 0xf>(v);  // row_shr:2
    v = dpp_shift_add<0x114, 0xf>(v);  // row_shr:4
    v = dpp_shift_add<0x118, 0xf>(v);  // row_shr:8
    v = dpp_shift_add<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3
    v = dpp_shift_add<0x143, 0xc>(v);  // row_bcast:31 -> rows 2,3
    return v;
}

template <int NS>
__global__ __launch_bounds__(64, 2) void gru_valu_kernel(GruArgs a)
{
    __shared__ __attribute__((aligned(16))) float hs[NS][kH];
    const int j = threadIdx.x;  // lane == hidden unit
    const int64_t s0 = (int64_t)blockIdx.x * NS;
    const int64_t T = a.T;

    float wr[kH], wz[kH], wn[kH];
    {
        const f32x4 *pr = (const f32x4 *)(a.w_hh + (size_t)(0 * kH + j) * kH);
        const f32x4 *pz = (const f32x4 *)(a.w_hh + (size_t)(1 * kH + j) * kH);
        const f32x4 *pn = (const f32x4 *)(a.w_hh + (size_t)(2 * kH + j) * kH);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const f32x4 r4 = pr[c], z4 = pz[c], n4 = pn[c];
#pragma unroll
            for (int e = 0; e < 4; ++e) { wr[4 * c + e] = r4[e]; wz[4 * c + e] = z4[e]; wn[4 * c + e] = n4[e]; }
        }
    }
    const float wir = a.w_ih[j], wiz = a.w_ih[kH + j], win = a.w_ih[2 * kH + j];
    const float br = a.b_ih[j] + a.b_hh[j], bz = a.b_ih[kH + j] + a.b_hh[kH + j];
    const float bin_ = a.b_ih[2 * kH + j], bhn = a.b_hh[2 * kH + j];
    const float wo = a.w_o[j];
    const float bo = a.b_o ? a.b_o[0] : 0.0f;

    float h[NS];
    bool valid[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        valid[i] = (s0 + i) < a.B;
        h[i] = (a.h_state && valid[i]) ? a.h_state[(s0 + i) * kH + j] : 0.0f;
        hs[i][j] = h[i];
    }
    __syncthreads();

    float xnext[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) xnext[i] = (valid[i] && j < T) ? a.x[(s0 + i) * a.xs + j] : 0.0f;

    for (int64_t t0 = 0; t0 < T; t0 += 64) {
        float xt[NS], yt[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            xt[i] = xnext[i];
            yt[i] = 0.0f;
            const int64_t tn = t0 + 64 + j;
            xnext[i] = (valid[i] && tn < T) ? a.x[(s0 + i) * a.xs + tn] : 0.0f;
        }
        const int nt = (int)((T - t0) < 64 ? (T - t0) : 64);
        for (int tt = 0; tt < nt; ++tt) {
            float ar[NS], az[NS], an[NS], gin[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const float xs = __builtin_bit_cast(
                    float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xt[i]), tt));
                ar[i] = __builtin_fmaf(wir, xs, br);
                az[i] = __builtin_fmaf(wiz, xs, bz);
                gin[i] = __builtin_fmaf(win, xs, bin_);
                an[i] = bhn;
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) {
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const f32x4 hv = ((const f32x4 *)hs[i])[c];  // same address in every lane: LDS broadcast
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        ar[i] = __builtin_fmaf(wr[4 * c + e], hv[e], ar[i]);
                        az[i] = __builtin_fmaf(wz[4 * c + e], hv[e], az[i]);
                        an[i] = __builtin_fmaf(wn[4 * c + e], hv[e], an[i]);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const float r = sigmoid_f32(ar[i]);
                const float n = tanh_f32(__builtin_fmaf(r, an[i], gin[i]));
                const float z = sigmoid_f32(az[i]);
                h[i] = __builtin_fmaf(z, h[i] - n, n);
                hs[i][j] = h[i];
                const float tot = wave_sum_lane63(wo * h[i]);
                const float yv = __builtin_bit_cast(
                    float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tot), 63)) + bo;
                yt[i] = (j == tt) ? yv : yt[i];
            }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < NS; ++i)
            if (valid[i] && t0 + j < T) a.y[(s0 + i) * a.ys + t0 + j] = yt[i];
    }
    if (a.h_state) {
#pragma unroll
        for (int i = 0; i < NS; ++i)
            if (valid[i]) a.h_state[(s0 + i) * kH + j] = h[i];
    }
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
hipError_t launch_gru_mfma(const GruArgs &a, hipStream_t stream)
{
    // Ask for more than half of the CU's 160 KiB LDS so that two workgroups never share a CU
    // (each workgroup wants all four SIMDs to itself).
    static const size_t smem_bytes = 96 * 1024;
    static_assert(MFMA_SMEM_FLOATS * sizeof(float) <= 96 * 1024, "LDS carve-up");
    {
        hipError_t e = hipFuncSetAttribute((const void *)gru_mfma_kernel<false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_bytes);
        if (e != hipSuccess) return e;
    }
    const unsigned grid = (unsigned)((a.B + SG - 1) / SG);
    if (a.dbg) {
        hipError_t e = hipFuncSetAttribute((const void *)gru_mfma_kernel<true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(gru_mfma_kernel<true>, dim3(grid), dim3(256), smem_bytes, stream, a);
    } else {
        hipLaunchKernelGGL(gru_mfma_kernel<false>, dim3(grid), dim3(256), smem_bytes, stream, a);
    }
    return hipGetLastError();
}

hipError_t launch_gru_valu(const GruArgs &a, hipStream_t stream)
{
    constexpr int NS = 2;
    const unsigned grid = (unsigned)((a.B + NS - 1) / NS);
    hipLaunchKernelGGL(gru_valu_kernel<NS>, dim3(grid), dim3(64), 0, stream, a);
    return hipGetLastError();
}

}  // namespace ntm
