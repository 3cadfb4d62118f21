// K1s: GRU-HS[H] + head for the SMALL hidden sizes H = 8, 16, 32 -- the reference's constructor default is
// hidden_size = 8 (code/model.py:22), its training default 16 (code/train.py:50); every shipped checkpoint is
// HS[64] and runs on the matrix-pipe / low-latency kernels instead.
//
// One wavefront advances S = 64 / H streams, no workgroup barrier anywhere:
//   lane (s, u) = (lane / H, lane % H) owns hidden unit u of stream s: rows u of W_r, W_z, W_n resident in 3H VGPRs
//   (with -log2e / 2 log2e folded in, so sigmoid / tanh start at v_exp_f32), v_pk_fma_f32 GEMV over the H values
//   of h, which every lane of the group fetches as H/4 ds_read_b128 from the wave's LDS copy (same address within
//   the group: LDS broadcast); gates lane-local; head y_t = w_o . h_t + b_o by a DPP row reduction inside the
//   H-lane group (row_shr 1,2,4[,8][, row_bcast15]).  x and y move in 64-sample tiles through LDS, coalesced.
// Exact fp32; chunked and one-shot launches agree bit for bit (no implicit contraction, fixed summation order).
#include "ntm_common.h"

namespace ntm {

typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr float LOG2E = 1.44269504088896340736f;

__device__ __forceinline__ void lds_fence_wave_s()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
    return v + __builtin_bit_cast(float, moved);
}

// sum over the H lanes of a group; valid in the group's LAST lane
template <int H>
__device__ __forceinline__ float group_sum_last(float v)
{
    v = dpp_add<0x111, 0xf>(v);                          // row_shr:1
    v = dpp_add<0x112, 0xf>(v);                          // row_shr:2
    v = dpp_add<0x114, 0xf>(v);                          // row_shr:4
    if constexpr (H >= 16) v = dpp_add<0x118, 0xf>(v);   // row_shr:8
    if constexpr (H >= 32) v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 -> rows 1,3
    return v;
}

template <int H>
__global__ __launch_bounds__(64) void gru_small_kernel(GruArgs a)
{
#pragma clang fp contract(off)
    constexpr int S = 64 / H;       // streams per wavefront
    constexpr int TT = 64;          // samples per x / y tile
    __shared__ __attribute__((aligned(16))) float hs[S][H];
    __shared__ float xt[S][TT + 1];
    __shared__ float yt[S][TT + 1];

    const int lane = threadIdx.x, s = lane / H, u = lane % H;
    const int64_t s0 = (int64_t)blockIdx.x * S;
    const int64_t T = a.T;
    const bool valid = (s0 + s) < a.B;

    constexpr float SRZ = -LOG2E, SN = 2.0f * LOG2E;
    f32x2 Wr[H / 2], Wz[H / 2], Wn[H / 2];
#pragma unroll
    for (int k = 0; k < H / 2; ++k) {
        const float *pr = a.w_hh + (size_t)(0 * H + u) * H + 2 * k;
        const float *pz = a.w_hh + (size_t)(1 * H + u) * H + 2 * k;
        const float *pn = a.w_hh + (size_t)(2 * H + u) * H + 2 * k;
        Wr[k] = (f32x2){pr[0] * SRZ, pr[1] * SRZ};
        Wz[k] = (f32x2){pz[0] * SRZ, pz[1] * SRZ};
        Wn[k] = (f32x2){pn[0] * SN, pn[1] * SN};
    }
    const float wir = a.w_ih[u] * SRZ, wiz = a.w_ih[H + u] * SRZ, win = a.w_ih[2 * H + u] * SN;
    const float br = (a.b_ih[u] + a.b_hh[u]) * SRZ, bz = (a.b_ih[H + u] + a.b_hh[H + u]) * SRZ;
    const float bin_ = a.b_ih[2 * H + u] * SN, bhn = a.b_hh[2 * H + u] * SN;
    const float wo = a.w_o[u];
    const float bo = a.b_o ? a.b_o[0] : 0.0f;
    float hold = (a.h_state && valid) ? a.h_state[(s0 + s) * H + u] : 0.0f;
    hs[s][u] = hold;

    // tile 0 into LDS, tile 1 parked in registers (lane i <-> sample i of the tile, one row per stream)
    float xnext[S];
#pragma unroll
    for (int ss = 0; ss < S; ++ss) {
        const bool ok = (s0 + ss) < a.B;
        xt[ss][lane] = (ok && lane < T) ? a.x[(s0 + ss) * a.xs + lane] : 0.0f;
        xnext[ss] = (ok && TT + lane < T) ? a.x[(s0 + ss) * a.xs + TT + lane] : 0.0f;
    }
    lds_fence_wave_s();

    for (int64_t t0 = 0; t0 < T; t0 += TT) {
        const int nt = (int)((T - t0) < TT ? (T - t0) : TT);
        for (int tt = 0; tt < nt; ++tt) {
            const float x = xt[s][tt];
            f32x2 ar0 = {0.0f, 0.0f}, ar1 = ar0, az0 = ar0, az1 = ar0, an0 = ar0, an1 = ar0;
#pragma unroll
            for (int c = 0; c < H / 4; ++c) {
                const f32x4 hv = *(const f32x4 *)&hs[s][4 * c];
                const f32x2 ha = {hv[0], hv[1]}, hb = {hv[2], hv[3]};
                ar0 = __builtin_elementwise_fma(Wr[2 * c], ha, ar0); ar1 = __builtin_elementwise_fma(Wr[2 * c + 1], hb, ar1);
                az0 = __builtin_elementwise_fma(Wz[2 * c], ha, az0); az1 = __builtin_elementwise_fma(Wz[2 * c + 1], hb, az1);
                an0 = __builtin_elementwise_fma(Wn[2 * c], ha, an0); an1 = __builtin_elementwise_fma(Wn[2 * c + 1], hb, an1);
            }
            const f32x2 sr = ar0 + ar1, sz = az0 + az1, sn = an0 + an1;
            const float pr_ = __builtin_fmaf(wir, x, br) + (sr[0] + sr[1]);
            const float pz_ = __builtin_fmaf(wiz, x, bz) + (sz[0] + sz[1]);
            const float gh = bhn + (sn[0] + sn[1]);
            const float gi = __builtin_fmaf(win, x, bin_);
            // gates on pre-scaled arguments: r, z = 1/(1 + 2^p);  n = 1 - 2/(1 + 2^q)
            const float r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pr_));
            const float z = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pz_));
            const float en = __builtin_amdgcn_exp2f(__builtin_fmaf(r, gh, gi));
            const float n = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + en), 1.0f);
            hold = __builtin_fmaf(z, hold - n, n);
            lds_fence_wave_s();                         // every lane has read h_{t-1}
            hs[s][u] = hold;
            const float yv = group_sum_last<H>(wo * hold) + bo;
            if (u == H - 1) yt[s][tt] = yv;
            lds_fence_wave_s();
        }
        // flush the y tile, bring in the next x tile, fetch the one after
#pragma unroll
        for (int ss = 0; ss < S; ++ss) {
            const bool ok = (s0 + ss) < a.B;
            if (ok && lane < nt) a.y[(s0 + ss) * a.ys + t0 + lane] = yt[ss][lane];
            xt[ss][lane] = xnext[ss];
            const int64_t nx = t0 + 2 * TT + lane;
            xnext[ss] = (ok && nx < T) ? a.x[(s0 + ss) * a.xs + nx] : 0.0f;
        }
        lds_fence_wave_s();
    }
    if (a.h_state && valid) a.h_state[(s0 + s) * H + u] = hold;
}

}   // namespace

hipError_t launch_gru_small(const GruArgs &a, int H, hipStream_t stream)
{
    if (a.B == 0) return hipSuccess;
    const int S = 64 / H;
    const unsigned grid = (unsigned)((a.B + S - 1) / S);
    switch (H) {
        case 8: hipLaunchKernelGGL(gru_small_kernel<8>, dim3(grid), dim3(64), 0, stream, a); break;
        case 16: hipLaunchKernelGGL(gru_small_kernel<16>, dim3(grid), dim3(64), 0, stream, a); break;
        case 32: hipLaunchKernelGGL(gru_small_kernel<32>, dim3(grid), dim3(64), 0, stream, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}   // namespace ntm
