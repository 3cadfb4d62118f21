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
//
// ANY hidden size (round 5; the reference's `--HIDDEN_SIZE` is a free integer, code/train.py:50, code/model.py:22,44-45):
//   H <= 64 that is not a power of two runs the same kernel at the next power of two HP (PAD = true): rows and columns
//   beyond H are zero, so the padded units stay exactly 0 (r = z = 1/2, n = tanh(0) = 0, h = z h = 0) and add exact zeros
//   to every sum; HP = 64 (H = 33 .. 63) is one stream per wavefront with 192 weight VGPRs per lane.  H = 64 itself has
//   the matrix-pipe / low-latency kernels.
//   64 < H <= 128: gru_wide_kernel<true> -- a 4-wave workgroup per stream, thread (unit, K half) keeps its 3 x 64 weights
//   in VGPRs, the two halves of a unit sit in adjacent lanes (one DPP add), h through LDS, ONE barrier per step.
//   128 < H <= 1024: gru_wide_kernel<false> -- the same workgroup with the weights streamed from L2 every step (rows go
//   round the waves, lanes stride over the columns: coalesced loads, a DPP sum per row): plain and correct, L2-bound,
//   nothing more (no shipped checkpoint or script is wider than 64).
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
    if constexpr (H >= 64) v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 -> rows 2,3
    return v;
}

// PAD = false: the hidden size IS H.  PAD = true: the hidden size is a.H < H, rows / columns [a.H, H) are zero padding.
template <int H, bool PAD = false>
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
    const int HR = PAD ? a.H : H;            // the model's hidden size: row stride of every parameter and of h_state
    const bool unit = !PAD || u < HR;        // this lane owns a real hidden unit
    f32x2 Wr[H / 2], Wz[H / 2], Wn[H / 2];
#pragma unroll
    for (int k = 0; k < H / 2; ++k) {
        if constexpr (!PAD) {
            const float *pr = a.w_hh + (size_t)(0 * H + u) * H + 2 * k;
            const float *pz = a.w_hh + (size_t)(1 * H + u) * H + 2 * k;
            const float *pn = a.w_hh + (size_t)(2 * H + u) * H + 2 * k;
            Wr[k] = (f32x2){pr[0] * SRZ, pr[1] * SRZ};
            Wz[k] = (f32x2){pz[0] * SRZ, pz[1] * SRZ};
            Wn[k] = (f32x2){pn[0] * SN, pn[1] * SN};
        } else {
            auto at = [&](int g, int c) { return (unit && c < HR) ? a.w_hh[(size_t)(g * HR + u) * HR + c] : 0.0f; };
            Wr[k] = (f32x2){at(0, 2 * k) * SRZ, at(0, 2 * k + 1) * SRZ};
            Wz[k] = (f32x2){at(1, 2 * k) * SRZ, at(1, 2 * k + 1) * SRZ};
            Wn[k] = (f32x2){at(2, 2 * k) * SN, at(2, 2 * k + 1) * SN};
        }
    }
    auto vec = [&](const float *p, int g) { return unit ? p[g * HR + u] : 0.0f; };
    const float wir = vec(a.w_ih, 0) * SRZ, wiz = vec(a.w_ih, 1) * SRZ, win = vec(a.w_ih, 2) * SN;
    const float br = (vec(a.b_ih, 0) + vec(a.b_hh, 0)) * SRZ, bz = (vec(a.b_ih, 1) + vec(a.b_hh, 1)) * SRZ;
    const float bin_ = vec(a.b_ih, 2) * SN, bhn = vec(a.b_hh, 2) * SN;
    const float wo = unit ? a.w_o[u] : 0.0f;
    const float bo = a.b_o ? a.b_o[0] : 0.0f;
    float hold = (a.h_state && valid && unit) ? a.h_state[(s0 + s) * HR + u] : 0.0f;
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
    if (a.h_state && valid && unit) a.h_state[(s0 + s) * HR + u] = hold;
}

// ---- 64 < H <= 1024: a workgroup of 4 waves per stream ------------------------------------------------------------
// REGW = true (H <= 128): thread tid = 2 u + ks owns the K half ks of unit u for the three gates: 3 x 64 weights in VGPRs
//   (zero beyond H), h broadcast-read from LDS as b128, the two halves meet by one DPP quad_perm add; lane ks = 0 does the
//   gates of its unit.  REGW = false: the weights come from L2 every step, a row per wave at a time (see the step).
// The head: wo . h summed per wave by DPP, the four wave partials parked in LDS by step parity and added by the thread
// that stores y one step later -- so a step has ONE barrier.  y leaves in 64-sample tiles.
template <bool REGW>
__global__ __launch_bounds__(256) void gru_wide_kernel(GruArgs a)
{
#pragma clang fp contract(off)
    constexpr int HMAX = REGW ? 128 : 1024;
    constexpr int TT = 64;
    __shared__ __attribute__((aligned(16))) float hs[2][HMAX];
    __shared__ float part[2][4];
    __shared__ float yt[TT];
    __shared__ float xt[2][TT];
    __shared__ float gh[REGW ? 1 : 3 * HMAX];      // REGW = false: W_hh . h of the step, all three gates
    const int tid = threadIdx.x, lane = tid & 63;
    // (scalar: the row loop of the REGW = false step carries DPP sums and must not be exec-masked, tools/check_dpp_exec.py)
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H;
    const int64_t b = blockIdx.x, T = a.T;
    const float bo = a.b_o ? a.b_o[0] : 0.0f;
    const float *xrow = a.x + b * a.xs;
    float *yrow = a.y + b * a.ys;

    // REGW: unit u = tid >> 1, K half ks = tid & 1 covering columns [64 ks, 64 ks + 64)
    const int u = tid >> 1, ks = tid & 1;
    float W[REGW ? 3 : 1][REGW ? 64 : 1];
    float wi[3] = {0, 0, 0}, bsum[2] = {0, 0}, bin_ = 0, bhn = 0, wo = 0, hold = 0;
    if constexpr (REGW) {
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                const int c = 64 * ks + k;
                W[g][k] = (u < H && c < H) ? a.w_hh[(size_t)(g * H + u) * H + c] : 0.0f;
            }
        if (u < H && ks == 0) {
            for (int g = 0; g < 3; ++g) wi[g] = a.w_ih[g * H + u];
            bsum[0] = a.b_ih[u] + a.b_hh[u];
            bsum[1] = a.b_ih[H + u] + a.b_hh[H + u];
            bin_ = a.b_ih[2 * H + u];
            bhn = a.b_hh[2 * H + u];
            wo = a.w_o[u];
            hold = a.h_state ? a.h_state[b * H + u] : 0.0f;
        }
    }
    for (int i = tid; i < HMAX; i += 256) {
        hs[0][i] = (i < H && a.h_state) ? a.h_state[b * H + i] : 0.0f;
        hs[1][i] = 0.0f;
    }
    if (tid < 4) { part[0][tid] = 0.0f; part[1][tid] = 0.0f; }
    if (tid < TT) xt[0][tid] = tid < T ? xrow[tid] : 0.0f;
    __syncthreads();

    auto wave_sum = [&](float v) {          // sum over the wave's 64 lanes, valid in lane 63
        v = dpp_add<0x111, 0xf>(v); v = dpp_add<0x112, 0xf>(v); v = dpp_add<0x114, 0xf>(v); v = dpp_add<0x118, 0xf>(v);
        v = dpp_add<0x142, 0xa>(v); v = dpp_add<0x143, 0xc>(v);
        return v;
    };
    for (int64_t t = 0; t < T; ++t) {
        const int cur = (int)(t & 1);
        const float x = xt[(t >> 6) & 1][t & (TT - 1)];
        float hp = 0.0f;                      // this thread's part of wo . h_t
        if constexpr (REGW) {
            float ar = 0.0f, az = 0.0f, an = 0.0f;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const f32x4 hv = *(const f32x4 *)&hs[cur][64 * ks + 4 * c];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ar = __builtin_fmaf(W[0][4 * c + e], hv[e], ar);
                    az = __builtin_fmaf(W[1][4 * c + e], hv[e], az);
                    an = __builtin_fmaf(W[2][4 * c + e], hv[e], an);
                }
            }
            // the other K half sits in the neighbouring lane: quad_perm [1,0,3,2]
            auto pair = [&](float v) {
                const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false);
                return v + __builtin_bit_cast(float, o);
            };
            ar = pair(ar); az = pair(az); an = pair(an);
            if (ks == 0 && u < H) {
                const float r = sigmoid_f32(__builtin_fmaf(wi[0], x, bsum[0]) + ar);
                const float z = sigmoid_f32(__builtin_fmaf(wi[1], x, bsum[1]) + az);
                const float n = tanh_f32(__builtin_fmaf(r, an + bhn, __builtin_fmaf(wi[2], x, bin_)));
                hold = __builtin_fmaf(z, hold - n, n);
                hs[cur ^ 1][u] = hold;
                hp = wo * hold;
            }
        } else {
            // rows of the stacked [3H, H] matrix go round the four waves; a wave's lanes stride over the COLUMNS of its row
            // (coalesced 256-byte loads from L2: the first version gave every thread its own rows -- 64 cache lines per load
            // instruction, 1.8 ms per step at H = 200 and 4096 streams) and meet in a DPP sum
            for (int r = wv; r < 3 * H; r += 4) {
                const float *row = a.w_hh + (size_t)r * H;
                float acc = 0.0f;
                for (int c = lane; c < H; c += 64) acc = __builtin_fmaf(row[c], hs[cur][c], acc);
                acc = wave_sum(acc);
                if (lane == 63) gh[r] = acc;
            }
            __syncthreads();
            for (int uu = tid; uu < H; uu += 256) {
                const float r = sigmoid_f32(__builtin_fmaf(a.w_ih[uu], x, a.b_ih[uu] + a.b_hh[uu]) + gh[uu]);
                const float z = sigmoid_f32(__builtin_fmaf(a.w_ih[H + uu], x, a.b_ih[H + uu] + a.b_hh[H + uu]) + gh[H + uu]);
                const float n = tanh_f32(__builtin_fmaf(r, gh[2 * H + uu] + a.b_hh[2 * H + uu], __builtin_fmaf(a.w_ih[2 * H + uu], x, a.b_ih[2 * H + uu])));
                const float hnew = __builtin_fmaf(z, hs[cur][uu] - n, n);
                hs[cur ^ 1][uu] = hnew;
                hp += a.w_o[uu] * hnew;
            }
        }
        hp = wave_sum(hp);
        if (lane == 63) part[cur][wv] = hp;
        // y of the PREVIOUS step: its four partials were parked before the barrier that ended that step
        if (tid == 0 && t > 0) yt[(t - 1) & (TT - 1)] = ((part[cur ^ 1][0] + part[cur ^ 1][1]) + (part[cur ^ 1][2] + part[cur ^ 1][3])) + bo;
        __syncthreads();
        if ((t & (TT - 1)) == 0) {
            // y tile [t - 64, t) is complete (its last sample was written by thread 0 BEFORE this step's barrier) and leaves
            // now: yt[i] is next written behind the barrier of step t + i, and thread 0 reads its own yt[0] first.  The x
            // tile after the current one comes in: first read 64 barriers from here.
            if (tid < TT) {
                if (t > 0) yrow[t - TT + tid] = yt[tid];
                xt[((t >> 6) + 1) & 1][tid] = (t + TT + tid < T) ? xrow[t + TT + tid] : 0.0f;
            }
        }
    }
    // the last step's y, then the ragged tail tile
    if (tid == 0 && T > 0) {
        const int cur = (int)((T - 1) & 1);
        yt[(T - 1) & (TT - 1)] = ((part[cur][0] + part[cur][1]) + (part[cur][2] + part[cur][3])) + bo;
    }
    __syncthreads();
    {
        const int64_t t0 = ((T - 1) / TT) * TT;
        if (T > 0 && t0 + tid < T && tid < TT) yrow[t0 + tid] = yt[tid];
    }
    if (a.h_state)
        for (int i = tid; i < H; i += 256) a.h_state[b * H + i] = hs[T & 1][i];
}

// General input_size / output_size (code/model.py:22,44-45: nn.GRU(input_size, H) + nn.Linear(H, output_size); no caller of the
// reference uses sizes other than 1, so this is protocol completeness, plain and correct like gru_wide_kernel<false>, nothing more).
// The reference reinterprets (B, C, T) as (B, T, C) with `reshape` (code/model.py:77,87): per stream the input row IS the [T][I]
// matrix the GRU reads and the output row IS the [T][O] matrix the head writes.  A 4-wave workgroup per stream; per step the
// rows of W_hh go round the waves (coalesced loads, a DPP sum per row), the gates read W_ih . x_t from L2 (I products per gate,
// summed in i = 0..I-1 order), the O head rows go round the waves again; three barriers per step.
__global__ __launch_bounds__(256) void gru_io_kernel(GruArgs a, int I, int O)
{
#pragma clang fp contract(off)
    constexpr int HMAX = 1024;
    __shared__ float hs[2][HMAX];
    __shared__ float gh[3 * HMAX];
    __shared__ float xs[HMAX];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H;
    const int64_t b = blockIdx.x, T = a.T;
    const float *xrow = a.x + b * a.xs;
    float *yrow = a.y + b * a.ys;
    for (int i = tid; i < HMAX; i += 256) {
        hs[0][i] = (i < H && a.h_state) ? a.h_state[b * H + i] : 0.0f;
        hs[1][i] = 0.0f;
    }
    auto wave_sum = [&](float v) {          // sum over the wave's 64 lanes, valid in lane 63
        v = dpp_add<0x111, 0xf>(v); v = dpp_add<0x112, 0xf>(v); v = dpp_add<0x114, 0xf>(v); v = dpp_add<0x118, 0xf>(v);
        v = dpp_add<0x142, 0xa>(v); v = dpp_add<0x143, 0xc>(v);
        return v;
    };
    __syncthreads();
    for (int64_t t = 0; t < T; ++t) {
        const int cur = (int)(t & 1);
        for (int i = tid; i < I; i += 256) xs[i] = xrow[t * I + i];
        for (int r = wv; r < 3 * H; r += 4) {
            const float *row = a.w_hh + (size_t)r * H;
            float acc = 0.0f;
            for (int c = lane; c < H; c += 64) acc = __builtin_fmaf(row[c], hs[cur][c], acc);
            acc = wave_sum(acc);
            if (lane == 63) gh[r] = acc;
        }
        __syncthreads();
        for (int u = tid; u < H; u += 256) {
            float gi[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const float *wr = a.w_ih + (size_t)(g * H + u) * I;
                float acc = 0.0f;
                for (int i = 0; i < I; ++i) acc = __builtin_fmaf(wr[i], xs[i], acc);
                gi[g] = acc + a.b_ih[g * H + u];
            }
            const float r = sigmoid_f32(gi[0] + (gh[u] + a.b_hh[u]));
            const float z = sigmoid_f32(gi[1] + (gh[H + u] + a.b_hh[H + u]));
            const float n = tanh_f32(__builtin_fmaf(r, gh[2 * H + u] + a.b_hh[2 * H + u], gi[2]));
            hs[cur ^ 1][u] = __builtin_fmaf(z, hs[cur][u] - n, n);
        }
        __syncthreads();
        for (int o = wv; o < O; o += 4) {
            const float *row = a.w_o + (size_t)o * H;
            float acc = 0.0f;
            for (int c = lane; c < H; c += 64) acc = __builtin_fmaf(row[c], hs[cur ^ 1][c], acc);
            acc = wave_sum(acc);
            if (lane == 63) yrow[t * O + o] = a.b_o ? acc + a.b_o[o] : acc;
        }
        __syncthreads();                    // xs, gh and hs[cur] are rewritten by the next step
    }
    if (a.h_state)
        for (int i = tid; i < H; i += 256) a.h_state[b * H + i] = hs[T & 1][i];
}

}   // namespace

hipError_t launch_gru_io(const GruArgs &a0, int H, int I, int O, hipStream_t stream)
{
    if (a0.B == 0 || a0.T == 0) return hipSuccess;
    if (H < 1 || H > 1024 || I < 1 || I > 1024 || O < 1 || O > 1024 || a0.B > 0x7fffffff) return hipErrorInvalidValue;
    GruArgs a = a0;
    a.H = H;
    hipLaunchKernelGGL(gru_io_kernel, dim3((unsigned)a.B), dim3(256), 0, stream, a, I, O);
    return hipGetLastError();
}

// Any hidden size but 64 (ntm_api.hip routes H = 64 to the matrix-pipe / low-latency kernels).
hipError_t launch_gru_small(const GruArgs &a0, int H, hipStream_t stream)
{
    if (a0.B == 0) return hipSuccess;
    if (H < 1 || H > 1024) return hipErrorInvalidValue;
    GruArgs a = a0;
    a.H = H;
    if (H > 64) {
        if (a.B > 0x7fffffff) return hipErrorInvalidValue;
        if (H <= 128) hipLaunchKernelGGL(gru_wide_kernel<true>, dim3((unsigned)a.B), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL(gru_wide_kernel<false>, dim3((unsigned)a.B), dim3(256), 0, stream, a);
        return hipGetLastError();
    }
    int HP = 8;
    while (HP < H) HP *= 2;
    const int S = 64 / HP;
    const unsigned grid = (unsigned)((a.B + S - 1) / S);
    const bool pad = HP != H;
    switch (HP) {
        case 8: if (pad) hipLaunchKernelGGL((gru_small_kernel<8, true>), dim3(grid), dim3(64), 0, stream, a);
                else hipLaunchKernelGGL((gru_small_kernel<8>), dim3(grid), dim3(64), 0, stream, a); break;
        case 16: if (pad) hipLaunchKernelGGL((gru_small_kernel<16, true>), dim3(grid), dim3(64), 0, stream, a);
                 else hipLaunchKernelGGL((gru_small_kernel<16>), dim3(grid), dim3(64), 0, stream, a); break;
        case 32: if (pad) hipLaunchKernelGGL((gru_small_kernel<32, true>), dim3(grid), dim3(64), 0, stream, a);
                 else hipLaunchKernelGGL((gru_small_kernel<32>), dim3(grid), dim3(64), 0, stream, a); break;
        case 64: hipLaunchKernelGGL((gru_small_kernel<64, true>), dim3(grid), dim3(64), 0, stream, a); break;   // H = 33 .. 63
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}   // namespace ntm
