// K1e: low-latency GRU kernel for SMALL batches (NTM_GRU_LAT; NTM_GRU_AUTO picks it when there are too few
// streams to fill the matrix-pipe kernel: the warm-start of predict() -- ONE stream, code/model.py:58-65 -- and
// evaluation batches like BASELINE configs[0], 16 x 8192).
//
// The MFMA2 kernel spends 2151 cycles per step on 16 streams at once and needs 16 streams per workgroup to be
// efficient; with B streams only ceil(B/16) CUs work.  Here ONE workgroup (4 waves) advances ONE stream, so B
// streams occupy B workgroups (several per CU), and the step is cut four ways along K:
//   wave w, lane u:  partial_g = sum_{k in [16w,16w+16)} W_g[u][k] h[k]   (g = r,z,n; weights resident in 48 VGPRs,
//                    the 16 h values arrive as 4 uniform-address ds_read_b128 from the wave's private h copy)
//   exchange of the 3 x 4 partial sums through LDS, ONE s_barrier per step, fixed summation order
//   every wave then evaluates the gates of all 64 units redundantly (lane u = unit u) and refreshes its private
//   h copy -- no second barrier; the head y_t = w_o . h_t + b_o is a DPP wave reduction done by wave (t mod 4)
//   one step later, behind that step's barrier, in the shadow of the partial-sum reads.
// x and y move in 256-sample tiles through LDS (coalesced global accesses).
// Exact fp32 like the other exact kernels (different summation order: K split in four).
#include "ntm_common.h"

#include <cstdlib>
#include <type_traits>

namespace ntm {

typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int LT = 256;                  // samples per x / y tile
constexpr float LOG2E = 1.44269504088896340736f;

__device__ __forceinline__ void lds_fence_wave()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_shift_add(float v)
{
    const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
    return v + __builtin_bit_cast(float, moved);
}

// Sum over the 64 lanes on the VALU (row_shr / row_bcast scan, no LDS round trips); total valid in lane 63.
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

__global__ __launch_bounds__(256) void gru_lat_kernel(GruArgs a)
{
    // no implicit contraction: hipcc peels the first step and would fuse the weight pre-scaling into ITS adds
    // (fma(2 log2e, b_hn, sum) instead of the rounded product), so a launch that starts at sample n would differ
    // in the last bit from one that passes through n -- chunked and one-shot predict() must agree exactly
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) float hc[4][kH];            // per-wave private copy of h
    __shared__ __attribute__((aligned(16))) float part[2][3][kH][4];    // [step parity][gate][unit][wave]
    __shared__ float xt[2][LT];
    __shared__ float yt[2][LT];

    const int tid = threadIdx.x, u = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t s = blockIdx.x;
    const int64_t T = a.T;
    const float *xs = a.x + s * a.xs;
    float *ys = a.y + s * a.ys;

    // resident weights: rows u of the three gates, columns 16w .. 16w+15, with -log2e / 2 log2e folded in so
    // that sigmoid / tanh start at v_exp_f32
    constexpr float SRZ = -LOG2E, SN = 2.0f * LOG2E;
    f32x2 Wr[8], Wz[8], Wn[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float *pr = a.w_hh + (size_t)(0 * kH + u) * kH + 16 * w + 2 * k;
        const float *pz = a.w_hh + (size_t)(1 * kH + u) * kH + 16 * w + 2 * k;
        const float *pn = a.w_hh + (size_t)(2 * kH + u) * kH + 16 * w + 2 * k;
        Wr[k] = (f32x2){pr[0] * SRZ, pr[1] * SRZ};
        Wz[k] = (f32x2){pz[0] * SRZ, pz[1] * SRZ};
        Wn[k] = (f32x2){pn[0] * SN, pn[1] * SN};
    }
    const float wir = a.w_ih[u] * SRZ, wiz = a.w_ih[kH + u] * SRZ, win = a.w_ih[2 * kH + u] * SN;
    const float br = (a.b_ih[u] + a.b_hh[u]) * SRZ, bz = (a.b_ih[kH + u] + a.b_hh[kH + u]) * SRZ;
    const float bin_ = a.b_ih[2 * kH + u] * SN, bhn = a.b_hh[2 * kH + u] * SN;
    const float bo = a.b_o ? a.b_o[0] : 0.0f;
    float hold = a.h_state ? a.h_state[s * kH + u] : 0.0f;

    const float wo = a.w_o[u];
    hc[w][u] = hold;
    if (tid < T) xt[0][tid] = xs[tid];
    float xnext = (LT + tid < T) ? xs[LT + tid] : 0.0f;      // tile 1, parked in a register until mid-tile
    __syncthreads();

    // one step; ph = t mod 256, tb = tile parity.  At step t the K quarter in hand is h_{t-1}, so the head partial
    // computed here belongs to sample t-1: slot ph-1 of this tile, or slot 255 of the previous tile's buffer
    // (at t = 0 that is a scratch write: buffer 1 is filled by samples 256.. long before it is flushed).
    auto step = [&](const int ph, const int tb) {
        const int par = ph & 1;
        // ---- this wave's quarter of h (uniform addresses: LDS broadcast) and the input sample ----
        // (v_readlane from the wave's own lanes 16w..16w+15 was tried instead of the private LDS copy: 16 VALU
        //  slots per step cost more than the 4 broadcast reads, 423 vs 405 ns at B = 1 and worse with more streams)
        const f32x4 h0 = *(const f32x4 *)&hc[w][16 * w + 0], h1 = *(const f32x4 *)&hc[w][16 * w + 4];
        const f32x4 h2 = *(const f32x4 *)&hc[w][16 * w + 8], h3 = *(const f32x4 *)&hc[w][16 * w + 12];
        const float x = xt[tb][ph];
        const f32x2 hq[8] = {{h0[0], h0[1]}, {h0[2], h0[3]}, {h1[0], h1[1]}, {h1[2], h1[3]},
                             {h2[0], h2[1]}, {h2[2], h2[3]}, {h3[0], h3[1]}, {h3[2], h3[3]}};
        // ---- partial dot products (two packed accumulators per row keep the chains short) ----
        f32x2 ar0 = Wr[0] * hq[0], ar1 = Wr[1] * hq[1], az0 = Wz[0] * hq[0], az1 = Wz[1] * hq[1];
        f32x2 an0 = Wn[0] * hq[0], an1 = Wn[1] * hq[1];
#pragma unroll
        for (int k = 2; k < 8; k += 2) {
            ar0 = __builtin_elementwise_fma(Wr[k], hq[k], ar0); ar1 = __builtin_elementwise_fma(Wr[k + 1], hq[k + 1], ar1);
            az0 = __builtin_elementwise_fma(Wz[k], hq[k], az0); az1 = __builtin_elementwise_fma(Wz[k + 1], hq[k + 1], az1);
            an0 = __builtin_elementwise_fma(Wn[k], hq[k], an0); an1 = __builtin_elementwise_fma(Wn[k + 1], hq[k + 1], an1);
        }
        const f32x2 sr = ar0 + ar1, sz = az0 + az1, sn = an0 + an1;
        part[par][0][u][w] = sr[0] + sr[1];
        part[par][1][u][w] = sz[0] + sz[1];
        part[par][2][u][w] = sn[0] + sn[1];
        // input terms while the exchange is in flight
        const float cr = __builtin_fmaf(wir, x, br), cz = __builtin_fmaf(wiz, x, bz), gi = __builtin_fmaf(win, x, bin_);
        __syncthreads();                                   // the step's only barrier
        const f32x4 qr = *(const f32x4 *)&part[par][0][u][0];
        const f32x4 qz = *(const f32x4 *)&part[par][1][u][0];
        const f32x4 qn = *(const f32x4 *)&part[par][2][u][0];
        // head of the PREVIOUS sample (`hold` still is h_{t-1}) on one wave per step, as a DPP wave reduction issued
        // right here: its dependent chain runs in the shadow of the three LDS reads above.  (Issued BEFORE the
        // barrier it delayed the barrier for every wave: 405 instead of 362 ns per step.)
        if ((ph & 3) == w) {
            const float yv = wave_sum_lane63(wo * hold) + bo;
            if (u == 63) { if (ph > 0) yt[tb][ph - 1] = yv; else yt[tb ^ 1][LT - 1] = yv; }
        }
        const float pr_ = cr + ((qr[0] + qr[1]) + (qr[2] + qr[3]));
        const float pz_ = cz + ((qz[0] + qz[1]) + (qz[2] + qz[3]));
        const float gh = bhn + ((qn[0] + qn[1]) + (qn[2] + qn[3]));
        // ---- gates (pre-scaled arguments): r, z = 1/(1 + 2^p);  n = 1 - 2/(1 + 2^q) ----
        const float r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pr_));
        const float z = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pz_));
        const float en = __builtin_amdgcn_exp2f(__builtin_fmaf(r, gh, gi));
        const float n = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + en), 1.0f);
        hold = __builtin_fmaf(z, hold - n, n);
        hc[w][u] = hold;                                   // private copy: read back by this wave only
        lds_fence_wave();
    };

    // tiles of 256 steps; the tile housekeeping sits between runs of steps, not inside the step
    for (int64_t tile0 = 0; tile0 < T; tile0 += LT) {
        const int ns = (int)((T - tile0) < LT ? (T - tile0) : LT);
        const int tb = (int)((tile0 >> 8) & 1);
        int ph = 0;
        for (; ph < (ns < 3 ? ns : 3); ++ph) step(ph, tb);
        if (ns > 2 && tile0 >= LT) ys[tile0 - LT + tid] = yt[tb ^ 1][tid];      // previous y tile is complete
        for (; ph < (ns < 129 ? ns : 129); ++ph) step(ph, tb);
        if (ns > 128) {                                     // park the next x tile, fetch the one after
            xt[tb ^ 1][tid] = xnext;
            const int64_t nx = tile0 + 2 * LT + tid;
            xnext = nx < T ? xs[nx] : 0.0f;
        }
        for (; ph < ns; ++ph) step(ph, tb);
    }
    if (T > 0 && w == 0) {                                  // head of the last sample
        const float yv = wave_sum_lane63(wo * hold) + bo;
        if (u == 63) yt[(int)(((T - 1) >> 8) & 1)][(int)((T - 1) & (LT - 1))] = yv;
    }
    __syncthreads();
    // last (partial) y tile(s)
    const int64_t last0 = ((T - 1) >> 8) * LT;
    if (T > 0) {
        if (last0 + tid < T) ys[last0 + tid] = yt[(last0 >> 8) & 1][tid];
        // the tile before the last one is flushed at ph == 2 of the last tile only if the last tile got that far
        if (last0 >= LT && (T - 1 - last0) < 2) ys[last0 - LT + tid] = yt[((last0 >> 8) & 1) ^ 1][tid];
    }
    if (a.h_state && w == 0) a.h_state[s * kH + u] = hold;
}


// ---- round 5: the same kernel with the step cut along UNITS between the waves and along K inside a quad of lanes ----------
// gru_lat_kernel above needs TWO LDS round trips per step (partial sums out / in around the barrier, then the wave's private
// copy of h out / in).  Here wave w owns units 16w .. 16w+15 outright: lane l = 4 ul + kq holds the K quarter kq of unit
// 16w + ul for the three gates (48 weights, as before), the four quarters of a unit meet by two DPP quad_perm adds (every
// lane of the quad gets the same bits), the quad evaluates the gates redundantly and lane kq = 0 publishes h_t -- ONE LDS
// round trip per step: write h_t -> barrier -> four broadcast ds_read_b128 of the K quarter (h double-buffered by step parity,
// so one barrier orders both the reads of h_{t-1} and the writes of h_t).  The head: the wave on duty (t mod 4) reads all 64
// values of h_{t-1} from the same buffer (lane = unit) and sums w_o . h by DPP in the shadow of the K-quarter reads, one sample
// behind the recurrence, as gru_lat_kernel does.  Same arithmetic per unit except for the summation tree of the dot product.
template <int PERM>
__device__ __forceinline__ float quad_add(float v)
{
    const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), PERM, 0xf, 0xf, true);
    return v + __builtin_bit_cast(float, o);
}

__global__ __launch_bounds__(256) void gru_lat2_kernel(GruArgs a)
{
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) float hb[2][kH];            // h by step parity
    __shared__ float xt[2][LT];
    __shared__ float yt[2][LT];

    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ul = l >> 2, kq = l & 3;
    const int u = 16 * w + ul;                                          // this quad's hidden unit
    const int64_t s = blockIdx.x;
    const int64_t T = a.T;
    const float *xs = a.x + s * a.xs;
    float *ys = a.y + s * a.ys;

    constexpr float SRZ = -LOG2E, SN = 2.0f * LOG2E;
    f32x2 Wr[8], Wz[8], Wn[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float *pr = a.w_hh + (size_t)(0 * kH + u) * kH + 16 * kq + 2 * k;
        const float *pz = a.w_hh + (size_t)(1 * kH + u) * kH + 16 * kq + 2 * k;
        const float *pn = a.w_hh + (size_t)(2 * kH + u) * kH + 16 * kq + 2 * k;
        Wr[k] = (f32x2){pr[0] * SRZ, pr[1] * SRZ};
        Wz[k] = (f32x2){pz[0] * SRZ, pz[1] * SRZ};
        Wn[k] = (f32x2){pn[0] * SN, pn[1] * SN};
    }
    const float wir = a.w_ih[u] * SRZ, wiz = a.w_ih[kH + u] * SRZ, win = a.w_ih[2 * kH + u] * SN;
    const float br = (a.b_ih[u] + a.b_hh[u]) * SRZ, bz = (a.b_ih[kH + u] + a.b_hh[kH + u]) * SRZ;
    const float bin_ = a.b_ih[2 * kH + u] * SN, bhn = a.b_hh[2 * kH + u] * SN;
    const float bo = a.b_o ? a.b_o[0] : 0.0f;
    const float wo_l = a.w_o[l];                                        // head weights by LANE: the wave on duty sums all 64 units
    float hold = a.h_state ? a.h_state[s * kH + u] : 0.0f;

    if (kq == 0) hb[0][u] = hold;
    if (tid < T) xt[0][tid] = xs[tid];
    float xnext = (LT + tid < T) ? xs[LT + tid] : 0.0f;
    __syncthreads();

    // step t = tile + ph: hb[t & 1] holds h_{t-1}; tb = tile parity (of the x / y buffers)
    auto step = [&](const int ph, const int tb) {
        const int par = ph & 1;                                         // == t & 1 (tiles are 256 steps)
        const f32x4 h0 = *(const f32x4 *)&hb[par][16 * kq + 0], h1 = *(const f32x4 *)&hb[par][16 * kq + 4];
        const f32x4 h2 = *(const f32x4 *)&hb[par][16 * kq + 8], h3 = *(const f32x4 *)&hb[par][16 * kq + 12];
        const float x = xt[tb][ph];
        // the head of sample t-1 on ONE wave per step (wave-uniform branch), from the same buffer: a fifth read, a DPP wave
        // sum in the shadow of the reads above
        if ((ph & 3) == w) {
            const float yv = wave_sum_lane63(wo_l * hb[par][l]) + bo;
            if (l == 63) { if (ph > 0) yt[tb][ph - 1] = yv; else yt[tb ^ 1][LT - 1] = yv; }
        }
        const f32x2 hq[8] = {{h0[0], h0[1]}, {h0[2], h0[3]}, {h1[0], h1[1]}, {h1[2], h1[3]},
                             {h2[0], h2[1]}, {h2[2], h2[3]}, {h3[0], h3[1]}, {h3[2], h3[3]}};
        f32x2 ar0 = Wr[0] * hq[0], ar1 = Wr[1] * hq[1], az0 = Wz[0] * hq[0], az1 = Wz[1] * hq[1];
        f32x2 an0 = Wn[0] * hq[0], an1 = Wn[1] * hq[1];
#pragma unroll
        for (int k = 2; k < 8; k += 2) {
            ar0 = __builtin_elementwise_fma(Wr[k], hq[k], ar0); ar1 = __builtin_elementwise_fma(Wr[k + 1], hq[k + 1], ar1);
            az0 = __builtin_elementwise_fma(Wz[k], hq[k], az0); az1 = __builtin_elementwise_fma(Wz[k + 1], hq[k + 1], az1);
            an0 = __builtin_elementwise_fma(Wn[k], hq[k], an0); an1 = __builtin_elementwise_fma(Wn[k + 1], hq[k + 1], an1);
        }
        const f32x2 sr = ar0 + ar1, sz = az0 + az1, sn = an0 + an1;
        // the unit's four K quarters: quad_perm [1,0,3,2] then [2,3,0,1] -- ((q0 + q1) + (q2 + q3)) in every lane of the quad
        const float qr = quad_add<0x4E>(quad_add<0xB1>(sr[0] + sr[1]));
        const float qz = quad_add<0x4E>(quad_add<0xB1>(sz[0] + sz[1]));
        const float qn = quad_add<0x4E>(quad_add<0xB1>(sn[0] + sn[1]));
        const float cr = __builtin_fmaf(wir, x, br), cz = __builtin_fmaf(wiz, x, bz), gi = __builtin_fmaf(win, x, bin_);
        const float pr_ = cr + qr, pz_ = cz + qz, gh = bhn + qn;
        const float r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pr_));
        const float z = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pz_));
        const float en = __builtin_amdgcn_exp2f(__builtin_fmaf(r, gh, gi));
        const float n = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + en), 1.0f);
        hold = __builtin_fmaf(z, hold - n, n);
        if (kq == 0) hb[par ^ 1][u] = hold;
        __syncthreads();                                   // the step's only barrier
    };

    for (int64_t tile0 = 0; tile0 < T; tile0 += LT) {
        const int ns = (int)((T - tile0) < LT ? (T - tile0) : LT);
        const int tb = (int)((tile0 >> 8) & 1);
        int ph = 0;
        for (; ph < (ns < 3 ? ns : 3); ++ph) step(ph, tb);
        if (ns > 2 && tile0 >= LT) ys[tile0 - LT + tid] = yt[tb ^ 1][tid];      // previous y tile is complete
        for (; ph < (ns < 129 ? ns : 129); ++ph) step(ph, tb);
        if (ns > 128) {
            xt[tb ^ 1][tid] = xnext;
            const int64_t nx = tile0 + 2 * LT + tid;
            xnext = nx < T ? xs[nx] : 0.0f;
        }
        for (; ph < ns; ++ph) step(ph, tb);
    }
    if (T > 0 && w == 0) {                                  // head of the last sample (h_{T-1} sits in hb[T & 1])
        const float yv = wave_sum_lane63(wo_l * hb[(int)(T & 1)][l]) + bo;
        if (l == 63) yt[(int)(((T - 1) >> 8) & 1)][(int)((T - 1) & (LT - 1))] = yv;
    }
    __syncthreads();
    const int64_t last0 = ((T - 1) >> 8) * LT;
    if (T > 0) {
        if (last0 + tid < T) ys[last0 + tid] = yt[(last0 >> 8) & 1][tid];
        if (last0 >= LT && (T - 1 - last0) < 2) ys[last0 - LT + tid] = yt[((last0 >> 8) & 1) ^ 1][tid];
    }
    if (a.h_state && kq == 0) a.h_state[s * kH + u] = hold;
}

}   // namespace

hipError_t launch_gru_lat(const GruArgs &a, hipStream_t stream)
{
    if (a.B == 0) return hipSuccess;
    static const bool old_kernel = getenv("NTM_LAT_OLD") != nullptr;      // development A/B only
    if (old_kernel) hipLaunchKernelGGL(gru_lat_kernel, dim3((unsigned)a.B), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(gru_lat2_kernel, dim3((unsigned)a.B), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}   // namespace ntm
