// K1e: low-latency GRU kernel for SMALL batches (NTM_GRU_LAT; NTM_GRU_AUTO picks it when there are too few
// streams to fill the matrix-pipe kernel: the warm-start of predict() -- ONE stream, code/model.py:58-65 -- and
// evaluation batches like BASELINE configs[0], 16 x 8192, or the reference's real test sets: dozens of 10-second segments).
//
// The MFMA2 kernel spends 2012 cycles per step on 16 streams at once and needs 16 streams per workgroup to be
// efficient; with B streams only ceil(B/16) CUs work.  Here ONE workgroup (4 waves) advances ONE stream, so B
// streams occupy B workgroups (several per CU); see the comment above the kernel for how the step is cut.
// Exact fp32 like the other exact kernels (different summation order: K split in four).
#include "ntm_common.h"

#include <type_traits>

namespace ntm {

typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int LT = 256;                  // samples per x / y tile
constexpr float LOG2E = 1.44269504088896340736f;

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

// Rounds 1-4 cut the step four ways along K BETWEEN the waves (wave w, lane u = unit u: columns 16w .. 16w+15 of all 192 rows):
// that needs TWO LDS round trips per step -- the partial sums out / in around the barrier, then each wave's private copy of h
// out / in (every wave evaluated all 64 gates redundantly) -- 342 ns per step.  Round 5: wave w owns units 16w .. 16w+15
// outright: lane l = 4 ul + kq holds the K quarter kq of unit 16w + ul for the three gates (48 weights, as before), the four quarters of a unit meet by two DPP quad_perm adds (every
// lane of the quad gets the same bits), the quad evaluates the gates redundantly and lane kq = 0 publishes h_t -- ONE LDS
// round trip per step: write h_t -> barrier -> four broadcast ds_read_b128 of the K quarter (h double-buffered by step parity,
// so one barrier orders both the reads of h_{t-1} and the writes of h_t).  The head: the wave on duty (t mod 4) reads all 64
// values of h_{t-1} from the same buffer (lane = unit) and sums w_o . h by DPP in the shadow of the K-quarter reads, one sample
// behind the recurrence -- on a fifth wave of its own when the workgroup has a CU to itself.  242-246 ns per step for B <= 256
// (342-344 before), 352 at B = 512 (426), 578 at B = 1024 (692).
// x and y move in 256-sample tiles through LDS (coalesced global accesses); the tile housekeeping sits between runs of steps.
template <int PERM>
__device__ __forceinline__ float quad_add(float v)
{
    const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), PERM, 0xf, 0xf, true);
    return v + __builtin_bit_cast(float, o);
}

// HEADW: a fifth wave does the head (B <= the number of CUs: a workgroup has a CU to itself); without it the head is a duty
// that rotates among the four compute waves (more streams than CUs: a fifth wave per workgroup costs occupancy)
template <bool HEADW>
__global__ __launch_bounds__(HEADW ? 320 : 256) void gru_lat_kernel(GruArgs a)
{
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) float hb[2][kH];            // h by step parity
    __shared__ float xt[2][LT];
    __shared__ float yt[2][LT];

    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ul = l >> 2, kq = l & 3;
    const bool head_wave = HEADW && w == 4;                             // the fifth wave: head + nothing else (see the step)
    const int u = 16 * (w & 3) + ul;                                    // this quad's hidden unit (the head wave: unused)
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

    if (kq == 0 && !head_wave) hb[0][u] = hold;
    if (tid < LT && tid < T) xt[0][tid] = xs[tid];
    float xnext = (tid < LT && LT + tid < T) ? xs[LT + tid] : 0.0f;
    __syncthreads();

    // step t = tile + ph: hb[t & 1] holds h_{t-1}; tb = tile parity (of the x / y buffers)
    const float *const hq_rd = &hb[0][16 * kq];                         // this lane's K quarter / unit in buffer 0
    float *const hu_wr = &hb[0][u];                                     // (buffer 1: a compile-time + kH in the unrolled loop)
    auto step = [&](const int ph, const int tb, auto par_c) {
        constexpr int par = decltype(par_c)::value;                     // == t & 1 (tiles are 256 steps): compile time
        auto head = [&]() {              // y of sample t-1: all 64 values of h_{t-1} from the exchange buffer (lane = unit), DPP sum
            const float yv = wave_sum_lane63(wo_l * hb[par][l]) + bo;
            if (l == 63) { if (ph > 0) yt[tb][ph - 1] = yv; else yt[tb ^ 1][LT - 1] = yv; }
        };
        if constexpr (HEADW) {
            if (head_wave) {             // on a wave of its own, off the compute waves' critical path: 246 ns per step instead of 287
                head();
                __syncthreads();
                return;
            }
        }
        const f32x4 h0 = *(const f32x4 *)(hq_rd + par * kH + 0), h1 = *(const f32x4 *)(hq_rd + par * kH + 4);
        const f32x4 h2 = *(const f32x4 *)(hq_rd + par * kH + 8), h3 = *(const f32x4 *)(hq_rd + par * kH + 12);
        const float x = xt[tb][ph];
        if constexpr (!HEADW) {          // as a duty that rotates among the compute waves, in the shadow of the reads above
            if ((ph & 3) == w) head();
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
        hu_wr[(par ^ 1) * kH] = hold;                      // all four lanes of the quad store the same bits to the same word
        __syncthreads();                                   // the step's only barrier
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    // steps [p0, p1) of the tile, two at a time with the parity known at compile time
    auto run = [&](int p0, const int p1, const int tb) {
        if (p0 < p1 && (p0 & 1)) { step(p0, tb, P1{}); ++p0; }
        for (; p0 + 1 < p1; p0 += 2) { step(p0, tb, P0{}); step(p0 + 1, tb, P1{}); }
        if (p0 < p1) step(p0, tb, P0{});
    };

    for (int64_t tile0 = 0; tile0 < T; tile0 += LT) {
        const int ns = (int)((T - tile0) < LT ? (T - tile0) : LT);
        const int tb = (int)((tile0 >> 8) & 1);
        run(0, ns < 3 ? ns : 3, tb);
        if (ns > 2 && tile0 >= LT && tid < LT) ys[tile0 - LT + tid] = yt[tb ^ 1][tid];      // previous y tile is complete
        run(3, ns < 129 ? ns : 129, tb);
        if (ns > 128 && tid < LT) {
            xt[tb ^ 1][tid] = xnext;
            const int64_t nx = tile0 + 2 * LT + tid;
            xnext = nx < T ? xs[nx] : 0.0f;
        }
        run(129, ns, tb);
    }
    if (T > 0 && (HEADW ? head_wave : w == 0)) {            // head of the last sample (h_{T-1} sits in hb[T & 1])
        const float yv = wave_sum_lane63(wo_l * hb[(int)(T & 1)][l]) + bo;
        if (l == 63) yt[(int)(((T - 1) >> 8) & 1)][(int)((T - 1) & (LT - 1))] = yv;
    }
    __syncthreads();
    const int64_t last0 = ((T - 1) >> 8) * LT;
    if (T > 0 && tid < LT) {
        if (last0 + tid < T) ys[last0 + tid] = yt[(last0 >> 8) & 1][tid];
        if (last0 >= LT && (T - 1 - last0) < 2) ys[last0 - LT + tid] = yt[((last0 >> 8) & 1) ^ 1][tid];
    }
    if (a.h_state && kq == 0 && !head_wave) a.h_state[s * kH + u] = hold;
}

}   // namespace

hipError_t launch_gru_lat(const GruArgs &a, hipStream_t stream)
{
    if (a.B == 0) return hipSuccess;
    if (a.B <= device_cus()) hipLaunchKernelGGL(gru_lat_kernel<true>, dim3((unsigned)a.B), dim3(320), 0, stream, a);
    else hipLaunchKernelGGL(gru_lat_kernel<false>, dim3((unsigned)a.B), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}   // namespace ntm
