// GRU-HS[64] + head, MFMA variant 2 ("own quarter first, blocked issue"): the default K1 kernel.
//
// Same mapping as gru_mfma_kernel (16 streams per 4-wave workgroup, wave w owns hidden units
// [16w,16w+16) of all three gates, W_hh resident in VGPRs as v_mfma_f32_16x16x4_f32 A-operands, exact
// fp32), re-ordered around two measured facts (tools/ubench/*.hip, MI355X):
//   (1) a VALU op issued between two f32 MFMAs of the SAME wave costs ~8 cycles of MFMA issue, a VALU op
//       in a contiguous VALU block ~5: the f32 matrix pipe and the VALU do not overlap within a wave, so
//       all 48 MFMAs of a step are issued back to back and the gate math follows as one block;
//   (2) an LDS round trip (write -> barrier -> read) is ~200-300 cycles of pure latency.
// Hence:
//   * K permutation unit(s,k) = 16(s>>2) + 4k + (s&3): K-step s consumes only units of the quarter of h
//     that wave s>>2 produced, and for a wave's four OWN K-steps (s = 4w+v) the B operand of lane
//     (k,j) is unit 16w+4k+v of stream j -- exactly register v of the C/D fragment the same lane just
//     produced (C/D layout: lane (q,j) = stream j, units 16w+4q+v).  So the new h feeds the next
//     step's first 12 MFMAs with no data movement at all, and leaves for the other three waves as ONE
//     ds_write_b128 of those same registers.  (transpose_groups4 below is kept for the unit test of
//     the permlane-swap idiom; the kernel no longer needs it.)
//   * step t starts with 12 MFMAs (r,n,z x 4 own K-steps) straight out of registers; behind the third
//     one sits the step's only barrier (all h_{t-1} writes are complete), then three ds_read_b128
//     fetch the other three quarters while the remaining own MFMAs run; the other 36 MFMAs follow.
//   * K-steps are numbered per wave (sigma = (s - 4w) mod 16) so that "own first" uses fixed registers.
//   * the VALU block after the MFMAs: input terms of step t+1, the three gates on packed fp32 ops
//     (v_pk_fma/add) and v_exp/v_rcp, the blend, the head partial of y_t over the lane's 4 units (the
//     16 partials per sample are summed when a 64-sample tile is flushed), the transpose.
//   * PRESCALE (default on): -log2(e) resp. 2 log2(e) are folded into W_hh / W_ih / biases when they
//     are loaded, so sigmoid and tanh start directly with v_exp_f32 (saves 12 VALU ops per step).
#include "ntm_common.h"
#include "delay_math.h"

#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace ntm {

#ifndef NTM2_NB
#define NTM2_NB 1
#endif
#ifndef NTM2_HKTRIM
#define NTM2_HKTRIM 1     // whole-tile forms of the x-tile load and the y-tile flush in the unrolled loop (0: the general forms everywhere)
#endif
#ifndef NTM2_ORDER
#define NTM2_ORDER 0      // order of the three accumulator chains inside a K-step: 0 = r, n, z   1 = r, z, n
#endif

namespace m2 {
constexpr int SG = 16;            // streams per workgroup
constexpr int TT = 64;            // samples per x / y staging tile
constexpr int HB_J = 20;          // floats per (k, stream) row: 16 K-steps + 4 pad (conflict-free b128)
constexpr int HB_K = SG * HB_J;   // 320
constexpr int HB = 4 * HB_K;      // 1280 floats per buffer
constexpr int XS = TT + 1;        // x tile row (conflict-free column reads)
constexpr int YS = TT + 4;        // y partial row: 16-B aligned rows for ds_read_b128 at flush
constexpr int YP_Q = SG * YS;     // 1088 floats per partial plane
constexpr int HB3 = 2 * 3 * 64 * 4; // ENGINE 2 exchange buffer: [K half][bf16 piece][lane] x 16 B = 1536 floats
constexpr int hb_floats(int engine) { return engine == 2 ? HB3 : HB; }
constexpr int smem_floats(int ypn, int engine = 0) { return 2 * hb_floats(engine) + 2 * SG * XS + 2 * ypn * YP_Q; }   // ypn partial planes per y tile
static_assert((2 * HB + 2 * SG * XS) % 4 == 0 && (2 * HB3 + 2 * SG * XS) % 4 == 0 && YS % 4 == 0 && YP_Q % 4 == 0,
              "y partial rows must be 16-B aligned");
}  // namespace m2

__device__ __forceinline__ f32x4 mfma16x(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// 4x4 transpose between the wave's four 16-lane groups and four registers:
// out[i] at lane group k  =  in[k] at lane group i   (same lane-in-group).
__device__ __forceinline__ void transpose_groups4(float (&r)[4])
{
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    unsigned a0 = __builtin_bit_cast(unsigned, r[0]), a1 = __builtin_bit_cast(unsigned, r[1]);
    unsigned a2 = __builtin_bit_cast(unsigned, r[2]), a3 = __builtin_bit_cast(unsigned, r[3]);
    // v_permlane32_swap X, Y : X[32..63] <-> Y[0..31]
    u32x2 p = __builtin_amdgcn_permlane32_swap(a0, a2, false, false);
    a0 = p[0]; a2 = p[1];
    p = __builtin_amdgcn_permlane32_swap(a1, a3, false, false);
    a1 = p[0]; a3 = p[1];
    // v_permlane16_swap X, Y : X rows 1,3 <-> Y rows 0,2
    p = __builtin_amdgcn_permlane16_swap(a0, a1, false, false);
    a0 = p[0]; a1 = p[1];
    p = __builtin_amdgcn_permlane16_swap(a2, a3, false, false);
    a2 = p[0]; a3 = p[1];
    r[0] = __builtin_bit_cast(float, a0); r[1] = __builtin_bit_cast(float, a1);
    r[2] = __builtin_bit_cast(float, a2); r[3] = __builtin_bit_cast(float, a3);
}

#ifdef NTM_LAB
__global__ __launch_bounds__(256) void debug_transpose_kernel(const float *in, float *out)
{
    float r[4];
    for (int v = 0; v < 4; ++v) r[v] = in[threadIdx.x * 4 + v];
    transpose_groups4(r);
    for (int v = 0; v < 4; ++v) out[threadIdx.x * 4 + v] = r[v];
}
#endif

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// fp32 pair -> three bf16 pairs, round to nearest each time: v = p0 + p1 + p2 EXACTLY (8 + 8 + 8 significant bits, the
// residuals v - p0 and v - p0 - p1 are exact in fp32; bf16 has fp32's exponent range).  v_cvt_pk_bf16_f32 packs a pair
// into one dword; 9 vector instructions per pair.
__device__ __forceinline__ void split_bf16x3(const f32x2 v, bf16x2 &p0, bf16x2 &p1, bf16x2 &p2)
{
    p0 = __builtin_convertvector(v, bf16x2);
    const f32x2 r1 = v - __builtin_convertvector(p0, f32x2);
    p1 = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(p1, f32x2);
    p2 = __builtin_convertvector(r2, bf16x2);
}

// fp32 x4 -> fp16 hi + fp16 lo (round to nearest both times): v = hi + lo to ~22 bits
__device__ __forceinline__ void split_f16(const f32x4 v, f16x4 &hi, f16x4 &lo)
{
    hi = __builtin_convertvector(v, f16x4);
    lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), f16x4);
}
__device__ __forceinline__ f32x4 pack_hl(const f16x4 hi, const f16x4 lo)
{
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    const f16x8 v = __builtin_shufflevector(hi, lo, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(f32x4, v);
}

// STAMP = true is a DIAGNOSTIC build (ntm_debug_gru_stamps): s_memtime is issued (not waited for) at six
// points of the step; the differences are accumulated once per step after a single wait.
#define NTM2_STAMP(k)                                                                       \
    if constexpr (STAMP) {                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        asm volatile("s_memtime %0" : "=s"(ts_[k])::"memory");                              \
        __builtin_amdgcn_sched_barrier(0);                                                  \
    }

// ABL != 0 are DIAGNOSTIC instantiations (wrong results on purpose, never timed as product):
//   1 no gate math   2 no LDS exchange of h (barrier kept)   4 no head partial   8 no MFMAs on the
//   other quarters   16 no barrier   32 no tile housekeeping
// ENGINE 0: exact fp32 (v_mfma_f32_16x16x4_f32).
// ENGINE 1: "f16x3" -- W and h are split into fp16 hi + lo parts (W = Wh + Wl, h = hh + hl, 22 significant
//   bits each) and W.h is evaluated as Wh.hh + Wh.hl + Wl.hh on v_mfma_f32_16x16x16_f16 with fp32
//   accumulation (products of fp16 pairs are exact in fp32; the dropped Wl.hl term is ~2^-22 relative).
//   Everything outside the GEMV (state, gates, head) stays fp32.  Measured error vs the reference is
//   the same as ENGINE 0's (docs/DESIGN_measurement_log_r1_r5.md, "f16x3 on every golden").  Per step and wave 27 MFMAs of ~17 cycles instead of 48 x 32:
//   own quarter 9 x K16, the two other quarters that are adjacent in the exchange row as 9 x K32
//   (v_mfma_f32_16x16x32_f16), the remaining quarter 9 x K16.
// ENGINE 2: "bf16x3" -- W and h are each split into THREE bf16 pieces (24 significant bits: the fp32 operands exactly, over
//   fp32's exponent range) and W.h is the sum of the partial products W_p.h_q, p + q <= 3 of {1,2,3} (all nine but W_3.h_3,
//   which is <= 2^-32 of |W||h|: 1/256 of ONE fp32 rounding of the product), each exact in fp32, on
//   v_mfma_f32_16x16x32_bf16 with fp32 accumulation: 8 products x 2 K halves x 3 gates = 48 MFMAs of ~16 cycles per wave
//   and step instead of 48 x 32.  The bf16 matrix pipe leaves half of its cycles to the vector pipe (the fp32 one does not),
//   so the r and n gate math is issued INSIDE the MFMA block; the step has its own order (step_b below):
//     barrier 1 (the hi pieces of h_{t-1} are visible) -> ds_read hi; in the shadow of that round trip the mid / lo pieces
//     are formed and published, the head partial of y_{t-1} parked -> 18 MFMAs on the hi pieces -> barrier 2 -> ds_read
//     mid / lo -> 30 MFMAs (r chain, n chain with the r sigmoid in its gaps, z chain with the n gate in its gaps) -> z
//     sigmoid, blend -> hi piece of h_t published.
//   Exchange layout: buffer[K half m][piece p][lane l] = 16 B = [quarter 2m: units 4q..4q+3 | quarter 2m+1: the same] of
//   stream j (l = 16 q + j) -- the B operand of the K = 32 MFMA as ONE conflict-free ds_read_b128; wave w writes its 8 bytes
//   at half w >> 1, offset 8 (w & 1).  Everything outside the GEMV (state, gates, head) is ENGINE 0's fp32 code.
// YPN = 16: every lane parks its 4-unit head partial (16 planes per y tile, 152 KB of LDS, one workgroup per
//   CU -- right for B <= 4096 where a CU has a single 16-stream group anyway).
// YPN = 4: the four lane groups of a wave are summed first (two permlane swaps), 4 planes, 53 KB of LDS:
//   two or three workgroups share a CU and one group's MFMAs overlap another group's gate math
//   (overlap only exists across waves, docs/DESIGN_measurement_log_r1_r5.md §4) -- used when B >= 8192.
// FUSE = true: the DiffDelRNN step (code/model.py:393-424) in ONE launch.  `a.y` is then pre_d (the GRU + bias-free head)
//   and the time-varying delay line (code/model.py:269-320) writes a.yd from it inside the y-tile housekeeping, one
//   64-sample tile behind the recurrence and spread over three compile-time positions of a tile so that no load
//   latency lands on the recurrence's critical path:
//     phase  2 of tile i    the thread's 4 delays of tile i-1 are fetched (next to the flush that stores pre_d tile i-1);
//                           before that, the taps fetched for tile i-2 are consumed and y tile i-2 is stored;
//     phase 34              s_waitcnt vmcnt(0): every thread's pre_d stores of tile i-1 have completed, and the barrier
//                           of step 35 orders them before any load issued after it (one workgroup = one CU = one L1);
//     phase 36              the delays are classified and the taps of tile i-1 fetched FROM THE KERNEL'S OWN pre_d
//                           OUTPUT (L2: the taps lie at most D samples back): the two taps of a sample are adjacent,
//                           one 8-byte load per sample at 4-byte alignment, when all 4 delays lie in [0, D) and no
//                           tap falls into the carried history -- whatever the integer parts do; everything else
//                           (history taps, k = D, d < 0, NaN, d > D) takes delay_sample() at consume time.
//   The arithmetic is delay_math.h's, i.e. bit-identical to the separate pass (delay_apply_kernel) on the same pre_d.
//   The range check (d > D or NaN: the reference's assert, code/model.py:284) raises the caller's sticky flag; the
//   carried buffer is moved on by delay_update_kernel afterwards (it must see the flag of EVERY workgroup).
// ESR = true: the loss leg of the step (code/test-model.py:386-388) in the same launch.  The per-stream sums
//   sum (t - y)^2 and sum t^2 over samples [esr_skip, T) ride in the y-tile flush: a thread's 4 outputs of the tile are in
//   registers there anyway; the matching 16 bytes of the target were fetched one tile earlier (loads of a step ahead of its
//   stores); products and sums in fp64 as esr_sums_kernel forms them (same per-sample terms, other summation order).
//   As a separate pass on a side stream the 2 GB ESR kernel overlapped the NEXT step's launch and cost that launch
//   0.37 ms -- its vector instructions take issue slots and datapath from the one wave per SIMD that feeds the matrix
//   pipe; here it is ~25 instructions per thread and tile.
// DCP = true (with ESR; for FUSE on the DELAYED output, in the fused delay stage): the DCPreESR entry of the loss dict
//   (code/test-model.py:252) in the same flush.  Both
//   signals e = t - y and t pass H(z) = (1 - z^-1)/(1 - R z^-1) from zero state at esr_skip and sum f(e)^2, sum f(t)^2 are
//   accumulated per stream (ntm_esr_dcpre_sums' definition).  The 16 threads that flush a stream's tile are one DPP row:
//   each runs its 4 samples from zero state, a 4-step row_shr scan with the constant ratios R^4, R^8, R^16, R^32 gives
//   every lane the state entering it, lane 15's end state and last input are broadcast to the row (row_newbcast:15) for
//   the next tile -- no LDS traffic (the step barrier's hand-counted lgkmcnt stays valid), ~70 vector instructions per
//   thread and tile.  fp32 filter (scan order instead of the streaming kernel's; both agree with the sequential
//   recursion of the oracle to ~1e-6 relative in the sums), fp64 sums.
#ifndef NTM3_ASM
#define NTM3_ASM 1        // ENGINE 2: the hand-scheduled MFMA / gate block (0: the compiler-scheduled form, same arithmetic, for the A/B)
#endif
template <bool PRESCALE, bool STAMP, int ABL = 0, int ENGINE = 0, int YPN = 16, bool FUSE = false, bool ESR = false, bool DCP = false>
__global__ __launch_bounds__(256, 1) void gru_mfma2_kernel(GruArgs a)
{
    using namespace m2;
    unsigned long long seg[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, last_ = 0, ts_[6];    // [6..11]: the phase-2 steps alone
    (void)seg; (void)last_; (void)ts_;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *hb = smem;                  // [2][4 k][16 stream][20]      (ENGINE 2: [2][2 K halves][3 pieces][64 lanes][4])
    float *xb = hb + 2 * hb_floats(ENGINE);   // [2][16][65]
    float *yp = xb + 2 * SG * XS;      // [2][16 (w,q)][16][68]   (offset 4640 floats: 16-B aligned)

    const int tid = threadIdx.x;
    const int l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = l >> 4, j = l & 15;  // q doubles as the K slot k of the A/B operand layouts
    const int64_t s0 = (int64_t)blockIdx.x * SG;
    const int64_t T = a.T;
    const bool valid = (s0 + j) < a.B;
    constexpr int NB = NTM2_NB;   // own K-steps issued before the barrier (ENGINE 0)
    constexpr float LOG2E = 1.44269504088896340736f;
    constexpr float SRZ = PRESCALE ? -LOG2E : 1.0f;        // scale of the r and z rows
    constexpr float SN = PRESCALE ? 2.0f * LOG2E : 1.0f;   // scale of the n rows

    // ---- resident operands ---------------------------------------------------------------
    // ENGINE 0: A[sigma] = W_g[16w + (l&15)][unit(s, l>>4)],  s = (sigma + 4w) & 15,
    //           unit(s,k) = 16(s>>2) + 4k + (s&3)
    // ENGINE 1: hi/lo fp16 parts of W_g[16w + (l&15)][16 Q + 4(l>>4) + 0..3] for quarter Q:
    //           Ah/Al[g][0] Q = w (own), Ah/Al[g][1] Q = w^1 (single), A8h/A8l[g] Q = pa, pa+1 (pair,
    //           pa = 2 for waves 0,1 and 0 for waves 2,3: the half of the row that does not hold w)
    float Ar[16], Az[16], An[16];
    f16x4 Ah[3][2], Al[3][2];
    f16x8 A8h[3], A8l[3];
    // ENGINE 2: A3[g][m][p] = bf16 piece p of W_g[16w + (l&15)][16 (2m) + 4(l>>4) + 0..3 | 16 (2m+1) + 4(l>>4) + 0..3], g = r, n, z
    bf16x8 A3[3][2][3];
    const int pa = (w < 2) ? 2 : 0;
    {
        const int row = 16 * w + j;
        const float *pr = a.w_hh + (size_t)(0 * kH + row) * kH + 4 * q;
        const float *pz = a.w_hh + (size_t)(1 * kH + row) * kH + 4 * q;
        const float *pn = a.w_hh + (size_t)(2 * kH + row) * kH + 4 * q;
        if constexpr (ENGINE == 0) {
#pragma unroll
            for (int sg = 0; sg < 16; ++sg) {
                const int s = (sg + 4 * w) & 15;
                const int u0 = 16 * (s >> 2) + (s & 3);
                Ar[sg] = pr[u0] * SRZ; Az[sg] = pz[u0] * SRZ; An[sg] = pn[u0] * SN;
            }
        } else if constexpr (ENGINE == 2) {
            const float *pg[3] = {pr, pn, pz};
            const float sc[3] = {SRZ, SN, SRZ};
#pragma unroll
            for (int g = 0; g < 3; ++g) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    bf16x2 p0[4], p1[4], p2[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {       // i = 0,1: quarter 2m, units 4q + 0..3;  i = 2,3: quarter 2m + 1
                        const float *pq = pg[g] + 16 * (2 * m + (i >> 1)) + 2 * (i & 1);
                        split_bf16x3((f32x2){pq[0] * sc[g], pq[1] * sc[g]}, p0[i], p1[i], p2[i]);
                    }
                    A3[g][m][0] = (bf16x8){p0[0][0], p0[0][1], p0[1][0], p0[1][1], p0[2][0], p0[2][1], p0[3][0], p0[3][1]};
                    A3[g][m][1] = (bf16x8){p1[0][0], p1[0][1], p1[1][0], p1[1][1], p1[2][0], p1[2][1], p1[3][0], p1[3][1]};
                    A3[g][m][2] = (bf16x8){p2[0][0], p2[0][1], p2[1][0], p2[1][1], p2[2][0], p2[2][1], p2[3][0], p2[3][1]};
                }
            }
        } else {
            const float *pg[3] = {pr, pn, pz};
            const float sc[3] = {SRZ, SN, SRZ};
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                auto quarter = [&](int Q, f16x4 &hi, f16x4 &lo) {
                    const float *p = pg[g] + 16 * Q;
                    split_f16((f32x4){p[0] * sc[g], p[1] * sc[g], p[2] * sc[g], p[3] * sc[g]}, hi, lo);
                };
                f16x4 h0, l0, h1, l1;
                quarter(w, Ah[g][0], Al[g][0]);
                quarter(w ^ 1, Ah[g][1], Al[g][1]);
                quarter(pa, h0, l0);
                quarter(pa + 1, h1, l1);
                A8h[g] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                A8l[g] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
    }
    // per-lane gate parameters of units 16w+4q+v, as packed pairs (v = 0,1 | 2,3)
    f32x2 wir[2], wiz[2], win[2], br[2], bz[2], bin_[2], bhn[2], wo[2], hold[2];
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
        wo[v >> 1][v & 1] = a.w_o[u];
        hold[v >> 1][v & 1] = (a.h_state && valid) ? a.h_state[(s0 + j) * kH + u] : 0.0f;
    }
    const float bo = a.b_o ? a.b_o[0] : 0.0f;

    auto load_x_tile = [&](int64_t tile, float (&xr)[4]) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int e = tid + 256 * c;
            const int64_t st = s0 + (e >> 6), tt = tile * TT + (e & 63);
            xr[c] = (st < a.B && tt < T) ? a.x[st * a.xs + tt] : 0.0f;
        }
    };
#if NTM2_HKTRIM
    // The same loads for a tile that lies entirely inside [0, T): no bound on the sample index, row pointers computed
    // once (an absent stream reads the workgroup's first row: its column of the batch is never stored).  At one wave per
    // SIMD every instruction of a housekeeping step costs the recurrence ~4.5 cycles, a taken branch ~20 (measurement log 4 K2f).
    const float *xrow[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int e = tid + 256 * c;
        const int64_t st = s0 + (e >> 6);
        xrow[c] = a.x + (st < a.B ? st : s0) * a.xs + (e & 63);
    }
    auto load_x_tile_whole = [&](int64_t tile, float (&xr)[4]) {
#pragma unroll
        for (int c = 0; c < 4; ++c) xr[c] = xrow[c][tile * TT];
    };
#endif
    auto store_x_tile = [&](int64_t tile, const float (&xr)[4]) {
        float *dst = xb + (tile & 1) * SG * XS;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int e = tid + 256 * c;
            dst[(e >> 6) * XS + (e & 63)] = xr[c];
        }
    };
    // y tile flush: thread -> stream tid>>4, samples 4*(tid&15)..+3; the 16 partial planes (wave, lane
    // group) are summed in a fixed order (deterministic), one ds_read_b128 per plane.
    const bool y_vec_ok = ((reinterpret_cast<uintptr_t>(a.y) & 15) == 0) && ((a.ys & 3) == 0);
    typedef float f32x4y __attribute__((ext_vector_type(4), aligned(4)));     // 16-byte store at 4-byte alignment
    const bool y_row_ok = (s0 + (tid >> 4)) < a.B;
    float *const y_row = a.y + (y_row_ok ? s0 + (tid >> 4) : s0) * a.ys + 4 * (tid & 15);
    (void)y_row;
    constexpr bool ESR_FLUSH = ESR && !FUSE;     // the sums are taken on y: in the flush for the GRU,
    constexpr bool ESR_DL = ESR && FUSE;         // in the fused delay stage (on the DELAYED output) for the DiffDelGRU
    auto flush_sum = [&](int64_t tile) -> f32x4 {
        const float *src = yp + (tile & 1) * YPN * YP_Q + (tid >> 4) * YS + 4 * (tid & 15);
        // (hipcc keeps four of the reads in flight; issuing all sixteen ahead of the first add was measured: +38 VGPRs and
        //  the plain launch 0.35 ms SLOWER, 55.35 against 55.00 ms on one box -- the regular steps' code changed with it)
        f32x4 v = {bo, bo, bo, bo};
#pragma unroll
        for (int pl = 0; pl < YPN; ++pl) v += *(const f32x4 *)(src + pl * YP_Q);
        return v;
    };
    // ---- ESR: per-thread fp64 sums over the thread's sample column of its stream; the target tile in flight ----------
    double esr_e = 0.0, esr_t = 0.0;
    f32x4 esr_tg = {0.0f, 0.0f, 0.0f, 0.0f};
    int64_t esr_tile = -1;                                   // which tile esr_tg holds (wave-uniform)
    const float *const esr_row = ESR ? a.tgt + (y_row_ok ? s0 + (tid >> 4) : s0) * T + 4 * (tid & 15) : nullptr;
    (void)esr_e; (void)esr_t; (void)esr_tg; (void)esr_tile; (void)esr_row;
    // ONE unconditional 16-byte load into esr_tg on the common path; the general form (first flush of a launch, ragged
    // tail) fetches into registers of its own -- a second, conditional load path into the SAME registers makes hipcc drain
    // the VM counter in front of the common one (measurement log 4 K2f)
    auto esr_fetch_whole = [&](int64_t tile) {
        esr_tg = *(const f32x4y *)(esr_row + tile * TT);
        esr_tile = tile;
    };
    // ---- DCP: the one-pole filter state of the stream's row (row-uniform), the fp64 sums of this thread's columns ----
    static_assert(!DCP || ESR, "the DCPreESR sums ride beside the ESR sums (GRU: in the y-tile flush; DiffDelGRU: in the fused delay stage)");
    double dcp_e = 0.0, dcp_t = 0.0;
    float dcp_ce = 0.0f, dcp_ct = 0.0f;                      // filter outputs at the last sample of the previous tile
    float dcp_le = 0.0f, dcp_lt = 0.0f;                      // filter inputs there
    const float dR1 = a.dcp_R, dR2 = dR1 * dR1, dR3 = dR2 * dR1, dR4 = dR2 * dR2, dR8 = dR4 * dR4, dR16 = dR8 * dR8, dR32 = dR16 * dR16;
    float dRc = 1.0f;                                        // R^(4 c): decay from the tile's start to this lane's first sample
    if constexpr (DCP) {
        for (int i = 0; i < (tid & 15); ++i) dRc *= dR4;
    }
    (void)dcp_e; (void)dcp_t; (void)dcp_ce; (void)dcp_ct; (void)dcp_le; (void)dcp_lt; (void)dR3; (void)dR32; (void)dRc;
    f32x4 esr_used = {0.0f, 0.0f, 0.0f, 0.0f};               // the target values the last esr_accumulate saw
    (void)esr_used;
    // (bound_ctrl = 1: a lane without a source reads 0, so the move needs no `old` register set up in front of it and
    //  hipcc can fold it into the consuming v_fmac as a DPP operand)
    auto dpp_shr = [](float x, auto n_c) {                   // lane c of a 16-lane row <- lane c - N (0 where there is none)
        constexpr int N = decltype(n_c)::value;
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x110 + N, 0xf, 0xf, true));
    };
    auto dpp_last = [](float x) {                            // every lane of a row <- lane 15 of the row (row_newbcast:15)
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x15F, 0xf, 0xf, true));
    };
    // one signal of one tile: u[0..3] = this thread's 4 filter inputs (zero outside [skip, T)), `last` / `state` the row's
    // carried input / output; -> the 4 filter outputs in u, carries moved on
    auto dcp_filter = [&](float (&u)[4], float &last, float &state) {
        using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        using I4 = std::integral_constant<int, 4>; using I8 = std::integral_constant<int, 8>;
        // the input sample before this thread's four: lane c - 1's last one, for lane 0 of the row the carried one -- as ONE
        // DPP move whose `old` operand is the carried value (row_shr:1 leaves lane 0 of a row unwritten).  Written as a
        // select, hipcc sinks the DPP move into the branch of lanes 1..15, where lane 1 then reads an EXEC-disabled lane 0
        // and gets nothing.
        const float prev = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, last), __builtin_bit_cast(int, u[3]),
                                                                                 0x111, 0xf, 0xf, false));
        const float f0 = u[0] - prev;
        const float f1 = __builtin_fmaf(dR1, f0, u[1] - u[0]);
        const float f2 = __builtin_fmaf(dR1, f1, u[2] - u[1]);
        const float f3 = __builtin_fmaf(dR1, f2, u[3] - u[2]);
        last = dpp_last(u[3]);
        float b = f3;                                         // inclusive scan of the lanes' end values, ratio R^4 per lane
        b = __builtin_fmaf(dR4, dpp_shr(b, I1{}), b);
        b = __builtin_fmaf(dR8, dpp_shr(b, I2{}), b);
        b = __builtin_fmaf(dR16, dpp_shr(b, I4{}), b);
        b = __builtin_fmaf(dR32, dpp_shr(b, I8{}), b);
        const float in = __builtin_fmaf(dRc, state, dpp_shr(b, I1{}));      // state entering this lane's first sample
        u[0] = __builtin_fmaf(dR1, in, f0);
        u[1] = __builtin_fmaf(dR2, in, f1);
        u[2] = __builtin_fmaf(dR3, in, f2);
        u[3] = __builtin_fmaf(dR4, in, f3);
        state = dpp_last(u[3]);
    };
    auto dcp_accumulate = [&](int64_t tile, const f32x4 v, const f32x4 tg, auto whole_c) {
        constexpr bool WHOLE = decltype(whole_c)::value;
        const int64_t gt = tile * TT + 4 * (tid & 15);
        float ue[4], ut[4];
        bool in[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            in[c] = WHOLE ? gt >= a.esr_skip : (gt + c >= a.esr_skip && gt + c < T);
            ut[c] = in[c] ? tg[c] : 0.0f;
            ue[c] = in[c] ? tg[c] - v[c] : 0.0f;
        }
        dcp_filter(ue, dcp_le, dcp_ce);
        dcp_filter(ut, dcp_lt, dcp_ct);
        // the thread's four squares are added in fp32 (relative error 1e-7 of a 4-term sum), the tile's contribution goes
        // into the fp64 sum: 2 conversions per tile instead of 8 (the streaming kernel converts every sample; the
        // difference is far inside the 2e-6 by which the two filters' evaluation orders differ)
        float qe = 0.0f, qt = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool on = WHOLE || in[c];                   // (whole tiles ahead of esr_skip add exact zeros)
            qe = __builtin_fmaf(on ? ue[c] : 0.0f, ue[c], qe);
            qt = __builtin_fmaf(on ? ut[c] : 0.0f, ut[c], qt);
        }
        dcp_e += (double)qe;
        dcp_t += (double)qt;
    };
    auto esr_accumulate = [&](int64_t tile, const f32x4 v, auto whole_c) {
        constexpr bool WHOLE = decltype(whole_c)::value;
        const int64_t gt = tile * TT + 4 * (tid & 15);
        f32x4 tg = esr_tg;
        if (__builtin_expect(esr_tile != tile, 0)) {
#pragma unroll
            for (int c = 0; c < 4; ++c) tg[c] = gt + c < T ? esr_row[tile * TT + c] : 0.0f;
        }
        if constexpr (DCP) dcp_accumulate(tile, v, tg, whole_c);
        if (WHOLE ? gt >= a.esr_skip : true) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (!WHOLE && !(gt + c >= a.esr_skip && gt + c < T)) continue;
                const float e = tg[c] - v[c];
                esr_e += (double)e * (double)e;
                esr_t += (double)tg[c] * (double)tg[c];
            }
        }
    };
    auto flush_store = [&](int64_t tile, const f32x4 v, auto whole_c) {
        constexpr bool WHOLE = decltype(whole_c)::value && NTM2_HKTRIM;
        if constexpr (WHOLE) {
            // a tile that lies entirely inside [0, T): one predicated 16-byte store (any alignment), nothing else
            if (y_row_ok) {
                *(f32x4y *)(y_row + tile * TT) = v;
                if constexpr (FUSE) {
                    if (__builtin_expect(a.warmup != 0, 0))      // warm-up mode: the delay line passes pre_d through
                        *(f32x4y *)(a.yd + (s0 + (tid >> 4)) * T + 4 * (tid & 15) + tile * TT) = v;
                }
            }
            return;
        }
        const int64_t gs = s0 + (tid >> 4), gt = tile * TT + 4 * (tid & 15);
        if (gs < a.B) {
            float *dst = a.y + gs * a.ys + gt;
            if (__builtin_expect(y_vec_ok && gt + 3 < T, 1)) {
                *(f32x4 *)dst = v;
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (gt + c < T) dst[c] = v[c];
            }
            if constexpr (FUSE) {
                if (__builtin_expect(a.warmup != 0, 0)) {          // warm-up mode: the delay line passes pre_d through (code/model.py:288-292)
                    float *dw = a.yd + gs * T + gt;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (gt + c < T) dw[c] = v[c];
                }
            }
        }
    };
    auto flush_y_tile = [&](int64_t tile, auto whole_c) {      // sum + (ESR) + store in one go: epilogue, non-ESR steps
        const f32x4 v = flush_sum(tile);
        if constexpr (ESR_FLUSH) esr_accumulate(tile, v, std::false_type{});
        flush_store(tile, v, whole_c);
    };

    // ---- FUSE: the delay line, one tile behind the recurrence (see the comment above the kernel) ----------------
    // thread -> stream tid>>4, samples 4*(tid&15)..+3 of a tile, as in the flush.  All addresses are the workgroup's
    // uniform row-block base + a 32-bit element offset (the launcher bounds 16 T below 2^31).
    const bool dl_row = FUSE && (s0 + (tid >> 4)) < a.B;
    // the workgroup's 16 rows of pre_d / d / y as wave-uniform BYTE bases + 32-bit byte offsets (the launcher bounds
    // 64 T below 2^32): the accesses take the scalar-base addressing mode, one v_add per address
    const char *const dl_xw = FUSE ? (const char *)(a.y + s0 * a.ys) : nullptr;     // pre_d (ys == T)
    const char *const dl_dw = FUSE ? (const char *)(a.dd + s0 * T) : nullptr;
    char *const dl_yw = FUSE ? (char *)(a.yd + s0 * T) : nullptr;
    const int dl_T = (int)T;
    const unsigned dl_ro = dl_row ? (unsigned)(tid >> 4) * (unsigned)dl_T * 4u : 0u;  // byte offset of this thread's row (an
                                                                                     // absent stream reads row 0, stores nothing)
    // the delay part runs unless this is a warm-up call or an earlier violation froze the state (sticky flag)
    const bool dl_on = FUSE && !a.warmup && !(a.dl_flag && *a.dl_flag);
    const unsigned dl_dbits = __float_as_uint((float)a.D);                   // 0 <= d < D  <=>  bits(d) < bits(D) (unsigned)
    f32x4 dl_d = {0.0f, 0.0f, 0.0f, 0.0f};                                   // the 4 delays of the tile in flight
    f32x2u dl_t0 = {0.0f, 0.0f}, dl_t1 = dl_t0, dl_t2 = dl_t0, dl_t3 = dl_t0; // (x[n-k-1], x[n-k]) of its 4 samples
    bool dl_fast = false;
    int dl_bad = 0;
    int dl_tile = 0;                                                         // tile in flight (wave-uniform)
    int dl_stage = 0;                                                        // 0 nothing in flight, 1 delays fetched, 2 taps fetched
    (void)dl_xw; (void)dl_dw; (void)dl_yw; (void)dl_ro; (void)dl_on; (void)dl_dbits; (void)dl_fast; (void)dl_bad;
    // Each of the three stages holds ONE global access per array on its common path (16-byte accesses at 4-byte alignment,
    // so rows need no alignment): hipcc's wait insertion merges the pending-load state of every branch of a step, and a
    // second, conditional load path into the same registers made it drain the VM counter in front of the common one.
    // WHOLE (compile time): the tile lies entirely inside [0, T) -- true at the housekeeping positions of the unrolled
    // whole-tile loop, whose delay tile is an earlier one -- so no bound on T is tested; otherwise the ragged tail of a
    // row (its last T % 4 samples) is left to the general form entirely.
    auto dl_load_d = [&](int tile, auto whole_c) { // stage 0 -> 1
        constexpr bool WHOLE = decltype(whole_c)::value;
        const int gt = tile * TT + 4 * (tid & 15);
        if (WHOLE || gt + 3 < dl_T) dl_d = *(const f32x4u *)(dl_dw + (dl_ro + 4u * (unsigned)gt));
        dl_tile = tile;
        dl_stage = 1;
    };
    auto dl_issue_taps = [&](auto whole_c) {       // stage 1 -> 2: pre_d up to the end of dl_tile is visible
        constexpr bool WHOLE = decltype(whole_c)::value;
        const int gt = dl_tile * TT + 4 * (tid & 15);
        // sample c = output n = gt + c with integer part k_c: its taps x[n-k-1], x[n-k] are ONE 8-byte load at element
        // i_c = n - k - 1.  Fast when every d lies in [0, D) (then k + 1 <= D; NaN and negative values fail the unsigned
        // compare of the bit patterns) and every i_c >= 0 (no tap in the carried history); i_c + 1 = n - k <= n < T.
        // (__float_as_uint, not __builtin_bit_cast: hipcc of ROCm 7.2 folds a bit_cast of an ext-vector ELEMENT to element 0)
        const int i0 = gt - 1 - (int)floorf(dl_d[0]), i1 = gt - (int)floorf(dl_d[1]);
        const int i2 = gt + 1 - (int)floorf(dl_d[2]), i3 = gt + 2 - (int)floorf(dl_d[3]);
        const unsigned b0 = __float_as_uint(dl_d[0]), b1 = __float_as_uint(dl_d[1]);
        const unsigned b2 = __float_as_uint(dl_d[2]), b3 = __float_as_uint(dl_d[3]);
        const unsigned bmax = max(max(b0, b1), max(b2, b3));
        const int imin = min(min(i0, i1), min(i2, i3));
        dl_fast = bmax < dl_dbits && imin >= 0 && (WHOLE || gt + 3 < dl_T);
        // unconditional loads (lanes off the fast path read the row's first samples: harmless, in range for T >= 2)
        if (WHOLE || dl_T >= 2) {
            dl_t0 = *(const f32x2u *)(dl_xw + (dl_ro + 4u * (unsigned)(dl_fast ? i0 : 0)));
            dl_t1 = *(const f32x2u *)(dl_xw + (dl_ro + 4u * (unsigned)(dl_fast ? i1 : 0)));
            dl_t2 = *(const f32x2u *)(dl_xw + (dl_ro + 4u * (unsigned)(dl_fast ? i2 : 0)));
            dl_t3 = *(const f32x2u *)(dl_xw + (dl_ro + 4u * (unsigned)(dl_fast ? i3 : 0)));
        }
        dl_stage = 2;
    };
    // stage 2 -> 0 in two halves, because hipcc also drains the VM counter between a global STORE and a later global
    // LOAD: a step issues all its loads first, then all its stores.
    f32x4 dl_out = {0.0f, 0.0f, 0.0f, 0.0f};
    int dl_out_gt = 0;
    bool dl_out_full = false;
    auto dl_compute = [&](auto whole_c) {          // y of dl_tile into registers (consumes the taps and the delays)
        constexpr bool WHOLE = decltype(whole_c)::value;
        const int gt = dl_tile * TT + 4 * (tid & 15);
        dl_stage = 0;
        dl_out = (f32x4){delay_pair(dl_d[0], dl_t0), delay_pair(dl_d[1], dl_t1), delay_pair(dl_d[2], dl_t2), delay_pair(dl_d[3], dl_t3)};
        dl_out_gt = gt;
        dl_out_full = dl_row && (WHOLE || gt + 3 < dl_T);
        const bool gen = dl_row && (WHOLE || gt < dl_T) && !dl_fast;
        if (__builtin_expect(__any(gen), 0)) {
            // rare (never under wow and flutter once n > k): history taps, k = D, d < 0, NaN, d > D, and the ragged tail of
            // a row, whose delays are fetched and whose outputs are stored right here
            if (gen)
                delay_general4((const float *)(dl_xw + dl_ro), (const float *)(dl_dw + dl_ro), (float *)(dl_yw + dl_ro),
                               a.dl_buf + (s0 + (tid >> 4)) * (int64_t)a.D, a.D, gt, dl_T, dl_out_full, dl_d, dl_out, dl_bad);
        }
    };
    auto dl_store = [&]() {
        if (dl_out_full) *(f32x4u *)(dl_yw + (dl_ro + 4u * (unsigned)dl_out_gt)) = dl_out;
        dl_out_full = false;
    };

    float xr[4];
    load_x_tile(0, xr);
    store_x_tile(0, xr);
    int64_t next_flush = 0;

    // h_0: already in B-operand form for the own K-steps; publish it, prepare the input terms of step 0
    float hT[4] = {hold[0][0], hold[0][1], hold[1][0], hold[1][1]};
    f16x4 hTh, hTl;                      // ENGINE 1: the same four values as fp16 hi / lo parts
    split_f16((f32x4){hT[0], hT[1], hT[2], hT[3]}, hTh, hTl);
    // ENGINE 2: the hi pieces of this lane's four units (packed pairs), formed when h_t is; its exchange addresses
    bf16x2 h3hi[2];
    h3hi[0] = __builtin_convertvector(hold[0], bf16x2);
    h3hi[1] = __builtin_convertvector(hold[1], bf16x2);
    float *const h3wr = hb + ((w >> 1) * 3 * 64 + l) * 4 + 2 * (w & 1);    // + 256 p + HB3 buffer (floats)
    const float *const h3rd = hb + l * 4;                                  // + 256 (3 m + p) + HB3 buffer
    (void)h3wr; (void)h3rd;
    if constexpr (ENGINE == 0) *(f32x4 *)&hb[0 * HB + q * HB_K + j * HB_J + 4 * w] = (f32x4){hT[0], hT[1], hT[2], hT[3]};
    else if constexpr (ENGINE == 2) *(bf16x4 *)h3wr = (bf16x4){h3hi[0][0], h3hi[0][1], h3hi[1][0], h3hi[1][1]};
    else {
        // ENGINE 1 exchange row (kg, j): [hi q0 | hi q1 | hi q2 | hi q3 | lo q0 | lo q1 | lo q2 | lo q3], 8 B each
        *(f16x4 *)&hb[0 * HB + q * HB_K + j * HB_J + 2 * w] = hTh;
        *(f16x4 *)&hb[0 * HB + q * HB_K + j * HB_J + 8 + 2 * w] = hTl;
    }
    __syncthreads();   // x tile 0 visible (the h_0 writes are covered by step 0's barrier as well)
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
    float *const yp_lane = yp + (YPN == 16 ? w * 4 + q : w) * YP_Q + j * YS;
    // LDS addresses of the h exchange (buffer 0; buffer 1 is a compile-time +HB in the unrolled loop)
    const float *const hrd1 = hb + q * HB_K + j * HB_J + 4 * ((w + 1) & 3);
    const float *const hrd2 = hb + q * HB_K + j * HB_J + 4 * ((w + 2) & 3);
    const float *const hrd3 = hb + q * HB_K + j * HB_J + 4 * ((w + 3) & 3);
    float *const hwr = hb + q * HB_K + j * HB_J + 4 * w;
    float *const hrow = hb + q * HB_K + j * HB_J;   // ENGINE 1 addressing
    const int ps = w ^ 1;

    // hk_c: tile housekeeping of a step, decided at COMPILE time in whole tiles (0 none, 1 the ph == 2 work, 2 the ph == 34
    // work, 3 the ph == 36 work of FUSE) so that the step carries no phase tests; -1 = test ph at run time (ragged last tile)
    auto housekeeping = [&](const int64_t t, const int ph, const int64_t tile, auto hk_c) __attribute__((always_inline)) {
        constexpr int HK = decltype(hk_c)::value;
        (void)t; (void)ph; (void)tile;
    if constexpr (!(ABL & 32)) {
        if (HK == 1 || (HK < 0 && ph == 2)) {
            if constexpr (FUSE) {
                // y of the tile whose taps were fetched at phase 36 of the previous tile, into registers; below, this
                // tile's delays -- every load of the step ahead of every store.
                // The wait is the BUILTIN (vmcnt(0) only: simm16 0x0F70 leaves expcnt / lgkmcnt alone) so that hipcc's
                // own wait bookkeeping sees it: with the taps' consumers behind run-time stage tests it otherwise
                // keeps "a load into these registers may be pending" alive around the loop and drains the VM counter
                // in front of the NEXT loads -- the x tile fetched just before them, a full memory round trip.
                __builtin_amdgcn_s_waitcnt(0x0F70);
                if (__builtin_expect(dl_stage == 2, 1)) {
                    dl_compute(std::bool_constant<(HK > 0)>{});
                    // the loss leg of the DiffDelGRU step: the ESR terms of the tile whose y has just been formed
                    if constexpr (ESR_DL) esr_accumulate(dl_tile, dl_out, std::bool_constant<(HK > 0)>{});
                }
            }
            // the x loads go out before the flush's stores (no VMEM drain between them)
            f32x4 fv = {0.0f, 0.0f, 0.0f, 0.0f};
            if constexpr (ESR_FLUSH) {
                // the flushed tile's sums and its ESR terms first: they consume the target fetched one tile ago, and
                // nothing newer may be in flight when hipcc waits for it (it waits for everything)
                __builtin_amdgcn_s_waitcnt(0x0F70);
                if (__builtin_expect(t > 65, 1)) {
                    fv = flush_sum(next_flush);
                    esr_accumulate(next_flush, fv, std::bool_constant<(HK > 0)>{});
                }
            }
#if NTM2_HKTRIM
            if (__builtin_expect((tile + 2) * TT <= T, 1)) load_x_tile_whole(tile + 1, xr);
            else if ((tile + 1) * TT < T) load_x_tile(tile + 1, xr);
#else
            if (__builtin_expect((tile + 1) * TT < T, 1)) load_x_tile(tile + 1, xr);
#endif
            if constexpr (ESR_FLUSH) {
                // the next flush's target (tile next_flush + 1 = this tile when t > 65): whole inside the unrolled loop
                if (__builtin_expect(t > 65, 1)) {
                    if (HK > 0) esr_fetch_whole(next_flush + 1);
                    flush_store(next_flush, fv, std::bool_constant<(HK > 0)>{});
                    ++next_flush;
                }
            }
            if constexpr (FUSE) {
                if (__builtin_expect(dl_on && t > 65, 1)) {
                    dl_load_d((int)next_flush, std::bool_constant<(HK > 0)>{});
                    if constexpr (ESR_DL) {
                        if (HK > 0) esr_fetch_whole(next_flush);       // the target of the delay tile now in flight
                    }
                }
                dl_store();
            }
            if constexpr (!ESR_FLUSH) {
                if (__builtin_expect(t > 65, 1)) { flush_y_tile(next_flush, std::bool_constant<(HK > 0)>{}); ++next_flush; }
            }
        } else if (HK == 2 || (HK < 0 && ph == 34)) {
            if (__builtin_expect((tile + 1) * TT < T, 1)) store_x_tile(tile + 1, xr);
            // FUSE: this thread's pre_d stores of phase 2 have completed; step 35's barrier makes that true of the
            // whole workgroup before phase 36 reads them back
            if constexpr (FUSE) __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0), as a MachineInstr hipcc accounts for
        } else if (FUSE && (HK == 3 || (HK < 0 && ph == 36))) {
            if (__builtin_expect(dl_stage == 1, 1)) dl_issue_taps(std::bool_constant<(HK > 0)>{});
        }
    }

    };
    if constexpr (STAMP) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_)::"memory");
    // hk_c: tile housekeeping of this step, decided at COMPILE time in whole tiles (0 none, 1 the ph == 2 work,
    // 2 the ph == 34 work) so that the step carries no phase tests; -1 = test ph at run time (ragged last tile)
    auto step = [&](const int64_t t, auto cur_c, auto hk_c) {
        constexpr int cur = decltype(cur_c)::value;   // == t & 1: which exchange buffer holds h_{t-1}
        constexpr int HK = decltype(hk_c)::value;
        const int ph = (int)(t & 63);
        const int64_t tile = t >> 6;
        float hB[16];
        f16x4 Bh[2], Bl[2];
        f16x8 B8h, B8l;
#pragma unroll
        for (int i = 0; i < 4; ++i) hB[i] = hT[i];
        Bh[0] = hTh; Bl[0] = hTl;

        // ---- the MFMAs of the step, issued back to back: the own quarter comes straight from
        //      registers, the rest from the three ds_read_b128 issued behind the barrier ---------------
        f32x4 acc_r = {cr[0][0], cr[0][1], cr[1][0], cr[1][1]};
        f32x4 acc_n = {bhn[0][0], bhn[0][1], bhn[1][0], bhn[1][1]};
        f32x4 acc_z = {cz[0][0], cz[0][1], cz[1][0], cz[1][1]};
        if constexpr (ENGINE == 0) {
#pragma unroll
            for (int sg = 0; sg < NB; ++sg) {
                acc_r = mfma16x(Ar[sg], hB[sg], acc_r);
#if NTM2_ORDER
                acc_z = mfma16x(Az[sg], hB[sg], acc_z);
                acc_n = mfma16x(An[sg], hB[sg], acc_n);
#else
                acc_n = mfma16x(An[sg], hB[sg], acc_n);
                acc_z = mfma16x(Az[sg], hB[sg], acc_z);
#endif
            }
        } else {
            acc_r = __builtin_amdgcn_mfma_f32_16x16x16f16(Ah[0][0], Bh[0], acc_r, 0, 0, 0);
            acc_n = __builtin_amdgcn_mfma_f32_16x16x16f16(Ah[1][0], Bh[0], acc_n, 0, 0, 0);
            acc_z = __builtin_amdgcn_mfma_f32_16x16x16f16(Ah[2][0], Bh[0], acc_z, 0, 0, 0);
        }
        // the step's only barrier: every wave's ds_write_b128 of h_{t-1} has completed (lgkmcnt(1): LDS
        // ops retire in order and the only younger one is the y partial write, which may stay in
        // flight -- its readers are two barriers away); three MFMAs are already in the pipe.
        if constexpr (ABL & 16) asm volatile("" : "+v"(acc_r), "+v"(acc_n), "+v"(acc_z)::"memory");
        else if constexpr (STAMP) {
            NTM2_STAMP(0)   // step start .. own MFMAs 1-3 issued
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(acc_r), "+v"(acc_n), "+v"(acc_z)::"memory");
            NTM2_STAMP(1)   // wait for the own h write
            asm volatile("s_barrier" : "+v"(acc_r), "+v"(acc_n), "+v"(acc_z)::"memory");
            NTM2_STAMP(2)   // barrier
        } else asm volatile("s_waitcnt lgkmcnt(1)\n\ts_barrier" : "+v"(acc_r), "+v"(acc_n), "+v"(acc_z)::"memory");
        if constexpr (ABL & 2) {
#pragma unroll
            for (int i = 4; i < 16; ++i) hB[i] = hT[i & 3];
        } else if constexpr (ENGINE == 0) {
            const f32x4 v1 = *(const f32x4 *)(hrd1 + cur * HB);
            const f32x4 v2 = *(const f32x4 *)(hrd2 + cur * HB);
            const f32x4 v3 = *(const f32x4 *)(hrd3 + cur * HB);
            if constexpr (ENGINE == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { hB[4 + i] = v1[i]; hB[8 + i] = v2[i]; hB[12 + i] = v3[i]; }
            }
        }
        if constexpr (ENGINE == 1 && !(ABL & 2)) {
            B8h = *(const f16x8 *)(hrow + cur * HB + 2 * pa);
            B8l = *(const f16x8 *)(hrow + cur * HB + 8 + 2 * pa);
            Bh[1] = *(const f16x4 *)(hrow + cur * HB + 2 * ps);
            Bl[1] = *(const f16x4 *)(hrow + cur * HB + 8 + 2 * ps);
        }
        // x of step t+1 (its tile was staged at ph 34 of the previous tile at the latest)
        float xn = xb[(((t + 1) >> 6) & 1) * SG * XS + j * XS + (int)((t + 1) & 63)];
        if constexpr (ENGINE == 0) {
#pragma unroll
            for (int sg = NB; sg < ((ABL & 8) ? 4 : 16); ++sg) {
                acc_r = mfma16x(Ar[sg], hB[sg], acc_r);
#if NTM2_ORDER
                acc_z = mfma16x(Az[sg], hB[sg], acc_z);
                acc_n = mfma16x(An[sg], hB[sg], acc_n);
#else
                acc_n = mfma16x(An[sg], hB[sg], acc_n);
                acc_z = mfma16x(Az[sg], hB[sg], acc_z);
#endif
                if (sg == 3) asm volatile("" : "+v"(acc_r), "+v"(acc_n), "+v"(acc_z));
            }
        } else {
            // own quarter: the two cross terms; then the adjacent pair of other quarters as K=32 MFMAs and
            // the remaining quarter as K=16, three terms each
            acc_r = __builtin_amdgcn_mfma_f32_16x16x16f16(Ah[0][0], Bl[0], acc_r, 0, 0, 0);
            acc_n = __builtin_amdgcn_mfma_f32_16x16x16f16(Ah[1][0], Bl[0], acc_n, 0, 0, 0);
            acc_z = __builtin_amdgcn_mfma_f32_16x16x16f16(Ah[2][0], Bl[0], acc_z, 0, 0, 0);
            acc_r = __builtin_amdgcn_mfma_f32_16x16x16f16(Al[0][0], Bh[0], acc_r, 0, 0, 0);
            acc_n = __builtin_amdgcn_mfma_f32_16x16x16f16(Al[1][0], Bh[0], acc_n, 0, 0, 0);
            acc_z = __builtin_amdgcn_mfma_f32_16x16x16f16(Al[2][0], Bh[0], acc_z, 0, 0, 0);
            asm volatile("" : "+v"(acc_r), "+v"(acc_n), "+v"(acc_z));
            if constexpr (!(ABL & 8)) {
                acc_r = __builtin_amdgcn_mfma_f32_16x16x32_f16(A8h[0], B8h, acc_r, 0, 0, 0);
                acc_n = __builtin_amdgcn_mfma_f32_16x16x32_f16(A8h[1], B8h, acc_n, 0, 0, 0);
                acc_z = __builtin_amdgcn_mfma_f32_16x16x32_f16(A8h[2], B8h, acc_z, 0, 0, 0);
                acc_r = __builtin_amdgcn_mfma_f32_16x16x32_f16(A8h[0], B8l, acc_r, 0, 0, 0);
                acc_n = __builtin_amdgcn_mfma_f32_16x16x32_f16(A8h[1], B8l, acc_n, 0, 0, 0);
                acc_z = __builtin_amdgcn_mfma_f32_16x16x32_f16(A8h[2], B8l, acc_z, 0, 0, 0);
                acc_r = __builtin_amdgcn_mfma_f32_16x16x32_f16(A8l[0], B8h, acc_r, 0, 0, 0);
                acc_n = __builtin_amdgcn_mfma_f32_16x16x32_f16(A8l[1], B8h, acc_n, 0, 0, 0);
                acc_z = __builtin_amdgcn_mfma_f32_16x16x32_f16(A8l[2], B8h, acc_z, 0, 0, 0);
                acc_r = __builtin_amdgcn_mfma_f32_16x16x16f16(Ah[0][1], Bh[1], acc_r, 0, 0, 0);
                acc_n = __builtin_amdgcn_mfma_f32_16x16x16f16(Ah[1][1], Bh[1], acc_n, 0, 0, 0);
                acc_z = __builtin_amdgcn_mfma_f32_16x16x16f16(Ah[2][1], Bh[1], acc_z, 0, 0, 0);
                acc_r = __builtin_amdgcn_mfma_f32_16x16x16f16(Ah[0][1], Bl[1], acc_r, 0, 0, 0);
                acc_n = __builtin_amdgcn_mfma_f32_16x16x16f16(Ah[1][1], Bl[1], acc_n, 0, 0, 0);
                acc_z = __builtin_amdgcn_mfma_f32_16x16x16f16(Ah[2][1], Bl[1], acc_z, 0, 0, 0);
                acc_r = __builtin_amdgcn_mfma_f32_16x16x16f16(Al[0][1], Bh[1], acc_r, 0, 0, 0);
                acc_n = __builtin_amdgcn_mfma_f32_16x16x16f16(Al[1][1], Bh[1], acc_n, 0, 0, 0);
                acc_z = __builtin_amdgcn_mfma_f32_16x16x16f16(Al[2][1], Bh[1], acc_z, 0, 0, 0);
            }
        }
        // (xn is tied in so that its consumers -- the input terms of step t+1 -- stay out of the MFMA block)
        asm volatile("" : "+v"(acc_r), "+v"(acc_n), "+v"(acc_z), "+v"(xn));
        NTM2_STAMP(3)   // LDS reads + the other 45 MFMAs issued

        // tile housekeeping, once per 64 steps each (y partials of the previous tile are complete and
        // visible once step 64i+65 has passed its barrier)
        housekeeping(t, ph, tile, hk_c);

        // ---- the VALU block ------------------------------------------------------------------------
        // input terms of step t+1 first: they do not wait for the last MFMAs to drain
        const f32x2 xx = {xn, xn};
        f32x2 ncr[2], ncz[2], ngi[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            ncr[p] = __builtin_elementwise_fma(wir[p], xx, br[p]);
            ncz[p] = __builtin_elementwise_fma(wiz[p], xx, bz[p]);
            ngi[p] = __builtin_elementwise_fma(win[p], xx, bin_[p]);
        }
        // (pinned ahead of the gates: they fill the ~40 cycles the last MFMAs need to drain)
        asm volatile("" : "+v"(ncr[0]), "+v"(ncr[1]), "+v"(ncz[0]), "+v"(ncz[1]), "+v"(ngi[0]), "+v"(ngi[1]),
                          "+v"(acc_r), "+v"(acc_n), "+v"(acc_z));

        const f32x2 one = {1.0f, 1.0f};
        f32x2 hn[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            f32x2 ar = {acc_r[2 * p], acc_r[2 * p + 1]}, an = {acc_n[2 * p], acc_n[2 * p + 1]};
            f32x2 az = {acc_z[2 * p], acc_z[2 * p + 1]};
            if constexpr (ABL & 1) { hn[p] = (ar + an) + (az + gi[p]); continue; }
            if (!PRESCALE) { ar *= -LOG2E; az *= -LOG2E; }
            f32x2 er = {__builtin_amdgcn_exp2f(ar[0]), __builtin_amdgcn_exp2f(ar[1])};
            f32x2 ez = {__builtin_amdgcn_exp2f(az[0]), __builtin_amdgcn_exp2f(az[1])};
            er += one; ez += one;
            const f32x2 r = {__builtin_amdgcn_rcpf(er[0]), __builtin_amdgcn_rcpf(er[1])};
            const f32x2 z = {__builtin_amdgcn_rcpf(ez[0]), __builtin_amdgcn_rcpf(ez[1])};
            f32x2 pn = __builtin_elementwise_fma(r, an, gi[p]);        // (2 log2e) (gi_n + r gh_n) if PRESCALE
            if (!PRESCALE) pn *= 2.0f * LOG2E;
            f32x2 en = {__builtin_amdgcn_exp2f(pn[0]), __builtin_amdgcn_exp2f(pn[1])};
            en += one;
            const f32x2 rn = {__builtin_amdgcn_rcpf(en[0]), __builtin_amdgcn_rcpf(en[1])};
            // tanh as 1 - 2 / (1 + 2^p).  The form sign(p) (1 - t) rcp(1 + t), t = 2^-|p|, has no cancellation for small |n| but cost
            // 1.8 % of the step and left the full-batch worst error at 6.0e-6 (6.8e-6 here): measured in round 6, not adopted
            // (profiles/r06_b_full_batch_parity.jsonl)
            const f32x2 n = __builtin_elementwise_fma((f32x2){-2.0f, -2.0f}, rn, one);
            hn[p] = __builtin_elementwise_fma(z, hold[p] - n, n);                         // n + z (h - n)
        }
        asm volatile("" : "+v"(hn[0]), "+v"(hn[1]));
        NTM2_STAMP(4)   // housekeeping + input terms + gates
        // publish h_t first (it is on the critical path of every wave's next step) ...
#pragma unroll
        for (int p = 0; p < 2; ++p) { hold[p] = hn[p]; cr[p] = ncr[p]; cz[p] = ncz[p]; gi[p] = ngi[p]; }
        hT[0] = hn[0][0]; hT[1] = hn[0][1]; hT[2] = hn[1][0]; hT[3] = hn[1][1];
        if constexpr (ENGINE == 1) split_f16((f32x4){hT[0], hT[1], hT[2], hT[3]}, hTh, hTl);
        if constexpr (!(ABL & 2)) {
            if constexpr (ENGINE == 0) *(f32x4 *)(hwr + (cur ^ 1) * HB) = (f32x4){hT[0], hT[1], hT[2], hT[3]};
            else {
                *(f16x4 *)(hrow + (cur ^ 1) * HB + 2 * w) = hTh;
                *(f16x4 *)(hrow + (cur ^ 1) * HB + 8 + 2 * w) = hTl;
            }
        }
        asm volatile("" ::: "memory");   // keep the two LDS writes in this order (see the barrier's lgkmcnt)
        // ... then the head partial of y_t over this lane's four units
        if constexpr (!(ABL & 4)) {
            const f32x2 pp = __builtin_elementwise_fma(hn[1], wo[1], hn[0] * wo[0]);
            float hp = pp[0] + pp[1];
            if constexpr (YPN == 4) {   // sum the wave's four lane groups; all four then store the same value
                // v_permlane*_swap exchanges halves of TWO registers in place; with a copy of hp as the second
                // one, the two results sum to hp(l) + hp(l^32) resp. hp(l) + hp(l^16).  Written as asm: hipcc
                // (ROCm 7.2) folds the builtin's two results into one register when both operands hold the
                // same value.  Wait states around the swap are inside the string (cdna_hip_programming.md 5.7).
                float hq;
                asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(hp), "=&v"(hq));
                hp += hq;
                asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(hp), "=&v"(hq));
                hp += hq;
            }
            yp_lane[(tile & 1) * YPN * YP_Q + ph] = hp;
        } else {
            yp_lane[0] = 0.0f;           // keep the LDS-op count the barrier's lgkmcnt(1) relies on
        }
        NTM2_STAMP(5)   // h write + head partial + y write issued
        if constexpr (STAMP) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            seg[0] += ts_[0] - last_;
#pragma unroll
            for (int k = 1; k < 6; ++k) seg[k] += ts_[k] - ts_[k - 1];
            if (HK == 1 || (HK < 0 && ph == 2)) {      // the heavy housekeeping step by itself
                seg[6] += ts_[0] - last_;
#pragma unroll
                for (int k = 1; k < 6; ++k) seg[6 + k] += ts_[k] - ts_[k - 1];
            }
            last_ = ts_[5];
        }
    };
    // ---- ENGINE 2: the step in its own order (comment above the kernel) ------------------------------------------------
    // head partial of y_{tp} over this lane's four units of h_{tp} (= hold), parked for the tile flush
    auto park_head = [&](const int64_t tp) __attribute__((always_inline)) {
        const f32x2 pp = __builtin_elementwise_fma(hold[1], wo[1], hold[0] * wo[0]);
        float hp = pp[0] + pp[1];
        if constexpr (YPN == 4) {       // as in step(): the wave's four lane groups summed, all four store the same value
            float hq;
            asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(hp), "=&v"(hq));
            hp += hq;
            asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(hp), "=&v"(hq));
            hp += hq;
        }
        yp_lane[(int)((tp >> 6) & 1) * YPN * YP_Q + (int)(tp & 63)] = hp;
    };
#define NTM3_MF(ACC, G, M, P, B) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A3[G][M][P], B, ACC, 0, 0, 0)
    (void)step;
#if NTM3_ASM
    // state of the asm form (held in fixed registers inside step_b): the fp32 h of this lane's four units and its packed hi
    // pieces; the seeds of the r / z accumulators (the input terms of the step they will run; step 0: from the prologue); the
    // accumulators and the six B operands in flight between the statements; LDS byte addresses for the ds_* inside the strings
    typedef const __attribute__((address_space(3))) float *lds_cptr;
    f32x4 h4 = {hold[0][0], hold[0][1], hold[1][0], hold[1][1]};
    f32x2 p0v = __builtin_bit_cast(f32x2, (bf16x4){h3hi[0][0], h3hi[0][1], h3hi[1][0], h3hi[1][1]});
    // (step 0 is an even step: it reads parity 0 of the seed / gi_n pairs)
    f32x4 seed_r = {cr[0][0], cr[0][1], cr[1][0], cr[1][1]};
    f32x4 seed_z[2] = {{cz[0][0], cz[0][1], cz[1][0], cz[1][1]}, {0.0f, 0.0f, 0.0f, 0.0f}};
    f32x4 gi4[2] = {{gi[0][0], gi[0][1], gi[1][0], gi[1][1]}, {0.0f, 0.0f, 0.0f, 0.0f}};
    const f32x4 bhn4 = {bhn[0][0], bhn[0][1], bhn[1][0], bhn[1][1]};
    const f32x4 wir4 = {wir[0][0], wir[0][1], wir[1][0], wir[1][1]}, br4 = {br[0][0], br[0][1], br[1][0], br[1][1]};
    const f32x4 wiz4 = {wiz[0][0], wiz[0][1], wiz[1][0], wiz[1][1]}, bz4 = {bz[0][0], bz[0][1], bz[1][0], bz[1][1]};
    const f32x4 win4 = {win[0][0], win[0][1], win[1][0], win[1][1]}, bin4 = {bin_[0][0], bin_[0][1], bin_[1][0], bin_[1][1]};
    const f32x4 wo4 = {wo[0][0], wo[0][1], wo[1][0], wo[1][1]};
    f32x4 acc3[3];
    bf16x8 B3[6];
    float x3_next = xb[j * XS + 1];          // x of step 1 (tile 0 is staged; a position beyond T holds 0)
    const unsigned rd_addr = (unsigned)(uintptr_t)(lds_cptr)h3rd, wr_addr = (unsigned)(uintptr_t)(lds_cptr)h3wr;
    const unsigned yp_addr = (unsigned)(uintptr_t)(lds_cptr)yp_lane;
    (void)x3_next; (void)gi4; (void)wir4; (void)br4; (void)wiz4; (void)bz4; (void)win4; (void)bin4; (void)wo4;
    (void)h4; (void)p0v; (void)seed_r; (void)seed_z; (void)bhn4; (void)acc3; (void)B3; (void)rd_addr; (void)wr_addr; (void)yp_addr;
#endif
    auto step_b = [&](const int64_t t, auto cur_c, auto hk_c) {
#if NTM3_ASM
        // ---- the hand-scheduled form: four asm statements from barrier 1 to the publication of h_t's hi piece.  The step is
        //      paced by the matrix pipe, and which vector instruction sits in which MFMA gap decides its length
        //      (tools/ubench/mfma_bf16_gap.hip, cycles per v_mfma_f32_16x16x32_bf16 of one wave: alone 17.6; + one v_exp_f32
        //      18.3; + two v_add_f32 / v_fma_f32 18.5; + v_exp_f32 + v_add_f32 22.6; + four v_add_f32 26.6; + ONE
        //      v_pk_add_f32 34.5 -- packed fp32 ops do not overlap a bf16 MFMA at all, so the gaps hold only plain ops, at most
        //      one transcendental or two simple ones each; the compiler-scheduled form below, with packed ops and hipcc's
        //      own wait states around MFMA results, runs ~1690 cycles per step).
        //      Order of the 48 MFMAs: r hi (1) | barrier 2, ds_read mid / lo | r hi (5), n hi (6); r mid+lo (10)
        //      [the input terms of step t+1, the head partial of y_{t-1}]; n mid+lo (10) [r sigmoid]; z hi + mid + lo (16)
        //      [n gate]; then the z sigmoid, the blend, the hi piece of h_t.  Per accumulator the order of the products is the
        //      compiler form's, so the two forms agree bit for bit (tests/test_gpu_round6.py).
        //      Registers: the strings name single elements of tuples, which asm operands cannot express, so everything they
        //      touch lives in FIXED registers, declared to hipcc as physical-register operands / clobbers:
        //        v[100:103] acc r -> r     v[104:107] acc n     v[108:111] acc z       v[124:127] h (fp32)  v[128:129] hi pieces
        //        v[130:141] scratch: mid / lo pieces and the split's temporaries in the first statement, the head partial in the
        //        second, n gate (130:133), z sigmoid (134:137) and h - n (138:141) in the last
        //        v[142:165] B operands (hi, mid, lo x 2 K halves)    v[166:169] seed of the r accumulator (W_ir x + b)
        //        v[170:173] / v[210:213] seed of the z accumulator and v[174:177] / v[206:209] gi_n, by step parity (the
        //        values of step t+1 are formed before step t has used its own)    v[178:205] W_ih, biases, head weights
        //      Waits inside the strings are by hand: lgkmcnt(0), and one lgkmcnt(2) in the first statement -- behind its own full
        //      wait, where the only four LDS ops in flight are that statement's two hi reads and two mid / lo writes (LDS returns
        //      in order, so an LDS op hipcc has in flight elsewhere is harmless); MFMA result -> vector read >= 28 cycles after the MFMA's issue; transcendental result -> next
        //      vector use one state later; permlane swaps as in park_head.
        constexpr int cur = decltype(cur_c)::value;   // == t & 1: which exchange buffer holds h_{t-1}
        static_assert(PRESCALE, "the asm form of the bf16x3 step folds the log2(e) factors into the weights");
        const int ph = (int)(t & 63);
        const int64_t tile = t >> 6;
        NTM2_STAMP(0)
        housekeeping(t, ph, tile, hk_c);
        // x of step t+1 was fetched during step t-1 (behind the r chain: an LDS read issued here would sit between the hi-piece
        // write and barrier 1 and hold the barrier up for its own latency); the slot the head partial of y_{t-1} parks in
        const float xn = x3_next;
        const unsigned ya = yp_addr + 4u * (unsigned)((int)(((t - 1) >> 6) & 1) * YPN * YP_Q + (int)((t - 1) & 63));
        NTM2_STAMP(1)
#define MF3(D, A, B, C) "v_mfma_f32_16x16x32_bf16 " D ", " A ", " B ", " C "\n\t"
#define NEG " neg_lo:[0,1] neg_hi:[0,1]\n\t"
#define FMA(D, A, B, C) "v_fma_f32 " D ", " A ", " B ", " C "\n\t"
#define NTM3_S1(O_H0, O_H1, O_M0, O_M1, O_L0, O_L1, O_WM, O_WL)                                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier\n\t"                                                                      \
                     "ds_read_b128 v[142:145], %[rd] offset:" O_H0 "\n\tds_read_b128 v[146:149], %[rd] offset:" O_H1 "\n\t"      \
                     "v_lshlrev_b32 v134, 16, v128\n\tv_and_b32 v135, 0xffff0000, v128\n\t"                                        \
                     "v_lshlrev_b32 v136, 16, v129\n\tv_and_b32 v137, 0xffff0000, v129\n\t"                                        \
                     "v_pk_add_f32 v[138:139], v[124:125], v[134:135]" NEG "v_pk_add_f32 v[140:141], v[126:127], v[136:137]" NEG   \
                     "v_cvt_pk_bf16_f32 v130, v138, v139\n\tv_cvt_pk_bf16_f32 v131, v140, v141\n\t"                                \
                     "v_lshlrev_b32 v134, 16, v130\n\tv_and_b32 v135, 0xffff0000, v130\n\t"                                        \
                     "v_lshlrev_b32 v136, 16, v131\n\tv_and_b32 v137, 0xffff0000, v131\n\t"                                        \
                     "v_pk_add_f32 v[138:139], v[138:139], v[134:135]" NEG "v_pk_add_f32 v[140:141], v[140:141], v[136:137]" NEG   \
                     "v_cvt_pk_bf16_f32 v132, v138, v139\n\tv_cvt_pk_bf16_f32 v133, v140, v141\n\t"                                \
                     "ds_write_b64 %[wr], v[130:131] offset:" O_WM "\n\tds_write_b64 %[wr], v[132:133] offset:" O_WL "\n\t"      \
                     "s_waitcnt lgkmcnt(2)\n\t"                                                                                   \
                     MF3("v[100:103]", "%[r00]", "v[142:145]", "v[166:169]")                                                       \
                     "s_waitcnt lgkmcnt(0)\n\ts_barrier\n\t"                                                                     \
                     "ds_read_b128 v[150:153], %[rd] offset:" O_M0 "\n\tds_read_b128 v[154:157], %[rd] offset:" O_M1 "\n\t"      \
                     "ds_read_b128 v[158:161], %[rd] offset:" O_L0 "\n\tds_read_b128 v[162:165], %[rd] offset:" O_L1 "\n\t"      \
                     MF3("v[100:103]", "%[r10]", "v[146:149]", "v[100:103]")                                                       \
                     MF3("v[104:107]", "%[n00]", "v[142:145]", "%[bhn]") MF3("v[104:107]", "%[n10]", "v[146:149]", "v[104:107]")  \
                     MF3("v[100:103]", "%[r01]", "v[142:145]", "v[100:103]") MF3("v[100:103]", "%[r11]", "v[146:149]", "v[100:103]") \
                     MF3("v[100:103]", "%[r02]", "v[142:145]", "v[100:103]") MF3("v[100:103]", "%[r12]", "v[146:149]", "v[100:103]") \
                     MF3("v[104:107]", "%[n01]", "v[142:145]", "v[104:107]") MF3("v[104:107]", "%[n11]", "v[146:149]", "v[104:107]") \
                     MF3("v[104:107]", "%[n02]", "v[142:145]", "v[104:107]") "v_mfma_f32_16x16x32_bf16 v[104:107], %[n12], v[146:149], v[104:107]" \
                     : "=&{v[100:103]}"(acc3[0]), "=&{v[104:107]}"(acc3[1]), "=&{v[142:145]}"(B3[0]), "=&{v[146:149]}"(B3[1]),      \
                       "=&{v[150:153]}"(B3[2]), "=&{v[154:157]}"(B3[3]), "=&{v[158:161]}"(B3[4]), "=&{v[162:165]}"(B3[5])          \
                     : [rd] "v"(rd_addr), [wr] "v"(wr_addr), "{v[166:169]}"(seed_r), "{v[124:127]}"(h4), "{v[128:129]}"(p0v),      \
                       [bhn] "v"(bhn4), [r00] "v"(A3[0][0][0]), [r10] "v"(A3[0][1][0]), [r01] "v"(A3[0][0][1]), [r11] "v"(A3[0][1][1]), \
                       [r02] "v"(A3[0][0][2]), [r12] "v"(A3[0][1][2]), [n00] "v"(A3[1][0][0]), [n10] "v"(A3[1][1][0]),             \
                       [n01] "v"(A3[1][0][1]), [n11] "v"(A3[1][1][1]), [n02] "v"(A3[1][0][2]), [n12] "v"(A3[1][1][2])              \
                     : "memory", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141")
        if constexpr (cur == 0) NTM3_S1("0", "3072", "1024", "4096", "2048", "5120", "1024", "2048");
        else NTM3_S1("6144", "9216", "7168", "10240", "8192", "11264", "7168", "8192");
#undef NTM3_S1
        NTM2_STAMP(2)
        // r chain, mid and lo pieces; in its gaps, two plain ops each: the input terms of step t+1 (seeds of its r and z
        // accumulators, gi_n) and the head partial of y_{t-1} over this lane's four units of h_{t-1}
#define NTM3_S2(SZN, SZN0, SZN1, SZN2, SZN3, GIN, GIN0, GIN1, GIN2, GIN3, PARK)                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)\n\t"                                                                                   \
                     MF3("v[100:103]", "%[r00]", "v[150:153]", "v[100:103]") FMA("v166", "v178", "%[xn]", "v182") FMA("v167", "v179", "%[xn]", "v183") \
                     MF3("v[100:103]", "%[r10]", "v[154:157]", "v[100:103]") FMA("v168", "v180", "%[xn]", "v184") FMA("v169", "v181", "%[xn]", "v185") \
                     MF3("v[100:103]", "%[r01]", "v[150:153]", "v[100:103]") FMA(SZN0, "v186", "%[xn]", "v190") FMA(SZN1, "v187", "%[xn]", "v191")   \
                     MF3("v[100:103]", "%[r11]", "v[154:157]", "v[100:103]") FMA(SZN2, "v188", "%[xn]", "v192") FMA(SZN3, "v189", "%[xn]", "v193")   \
                     MF3("v[100:103]", "%[r02]", "v[150:153]", "v[100:103]") FMA(GIN0, "v194", "%[xn]", "v198") FMA(GIN1, "v195", "%[xn]", "v199")   \
                     MF3("v[100:103]", "%[r12]", "v[154:157]", "v[100:103]") FMA(GIN2, "v196", "%[xn]", "v200") FMA(GIN3, "v197", "%[xn]", "v201")   \
                     MF3("v[100:103]", "%[r00]", "v[158:161]", "v[100:103]") "v_mul_f32 v134, v124, v202\n\tv_mul_f32 v135, v125, v203\n\t"        \
                     MF3("v[100:103]", "%[r10]", "v[162:165]", "v[100:103]") FMA("v134", "v126", "v204", "v134") FMA("v135", "v127", "v205", "v135") \
                     MF3("v[100:103]", "%[r01]", "v[158:161]", "v[100:103]") PARK                                                  \
                     : "+{v[100:103]}"(acc3[0]), "=&{v[166:169]}"(seed_r), "=&{" SZN "}"(seed_z[cur ^ 1]), "=&{" GIN "}"(gi4[cur ^ 1]) \
                     : "{v[150:153]}"(B3[2]), "{v[154:157]}"(B3[3]), "{v[158:161]}"(B3[4]), "{v[162:165]}"(B3[5]),                 \
                       [r00] "v"(A3[0][0][0]), [r10] "v"(A3[0][1][0]), [r01] "v"(A3[0][0][1]), [r11] "v"(A3[0][1][1]),             \
                       [r02] "v"(A3[0][0][2]), [r12] "v"(A3[0][1][2]), [xn] "v"(xn), [ya] "v"(ya), "{v[124:127]}"(h4),             \
                       "{v[178:181]}"(wir4), "{v[182:185]}"(br4), "{v[186:189]}"(wiz4), "{v[190:193]}"(bz4), "{v[194:197]}"(win4), \
                       "{v[198:201]}"(bin4), "{v[202:205]}"(wo4)                                                                  \
                     : "memory", "v134", "v135")
        // (the park sequence ends the statement: the last sum, for YPN = 4 the two permlane swaps over the wave's four lane groups
        //  -- all four then store the same value --, the store, and the chain's last MFMA)
#define NTM3_PARK16 "v_add_f32 v134, v134, v135\n\tds_write_b32 %[ya], v134\n\t"                                                   \
                    "v_mfma_f32_16x16x32_bf16 v[100:103], %[r11], v[162:165], v[100:103]"
#define NTM3_PARK4 "v_add_f32 v134, v134, v135\n\tv_mov_b32 v135, v134\n\t"                                                        \
                   MF3("v[100:103]", "%[r11]", "v[162:165]", "v[100:103]")                                                         \
                   "s_nop 0\n\tv_permlane32_swap_b32 v134, v135\n\ts_nop 1\n\tv_add_f32 v134, v134, v135\n\tv_mov_b32 v135, v134\n\t" \
                   "s_nop 1\n\tv_permlane16_swap_b32 v134, v135\n\ts_nop 1\n\tv_add_f32 v134, v134, v135\n\tds_write_b32 %[ya], v134"
        if constexpr (cur == 0 && YPN == 16) NTM3_S2("v[210:213]", "v210", "v211", "v212", "v213", "v[206:209]", "v206", "v207", "v208", "v209", NTM3_PARK16);
        else if constexpr (cur == 1 && YPN == 16) NTM3_S2("v[170:173]", "v170", "v171", "v172", "v173", "v[174:177]", "v174", "v175", "v176", "v177", NTM3_PARK16);
        else if constexpr (cur == 0) NTM3_S2("v[210:213]", "v210", "v211", "v212", "v213", "v[206:209]", "v206", "v207", "v208", "v209", NTM3_PARK4);
        else NTM3_S2("v[170:173]", "v170", "v171", "v172", "v173", "v[174:177]", "v174", "v175", "v176", "v177", NTM3_PARK4);
#undef NTM3_S2
#undef NTM3_PARK16
#undef NTM3_PARK4
        NTM2_STAMP(3)
        // (hipcc-issued; its tile was staged at phase 34 of the previous tile at the latest) x of step t+2, for the next step's input terms
        x3_next = xb[(((t + 2) >> 6) & 1) * SG * XS + j * XS + (int)((t + 2) & 63)];
        // n chain, mid and lo pieces; from its second gap on the r sigmoid, in place: 4 x v_exp, 4 x v_add, 3 of the 4 x v_rcp
        asm volatile(MF3("v[104:107]", "%[n00]", "v[150:153]", "v[104:107]")
                     MF3("v[104:107]", "%[n10]", "v[154:157]", "v[104:107]") "v_exp_f32 v100, v100\n\t"
                     MF3("v[104:107]", "%[n01]", "v[150:153]", "v[104:107]") "v_exp_f32 v101, v101\n\t"
                     MF3("v[104:107]", "%[n11]", "v[154:157]", "v[104:107]") "v_exp_f32 v102, v102\n\t"
                     MF3("v[104:107]", "%[n02]", "v[150:153]", "v[104:107]") "v_exp_f32 v103, v103\n\t"
                     MF3("v[104:107]", "%[n12]", "v[154:157]", "v[104:107]") "v_add_f32 v100, 1.0, v100\n\tv_add_f32 v101, 1.0, v101\n\t"
                     MF3("v[104:107]", "%[n00]", "v[158:161]", "v[104:107]") "v_add_f32 v102, 1.0, v102\n\tv_add_f32 v103, 1.0, v103\n\t"
                     MF3("v[104:107]", "%[n10]", "v[162:165]", "v[104:107]") "v_rcp_f32 v100, v100\n\t"
                     MF3("v[104:107]", "%[n01]", "v[158:161]", "v[104:107]") "v_rcp_f32 v101, v101\n\t"
                     MF3("v[104:107]", "%[n11]", "v[162:165]", "v[104:107]") "v_rcp_f32 v102, v102"
                     : "+{v[104:107]}"(acc3[1]), "+{v[100:103]}"(acc3[0])
                     : "{v[150:153]}"(B3[2]), "{v[154:157]}"(B3[3]), "{v[158:161]}"(B3[4]), "{v[162:165]}"(B3[5]),
                       [n00] "v"(A3[1][0][0]), [n10] "v"(A3[1][1][0]), [n01] "v"(A3[1][0][1]), [n11] "v"(A3[1][1][1]),
                       [n02] "v"(A3[1][0][2]), [n12] "v"(A3[1][1][2]));
        NTM2_STAMP(4)
        // z chain, all three pieces (16 MFMAs); in its gaps the last v_rcp of the r sigmoid, then the n gate: p_n = r gh_n + gi_n,
        // 4 x v_exp, + 1, 4 x v_rcp, n = 1 - 2 / (...); under the last MFMA's 28 cycles h - n; then the z sigmoid, the blend
        // n + z (h - n) into the h registers, the hi pieces of h_t and their publication -- every wave's next step waits for it
#define NTM3_S4(O_WH, SZC, GIC, GI0, GI1, GI2, GI3)                                                                                      \
        asm volatile(MF3("v[108:111]", "%[z00]", "v[142:145]", SZC) "v_rcp_f32 v103, v103\n\t"                                      \
                     MF3("v[108:111]", "%[z10]", "v[146:149]", "v[108:111]") FMA("v130", "v100", "v104", GI0) FMA("v131", "v101", "v105", GI1) \
                     MF3("v[108:111]", "%[z01]", "v[142:145]", "v[108:111]") FMA("v132", "v102", "v106", GI2) FMA("v133", "v103", "v107", GI3) \
                     MF3("v[108:111]", "%[z11]", "v[146:149]", "v[108:111]") "v_exp_f32 v130, v130\n\t"                            \
                     MF3("v[108:111]", "%[z02]", "v[142:145]", "v[108:111]") "v_exp_f32 v131, v131\n\t"                            \
                     MF3("v[108:111]", "%[z12]", "v[146:149]", "v[108:111]") "v_exp_f32 v132, v132\n\t"                            \
                     MF3("v[108:111]", "%[z00]", "v[150:153]", "v[108:111]") "v_exp_f32 v133, v133\n\t"                            \
                     MF3("v[108:111]", "%[z10]", "v[154:157]", "v[108:111]") "v_add_f32 v130, 1.0, v130\n\tv_add_f32 v131, 1.0, v131\n\t" \
                     MF3("v[108:111]", "%[z01]", "v[150:153]", "v[108:111]") "v_add_f32 v132, 1.0, v132\n\tv_add_f32 v133, 1.0, v133\n\t" \
                     MF3("v[108:111]", "%[z11]", "v[154:157]", "v[108:111]") "v_rcp_f32 v130, v130\n\t"                            \
                     MF3("v[108:111]", "%[z02]", "v[150:153]", "v[108:111]") "v_rcp_f32 v131, v131\n\t"                            \
                     MF3("v[108:111]", "%[z12]", "v[154:157]", "v[108:111]") "v_rcp_f32 v132, v132\n\t"                            \
                     MF3("v[108:111]", "%[z00]", "v[158:161]", "v[108:111]") "v_rcp_f32 v133, v133\n\t"                            \
                     MF3("v[108:111]", "%[z10]", "v[162:165]", "v[108:111]") FMA("v130", "v130", "-2.0", "1.0") FMA("v131", "v131", "-2.0", "1.0") \
                     MF3("v[108:111]", "%[z01]", "v[158:161]", "v[108:111]") FMA("v132", "v132", "-2.0", "1.0") FMA("v133", "v133", "-2.0", "1.0") \
                     MF3("v[108:111]", "%[z11]", "v[162:165]", "v[108:111]")                                                       \
                     "v_sub_f32 v138, v124, v130\n\tv_sub_f32 v139, v125, v131\n\tv_sub_f32 v140, v126, v132\n\tv_sub_f32 v141, v127, v133\n\t" \
                     "s_nop 1\n\t"                                                                                                \
                     "v_exp_f32 v134, v108\n\tv_exp_f32 v135, v109\n\tv_exp_f32 v136, v110\n\tv_exp_f32 v137, v111\n\t"            \
                     "v_pk_add_f32 v[134:135], v[134:135], 1.0 op_sel_hi:[1,0]\n\t"                                                \
                     "v_pk_add_f32 v[136:137], v[136:137], 1.0 op_sel_hi:[1,0]\n\t"                                                \
                     "v_rcp_f32 v134, v134\n\tv_rcp_f32 v135, v135\n\tv_rcp_f32 v136, v136\n\tv_rcp_f32 v137, v137\n\t"            \
                     "v_pk_fma_f32 v[124:125], v[134:135], v[138:139], v[130:131]\n\t"                                             \
                     "v_pk_fma_f32 v[126:127], v[136:137], v[140:141], v[132:133]\n\t"                                             \
                     "v_cvt_pk_bf16_f32 v128, v124, v125\n\tv_cvt_pk_bf16_f32 v129, v126, v127\n\t"                                \
                     "ds_write_b64 %[wr], v[128:129] offset:" O_WH                                                                \
                     : "+{v[124:127]}"(h4), "=&{v[128:129]}"(p0v), "=&{v[108:111]}"(acc3[2]), "+{v[100:103]}"(acc3[0])              \
                     : "{v[104:107]}"(acc3[1]), "{" SZC "}"(seed_z[cur]), "{" GIC "}"(gi4[cur]), [wr] "v"(wr_addr),                 \
                       "{v[142:145]}"(B3[0]), "{v[146:149]}"(B3[1]),                                                              \
                       "{v[150:153]}"(B3[2]), "{v[154:157]}"(B3[3]), "{v[158:161]}"(B3[4]), "{v[162:165]}"(B3[5]),                 \
                       [z00] "v"(A3[2][0][0]), [z10] "v"(A3[2][1][0]), [z01] "v"(A3[2][0][1]), [z11] "v"(A3[2][1][1]),             \
                       [z02] "v"(A3[2][0][2]), [z12] "v"(A3[2][1][2])                                                             \
                     : "memory", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141")
        if constexpr (cur == 0) NTM3_S4("6144", "v[170:173]", "v[174:177]", "v174", "v175", "v176", "v177");
        else NTM3_S4("0", "v[210:213]", "v[206:209]", "v206", "v207", "v208", "v209");
#undef NTM3_S4
#undef FMA
#undef NEG
#undef MF3
        NTM2_STAMP(5)
        if constexpr (STAMP) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            seg[0] += ts_[0] - last_;
#pragma unroll
            for (int k = 1; k < 6; ++k) seg[k] += ts_[k] - ts_[k - 1];
            last_ = ts_[5];
        }
#else
        // ---- the compiler-scheduled form (NTM3_ASM = 0; libntm_bf16x3c.so of `make exp`): the same arithmetic, for the A/B
        constexpr int cur = decltype(cur_c)::value;   // == t & 1: which exchange buffer holds h_{t-1}
        const int ph = (int)(t & 63);
        const int64_t tile = t >> 6;
        // barrier 1: every wave's hi piece of h_{t-1} has landed (its own writes: all retired, lgkmcnt(0))
        NTM2_STAMP(0)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        NTM2_STAMP(1)
        const bf16x8 Bh0 = *(const bf16x8 *)(h3rd + cur * HB3);
        const bf16x8 Bh1 = *(const bf16x8 *)(h3rd + cur * HB3 + 3 * 256);
        // ---- in the shadow of that round trip: the mid and lo pieces of this wave's quarter, published; the head partial of
        //      y_{t-1} (t = 0: h_0's, into a slot that sample 127 rewrites before any flush reads it); tile housekeeping
        {
            bf16x2 m_[2], l_[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const f32x2 r1 = hold[p] - __builtin_convertvector(h3hi[p], f32x2);
                m_[p] = __builtin_convertvector(r1, bf16x2);
                const f32x2 r2 = r1 - __builtin_convertvector(m_[p], f32x2);
                l_[p] = __builtin_convertvector(r2, bf16x2);
            }
            *(bf16x4 *)(h3wr + cur * HB3 + 256) = (bf16x4){m_[0][0], m_[0][1], m_[1][0], m_[1][1]};
            *(bf16x4 *)(h3wr + cur * HB3 + 512) = (bf16x4){l_[0][0], l_[0][1], l_[1][0], l_[1][1]};
        }
        park_head(t - 1);
        housekeeping(t, ph, tile, hk_c);
        NTM2_STAMP(2)

        // ---- the 18 MFMAs on the hi pieces (W_1, W_2, W_3 . h_1); behind the first six: barrier 2 (every wave's mid / lo
        //      pieces have landed), their reads and x of step t+1
        f32x4 acc_r = {cr[0][0], cr[0][1], cr[1][0], cr[1][1]};
        f32x4 acc_n = {bhn[0][0], bhn[0][1], bhn[1][0], bhn[1][1]};
        f32x4 acc_z = {cz[0][0], cz[0][1], cz[1][0], cz[1][1]};
        NTM3_MF(acc_r, 0, 0, 0, Bh0); NTM3_MF(acc_n, 1, 0, 0, Bh0); NTM3_MF(acc_z, 2, 0, 0, Bh0);
        NTM3_MF(acc_r, 0, 1, 0, Bh1); NTM3_MF(acc_n, 1, 1, 0, Bh1); NTM3_MF(acc_z, 2, 1, 0, Bh1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+v"(acc_r), "+v"(acc_n), "+v"(acc_z)::"memory");
        const bf16x8 Bm0 = *(const bf16x8 *)(h3rd + cur * HB3 + 1 * 256);
        const bf16x8 Bm1 = *(const bf16x8 *)(h3rd + cur * HB3 + 4 * 256);
        const bf16x8 Bl0 = *(const bf16x8 *)(h3rd + cur * HB3 + 2 * 256);
        const bf16x8 Bl1 = *(const bf16x8 *)(h3rd + cur * HB3 + 5 * 256);
        float xn = xb[(((t + 1) >> 6) & 1) * SG * XS + j * XS + (int)((t + 1) & 63)];
#pragma unroll
        for (int p = 1; p < 3; ++p) {
            NTM3_MF(acc_r, 0, 0, p, Bh0); NTM3_MF(acc_n, 1, 0, p, Bh0); NTM3_MF(acc_z, 2, 0, p, Bh0);
            NTM3_MF(acc_r, 0, 1, p, Bh1); NTM3_MF(acc_n, 1, 1, p, Bh1); NTM3_MF(acc_z, 2, 1, p, Bh1);
        }
        asm volatile("" : "+v"(acc_r), "+v"(acc_n), "+v"(acc_z), "+v"(xn));
        NTM2_STAMP(3)

        // ---- the 30 MFMAs on the mid and lo pieces, chain after chain; the vector pipe works in their gaps (one op per gap:
        //      a bf16 MFMA holds the vector issue for 8 of its 16 cycles).  Each `asm` pins one gap: the MFMA and the vector op
        //      in front of it are issued there, their consumers behind it.
        // r chain; in its gaps the input terms of step t+1
        const f32x2 xx = {xn, xn};
        f32x2 ncr[2], ncz[2], ngi[2];
        NTM3_MF(acc_r, 0, 0, 0, Bm0); NTM3_MF(acc_r, 0, 1, 0, Bm1);
        NTM3_MF(acc_r, 0, 0, 1, Bm0); NTM3_MF(acc_r, 0, 1, 1, Bm1);
        NTM3_MF(acc_r, 0, 0, 2, Bm0);
        ncr[0] = __builtin_elementwise_fma(wir[0], xx, br[0]);
        asm volatile("" : "+v"(acc_r), "+v"(ncr[0]));
        NTM3_MF(acc_r, 0, 1, 2, Bm1);
        ncr[1] = __builtin_elementwise_fma(wir[1], xx, br[1]);
        asm volatile("" : "+v"(acc_r), "+v"(ncr[1]));
        NTM3_MF(acc_r, 0, 0, 0, Bl0);
        ncz[0] = __builtin_elementwise_fma(wiz[0], xx, bz[0]);
        asm volatile("" : "+v"(acc_r), "+v"(ncz[0]));
        NTM3_MF(acc_r, 0, 1, 0, Bl1);
        ncz[1] = __builtin_elementwise_fma(wiz[1], xx, bz[1]);
        asm volatile("" : "+v"(acc_r), "+v"(ncz[1]));
        NTM3_MF(acc_r, 0, 0, 1, Bl0);
        ngi[0] = __builtin_elementwise_fma(win[0], xx, bin_[0]);
        asm volatile("" : "+v"(acc_r), "+v"(ngi[0]));
        NTM3_MF(acc_r, 0, 1, 1, Bl1);
        ngi[1] = __builtin_elementwise_fma(win[1], xx, bin_[1]);
        asm volatile("" : "+v"(acc_r), "+v"(ngi[1]), "+v"(acc_n));
        if (!PRESCALE) acc_r *= -LOG2E;
        // n chain; in its gaps the r sigmoid: 4 x v_exp, 2 x v_pk_add, 4 x v_rcp
        const f32x2 one = {1.0f, 1.0f};
        float e0, e1, e2, e3;
        NTM3_MF(acc_n, 1, 0, 0, Bm0); e0 = __builtin_amdgcn_exp2f(acc_r[0]); asm volatile("" : "+v"(acc_n), "+v"(e0), "+v"(acc_r));
        NTM3_MF(acc_n, 1, 1, 0, Bm1); e1 = __builtin_amdgcn_exp2f(acc_r[1]); asm volatile("" : "+v"(acc_n), "+v"(e1), "+v"(acc_r));
        NTM3_MF(acc_n, 1, 0, 1, Bm0); e2 = __builtin_amdgcn_exp2f(acc_r[2]); asm volatile("" : "+v"(acc_n), "+v"(e2), "+v"(acc_r));
        NTM3_MF(acc_n, 1, 1, 1, Bm1); e3 = __builtin_amdgcn_exp2f(acc_r[3]); asm volatile("" : "+v"(acc_n), "+v"(e3));
        f32x2 s01 = {e0, e1}, s23 = {e2, e3};
        NTM3_MF(acc_n, 1, 0, 2, Bm0); s01 += one; asm volatile("" : "+v"(acc_n), "+v"(s01), "+v"(s23));
        NTM3_MF(acc_n, 1, 1, 2, Bm1); s23 += one; asm volatile("" : "+v"(acc_n), "+v"(s23), "+v"(s01));
        float r0, r1, r2, r3;
        NTM3_MF(acc_n, 1, 0, 0, Bl0); r0 = __builtin_amdgcn_rcpf(s01[0]); asm volatile("" : "+v"(acc_n), "+v"(r0), "+v"(s01), "+v"(s23));
        NTM3_MF(acc_n, 1, 1, 0, Bl1); r1 = __builtin_amdgcn_rcpf(s01[1]); asm volatile("" : "+v"(acc_n), "+v"(r1), "+v"(s23));
        NTM3_MF(acc_n, 1, 0, 1, Bl0); r2 = __builtin_amdgcn_rcpf(s23[0]); asm volatile("" : "+v"(acc_n), "+v"(r2), "+v"(s23));
        NTM3_MF(acc_n, 1, 1, 1, Bl1); r3 = __builtin_amdgcn_rcpf(s23[1]); asm volatile("" : "+v"(acc_n), "+v"(r3), "+v"(acc_z));
        const f32x2 rr[2] = {{r0, r1}, {r2, r3}};
        // z chain; in its gaps the n gate: tanh(gi_n + r gh_n) = 1 - 2 / (1 + 2^(2 log2e (...)))
        f32x2 pn0, pn1;
        NTM3_MF(acc_z, 2, 0, 0, Bm0);
        pn0 = __builtin_elementwise_fma(rr[0], (f32x2){acc_n[0], acc_n[1]}, gi[0]);
        pn1 = __builtin_elementwise_fma(rr[1], (f32x2){acc_n[2], acc_n[3]}, gi[1]);
        if (!PRESCALE) { pn0 *= 2.0f * LOG2E; pn1 *= 2.0f * LOG2E; }
        asm volatile("" : "+v"(acc_z), "+v"(pn0), "+v"(pn1));
        float f0, f1, f2, f3;
        NTM3_MF(acc_z, 2, 1, 0, Bm1); f0 = __builtin_amdgcn_exp2f(pn0[0]); asm volatile("" : "+v"(acc_z), "+v"(f0), "+v"(pn0), "+v"(pn1));
        NTM3_MF(acc_z, 2, 0, 1, Bm0); f1 = __builtin_amdgcn_exp2f(pn0[1]); asm volatile("" : "+v"(acc_z), "+v"(f1), "+v"(pn1));
        NTM3_MF(acc_z, 2, 1, 1, Bm1); f2 = __builtin_amdgcn_exp2f(pn1[0]); asm volatile("" : "+v"(acc_z), "+v"(f2), "+v"(pn1));
        NTM3_MF(acc_z, 2, 0, 2, Bm0); f3 = __builtin_amdgcn_exp2f(pn1[1]); asm volatile("" : "+v"(acc_z), "+v"(f3));
        f32x2 u01 = {f0, f1}, u23 = {f2, f3};
        NTM3_MF(acc_z, 2, 1, 2, Bm1); u01 += one; u23 += one; asm volatile("" : "+v"(acc_z), "+v"(u01), "+v"(u23));
        float g0, g1, g2, g3;
        NTM3_MF(acc_z, 2, 0, 0, Bl0); g0 = __builtin_amdgcn_rcpf(u01[0]); asm volatile("" : "+v"(acc_z), "+v"(g0), "+v"(u01), "+v"(u23));
        NTM3_MF(acc_z, 2, 1, 0, Bl1); g1 = __builtin_amdgcn_rcpf(u01[1]); asm volatile("" : "+v"(acc_z), "+v"(g1), "+v"(u23));
        NTM3_MF(acc_z, 2, 0, 1, Bl0); g2 = __builtin_amdgcn_rcpf(u23[0]); asm volatile("" : "+v"(acc_z), "+v"(g2), "+v"(u23));
        NTM3_MF(acc_z, 2, 1, 1, Bl1); g3 = __builtin_amdgcn_rcpf(u23[1]); asm volatile("" : "+v"(acc_z), "+v"(g3));
        NTM2_STAMP(4)
        // while the last MFMAs drain: n = 1 - 2 / (1 + 2^...), h - n
        const f32x2 n0 = __builtin_elementwise_fma((f32x2){-2.0f, -2.0f}, (f32x2){g0, g1}, one);
        const f32x2 n1 = __builtin_elementwise_fma((f32x2){-2.0f, -2.0f}, (f32x2){g2, g3}, one);
        f32x2 d0 = hold[0] - n0, d1 = hold[1] - n1;
        asm volatile("" : "+v"(d0), "+v"(d1), "+v"(acc_z));
        // ---- the tail: z sigmoid, blend
        if (!PRESCALE) acc_z *= -LOG2E;
        f32x2 v01 = {__builtin_amdgcn_exp2f(acc_z[0]), __builtin_amdgcn_exp2f(acc_z[1])};
        f32x2 v23 = {__builtin_amdgcn_exp2f(acc_z[2]), __builtin_amdgcn_exp2f(acc_z[3])};
        v01 += one; v23 += one;
        const f32x2 z0 = {__builtin_amdgcn_rcpf(v01[0]), __builtin_amdgcn_rcpf(v01[1])};
        const f32x2 z1 = {__builtin_amdgcn_rcpf(v23[0]), __builtin_amdgcn_rcpf(v23[1])};
        hold[0] = __builtin_elementwise_fma(z0, d0, n0);               // n + z (h - n)
        hold[1] = __builtin_elementwise_fma(z1, d1, n1);
        // publish the hi piece of h_t (on the critical path of every wave's next step); the rest follows behind barrier 1
        h3hi[0] = __builtin_convertvector(hold[0], bf16x2);
        h3hi[1] = __builtin_convertvector(hold[1], bf16x2);
        *(bf16x4 *)(h3wr + (cur ^ 1) * HB3) = (bf16x4){h3hi[0][0], h3hi[0][1], h3hi[1][0], h3hi[1][1]};
#pragma unroll
        for (int p = 0; p < 2; ++p) { cr[p] = ncr[p]; cz[p] = ncz[p]; gi[p] = ngi[p]; }
        NTM2_STAMP(5)
        if constexpr (STAMP) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            seg[0] += ts_[0] - last_;
#pragma unroll
            for (int k = 1; k < 6; ++k) seg[k] += ts_[k] - ts_[k - 1];
            last_ = ts_[5];
        }
    #endif
    };
#undef NTM3_MF
    {
        using C0 = std::integral_constant<int, 0>;
        using C1 = std::integral_constant<int, 1>;
        using H0 = std::integral_constant<int, 0>;
        using HA = std::integral_constant<int, 1>;
        using HB = std::integral_constant<int, 2>;
        using HC = std::integral_constant<int, 3>;
        using HR = std::integral_constant<int, -1>;
        // (always_inline: the step bodies must stay part of this kernel's one basic-block chain whatever the wrapper costs)
        auto do_step = [&](const int64_t t, auto cur_c, auto hk_c) __attribute__((always_inline)) {
            if constexpr (ENGINE == 2) step_b(t, cur_c, hk_c);
            else step(t, cur_c, hk_c);
        };
        const int64_t full = (T / TT) * TT;
        for (int64_t t0 = 0; t0 < full; t0 += TT) {          // whole tiles: housekeeping at compile-time positions
            do_step(t0, C0{}, H0{}); do_step(t0 + 1, C1{}, H0{});
            do_step(t0 + 2, C0{}, HA{}); do_step(t0 + 3, C1{}, H0{});
            for (int p = 4; p < 34; p += 2) { do_step(t0 + p, C0{}, H0{}); do_step(t0 + p + 1, C1{}, H0{}); }
            do_step(t0 + 34, C0{}, HB{}); do_step(t0 + 35, C1{}, H0{});
            if constexpr (FUSE) {
                do_step(t0 + 36, C0{}, HC{}); do_step(t0 + 37, C1{}, H0{});
                for (int p = 38; p < TT; p += 2) { do_step(t0 + p, C0{}, H0{}); do_step(t0 + p + 1, C1{}, H0{}); }
            } else {
                for (int p = 36; p < TT; p += 2) { do_step(t0 + p, C0{}, H0{}); do_step(t0 + p + 1, C1{}, H0{}); }
            }
        }
        for (int64_t t = full; t < T; t += 2) {               // ragged last tile: phase tests at run time
            do_step(t, C0{}, HR{});
            if (t + 1 < T) do_step(t + 1, C1{}, HR{});
        }
    }
    if constexpr (STAMP) {
        if (a.dbg && l == 0)
            for (int k = 0; k < 12; ++k) a.dbg[((size_t)blockIdx.x * 4 + w) * 12 + k] = seg[k];
    }

    // ---- epilogue: remaining y tiles, final state ---------------------------------------------------
    if constexpr (ENGINE == 2) {
#if NTM3_ASM
        hold[0] = (f32x2){h4[0], h4[1]};
        hold[1] = (f32x2){h4[2], h4[3]};
#endif
        park_head(T - 1);      // (step_b parks the head partial of a sample one step later)
    }
    __syncthreads();
    while (next_flush * TT < T) { flush_y_tile(next_flush, std::false_type{}); ++next_flush; }
    if constexpr (FUSE) {
        if (dl_on) {
            // every pre_d tile is stored: make the stores visible to the workgroup, then run the delay tiles that are left
            // (the one in flight first) back to back -- at most two tiles plus the ragged tail per launch
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            int next_delay = 0;
            if (dl_stage == 1) dl_issue_taps(std::false_type{});
            if (dl_stage == 2) {
                next_delay = dl_tile + 1;
                dl_compute(std::false_type{});
                if constexpr (ESR_DL) esr_accumulate(dl_tile, dl_out, std::false_type{});
                dl_store();
            }
            for (; (int64_t)next_delay * TT < T; ++next_delay) {
                dl_load_d(next_delay, std::false_type{});
                dl_issue_taps(std::false_type{});
                dl_compute(std::false_type{});
                if constexpr (ESR_DL) esr_accumulate(next_delay, dl_out, std::false_type{});
                dl_store();
            }
            if (a.dl_flag && __any(dl_bad) && l == 0) atomicOr(a.dl_flag, 1);
        }
    }
    if constexpr (ESR) {
        // the 16 threads of a stream (one lane group of 16) add their columns in a fixed butterfly order
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            esr_e += __shfl_xor(esr_e, off, 16);
            esr_t += __shfl_xor(esr_t, off, 16);
        }
        if (y_row_ok && (tid & 15) == 0) {
            a.esr_out[(s0 + (tid >> 4)) * 2 + 0] = esr_e;
            a.esr_out[(s0 + (tid >> 4)) * 2 + 1] = esr_t;
        }
        if constexpr (DCP) {
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) {
                dcp_e += __shfl_xor(dcp_e, off, 16);
                dcp_t += __shfl_xor(dcp_t, off, 16);
            }
            if (y_row_ok && (tid & 15) == 0) {
                a.dcp_out[(s0 + (tid >> 4)) * 2 + 0] = dcp_e;
                a.dcp_out[(s0 + (tid >> 4)) * 2 + 1] = dcp_t;
            }
        }
    }

    if (a.h_state && valid) {
#pragma unroll
        for (int v = 0; v < 4; ++v) a.h_state[(s0 + j) * kH + 16 * w + 4 * q + v] = hold[v >> 1][v & 1];
    }
}

// The dynamic-LDS attribute is per (kernel, device): set once for each and remembered (one process may drive several
// devices; concurrent first calls at worst set it twice).  Real-time style callers issue thousands of short launches.
template <typename K, K kernel>
static hipError_t launch_m2(size_t smem_bytes, unsigned grid, const GruArgs &a, hipStream_t stream)
{
    static std::atomic<uint64_t> configured{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = dev < 64 ? (uint64_t)1 << dev : 0;
    if (!(configured.load(std::memory_order_relaxed) & bit) || !bit) {
        e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_bytes);
        if (e != hipSuccess) return e;
        configured.fetch_or(bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), smem_bytes, stream, a);
    return hipGetLastError();
}
#define NTM2_LAUNCH(KERNEL, SMEM) launch_m2<decltype(&KERNEL), &KERNEL>(SMEM, grid, a, stream)

hipError_t launch_gru_mfma2(const GruArgs &a, hipStream_t stream)
{
    constexpr size_t smem16 = m2::smem_floats(16) * sizeof(float);   // 151 680 B: one workgroup per CU
    constexpr size_t smem4 = m2::smem_floats(4) * sizeof(float);     //  53 376 B: up to three per CU
    static_assert(smem16 <= 160 * 1024, "LDS carve-up");
    const unsigned grid = (unsigned)((a.B + m2::SG - 1) / m2::SG);
    // More stream groups than CUs: the small-LDS build lets two or three groups share a CU instead of running a
    // second round of workgroups (B = 6144: 7.1 ms instead of 7.7 per 4096 steps; B >= 8192: 0.76-0.80 of peak).
    const bool many = grid > (unsigned)device_cus();
#ifdef NTM_LAB
    // libntm_lab.so only: the diagnostic instantiations (s_memtime stamps, timing ablations) behind
    // ntm_debug_gru_stamps / ntm_debug_gru_ablate -- never compiled into the product library
#define NTM2_ABL_CASE(M) case M: return NTM2_LAUNCH((gru_mfma2_kernel<true, false, M>), smem16);
    switch (a.abl) {
        NTM2_ABL_CASE(1) NTM2_ABL_CASE(2) NTM2_ABL_CASE(4) NTM2_ABL_CASE(8) NTM2_ABL_CASE(16) NTM2_ABL_CASE(32)
        NTM2_ABL_CASE(3) NTM2_ABL_CASE(7) NTM2_ABL_CASE(18) NTM2_ABL_CASE(39) NTM2_ABL_CASE(55) NTM2_ABL_CASE(63)
        default: break;
    }
    if (a.dbg && a.engine == 2) return NTM2_LAUNCH((gru_mfma2_kernel<true, true, 0, 2, 16>), m2::smem_floats(16, 2) * sizeof(float));
    if (a.dbg && a.tgt) return NTM2_LAUNCH((gru_mfma2_kernel<true, true, 0, 0, 16, false, true>), smem16);
    if (a.dbg) return NTM2_LAUNCH((gru_mfma2_kernel<true, true>), smem16);
#else
    if (a.abl || a.dbg) return hipErrorInvalidValue;      // diagnostics live in libntm_lab.so
#endif
    constexpr size_t smem16b = m2::smem_floats(16, 2) * sizeof(float);   // ENGINE 2: 2 KB more exchange buffer
    constexpr size_t smem4b = m2::smem_floats(4, 2) * sizeof(float);
    static_assert(smem16b <= 160 * 1024, "LDS carve-up");
    if (a.tgt) {        // predict + ESR sums in one launch (exact fp32 engine)
        if (!a.esr_out || a.engine || (a.esr_skip & 3) || a.esr_skip < 0) return hipErrorInvalidValue;
        if (a.dcp_out)  // ... + the DCPreESR sums
            return many ? NTM2_LAUNCH((gru_mfma2_kernel<true, false, 0, 0, 4, false, true, true>), smem4)
                        : NTM2_LAUNCH((gru_mfma2_kernel<true, false, 0, 0, 16, false, true, true>), smem16);
        return many ? NTM2_LAUNCH((gru_mfma2_kernel<true, false, 0, 0, 4, false, true>), smem4)
                    : NTM2_LAUNCH((gru_mfma2_kernel<true, false, 0, 0, 16, false, true>), smem16);
    }
    if (a.engine == 2)
        return many ? NTM2_LAUNCH((gru_mfma2_kernel<true, false, 0, 2, 4>), smem4b)
                    : NTM2_LAUNCH((gru_mfma2_kernel<true, false, 0, 2, 16>), smem16b);
    if (a.engine == 1)
        return many ? NTM2_LAUNCH((gru_mfma2_kernel<true, false, 0, 1, 4>), smem4)
                    : NTM2_LAUNCH((gru_mfma2_kernel<true, false, 0, 1, 16>), smem16);
    return many ? NTM2_LAUNCH((gru_mfma2_kernel<true, false, 0, 0, 4>), smem4)
                : NTM2_LAUNCH((gru_mfma2_kernel<true, false, 0, 0, 16>), smem16);
}

// The DiffDelRNN step in one launch (exact fp32 engine).  The carried delay buffer is moved on by delay_update_kernel
// (aux_kernels.hip) behind this launch.
hipError_t launch_gru_mfma2_fused(const GruArgs &a, hipStream_t stream)
{
    constexpr size_t smem16 = m2::smem_floats(16) * sizeof(float);
    constexpr size_t smem4 = m2::smem_floats(4) * sizeof(float);
    if (!a.dd || !a.yd || (a.D > 0 && !a.dl_buf) || a.abl || a.dbg || a.engine) return hipErrorInvalidValue;
    if (a.T >= (1LL << 26) || a.ys != a.T) return hipErrorInvalidValue;   // contiguous rows; 32-bit BYTE offsets inside a 16-row block
    const unsigned grid = (unsigned)((a.B + m2::SG - 1) / m2::SG);
    const bool many = grid > (unsigned)device_cus();
    if (a.tgt) {        // + the ESR sums of the delayed output against a target, in the fused delay stage
        if (!a.esr_out || (a.esr_skip & 3) || a.esr_skip < 0 || a.warmup) return hipErrorInvalidValue;
        if (a.dcp_out)  // ... + its DCPreESR sums
            return many ? NTM2_LAUNCH((gru_mfma2_kernel<true, false, 0, 0, 4, true, true, true>), smem4)
                        : NTM2_LAUNCH((gru_mfma2_kernel<true, false, 0, 0, 16, true, true, true>), smem16);
        return many ? NTM2_LAUNCH((gru_mfma2_kernel<true, false, 0, 0, 4, true, true>), smem4)
                    : NTM2_LAUNCH((gru_mfma2_kernel<true, false, 0, 0, 16, true, true>), smem16);
    }
    return many ? NTM2_LAUNCH((gru_mfma2_kernel<true, false, 0, 0, 4, true>), smem4)
                : NTM2_LAUNCH((gru_mfma2_kernel<true, false, 0, 0, 16, true>), smem16);
}

#ifdef NTM_LAB
hipError_t launch_debug_transpose(const float *in, float *out, hipStream_t stream)
{
    hipLaunchKernelGGL(debug_transpose_kernel, dim3(1), dim3(256), 0, stream, in, out);
    return hipGetLastError();
}
#endif

}  // namespace ntm
