// K4: builder-defined causal dilated-Conv1d TCN (BASELINE configs[3]; the reference has no TCN).
//   block:  out[n][co] = PReLU(b[co] + sum_{ci,k} W[ci][k][co] in[n-(K-1-k)dil][ci]) + sum_ci R[ci][co] in[n][ci]
// Activations between blocks are channels-last [B][T][32].  Blocks with 32 input channels run on the matrix
// pipe (exact fp32 v_mfma_f32_16x16x4_f32): M = 16 output channels (A = weights, resident in VGPRs: 112 per
// lane), N = 16 samples (B = input rows from LDS), K = 32x13 taps + 32 residual; the 1-channel first block and
// the 1x1 output conv are plain VALU kernels (< 4 % of the flops).
//
// History (MI355X, 512 x 65 536): a first MFMA version that fetched B operands straight from global memory
// in natural sample order ran 12.7 ms per block with the matrix pipe 50 % busy: every input row was fetched
// 13 x (taps) x 2 (waves sharing a sample range) = 26 times through L1, for every dilation alike.  The
// polyphase tile order + LDS staging below fetches each row once per tile: 8.3 ms per block.
#include "ntm_common.h"

#include <cstdlib>
#include <type_traits>

namespace ntm {

__device__ float tcn_zeros[16 * 32];   // zero page (one row block) for rows before the start of a stream / of no tile

#ifdef NTM_LAB
// DIAGNOSTIC build (libntm_lab.so only, never timed as product): tcn_block_pg_kernel<false> takes s_memtime at six points
// of every iteration and one wave leaves the per-segment sums here (ntm_lab_tcn_stamps, tools/attic/tcn_stamp_probe.py).  The
// wait sits inside the asm: s_memtime returns asynchronously and would otherwise land in a register pair the compiler
// has already given to something else.
__device__ unsigned long long tcn_stamp_out[8 + 3 * 64];   // [0..5] segment sums, [6] iterations, then per iteration: start, MFMA block, total
__device__ unsigned long long *tcn_trace_buf;       // optional: 4 words per workgroup (start, end, XCC | HW_ID, cycles in the loop)
#define TCN_STAMP(k)                                                                                   \
    if constexpr (!FUSE_OUT) {                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_[k])::"memory");                 \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    }
#else
#define TCN_STAMP(k)
#endif

constexpr int TCN_PAD_FLOATS = 16 * 32;   // behind each activation buffer: a row block may reach 15 rows past T (read, never used)

constexpr int TC = 32;    // channels
constexpr int TK = 13;    // kernel size

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

// ---- first block, dilation 1 (the frozen spec) ---------------------------------------------------------------
// A workgroup walks FC chunks of 256 consecutive samples of one stream; thread -> (8 consecutive samples, group of 4
// output channels): a wave writes 8 KiB contiguous per chunk.  The 8 samples share a window of 20 inputs; a chunk's
// 256 samples + 12 of history go through LDS (double-buffered: one coalesced, branch-free load per thread -- index
// clamped into the stream, zero selected outside -- issued a chunk ahead, one barrier per chunk), the window is five
// ds_read_b128.  The 13 weight vectors are loaded once per workgroup and stay in registers.  (History: with the
// window loaded per thread every load sat behind its own bounds branch, and every 256-sample workgroup re-loaded the
// weights: 8.9 ms for the 34 GB of output; a pure store of that array takes 5.9 ms, tools/ubench/store_pattern.hip.)
// Same operation order per output as the generic kernel (bit-identical results).
constexpr int FC = 8;                          // chunks per workgroup

__global__ __launch_bounds__(256) void tcn_first_d1_kernel(const float *x, float *out, const float *W, const float *bias,
                                                           const float *alpha, const float *R, int64_t T)
{
    constexpr int S = 8, NB = 32 * S;         // samples per thread and chunk / per chunk
    const int64_t b = blockIdx.x;
    const int c4 = threadIdx.x & 7, g = threadIdx.x >> 3;
    const int64_t first = (int64_t)blockIdx.y * (FC * NB);
    const float *xb = x + b * T;
    __shared__ __attribute__((aligned(16))) float xs[2][NB + 16];
    auto chunk_load = [&](int64_t base, int i) -> float {        // element i of the chunk's window [base - 12, base + 256)
        const int64_t src = base - (TK - 1) + i;
        const int64_t sc = src < 0 ? 0 : (src > T - 1 ? T - 1 : src);
        const float v = xb[sc];
        return sc == src ? v : 0.0f;
    };
    const int i1 = 256 + (threadIdx.x & 15);                     // second element (only NB + 12 - 256 = 12 are needed; 16 stored)
    float va = chunk_load(first, threadIdx.x), vb = chunk_load(first, i1 < NB + TK - 1 ? i1 : NB + TK - 2);
    f32x4 wv[TK];
#pragma unroll
    for (int k = 0; k < TK; ++k) wv[k] = *(const f32x4 *)(W + k * TC + 4 * c4);
    const f32x4 bi = *(const f32x4 *)(bias + 4 * c4), al = *(const f32x4 *)(alpha + 4 * c4), rv = *(const f32x4 *)(R + 4 * c4);
    for (int c = 0; c < FC; ++c) {
        const int64_t base = first + (int64_t)c * NB;
        if (base >= T) break;
        float *xc = xs[c & 1];
        xc[threadIdx.x] = va;
        if (threadIdx.x < 16) xc[i1 < NB + TK - 1 ? i1 : NB + TK - 2] = vb;
        __syncthreads();                       // (the buffer written two chunks ago was last read before the previous barrier)
        if (c + 1 < FC && base + NB < T) {     // next chunk's window, in flight during this chunk's arithmetic
            va = chunk_load(base + NB, threadIdx.x);
            vb = chunk_load(base + NB, i1 < NB + TK - 1 ? i1 : NB + TK - 2);
        }
        const int64_t n0 = base + g * S;
        float xw[S + TK - 1];
#pragma unroll
        for (int i = 0; i < (S + TK - 1) / 4; ++i) {
            const f32x4 v = *(const f32x4 *)&xc[g * S + 4 * i];
#pragma unroll
            for (int e = 0; e < 4; ++e) xw[4 * i + e] = v[e];
        }
        float *ob = out + (b * T + n0) * TC + 4 * c4;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            f32x4 acc = bi;
#pragma unroll
            for (int k = 0; k < TK; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(wv[k][e], xw[s + k], acc[e]);
            const float x0 = xw[s + TK - 1];
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(rv[e], x0, acc[e] >= 0.0f ? acc[e] : al[e] * acc[e]);
            if (n0 + s < T) *(f32x4 *)(ob + (int64_t)s * TC) = v;
        }
    }
}

// PReLU(u) + r without compare / select pairs: hi = max(u, 0), lo = u - hi (exact: one of the two is u, the other 0),
// alpha lo + hi is then u for u >= 0 and alpha u (one rounding, like the product in the select form) for u < 0 -- the
// same values as `u >= 0 ? u : alpha u`.  v_max_f32 through asm: the builtin adds a canonicalising v_max per element.
__device__ __forceinline__ f32x4 prelu_plus(const f32x4 u, const f32x4 alpha, const f32x4 r)
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x4 o;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x2 uu = {u[2 * h], u[2 * h + 1]}, aa = {alpha[2 * h], alpha[2 * h + 1]}, rr = {r[2 * h], r[2 * h + 1]};
        f32x2 hi;
        asm("v_max_f32 %0, 0, %1" : "=v"(hi[0]) : "v"(uu[0]));
        asm("v_max_f32 %0, 0, %1" : "=v"(hi[1]) : "v"(uu[1]));
        f32x2 lo;                                                    // u - hi in ONE packed op (hipcc emits two v_sub_f32)
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"(uu), "v"(hi));
        const f32x2 t = __builtin_elementwise_fma(aa, lo, hi) + rr;
        o[2 * h] = t[0]; o[2 * h + 1] = t[1];
    }
    return o;
}

// lane partial of the 1x1 output conv over the lane's 4 channels, on packed ops: (w0 v0 + w2 v2) + (w1 v1 + w3 v3)
__device__ __forceinline__ float dot4(const f32x4 wv, const f32x4 v)
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 t = __builtin_elementwise_fma((f32x2){wv[2], wv[3]}, (f32x2){v[2], v[3]}, (f32x2){wv[0], wv[1]} * (f32x2){v[0], v[1]});
    return t[0] + t[1];
}

// p(l) + p(l ^ 16), then that + the same of lane l ^ 32: the sum over the wave's four 16-lane groups, in every lane.
// v_permlane*_swap exchanges halves of TWO registers in place; with a copy of p as the second one the two results add
// up to p(l) + p(l ^ 32) resp. p(l) + p(l ^ 16) (as in gru_mfma2.hip; asm because hipcc folds the builtin's two results
// when both operands hold the same value; the wait states around the swap are inside the string).
__device__ __forceinline__ float sum_lane_groups(float p)
{
    float t;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(p), "=&v"(t));
    p += t;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(p), "=&v"(t));
    return p + t;
}

// ---- 32 -> 32 channel block, polyphase tile order + LDS staging (dilations below 512) ----------------
// A tile is 16 outputs of ONE phase of the dilation: n = p + (m0 + j) dil, j = 0..15.  Its 13 taps read the
// rows m0 + j + k - 12 of the same phase, so consecutive taps reuse the same 28 input rows: they are staged
// once in LDS (register-staged one iteration ahead) and the B operands of all 13 x 8 K-steps are
// ds_read_b128 from there -- every input row is fetched from global memory once per tile (1.75x instead of
// 26x), for every dilation alike.  Tiles of a stream are numbered tau = p * tpp + m0/16 (tpp = tiles per
// phase); a workgroup takes TPW consecutive tiles, 8 per iteration: wave pair ng = w>>1 owns 4 of them,
// wave mt = w&1 of the pair computes output channels 16 mt .. +15.
constexpr int TROWS = 28;                  // 16 outputs + 12 taps of history
constexpr int TRS = 36;                    // floats per LDS row: 32 channels + 4 pad (b128 reads conflict-light)
constexpr int TILE_F = TROWS * TRS;        // 1008 floats per tile window
constexpr int TPW = 512;                   // tiles per workgroup
constexpr int TCN2_SMEM_FLOATS = 2 * 2 * 4 * TILE_F;   // [buffer][pair][tile]  = 64 512 B

// FUSE_OUT (the last block): the 1x1 output conv y[n] = ob + sum_c ow[c] act[n][c] is applied to the tile while it
// is still in registers -- lane partial over its 4 channels, two cross-lane adds over the 4 lane groups, the
// two waves of a pair (channel halves) meet through LDS behind the iteration's barrier -- and the [B][T][32]
// activation of the last block is never written or read back (2 x 34 GB at 4096 x 65 536).
template <bool FUSE_OUT>
__global__ __launch_bounds__(256, 2) void tcn_block_mfma2_kernel(const float *in, float *out, const float *W,
                                                                 const float *bias, const float *alpha,
                                                                 const float *R, int dil, int64_t T64, int tpp,
                                                                 int total_tiles, const float *ow, const float *obias,
                                                                 float *yout)
{
    extern __shared__ __attribute__((aligned(16))) float tsm[];
    __shared__ float ypp[2][2][4][16];        // FUSE_OUT: [iteration parity][pair][tile][sample] partial of wave mt = 1
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = w & 1, ng = w >> 1;
    const int q = l >> 4, j = l & 15;
    const int64_t b = blockIdx.x;
    const int T = (int)T64;                   // launch_tcn: T < 2^31 - 2^25, dil <= 2^20 (32-bit sample arithmetic below)
    const float *ib = in + b * T64 * TC;
    float *ob = out + b * T64 * TC;
    const int tile0 = blockIdx.y * TPW;
    const int tiles_end = (tile0 + TPW < total_tiles) ? tile0 + TPW : total_tiles;
    const int niter = (tiles_end - tile0 + 7) / 8;

    float Aw[TK][8], Ar[8];
#pragma unroll
    for (int k = 0; k < TK; ++k)
#pragma unroll
        for (int s = 0; s < 8; ++s) Aw[k][s] = W[((8 * q + s) * TK + k) * TC + 16 * mt + j];
#pragma unroll
    for (int s = 0; s < 8; ++s) Ar[s] = R[(8 * q + s) * TC + 16 * mt + j];
    f32x4 bi, al, owv = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int v = 0; v < 4; ++v) { bi[v] = bias[16 * mt + 4 * q + v]; al[v] = alpha[16 * mt + 4 * q + v]; }
    if constexpr (FUSE_OUT) {
#pragma unroll
        for (int v = 0; v < 4; ++v) owv[v] = ow[16 * mt + 4 * q + v];
    }
    float ypart[4] = {0.0f, 0.0f, 0.0f, 0.0f};      // FUSE_OUT, wave mt = 0: own half of the last iteration's outputs
    int yn[4] = {T, T, T, T};
    float *yb = FUSE_OUT ? yout + b * T64 : nullptr;
    const float ob0 = FUSE_OUT ? obias[0] : 0.0f;
    auto finish_y = [&](int parity) {              // after the barrier: wave mt = 0 adds the other half and stores
        if constexpr (FUSE_OUT) {
            if (mt == 0 && q == 0) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    if (yn[nt] < T) *(float *)((char *)yb + 4u * (unsigned)yn[nt]) = (ypart[nt] + ypp[parity][ng][nt][j]) + ob0;
            }
        }
    };

    // Staging.  A tile's window = 12 history rows + 16 output rows, 8 pieces of 16 B each; the pair's 128 lanes take it
    // in two rounds: H = the 96 history pieces (lanes 96..127 repeat piece 95), O = the 128 output-row pieces.  The
    // tile is the same for every lane of a load, so everything that depends on the tile is SCALAR -- the window base
    // (or the zero page: history of the first tile of a phase, tiles past the end), and the number of samples left
    // up to T -- and the per-lane part is a constant 32-bit offset plus one compare/select for rows >= T: 3 vector
    // instructions per tile and no branch.  (The first version spread the 896 pieces of four tiles over 7 loads per
    // lane and selected tile, base and validity per lane: 84 vector instructions and a dozen branches per iteration,
    // 5 % of the launch -- vector instructions of either wave on a SIMD take matrix-pipe cycles, see gru_mfma2.hip.)
    const int e0 = mt * 64 + l;
    const int pH = e0 < 96 ? e0 : 95;
    const int rH = pH >> 3, cH = pH & 7, rO = e0 >> 3, cO = e0 & 7;
    const unsigned offH = (unsigned)(rH * dil * TC + 4 * cH) * 4u;                   // bytes from the window's first row
    const unsigned offO = (unsigned)(rO * dil * TC + 4 * cO) * 4u;
    const int ldsH = ng * 4 * TILE_F + rH * TRS + 4 * cH, ldsO = ng * 4 * TILE_F + (TK - 1 + rO) * TRS + 4 * cO;   // + nt * TILE_F
    const unsigned offS = (unsigned)(j * dil * TC + 16 * mt + 4 * q) * 4u;                                         // the lane's output row
    const int jdS = j * dil;
    const unsigned hist_bytes = (unsigned)(TK - 1) * (unsigned)dil * TC * 4u;

    // Tile tau = p * tpp + t (phase p, tile t of the phase).  The pair's four tiles advance by 8 per iteration; (p, t)
    // are carried as wave-uniform counters: t += 8 mod tpp with carry into p, as compare + selects (any tpp).
    struct TilePos { int tau, p, t; };
    const int adv_p = 8 / tpp, adv_t = 8 - adv_p * tpp;
    auto pos_init = [&](int tau) { TilePos r; r.tau = tau; r.p = tau / tpp; r.t = tau - r.p * tpp; return r; };
    auto pos_advance = [&](TilePos &r) {
        r.tau += 8; r.t += adv_t; r.p += adv_p;
        const bool wrap = r.t >= tpp;
        r.t = wrap ? r.t - tpp : r.t;
        r.p = wrap ? r.p + 1 : r.p;
    };
    // first output sample n0 of the tile and rem = samples from n0 to T (<= 0: none; always for tiles past the end)
    auto tile_span = [&](const TilePos &r, int &n0, int &rem) {
        n0 = r.p + 16 * r.t * dil;
        rem = r.tau < tiles_end ? T - n0 : 0;
    };
    TilePos pc[4], ps[4];                    // tiles of the iteration being computed / being staged
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) pc[nt] = ps[nt] = pos_init(tile0 + 4 * ng + nt);
    f32x4 sreg[8];
    auto stage_load = [&]() {               // loads the windows of the tiles in `ps`
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            int n0, rem;
            tile_span(ps[nt], n0, rem);
            const bool some = rem > 0, hist = some && ps[nt].t != 0;
            const char *bO = some ? (const char *)(ib + (int64_t)n0 * TC) : (const char *)tcn_zeros;
            const char *bH = hist ? bO - hist_bytes : (const char *)tcn_zeros;
            // scalar byte limits; a lane offset beyond them is clamped to an in-bounds 16-byte piece whose content nobody
            // uses (rows >= T feed only outputs >= T) resp. to the zero page (offsets are multiples of 16)
            const int remc = rem < (1 << 24) ? rem : (1 << 24);
            const unsigned limO = some ? (unsigned)(remc - 1) * (TC * 4u) + (TC * 4u - 16u) : 0u;
            const unsigned limH = hist ? 0xffffffffu : (TC * 4u - 16u);
            sreg[2 * nt] = *(const f32x4 *)(bH + __builtin_elementwise_min(offH, limH));
            sreg[2 * nt + 1] = *(const f32x4 *)(bO + __builtin_elementwise_min(offO, limO));
        }
    };
    auto stage_store = [&](int buf) {
        float *dst = tsm + buf * (TCN2_SMEM_FLOATS / 2);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            *(f32x4 *)&dst[ldsH + nt * TILE_F] = sreg[2 * nt];
            *(f32x4 *)&dst[ldsO + nt * TILE_F] = sreg[2 * nt + 1];
        }
    };
    const int rd_base = (ng * 4) * TILE_F + j * TRS + 8 * q;   // + nt*TILE_F + k*TRS (+4 for the upper 4 channels)

    if (niter <= 0) return;
    stage_load();
    stage_store(0);
    __syncthreads();
    // every load of the prologue (weights, bias) is complete before the loop: otherwise hipcc parks a vmcnt wait for them
    // at the loop head, which on every later iteration waits for the epilogue's stores instead (loads and stores
    // share the counter on gfx9-class hardware)
    __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0), other counters untouched
    for (int it = 0; it < niter; ++it) {
        const int buf = it & 1;
        // the next iteration's windows (past the last iteration: tiles >= tiles_end, i.e. the zero page)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) pos_advance(ps[nt]);
        stage_load();
        __builtin_amdgcn_sched_barrier(0);      // the loads go out HERE, a whole iteration of MFMAs ahead of their use (hipcc sinks them to the epilogue otherwise)
        const float *tb = tsm + buf * (TCN2_SMEM_FLOATS / 2) + rd_base;
        f32x4 acc[4], res[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { acc[nt] = bi; res[nt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; }
#pragma unroll
        for (int k = 0; k < TK; ++k) {
            f32x4 lo[4], hi[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                lo[nt] = *(const f32x4 *)(tb + nt * TILE_F + k * TRS);
                hi[nt] = *(const f32x4 *)(tb + nt * TILE_F + k * TRS + 4);
            }
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const float bv = s < 4 ? lo[nt][s] : hi[nt][s - 4];
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Aw[k][s], bv, acc[nt], 0, 0, 0);
                    if (k == TK - 1) res[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ar[s], bv, res[nt], 0, 0, 0);
                }
        }
        // The staged windows go to LDS BEFORE the epilogue's global stores: the wait for the loads (issued a whole MFMA
        // block ago) is then free.  Behind the stores it is a vmcnt(0) that also waits for the stores' acknowledgement,
        // ~1500 cycles per iteration (stores and loads share the counter; the stores sit in a divergent branch, so hipcc
        // cannot count them and waits for everything).
        stage_store(buf ^ 1);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            int n0, rem;
            tile_span(pc[nt], n0, rem);
            const f32x4 v = prelu_plus(acc[nt], al, res[nt]);
            if constexpr (FUSE_OUT) {
                const float p = sum_lane_groups(dot4(owv, v));
                if (mt == 1) { if (q == 0) ypp[it & 1][ng][nt][j] = p; }
                else { ypart[nt] = p; yn[nt] = jdS < rem ? n0 + jdS : T; }
            } else {
                if (jdS < rem) *(f32x4 *)((char *)(ob + (int64_t)n0 * TC) + offS) = v;
            }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) pos_advance(pc[nt]);
        __syncthreads();
        finish_y(it & 1);
    }
}

// ---- 32 -> 32 channel block for LARGE dilations: phase-GROUP tiles with a sliding window -------------------------
// With dil = 1000 and T = 65 536 a phase holds 66 outputs: in polyphase tiles of 16 that is 5 tiles = 80 slots (17.5 %
// wasted), every staged row is a separate 128-byte line dil rows away from the next, and the fused output conv stores y
// with the polyphase stride (4-byte stores dil samples apart): 81.7 ms against 58.4 ms for the small dilations.
// Here a tile is 16 ADJACENT phases at ONE time index: outputs n = 16 g + j + m dil, j = 0..15 (group g, time index m).
// Tap k reads the rows 16 g + j + (m - 12 + k) dil: the 16 rows of "row block" (g, m - 12 + k), 2 KB contiguous in
// memory.  A wave pair walks m = 0, 1, 2, ... for its group and keeps the last 13 row blocks in an LDS ring of 16
// slots: ONE new 2 KB block per tile (a single 16-byte load per lane), no padding along time, and the fused
// output conv stores 16 consecutive samples (64 bytes).  Same MFMA order per output as the polyphase kernel
// (bit-identical results).  Two tiles (m, m + 1) per iteration and barrier.  Wasted slots: only the phases 16 g + j >=
// dil of the last group (8 of 1008 for dil = 1000).
constexpr int PG_SLOTS = 16;                 // ring of row blocks per pair (13 live + 2 being filled, power of two)
constexpr int PG_BLK_F = 16 * TRS;           // floats per row block
constexpr int PG_SMEM_FLOATS = 2 * PG_SLOTS * PG_BLK_F;     // two pairs: 73 728 B -> two workgroups per CU

template <int V> struct IntC { static constexpr int value = V; };

template <bool FUSE_OUT>
__global__ __launch_bounds__(256, 2) void tcn_block_pg_kernel(const float *in, float *out, const float *W,
                                                              const float *bias, const float *alpha, const float *R,
                                                              int dil, int64_t T64, int groups, const float *ow,
                                                              const float *obias, float *yout)
{
    extern __shared__ __attribute__((aligned(16))) float tsm[];
    // FUSE_OUT: the lanes' output-conv partials, [iteration parity][pair][tile][sample j][mt * 4 + q] (the 8 partials of a
    // sample are 32 contiguous bytes)
    __shared__ __attribute__((aligned(16))) float ypp[2][2][2][16][8];
    const int tid = threadIdx.x, l = tid & 63;
#ifdef NTM_LAB
    unsigned long long t_begin_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_begin_)::"memory");
#endif
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = w & 1, ng = w >> 1;
    const int q = l >> 4, j = l & 15;
    const int64_t b = blockIdx.x;
    const int g = 2 * blockIdx.y + ng;                   // this pair's phase group (may be past the end: idles)
    const bool gvalid = g < groups;
    const int T = (int)T64;                              // launch_tcn: T < 2^31, dil <= 2^20 (32-bit sample arithmetic)
    const float *ib = in + b * T64 * TC;
    float *ob = out + b * T64 * TC;
    const int mtot = (int)((T64 + dil - 1) / dil);       // time indices
    const int niter = (mtot + 1) / 2;

    float Aw[TK][8], Ar[8];
#pragma unroll
    for (int k = 0; k < TK; ++k)
#pragma unroll
        for (int s = 0; s < 8; ++s) Aw[k][s] = W[((8 * q + s) * TK + k) * TC + 16 * mt + j];
#pragma unroll
    for (int s = 0; s < 8; ++s) Ar[s] = R[(8 * q + s) * TC + 16 * mt + j];
    f32x4 bi, al, owv = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int v = 0; v < 4; ++v) { bi[v] = bias[16 * mt + 4 * q + v]; al[v] = alpha[16 * mt + 4 * q + v]; }
    if constexpr (FUSE_OUT) {
#pragma unroll
        for (int v = 0; v < 4; ++v) owv[v] = ow[16 * mt + 4 * q + v];
    }
    float *yb = FUSE_OUT ? yout + b * T64 : nullptr;
    const float ob0 = FUSE_OUT ? obias[0] : 0.0f;

    // staging: a row block = 16 rows x 8 pieces of 16 B = 128 pieces = one per lane of the pair, 2 KB contiguous.  The
    // block is the same for every lane: a scalar base (the block, or the zero page when the group / the time index does
    // not exist) plus a constant lane offset -- no vector arithmetic at all.  A block that straddles T is read whole:
    // its rows >= T feed only outputs >= T, and they are in bounds because launch_tcn pads every activation buffer by
    // one row block.
    const int e = mt * 64 + l, st_r = e >> 3, st_c8 = e & 7;
    const unsigned voff = (unsigned)(st_r * TC + 4 * st_c8) * 4u;
    float *ring = tsm + ng * (PG_SLOTS * PG_BLK_F);
    float *const st_lds = ring + st_r * TRS + 4 * st_c8;                       // + slot * PG_BLK_F
    const float *const rd_lds = ring + j * TRS + 8 * q;                        // + slot * PG_BLK_F (+ 4: upper 4 channels)
    const int g16 = 16 * g;
    auto block_load = [&](int mb) -> f32x4 {             // mb >= 0.  (rows of aliased phases >= dil are real rows too)
        const int n0 = g16 + mb * dil;                   // < T + dil + 16 g: no overflow
        const char *base = (gvalid && n0 < T) ? (const char *)(ib + (int64_t)n0 * TC) : (const char *)tcn_zeros;
        return *(const f32x4 *)(base + voff);
    };
    // the lane's output sample inside a row block: row j if its phase 16 g + j exists, else "never valid"
    const int jrow = (gvalid && g16 + j < dil) ? j : 0x40000000;
    const unsigned offS = (unsigned)(j * TC + 16 * mt + 4 * q) * 4u;

    if (niter <= 0) return;
    // prologue: blocks -12 .. -1 are zeros, blocks 0 and 1 come from memory
#pragma unroll
    for (int sl = PG_SLOTS - (TK - 1); sl < PG_SLOTS; ++sl) *(f32x4 *)(st_lds + sl * PG_BLK_F) = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    *(f32x4 *)(st_lds + 0 * PG_BLK_F) = block_load(0);
    *(f32x4 *)(st_lds + 1 * PG_BLK_F) = block_load(1);
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0): no prologue load is pending at the loop head (see tcn_block_mfma2_kernel)

    // One iteration = the tiles m0 = 2 it and m0 + 1.  The ring slot of row block mb is mb & 15, so the slots an iteration
    // touches depend only on PH = it & 7: the loop is unrolled over the 8 phases and every LDS address is the lane's
    // base register + an immediate (the rolled form needed a v_add per block -- 17 of the ~100 vector instructions
    // per iteration, and a vector instruction of either wave on a SIMD costs ~8 matrix-pipe cycles).
    unsigned long long seg_[6] = {0, 0, 0, 0, 0, 0}, last_ = 0, ts_[6];
    (void)seg_; (void)last_; (void)ts_;
    auto iteration = [&](const int it, auto ph_c) {
        constexpr int PH = decltype(ph_c)::value;
        constexpr int S0 = 2 * PH;                        // slot of row block m0
        const int m0 = 2 * it;
        TCN_STAMP(0)
        const f32x4 nb0 = block_load(m0 + 2), nb1 = block_load(m0 + 3);     // the next iteration's new blocks
        __builtin_amdgcn_sched_barrier(0);
        TCN_STAMP(1)   // block loads issued
        f32x4 acc[2], res[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) { acc[nt] = bi; res[nt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; }
        // B operands.  Tap k of tile m0 + 1 reads the row block that tap k + 1 of tile m0 reads: the iteration walks
        // the 14 blocks m0 - 12 .. m0 + 1 once (28 ds_read_b128 instead of 52) and feeds each to both tiles, taps in
        // ascending order per output as before.  The reads of block s + 1 are issued BEFORE the MFMAs of block s
        // (register double buffer, the scheduler pinned by sched_barrier): left to itself hipcc puts each read right in
        // front of its first MFMA and every block waits out the LDS latency.
        f32x4 lo[2], hi[2];
#define PG_READ(s, buf)                                                                         \
        {                                                                                       \
            const int sl = (S0 - (TK - 1) + (s) + 2 * PG_SLOTS) & (PG_SLOTS - 1);               \
            lo[buf] = *(const f32x4 *)(rd_lds + sl * PG_BLK_F);                                 \
            hi[buf] = *(const f32x4 *)(rd_lds + sl * PG_BLK_F + 4);                             \
        }
        PG_READ(0, 0)
#pragma unroll
        for (int s = 0; s <= TK; ++s) {
            const int cb = s & 1;
            if (s + 1 <= TK) PG_READ(s + 1, cb ^ 1)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float bv = c < 4 ? lo[cb][c] : hi[cb][c - 4];
                if (s < TK) acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(Aw[s < TK ? s : 0][c], bv, acc[0], 0, 0, 0);
                if (s >= 1) acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(Aw[s >= 1 ? s - 1 : 0][c], bv, acc[1], 0, 0, 0);
                if (s == TK - 1) res[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ar[c], bv, res[0], 0, 0, 0);
                if (s == TK) res[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ar[c], bv, res[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef PG_READ
        TCN_STAMP(2)   // MFMA block (14 row blocks)
        // new row blocks -> ring BEFORE the epilogue's global stores (see tcn_block_mfma2_kernel); their slots are
        // those of blocks m0 - 14, m0 - 13, which no tile of this iteration reads
        *(f32x4 *)(st_lds + ((S0 + 2) & (PG_SLOTS - 1)) * PG_BLK_F) = nb0;
        *(f32x4 *)(st_lds + ((S0 + 3) & (PG_SLOTS - 1)) * PG_BLK_F) = nb1;
        TCN_STAMP(3)   // ring stores (vmcnt wait)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n0 = g16 + (m0 + nt) * dil;          // first sample of the tile's row block
            const int rem = T - n0;                         // rows left up to T (<= 0: none)
            const f32x4 v = prelu_plus(acc[nt], al, res[nt]);
            if constexpr (FUSE_OUT) {
                // fused 1x1 output conv: every lane parks its partial over its 4 channels in LDS (no cross-lane
                // arithmetic here: that was 6 vector instructions per tile in BOTH waves); the sums follow the barrier
                ypp[PH & 1][ng][nt][j][mt * 4 + q] = dot4(owv, v);
            } else {
                if (jrow < rem) *(f32x4 *)((char *)(ob + (int64_t)n0 * TC) + offS) = v;
            }
        }
        TCN_STAMP(4)   // epilogue
        __syncthreads();
        TCN_STAMP(5)   // barrier
#ifdef NTM_LAB
        if constexpr (!FUSE_OUT) {
            for (int i = 0; i < 5; ++i) seg_[i] += ts_[i + 1] - ts_[i];
            if (it > 0) seg_[5] += ts_[0] - last_;      // loop back-edge
            last_ = ts_[5];
            if (blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2 && tid == 0 && it < 64) {
                tcn_stamp_out[8 + 3 * it] = ts_[0] - t_begin_;
                tcn_stamp_out[8 + 3 * it + 1] = ts_[2] - ts_[1];
                tcn_stamp_out[8 + 3 * it + 2] = ts_[5] - ts_[0];
            }
        }
#endif
        if constexpr (FUSE_OUT) {
            // wave mt finishes tile nt = mt: the 8 partials of sample j (two ds_read_b128), summed in a fixed order,
            // 16 consecutive samples = one 64-byte store by the lanes q = 0
            const int n0 = g16 + (m0 + mt) * dil, rem = T - n0;
            const f32x4 pa = *(const f32x4 *)&ypp[PH & 1][ng][mt][j][0], pb = *(const f32x4 *)&ypp[PH & 1][ng][mt][j][4];
            const float ysum = (((pa[0] + pa[1]) + (pa[2] + pa[3])) + ((pb[0] + pb[1]) + (pb[2] + pb[3]))) + ob0;
            if (q == 0 && jrow < rem) *(float *)((char *)(yb + n0) + 4 * j) = ysum;
        }
    };
    for (int it = 0; it < niter; it += 8) {
        iteration(it, IntC<0>{});
        if (it + 1 >= niter) break;
        iteration(it + 1, IntC<1>{});
        if (it + 2 >= niter) break;
        iteration(it + 2, IntC<2>{});
        if (it + 3 >= niter) break;
        iteration(it + 3, IntC<3>{});
        if (it + 4 >= niter) break;
        iteration(it + 4, IntC<4>{});
        if (it + 5 >= niter) break;
        iteration(it + 5, IntC<5>{});
        if (it + 6 >= niter) break;
        iteration(it + 6, IntC<6>{});
        if (it + 7 >= niter) break;
        iteration(it + 7, IntC<7>{});
    }
#ifdef NTM_LAB
    if (!FUSE_OUT && blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2 && tid == 0) {     // a workgroup from the middle of the launch
        for (int i = 0; i < 6; ++i) tcn_stamp_out[i] = seg_[i];
        tcn_stamp_out[6] = (unsigned long long)niter;
    }
    if (!FUSE_OUT && tid == 0 && tcn_trace_buf) {
        unsigned long long t_end;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_end)::"memory");
        unsigned long long *tr = tcn_trace_buf + 4 * ((size_t)blockIdx.x + (size_t)gridDim.x * blockIdx.y);
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        unsigned long long loop = 0;
        for (int i = 0; i < 6; ++i) loop += seg_[i];
        tr[0] = t_begin_; tr[1] = t_end; tr[2] = ((unsigned long long)xcc << 32) | hw; tr[3] = loop;
    }
#endif
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
    if (T >= ((int64_t)1 << 31) - (1 << 25)) return hipErrorInvalidValue;   // 32-bit sample arithmetic in the block kernels
    for (int l = 0; l < L; ++l)
        if (dil[l] <= 0 || dil[l] > (1 << 20)) return hipErrorInvalidValue;
    float *bufA = scratch, *bufB = scratch + (size_t)B * C * T + TCN_PAD_FLOATS;     // each followed by one row block of padding
    const float *p = params;
    const dim3 grid1((unsigned)B, (unsigned)((T + 255) / 256));
    const dim3 gridf((unsigned)B, (unsigned)((T + 31) / 32));
    const float *in = x;
    int cin = 1;
    for (int l = 0; l < L; ++l) {
        const float *W = p;      p += (size_t)C * cin * K;
        const float *bias = p;   p += C;
        const float *alpha = p;  p += C;
        const float *R = p;      p += (size_t)C * cin;
        float *out = (l & 1) ? bufB : bufA;
        if (cin == 1 && dil[l] == 1)
            hipLaunchKernelGGL(tcn_first_d1_kernel, dim3((unsigned)B, (unsigned)((T + FC * 256 - 1) / (FC * 256))), dim3(256), 0, stream, in,
                               out, W, bias, alpha, R, T);
        else if (cin == 1) hipLaunchKernelGGL(tcn_first_kernel, gridf, dim3(256), 0, stream, in, out, W, bias, alpha, R, dil[l], T);
        else {
            if (dil[l] >= 512) {
                // large dilation: phase-group tiles with a sliding window (two groups per workgroup)
                const int groups = (dil[l] + 15) / 16;
                const dim3 gridg((unsigned)B, (unsigned)((groups + 1) / 2));
                size_t smem = PG_SMEM_FLOATS * sizeof(float);
#ifdef NTM_LAB
                if (getenv("NTM_LAB_TCN_ONE_WG")) smem = 90 * 1024;       // diagnostic: one workgroup per CU, one wave per SIMD
#endif
                if (l == L - 1) {
                    auto k = tcn_block_pg_kernel<true>;
                    hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
                    if (e != hipSuccess) return e;
                    hipLaunchKernelGGL(k, gridg, dim3(256), smem, stream, in, out, W, bias, alpha, R, dil[l], T, groups, p, p + C, y);
                    return hipGetLastError();
                }
                auto k = tcn_block_pg_kernel<false>;
                hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(k, gridg, dim3(256), smem, stream, in, out, W, bias, alpha, R, dil[l], T, groups, nullptr,
                                   nullptr, nullptr);
                in = out;
                cin = C;
                continue;
            }
            // polyphase tiles: M = ceil(T / dil) outputs per phase, tpp = ceil(M / 16) tiles per phase
            const int64_t M = (T + dil[l] - 1) / dil[l];
            const int tpp = (int)((M + 15) / 16);
            const int64_t total = (int64_t)dil[l] * tpp;
            if (dil[l] <= 0 || total > (int64_t)1 << 30) return hipErrorInvalidValue;
            const dim3 gridp((unsigned)B, (unsigned)((total + TPW - 1) / TPW));
            if (l == L - 1) {      // last block: 1x1 output conv fused (out_w, out_b follow this block's parameters)
                hipLaunchKernelGGL(tcn_block_mfma2_kernel<true>, gridp, dim3(256), TCN2_SMEM_FLOATS * sizeof(float), stream,
                                   in, out, W, bias, alpha, R, dil[l], T, tpp, (int)total, p, p + C, y);
                return hipGetLastError();
            }
            hipLaunchKernelGGL(tcn_block_mfma2_kernel<false>, gridp, dim3(256), TCN2_SMEM_FLOATS * sizeof(float), stream, in,
                               out, W, bias, alpha, R, dil[l], T, tpp, (int)total, nullptr, nullptr, nullptr);
        }
        in = out;
        cin = C;
    }
    if (cin != C) return hipErrorInvalidValue;   // L == 0
    hipLaunchKernelGGL(tcn_out_kernel, grid1, dim3(256), 0, stream, in, y, p, p + C, T);   // single-block network
    return hipGetLastError();
}

}  // namespace ntm

#ifdef NTM_LAB
// laboratory entry points (include/ntm_lab.h): the same forward through the STAMP build, and its segment sums
extern "C" int ntm_lab_tcn_forward(const float *params, int L, int C, int K, const int *dil, const float *x, float *y, int64_t B,
                                   int64_t T, float *scratch, void *stream)
{
    return (int)ntm::launch_tcn(params, L, C, K, dil, x, y, B, T, scratch, (hipStream_t)stream);
}
extern "C" int ntm_lab_tcn_trace(unsigned long long *device_buf)      // nullptr: off; else 4 words per workgroup of the stamped launch
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(ntm::tcn_trace_buf), &device_buf, sizeof(device_buf));
}
extern "C" int ntm_lab_tcn_stamps(unsigned long long *host7)      // 8 + 3 * 64 words
{
    return (int)hipMemcpyFromSymbol(host7, HIP_SYMBOL(ntm::tcn_stamp_out), (8 + 3 * 64) * sizeof(unsigned long long));
}
#endif
