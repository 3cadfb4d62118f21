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

#include <type_traits>

namespace ntm {

__device__ float tcn_zeros[32];   // zero page for taps that fall before the start of a stream

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

// ---- first block, dilation 1 (the frozen spec): thread -> (8 consecutive samples, group of 4 output channels) --
// The 8 samples share a window of 20 inputs (loaded once, 13 x 8 in the generic kernel) and the 13 weight
// vectors stay in registers; the kernel then runs at the rate of its 34 GB of output.  Same operation order per
// output as the generic kernel (bit-identical results).
__global__ __launch_bounds__(256) void tcn_first_d1_kernel(const float *x, float *out, const float *W, const float *bias,
                                                           const float *alpha, const float *R, int64_t T)
{
    constexpr int S = 8;
    const int64_t b = blockIdx.x;
    const int c4 = threadIdx.x & 7;
    const int64_t n0 = ((int64_t)blockIdx.y * 32 + (threadIdx.x >> 3)) * S;
    if (n0 >= T) return;
    const float *xb = x + b * T;
    float xw[S + TK - 1];
#pragma unroll
    for (int i = 0; i < S + TK - 1; ++i) {
        const int64_t src = n0 - (TK - 1) + i;
        xw[i] = (src >= 0 && src < T) ? xb[src] : 0.0f;
    }
    f32x4 wv[TK];
#pragma unroll
    for (int k = 0; k < TK; ++k) wv[k] = *(const f32x4 *)(W + k * TC + 4 * c4);
    const f32x4 bi = *(const f32x4 *)(bias + 4 * c4), al = *(const f32x4 *)(alpha + 4 * c4), rv = *(const f32x4 *)(R + 4 * c4);
    float *ob = out + (b * T + n0) * TC + 4 * c4;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        if (n0 + s >= T) break;
        f32x4 acc = bi;
#pragma unroll
        for (int k = 0; k < TK; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(wv[k][e], xw[s + k], acc[e]);
        const float x0 = xw[s + TK - 1];
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(rv[e], x0, acc[e] >= 0.0f ? acc[e] : al[e] * acc[e]);
        *(f32x4 *)(ob + (int64_t)s * TC) = v;
    }
}

// ---- 32 -> 32 channel block, polyphase tile order + LDS staging (the one launch_tcn uses) -------------
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
                                                                 const float *R, int dil, int64_t T, int tpp,
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
    const float *ib = in + b * T * TC;
    float *ob = out + b * T * TC;
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
    int64_t yn[4] = {T, T, T, T};
    float *yb = FUSE_OUT ? yout + b * T : nullptr;
    const float ob0 = FUSE_OUT ? obias[0] : 0.0f;
    auto finish_y = [&](int parity) {              // after the barrier: wave mt = 0 adds the other half and stores
        if constexpr (FUSE_OUT) {
            if (mt == 0 && q == 0) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    if (yn[nt] < T) yb[yn[nt]] = (ypart[nt] + ypp[parity][ng][nt][j]) + ob0;
            }
        }
    };

    // staging: the pair's 4 windows = 112 rows x 8 pieces of 16 B = 896 pieces over its 128 lanes (7 each)
    int st_nt[7], st_roff[7], st_lds[7];
    int64_t st_goff[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) {
        const int e = mt * 64 + l + 128 * c;
        const int nt = e / 224, r = (e % 224) >> 3, c8 = e & 7;
        st_nt[c] = nt;
        st_roff[c] = r - (TK - 1);                                  // row relative to the tile's first output
        st_goff[c] = (int64_t)(r - (TK - 1)) * dil * TC + 4 * c8;    // floats, relative to the tile's first output
        st_lds[c] = (ng * 4 + nt) * TILE_F + r * TRS + 4 * c8;
    }
    // Tile tau = p * tpp + t (phase p, tile t of the phase).  The pair's four tiles advance by 8 per iteration; their
    // (p, t) are carried as wave-uniform counters (one division per tile before the loop instead of eight per
    // iteration: scalar instructions inside the MFMA stream are not free, see gru_mfma2.hip).
    struct TilePos { int tau, p, t; };
    auto pos_init = [&](int tau) { TilePos q; q.tau = tau; q.p = tau / tpp; q.t = tau - q.p * tpp; return q; };
    auto pos_advance = [&](TilePos &q) {
        q.tau += 8; q.t += 8;
        while (q.t >= tpp) { q.t -= tpp; ++q.p; }
    };
    // first output sample of the tile (>= T for tiles beyond the end) and whether the tile starts a phase
    auto tile_n0 = [&](const TilePos &q, bool &first) -> int64_t {
        if (q.tau >= tiles_end) { first = false; return T; }
        first = (q.t == 0);
        return (int64_t)q.p + (int64_t)(16 * q.t) * dil;
    };
    TilePos pc[4], ps[4];                    // tiles of the iteration being computed / being staged
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) pc[nt] = ps[nt] = pos_init(tile0 + 4 * ng + nt);
    f32x4 sreg[7];
    auto stage_load = [&]() {               // loads the windows of the tiles in `ps`
        int64_t n0[4]; bool fst[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) n0[nt] = tile_n0(ps[nt], fst[nt]);
#pragma unroll
        for (int c = 0; c < 7; ++c) {
            const int nt = st_nt[c];
            const int64_t base = nt == 0 ? n0[0] : nt == 1 ? n0[1] : nt == 2 ? n0[2] : n0[3];
            const bool f = nt == 0 ? fst[0] : nt == 1 ? fst[1] : nt == 2 ? fst[2] : fst[3];
            const int64_t n = base + (int64_t)st_roff[c] * dil;
            const bool ok = n < T && !(f && st_roff[c] < 0);
            const float *ptr = ok ? ib + base * TC + st_goff[c] : tcn_zeros + (st_goff[c] & 31);
            sreg[c] = *(const f32x4 *)ptr;
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int c = 0; c < 7; ++c) *(f32x4 *)&tsm[buf * (TCN2_SMEM_FLOATS / 2) + st_lds[c]] = sreg[c];
    };
    const int rd_base = (ng * 4) * TILE_F + j * TRS + 8 * q;   // + nt*TILE_F + k*TRS (+4 for the upper 4 channels)

    if (niter <= 0) return;
    stage_load();
    stage_store(0);
    __syncthreads();
    for (int it = 0; it < niter; ++it) {
        const int buf = it & 1;
        if (it + 1 < niter) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) pos_advance(ps[nt]);
            stage_load();
        }
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
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            bool fst;
            const int64_t n = tile_n0(pc[nt], fst) + (int64_t)j * dil;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float u = acc[nt][e];
                v[e] = (u >= 0.0f ? u : al[e] * u) + res[nt][e];
            }
            if constexpr (FUSE_OUT) {
                float p = (owv[0] * v[0] + owv[1] * v[1]) + (owv[2] * v[2] + owv[3] * v[3]);
                p += __shfl_xor(p, 16, 64);
                p += __shfl_xor(p, 32, 64);
                if (mt == 1) { if (q == 0) ypp[it & 1][ng][nt][j] = p; }
                else { ypart[nt] = p; yn[nt] = n; }
            } else {
                if (n < T) *(f32x4 *)(ob + n * TC + 16 * mt + 4 * q) = v;
            }
        }
        if (it + 1 < niter) stage_store(buf ^ 1);
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

template <bool FUSE_OUT>
__global__ __launch_bounds__(256, 2) void tcn_block_pg_kernel(const float *in, float *out, const float *W,
                                                              const float *bias, const float *alpha, const float *R,
                                                              int dil, int64_t T, int groups, const float *ow,
                                                              const float *obias, float *yout)
{
    extern __shared__ __attribute__((aligned(16))) float tsm[];
    __shared__ float ypp[2][2][2][16];        // FUSE_OUT: [iteration parity][pair][tile][sample] partial of wave mt = 1
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = w & 1, ng = w >> 1;
    const int q = l >> 4, j = l & 15;
    const int64_t b = blockIdx.x;
    const int g = 2 * blockIdx.y + ng;                   // this pair's phase group (may be past the end: idles)
    const bool gvalid = g < groups;
    const float *ib = in + b * T * TC;
    float *ob = out + b * T * TC;
    const int mtot = (int)((T + dil - 1) / dil);         // time indices
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
    float ypart[2] = {0.0f, 0.0f};
    int64_t yn[2] = {T, T};
    float *yb = FUSE_OUT ? yout + b * T : nullptr;
    const float ob0 = FUSE_OUT ? obias[0] : 0.0f;
    auto finish_y = [&](int parity) {                  // after the barrier: wave mt = 0 adds the other half and stores
        if constexpr (FUSE_OUT) {
            if (mt == 0 && q == 0) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    if (yn[nt] < T) yb[yn[nt]] = (ypart[nt] + ypp[parity][ng][nt][j]) + ob0;
            }
        }
    };

    // staging: a row block = 16 rows x 8 pieces of 16 B = 128 pieces = one per lane of the pair
    const int e = mt * 64 + l, st_r = e >> 3, st_c8 = e & 7;
    float *ring = tsm + ng * (PG_SLOTS * PG_BLK_F);
    auto block_load = [&](int mb) -> f32x4 {             // row block (g, mb): rows 16 g + st_r + mb dil
        const int64_t n = (int64_t)16 * g + st_r + (int64_t)mb * dil;
        const bool ok = gvalid && mb >= 0 && n < T && 16 * g + st_r < dil + 16;   // (rows of aliased phases are real rows too)
        const float *ptr = ok ? ib + n * TC + 4 * st_c8 : tcn_zeros + 4 * st_c8;
        return *(const f32x4 *)ptr;
    };
    auto block_store = [&](int mb, f32x4 v) { *(f32x4 *)&ring[(mb & (PG_SLOTS - 1)) * PG_BLK_F + st_r * TRS + 4 * st_c8] = v; };
    const int rd_base = j * TRS + 8 * q;

    if (niter <= 0) return;
    // prologue: blocks -12 .. -1 are zeros, blocks 0 and 1 come from memory
    for (int mb = -12; mb < 0; ++mb) block_store(mb, (f32x4){0.0f, 0.0f, 0.0f, 0.0f});
    block_store(0, block_load(0));
    block_store(1, block_load(1));
    __syncthreads();
    for (int it = 0; it < niter; ++it) {
        const int m0 = 2 * it;
        const f32x4 nb0 = block_load(m0 + 2), nb1 = block_load(m0 + 3);     // the next iteration's new blocks
        f32x4 acc[2], res[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) { acc[nt] = bi; res[nt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; }
        // B operands: the ds_read_b128 of tap k + 1 are issued BEFORE the MFMAs of tap k (register double buffer, the
        // scheduler pinned by sched_barrier): left to itself hipcc puts each read right in front of its first MFMA and
        // every tap waits out the LDS latency (~110 of its 512 matrix-pipe cycles: the 0.83 of the first version)
        f32x4 lo[2][2], hi[2][2];
        auto tap_read = [&](int k, int buf) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const float *tb = ring + ((m0 + nt - (TK - 1) + k) & (PG_SLOTS - 1)) * PG_BLK_F + rd_base;
                lo[buf][nt] = *(const f32x4 *)tb;
                hi[buf][nt] = *(const f32x4 *)(tb + 4);
            }
        };
        tap_read(0, 0);
#pragma unroll
        for (int k = 0; k < TK; ++k) {
            const int cb = k & 1;
            if (k + 1 < TK) tap_read(k + 1, cb ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const float bv = s < 4 ? lo[cb][nt][s] : hi[cb][nt][s - 4];
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Aw[k][s], bv, acc[nt], 0, 0, 0);
                    if (k == TK - 1) res[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ar[s], bv, res[nt], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int p = 16 * g + j;                                        // phase of this lane's output column
            const int64_t n = (gvalid && p < dil && m0 + nt < mtot) ? (int64_t)p + (int64_t)(m0 + nt) * dil : T;
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float u = acc[nt][c];
                v[c] = (u >= 0.0f ? u : al[c] * u) + res[nt][c];
            }
            if constexpr (FUSE_OUT) {
                float pp = (owv[0] * v[0] + owv[1] * v[1]) + (owv[2] * v[2] + owv[3] * v[3]);
                pp += __shfl_xor(pp, 16, 64);
                pp += __shfl_xor(pp, 32, 64);
                if (mt == 1) { if (q == 0) ypp[it & 1][ng][nt][j] = pp; }
                else { ypart[nt] = pp; yn[nt] = n; }
            } else {
                if (n < T) *(f32x4 *)(ob + n * TC + 16 * mt + 4 * q) = v;
            }
        }
        block_store(m0 + 2, nb0);          // slots of blocks m0 - 14, m0 - 13: no tile of this iteration reads them
        block_store(m0 + 3, nb1);
        __syncthreads();
        finish_y(it & 1);
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
    const float *in = x;
    int cin = 1;
    for (int l = 0; l < L; ++l) {
        const float *W = p;      p += (size_t)C * cin * K;
        const float *bias = p;   p += C;
        const float *alpha = p;  p += C;
        const float *R = p;      p += (size_t)C * cin;
        float *out = (l & 1) ? bufB : bufA;
        if (cin == 1 && dil[l] == 1)
            hipLaunchKernelGGL(tcn_first_d1_kernel, dim3((unsigned)B, (unsigned)((T + 255) / 256)), dim3(256), 0, stream, in,
                               out, W, bias, alpha, R, T);
        else if (cin == 1) hipLaunchKernelGGL(tcn_first_kernel, gridf, dim3(256), 0, stream, in, out, W, bias, alpha, R, dil[l], T);
        else {
            if (dil[l] >= 512) {
                // large dilation: phase-group tiles with a sliding window (two groups per workgroup)
                const int groups = (dil[l] + 15) / 16;
                const dim3 gridg((unsigned)B, (unsigned)((groups + 1) / 2));
                const size_t smem = PG_SMEM_FLOATS * sizeof(float);
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
