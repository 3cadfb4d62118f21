// "Next" row N1: the spectral part of the evaluation metrics that follow the path in
// code/test-model.py:250-254,386-388 -- `MultiResolutionSTFTLoss()` of the un-vendored auraloss submodule
// (code/test-model.py:25).  One kernel per resolution produces, per stream, the sums the loss is made of;
// the host (model.py MRSTFTLoss) divides and averages.
//
// Arithmetic (auraloss.freq.STFTLoss at its defaults): X = torch.stft(x, n_fft, hop, win_length,
// hann_window(win_length)) -- centred frames, reflect padding of n_fft/2, window zero-padded to n_fft on both
// sides -- and mag = sqrt(clamp(re^2 + im^2, min = eps)).
//
// Mapping: one 64-lane wave transforms one frame PAIR: z = w*(y + i t) goes through ONE complex n_fft-point
// FFT, and the two real spectra are separated afterwards (Y[k] = (Z[k] + conj Z[N-k])/2,
// T[k] = (Z[k] - conj Z[N-k])/(2i)).  The FFT is a Stockham autosort FFT in three passes of radix 16/8/4
// (256 = 4.4.4.4, 512 = 8.8.8, 1024 = 16.4.16, 2048 = 16.8.16): each lane holds n_fft/64 points in registers
// and runs whole radix-R butterflies on them; passes exchange through the wave's own padded
// LDS buffer, so no workgroup barrier is needed inside the frame loop (LDS operations of one wave are
// processed in issue order).  Window values and pass twiddles are per-lane, frame-invariant constants kept
// in registers (computed once with sincospi).  A workgroup = 4 waves works on one (stream, frame chunk);
// sums are accumulated in fp64 per lane and written per (stream, chunk, wave) -- the host adds them in a
// fixed order (deterministic, no atomics).
//
// Bound: VALU (about 5 n log2 n flop per frame pair); the signals are read once from HBM per resolution
// (frames overlap 5x in L1/L2), 8 B/sample algorithmic.
#include "ntm_common.h"

namespace ntm {

struct StftArgs {
    const float *y, *t;
    int64_t B, T, skip;
    int hop, win, chunks, frames_per_chunk, n_frames, mode;
    float eps;
    double *out;
    // mode 2 (mel): row-compressed mel filter bank (device): filter m = mel_w[mel_start[m] .. mel_start[m+1]) on the
    // bins mel_first[m] ...
    const int *mel_first, *mel_start;
    const float *mel_w;
    int n_mels;
};

typedef float f2 __attribute__((ext_vector_type(2)));

// Complex arithmetic on packed fp32 pairs with the operand-select / negate modifiers of the VOP3P encoding written out:
// hipcc builds a complex product from v_pk_mul + TWO v_pk_fma (one per sign pattern) + a v_mov that splices their halves,
// and a multiplication by -i from v_xor + v_mov -- 90 of the ~660 vector instructions of a 1024-point frame pair, in a
// kernel that is bound by vector issue.
__device__ __forceinline__ f2 cmul(f2 a, f2 b)
{
    f2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));                     // (ax bx, ax by)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"             // (- ay by, + ay bx)
        : "+v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f2 mul_mi(f2 a) { return (f2){a.y, -a.x}; }             // a * (-i)
// The two real spectra inside Z = FFT(y + i t), UNSCALED:  2 Y[k] = Z[k] + conj Z[N-k],  2 T[k] = (Z[k] - conj Z[N-k]) / i,
// one packed add each.  (The factor 1/2 -- 1/4 on the powers -- is a power of two: it is applied once to the frame
// sums, with the clamp floor scaled the other way; results are bit-identical to scaling every bin.)
__device__ __forceinline__ void split_spectra(f2 zk, f2 zn, f2 &Y2, f2 &T2)
{
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(Y2) : "v"(zk), "v"(zn));                                   // (zk.x + zn.x, zk.y - zn.y)
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(T2) : "v"(zk), "v"(zn));      // (zk.y + zn.y, zn.x - zk.x)
}
// a + (-i) d  and  a - (-i) d  in one packed add each (the half swap and the sign ride on the second operand)
__device__ __forceinline__ f2 add_mi(f2 a, f2 d)
{
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(d));   // (a.x + d.y, a.y - d.x)
    return r;
}
__device__ __forceinline__ f2 sub_mi(f2 a, f2 d)
{
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(d));   // (a.x - d.y, a.y + d.x)
    return r;
}

__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- in-register DFTs (forward, e^{-2 pi i nk/R}), natural order in and out -------------------------------
__device__ __forceinline__ void dft2(f2 &a, f2 &b) { const f2 t = a; a = t + b; b = t - b; }

__device__ __forceinline__ void dft4(f2 &x0, f2 &x1, f2 &x2, f2 &x3)
{
    const f2 s02 = x0 + x2, d02 = x0 - x2, s13 = x1 + x3, d13 = x1 - x3;
    x0 = s02 + s13; x1 = add_mi(d02, d13); x2 = s02 - s13; x3 = sub_mi(d02, d13);
}

// the same with x2 standing for (-i) x2 (the W16^4 twiddle of the radix-16 butterfly, folded into the first adds)
__device__ __forceinline__ void dft4_x2mi(f2 &x0, f2 &x1, f2 &x2, f2 &x3)
{
    const f2 s02 = add_mi(x0, x2), d02 = sub_mi(x0, x2), s13 = x1 + x3, d13 = x1 - x3;
    x0 = s02 + s13; x1 = add_mi(d02, d13); x2 = s02 - s13; x3 = sub_mi(d02, d13);
}

constexpr float C8 = 0.70710678118654752f;                       // cos(pi/4)
constexpr float C16 = 0.92387953251128674f, S16 = 0.38268343236508977f;   // cos, sin(pi/8)

template <int R> __device__ __forceinline__ void dft(f2 (&x)[R]);

template <> __device__ __forceinline__ void dft<2>(f2 (&x)[2]) { dft2(x[0], x[1]); }
template <> __device__ __forceinline__ void dft<4>(f2 (&x)[4]) { dft4(x[0], x[1], x[2], x[3]); }

// R = 8 = 4 x 2:  n = 2 n1 + n2,  k = k1 + 4 k2
template <> __device__ __forceinline__ void dft<8>(f2 (&x)[8])
{
    dft4(x[0], x[2], x[4], x[6]);                                 // n2 = 0: a[0][k1] in x[2 k1]
    dft4(x[1], x[3], x[5], x[7]);                                 // n2 = 1: a[1][k1] in x[2 k1 + 1]
    x[3] = add_mi(x[3], x[3]) * (f2){C8, C8};                     // * W8^1 = (c, -c):  c (x + y, y - x)
    // (x[5] * W8^2 = x[5] * (-i) is folded into the last stage below)
    x[7] = sub_mi(x[7], x[7]) * (f2){-C8, -C8};                   // * W8^3 = (-c, -c): -c (x - y, x + y)
    f2 y[8];
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        if (k1 == 2) { y[k1] = add_mi(x[4], x[5]); y[k1 + 4] = sub_mi(x[4], x[5]); }
        else { y[k1] = x[2 * k1] + x[2 * k1 + 1]; y[k1 + 4] = x[2 * k1] - x[2 * k1 + 1]; }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = y[i];
}

// R = 16 = 4 x 4:  n = 4 n1 + n2,  k = k1 + 4 k2
template <> __device__ __forceinline__ void dft<16>(f2 (&x)[16])
{
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) dft4(x[n2], x[4 + n2], x[8 + n2], x[12 + n2]);    // a[n2][k1] in x[4 k1 + n2]
    // twiddles W16^(n2 k1)
    const f2 w1 = {C16, -S16}, w2 = {C8, -C8}, w3 = {S16, -C16}, w6 = {-C8, -C8}, w9 = {-C16, S16};
    x[4 * 1 + 1] = cmul(x[4 * 1 + 1], w1); x[4 * 1 + 2] = cmul(x[4 * 1 + 2], w2); x[4 * 1 + 3] = cmul(x[4 * 1 + 3], w3);
    x[4 * 2 + 1] = cmul(x[4 * 2 + 1], w2); /* x[10] * (-i): inside dft4_x2mi below */ x[4 * 2 + 3] = cmul(x[4 * 2 + 3], w6);
    x[4 * 3 + 1] = cmul(x[4 * 3 + 1], w3); x[4 * 3 + 2] = cmul(x[4 * 3 + 2], w6); x[4 * 3 + 3] = cmul(x[4 * 3 + 3], w9);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {                                                              // X[k1 + 4 k2] in x[4 k1 + k2]
        if (k1 == 2) dft4_x2mi(x[8], x[9], x[10], x[11]);
        else dft4(x[4 * k1], x[4 * k1 + 1], x[4 * k1 + 2], x[4 * k1 + 3]);
    }
    f2 y[16];
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) y[k1 + 4 * k2] = x[4 * k1 + k2];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = y[i];
}

// exchange-buffer index: one float2 of padding per 16 takes the strided Stockham stores from 32-way to two-way
// bank conflicts (PMC: the conflicts that remain cost 6-7 % of the wave cycles)
__device__ __forceinline__ constexpr int padi(int i) { return i + (i >> 4); }

// Frames of n_fft >= 256 use all 64 lanes of a wave (n_fft/64 points per lane); smaller frames use SUB = n_fft/4
// lanes each (4 points per lane) and a wave transforms 64/SUB frames side by side.
template <int N> struct Geo {
    static constexpr int SUB = N >= 256 ? 64 : N / 4;     // lanes per frame
    static constexpr int P = N / SUB;                     // points per lane
    static constexpr int FPW = 64 / SUB;                  // frames per wave and iteration
    static constexpr int NPAD = N + N / 16;               // padded exchange buffer of one frame (float2)
};

// Per-lane twiddles of one Stockham pass, radix R with sub-transform size Ns: butterfly b of this lane is
// j = sl + SUB b (sl = lane within the frame), k = j mod Ns, and input t is multiplied by
// exp(-2 pi i t k / (Ns R)).  Frame-invariant.
template <int N, int R, int Ns> struct PassTw {
    static constexpr int SUB = Geo<N>::SUB;
    static constexpr int NB = N / R / SUB;
    static constexpr int NBW = Ns <= SUB ? 1 : NB;     // Ns <= SUB: k = sl mod Ns is the same for every butterfly of the lane
    f2 w[Ns > 1 ? NBW * (R - 1) : 1];
    __device__ __forceinline__ void init(int sl)
    {
        if constexpr (Ns > 1) {
#pragma unroll
            for (int b = 0; b < NBW; ++b)
#pragma unroll
                for (int t = 1; t < R; ++t) {
                    const int k = (sl + SUB * b) & (Ns - 1);
                    float sn, cs;
                    sincospif(-2.0f * (float)(t * k) / (float)(Ns * R), &sn, &cs);
                    w[b * (R - 1) + t - 1] = (f2){cs, sn};
                }
        }
    }
    __device__ __forceinline__ f2 get(int b, int t) const { return w[(NBW == 1 ? 0 : b) * (R - 1) + t - 1]; }
};

// One Stockham pass over a frame's N points: v[q] holds point sl + SUB q; buf is the frame's exchange buffer.
template <int N, int R, int Ns, bool FIRST>
__device__ __forceinline__ void stockham_pass(f2 (&v)[Geo<N>::P], f2 *buf, const PassTw<N, R, Ns> &tw, int sl)
{
    constexpr int SUB = Geo<N>::SUB, P = Geo<N>::P, NB = N / R / SUB;
    static_assert(NB >= 1, "radix too large for the points a lane holds");
    if constexpr (!FIRST) {
        wave_lds_fence();
#pragma unroll
        for (int q = 0; q < P; ++q) v[q] = buf[padi(sl + SUB * q)];
        wave_lds_fence();
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int j = sl + SUB * b;
        const int k = j & (Ns - 1);
        f2 x[R];
#pragma unroll
        for (int t = 0; t < R; ++t) x[t] = v[b + t * NB];
        if constexpr (Ns > 1) {
#pragma unroll
            for (int t = 1; t < R; ++t) x[t] = cmul(x[t], tw.get(b, t));
        }
        dft<R>(x);
        const int o = (j - k) * R + k;
#pragma unroll
        for (int m = 0; m < R; ++m) buf[padi(o + m * Ns)] = x[m];
    }
}

// radix plans: 256 = 4.4.4.4, 512 = 8.8.8, 1024 = 16.4.16, 2048 = 16.8.16 (the small radix in the middle pass,
// where every butterfly of a lane shares its twiddles: fewer twiddle registers)
template <int LOG2N> struct Plan;
template <> struct Plan<6> { static constexpr int R0 = 4, R1 = 4, R2 = 4, R3 = 1; };      // 16 lanes per frame
template <> struct Plan<7> { static constexpr int R0 = 4, R1 = 4, R2 = 4, R3 = 2; };      // 32 lanes per frame
template <> struct Plan<8> { static constexpr int R0 = 4, R1 = 4, R2 = 4, R3 = 4; };
template <> struct Plan<9> { static constexpr int R0 = 8, R1 = 8, R2 = 8, R3 = 1; };
template <> struct Plan<10> { static constexpr int R0 = 16, R1 = 4, R2 = 16, R3 = 1; };
template <> struct Plan<11> { static constexpr int R0 = 16, R1 = 8, R2 = 16, R3 = 1; };

template <int LOG2N, int MODE>
__global__ __launch_bounds__(256, (LOG2N <= 10 ? 2 : 1)) void stft_sums_kernel(StftArgs a)
{
    constexpr int N = 1 << LOG2N;
    using G = Geo<N>;
    constexpr int SUB = G::SUB, P = G::P, FPW = G::FPW, NPAD = G::NPAD;
    using PL = Plan<LOG2N>;
    constexpr int R0 = PL::R0, R1 = PL::R1, R2 = PL::R2, R3 = PL::R3;
    static_assert(R0 * R1 * R2 * R3 == N, "radix plan");
    extern __shared__ f2 stft_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sl = lane & (SUB - 1), fs = lane / SUB;          // lane within the frame, frame slot within the wave
    f2 *buf = stft_smem + (wave * FPW + fs) * NPAD;            // this frame slot's exchange buffer

    PassTw<N, R1, R0> tw1;
    PassTw<N, R2, R0 * R1> tw2;
    PassTw<N, (R3 > 1 ? R3 : 2), R0 * R1 * R2 / (R3 > 1 ? 1 : 2)> tw3;      // (unused when R3 == 1)
    tw1.init(sl);
    tw2.init(sl);
    if constexpr (R3 > 1) tw3.init(sl);

    // window value of this lane's points n = sl + SUB q  (periodic Hann of `win` samples, centred in n_fft)
    const int left = (N - a.win) / 2;
    float wreg[P];
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const int n = sl + SUB * q - left;
        wreg[q] = (n >= 0 && n < a.win) ? 0.5f - 0.5f * cospif(2.0f * (float)n / (float)a.win) : 0.0f;
    }

    const int64_t stream = blockIdx.x / a.chunks;
    const int chunk = blockIdx.x % a.chunks;
    const int L = (int)(a.T - a.skip);
    const float *ys = a.y + stream * a.T + a.skip;
    const float *ts = a.t + stream * a.T + a.skip;
    const int f_begin = chunk * a.frames_per_chunk;
    const int f_end = min(f_begin + a.frames_per_chunk, a.n_frames);
    constexpr int FSTEP = 4 * FPW;                             // frames per workgroup and iteration

    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const float eps4 = 4.0f * a.eps;                           // floor of the unscaled (4 x) powers, see split_spectra
    // raw samples of the frame pair, fetched one frame ahead: ry/rt[q] = (y, t)[f hop + sl + SUB q - N/2],
    // reflected at the ends (torch.stft center=True, pad_mode="reflect"); frame slots past the end read frame 0
    float ry[P], rt[P];
    auto fetch = [&](int f) {
        if (f >= f_end) f = f_begin;                        // idle slot of a multi-frame wave: harmless reload
        const int start = f * a.hop - N / 2;                // (T - skip < 2^31 - n_fft: checked by the API)
        if (FPW == 1 && start >= 0 && start + N <= L) {     // interior frame (wave-uniform): no index math per load
            const float *py = ys + start + sl, *pt = ts + start + sl;
#pragma unroll
            for (int q = 0; q < P; ++q) { ry[q] = py[SUB * q]; rt[q] = pt[SUB * q]; }
        } else {
#pragma unroll
            for (int q = 0; q < P; ++q) {
                int i = start + sl + SUB * q;
                i = i < 0 ? -i : i;
                i = i >= L ? 2 * (L - 1) - i : i;
                ry[q] = ys[i]; rt[q] = ts[i];
            }
        }
    };
    const int f_first = f_begin + wave * FPW;                  // this wave's first frame (slot 0)
    if (f_first < f_end) fetch(f_first + fs);
    for (int f0 = f_first; f0 < f_end; f0 += FSTEP) {
        const int f = f0 + fs;
        const bool live = f < f_end;                           // (only a multi-frame wave can have idle slots)
        f2 v[P];
#pragma unroll
        for (int q = 0; q < P; ++q) v[q] = (f2){wreg[q] * ry[q], wreg[q] * rt[q]};
        // ---- Stockham autosort FFT, the frame slot's own LDS buffer between passes ----
        {
            PassTw<N, R0, 1> tw0;
            stockham_pass<N, R0, 1, true>(v, buf, tw0, sl);
        }
        stockham_pass<N, R1, R0, false>(v, buf, tw1, sl);
        // the next frame's samples are fetched here, when the biggest butterflies are done and their registers are
        // free again (fetching at the top of the frame cost n_fft = 2048 spills and 0.7 ms)
        if (f0 + FSTEP < f_end) fetch(f + FSTEP);
        stockham_pass<N, R2, R0 * R1, false>(v, buf, tw2, sl);
        if constexpr (R3 > 1) stockham_pass<N, R3, R0 * R1 * R2, false>(v, buf, tw3, sl);
        wave_lds_fence();
        // ---- separate the two real spectra; distance terms for bins k = 0 .. N/2 ----
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
        auto bin = [&](int k) {
            const f2 zk = buf[padi(k)], zn = buf[padi((N - k) & (N - 1))];
            f2 Y2, T2;
            split_spectra(zk, zn, Y2, T2);
            const f2 ysq = Y2 * Y2, tsq = T2 * T2;
            const float py0 = ysq.x + ysq.y, pt0 = tsq.x + tsq.y;                 // 4 x the powers
            const float py = fmaxf(py0, eps4), pt = fmaxf(pt0, eps4);
            if constexpr (MODE == 0) {
                // auraloss.freq.STFTLoss terms: magnitudes sqrt(clamp(power, eps))
                const float my = __builtin_amdgcn_sqrtf(py), mt = __builtin_amdgcn_sqrtf(pt);
                const float d = mt - my;
                s0 += d * d;
                s1 += pt;
                s2 += fabsf(__builtin_amdgcn_logf(py) - __builtin_amdgcn_logf(pt));   // log2 of the POWERS
                s3 += fabsf(d);
            } else {
                // power-spectrogram terms (torchaudio Spectrogram(power=2) as used by code/evaluation.py:75-84):
                // |P_y - P_t|, |log2 max(P_y, floor) - log2 max(P_t, floor)| (floor = eps), P_t, P_y
                s0 += fabsf(py0 - pt0);
                s1 += fabsf(__builtin_amdgcn_logf(py) - __builtin_amdgcn_logf(pt));
                s2 += pt0;
                s3 += py0;
            }
        };
        if constexpr (MODE != 2) {
            if (live) {
#pragma unroll
                for (int i = 0; i < P / 2; ++i) bin(sl + SUB * i);
                if (sl == 0) bin(N / 2);
            }
        } else {
            // mel entries of code/evaluation.py:86-92: mel = mel_basis @ power spectrogram (code/utilities/utilities.py:
            // 666), per frame.  The two power spectra (N/2 + 1 bins each) take the frame's exchange buffer over -- every
            // lane first reads its bins out of it -- and lane m then walks the bins of filters m, m + 64, ... (a triangle
            // spans 2 .. ~50 bins).  One frame per wave here (n_fft >= 256).
            static_assert(MODE != 2 || FPW == 1, "mel mode: one frame per wave");
            float py_[P / 2], pt_[P / 2], pyn = 0.0f, ptn = 0.0f;
            auto power = [&](int k, float &py0, float &pt0) {
                const f2 zk = buf[padi(k)], zn = buf[padi((N - k) & (N - 1))];
                f2 Y2, T2;
                split_spectra(zk, zn, Y2, T2);
                const f2 ysq = Y2 * Y2, tsq = T2 * T2;
                py0 = ysq.x + ysq.y;                                              // 4 x the powers
                pt0 = tsq.x + tsq.y;
            };
#pragma unroll
            for (int i = 0; i < P / 2; ++i) power(sl + SUB * i, py_[i], pt_[i]);
            if (sl == 0) power(N / 2, pyn, ptn);
            wave_lds_fence();                                   // every lane has read its bins: the buffer is free
            float *pw = reinterpret_cast<float *>(buf);         // [0 .. N/2] P_y, [N/2 + 1 .. N + 1] P_t
#pragma unroll
            for (int i = 0; i < P / 2; ++i) { pw[sl + SUB * i] = py_[i]; pw[N / 2 + 1 + sl + SUB * i] = pt_[i]; }
            if (sl == 0) { pw[N / 2] = pyn; pw[N + 1] = ptn; }
            wave_lds_fence();
            if (live) {
                for (int m = sl; m < a.n_mels; m += 64) {
                    const int b0 = a.mel_first[m], w0 = a.mel_start[m], nw = a.mel_start[m + 1] - w0;
                    float my = 0.0f, mt = 0.0f;
                    for (int q = 0; q < nw; ++q) {
                        const float wq = a.mel_w[w0 + q];
                        my = __builtin_fmaf(wq, pw[b0 + q], my);
                        mt = __builtin_fmaf(wq, pw[N / 2 + 1 + b0 + q], mt);
                    }
                    s0 += fabsf(my - mt);
                    s1 += fabsf(__builtin_amdgcn_logf(fmaxf(my, eps4)) - __builtin_amdgcn_logf(fmaxf(mt, eps4)));
                    s2 += mt;
                    s3 += my;
                }
            }
        }
        acc[0] += s0; acc[1] += s1; acc[2] += s2; acc[3] += s3;
        wave_lds_fence();
    }
    // mode 0: |ln mag_y - ln mag_t| = (ln 2 / 2) |log2 p_y - log2 p_t|;  mode 1: log10 = log10(2) log2
    // undo the factor 4 of the powers (2 of the magnitudes); the log differences do not carry it
    if constexpr (MODE == 0) { acc[0] *= 0.25; acc[1] *= 0.25; acc[3] *= 0.5; }
    else { acc[0] *= 0.25; acc[2] *= 0.25; acc[3] *= 0.25; }
    if constexpr (MODE == 0) acc[2] *= 0.34657359027997264;
    else acc[1] *= 0.30102999566398120;                       // modes 1 and 2: log10 = log10(2) log2
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        double s = acc[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) a.out[(((size_t)stream * a.chunks + chunk) * 4 + wave) * 4 + c] = s;
    }
}

template <int LOG2N, int MODE>
static hipError_t launch_mode(const StftArgs &a, hipStream_t stream)
{
    constexpr int N = 1 << LOG2N;
    const size_t smem = (size_t)4 * Geo<N>::FPW * Geo<N>::NPAD * sizeof(f2);
    auto k = stft_sums_kernel<LOG2N, MODE>;
    hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3((unsigned)(a.B * a.chunks)), dim3(256), smem, stream, a);
    return hipGetLastError();
}

template <int LOG2N>
static hipError_t launch_one(const StftArgs &a, hipStream_t stream)
{
    if constexpr (LOG2N >= 10) {             // the mel projection is compiled for the frame sizes it is used with
        if (a.mode == 2) return launch_mode<LOG2N, 2>(a, stream);
    }
    if (a.mode == 2) return hipErrorInvalidValue;
    return a.mode == 0 ? launch_mode<LOG2N, 0>(a, stream) : launch_mode<LOG2N, 1>(a, stream);
}

hipError_t launch_stft_sums(const float *y, const float *t, int64_t B, int64_t T, int64_t skip, int n_fft, int hop,
                            int win, float eps, int chunks, int mode, double *out, hipStream_t stream, int n_mels,
                            const int *mel_first, const int *mel_start, const float *mel_w)
{
    StftArgs a;
    a.mode = mode;
    a.n_mels = n_mels; a.mel_first = mel_first; a.mel_start = mel_start; a.mel_w = mel_w;
    a.y = y; a.t = t; a.B = B; a.T = T; a.skip = skip; a.hop = hop; a.win = win; a.chunks = chunks; a.eps = eps; a.out = out;
    a.n_frames = (int)(1 + (T - skip) / hop);
    a.frames_per_chunk = (a.n_frames + chunks - 1) / chunks;
    switch (n_fft) {
    case 64: return launch_one<6>(a, stream);
    case 128: return launch_one<7>(a, stream);
    case 256: return launch_one<8>(a, stream);
    case 512: return launch_one<9>(a, stream);
    case 1024: return launch_one<10>(a, stream);
    case 2048: return launch_one<11>(a, stream);
    default: return hipErrorInvalidValue;
    }
}

}   // namespace ntm
