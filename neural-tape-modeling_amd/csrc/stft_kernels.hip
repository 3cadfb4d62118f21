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
// T[k] = (Z[k] - conj Z[N-k])/(2i)).  The FFT is a Stockham autosort radix-4 (plus one radix-2 pass when
// log2 n_fft is odd); each lane holds n_fft/64 points in registers, passes exchange through the wave's own
// LDS buffer, so no workgroup barrier is needed inside the frame loop (LDS operations of one wave are
// processed in issue order).  The window values and their positions are per-lane constants kept in registers;
// twiddles come from an LDS table computed once per workgroup with sincospi.  A workgroup = 4 waves works on
// one (stream, frame chunk); sums are accumulated in fp64 per lane and written per (stream, chunk, wave) --
// the host adds them in a fixed order (deterministic, no atomics).
//
// Bound: VALU (about 5 n log2 n flop per frame pair); the signals are read once from HBM per resolution
// (frames overlap 5x in L1/L2), 8 B/sample algorithmic.
#include "ntm_common.h"

namespace ntm {

struct StftArgs {
    const float *y, *t;
    int64_t B, T, skip;
    int hop, win, chunks, frames_per_chunk, n_frames;
    float eps;
    double *out;
};

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 cmul(f2 a, f2 b) { return (f2){a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int LOG2N>
__global__ __launch_bounds__(256) void stft_sums_kernel(StftArgs a)
{
    constexpr int N = 1 << LOG2N, P = N / 64, NB4 = P / 4, NPASS4 = LOG2N / 2;
    constexpr bool ODD = (LOG2N & 1) != 0;
    extern __shared__ f2 stft_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f2 *tw = stft_smem;                       // [N]   exp(-2 pi i m / N)
    f2 *buf = stft_smem + N + wave * N;       // [N]   this wave's exchange buffer

    for (int m = tid; m < N; m += 256) {
        float s, c;
        sincospif(-2.0f * (float)m / (float)N, &s, &c);
        tw[m] = (f2){c, s};
    }
    // window value of this lane's points n = lane + 64 q  (periodic Hann of `win` samples, centred in n_fft)
    const int left = (N - a.win) / 2;
    float wreg[P];
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const int n = lane + 64 * q - left;
        wreg[q] = (n >= 0 && n < a.win) ? 0.5f - 0.5f * cospif(2.0f * (float)n / (float)a.win) : 0.0f;
    }
    __syncthreads();

    const int64_t stream = blockIdx.x / a.chunks;
    const int chunk = blockIdx.x % a.chunks;
    const int L = (int)(a.T - a.skip);
    const float *ys = a.y + stream * a.T + a.skip;
    const float *ts = a.t + stream * a.T + a.skip;
    const int f_begin = chunk * a.frames_per_chunk;
    const int f_end = min(f_begin + a.frames_per_chunk, a.n_frames);

    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int f = f_begin + wave; f < f_end; f += 4) {
        // ---- windowed frame pair into registers: v[q] = w[n] (y, t)[f hop + n - N/2], reflected at the ends ----
        f2 v[P];
        const int base = f * a.hop - N / 2 + lane;          // (T - skip < 2^31 - n_fft: checked by the API)
#pragma unroll
        for (int q = 0; q < P; ++q) {
            int i = base + 64 * q;
            i = i < 0 ? -i : i;
            i = i >= L ? 2 * (L - 1) - i : i;
            v[q] = (f2){wreg[q] * ys[i], wreg[q] * ts[i]};
        }
        // ---- radix-4 passes: butterfly b of this lane is j = lane + 64 b, inputs x[j + t N/4] = v[b + t NB4] ----
#pragma unroll
        for (int p = 0; p < NPASS4; ++p) {
            const int Ns = 1 << (2 * p);
            if (p > 0) {
                wave_lds_fence();
#pragma unroll
                for (int q = 0; q < P; ++q) v[q] = buf[lane + 64 * q];
                wave_lds_fence();
            }
#pragma unroll
            for (int b = 0; b < NB4; ++b) {
                const int j = lane + 64 * b;
                const int k = j & (Ns - 1);
                f2 x0 = v[b], x1 = v[b + NB4], x2 = v[b + 2 * NB4], x3 = v[b + 3 * NB4];
                if (p > 0) {
                    const f2 w1 = tw[k * (N / (4 * Ns))];
                    const f2 w2 = cmul(w1, w1), w3 = cmul(w1, w2);
                    x1 = cmul(x1, w1); x2 = cmul(x2, w2); x3 = cmul(x3, w3);
                }
                const f2 s02 = x0 + x2, d02 = x0 - x2, s13 = x1 + x3, d13 = x1 - x3;
                const f2 jd = (f2){d13.y, -d13.x};              // -i (x1 - x3)
                const int o = ((j - k) << 2) + k;
                buf[o] = s02 + s13;
                buf[o + Ns] = d02 + jd;
                buf[o + 2 * Ns] = s02 - s13;
                buf[o + 3 * Ns] = d02 - jd;
            }
        }
        if constexpr (ODD) {                                    // last pass radix 2: Ns = N/2, outputs in place
            wave_lds_fence();
#pragma unroll
            for (int q = 0; q < P; ++q) v[q] = buf[lane + 64 * q];
            wave_lds_fence();
#pragma unroll
            for (int b = 0; b < P / 2; ++b) {
                const int j = lane + 64 * b;
                const f2 x0 = v[b], x1 = cmul(v[b + P / 2], tw[j]);
                buf[j] = x0 + x1;
                buf[j + N / 2] = x0 - x1;
            }
        }
        wave_lds_fence();
        // ---- separate the two real spectra, magnitudes, distance terms for bins k = 0 .. N/2 ----
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
        auto bin = [&](int k) {
            const f2 zk = buf[k], zn = buf[(N - k) & (N - 1)];
            const float yr = 0.5f * (zk.x + zn.x), yi = 0.5f * (zk.y - zn.y);
            const float tr = 0.5f * (zk.y + zn.y), ti = 0.5f * (zn.x - zk.x);
            const float py = fmaxf(yr * yr + yi * yi, a.eps), pt = fmaxf(tr * tr + ti * ti, a.eps);
            const float my = __builtin_amdgcn_sqrtf(py), mt = __builtin_amdgcn_sqrtf(pt);
            const float d = mt - my;
            s0 += d * d;
            s1 += pt;
            s2 += fabsf(__builtin_amdgcn_logf(py) - __builtin_amdgcn_logf(pt));   // log2 of the POWERS
            s3 += fabsf(d);
        };
#pragma unroll
        for (int i = 0; i < P / 2; ++i) bin(lane + 64 * i);
        if (lane == 0) bin(N / 2);
        acc[0] += s0; acc[1] += s1; acc[2] += s2; acc[3] += s3;
        wave_lds_fence();
    }
    // |ln mag_y - ln mag_t| = (ln 2 / 2) |log2 p_y - log2 p_t|
    acc[2] *= 0.34657359027997264;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        double s = acc[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) a.out[(((size_t)stream * a.chunks + chunk) * 4 + wave) * 4 + c] = s;
    }
}

template <int LOG2N>
static hipError_t launch_one(const StftArgs &a, hipStream_t stream)
{
    constexpr int N = 1 << LOG2N;
    const size_t smem = (size_t)5 * N * sizeof(f2);
    auto k = stft_sums_kernel<LOG2N>;
    hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3((unsigned)(a.B * a.chunks)), dim3(256), smem, stream, a);
    return hipGetLastError();
}

hipError_t launch_stft_sums(const float *y, const float *t, int64_t B, int64_t T, int64_t skip, int n_fft, int hop,
                            int win, float eps, int chunks, double *out, hipStream_t stream)
{
    StftArgs a;
    a.y = y; a.t = t; a.B = B; a.T = T; a.skip = skip; a.hop = hop; a.win = win; a.chunks = chunks; a.eps = eps; a.out = out;
    a.n_frames = (int)(1 + (T - skip) / hop);
    a.frames_per_chunk = (a.n_frames + chunks - 1) / chunks;
    switch (n_fft) {
    case 256: return launch_one<8>(a, stream);
    case 512: return launch_one<9>(a, stream);
    case 1024: return launch_one<10>(a, stream);
    case 2048: return launch_one<11>(a, stream);
    default: return hipErrorInvalidValue;
    }
}

}   // namespace ntm
