// Shared device helpers for the libntm.so kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ntm {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kH = 64;  // hidden size of the matrix-pipe / low-latency kernels (NTM_HIDDEN); gru_small.hip: 8, 16, 32

// sigma(v) = 1/(1+e^-v) on v_exp_f32 / v_rcp_f32 (both ~1 ulp).  Saturates cleanly:
// e^-v -> inf gives 0, -> 0 gives 1.
__device__ __forceinline__ float sigmoid_f32(float v)
{
    const float e = __builtin_amdgcn_exp2f(v * -1.44269504088896340736f);
    return __builtin_amdgcn_rcpf(1.0f + e);
}

// tanh(v) = 1 - 2/(1+e^{2v}); abs error ~1e-7, exact limits +-1.
__device__ __forceinline__ float tanh_f32(float v)
{
    const float e = __builtin_amdgcn_exp2f(v * 2.88539008177792681472f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + e);
}

// Kernel arguments of every GRU variant (pointers are device pointers, strides in elements).
struct GruArgs {
    const float *w_ih, *w_hh, *b_ih, *b_hh, *w_o, *b_o;  // b_o may be null
    const float *x;
    float *y;
    float *h_state;  // [B,64] in/out, may be null
    int64_t B, T, xs, ys;
    unsigned long long *dbg;  // diagnostic stamp sums (ntm_debug_gru_stamps), else null
    int abl;                  // diagnostic ablation mask (ntm_debug_gru_ablate), else 0
    int engine;               // MFMA2 GEMV engine: 0 exact fp32, 1 split-fp16 x3 (NTM_GRU_F16X3), 2 split-bf16 x3 x3 (NTM_GRU_BF16X3)
    // fused DiffDelRNN step (gru_mfma2_kernel<FUSE>): `y` above is then pre_d, and the delay line writes yd
    const float *dd = nullptr;      // delay trajectory [B,T] in samples, contiguous
    float *yd = nullptr;            // delayed output [B,T], contiguous
    const float *dl_buf = nullptr;  // carried delay buffer [B,D] (read only here; delay_update_kernel moves it on)
    int32_t *dl_flag = nullptr;     // sticky range-violation flag (may be null)
    int D = 0;
    int warmup = 0;
    // predict + loss leg in one launch (gru_mfma2_kernel<ESR>): per-stream sums of (tgt - y)^2 and tgt^2 over [esr_skip, T)
    const float *tgt = nullptr;     // target [B,T], contiguous
    double *esr_out = nullptr;      // [B,2] fp64
    int64_t esr_skip = 0;           // multiple of 4
    double *dcp_out = nullptr;      // gru_mfma2_kernel<ESR, DCP>: [B,2] fp64 DC-pre-emphasised sums (non-null selects DCP)
    float dcp_R = 0.995f;           // pole of the DC blocker
    int H = 64;                     // hidden size as the caller's tensors have it (gru_small.hip: any size but 64; the
                                    // matrix-pipe / low-latency kernels are compiled for kH)
};

}  // namespace ntm

namespace ntm {
// compute units of the current device (256 on MI355X); the launch heuristics count stream groups against it
inline int device_cus()
{
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
        return n;
    return 256;
}

hipError_t launch_gru_mfma(const GruArgs &a, hipStream_t stream);
hipError_t launch_gru_valu(const GruArgs &a, hipStream_t stream);
hipError_t launch_gru_mfma2(const GruArgs &a, hipStream_t stream);
hipError_t launch_gru_mfma2_fused(const GruArgs &a, hipStream_t stream);   // GRU + head + delay line in one launch
hipError_t launch_gru_lat(const GruArgs &a, hipStream_t stream);
hipError_t launch_gru_small(const GruArgs &a, int H, hipStream_t stream);   // any H in [1, 1024] but 64
hipError_t launch_gru_io(const GruArgs &a, int H, int I, int O, hipStream_t stream);   // any input_size / output_size (gru_small.hip)
hipError_t launch_debug_transpose(const float *in, float *out, hipStream_t stream);
}  // namespace ntm
