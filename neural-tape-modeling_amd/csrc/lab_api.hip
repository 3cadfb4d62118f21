// extern "C" surface of libntm_lab.so (declared in include/ntm_lab.h): the LABORATORY -- older and experimental
// exact-fp32 GRU kernels kept as independent implementations for the parity tests and as measured dead ends, and
// the diagnostic builds of the product kernel (s_memtime stamps, timing ablations).  Nothing in the product path
// (libntm.so, neural-tape-modeling_amd/model.py with a product kernel_variant) loads this library.
#include "ntm_lab.h"
#include "ntm_common.h"

#include <cstdlib>
#include <string>

namespace {
thread_local std::string g_err;
int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
int hip_fail(hipError_t e, const char *where) { return fail(NTM_EHIP, std::string(where) + ": " + hipGetErrorString(e)); }
}  // namespace

extern "C" {

const char *ntm_lab_last_error(void) { return g_err.c_str(); }

int ntm_lab_gru_forward(const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh, const float *w_o,
                        const float *b_o, int H, const float *x, float *y, int64_t B, int64_t T, int64_t x_stride_b,
                        int64_t y_stride_b, float *h_state, int variant, void *stream)
{
    if (H != NTM_HIDDEN) return fail(NTM_EINVAL, "ntm_lab_gru_forward: the laboratory kernels are compiled for hidden size 64");
    if (B < 0 || T < 0) return fail(NTM_EINVAL, "ntm_lab_gru_forward: negative B or T");
    if (B == 0 || T == 0) return NTM_OK;
    if (!w_ih || !w_hh || !b_ih || !b_hh || !w_o || !x || !y) return fail(NTM_EINVAL, "ntm_lab_gru_forward: null pointer");
    if (x_stride_b < T || y_stride_b < T) return fail(NTM_EINVAL, "ntm_lab_gru_forward: stride < T");
    if (variant == NTM_GRU_VALU && (reinterpret_cast<uintptr_t>(w_hh) & 15))
        return fail(NTM_EINVAL, "ntm_lab_gru_forward: NTM_GRU_VALU reads W_hh with 16-byte loads; w_hh must be 16-byte aligned");
    ntm::GruArgs a{w_ih, w_hh, b_ih, b_hh, w_o, b_o, x, y, h_state, B, T, x_stride_b, y_stride_b, nullptr, 0, 0};
    hipError_t e;
    switch (variant) {
        case NTM_GRU_MFMA: e = ntm::launch_gru_mfma(a, (hipStream_t)stream); break;
        case NTM_GRU_VALU: e = ntm::launch_gru_valu(a, (hipStream_t)stream); break;
        default: return fail(NTM_EINVAL, "ntm_lab_gru_forward: not a laboratory variant (NTM_GRU_MFMA, _VALU)");
    }
    return e == hipSuccess ? NTM_OK : hip_fail(e, "ntm_lab_gru_forward");
}

int ntm_debug_gru_stamps(const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh, const float *w_o,
                         const float *b_o, const float *x, float *y, int64_t B, int64_t T, float *h_state,
                         uint64_t *stamps, int variant, void *stream)
{
    if (!stamps || !x || !y || B <= 0 || T <= 0) return fail(NTM_EINVAL, "ntm_debug_gru_stamps: bad argument");
    ntm::GruArgs a{w_ih, w_hh, b_ih, b_hh, w_o, b_o, x, y, h_state, B, T, T, T, (unsigned long long *)stamps, 0, 0};
    if (getenv("NTM_LAB_STAMP_ESR")) {          // diagnostic: the stamped build of the forward + ESR-sums variant (x as target)
        static double *scratch = nullptr;
        static int64_t cap = 0;
        if (cap < B) {
            if (scratch) (void)hipFree(scratch);
            if (hipMalloc(&scratch, (size_t)B * 2 * sizeof(double)) != hipSuccess) return fail(NTM_EHIP, "ntm_debug_gru_stamps: hipMalloc");
            cap = B;
        }
        a.tgt = x;
        a.esr_out = scratch;
        a.esr_skip = 0;
    }
    if (variant == NTM_GRU_BF16X3) a.engine = 2;          // the stamped build of the bf16x3 step order
    hipError_t e = variant == NTM_GRU_MFMA ? ntm::launch_gru_mfma(a, (hipStream_t)stream)
                                           : ntm::launch_gru_mfma2(a, (hipStream_t)stream);
    return e == hipSuccess ? NTM_OK : hip_fail(e, "ntm_debug_gru_stamps");
}

int ntm_debug_gru_ablate(const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh, const float *w_o,
                         const float *b_o, const float *x, float *y, int64_t B, int64_t T, float *h_state, int mask,
                         void *stream)
{
    if (!x || !y || B <= 0 || T <= 0 || mask <= 0) return fail(NTM_EINVAL, "ntm_debug_gru_ablate: bad argument");
    ntm::GruArgs a{w_ih, w_hh, b_ih, b_hh, w_o, b_o, x, y, h_state, B, T, T, T, nullptr, mask, 0};
    hipError_t e = ntm::launch_gru_mfma2(a, (hipStream_t)stream);
    return e == hipSuccess ? NTM_OK : hip_fail(e, "ntm_debug_gru_ablate");
}

int ntm_debug_transpose4(const float *in, float *out, void *stream)
{
    if (!in || !out) return fail(NTM_EINVAL, "ntm_debug_transpose4: null pointer");
    hipError_t e = ntm::launch_debug_transpose(in, out, (hipStream_t)stream);
    return e == hipSuccess ? NTM_OK : hip_fail(e, "ntm_debug_transpose4");
}

}  // extern "C"
