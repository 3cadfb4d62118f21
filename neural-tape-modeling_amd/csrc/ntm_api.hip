// extern "C" surface of libntm.so (declared in include/ntm.h): argument checking, variant choice,
// error reporting.  No allocation, no synchronisation, no global mutable state besides the
// thread-local error string and the per-device pool of two side streams ntm_tcn_forward keeps for chunked batches.
#include "ntm.h"
#include "ntm_common.h"

#include <mutex>
#include <string>

namespace ntm {
hipError_t launch_delay(const float *x, const float *d, float *y, int64_t B, int64_t T, float *dl_state, int D,
                        int warmup, int32_t *err_flag, hipStream_t stream);
hipError_t launch_delay_apply(const float *x, const float *d, float *y, int64_t B, int64_t T, const float *dl_state, int D,
                              int warmup, int32_t *err_flag, hipStream_t stream);
hipError_t launch_delay_update(const float *x, int64_t B, int64_t T, float *dl_state, int D, const int32_t *err_flag,
                               hipStream_t stream);
hipError_t launch_esr(const float *y, const float *t, int64_t B, int64_t T, int64_t skip, int splits, double *out,
                      hipStream_t stream);
int esr_default_splits(int64_t B, int64_t T, int64_t skip);
hipError_t launch_loss_scalars(const double *rows, int64_t B, double n, double eps, double *out4, hipStream_t stream);
hipError_t launch_esr_dcpre(const float *y, const float *t, int64_t B, int64_t T, int64_t skip, float R, double *out,
                            hipStream_t stream);
hipError_t launch_stft_sums(const float *y, const float *t, int64_t B, int64_t T, int64_t skip, int n_fft, int hop,
                            int win, float eps, int chunks, int mode, double *out, hipStream_t stream, int n_mels = 0,
                            const int *mel_first = nullptr, const int *mel_start = nullptr, const float *mel_w = nullptr);
hipError_t launch_demodulate(const float *x, float *out, int C, int64_t N, const int64_t *y_idx, int P, int64_t period,
                             int64_t shift, double *scratch, hipStream_t stream);
hipError_t launch_tape_record_field(const double *I, const double *bias, double *H, int64_t B, int64_t N, double gain,
                                    double gap, hipStream_t stream);
hipError_t launch_tape_hmag(const double *H, double *M, int64_t B, int64_t N, double *state, double Ts, const double *par,
                            hipStream_t stream);
hipError_t launch_resample_fir(const double *x, double *y, int64_t B, int64_t N, int64_t M, int up, int down, int width,
                               const double *ker, hipStream_t stream);
hipError_t launch_fir_f64(const double *x, double *y, int64_t B, int64_t N, const double *h, int taps, int clamp, hipStream_t stream);
hipError_t launch_tcn(const float *params, int L, int C, int K, const int *dil, const float *x, float *y, int64_t B,
                      int64_t T, float *scratch, hipStream_t stream);
}  // namespace ntm

namespace {
thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
int hip_fail(hipError_t e, const char *where)
{
    return fail(NTM_EHIP, std::string(where) + ": " + hipGetErrorString(e));
}
}  // namespace

extern "C" {

int ntm_abi_version(void) { return NTM_ABI_VERSION; }

const char *ntm_last_error(void) { return g_err.c_str(); }

int ntm_gru_forward_io(const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh, const float *w_o,
                       const float *b_o, int H, int I, int O, const float *x, float *y, int64_t B, int64_t T,
                       int64_t x_stride_b, int64_t y_stride_b, float *h_state, void *stream)
{
    if (H < 1 || H > NTM_MAX_HIDDEN) return fail(NTM_EINVAL, "ntm_gru_forward_io: hidden size must lie in [1, 1024]");
    if (I < 1 || I > 1024 || O < 1 || O > 1024) return fail(NTM_EINVAL, "ntm_gru_forward_io: input_size and output_size must lie in [1, 1024]");
    if (B < 0 || T < 0) return fail(NTM_EINVAL, "ntm_gru_forward_io: negative B or T");
    if (B == 0 || T == 0) return NTM_OK;
    if (!w_ih || !w_hh || !b_ih || !b_hh || !w_o || !x || !y) return fail(NTM_EINVAL, "ntm_gru_forward_io: null pointer");
    if (x_stride_b < T * I || y_stride_b < T * O) return fail(NTM_EINVAL, "ntm_gru_forward_io: row stride below T * size");
    if (B > 0x7fffffff) return fail(NTM_EINVAL, "ntm_gru_forward_io: at most 2^31 - 1 streams per call");
    if (x == y) return fail(NTM_EINVAL, "ntm_gru_forward_io: y must not alias x");
    ntm::GruArgs a{w_ih, w_hh, b_ih, b_hh, w_o, b_o, x, y, h_state, B, T, x_stride_b, y_stride_b, nullptr, 0, 0};
    hipError_t e = ntm::launch_gru_io(a, H, I, O, (hipStream_t)stream);
    return e == hipSuccess ? NTM_OK : hip_fail(e, "ntm_gru_forward_io");
}

int ntm_gru_forward_ex(const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh, const float *w_o,
                       const float *b_o, int H, const float *x, float *y, int64_t B, int64_t T, int64_t x_stride_b,
                       int64_t y_stride_b, float *h_state, int variant, void *stream)
{
    if (H < 1 || H > NTM_MAX_HIDDEN) return fail(NTM_EINVAL, "ntm_gru_forward: hidden size must lie in [1, 1024]");
    if (B < 0 || T < 0) return fail(NTM_EINVAL, "ntm_gru_forward: negative B or T");
    if (B == 0 || T == 0) return NTM_OK;
    if (!w_ih || !w_hh || !b_ih || !b_hh || !w_o || !x || !y) return fail(NTM_EINVAL, "ntm_gru_forward: null pointer");
    if (x_stride_b < T || y_stride_b < T) return fail(NTM_EINVAL, "ntm_gru_forward: stride < T");
    if (H != NTM_HIDDEN) {
        // every other hidden size (the reference's constructor / training defaults 8 and 16, code/model.py:22,
        // code/train.py:50, and whatever --HIDDEN_SIZE a user trained with): gru_small.hip -- 64/HP streams per wavefront at
        // the next power of two HP for H < 64, a workgroup per stream above; the matrix-pipe variants exist for H = 64 only
        if (variant != NTM_GRU_AUTO && variant != NTM_GRU_LAT && variant != NTM_GRU_VALU)
            return fail(NTM_EINVAL, "ntm_gru_forward: the matrix-pipe kernel variants are compiled for hidden size 64 only");
        if (H > NTM_HIDDEN && B > 0x7fffffff) return fail(NTM_EINVAL, "ntm_gru_forward: at most 2^31 - 1 streams per call for H > 64");
        ntm::GruArgs as{w_ih, w_hh, b_ih, b_hh, w_o, b_o, x, y, h_state, B, T, x_stride_b, y_stride_b, nullptr, 0, 0};
        hipError_t es = ntm::launch_gru_small(as, H, (hipStream_t)stream);
        return es == hipSuccess ? NTM_OK : hip_fail(es, "ntm_gru_forward");
    }
    if (variant == NTM_GRU_VALU && (reinterpret_cast<uintptr_t>(w_hh) & 15))
        return fail(NTM_EINVAL, "ntm_gru_forward: NTM_GRU_VALU reads W_hh with 16-byte loads; w_hh must be 16-byte aligned");
    ntm::GruArgs a{w_ih, w_hh, b_ih, b_hh, w_o, b_o, x, y, h_state, B, T, x_stride_b, y_stride_b, nullptr, 0, 0};
    hipError_t e;
    if (variant == NTM_GRU_AUTO) {
        // B <= 1024: the low-latency kernel.  Otherwise the matrix-pipe kernel; when B is a whole number of full
        // device rounds (16 streams x CUs) plus a remainder the low-latency kernel can take, the remainder goes there
        // instead of opening another round of workgroups (B = 4112: 5.4 ms instead of 7.2 per 4096 steps).
        const int64_t round = 16 * (int64_t)ntm::device_cus();
        const int64_t full = (B / round) * round, rem = B - full;
        if (B <= NTM_GRU_LAT_MAX_B) variant = NTM_GRU_LAT;
        else if (full > 0 && rem > 0 && rem <= NTM_GRU_LAT_MAX_B) {
            ntm::GruArgs a0 = a, a1 = a;
            a0.B = full;
            a1.B = rem;
            a1.x = x + full * x_stride_b;
            a1.y = y + full * y_stride_b;
            a1.h_state = h_state ? h_state + full * NTM_HIDDEN : nullptr;
            e = ntm::launch_gru_mfma2(a0, (hipStream_t)stream);
            if (e == hipSuccess) e = ntm::launch_gru_lat(a1, (hipStream_t)stream);
            return e == hipSuccess ? NTM_OK : hip_fail(e, "ntm_gru_forward");
        } else variant = NTM_GRU_MFMA2;
    }
    switch (variant) {
        case NTM_GRU_MFMA2: e = ntm::launch_gru_mfma2(a, (hipStream_t)stream); break;
        case NTM_GRU_LAT: e = ntm::launch_gru_lat(a, (hipStream_t)stream); break;
        case NTM_GRU_F16X3: a.engine = 1; e = ntm::launch_gru_mfma2(a, (hipStream_t)stream); break;
        case NTM_GRU_BF16X3: a.engine = 2; e = ntm::launch_gru_mfma2(a, (hipStream_t)stream); break;
        case NTM_GRU_MFMA: case NTM_GRU_VALU:
            return fail(NTM_EINVAL, "ntm_gru_forward: laboratory kernel variant -- those live in libntm_lab.so "
                                    "(ntm_lab_gru_forward, include/ntm_lab.h), not in the product library");
        case NTM_GRU_MFMA3: case NTM_GRU_MFMA4:
            return fail(NTM_EINVAL, "ntm_gru_forward: kernel variant retired in round 6 (a measured negative result)");
        default: return fail(NTM_EINVAL, "ntm_gru_forward: unknown kernel variant");
    }
    return e == hipSuccess ? NTM_OK : hip_fail(e, "ntm_gru_forward");
}

// ntm_gru_forward_esr (dcp_out == NULL) and ntm_gru_forward_losses share this body
static int gru_losses_impl(const char *who, const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh, const float *w_o,
                           const float *b_o, int H, const float *x, float *y, int64_t B, int64_t T, int64_t x_stride_b,
                           int64_t y_stride_b, float *h_state, const float *target, int64_t skip, double *esr_out, float R,
                           double *dcp_out, void *stream)
{
    const std::string w(who);
    // every argument is checked BEFORE anything is enqueued: an NTM_EINVAL leaves y, h_state and the sums untouched
    if (H < 1 || H > NTM_MAX_HIDDEN) return fail(NTM_EINVAL, w + ": hidden size must lie in [1, 1024]");
    if (B < 0 || T < 0 || skip < 0 || skip > T) return fail(NTM_EINVAL, w + ": bad size");
    if (dcp_out && !(R >= 0.0f && R < 1.0f)) return fail(NTM_EINVAL, w + ": R must be in [0,1)");
    if (B == 0) return NTM_OK;
    if (!target || !esr_out) return fail(NTM_EINVAL, w + ": null pointer");
    if (T == 0) {                                // no samples: the sums are zero
        hipError_t ez = hipMemsetAsync(esr_out, 0, (size_t)B * 2 * sizeof(double), (hipStream_t)stream);
        if (ez == hipSuccess && dcp_out) ez = hipMemsetAsync(dcp_out, 0, (size_t)B * 2 * sizeof(double), (hipStream_t)stream);
        return ez == hipSuccess ? NTM_OK : hip_fail(ez, who);
    }
    if (!w_ih || !w_hh || !b_ih || !b_hh || !w_o || !x || !y) return fail(NTM_EINVAL, w + ": null pointer");
    if (x_stride_b < T || y_stride_b < T) return fail(NTM_EINVAL, w + ": stride < T");
    if (target == y) return fail(NTM_EINVAL, w + ": target must not alias y");
    // streams the matrix-pipe kernel takes (as ntm_gru_forward's NTM_GRU_AUTO decides): there the sums ride in the launch
    int64_t fused = 0;
    if (H == NTM_HIDDEN && B > NTM_GRU_LAT_MAX_B && (skip & 3) == 0) {
        const int64_t round = 16 * (int64_t)ntm::device_cus();
        const int64_t full = (B / round) * round, rem = B - full;
        fused = (full > 0 && rem > 0 && rem <= NTM_GRU_LAT_MAX_B) ? full : B;
    }
    if (fused < B && y_stride_b != T)
        return fail(NTM_EINVAL, w + ": the streaming loss passes (streams outside the matrix-pipe launch) need contiguous y rows (stride T)");
    if (fused > 0) {
        ntm::GruArgs a{w_ih, w_hh, b_ih, b_hh, w_o, b_o, x, y, h_state, fused, T, x_stride_b, y_stride_b, nullptr, 0, 0};
        a.tgt = target;
        a.esr_out = esr_out;
        a.esr_skip = skip;
        a.dcp_out = dcp_out;
        a.dcp_R = R;
        hipError_t e = ntm::launch_gru_mfma2(a, (hipStream_t)stream);
        if (e != hipSuccess) return hip_fail(e, who);
    }
    if (fused < B) {            // the rest: the forward launch the library would pick, then the streaming passes (one row per stream)
        const int64_t r = B - fused;
        int rc = ntm_gru_forward(w_ih, w_hh, b_ih, b_hh, w_o, b_o, H, x + fused * x_stride_b, y + fused * y_stride_b, r, T,
                                 x_stride_b, y_stride_b, h_state ? h_state + fused * H : nullptr, stream);
        if (rc != NTM_OK) return rc;
        hipError_t e = ntm::launch_esr(y + fused * T, target + fused * T, r, T, skip, 1, esr_out + 2 * fused, (hipStream_t)stream);
        if (e == hipSuccess && dcp_out)
            e = ntm::launch_esr_dcpre(y + fused * T, target + fused * T, r, T, skip, R, dcp_out + 2 * fused, (hipStream_t)stream);
        if (e != hipSuccess) return hip_fail(e, who);
    }
    return NTM_OK;
}

int ntm_gru_forward_esr(const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh, const float *w_o,
                        const float *b_o, int H, const float *x, float *y, int64_t B, int64_t T, int64_t x_stride_b,
                        int64_t y_stride_b, float *h_state, const float *target, int64_t skip, double *esr_out, void *stream)
{
    return gru_losses_impl("ntm_gru_forward_esr", w_ih, w_hh, b_ih, b_hh, w_o, b_o, H, x, y, B, T, x_stride_b, y_stride_b, h_state, target,
                           skip, esr_out, 0.0f, nullptr, stream);
}

int ntm_gru_forward_losses(const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh, const float *w_o,
                           const float *b_o, int H, const float *x, float *y, int64_t B, int64_t T, int64_t x_stride_b,
                           int64_t y_stride_b, float *h_state, const float *target, int64_t skip, double *esr_out, float dcpre_R,
                           double *dcpre_out, void *stream)
{
    if (!dcpre_out) return fail(NTM_EINVAL, "ntm_gru_forward_losses: null pointer");
    if (dcpre_out == esr_out) return fail(NTM_EINVAL, "ntm_gru_forward_losses: esr_out and dcpre_out must be distinct");
    return gru_losses_impl("ntm_gru_forward_losses", w_ih, w_hh, b_ih, b_hh, w_o, b_o, H, x, y, B, T, x_stride_b, y_stride_b, h_state,
                           target, skip, esr_out, dcpre_R, dcpre_out, stream);
}

int ntm_gru_forward(const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh, const float *w_o,
                    const float *b_o, int H, const float *x, float *y, int64_t B, int64_t T, int64_t x_stride_b,
                    int64_t y_stride_b, float *h_state, void *stream)
{
    return ntm_gru_forward_ex(w_ih, w_hh, b_ih, b_hh, w_o, b_o, H, x, y, B, T, x_stride_b, y_stride_b, h_state,
                              NTM_GRU_AUTO, stream);
}

int ntm_delay_forward(const float *x, const float *d, float *y, int64_t B, int64_t T, float *dl_state, int D,
                      int warmup, int32_t *err_flag, void *stream)
{
    if (B < 0 || T < 0 || D < 0) return fail(NTM_EINVAL, "ntm_delay_forward: negative size");
    if (B == 0 || T == 0) return NTM_OK;
    if (!x || !d || !y || (D > 0 && !dl_state)) return fail(NTM_EINVAL, "ntm_delay_forward: null pointer");
    if (x == y) return fail(NTM_EINVAL, "ntm_delay_forward: y must not alias x");
    hipError_t e = ntm::launch_delay(x, d, y, B, T, dl_state, D, warmup, err_flag, (hipStream_t)stream);
    return e == hipSuccess ? NTM_OK : hip_fail(e, "ntm_delay_forward");
}

static int diffdel_impl(const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh, const float *w_o, int H,
                        const float *x, const float *d, float *y, float *pre_d, int64_t B, int64_t T, float *h_state,
                        float *dl_state, int D, int warmup, int32_t *err_flag, int mode, const float *target, int64_t skip,
                        double *esr_out, void *stream, float dcp_R = 0.0f, double *dcp_out = nullptr)
{
    // dcp_out != NULL (ntm_diffdel_gru_forward_losses): also the DC-pre-emphasised sums, beside the ESR sums wherever those are formed
    // target != NULL: also the per-stream ESR sums of y against target over [skip, T) (ntm_diffdel_gru_forward_esr): inside the
    // fused launch where it runs and skip is a multiple of 4, by the streaming pass (one row per stream) everywhere else
    const bool esr_in_kernel = target && (skip & 3) == 0 && !warmup;
    // every argument is checked BEFORE anything is enqueued: an NTM_EINVAL leaves y, pre_d, the states and esr_out untouched
    if (B < 0 || T < 0 || D < 0) return fail(NTM_EINVAL, "ntm_diffdel_gru_forward: negative size");
    if (H < 1 || H > NTM_MAX_HIDDEN) return fail(NTM_EINVAL, "ntm_diffdel_gru_forward: hidden size must lie in [1, 1024]");
    if (B == 0) return NTM_OK;
    if (!pre_d || pre_d == y) return fail(NTM_EINVAL, "ntm_diffdel_gru_forward: pre_d must be a distinct buffer");
    if (mode != NTM_DIFFDEL_AUTO && mode != NTM_DIFFDEL_TWO_PASS && mode != NTM_DIFFDEL_FUSED)
        return fail(NTM_EINVAL, "ntm_diffdel_gru_forward: unknown mode");
    if (mode == NTM_DIFFDEL_FUSED && H != NTM_HIDDEN)
        return fail(NTM_EINVAL, "ntm_diffdel_gru_forward: the fused kernel is compiled for hidden size 64 only");
    if (target && T == 0) {            // no samples: the sums are zero
        hipError_t ez = hipMemsetAsync(esr_out, 0, (size_t)B * 2 * sizeof(double), (hipStream_t)stream);
        if (ez == hipSuccess && dcp_out) ez = hipMemsetAsync(dcp_out, 0, (size_t)B * 2 * sizeof(double), (hipStream_t)stream);
        return ez == hipSuccess ? NTM_OK : hip_fail(ez, "ntm_diffdel_gru_forward_esr");
    }
    // how many streams take the fused matrix-pipe kernel: all of them when forced; under AUTO the streams
    // ntm_gru_forward would give to that kernel (B > NTM_GRU_LAT_MAX_B; a remainder of at most that many streams behind
    // whole device rounds goes to the low-latency kernel + the streaming delay pass, as there)
    int64_t fused = 0;
    constexpr int64_t kFusedMaxT = (int64_t)1 << 26;     // the fused kernel addresses a 16-row block with 32-bit byte offsets
    if (mode == NTM_DIFFDEL_FUSED) {
        if (T >= kFusedMaxT) return fail(NTM_EINVAL, "ntm_diffdel_gru_forward: the fused kernel takes T < 2^26 samples per call");
        fused = B;
    } else if (mode == NTM_DIFFDEL_AUTO && H == NTM_HIDDEN && B > NTM_GRU_LAT_MAX_B && T < kFusedMaxT) {
        const int64_t round = 16 * (int64_t)ntm::device_cus();
        const int64_t full = (B / round) * round, rem = B - full;
        fused = (full > 0 && rem > 0 && rem <= NTM_GRU_LAT_MAX_B) ? full : B;
    }
    // A warm-up call (code/model.py:288-292: the delay line only moves its buffer on, y = pre_d) takes the two-pass form in
    // every mode: the fused kernel's delay stage is idle then and reads no delays, while the reference evaluates its range
    // assert BEFORE the warm-up branch (code/model.py:284 vs :288) -- delay_apply_kernel does, in warm-up too.
    if (warmup) fused = 0;
    if (fused > 0) {
        if (T == 0) return NTM_OK;
        if (!w_ih || !w_hh || !b_ih || !b_hh || !w_o || !x || !d || !y || (D > 0 && !dl_state))
            return fail(NTM_EINVAL, "ntm_diffdel_gru_forward: null pointer");
        if (x == y || x == pre_d || d == y || d == pre_d)
            return fail(NTM_EINVAL, "ntm_diffdel_gru_forward: outputs must not alias x or d");
        ntm::GruArgs a{w_ih, w_hh, b_ih, b_hh, w_o, nullptr, x, pre_d, h_state, fused, T, T, T, nullptr, 0, 0};
        a.dd = d;
        a.yd = y;
        a.dl_buf = dl_state;
        a.dl_flag = err_flag;
        a.D = D;
        a.warmup = warmup;
        if (esr_in_kernel) {
            a.tgt = target;
            a.esr_out = esr_out;
            a.esr_skip = skip;
            a.dcp_out = dcp_out;
            a.dcp_R = dcp_R;
        }
        hipError_t e = ntm::launch_gru_mfma2_fused(a, (hipStream_t)stream);
        if (e != hipSuccess) return hip_fail(e, "ntm_diffdel_gru_forward");
    }
    if (fused < B) {            // the streams the fused kernel did not take: GRU launch, then the streaming delay pass
        const int64_t r = B - fused, o = fused * T;
        // (a warm-up call under NTM_DIFFDEL_FUSED keeps the matrix-pipe GRU kernel the fused launch would have run, so that
        // the state a forced mode leaves behind does not depend on which calls were warm-ups)
        const int gv = (warmup && mode == NTM_DIFFDEL_FUSED) ? NTM_GRU_MFMA2 : NTM_GRU_AUTO;
        int rc = ntm_gru_forward_ex(w_ih, w_hh, b_ih, b_hh, w_o, nullptr, H, x + o, pre_d + o, r, T, T, T,
                                    h_state ? h_state + fused * H : nullptr, gv, stream);
        if (rc != NTM_OK) return rc;
        if (r <= 0 || T <= 0) return NTM_OK;
        if (!d || !y || (D > 0 && !dl_state)) return fail(NTM_EINVAL, "ntm_delay_forward: null pointer");
        if (fused == 0) {
            rc = ntm_delay_forward(pre_d, d, y, B, T, dl_state, D, warmup, err_flag, stream);
            if (rc != NTM_OK || !target) return rc;
            hipError_t ee = ntm::launch_esr(y, target, B, T, skip, 1, esr_out, (hipStream_t)stream);
            if (ee == hipSuccess && dcp_out) ee = ntm::launch_esr_dcpre(y, target, B, T, skip, dcp_R, dcp_out, (hipStream_t)stream);
            return ee == hipSuccess ? NTM_OK : hip_fail(ee, "ntm_diffdel_gru_forward_esr");
        }
        // mixed: interpolate the remainder here, then ONE buffer update over all streams (it reads the flag both parts raise)
        hipError_t e = ntm::launch_delay_apply(pre_d + o, d + o, y + o, r, T, dl_state ? dl_state + fused * D : nullptr, D, warmup,
                                               err_flag, (hipStream_t)stream);
        if (e != hipSuccess) return hip_fail(e, "ntm_diffdel_gru_forward");
    }
    hipError_t e = ntm::launch_delay_update(pre_d, B, T, dl_state, D, err_flag, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "ntm_diffdel_gru_forward");
    if (target) {               // the streams whose sums did not ride in the fused launch
        const int64_t from = esr_in_kernel ? fused : 0;
        if (from < B) {
            e = ntm::launch_esr(y + from * T, target + from * T, B - from, T, skip, 1, esr_out + 2 * from, (hipStream_t)stream);
            if (e == hipSuccess && dcp_out)
                e = ntm::launch_esr_dcpre(y + from * T, target + from * T, B - from, T, skip, dcp_R, dcp_out + 2 * from, (hipStream_t)stream);
            if (e != hipSuccess) return hip_fail(e, "ntm_diffdel_gru_forward_esr");
        }
    }
    return NTM_OK;
}

int ntm_diffdel_gru_forward_ex(const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                               const float *w_o, int H, const float *x, const float *d, float *y, float *pre_d,
                               int64_t B, int64_t T, float *h_state, float *dl_state, int D, int warmup,
                               int32_t *err_flag, int mode, void *stream)
{
    return diffdel_impl(w_ih, w_hh, b_ih, b_hh, w_o, H, x, d, y, pre_d, B, T, h_state, dl_state, D, warmup, err_flag, mode, nullptr, 0,
                        nullptr, stream);
}

int ntm_diffdel_gru_forward_esr(const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                                const float *w_o, int H, const float *x, const float *d, float *y, float *pre_d,
                                int64_t B, int64_t T, float *h_state, float *dl_state, int D, int32_t *err_flag,
                                const float *target, int64_t skip, double *esr_out, void *stream)
{
    if (!target || !esr_out) return fail(NTM_EINVAL, "ntm_diffdel_gru_forward_esr: null pointer");
    if (skip < 0 || skip > T) return fail(NTM_EINVAL, "ntm_diffdel_gru_forward_esr: bad skip");
    if (target == y || target == pre_d) return fail(NTM_EINVAL, "ntm_diffdel_gru_forward_esr: target must not alias an output");
    return diffdel_impl(w_ih, w_hh, b_ih, b_hh, w_o, H, x, d, y, pre_d, B, T, h_state, dl_state, D, 0, err_flag, NTM_DIFFDEL_AUTO, target,
                        skip, esr_out, stream);
}

int ntm_diffdel_gru_forward_losses(const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                                   const float *w_o, int H, const float *x, const float *d, float *y, float *pre_d,
                                   int64_t B, int64_t T, float *h_state, float *dl_state, int D, int32_t *err_flag,
                                   const float *target, int64_t skip, double *esr_out, float dcpre_R, double *dcpre_out, void *stream)
{
    if (!target || !esr_out || !dcpre_out) return fail(NTM_EINVAL, "ntm_diffdel_gru_forward_losses: null pointer");
    if (dcpre_out == esr_out) return fail(NTM_EINVAL, "ntm_diffdel_gru_forward_losses: esr_out and dcpre_out must be distinct");
    if (!(dcpre_R >= 0.0f && dcpre_R < 1.0f)) return fail(NTM_EINVAL, "ntm_diffdel_gru_forward_losses: R must be in [0,1)");
    if (skip < 0 || skip > T) return fail(NTM_EINVAL, "ntm_diffdel_gru_forward_losses: bad skip");
    if (target == y || target == pre_d) return fail(NTM_EINVAL, "ntm_diffdel_gru_forward_losses: target must not alias an output");
    return diffdel_impl(w_ih, w_hh, b_ih, b_hh, w_o, H, x, d, y, pre_d, B, T, h_state, dl_state, D, 0, err_flag, NTM_DIFFDEL_AUTO, target,
                        skip, esr_out, stream, dcpre_R, dcpre_out);
}

int ntm_diffdel_gru_forward(const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                            const float *w_o, int H, const float *x, const float *d, float *y, float *pre_d,
                            int64_t B, int64_t T, float *h_state, float *dl_state, int D, int warmup,
                            int32_t *err_flag, void *stream)
{
    return ntm_diffdel_gru_forward_ex(w_ih, w_hh, b_ih, b_hh, w_o, H, x, d, y, pre_d, B, T, h_state, dl_state, D, warmup,
                                      err_flag, NTM_DIFFDEL_AUTO, stream);
}

int ntm_esr_splits(int64_t B, int64_t T, int64_t skip) { return ntm::esr_default_splits(B, T, skip); }

int ntm_esr_sums(const float *y, const float *t, int64_t B, int64_t T, int64_t skip, int splits, double *out, void *stream)
{
    if (B < 0 || T < 0 || skip < 0 || skip > T) return fail(NTM_EINVAL, "ntm_esr_sums: bad size");
    if (splits < 1 || splits > 65535) return fail(NTM_EINVAL, "ntm_esr_sums: splits must be in [1, 65535]");
    if (B == 0) return NTM_OK;
    if (!y || !t || !out) return fail(NTM_EINVAL, "ntm_esr_sums: null pointer");
    hipError_t e = ntm::launch_esr(y, t, B, T, skip, splits, out, (hipStream_t)stream);
    return e == hipSuccess ? NTM_OK : hip_fail(e, "ntm_esr_sums");
}

int ntm_loss_scalars(const double *esr_rows, int64_t B, int64_t n_samples, double eps, double *out4, void *stream)
{
    if (B < 0 || n_samples <= 0 || !(eps >= 0.0)) return fail(NTM_EINVAL, "ntm_loss_scalars: bad size or eps");
    if (!out4 || (B > 0 && !esr_rows)) return fail(NTM_EINVAL, "ntm_loss_scalars: null pointer");
    hipError_t e = ntm::launch_loss_scalars(esr_rows, B, (double)n_samples, eps, out4, (hipStream_t)stream);
    return e == hipSuccess ? NTM_OK : hip_fail(e, "ntm_loss_scalars");
}

int ntm_esr_dcpre_sums(const float *y, const float *t, int64_t B, int64_t T, int64_t skip, float R, double *out,
                       void *stream)
{
    if (B < 0 || T < 0 || skip < 0 || skip > T) return fail(NTM_EINVAL, "ntm_esr_dcpre_sums: bad size");
    if (!(R >= 0.0f && R < 1.0f)) return fail(NTM_EINVAL, "ntm_esr_dcpre_sums: R must be in [0,1)");
    if (B == 0) return NTM_OK;
    if (!y || !t || !out) return fail(NTM_EINVAL, "ntm_esr_dcpre_sums: null pointer");
    hipError_t e = ntm::launch_esr_dcpre(y, t, B, T, skip, R, out, (hipStream_t)stream);
    return e == hipSuccess ? NTM_OK : hip_fail(e, "ntm_esr_dcpre_sums");
}

static int stft_common(const char *who, const float *y, const float *t, int64_t B, int64_t T, int64_t skip, int n_fft, int hop,
                       int win_length, float floor_, int chunks, int mode, double *out, void *stream, int n_mels = 0,
                       const int *mel_first = nullptr, const int *mel_start = nullptr, const float *mel_w = nullptr)
{
    const std::string w(who);
    if (B < 0 || T < 0 || skip < 0 || skip > T) return fail(NTM_EINVAL, w + ": bad size");
    if (n_fft != 64 && n_fft != 128 && n_fft != 256 && n_fft != 512 && n_fft != 1024 && n_fft != 2048)
        return fail(NTM_EINVAL, w + ": n_fft must be a power of two from 64 to 2048");
    if (hop <= 0 || win_length <= 0 || win_length > n_fft) return fail(NTM_EINVAL, w + ": bad hop or win_length");
    if (!(floor_ > 0.0f)) return fail(NTM_EINVAL, w + ": the power floor must be positive");
    if (chunks < 1 || B * (int64_t)chunks > 0x7fffffff) return fail(NTM_EINVAL, w + ": bad chunks");
    if (B == 0) return NTM_OK;
    if (T - skip <= n_fft / 2) return fail(NTM_EINVAL, w + ": reflect padding needs T - skip > n_fft/2");
    if (T - skip > 0x7fffffff - 4096) return fail(NTM_EINVAL, w + ": T - skip must be below 2^31 - 4096");
    if (!y || !t || !out) return fail(NTM_EINVAL, w + ": null pointer");
    hipError_t e = ntm::launch_stft_sums(y, t, B, T, skip, n_fft, hop, win_length, floor_, chunks, mode, out, (hipStream_t)stream,
                                         n_mels, mel_first, mel_start, mel_w);
    return e == hipSuccess ? NTM_OK : hip_fail(e, who);
}

int ntm_stft_sums(const float *y, const float *t, int64_t B, int64_t T, int64_t skip, int n_fft, int hop, int win_length,
                  float power_eps, int chunks, double *out, void *stream)
{
    return stft_common("ntm_stft_sums", y, t, B, T, skip, n_fft, hop, win_length, power_eps, chunks, 0, out, stream);
}

int ntm_spec_sums(const float *y, const float *t, int64_t B, int64_t T, int64_t skip, int n_fft, int hop, int win_length,
                  float log_floor, int chunks, double *out, void *stream)
{
    return stft_common("ntm_spec_sums", y, t, B, T, skip, n_fft, hop, win_length, log_floor, chunks, 1, out, stream);
}

int ntm_mel_sums(const float *y, const float *t, int64_t B, int64_t T, int64_t skip, int n_fft, int hop, int win_length,
                 float log_floor, int chunks, int n_mels, const int32_t *mel_first, const int32_t *mel_start, const float *mel_w,
                 double *out, void *stream)
{
    if (n_fft != 1024 && n_fft != 2048) return fail(NTM_EINVAL, "ntm_mel_sums: n_fft must be 1024 or 2048");
    if (n_mels < 1 || n_mels > 4096) return fail(NTM_EINVAL, "ntm_mel_sums: bad n_mels");
    if (B > 0 && (!mel_first || !mel_start || !mel_w)) return fail(NTM_EINVAL, "ntm_mel_sums: null filter-bank pointer");
    return stft_common("ntm_mel_sums", y, t, B, T, skip, n_fft, hop, win_length, log_floor, chunks, 2, out, stream, n_mels,
                       mel_first, mel_start, mel_w);
}

int ntm_copy2d_async(void *dst, int64_t dst_pitch_bytes, const void *src, int64_t src_pitch_bytes, int64_t width_bytes,
                     int64_t rows, int kind, void *stream)
{
    if (width_bytes < 0 || rows < 0 || dst_pitch_bytes < width_bytes || src_pitch_bytes < width_bytes)
        return fail(NTM_EINVAL, "ntm_copy2d_async: bad size or pitch");
    if (kind != 0 && kind != 1) return fail(NTM_EINVAL, "ntm_copy2d_async: kind must be 0 (H2D) or 1 (D2H)");
    if (width_bytes == 0 || rows == 0) return NTM_OK;
    if (!dst || !src) return fail(NTM_EINVAL, "ntm_copy2d_async: null pointer");
    hipError_t e = hipMemcpy2DAsync(dst, (size_t)dst_pitch_bytes, src, (size_t)src_pitch_bytes, (size_t)width_bytes,
                                    (size_t)rows, kind == 0 ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost,
                                    (hipStream_t)stream);
    return e == hipSuccess ? NTM_OK : hip_fail(e, "ntm_copy2d_async");
}

int ntm_demodulate(const float *x, float *out, int C, int64_t N, const int64_t *y_idx, int P, int64_t period, int64_t shift,
                   double *scratch, void *stream)
{
    if (C < 0 || N < 0) return fail(NTM_EINVAL, "ntm_demodulate: negative size");
    if (P < 2) return fail(NTM_EINVAL, "ntm_demodulate: at least two pulses are needed");
    if (period <= 0) return fail(NTM_EINVAL, "ntm_demodulate: pulse period must be positive");
    if (C == 0 || N == 0) return NTM_OK;
    if (N < 2) return fail(NTM_EINVAL, "ntm_demodulate: N must be at least 2");
    if (!x || !out || !y_idx || !scratch) return fail(NTM_EINVAL, "ntm_demodulate: null pointer");
    if (x == out) return fail(NTM_EINVAL, "ntm_demodulate: out must not alias x");
    hipError_t e = ntm::launch_demodulate(x, out, C, N, y_idx, P, period, shift, scratch, (hipStream_t)stream);
    return e == hipSuccess ? NTM_OK : hip_fail(e, "ntm_demodulate");
}

int ntm_tape_record_field(const double *I, const double *bias, double *H, int64_t B, int64_t N, double gain, double gap,
                          void *stream)
{
    if (B < 0 || N < 0 || !(gap != 0.0)) return fail(NTM_EINVAL, "ntm_tape_record_field: bad size or gap");
    if (B == 0 || N == 0) return NTM_OK;
    if (!I || !H) return fail(NTM_EINVAL, "ntm_tape_record_field: null pointer");
    hipError_t e = ntm::launch_tape_record_field(I, bias, H, B, N, gain, gap, (hipStream_t)stream);
    return e == hipSuccess ? NTM_OK : hip_fail(e, "ntm_tape_record_field");
}

int ntm_tape_hmag(const double *H, double *M, int64_t B, int64_t N, double *state, double Ts, const double *params5,
                  void *stream)
{
    if (B < 0 || N < 0 || !(Ts > 0.0)) return fail(NTM_EINVAL, "ntm_tape_hmag: bad size or Ts");
    if (B == 0 || N == 0) return NTM_OK;
    if (!H || !M || !state || !params5) return fail(NTM_EINVAL, "ntm_tape_hmag: null pointer");
    hipError_t e = ntm::launch_tape_hmag(H, M, B, N, state, Ts, params5, (hipStream_t)stream);
    return e == hipSuccess ? NTM_OK : hip_fail(e, "ntm_tape_hmag");
}

int ntm_resample_fir(const double *x, double *y, int64_t B, int64_t N, int64_t M, int up, int down, int width,
                     const double *kernel, void *stream)
{
    if (B < 0 || N < 0 || M < 0 || up < 1 || down < 1 || width < 0) return fail(NTM_EINVAL, "ntm_resample_fir: bad size");
    if (B > 65535) return fail(NTM_EINVAL, "ntm_resample_fir: at most 65535 streams per call");
    if (B == 0 || M == 0) return NTM_OK;
    if (!x || !y || !kernel || x == y) return fail(NTM_EINVAL, "ntm_resample_fir: null or aliased pointer");
    hipError_t e = ntm::launch_resample_fir(x, y, B, N, M, up, down, width, kernel, (hipStream_t)stream);
    return e == hipSuccess ? NTM_OK : hip_fail(e, "ntm_resample_fir");
}

int ntm_fir_f64(const double *x, double *y, int64_t B, int64_t N, const double *h, int taps, int clamp, void *stream)
{
    if (B < 0 || N < 0 || taps < 1) return fail(NTM_EINVAL, "ntm_fir_f64: bad size");
    if (B > 65535) return fail(NTM_EINVAL, "ntm_fir_f64: at most 65535 streams per call");
    if (B == 0 || N == 0) return NTM_OK;
    if (!x || !y || !h || x == y) return fail(NTM_EINVAL, "ntm_fir_f64: null or aliased pointer");
    hipError_t e = ntm::launch_fir_f64(x, y, B, N, h, taps, clamp, (hipStream_t)stream);
    return e == hipSuccess ? NTM_OK : hip_fail(e, "ntm_fir_f64");
}

namespace {
constexpr int kMaxDevices = 64;
struct TcnLanes {
    std::mutex mu;
    bool ready = false;
    hipStream_t lane[2] = {nullptr, nullptr};
    hipEvent_t fork = nullptr, done[2] = {nullptr, nullptr};
};
TcnLanes g_tcn_lanes[kMaxDevices];      // never destroyed: the runtime may already be gone when statics are torn down
}  // namespace

// The batch is processed in chunks of streams (streams are independent): the scratch is bounded by TCN_SCRATCH_BUDGET floats
// whatever B is -- 4096 x 65 536 x 32 ch took two 34.4 GB buffers as one launch set, and the per-GPU shapes of BASELINE
// configs[4] (B >= 8192) did not fit at all.  A batch that needs more than one chunk runs on TWO lanes: chunk k goes to lane
// k & 1, each lane has its own pair of activation buffers and its own HIP stream (forked from / joined to the caller's
// stream by events), so the drain of one chunk's launch is filled by the other
// lane's launch and the HBM-bound first block of one chunk runs under the matrix-pipe blocks of the other.  Chunks are
// equal-sized (the last may be smaller).  (The lane streams / events live in a per-device pool: this is the library's only
// state besides the thread-local error string, and it holds no data.)
static const int64_t TCN_SCRATCH_BUDGET = 2000000000;      // floats in total (8 GB): 2 lanes x 2 activation buffers
static int64_t tcn_chunk_streams(int64_t B, int64_t T, int C)
{
    const int64_t per = T * (int64_t)C;
    if (B * per <= TCN_SCRATCH_BUDGET / 2) return B;       // one chunk, two buffers, the caller's stream
    int64_t most = TCN_SCRATCH_BUDGET / 4 / per;
    if (most < 1) most = 1;                                // one stream longer than the budget: that stream alone
    const int64_t chunks = (B + most - 1) / most;
    return (B + chunks - 1) / chunks;
}

int64_t ntm_tcn_chunk_streams(int64_t B, int64_t T, int C)
{
    if (B <= 0 || T <= 0 || C <= 0) return 0;
    return tcn_chunk_streams(B, T, C);
}

int64_t ntm_tcn_scratch_floats(int64_t B, int64_t T, int C)
{
    if (B <= 0 || T <= 0 || C <= 0) return 0;
    const int64_t bc = tcn_chunk_streams(B, T, C);
    // per lane: two activation buffers, each padded by one 16-row block; two lanes when the batch is chunked
    return (bc < B ? 2 : 1) * 2 * (bc * T * (int64_t)C + 16 * (int64_t)C);
}

int ntm_tcn_forward(const float *params, int L, int C, int K, const int *dil, const float *x, float *y, int64_t B,
                    int64_t T, float *scratch, void *stream)
{
    if (L <= 0 || K <= 0 || B < 0 || T < 0) return fail(NTM_EINVAL, "ntm_tcn_forward: bad size");
    if (C != 32 || K != 13) return fail(NTM_EINVAL, "ntm_tcn_forward: only C = 32 channels, K = 13 taps is compiled");
    if (B == 0 || T == 0) return NTM_OK;
    if (!params || !dil || !x || !y || !scratch) return fail(NTM_EINVAL, "ntm_tcn_forward: null pointer");
    if ((reinterpret_cast<uintptr_t>(params) & 15) || (reinterpret_cast<uintptr_t>(scratch) & 15))
        return fail(NTM_EINVAL, "ntm_tcn_forward: params and scratch must be 16-byte aligned");
    if (T >= ((int64_t)1 << 31) - (1 << 25)) return fail(NTM_EINVAL, "ntm_tcn_forward: T must be below 2^31 - 2^25 samples");
    for (int l = 0; l < L; ++l)
        if (dil[l] <= 0 || dil[l] > (1 << 20)) return fail(NTM_EINVAL, "ntm_tcn_forward: dilations must lie in [1, 2^20]");
    const int64_t bc = tcn_chunk_streams(B, T, C);
    hipStream_t user = (hipStream_t)stream;
    if (bc >= B) {
        hipError_t e = ntm::launch_tcn(params, L, C, K, dil, x, y, B, T, scratch, user);
        return e == hipSuccess ? NTM_OK : hip_fail(e, "ntm_tcn_forward");
    }
    // two lanes: the side streams and the three events are created once per device and kept for the life of the process
    // (round 4 created and destroyed them per call -- hipStreamDestroy may wait for the queue to drain, which would have
    // turned "asynchronous on `stream`" into a host-blocking call for every chunked batch, and stream creation cannot be
    // captured into a graph).  The pool's mutex is held while the call ENQUEUES (microseconds): calls on one device share
    // the lanes, which only serialises work that would contend for the same CUs anyway.
    const int64_t lane_floats = 2 * (bc * T * (int64_t)C + 16 * (int64_t)C);
    // A call under stream capture keeps to the caller's stream: the pooled lanes are shared by every caller on the device, and
    // a lane forked into one caller's capture would drag a concurrent, un-captured call of another thread into that capture
    // (or fail it with a capture-isolation error); stream and event creation cannot be captured either.  The chunks then run
    // one after the other on `stream`, still alternating the two scratch halves -- same results, no lane overlap in the graph.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(user, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    if (cap != hipStreamCaptureStatusNone) {
        hipError_t ec = hipSuccess;
        int kc = 0;
        for (int64_t b0 = 0; b0 < B && ec == hipSuccess; b0 += bc, ++kc) {
            const int64_t n = B - b0 < bc ? B - b0 : bc;
            ec = ntm::launch_tcn(params, L, C, K, dil, x + b0 * T, y + b0 * T, n, T, scratch + (kc & 1) * lane_floats, user);
        }
        return ec == hipSuccess ? NTM_OK : hip_fail(ec, "ntm_tcn_forward");
    }
    int devi = 0;
    hipError_t e = hipGetDevice(&devi);
    if (e != hipSuccess) return hip_fail(e, "ntm_tcn_forward");
    if (devi < 0 || devi >= kMaxDevices) return fail(NTM_EINVAL, "ntm_tcn_forward: device ordinal beyond the lane pool");
    TcnLanes &P = g_tcn_lanes[devi];
    std::lock_guard<std::mutex> hold(P.mu);
    if (!P.ready) {
        // built into locals and committed only when all five handles exist: a failure half way destroys what it made
        // (a retry used to overwrite, i.e. leak, the handles of the first attempt)
        hipEvent_t fork = nullptr, done[2] = {nullptr, nullptr};
        hipStream_t lane[2] = {nullptr, nullptr};
        e = hipEventCreateWithFlags(&fork, hipEventDisableTiming);
        for (int i = 0; i < 2 && e == hipSuccess; ++i) {
            e = hipStreamCreateWithFlags(&lane[i], hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
        }
        if (e != hipSuccess) {
            if (fork) (void)hipEventDestroy(fork);
            for (int i = 0; i < 2; ++i) {
                if (done[i]) (void)hipEventDestroy(done[i]);
                if (lane[i]) (void)hipStreamDestroy(lane[i]);
            }
            return hip_fail(e, "ntm_tcn_forward");
        }
        P.fork = fork;
        for (int i = 0; i < 2; ++i) { P.lane[i] = lane[i]; P.done[i] = done[i]; }
        P.ready = true;
    }
    e = hipEventRecord(P.fork, user);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipStreamWaitEvent(P.lane[i], P.fork, 0);
    int k = 0;
    for (int64_t b0 = 0; b0 < B && e == hipSuccess; b0 += bc, ++k) {
        const int64_t n = B - b0 < bc ? B - b0 : bc;
        e = ntm::launch_tcn(params, L, C, K, dil, x + b0 * T, y + b0 * T, n, T, scratch + (k & 1) * lane_floats, P.lane[k & 1]);
    }
    for (int i = 0; i < 2; ++i) {              // join -- also after an error, so that nothing outlives the caller's ordering
        if (hipEventRecord(P.done[i], P.lane[i]) == hipSuccess) (void)hipStreamWaitEvent(user, P.done[i], 0);
    }
    return e == hipSuccess ? NTM_OK : hip_fail(e, "ntm_tcn_forward");
}

}  // extern "C"
