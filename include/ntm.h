/*
 * ntm.h -- C ABI of libntm.so: the MI355X (gfx950) kernels for the neural tape-nonlinearity
 * forward path of 01tot10/neural-tape-modeling.
 *
 * The reference has no FFI for this path: it is a Python torch.nn.Module protocol
 * (code/model.py).  Each entry point below names the reference call it replaces; the Python
 * host layer (neural-tape-modeling_amd/model.py) rebuilds the reference's object protocol on top
 * of these and INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer owned by the caller (fp32) unless an entry point says otherwise (small
 *     host-side parameter arrays, the host side of ntm_copy2d_async); nothing is
 *     allocated or freed inside; no global state (carried state lives in the caller's h_state /
 *     dl_state buffers), so calls are re-entrant;
 *   - work is enqueued asynchronously on `stream` (a hipStream_t passed as void*; NULL = default);
 *   - audio tensors are [B, T] row-major views of the reference's (B,1,T) layout with explicit
 *     per-stream strides in ELEMENTS (>= T);
 *   - return value: 0 on success, negative NTM_E* on failure; ntm_last_error() gives the
 *     thread-local message.  There is NO CPU fallback: without a HIP device every call fails.
 */
#ifndef NTM_H
#define NTM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NTM_OK 0
#define NTM_EINVAL (-1)  /* bad argument (null pointer, negative size, unsupported H)  */
#define NTM_EHIP (-2)    /* HIP runtime error (launch failed, no device)               */
#define NTM_EDELAY (-3)  /* reserved for host-side delay-range checks                  */

#define NTM_ABI_VERSION 9 /* 2: hidden sizes 8/16/32/64; the delay line is one pass, no scratch, sticky error flag.  3: the TCN scratch is padded (ntm_tcn_scratch_floats grew), dilation / length limits.  4: ntm_diffdel_gru_forward is ONE fused launch where the matrix-pipe kernel runs (+ ntm_diffdel_gru_forward_ex).  5: ntm_gru_forward_esr, ntm_diffdel_gru_forward_esr.  6: ntm_tcn_forward works through the batch in stream chunks, ntm_tcn_scratch_floats is bounded (<= 2.0e9 floats + padding for any B), ntm_tcn_chunk_streams; DiffDelGRU warm-up calls always take the two-pass form.  7: ntm_loss_scalars (+ include/ntm_rccl.h, libntm_rccl.so).  8: any hidden size in [1, NTM_MAX_HIDDEN]; ntm_gru_forward_losses / ntm_diffdel_gru_forward_losses (ESR + DCPreESR sums in the recurrent launch).  9: NTM_GRU_BF16X3, ntm_gru_forward_io (any input_size / output_size) */

#define NTM_HIDDEN 64 /* hidden size of every shipped checkpoint (HS[64]): matrix-pipe and low-latency kernels.
                         Every other H in [1, NTM_MAX_HIDDEN] (the reference's `--HIDDEN_SIZE` is a free integer,
                         code/train.py:50; its constructor default is 8, code/model.py:22, its training default 16)
                         runs on gru_small.hip, any variant among AUTO / LAT / VALU: H = 8, 16, 32 one wavefront per
                         64/H streams; other H < 64 the same kernel zero-padded to the next power of two; 64 < H <=
                         128 a workgroup per stream with the weights in registers; above that a plain kernel that
                         streams the weights from L2 (correct, not fast).                                          */
#define NTM_MAX_HIDDEN 1024

/* GRU kernel variants (the laboratory ones: docs/DESIGN_measurement_log_r1_r5.md).  ntm_gru_forward_ex of libntm.so (the product) accepts NTM_GRU_AUTO, _MFMA2,
 * _LAT and the opt-in _F16X3 / _BF16X3; the others are LABORATORY kernels -- older or experimental exact-fp32 implementations
 * kept as independent checks and as measured dead ends -- compiled into libntm_lab.so only (include/ntm_lab.h).  */
#define NTM_GRU_AUTO 0  /* NTM_GRU_MFMA2, or NTM_GRU_LAT when B <= NTM_GRU_LAT_MAX_B         */
#define NTM_GRU_MFMA 1  /* 16 streams / workgroup, 4 waves, v_mfma_f32_16x16x4_f32, h in LDS */
#define NTM_GRU_VALU 2  /* 2 streams / wavefront, W_hh in VGPRs, h broadcast through LDS     */
#define NTM_GRU_MFMA2 3 /* as MFMA, own-quarter-first step order: LDS exchange hidden by MFMAs */
#define NTM_GRU_MFMA3 5 /* retired in round 6 (a measured negative result; the number stays reserved)     */
#define NTM_GRU_LAT 6   /* exact fp32, ONE stream per workgroup (K split over 4 waves): low latency, small B */
#define NTM_GRU_MFMA4 7 /* retired in round 6 (a measured negative result; the number stays reserved)     */
#define NTM_GRU_LAT_MAX_B 1024
#define NTM_GRU_F16X3 4 /* OPT-IN: MFMA2 with W.h as three fp16 hi/lo products, fp32 accumulate  */
#define NTM_GRU_BF16X3 8 /* OPT-IN: MFMA2 with W and h each split into THREE bf16 pieces (24 bits: the fp32 operands exactly) and
                            W.h as their eight partial products W_p.h_q, p + q <= 3 (W_3.h_3 <= 2^-32 |W||h| is dropped), each exact
                            in fp32, on v_mfma_f32_16x16x32_bf16 with fp32 accumulation; state, gates and head in fp32 as MFMA2 */

/* ABI version of this header; bumped on any signature change. */
int ntm_abi_version(void);

/* Thread-local message for the last failing call on this thread ("" if none). */
const char *ntm_last_error(void);

/*
 * Replaces  x, self.hidden = self.GRU(x, self.hidden); y = self.output(x)
 * at code/model.py:81-82 (RNN.forward) and :412-413 (DiffDelRNN.forward, where b_o == NULL
 * because the head is Linear(..., bias=False), code/model.py:365).
 *
 * Weights in the reference's state_dict layout, gate row order r,z,n:
 *   w_ih [3H]      GRU.weight_ih_l0 (3H,1)      w_hh [3H,H]   GRU.weight_hh_l0
 *   b_ih [3H]      GRU.bias_ih_l0               b_hh [3H]     GRU.bias_hh_l0
 *   w_o  [H]       output.weight (1,H)          b_o  [1] or NULL   output.bias
 * x [B,T] (stride x_stride_b) -> y [B,T] (stride y_stride_b).
 * 1 <= H <= NTM_MAX_HIDDEN.  h_state [B,H] is read as h_0 and overwritten with h_T (the reference's self.hidden);
 * NULL means h_0 = 0 and h_T is discarded.  B == 0 or T == 0 is a successful no-op.
 */
int ntm_gru_forward(const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                    const float *w_o, const float *b_o, int H, const float *x, float *y,
                    int64_t B, int64_t T, int64_t x_stride_b, int64_t y_stride_b, float *h_state,
                    void *stream);

/*
 * The same reference call for ANY input_size / output_size: self.GRU = nn.GRU(input_size, hidden_size, batch_first=True),
 * self.output = nn.Linear(hidden_size, output_size) (code/model.py:22,44-45), forward code/model.py:67-88.  The reference moves
 * between (B, C, T) and (B, T, C) with `reshape` (:77, :87) -- a reinterpretation of the row, not a transpose -- so per stream the
 * input row of I T floats IS the [T][I] matrix the GRU reads and the output row of O T floats IS the [T][O] matrix the head
 * writes: x [B] rows of T*I floats (x_stride_b >= T*I), y [B] rows of T*O floats.  w_ih [3H, I], w_o [O, H], b_o [O] or NULL;
 * h_state [B, H] in/out or NULL (zeros).  H, I, O in [1, 1024].  No caller of the reference uses sizes other than 1
 * (code/test-model.py:124-125): a plain, correct kernel (a workgroup per stream, weights from L2), not a fast one; pinned by
 * golden g22 from the reference's own forward().  The skip connection (y += x, output_size == input_size) is the caller's.
 */
int ntm_gru_forward_io(const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                       const float *w_o, const float *b_o, int H, int I, int O, const float *x, float *y,
                       int64_t B, int64_t T, int64_t x_stride_b, int64_t y_stride_b, float *h_state, void *stream);

/* Same, with an explicit kernel variant (NTM_GRU_*); used by bench.py / tests to A/B kernels. */
int ntm_gru_forward_ex(const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                       const float *w_o, const float *b_o, int H, const float *x, float *y,
                       int64_t B, int64_t T, int64_t x_stride_b, int64_t y_stride_b,
                       float *h_state, int variant, void *stream);

/*
 * RNN.forward + the ESR sums of the loss loop in ONE call: replaces
 *     output = model(input)  ...  loss_fcn(output[..., INIT_LEN:], target[..., INIT_LEN:])      (code/test-model.py:346,386-388)
 * for the ESR entry of the loss dict.  Arguments as ntm_gru_forward (kernel choice = NTM_GRU_AUTO), then
 * target [B,T] contiguous, skip (= INIT_LEN), and esr_out [B,2] fp64 (device):
 *     esr_out[2b] = sum_{n >= skip} (target - y)^2,   esr_out[2b+1] = sum_{n >= skip} target^2      of stream b
 * -- the per-sample terms of ntm_esr_sums (fp32 difference, fp64 products and sums), added in a fixed order.  Where the
 * matrix-pipe kernel runs (H = 64, B > NTM_GRU_LAT_MAX_B, skip a multiple of 4) the sums are accumulated inside that launch,
 * in the y-tile flush where the outputs sit in registers (the separate 2 GB pass, overlapped with the next launch, cost
 * that launch 0.37 ms at 4096 x 65 536); elsewhere it is the forward launch followed by the streaming pass (then y must
 * be contiguous).  target must not alias y.
 */
int ntm_gru_forward_esr(const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                        const float *w_o, const float *b_o, int H, const float *x, float *y,
                        int64_t B, int64_t T, int64_t x_stride_b, int64_t y_stride_b, float *h_state,
                        const float *target, int64_t skip, double *esr_out, void *stream);

/*
 * RNN.forward + BOTH time-domain entries of the loss dict in ONE call: replaces
 *     output = model(input)  ...  for key in loss_fcns: loss_fcns[key](output[..., INIT_LEN:], target[..., INIT_LEN:])
 * (code/test-model.py:250-252,346,386-388) for the ESR and the DCPreESR entries.  Arguments as ntm_gru_forward_esr, then the
 * pole dcpre_R of the DC blocker (0.995 upstream) and dcpre_out [B,2] fp64 (device), the sums of ntm_esr_dcpre_sums:
 *     dcpre_out[2b] = sum f(target - y)^2,   dcpre_out[2b+1] = sum f(target)^2,   f = (1 - z^-1)/(1 - R z^-1) from zero
 * state at sample `skip`.  Where the matrix-pipe kernel runs (H = 64, B > NTM_GRU_LAT_MAX_B, skip a multiple of 4) both pairs
 * of sums come out of that launch: the one-pole filter is a 16-lane scan inside the y-tile flush (the streaming pass reads
 * y and target again, 2 GB at 4096 x 65 536); elsewhere the forward launch is followed by the two streaming passes.  The fp32
 * filter is evaluated in scan order there and in the streaming kernel's order elsewhere: the two agree with each other and
 * with the sequential recursion to ~1e-6 relative in the sums (not bit for bit).
 */
int ntm_gru_forward_losses(const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                           const float *w_o, const float *b_o, int H, const float *x, float *y,
                           int64_t B, int64_t T, int64_t x_stride_b, int64_t y_stride_b, float *h_state,
                           const float *target, int64_t skip, double *esr_out, float dcpre_R,
                           double *dcpre_out, void *stream);

/*
 * Replaces TimeVaryingDelayLine.forward(x, dt, warmup), code/model.py:269-320.
 * x, d, y: [B,T] contiguous (stride T); d in SAMPLES.  dl_state [B,D] is the reference's
 * `self.buffer` (oldest sample first) and is updated in place to cat(buffer[T:], x[-D:]).
 * warmup != 0: y = x, only the buffer is updated (code/model.py:288-292).
 * y must not alias x.  One pass over the audio: d is read once, there is no separate range check.
 * err_flag (device int32, caller-owned, may be NULL): raised to 1 if any d > D (or NaN) -- the reference raises
 * AssertionError there (code/model.py:284) before touching its state; here y of the violating call is
 * unspecified and dl_state is left untouched.  The flag is STICKY: while it is non-zero every later call that is
 * given the same flag is a no-op (the state stays frozen at the last good call), so a caller that streams many
 * chunks may look at it once at the end instead of synchronising per chunk, and clears it (hipMemsetAsync)
 * when it re-initialises the state.
 */
int ntm_delay_forward(const float *x, const float *d, float *y, int64_t B, int64_t T,
                      float *dl_state, int D, int warmup, int32_t *err_flag, void *stream);

/*
 * Replaces DiffDelRNN.forward(x, del_traj, warmup), code/model.py:393-424:
 * the GRU + bias-free head writes pre_d, the delay line writes y.  Returns (y, pre_d) through the two
 * output pointers (distinct buffers, neither aliasing x); x, d, y, pre_d [B,T] contiguous; h_state, dl_state,
 * err_flag as above (a violation leaves dl_state untouched; h_state is h_T either way -- the reference assigns
 * self.hidden at :412 before its delay line asserts).
 *
 * Where the matrix-pipe kernel runs (H = 64, B > NTM_GRU_LAT_MAX_B) the step is ONE fused launch: the delay line
 * interpolates the kernel's own pre_d output inside the y-tile housekeeping of the recurrence, one 64-sample tile
 * behind it (taps read back through L2; d is read once, pre_d is never re-read from HBM), bit-identical to the separate
 * pass on the same pre_d; a small launch moves the carried buffer on afterwards.  Small batches and H < 64 run the GRU
 * launch followed by the streaming delay pass (NTM_DIFFDEL_TWO_PASS).  ntm_diffdel_gru_forward = mode NTM_DIFFDEL_AUTO.
 */
#define NTM_DIFFDEL_AUTO 0     /* fused where the matrix-pipe kernel runs, two passes elsewhere          */
#define NTM_DIFFDEL_TWO_PASS 1 /* ntm_gru_forward then ntm_delay_forward                                 */
#define NTM_DIFFDEL_FUSED 2    /* the fused kernel for every stream (H = 64), whatever B is (tests, A/B) */
int ntm_diffdel_gru_forward(const float *w_ih, const float *w_hh, const float *b_ih,
                            const float *b_hh, const float *w_o, int H, const float *x,
                            const float *d, float *y, float *pre_d, int64_t B, int64_t T,
                            float *h_state, float *dl_state, int D, int warmup, int32_t *err_flag,
                            void *stream);
int ntm_diffdel_gru_forward_ex(const float *w_ih, const float *w_hh, const float *b_ih,
                               const float *b_hh, const float *w_o, int H, const float *x,
                               const float *d, float *y, float *pre_d, int64_t B, int64_t T,
                               float *h_state, float *dl_state, int D, int warmup, int32_t *err_flag,
                               int mode, void *stream);
/*
 * DiffDelRNN.forward + the ESR sums of the loss loop (code/test-model.py:353,386-388) in one call: as
 * ntm_diffdel_gru_forward (warmup = 0, mode NTM_DIFFDEL_AUTO), then target [B,T] contiguous, skip (= INIT_LEN) and
 * esr_out [B,2] fp64 = sum (target - y)^2 | sum target^2 over samples [skip, T) of each stream, y being the DELAYED output.
 * Inside the fused launch (in its delay stage, where a thread's 4 outputs are in registers) when that launch runs and skip is a
 * multiple of 4; by the streaming ESR pass otherwise.  After a delay-range violation (err_flag raised) the sums are unspecified,
 * like y.
 */
int ntm_diffdel_gru_forward_esr(const float *w_ih, const float *w_hh, const float *b_ih,
                                const float *b_hh, const float *w_o, int H, const float *x,
                                const float *d, float *y, float *pre_d, int64_t B, int64_t T,
                                float *h_state, float *dl_state, int D, int32_t *err_flag,
                                const float *target, int64_t skip, double *esr_out, void *stream);

/* DiffDelRNN.forward + BOTH time-domain entries of the loss dict (ESR and DCPreESR of the DELAYED output, code/test-model.py:250-252,
 * 353,386-388) in one call: as ntm_diffdel_gru_forward_esr, plus dcpre_R / dcpre_out [B,2] fp64 as in ntm_gru_forward_losses.  Where the
 * fused step runs both pairs of sums come out of that ONE launch (accumulated in its delay stage); elsewhere the streaming passes follow. */
int ntm_diffdel_gru_forward_losses(const float *w_ih, const float *w_hh, const float *b_ih,
                                   const float *b_hh, const float *w_o, int H, const float *x,
                                   const float *d, float *y, float *pre_d, int64_t B, int64_t T,
                                   float *h_state, float *dl_state, int D, int32_t *err_flag,
                                   const float *target, int64_t skip, double *esr_out, float dcpre_R,
                                   double *dcpre_out, void *stream);

/*
 * Per-stream sums for the ESR loss that follows the path in code/test-model.py:250-254,386-388
 * (CoreAudioML ESRLoss, un-vendored): over samples [skip, T) of stream b, split over `splits` workgroups p:
 *   out[(b*splits + p)*2 + 0] = partial sum (t - y)^2      out[(b*splits + p)*2 + 1] = partial sum t^2   (fp64, device)
 * The caller adds the `splits` rows of a stream in index order: no atomics anywhere, the sums are bit-reproducible
 * from run to run for every B.  ntm_esr_splits() gives the split count that fills the device (1 for B >= 2048).
 * y, t: [B,T] contiguous.
 */
int ntm_esr_sums(const float *y, const float *t, int64_t B, int64_t T, int64_t skip, int splits, double *out,
                 void *stream);
int ntm_esr_splits(int64_t B, int64_t T, int64_t skip);

/*
 * This rank's four loss scalars from per-stream ESR rows (esr_out of ntm_gru_forward_esr / ntm_diffdel_gru_forward_esr, or the
 * summed partial rows of ntm_esr_sums): what the loss loop of code/test-model.py:386-398 aggregates over the segments --
 *     out4 = [ sum_b ESR_b,  B,  sum_b err2_b,  sum_b tgt2_b ],     ESR_b = (err2_b / n_samples) / (tgt2_b / n_samples + eps)
 * (eps = 1e-5: CoreAudioML's ESRLoss; n_samples = T - skip) -- fp64, device in, device out, one small launch, bit-reproducible.
 * A job sharded over ranks adds these four numbers over the ranks (ntm_rccl_allreduce_f64 in include/ntm_rccl.h, MPI, or
 * torch.distributed) and divides out4[0] by out4[1]: the mean over segments of the per-segment loss.  B == 0 gives zeros.
 */
int ntm_loss_scalars(const double *esr_rows, int64_t B, int64_t n_samples, double eps, double *out4, void *stream);

/*
 * As ntm_esr_sums, on the DC-blocked signals: both y and t pass H(z) = (1 - z^-1)/(1 - R z^-1) (zero state
 * at sample `skip`) before the sums are taken -- the `DCPreESR(dc_pre=True)` loss of code/test-model.py:252
 * and code/train.py:173-174 (GreyBoxDRC, un-vendored: parity unpinned; upstream truncates the impulse
 * response to 2000 taps, this is the untruncated recursion).  R = 0.995 upstream.
 */
int ntm_esr_dcpre_sums(const float *y, const float *t, int64_t B, int64_t T, int64_t skip, float R,
                       double *out, void *stream);

/*
 * "Next" row N1: per-stream sums of the multi-resolution STFT loss of code/test-model.py:25,253
 * (`MultiResolutionSTFTLoss()` of the un-vendored auraloss submodule: parity unpinned by the reference; the
 * arithmetic is pinned to torch.stft by tests/golden/g10).  For ONE resolution, over samples [skip, T) of each
 * stream:  X = stft(x, n_fft, hop, win_length, periodic Hann, centred, reflect padding),
 * mag = sqrt(max(re^2 + im^2, power_eps)), and with P = 4 * chunks partial rows per stream
 *   out[(b*P + p)*4 + 0..3] = sum (mag_t - mag_y)^2 | sum mag_t^2 | sum |ln mag_y - ln mag_t| | sum |mag_y - mag_t|
 * (fp64, device; the caller adds the P rows of a stream).  y = prediction, t = target, [B,T] contiguous.
 * n_fft in {64, 128, 256, 512, 1024, 2048}; 0 < win_length <= n_fft; T - skip > n_fft/2; power_eps > 0 (auraloss: 1e-8);
 * chunks >= 1 splits the frames of a stream over that many workgroups.  There are 1 + (T-skip)/hop frames
 * of n_fft/2 + 1 bins.
 */
int ntm_stft_sums(const float *y, const float *t, int64_t B, int64_t T, int64_t skip, int n_fft, int hop,
                  int win_length, float power_eps, int chunks, double *out, void *stream);

/*
 * Same transform and layout as ntm_stft_sums, with the POWER-spectrogram terms of the reference's validation
 * metric bundle (code/evaluation.py:75-84: `TimeFreqConverter` = torchaudio Spectrogram(n_fft, hop = n_fft/4,
 * power = 2), un-vendored torchaudio semantics = torch.stft, parity pinned to torch.stft by golden g13):
 *   out[..0..3] = sum |P_y - P_t| | sum |log10 max(P_y, log_floor) - log10 max(P_t, log_floor)| | sum P_t | sum P_y
 * (log_floor = 1e-5 in the reference).
 */
int ntm_spec_sums(const float *y, const float *t, int64_t B, int64_t T, int64_t skip, int n_fft, int hop,
                  int win_length, float log_floor, int chunks, double *out, void *stream);

/*
 * The two MEL entries of the same bundle (code/evaluation.py:86-92): the power spectrogram of
 * `TimeFreqConverter(n_fft = 2048, hop 512)` projected on the mel basis `librosa.filters.mel(sr, n_fft, 160)`
 * (code/utilities/utilities.py:639-646, :666) inside the transform kernel, then
 *   out[..0..3] = sum |mel_y - mel_t| | sum |log10 max(mel_y, log_floor) - log10 max(mel_t, log_floor)| | sum mel_t | sum mel_y
 * in the partial-row layout of ntm_stft_sums.  The filter bank comes in row-compressed form (device arrays): filter m
 * has weights mel_w[mel_start[m] .. mel_start[m+1]) on the bins mel_first[m], mel_first[m] + 1, ...  (host helper:
 * ntm_amd.utilities.mel_filterbank_sparse; librosa is un-vendored and absent: published algorithm, parity unpinned).
 * n_fft in {1024, 2048}; there are (1 + (T-skip)/hop) * n_mels cells per stream.
 */
int ntm_mel_sums(const float *y, const float *t, int64_t B, int64_t T, int64_t skip, int n_fft, int hop, int win_length,
                 float log_floor, int chunks, int n_mels, const int32_t *mel_first, const int32_t *mel_start,
                 const float *mel_w, double *out, void *stream);

/*
 * "Next" row N2 plumbing: pitched asynchronous copy between (pinned) host memory and the device, rows x
 * width_bytes with independent pitches -- what the segment feeder uses to send a TIME CHUNK of many segments
 * ([B, c0:c1] of a [B,T] batch) so that the copy of chunk c+1 overlaps the GRU launch on chunk c.
 * kind 0: host -> device, 1: device -> host.  Thin wrapper of hipMemcpy2DAsync on `stream`.
 */
int ntm_copy2d_async(void *dst, int64_t dst_pitch_bytes, const void *src, int64_t src_pitch_bytes,
                     int64_t width_bytes, int64_t rows, int kind, void *stream);

/*
 * "Next" row N3: DelayAnalyzer.demodulate, code/utilities/utilities.py:408-465 -- removes the time-varying
 * delay of a recording using the pulse indices of its pilot channel.  x, out: [C,N] fp32 device (out must not
 * alias x); y_idx [P] int64 device = output pulse indices (strictly increasing, P >= 2);
 * period = int(mean(diff(input pulse indices))) and shift = y_idx[0] - x_idx[0] are computed by the caller from
 * the INPUT pulse indices (:436, :458); scratch holds N doubles.  Interpolation runs in fp64 with scipy's
 * formulas, results are rounded to fp32 on store.
 */
int ntm_demodulate(const float *x, float *out, int C, int64_t N, const int64_t *y_idx, int P, int64_t period,
                   int64_t shift, double *scratch, void *stream);

/*
 * Record-head field of the reference's tape simulator, the stage in front of H_mag: I_rec = I + bias
 * (code/tape.py:476-510; bias [N] fp64 is shared by all streams, NULL = bias disabled) and
 * H = (gain * I_rec) / gap with gain = REC_N * REC_E, gap = REC_G (:512-514).  I, H: [B,N] fp64 device.
 */
int ntm_tape_record_field(const double *I, const double *bias, double *H, int64_t B, int64_t N, double gain,
                          double gap, void *stream);

/*
 * "Next" row N4: replaces Tape.H_mag, code/tape.py:516-551 (Jiles-Atherton hysteresis, RK4, fp64) of the
 * reference's white-box tape simulator.  H, M: [B,N] fp64 device, contiguous (oversampled rate);
 * state [B,3] fp64 device = (M_prev, H_prev, Hprime_prev), read and updated (zeros initially, :303-309);
 * Ts = 1/(fs*oversampling); params5 is a HOST array {Ms, A, alpha, K, c} (code/tape.py:251-256).
 */
int ntm_tape_hmag(const double *H, double *M, int64_t B, int64_t N, double *state, double Ts,
                  const double *params5, void *stream);

/*
 * The resamplers of Tape.__call__ (code/tape.py:330-332,471-474,553-558: torchaudio.transforms.Resample -- sinc
 * interpolation, Hann window, lowpass_filter_width 6, rolloff 0.99; torchaudio is un-vendored and absent here: the
 * published algorithm, parity unpinned).  Polyphase FIR in fp64 with a caller-supplied kernel table (device,
 * [up][2*width + down], built by ntm_amd.tape.sinc_resample_kernel):
 *   y[b][i*up + p] = sum_k kernel[p][k] * xpad[b][i*down + k],  xpad[j] = x[j - width] (zero outside [0, N)),  i*up + p < M.
 * x [B,N], y [B,M] fp64 device, contiguous.
 */
int ntm_resample_fir(const double *x, double *y, int64_t B, int64_t N, int64_t M, int up, int down, int width,
                     const double *kernel, void *stream);

/*
 * The playback-loss filter of Tape.H_play (code/tape.py:565-574: torchaudio.functional.lfilter with a = [1, 0, ...],
 * i.e. a FIR; clamp != 0 limits the output to [-1, 1] as lfilter's default does; zero initial state every call, as in the
 * reference):  y[b][n] = sum_{k < taps, k <= n} h[k] x[b][n-k].   x, y [B,N] fp64 device; h [taps] fp64 device.
 */
int ntm_fir_f64(const double *x, double *y, int64_t B, int64_t N, const double *h, int taps, int clamp, void *stream);

/*
 * Builder-defined causal dilated-Conv1d TCN (BASELINE.json config 4; the reference has no TCN:
 * code/micro_tcn is an empty submodule).  L causal blocks  out = PReLU(conv_dilated(in)) + conv1x1(in),
 * then a 1x1 conv to one channel.  params (device), block after block:
 *   W[C_in][K][C], b[C], alpha[C], R[C_in][C]   (C_in = 1 for block 0, C afterwards), then out_w[C], out_b[1].  x,y [B,T] contiguous; dil[L] host array; `scratch`
 * holds ntm_tcn_scratch_floats(B,T,C) floats (two activation buffers, each padded by one 16-row block: always ask this
 * function).  A batch whose activations exceed 8 GB is worked through in chunks of ntm_tcn_chunk_streams(B,T,C) streams
 * (streams are independent; same results bit for bit): the scratch never exceeds 2e9 floats (+ padding) whatever B is --
 * 8 GB for 32 768 x 65 536 instead of 2 x 275 GB.  Chunks alternate between two lanes, each with its own pair of
 * activation buffers inside `scratch` and its own HIP stream, forked from and joined to `stream` by events (the two side
 * streams and three events are created on the first chunked call on a device and kept -- the library's only state, it
 * holds no data): the call returns as soon as the work is enqueued, is ordered on `stream` like any other, and one
 * chunk's drain and HBM-bound first block run under the other chunk's matrix-pipe blocks.  A call made while `stream` is
 * being CAPTURED into a graph does not touch the shared lanes: its chunks are enqueued one after the other on `stream`
 * itself (same results; the lanes belong to every caller on the device, and a lane forked into one capture would pull
 * another thread's concurrent call into it).  Limits: T < 2^31 - 2^25,
 * 1 <= dil[l] <= 2^20 (NTM_EINVAL otherwise).
 */
int ntm_tcn_forward(const float *params, int L, int C, int K, const int *dil, const float *x,
                    float *y, int64_t B, int64_t T, float *scratch, void *stream);
int64_t ntm_tcn_scratch_floats(int64_t B, int64_t T, int C);
int64_t ntm_tcn_chunk_streams(int64_t B, int64_t T, int C);

#ifdef __cplusplus
}
#endif
#endif /* NTM_H */
