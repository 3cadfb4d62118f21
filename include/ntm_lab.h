/*
 * ntm_lab.h -- C ABI of libntm_lab.so: the LABORATORY beside the product library libntm.so (include/ntm.h).
 *
 * Older and experimental exact-fp32 GRU-HS[64] kernels (kept as independent implementations for the parity tests and
 * as documented, measured dead ends: docs/DESIGN_measurement_log_r1_r5.md) and the diagnostic builds of the product kernel.  Nothing on the
 * product path loads this library; same pointer / stream / error conventions as ntm.h.
 */
#ifndef NTM_LAB_H
#define NTM_LAB_H

#include "ntm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Thread-local message for the last failing laboratory call on this thread. */
const char *ntm_lab_last_error(void);

/*
 * Same contract as ntm_gru_forward_ex (code/model.py:81-82), for variant in
 *   NTM_GRU_MFMA   first matrix-pipe kernel (natural K order, LDS exchange exposed)
 *   NTM_GRU_VALU   one wavefront per two streams on v_fma_f32 (north_star's first idea)
 * (NTM_GRU_MFMA3 -- MFMA waves + partner VALU waves on the same SIMDs -- and NTM_GRU_MFMA4 -- one wavefront per 4 streams on
 *  v_mfma_f32_4x4x1_16B_f32 -- were measured negative results of rounds 2-5 and are retired: docs/DESIGN_measurement_log_r1_r5.md
 *  4 K1d / K1f, sources in the history up to round 5.)
 * H must be 64.
 */
int ntm_lab_gru_forward(const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                        const float *w_o, const float *b_o, int H, const float *x, float *y,
                        int64_t B, int64_t T, int64_t x_stride_b, int64_t y_stride_b,
                        float *h_state, int variant, void *stream);

/*
 * DIAGNOSTIC ONLY (never timed): an MFMA kernel (variant NTM_GRU_MFMA or NTM_GRU_MFMA2) with
 * s_memtime stamps.  stamps[(B+15)/16][4][12] (device, uint64) receives per-wave cycle sums of six
 * step segments over the whole launch in slots 0-5 (segment names: tools/stamp_profile.py) and, for NTM_GRU_MFMA2, the same
 * six segments over the phase-2 (housekeeping) steps alone in slots 6-11.  Outputs are the same as ntm_gru_forward.
 * With NTM_LAB_STAMP_ESR set in the environment the stamped build of the forward + ESR-sums variant runs (x as target).
 */
int ntm_debug_gru_stamps(const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                         const float *w_o, const float *b_o, const float *x, float *y, int64_t B,
                         int64_t T, float *h_state, uint64_t *stamps, int variant, void *stream);

/* DIAGNOSTIC ONLY (wrong results on purpose, for timing ablations of the MFMA2 kernel): mask bits
 * 1 no gate math, 2 no LDS exchange of h, 4 no head partial, 8 own-quarter MFMAs only, 16 no barrier,
 * 32 no tile housekeeping; only the combinations compiled in gru_mfma2.hip are accepted. */
int ntm_debug_gru_ablate(const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                         const float *w_o, const float *b_o, const float *x, float *y, int64_t B,
                         int64_t T, float *h_state, int mask, void *stream);

/* DIAGNOSTIC ONLY: the in-register 4x4 lane-group transpose used by the MFMA2 kernel, applied to one
 * 256-thread block: in/out [256][4] floats (device). */
int ntm_debug_transpose4(const float *in, float *out, void *stream);

/* DIAGNOSTIC ONLY: ntm_tcn_forward's arguments through a build of the TCN kernels whose phase-group block kernel (inner
 * blocks with dilation >= 512) takes s_memtime at six points of every iteration; ntm_lab_tcn_stamps copies the sums one
 * wave accumulated to host7[0..5] (issue of the block loads, MFMA block, ring stores, epilogue, barrier, loop back-edge;
 * clock ticks) and its iteration count to host7[6]; host7 must hold 8 + 3 * 64 words: from [8] on, per iteration, the
 * start (cycles since kernel entry), the duration of the MFMA block and of the whole iteration.  With NTM_LAB_TCN_ONE_WG set in the environment the launch asks
 * for 90 KB of LDS, i.e. one workgroup per CU and one wave per SIMD.  Returns a hipError_t value (0 = success). */
int ntm_lab_tcn_forward(const float *params, int L, int C, int K, const int *dil, const float *x, float *y, int64_t B,
                        int64_t T, float *scratch, void *stream);
int ntm_lab_tcn_stamps(unsigned long long *host7);
/* device buffer of 4 words per workgroup of the stamped launch (s_memtime at start and end, XCC_ID << 32 | HW_ID, cycles
 * inside the iteration loop), or NULL to switch the trace off */
int ntm_lab_tcn_trace(unsigned long long *device_buf);

#ifdef __cplusplus
}
#endif
#endif /* NTM_LAB_H */
