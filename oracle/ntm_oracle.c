/*
 * ntm_oracle.c -- CPU restatement of the reference's tape-nonlinearity forward path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product path (libntm.so, HIP) never calls it.
 *
 * Parity status: PINNED by golden vectors generated in the build container by importing the
 * reference itself (tools/make_goldens.py -> tests/golden/g1..g8); the reference ships no tests
 * of its own.  tests/test_oracle.py checks every function below against them.
 *
 * What is restated (file:line relative to the reference checkout):
 *   ntmo_gru_forward     code/model.py:67-88 (RNN.forward) and :393-415 (DiffDelRNN.forward, GRU part)
 *                        = torch.nn.GRU(1,H,batch_first=True) + torch.nn.Linear(H,1[,bias=False]),
 *                        a third-party dependency (PyTorch, unpinned in environment.yaml:7-8).  The
 *                        published GRU equations, gate row order (r,z,n), in the operation order of
 *                        ATen's CPU cell:  r=s(gi_r+gh_r) z=s(gi_z+gh_z) n=tanh(gi_n+r*gh_n)
 *                        h'=(h-n)*z+n ;  y=W_o.h'(+b_o)
 *   ntmo_delay_forward   code/model.py:269-320 (TimeVaryingDelayLine.forward) in closed form: of the
 *                        D+1 interpolation weights relu(1-|m-d|) at most the taps m=floor(d), floor(d)+1
 *                        are non-zero; every fp32 operation is kept in the reference's order so the
 *                        result is bit-identical (compile with -ffp-contract=off).
 *   ntmo_esr_sums        the un-vendored CoreAudioML ESRLoss used at code/test-model.py:250-254:
 *                        mean((t-y)^2)/(mean(t^2)+1e-5); this returns the two sums per stream.
 *                        PARITY UNPINNED for this one (source absent from the reference tree).
 *   ntmo_tcn_forward     builder-defined causal dilated Conv1d stack (no reference implementation
 *                        exists: code/micro_tcn is an empty submodule).  PARITY UNPINNED; checked
 *                        against torch.nn.functional.conv1d in tests.
 *
 * All arrays are dense row-major fp32.  Streams are independent; "_mt" variants split streams over
 * OpenMP threads (used only for the cpu_baseline timing).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

/* one stream: x[T] -> y[T], h[H] in/out.  wt = W_hh transposed [H][3H] so the inner loop runs over
 * the 3H outputs (vectorisable) while every output still accumulates its products in k = 0..H-1
 * order, i.e. the same sequential fp32 sum as the plain dot-product form. */
__attribute__((target_clones("avx2", "default")))
static void gru_stream(const float *w_ih, const float *wt, const float *b_ih, const float *b_hh,
                       const float *w_o, const float *b_o, int H, const float *x, float *y,
                       int64_t T, float *h)
{
    const int G = 3 * H;
    float *gh = (float *)malloc(sizeof(float) * (size_t)G);
    float *hn = (float *)malloc(sizeof(float) * (size_t)H);
    for (int64_t t = 0; t < T; ++t) {
        const float xt = x[t];
        for (int g = 0; g < G; ++g) gh[g] = 0.0f;
        for (int k = 0; k < H; ++k) {
            const float hk = h[k];
            const float *col = wt + (size_t)k * G;
            for (int g = 0; g < G; ++g) gh[g] += col[g] * hk;
        }
        for (int g = 0; g < G; ++g) gh[g] += b_hh[g];
        float yo = 0.0f;
        for (int j = 0; j < H; ++j) {
            const float gi_r = w_ih[j] * xt + b_ih[j];
            const float gi_z = w_ih[H + j] * xt + b_ih[H + j];
            const float gi_n = w_ih[2 * H + j] * xt + b_ih[2 * H + j];
            const float r = sigmoidf_(gi_r + gh[j]);
            const float z = sigmoidf_(gi_z + gh[H + j]);
            const float n = tanhf(gi_n + r * gh[2 * H + j]);
            hn[j] = (h[j] - n) * z + n;
            yo += w_o[j] * hn[j];
        }
        memcpy(h, hn, sizeof(float) * (size_t)H);
        y[t] = b_o ? yo + b_o[0] : yo;
    }
    free(gh);
    free(hn);
}

static float *transpose_whh(const float *w_hh, int H)
{
    const int G = 3 * H;
    float *wt = (float *)malloc(sizeof(float) * (size_t)G * (size_t)H);
    for (int g = 0; g < G; ++g)
        for (int k = 0; k < H; ++k) wt[(size_t)k * G + g] = w_hh[(size_t)g * H + k];
    return wt;
}

/* code/model.py:81-82.  h_state [B,H] in/out (caller passes zeros for hidden=None). */
int ntmo_gru_forward(const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                     const float *w_o, const float *b_o, int H, const float *x, float *y,
                     int64_t B, int64_t T, float *h_state)
{
    if (H <= 0 || B < 0 || T < 0) return -1;
    float *wt = transpose_whh(w_hh, H);
    for (int64_t b = 0; b < B; ++b)
        gru_stream(w_ih, wt, b_ih, b_hh, w_o, b_o, H, x + b * T, y + b * T, T, h_state + b * H);
    free(wt);
    return 0;
}

int ntmo_gru_forward_mt(const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                        const float *w_o, const float *b_o, int H, const float *x, float *y,
                        int64_t B, int64_t T, float *h_state, int threads)
{
    if (H <= 0 || B < 0 || T < 0) return -1;
    float *wt = transpose_whh(w_hh, H);
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int64_t b = 0; b < B; ++b)
        gru_stream(w_ih, wt, b_ih, b_hh, w_o, b_o, H, x + b * T, y + b * T, T, h_state + b * H);
    free(wt);
    return 0;
}

/*
 * General input_size / output_size (code/model.py:22,44-45: nn.GRU(input_size, H, batch_first=True) + nn.Linear(H, output_size)).
 * The reference moves between (B, C, T) and (B, T, C) with `reshape` (code/model.py:77,87) -- a reinterpretation of the row, not a
 * transpose -- so per stream the input row of C T floats IS the [T][I] matrix the GRU reads, and the [T][O] matrix the head writes IS
 * the output row: x [B, T*I], y [B, T*O].  w_ih [3H, I], w_o [O, H], b_o [O] or NULL.  Same operation order as gru_stream
 * (the input products of a gate are summed in i = 0..I-1 order on top of b_ih).  No caller of the reference uses sizes other
 * than 1 (code/test-model.py:124-125, code/train.py:87-88); pinned by golden g22 from the reference's own forward().
 */
int ntmo_gru_forward_io(const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh, const float *w_o,
                        const float *b_o, int H, int I, int O, const float *x, float *y, int64_t B, int64_t T, float *h_state)
{
    if (H <= 0 || I <= 0 || O <= 0 || B < 0 || T < 0) return -1;
    const int G = 3 * H;
    float *wt = transpose_whh(w_hh, H);
    float *gh = (float *)malloc(sizeof(float) * (size_t)G);
    float *gi = (float *)malloc(sizeof(float) * (size_t)G);
    float *hn = (float *)malloc(sizeof(float) * (size_t)H);
    for (int64_t b = 0; b < B; ++b) {
        float *h = h_state + b * H;
        const float *xb = x + b * T * I;
        float *yb = y + b * T * O;
        for (int64_t t = 0; t < T; ++t) {
            for (int g = 0; g < G; ++g) gh[g] = 0.0f;
            for (int k = 0; k < H; ++k) {
                const float hk = h[k];
                const float *col = wt + (size_t)k * G;
                for (int g = 0; g < G; ++g) gh[g] += col[g] * hk;
            }
            for (int g = 0; g < G; ++g) {
                gh[g] += b_hh[g];
                float acc = 0.0f;
                for (int i = 0; i < I; ++i) acc += w_ih[(size_t)g * I + i] * xb[t * I + i];
                gi[g] = acc + b_ih[g];
            }
            for (int j = 0; j < H; ++j) {
                const float r = sigmoidf_(gi[j] + gh[j]);
                const float z = sigmoidf_(gi[H + j] + gh[H + j]);
                const float n = tanhf(gi[2 * H + j] + r * gh[2 * H + j]);
                hn[j] = (h[j] - n) * z + n;
            }
            memcpy(h, hn, sizeof(float) * (size_t)H);
            for (int o = 0; o < O; ++o) {
                float yo = 0.0f;
                for (int j = 0; j < H; ++j) yo += w_o[(size_t)o * H + j] * h[j];
                yb[t * O + o] = b_o ? yo + b_o[o] : yo;
            }
        }
    }
    free(wt); free(gh); free(gi); free(hn);
    return 0;
}

/*
 * fp64 mode of the same restatement (code/model.py:81-82 evaluated in double: fp32 parameters and input, `double` state,
 * accumulators, exp / tanh): the yardstick that separates the device's rounding from the fp32 oracle's own -- per stream
 * |hip - f64| against |oracle32 - f64| (tests/test_gpu_round6.py).  Not a parity target: the reference computes in fp32
 * (model.py:76 casts to float); pinned to torch's double GRU by golden g6's fp64 twin (tests/test_oracle.py).
 * y [B,T] and h_state [B,H] are double.
 */
static void gru_stream_f64(const float *w_ih, const double *wt, const float *b_ih, const float *b_hh,
                           const float *w_o, const float *b_o, int H, const float *x, double *y, int64_t T, double *h)
{
    const int G = 3 * H;
    double *gh = (double *)malloc(sizeof(double) * (size_t)G);
    double *hn = (double *)malloc(sizeof(double) * (size_t)H);
    for (int64_t t = 0; t < T; ++t) {
        const double xt = (double)x[t];
        for (int g = 0; g < G; ++g) gh[g] = 0.0;
        for (int k = 0; k < H; ++k) {
            const double hk = h[k];
            const double *col = wt + (size_t)k * G;
            for (int g = 0; g < G; ++g) gh[g] += col[g] * hk;
        }
        for (int g = 0; g < G; ++g) gh[g] += (double)b_hh[g];
        double yo = 0.0;
        for (int j = 0; j < H; ++j) {
            const double gi_r = (double)w_ih[j] * xt + (double)b_ih[j];
            const double gi_z = (double)w_ih[H + j] * xt + (double)b_ih[H + j];
            const double gi_n = (double)w_ih[2 * H + j] * xt + (double)b_ih[2 * H + j];
            const double r = 1.0 / (1.0 + exp(-(gi_r + gh[j])));
            const double z = 1.0 / (1.0 + exp(-(gi_z + gh[H + j])));
            const double n = tanh(gi_n + r * gh[2 * H + j]);
            hn[j] = (h[j] - n) * z + n;
            yo += (double)w_o[j] * hn[j];
        }
        memcpy(h, hn, sizeof(double) * (size_t)H);
        y[t] = b_o ? yo + (double)b_o[0] : yo;
    }
    free(gh);
    free(hn);
}

int ntmo_gru_forward_f64_mt(const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                            const float *w_o, const float *b_o, int H, const float *x, double *y,
                            int64_t B, int64_t T, double *h_state, int threads)
{
    if (H <= 0 || B < 0 || T < 0) return -1;
    const int G = 3 * H;
    double *wt = (double *)malloc(sizeof(double) * (size_t)G * (size_t)H);
    for (int g = 0; g < G; ++g)
        for (int k = 0; k < H; ++k) wt[(size_t)k * G + g] = (double)w_hh[(size_t)g * H + k];
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
    for (int64_t b = 0; b < B; ++b)
        gru_stream_f64(w_ih, wt, b_ih, b_hh, w_o, b_o, H, x + b * T, y + b * T, T, h_state + b * H);
    free(wt);
    return 0;
}

/* the delay line (code/model.py:269-320, closed form as below) on a double signal: the trajectory d stays the caller's fp32
 * array (it is an INPUT), tap index and the two weights are evaluated in double.  Same return convention as ntmo_delay_forward. */
int ntmo_delay_forward_f64(const double *x, const float *d, double *y, int64_t B, int64_t T, double *dl_state, int D, int warmup)
{
    if (D < 0 || B < 0 || T < 0) return -1;
    for (int64_t i = 0; i < B * T; ++i)
        if (d[i] > (float)D) return 1;
    double *nb = (double *)malloc(sizeof(double) * (size_t)(D > 0 ? D : 1));
    for (int64_t b = 0; b < B; ++b) {
        const double *xb = x + b * T;
        const float *db = d + b * T;
        double *yb = y + b * T, *buf = dl_state + b * (int64_t)D;
        if (warmup) {
            if (yb != xb) memcpy(yb, xb, sizeof(double) * (size_t)T);
        } else {
            for (int64_t n = 0; n < T; ++n) {
                const double dn = (double)db[n];
                const int64_t k = (int64_t)floor(dn);
                double acc = 0.0;
                for (int64_t m = k + 1; m >= k; --m) {
                    if (m < 0 || m > D) continue;
                    const double w = 1.0 - fabs((double)m - dn);
                    if (!(w > 0.0)) continue;
                    const int64_t src = n - m;
                    acc += w * (src >= 0 ? xb[src] : buf[D + src]);
                }
                yb[n] = acc;
            }
        }
        if (T >= D) {
            memcpy(nb, xb + (T - D), sizeof(double) * (size_t)D);
        } else {
            memcpy(nb, buf + T, sizeof(double) * (size_t)(D - T));
            memcpy(nb + (D - T), xb, sizeof(double) * (size_t)T);
        }
        memcpy(buf, nb, sizeof(double) * (size_t)D);
    }
    free(nb);
    return 0;
}

/*
 * code/model.py:269-320.  x,d,y [B,T]; dl_state [B,D] = the reference's `buffer` (oldest first).
 * Returns 0, or 1 if max(d) > D (the reference's `assert self.max_delay >= torch.max(dt)`, :284;
 * nothing is written in that case).  y may not alias x.
 */
int ntmo_delay_forward(const float *x, const float *d, float *y, int64_t B, int64_t T,
                       float *dl_state, int D, int warmup)
{
    if (D < 0 || B < 0 || T < 0) return -1;
    for (int64_t i = 0; i < B * T; ++i)
        if (d[i] > (float)D) return 1;
    float *nb = (float *)malloc(sizeof(float) * (size_t)(D > 0 ? D : 1));
    for (int64_t b = 0; b < B; ++b) {
        const float *xb = x + b * T, *db = d + b * T;
        float *yb = y + b * T, *buf = dl_state + b * (int64_t)D;
        if (warmup) {
            if (yb != xb) memcpy(yb, xb, sizeof(float) * (size_t)T);     /* :288-292 returns x */
        } else {
            for (int64_t n = 0; n < T; ++n) {
                const float dn = db[n];
                const int64_t k = (int64_t)floorf(dn);
                float acc = 0.0f;
                /* j ascending in the reference <=> tap delay m descending: m=k+1 first, then m=k */
                for (int64_t m = k + 1; m >= k; --m) {
                    if (m < 0 || m > D) continue;
                    float w = 1.0f - fabsf((float)m - dn);
                    if (!(w > 0.0f)) continue;
                    const int64_t src = n - m;
                    const float xv = src >= 0 ? xb[src] : buf[D + src];
                    acc = acc + w * xv;
                }
                yb[n] = acc;
            }
        }
        /* buffer = cat(buffer[T:], x[-D:])   (:314-315) */
        if (T >= D) {
            memcpy(nb, xb + (T - D), sizeof(float) * (size_t)D);
        } else {
            memcpy(nb, buf + T, sizeof(float) * (size_t)(D - T));
            memcpy(nb + (D - T), xb, sizeof(float) * (size_t)T);
        }
        memcpy(buf, nb, sizeof(float) * (size_t)D);
    }
    free(nb);
    return 0;
}

/* per-stream sums over samples [skip,T): out[b*2+0]=sum (t-y)^2, out[b*2+1]=sum t^2 (fp64) */
int ntmo_esr_sums(const float *y, const float *t, int64_t B, int64_t T, int64_t skip, double *out)
{
    if (skip < 0 || skip > T) return -1;
    for (int64_t b = 0; b < B; ++b) {
        double se = 0.0, st = 0.0;
        for (int64_t n = skip; n < T; ++n) {
            const float e = t[b * T + n] - y[b * T + n];
            se += (double)e * (double)e;
            st += (double)t[b * T + n] * (double)t[b * T + n];
        }
        out[b * 2 + 0] = se;
        out[b * 2 + 1] = st;
    }
    return 0;
}

/*
 * DC-pre-emphasised ESR sums (GreyBoxDRC loss_funcs.ESRLoss(dc_pre=True), un-vendored: PARITY UNPINNED).
 * Both signals pass the DC blocker  H(z) = (1 - z^-1) / (1 - R z^-1)  (zero state at sample `skip`, the
 * reference cuts INIT_LEN before calling the loss) and the ESR sums are taken on the filtered signals.
 * Upstream realises H as its impulse response truncated to 2000 taps; R^2000 = 4.4e-5 for R = 0.995, the
 * recursion below is the untruncated filter.  The filter is linear, so f(t) - f(y) = f(t - y).
 * out[2b] = sum f(t-y)^2, out[2b+1] = sum f(t)^2 (fp64 sums of the fp32 filter outputs).
 */
int ntmo_esr_dcpre_sums(const float *y, const float *t, int64_t B, int64_t T, int64_t skip, float R, double *out)
{
    if (skip < 0 || skip > T) return -1;
    for (int64_t b = 0; b < B; ++b) {
        double se = 0.0, st = 0.0;
        float fe = 0.0f, ft = 0.0f, pe = 0.0f, pt = 0.0f;
        for (int64_t n = skip; n < T; ++n) {
            const float tv = t[b * T + n], e = tv - y[b * T + n];
            fe = (e - pe) + R * fe;
            ft = (tv - pt) + R * ft;
            pe = e; pt = tv;
            se += (double)fe * (double)fe;
            st += (double)ft * (double)ft;
        }
        out[b * 2 + 0] = se;
        out[b * 2 + 1] = st;
    }
    return 0;
}

/*
 * Jiles-Atherton magnetisation stage of the reference's white-box tape simulator ("next" row N4):
 * code/tape.py:516-551 (Tape.H_mag: trapezoidal dH/dt, RK4, clamp to +-Ms) with code/tape.py:587-635
 * (Tape._f), fp64, operation for operation -- including the reference's L'(.) evaluated on L(Q) instead of
 * on Q (:598-603).  PINNED by tests/golden/g9_tape_hmag.npz (generated from the reference's own H_mag).
 * H, M [B,N]; state [B,3] = (M_prev, H_prev, Hprime_prev) in/out; par = {Ms, A, alpha, K, c}.
 */
static double ja_f(double Mn, double Hn, double Hp, const double *par)
{
    const double Ms = par[0], A = par[1], alpha = par[2], K = par[3], c = par[4];
    const double Q = (Hn + alpha * Mn) / A;
    const double LQ = fabs(Q) > 1e-4 ? (1.0 / tanh(Q)) - 1.0 / Q : Q / 3.0;
    double LpQ;
    if (fabs(LQ) > 1e-4) { const double ct = 1.0 / tanh(LQ); LpQ = 1.0 / (LQ * LQ) - ct * ct + 1.0; } else LpQ = 1.0 / 3.0;
    const double M_diff = Ms * LQ - Mn;
    const double dS = Hp > 0.0 ? 1.0 : -1.0;
    const double sgn = M_diff > 0.0 ? 1.0 : (M_diff < 0.0 ? -1.0 : 0.0);
    const double dM = (dS == sgn) ? 1.0 : 0.0;
    const double t1n = (1.0 - c) * dM * M_diff;
    const double t1d = (1.0 - c) * dS * K - alpha * M_diff;
    const double t1 = (t1n / t1d) * Hp;
    const double t2 = c * (Ms / A) * Hp * LpQ;
    const double t3 = 1.0 - c * alpha * (Ms / A) * LpQ;
    return (t1 + t2) / t3;
}

int ntmo_tape_hmag(const double *H, double *M, int64_t B, int64_t N, double *state, double Ts, const double *par)
{
    for (int64_t b = 0; b < B; ++b) {
        double Mp = state[3 * b], Hpv = state[3 * b + 1], Hpp = state[3 * b + 2];
        for (int64_t n = 0; n < N; ++n) {
            const double Hn = H[b * N + n];
            const double Hprime = 2.0 * (Hn - Hpv) / Ts - Hpp;
            const double k1 = Ts * ja_f(Mp, Hpv, Hpp, par);
            const double k2 = Ts * ja_f(Mp + k1 / 2.0, (Hn + Hpv) / 2.0, (Hprime + Hpp) / 2.0, par);
            const double k3 = Ts * ja_f(Mp + k2 / 2.0, (Hn + Hpv) / 2.0, (Hprime + Hpp) / 2.0, par);
            const double k4 = Ts * ja_f(Mp + k3, Hn, Hprime, par);
            double m = Mp + k1 / 6.0 + k2 / 3.0 + k3 / 3.0 + k4 / 6.0;
            m = m < -par[0] ? -par[0] : (m > par[0] ? par[0] : m);
            M[b * N + n] = m;
            Hpv = Hn; Hpp = Hprime; Mp = m;
        }
        state[3 * b] = Mp; state[3 * b + 1] = Hpv; state[3 * b + 2] = Hpp;
    }
    return 0;
}

/*
 * Builder-defined TCN (DESIGN.md "K4"): L causal blocks; block i:
 *   u = causal_dilated_conv1d(in, W_i[C_out,C_in,K], b_i, dilation dil[i])   (zero history)
 *   v = PReLU(u, a_i[C_out])
 *   out = v + (res_w_i ? conv1x1(in, res_w_i[C_out,C_in]) : in)
 * followed by a 1x1 output conv (C->1, bias).  x [B,T] (C_in of block 0 is 1), y [B,T].
 * Parameters are packed block after block: W[C_in][K][C_out], b[C_out], a[C_out], res_w[C_in][C_out]
 * (always present), then out_w[C], out_b[1]  (output channel fastest: the layout the kernel streams).
 */
static void tcn_stream(const float *params, int L, int C, int K, const int *dil, const float *x, float *y, int64_t T)
{
    float *a = (float *)malloc(sizeof(float) * (size_t)C * (size_t)T);
    float *c = (float *)malloc(sizeof(float) * (size_t)C * (size_t)T);
    int cin = 1;
    const float *p = params;
    memcpy(a, x, sizeof(float) * (size_t)T);
    for (int l = 0; l < L; ++l) {
        const float *W = p;            p += (size_t)C * cin * K;
        const float *bias = p;         p += C;
        const float *alpha = p;        p += C;
        const float *rw = p;           p += (size_t)C * cin;
        for (int co = 0; co < C; ++co) {
            for (int64_t n = 0; n < T; ++n) {
                float u = bias[co];
                for (int ci = 0; ci < cin; ++ci)
                    for (int k = 0; k < K; ++k) {
                        const int64_t src = n - (int64_t)(K - 1 - k) * dil[l];
                        if (src >= 0) u += W[((size_t)ci * K + k) * C + co] * a[(size_t)ci * T + src];
                    }
                const float v = u >= 0.0f ? u : alpha[co] * u;
                float r = 0.0f;
                for (int ci = 0; ci < cin; ++ci) r += rw[(size_t)ci * C + co] * a[(size_t)ci * T + n];
                c[(size_t)co * T + n] = v + r;
            }
        }
        float *tmp = a; a = c; c = tmp;
        cin = C;
    }
    const float *ow = p, *ob = p + C;
    for (int64_t n = 0; n < T; ++n) {
        float acc = ob[0];
        for (int ci = 0; ci < C; ++ci) acc += ow[ci] * a[(size_t)ci * T + n];
        y[n] = acc;
    }
    free(a);
    free(c);
}

int ntmo_tcn_forward(const float *params, int L, int C, int K, const int *dil, const float *x,
                     float *y, int64_t B, int64_t T)
{
    if (L <= 0 || C <= 0 || K <= 0) return -1;
    for (int64_t b = 0; b < B; ++b) tcn_stream(params, L, C, K, dil, x + b * T, y + b * T, T);
    return 0;
}

/* streams over OpenMP threads (the full-size parity tests check several 65 536-sample streams) */
int ntmo_tcn_forward_mt(const float *params, int L, int C, int K, const int *dil, const float *x,
                        float *y, int64_t B, int64_t T, int threads)
{
    if (L <= 0 || C <= 0 || K <= 0) return -1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int64_t b = 0; b < B; ++b) tcn_stream(params, L, C, K, dil, x + b * T, y + b * T, T);
    return 0;
}
