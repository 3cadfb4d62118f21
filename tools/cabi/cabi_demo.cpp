// Stand-alone caller of the C ABI (include/ntm.h) with NO torch and NO Python in the process: raw HIP
// allocations, raw weight file, one ntm_gru_forward call, result written as raw fp32 for the test to compare.
//   usage: cabi_demo <w0.bin> <x.f32> <B> <T> <y_out.f32> <h_out.f32>
// and the rest of the hot path -- DiffDelRNN.forward (code/model.py:393-424) in two chunks with carried state, a
// refused call (pre_d == y), a delay-range violation (sticky flag, state untouched), then the ESR sums of y against
// pre_d (code/test-model.py:386-388):
//   usage: cabi_demo diffdel <w2.bin> <x.f32> <d.f32> <B> <T> <D> <split> <out_prefix>
//          writes <out_prefix>.y / .pre / .h / .buf (fp32), .esr (fp64 [B][2]), prints the flag values
// and the sharded evaluation as a torch-free rank runs it -- forward + ESR sums in one call, this rank's four loss scalars, the
// ONE collective of the path over RCCL (include/ntm_rccl.h; here a communicator of one rank: the box has one GPU):
//   usage: cabi_demo reduce <w0.bin> <x.f32> <target.f32> <B> <T> <skip>      prints the four reduced scalars and the job ESR
// w0.bin layout (neural-tape-modeling_amd/weights/manifest.json): W_ih[192] W_hh[192*64] b_ih[192] b_hh[192]
// W_o[64] b_o[1], little-endian fp32 -- the reference's state_dict order (code/model.py:44-45).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "ntm.h"
#ifndef NTM_NO_RCCL          // (tools/cabi/Makefile: a box without the RCCL development files builds the demo without `reduce`)
#include "ntm_rccl.h"
#endif

static std::vector<float> read_f32(const char *path, size_t want)
{
    std::vector<float> v(want);
    FILE *f = fopen(path, "rb");
    if (!f || fread(v.data(), sizeof(float), want, f) != want) { fprintf(stderr, "cannot read %zu floats from %s\n", want, path); exit(2); }
    fclose(f);
    return v;
}

#define HIP_OK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); return 3; } } while (0)

static void write_raw(const std::string &path, const void *p, size_t bytes)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f || fwrite(p, 1, bytes, f) != bytes) { fprintf(stderr, "cannot write %s\n", path.c_str()); exit(2); }
    fclose(f);
}

static int diffdel_main(int argc, char **argv)
{
    if (argc != 10) { fprintf(stderr, "usage: %s diffdel w2.bin x.f32 d.f32 B T D split out_prefix\n", argv[0]); return 1; }
    const long B = atol(argv[5]), T = atol(argv[6]), D = atol(argv[7]), split = atol(argv[8]);
    const std::string out = argv[9];
    const size_t nw = 192 + 192 * 64 + 192 + 192 + 64;            // DiffDelRNN: the head has no bias (code/model.py:365)
    std::vector<float> w = read_f32(argv[2], nw), x = read_f32(argv[3], (size_t)B * T), d = read_f32(argv[4], (size_t)B * T);
    float *dw, *dx, *dd, *dy, *dp, *dh, *db;
    int32_t *flag;
    double *esr;
    const size_t nb = (size_t)B * T * sizeof(float);
    HIP_OK(hipMalloc(&dw, nw * sizeof(float)));
    HIP_OK(hipMalloc(&dx, nb)); HIP_OK(hipMalloc(&dd, nb)); HIP_OK(hipMalloc(&dy, nb)); HIP_OK(hipMalloc(&dp, nb));
    HIP_OK(hipMalloc(&dh, (size_t)B * 64 * sizeof(float)));
    HIP_OK(hipMalloc(&db, (size_t)B * D * sizeof(float)));
    HIP_OK(hipMalloc(&flag, sizeof(int32_t)));
    HIP_OK(hipMemcpy(dw, w.data(), nw * sizeof(float), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dx, x.data(), nb, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dd, d.data(), nb, hipMemcpyHostToDevice));
    HIP_OK(hipMemset(dh, 0, (size_t)B * 64 * sizeof(float)));
    HIP_OK(hipMemset(db, 0, (size_t)B * D * sizeof(float)));
    HIP_OK(hipMemset(flag, 0, sizeof(int32_t)));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    const float *w_ih = dw, *w_hh = dw + 192, *b_ih = w_hh + 192 * 64, *b_hh = b_ih + 192, *w_o = b_hh + 192;
    // the audio arrays of this entry point are contiguous [B,T]: the two time chunks are separate arrays
    std::vector<float> xa((size_t)B * split), xb((size_t)B * (T - split)), da(xa.size()), dbv(xb.size());
    for (long b = 0; b < B; ++b)
        for (long n = 0; n < T; ++n) {
            float &xs = n < split ? xa[b * split + n] : xb[b * (T - split) + (n - split)];
            float &ds = n < split ? da[b * split + n] : dbv[b * (T - split) + (n - split)];
            xs = x[b * T + n]; ds = d[b * T + n];
        }
    float *cx[2] = {dx, dx + B * split}, *cd[2] = {dd, dd + B * split}, *cy[2] = {dy, dy + B * split}, *cp[2] = {dp, dp + B * split};
    HIP_OK(hipMemcpy(cx[0], xa.data(), xa.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(cx[1], xb.data(), xb.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(cd[0], da.data(), da.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(cd[1], dbv.data(), dbv.size() * sizeof(float), hipMemcpyHostToDevice));
    const long len[2] = {split, T - split};
    // refused: pre_d must be a buffer of its own
    if (ntm_diffdel_gru_forward(w_ih, w_hh, b_ih, b_hh, w_o, NTM_HIDDEN, cx[0], cd[0], cy[0], cy[0], B, len[0], dh, db, (int)D, 0,
                                flag, stream) != NTM_EINVAL) return 6;
    const std::string refusal = ntm_last_error();
    for (int c = 0; c < 2; ++c) {
        int rc = ntm_diffdel_gru_forward(w_ih, w_hh, b_ih, b_hh, w_o, NTM_HIDDEN, cx[c], cd[c], cy[c], cp[c], B, len[c], dh, db,
                                         (int)D, 0, flag, stream);
        if (rc != NTM_OK) { fprintf(stderr, "ntm_diffdel_gru_forward: %d %s\n", rc, ntm_last_error()); return 5; }
    }
    HIP_OK(hipStreamSynchronize(stream));
    int32_t f0 = -1, f1 = -1;
    HIP_OK(hipMemcpy(&f0, flag, sizeof(f0), hipMemcpyDeviceToHost));
    std::vector<float> y((size_t)B * T), pre(y.size()), h((size_t)B * 64), buf((size_t)B * D), ya(xa.size()), yb(xb.size()), pa(xa.size()), pb(xb.size());
    HIP_OK(hipMemcpy(ya.data(), cy[0], ya.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(yb.data(), cy[1], yb.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(pa.data(), cp[0], pa.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(pb.data(), cp[1], pb.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (long b = 0; b < B; ++b)
        for (long n = 0; n < T; ++n) {
            y[b * T + n] = n < split ? ya[b * split + n] : yb[b * (T - split) + (n - split)];
            pre[b * T + n] = n < split ? pa[b * split + n] : pb[b * (T - split) + (n - split)];
        }
    HIP_OK(hipMemcpy(h.data(), dh, h.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(buf.data(), db, buf.size() * sizeof(float), hipMemcpyDeviceToHost));
    // ESR sums of the second chunk's y against its pre_d, straight from the device arrays (splits = what the library suggests)
    const int splits = ntm_esr_splits(B, len[1], 0);
    HIP_OK(hipMalloc(&esr, (size_t)B * splits * 2 * sizeof(double)));
    if (ntm_esr_sums(cy[1], cp[1], B, len[1], 0, splits, esr, stream) != NTM_OK) { fprintf(stderr, "ntm_esr_sums: %s\n", ntm_last_error()); return 7; }
    HIP_OK(hipStreamSynchronize(stream));
    std::vector<double> part((size_t)B * splits * 2), sums((size_t)B * 2, 0.0);
    HIP_OK(hipMemcpy(part.data(), esr, part.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (long b = 0; b < B; ++b)
        for (int p = 0; p < splits; ++p) { sums[2 * b] += part[(b * splits + p) * 2]; sums[2 * b + 1] += part[(b * splits + p) * 2 + 1]; }
    // a delay beyond D: the flag goes up, y of that call is unspecified, h is updated (the GRU ran), the delay buffer is NOT
    std::vector<float> dbad(da);
    dbad[dbad.size() / 2] = (float)D + 0.5f;
    HIP_OK(hipMemcpy(cd[0], dbad.data(), dbad.size() * sizeof(float), hipMemcpyHostToDevice));
    if (ntm_diffdel_gru_forward(w_ih, w_hh, b_ih, b_hh, w_o, NTM_HIDDEN, cx[0], cd[0], cy[0], cp[0], B, len[0], dh, db, (int)D, 0, flag,
                                stream) != NTM_OK) return 8;
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(&f1, flag, sizeof(f1), hipMemcpyDeviceToHost));
    std::vector<float> buf2(buf.size());
    HIP_OK(hipMemcpy(buf2.data(), db, buf2.size() * sizeof(float), hipMemcpyDeviceToHost));
    const bool frozen = buf2 == buf;
    write_raw(out + ".y", y.data(), y.size() * sizeof(float));
    write_raw(out + ".pre", pre.data(), pre.size() * sizeof(float));
    write_raw(out + ".h", h.data(), h.size() * sizeof(float));
    write_raw(out + ".buf", buf.data(), buf.size() * sizeof(float));
    write_raw(out + ".esr", sums.data(), sums.size() * sizeof(double));
    printf("ok diffdel B=%ld T=%ld D=%ld flag_after_good_calls=%d flag_after_violation=%d buffer_frozen=%d refusal=\"%s\"\n", B, T, D, f0, f1,
           (int)frozen, refusal.c_str());
    return 0;
}

#ifdef NTM_NO_RCCL
static int reduce_main(int, char **) { fprintf(stderr, "built without RCCL (rccl/rccl.h absent at build time): no `reduce`\n"); return 16; }
#else
static int reduce_main(int argc, char **argv)
{
    if (argc != 8) { fprintf(stderr, "usage: %s reduce w0.bin x.f32 target.f32 B T skip\n", argv[0]); return 1; }
    const long B = atol(argv[5]), T = atol(argv[6]), skip = atol(argv[7]);
    const size_t nw = 192 + 192 * 64 + 192 + 192 + 64 + 1;
    std::vector<float> w = read_f32(argv[2], nw), x = read_f32(argv[3], (size_t)B * T), t = read_f32(argv[4], (size_t)B * T);
    float *dw, *dx, *dt, *dy, *dh;
    double *esr, *v4;
    const size_t nb = (size_t)B * T * sizeof(float);
    HIP_OK(hipMalloc(&dw, nw * sizeof(float)));
    HIP_OK(hipMalloc(&dx, nb)); HIP_OK(hipMalloc(&dt, nb)); HIP_OK(hipMalloc(&dy, nb));
    HIP_OK(hipMalloc(&dh, (size_t)B * 64 * sizeof(float)));
    HIP_OK(hipMalloc(&esr, (size_t)B * 2 * sizeof(double)));
    HIP_OK(hipMalloc(&v4, 4 * sizeof(double)));
    HIP_OK(hipMemcpy(dw, w.data(), nw * sizeof(float), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dx, x.data(), nb, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dt, t.data(), nb, hipMemcpyHostToDevice));
    HIP_OK(hipMemset(dh, 0, (size_t)B * 64 * sizeof(float)));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    const float *w_ih = dw, *w_hh = dw + 192, *b_ih = w_hh + 192 * 64, *b_hh = b_ih + 192, *w_o = b_hh + 192, *b_o = w_o + 64;
    // communicator: rank 0 makes the id; with more ranks the launcher would ship these 128 bytes to the others
    unsigned char id[NTM_RCCL_ID_BYTES];
    void *comm = nullptr;
    if (ntm_rccl_unique_id(id) != 0 || ntm_rccl_comm_create(&comm, 1, 0, id) != 0) { fprintf(stderr, "rccl: %s\n", ntm_rccl_last_error()); return 10; }
    if (ntm_rccl_comm_create(&comm, 2, 2, id) == 0) return 11;                    // rank out of range: refused before RCCL is called
    if (ntm_gru_forward_esr(w_ih, w_hh, b_ih, b_hh, w_o, b_o, NTM_HIDDEN, dx, dy, B, T, T, T, dh, dt, skip, esr, stream) != NTM_OK) {
        fprintf(stderr, "ntm_gru_forward_esr: %s\n", ntm_last_error()); return 7;
    }
    if (ntm_loss_scalars(esr, B, T - skip, 1e-5, v4, stream) != NTM_OK) { fprintf(stderr, "ntm_loss_scalars: %s\n", ntm_last_error()); return 12; }
    if (ntm_loss_scalars(esr, B, 0, 1e-5, v4, stream) == NTM_OK) return 13;       // no samples: refused
    if (ntm_rccl_allreduce_f64(v4, 4, comm, stream) != 0) { fprintf(stderr, "rccl: %s\n", ntm_rccl_last_error()); return 14; }
    HIP_OK(hipStreamSynchronize(stream));
    double h4[4];
    HIP_OK(hipMemcpy(h4, v4, sizeof(h4), hipMemcpyDeviceToHost));
    if (ntm_rccl_comm_destroy(comm) != 0) { fprintf(stderr, "rccl: %s\n", ntm_rccl_last_error()); return 15; }
    printf("ok reduce ranks=1 sum_esr=%.17g segments=%.17g sum_err2=%.17g sum_tgt2=%.17g job_esr=%.17g\n", h4[0], h4[1], h4[2], h4[3], h4[0] / h4[1]);
    return 0;
}
#endif

int main(int argc, char **argv)
{
    if (argc > 1 && std::string(argv[1]) == "diffdel") return diffdel_main(argc, argv);
    if (argc > 1 && std::string(argv[1]) == "reduce") return reduce_main(argc, argv);
    if (argc != 7) { fprintf(stderr, "usage: %s w0.bin x.f32 B T y_out.f32 h_out.f32\n", argv[0]); return 1; }
    const long B = atol(argv[3]), T = atol(argv[4]);
    const size_t nw = 192 + 192 * 64 + 192 + 192 + 64 + 1;
    std::vector<float> w = read_f32(argv[1], nw), x = read_f32(argv[2], (size_t)B * T);
    float *dw, *dx, *dy, *dh;
    HIP_OK(hipMalloc(&dw, nw * sizeof(float)));
    HIP_OK(hipMalloc(&dx, x.size() * sizeof(float)));
    HIP_OK(hipMalloc(&dy, x.size() * sizeof(float)));
    HIP_OK(hipMalloc(&dh, (size_t)B * 64 * sizeof(float)));
    HIP_OK(hipMemcpy(dw, w.data(), nw * sizeof(float), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dx, x.data(), x.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_OK(hipMemset(dh, 0, (size_t)B * 64 * sizeof(float)));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    const float *w_ih = dw, *w_hh = dw + 192, *b_ih = w_hh + 192 * 64, *b_hh = b_ih + 192, *w_o = b_hh + 192, *b_o = w_o + 64;
    if (ntm_abi_version() != NTM_ABI_VERSION) { fprintf(stderr, "unexpected ABI version\n"); return 4; }
    int rc = ntm_gru_forward(w_ih, w_hh, b_ih, b_hh, w_o, b_o, NTM_HIDDEN, dx, dy, B, T, T, T, dh, stream);
    if (rc != NTM_OK) { fprintf(stderr, "ntm_gru_forward: %d %s\n", rc, ntm_last_error()); return 5; }
    // error path: a hidden size outside [1, NTM_MAX_HIDDEN] must be refused with a message, not crash
    if (ntm_gru_forward(w_ih, w_hh, b_ih, b_hh, w_o, b_o, NTM_MAX_HIDDEN + 1, dx, dy, B, T, T, T, dh, stream) == NTM_OK) return 6;
    if (ntm_gru_forward(w_ih, w_hh, b_ih, b_hh, w_o, b_o, 0, dx, dy, B, T, T, T, dh, stream) == NTM_OK) return 6;
    // forward + the ESR sums of the loss loop in ONE call (ntm_gru_forward_esr; target = the input here), against the
    // streaming pass on the same output: the sums agree to fp64 summation order
    double *e1, *e2;
    const int splits = ntm_esr_splits(B, T, 0);
    HIP_OK(hipMalloc(&e1, (size_t)B * 2 * sizeof(double)));
    HIP_OK(hipMalloc(&e2, (size_t)B * splits * 2 * sizeof(double)));
    float *dy2, *dh2;
    HIP_OK(hipMalloc(&dy2, x.size() * sizeof(float)));
    HIP_OK(hipMalloc(&dh2, (size_t)B * 64 * sizeof(float)));
    HIP_OK(hipMemset(dh2, 0, (size_t)B * 64 * sizeof(float)));
    if (ntm_gru_forward_esr(w_ih, w_hh, b_ih, b_hh, w_o, b_o, NTM_HIDDEN, dx, dy2, B, T, T, T, dh2, dx, 0, e1, stream) != NTM_OK) {
        fprintf(stderr, "ntm_gru_forward_esr: %s\n", ntm_last_error()); return 7;
    }
    if (ntm_esr_sums(dy2, dx, B, T, 0, splits, e2, stream) != NTM_OK) return 8;
    HIP_OK(hipStreamSynchronize(stream));
    std::vector<double> v1((size_t)B * 2), v2((size_t)B * splits * 2);
    HIP_OK(hipMemcpy(v1.data(), e1, v1.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(v2.data(), e2, v2.size() * sizeof(double), hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (long b = 0; b < B; ++b)
        for (int c = 0; c < 2; ++c) {
            double ref = 0.0;
            for (int p = 0; p < splits; ++p) ref += v2[(b * splits + p) * 2 + c];
            const double rel = ref != 0.0 ? (v1[b * 2 + c] - ref) / ref : v1[b * 2 + c];
            worst = rel < 0 ? (-rel > worst ? -rel : worst) : (rel > worst ? rel : worst);
        }
    if (worst > 1e-12) { fprintf(stderr, "ntm_gru_forward_esr sums differ from ntm_esr_sums: %g\n", worst); return 9; }
    // forward + BOTH time-domain losses in one call (ntm_gru_forward_losses) against ntm_esr_dcpre_sums on the same output
    double *d1, *d2, *e3;
    HIP_OK(hipMalloc(&d1, (size_t)B * 2 * sizeof(double)));
    HIP_OK(hipMalloc(&d2, (size_t)B * 2 * sizeof(double)));
    HIP_OK(hipMalloc(&e3, (size_t)B * 2 * sizeof(double)));
    HIP_OK(hipMemset(dh2, 0, (size_t)B * 64 * sizeof(float)));
    if (ntm_gru_forward_losses(w_ih, w_hh, b_ih, b_hh, w_o, b_o, NTM_HIDDEN, dx, dy2, B, T, T, T, dh2, dx, 0, e3, 0.995f, d1, stream) != NTM_OK) {
        fprintf(stderr, "ntm_gru_forward_losses: %s\n", ntm_last_error()); return 10;
    }
    if (ntm_esr_dcpre_sums(dy2, dx, B, T, 0, 0.995f, d2, stream) != NTM_OK) return 11;
    HIP_OK(hipStreamSynchronize(stream));
    std::vector<double> w1((size_t)B * 2), w2((size_t)B * 2), w3((size_t)B * 2);
    HIP_OK(hipMemcpy(w1.data(), d1, w1.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(w2.data(), d2, w2.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(w3.data(), e3, w3.size() * sizeof(double), hipMemcpyDeviceToHost));
    double worst_dc = 0.0;
    for (size_t i = 0; i < w1.size(); ++i) {
        const double rel = w2[i] != 0.0 ? (w1[i] - w2[i]) / w2[i] : w1[i];
        worst_dc = rel < 0 ? (-rel > worst_dc ? -rel : worst_dc) : (rel > worst_dc ? rel : worst_dc);
        if (w3[i] != v1[i]) { fprintf(stderr, "ntm_gru_forward_losses: ESR sums differ from ntm_gru_forward_esr's\n"); return 12; }
    }
    if (worst_dc > 1e-5) { fprintf(stderr, "ntm_gru_forward_losses DCPreESR sums differ from ntm_esr_dcpre_sums: %g\n", worst_dc); return 13; }
    HIP_OK(hipStreamSynchronize(stream));
    std::vector<float> y(x.size()), h((size_t)B * 64);
    HIP_OK(hipMemcpy(y.data(), dy, y.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(h.data(), dh, h.size() * sizeof(float), hipMemcpyDeviceToHost));
    FILE *f = fopen(argv[5], "wb"); fwrite(y.data(), sizeof(float), y.size(), f); fclose(f);
    f = fopen(argv[6], "wb"); fwrite(h.data(), sizeof(float), h.size(), f); fclose(f);
    printf("ok B=%ld T=%ld forward_esr_vs_esr_sums_rel=%.1e forward_losses_dcpre_rel=%.1e last_error_after_refusal=\"%s\"\n", B, T, worst,
           worst_dc, ntm_last_error());
    return 0;
}
