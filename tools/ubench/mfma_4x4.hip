// Micro-benchmark for the wave-per-4-streams GRU mapping: v_mfma_f32_4x4x1_16B_f32 on gfx950.
//  (1) operand layout: A lane 4*blk+i = A[blk][i], B lane 4*blk+j = B[blk][j], D lane 4*blk+j reg i = D[blk][i][j]?
//  (2) BLGP = 4+g broadcasts the B lanes of 16-lane group g to all four groups?
//  (3) issue rate: back to back on 1 / 3 accumulator chains, with and without BLGP, A operands from a large register
//      array (192 live registers, as the resident weights would be)
//  (4) drain: cycles from the last MFMA of a chain to a VALU op that reads its result
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const float *A, const float *B, float *D, int blgp)
{
    const int l = threadIdx.x;
    f32x4 c = {0, 0, 0, 0};
    switch (blgp) {
        case 0: c = __builtin_amdgcn_mfma_f32_4x4x1f32(A[l], B[l], c, 0, 0, 0); break;
        case 3: c = __builtin_amdgcn_mfma_f32_4x4x1f32(A[l], B[l], c, 0, 0, 3); break;
        case 4: c = __builtin_amdgcn_mfma_f32_4x4x1f32(A[l], B[l], c, 0, 0, 4); break;
        case 5: c = __builtin_amdgcn_mfma_f32_4x4x1f32(A[l], B[l], c, 0, 0, 5); break;
        case 6: c = __builtin_amdgcn_mfma_f32_4x4x1f32(A[l], B[l], c, 0, 0, 6); break;
        case 7: c = __builtin_amdgcn_mfma_f32_4x4x1f32(A[l], B[l], c, 0, 0, 7); break;
    }
    for (int i = 0; i < 4; ++i) D[l * 4 + i] = c[i];
}

template <int CHAINS, int BLGP, int NW>
__global__ __launch_bounds__(64) void rate_kernel(const float *w, float *out, unsigned long long *cyc, int iters)
{
    float W[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) W[i] = w[i * 64 + threadIdx.x];
    f32x4 acc[3] = {{0, 0, 0, 0}, {1, 1, 1, 1}, {2, 2, 2, 2}};
    float b = 1.0f + threadIdx.x * 0.002f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NW; ++m) {
            acc[m % CHAINS] = __builtin_amdgcn_mfma_f32_4x4x1f32(W[m], b, acc[m % CHAINS], 0, 0, BLGP);
        }
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(b));
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// drain: NM MFMAs on 3 chains, then one v_add on each result, repeated; compare with the same without the adds
template <bool USE>
__global__ __launch_bounds__(64) void drain_kernel(float *out, unsigned long long *cyc, int iters)
{
    f32x4 acc[3] = {{0, 0, 0, 0}, {1, 1, 1, 1}, {2, 2, 2, 2}};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f, s = 0;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 48; ++m) acc[m % 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[m % 3], 0, 0, 0);
        if (USE) {
            s += acc[0][0] + acc[1][0] + acc[2][0];
            a = s * 1e-30f;
        }
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(a), "+v"(s));
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int CHAINS, int BLGP, int NW> void rate(const char *name, const float *w)
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 64 * 4); hipMalloc(&cyc, 8);
    const int iters = 500;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((rate_kernel<CHAINS, BLGP, NW>), dim3(1), dim3(64), 0, 0, w, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-40s chains=%d blgp=%d A regs=%3d : %.2f cycles per MFMA\n", name, CHAINS, BLGP, NW, (double)h / (iters * (double)NW));
    hipFree(out); hipFree(cyc);
}

int main()
{
    // ---- layout ----
    std::vector<float> A(64), B(64), D(256);
    for (int l = 0; l < 64; ++l) { A[l] = 1 + l; B[l] = 100 + l; }      // distinct primes would be nicer; products are unique enough
    float *dA, *dB, *dD;
    hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice);
    for (int blgp : {0, 3, 4, 5, 6, 7}) {
        hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD, blgp);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
        // hypothesis: D[l][i] = A[4*blk + i] * B[src(l)]   with blk = l/4 and src(l) given by the BLGP pattern
        int ok_plain = 1, src_ok = 1;
        int srcs[64];
        for (int l = 0; l < 64; ++l) {
            const int blk = l / 4;
            // infer the source lane of B from reg 0
            const float bsrc = D[l * 4 + 0] / A[4 * blk + 0];
            srcs[l] = (int)(bsrc + 0.5f) - 100;
            for (int i = 0; i < 4; ++i) if (D[l * 4 + i] != A[4 * blk + i] * bsrc) ok_plain = 0;
            (void)src_ok;
        }
        printf("blgp %d: D[lane][i] == A[4*blk+i] * B[src]: %s; src lanes of lanes 0,1,4,16,17,32,48,63: %d %d %d %d %d %d %d %d\n", blgp,
               ok_plain ? "yes" : "NO", srcs[0], srcs[1], srcs[4], srcs[16], srcs[17], srcs[32], srcs[48], srcs[63]);
    }
    // ---- rates ----
    std::vector<float> w(192 * 64, 0.001f);
    float *dw; hipMalloc(&dw, w.size() * 4); hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    rate<1, 0, 48>("4x4x1 back to back", dw); rate<3, 0, 48>("4x4x1 back to back", dw); rate<3, 4, 48>("4x4x1 back to back", dw);
    rate<3, 0, 192>("4x4x1, 192 resident A registers", dw); rate<3, 5, 192>("4x4x1, 192 resident A registers", dw);
    // ---- drain ----
    for (int use = 0; use < 2; ++use) {
        float *out; unsigned long long *cyc;
        hipMalloc(&out, 256); hipMalloc(&cyc, 8);
        const int iters = 2000;
        for (int r = 0; r < 2; ++r) {
            if (use) hipLaunchKernelGGL(drain_kernel<true>, dim3(1), dim3(64), 0, 0, out, cyc, iters);
            else hipLaunchKernelGGL(drain_kernel<false>, dim3(1), dim3(64), 0, 0, out, cyc, iters);
        }
        hipDeviceSynchronize();
        unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("48 MFMAs (3 chains) %s: %.1f cycles per iteration\n", use ? "+ dependent v_add chain" : "alone", (double)h / iters);
    }
    return 0;
}
