// Micro-benchmark (gfx950): how long the serial section between barrier 1 and the first MFMA of the bf16x3 GRU step takes by
// itself (csrc/gru_mfma2.hip step_b, first asm statement): two ds_read_b128, the second and third bf16 pieces of four fp32
// values (16 dependent vector ops), two ds_write_b64, s_waitcnt.  One wave per SIMD, s_memtime around 1000 repetitions.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
#define NEG " neg_lo:[0,1] neg_hi:[0,1]\n\t"
#define SPLIT                                                                                                             \
    "v_lshlrev_b32 v134, 16, v128\n\tv_and_b32 v135, 0xffff0000, v128\n\tv_lshlrev_b32 v136, 16, v129\n\tv_and_b32 v137, 0xffff0000, v129\n\t" \
    "v_pk_add_f32 v[138:139], v[124:125], v[134:135]" NEG "v_pk_add_f32 v[140:141], v[126:127], v[136:137]" NEG             \
    "v_cvt_pk_bf16_f32 v130, v138, v139\n\tv_cvt_pk_bf16_f32 v131, v140, v141\n\t"                                          \
    "v_lshlrev_b32 v134, 16, v130\n\tv_and_b32 v135, 0xffff0000, v130\n\tv_lshlrev_b32 v136, 16, v131\n\tv_and_b32 v137, 0xffff0000, v131\n\t" \
    "v_pk_add_f32 v[138:139], v[138:139], v[134:135]" NEG "v_pk_add_f32 v[140:141], v[140:141], v[136:137]" NEG             \
    "v_cvt_pk_bf16_f32 v132, v138, v139\n\tv_cvt_pk_bf16_f32 v133, v140, v141\n\t"
#define SPLIT_SCALAR                                                                                                      \
    "v_lshlrev_b32 v134, 16, v128\n\tv_and_b32 v135, 0xffff0000, v128\n\tv_lshlrev_b32 v136, 16, v129\n\tv_and_b32 v137, 0xffff0000, v129\n\t" \
    "v_sub_f32 v138, v124, v134\n\tv_sub_f32 v139, v125, v135\n\tv_sub_f32 v140, v126, v136\n\tv_sub_f32 v141, v127, v137\n\t" \
    "v_cvt_pk_bf16_f32 v130, v138, v139\n\tv_cvt_pk_bf16_f32 v131, v140, v141\n\t"                                          \
    "v_lshlrev_b32 v134, 16, v130\n\tv_and_b32 v135, 0xffff0000, v130\n\tv_lshlrev_b32 v136, 16, v131\n\tv_and_b32 v137, 0xffff0000, v131\n\t" \
    "v_sub_f32 v138, v138, v134\n\tv_sub_f32 v139, v139, v135\n\tv_sub_f32 v140, v140, v136\n\tv_sub_f32 v141, v141, v137\n\t" \
    "v_cvt_pk_bf16_f32 v132, v138, v139\n\tv_cvt_pk_bf16_f32 v133, v140, v141\n\t"
// MODE 0: the split alone   1: reads + wait (the LDS read latency)   2: reads, split, writes, wait for the reads (the product's section)
// 3: as 2 with plain v_sub_f32 instead of v_pk_add_f32   4: write b64, wait, barrier (the publication of the hi piece)
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[4096];
    f32x4 h = {0.1f * threadIdx.x, 0.2f, 0.3f, 0.4f};
    f32x2 p0 = {1.0f, 2.0f};
    f32x4 b0, b1;
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned rd = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)lds + 16u * threadIdx.x;
    unsigned long long t0, t1;
    STAMP(t0);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0)
            asm volatile(SPLIT : "+{v[124:127]}"(h), "+{v[128:129]}"(p0)::"v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141");
        else if (MODE == 1)
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:4096\n\ts_waitcnt lgkmcnt(0)" : "=&v"(b0), "=&v"(b1) : "v"(rd) : "memory");
        else if (MODE == 2)
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:4096\n\t" SPLIT
                         "ds_write_b64 %4, v[130:131] offset:8192\n\tds_write_b64 %4, v[132:133] offset:8200\n\ts_waitcnt lgkmcnt(2)"
                         : "=&v"(b0), "=&v"(b1), "+{v[124:127]}"(h), "+{v[128:129]}"(p0) : "v"(rd)
                         : "memory", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141");
        else if (MODE == 3)
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:4096\n\t" SPLIT_SCALAR
                         "ds_write_b64 %4, v[130:131] offset:8192\n\tds_write_b64 %4, v[132:133] offset:8200\n\ts_waitcnt lgkmcnt(2)"
                         : "=&v"(b0), "=&v"(b1), "+{v[124:127]}"(h), "+{v[128:129]}"(p0) : "v"(rd)
                         : "memory", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141");
        else
            asm volatile("ds_write_b64 %1, %0 offset:8192\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"v"(p0), "v"(rd) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    STAMP(t1);
    out[threadIdx.x] = h[0] + p0[0] + b0[0] + b1[1];
    if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
template <int MODE> void run(const char *what)
{
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 1024); (void)hipMalloc(&cyc, 32);
    const int it = 1000;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(256), 0, 0, out, cyc, it);
    (void)hipDeviceSynchronize();
    unsigned long long h[4]; (void)hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
    printf("%-90s %.1f cycles\n", what, (double)h[0] / it);
}
int main()
{
    run<0>("split of 4 values into mid / lo bf16 pieces (16 vector ops, packed subtractions)");
    run<1>("2 x ds_read_b128 + s_waitcnt lgkmcnt(0)");
    run<2>("2 x ds_read_b128, split, 2 x ds_write_b64, s_waitcnt lgkmcnt(2)   (the product's section)");
    run<3>("the same with plain v_sub_f32 instead of v_pk_add_f32");
    run<4>("ds_write_b64, s_waitcnt lgkmcnt(0), s_barrier   (publication of the hi piece)");
    return 0;
}
