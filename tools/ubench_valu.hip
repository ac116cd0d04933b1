// tools/ubench_valu.hip — issue-cost microbenchmarks behind the round-2 FFN design (NOT part of the library or tests):
// packed-f16 FMA, f16 transcendentals and the 16-deep bf16 MFMA, alone and interleaved with v_mfma_f32_16x16x32_bf16,
// at one and two waves per SIMD.   hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o tools/bin/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

template <int MODE> __global__ void k(float* out, long long* cyc, int iters) {
    const int lane = threadIdx.x;
    h2 a0 = {(_Float16)(lane * 0.001f), (_Float16)0.5f}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    h2 w = {(_Float16)0.999f, (_Float16)1.001f}, u = {(_Float16)0.001f, (_Float16)-0.001f};
    float f0 = lane, f1 = lane + 1, f2 = lane + 2, f3 = lane + 3, f4 = 4, f5 = 5, f6 = 6, f7 = 7;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    bf8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (__bf16)(0.01f * (lane + i)); B[i] = (__bf16)(0.02f * (lane - i)); }
    s4 A4 = {1, 2, 3, 4}, B4 = {5, 6, 7, 8};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {  // 16 x v_pk_fma_f16, 8 independent chains
            asm volatile(REP4("v_pk_fma_f16 %0, %0, %8, %9\n v_pk_fma_f16 %1, %1, %8, %9\n v_pk_fma_f16 %2, %2, %8, %9\n v_pk_fma_f16 %3, %3, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w), "v"(u));
        } else if constexpr (MODE == 1) {  // 16 x v_fma_f32
            asm volatile(REP4("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
                         : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(f4), "v"(f5));
        } else if constexpr (MODE == 2) {  // 16 x mfma 16x16x32 bf16, 4 accumulators
            asm volatile(REP4("v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n v_mfma_f32_16x16x32_bf16 %1, %4, %5, %1\n v_mfma_f32_16x16x32_bf16 %2, %4, %5, %2\n v_mfma_f32_16x16x32_bf16 %3, %4, %5, %3\n")
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(A), "v"(B));
        } else if constexpr (MODE == 3) {  // 16 x mfma 16x16x16 bf16
            asm volatile(REP4("v_mfma_f32_16x16x16_bf16 %0, %4, %5, %0\n v_mfma_f32_16x16x16_bf16 %1, %4, %5, %1\n v_mfma_f32_16x16x16_bf16 %2, %4, %5, %2\n v_mfma_f32_16x16x16_bf16 %3, %4, %5, %3\n")
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(A4), "v"(B4));
        } else if constexpr (MODE == 4) {  // 16 x (mfma + 2 pk_fma)
            asm volatile(REP4("v_mfma_f32_16x16x32_bf16 %0, %12, %13, %0\n v_pk_fma_f16 %4, %4, %14, %15\n v_pk_fma_f16 %5, %5, %14, %15\n"
                              "v_mfma_f32_16x16x32_bf16 %1, %12, %13, %1\n v_pk_fma_f16 %6, %6, %14, %15\n v_pk_fma_f16 %7, %7, %14, %15\n"
                              "v_mfma_f32_16x16x32_bf16 %2, %12, %13, %2\n v_pk_fma_f16 %8, %8, %14, %15\n v_pk_fma_f16 %9, %9, %14, %15\n"
                              "v_mfma_f32_16x16x32_bf16 %3, %12, %13, %3\n v_pk_fma_f16 %10, %10, %14, %15\n v_pk_fma_f16 %11, %11, %14, %15\n")
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(A), "v"(B), "v"(w), "v"(u));
        } else if constexpr (MODE == 5) {  // 16 x (mfma + 4 pk_fma)
            asm volatile(REP4("v_mfma_f32_16x16x32_bf16 %0, %12, %13, %0\n v_pk_fma_f16 %4, %4, %14, %15\n v_pk_fma_f16 %5, %5, %14, %15\n v_pk_fma_f16 %6, %6, %14, %15\n v_pk_fma_f16 %7, %7, %14, %15\n"
                              "v_mfma_f32_16x16x32_bf16 %1, %12, %13, %1\n v_pk_fma_f16 %8, %8, %14, %15\n v_pk_fma_f16 %9, %9, %14, %15\n v_pk_fma_f16 %10, %10, %14, %15\n v_pk_fma_f16 %11, %11, %14, %15\n"
                              "v_mfma_f32_16x16x32_bf16 %2, %12, %13, %2\n v_pk_fma_f16 %4, %4, %14, %15\n v_pk_fma_f16 %5, %5, %14, %15\n v_pk_fma_f16 %6, %6, %14, %15\n v_pk_fma_f16 %7, %7, %14, %15\n"
                              "v_mfma_f32_16x16x32_bf16 %3, %12, %13, %3\n v_pk_fma_f16 %8, %8, %14, %15\n v_pk_fma_f16 %9, %9, %14, %15\n v_pk_fma_f16 %10, %10, %14, %15\n v_pk_fma_f16 %11, %11, %14, %15\n")
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(A), "v"(B), "v"(w), "v"(u));
        } else if constexpr (MODE == 6) {  // 16 x v_exp_f16
            asm volatile(REP4("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else if constexpr (MODE == 7) {  // 16 x v_rcp_f16
            asm volatile(REP4("v_rcp_f16 %0, %0\n v_rcp_f16 %1, %1\n v_rcp_f16 %2, %2\n v_rcp_f16 %3, %3\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else if constexpr (MODE == 8) {  // 16 x v_exp_f32
            asm volatile(REP4("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n") : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));
        } else if constexpr (MODE == 9) {  // 16 x v_cvt_pkrtz_f16_f32
            asm volatile(REP4("v_cvt_pkrtz_f16_f32 %0, %4, %5\n v_cvt_pkrtz_f16_f32 %1, %5, %6\n v_cvt_pkrtz_f16_f32 %2, %6, %7\n v_cvt_pkrtz_f16_f32 %3, %7, %4\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3));
        } else if constexpr (MODE == 10) {  // 16 x v_pk_mul_f16
            asm volatile(REP4("v_pk_mul_f16 %0, %0, %4\n v_pk_mul_f16 %1, %1, %4\n v_pk_mul_f16 %2, %2, %4\n v_pk_mul_f16 %3, %3, %4\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w));
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = (float)a0[0] + (float)a1[1] + (float)a2[0] + (float)a3[0] + (float)a4[0] + (float)a5[0] + (float)a6[0] + (float)a7[0]
              + f0 + f1 + f2 + f3 + c0[0] + c1[1] + c2[2] + c3[3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE> void run(const char* name, int per_iter) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    const int iters = 2000;
    for (int threads : {256, 512}) {   // one / two waves per SIMD, one workgroup per CU
        k<MODE><<<256, threads>>>(out, cyc, iters);
        hipDeviceSynchronize();
        long long h[256 * 8];
        hipMemcpy(h, cyc, 256 * (threads / 64) * 8, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 256 * (threads / 64); ++i) s += h[i];
        s /= 256 * (threads / 64);
        printf("%-34s %d waves/SIMD: %7.2f cycles per group (wave view), %7.2f per SIMD\n", name, threads / 256,
               s / ((double)iters * 16 / per_iter) , s / ((double)iters * 16 / per_iter) / (threads / 256));
    }
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("v_pk_fma_f16", 1);
    run<1>("v_fma_f32", 1);
    run<10>("v_pk_mul_f16", 1);
    run<2>("mfma_16x16x32_bf16", 1);
    run<3>("mfma_16x16x16_bf16", 1);
    run<4>("mfma + 2 pk_fma (per mfma)", 1);
    run<5>("mfma + 4 pk_fma (per mfma)", 1);
    run<6>("v_exp_f16", 1);
    run<7>("v_rcp_f16", 1);
    run<8>("v_exp_f32", 1);
    run<9>("v_cvt_pkrtz_f16_f32", 1);
    return 0;
}
