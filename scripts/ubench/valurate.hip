// Microbenchmark: issue rate of the VALU instructions the forward epilogue is made of (gfx950).
// One wave per SIMD (256 blocks x 256 threads), 8 independent dependency chains per lane.
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAINS 8
#define UNROLL 4

template <int OP> __global__ __launch_bounds__(256) void k(int iters, double *out) {
    double d[CHAINS];
    float f[CHAINS];
    int i32[CHAINS];
    for (int c = 0; c < CHAINS; ++c) { d[c] = threadIdx.x * 1e-3 + c; f[c] = threadIdx.x * 1e-3f + c; i32[c] = threadIdx.x + c; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[c]));
                if (OP == 1) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[c]));
                if (OP == 2) asm volatile("v_add_f64 %0, %0, %0" : "+v"(d[c]));
                if (OP == 3) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(d[c]));
                if (OP == 4) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[c]) : "v"(i32[c]));
                if (OP == 5) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i32[c]) : "v"(d[c]));
                if (OP == 6) asm volatile("v_rndne_f64 %0, %0" : "+v"(d[c]));
                if (OP == 7) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[c]) : "v"(i32[c]));
                if (OP == 8) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[c]) : "v"(d[c]));
                if (OP == 9) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[c]) : "v"(f[c]));
                if (OP == 10) asm volatile("v_perm_b32 %0, %0, %0, %0" : "+v"(i32[c]));
                if (OP == 11) asm volatile("v_add_u32 %0, %0, %0" : "+v"(i32[c]));
                if (OP == 12) asm volatile("v_lshl_add_u32 %0, %0, 8, %0" : "+v"(i32[c]));
                if (OP == 13) asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(i32[c]));
                if (OP == 14) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[c]) : "v"(i32[c]));
                if (OP == 15) asm volatile("v_exp_f32 %0, %0" : "+v"(f[c]));
                if (OP == 16) asm volatile("v_mad_i64_i32 %0, vcc, %1, %1, %0" : "+v"(d[c]) : "v"(i32[c]) : "vcc");
                if (OP == 17) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(d[c]));
                if (OP == 18) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(i32[c]) : "v"(f[c]));
                if (OP == 19) asm volatile("v_rndne_f32 %0, %0" : "+v"(f[c]));
                if (OP == 20) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(f[c]) : "v"(i32[c]));
                if (OP == 21) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(d[c]));
                if (OP == 22) asm volatile("v_xor_b32 %0, %0, %0" : "+v"(i32[c]));
                if (OP == 23) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(i32[c]));
            }
        }
    }
    double s = 0; for (int c = 0; c < CHAINS; ++c) s += d[c] + f[c] + i32[c];
    if (s == 1.2345) out[0] = s;
}

template <int OP> void run(const char *name, double *d, int blocks_threads) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(blocks_threads), 0, 0, iters, d);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(blocks_threads), 0, 0, iters, d);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * UNROLL * CHAINS * (blocks_threads / 256);
    printf("%-16s waves/SIMD=%d  %.3f ms  %.2f ns/instr/SIMD\n", name, blocks_threads / 256, ms, ms * 1e6 / n);
}
#define R(op, name) run<op>(name, d, 256); run<op>(name, d, 512);
int main() {
    double *d; hipMalloc(&d, 64);
    R(0, "v_fma_f32") R(1, "v_fma_f64") R(2, "v_add_f64") R(3, "v_mul_f64") R(4, "v_cvt_f64_i32") R(5, "v_cvt_i32_f64")
    R(6, "v_rndne_f64") R(7, "v_ldexp_f64") R(8, "v_cvt_f32_f64") R(9, "v_cvt_f64_f32") R(10, "v_perm_b32") R(11, "v_add_u32")
    R(12, "v_lshl_add_u32") R(13, "v_cndmask_b32") R(14, "v_cvt_f32_i32") R(15, "v_exp_f32") R(16, "v_mad_i64_i32")
    R(17, "v_pk_fma_f32") R(18, "v_cvt_i32_f32") R(19, "v_rndne_f32") R(20, "v_ldexp_f32") R(21, "v_lshlrev_b64") R(22, "v_xor_b32") R(23, "v_bfe_u32")
    return 0;
}
