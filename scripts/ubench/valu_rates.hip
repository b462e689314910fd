// Issue cost of the vector instructions of the forward epilogue on gfx950: cycles per wave-instruction with one wave
// or two waves per SIMD, 8 independent chains per wave (throughput, not latency).  Build: hipcc --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP, int CH>
__global__ void k(double *out, long long *cyc, int iters) {
    double d[8];
    float f[8];
    int n[8];
    long long l[8];
    for (int c = 0; c < 8; ++c) {
        d[c] = 1.0 + 1e-9 * (threadIdx.x + c);
        f[c] = 1.0f + 1e-6f * (threadIdx.x + c);
        n[c] = threadIdx.x * 7 + c;
        l[c] = threadIdx.x + c;
    }
    const double k1 = 1.0000001, k2 = 1e-12;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#define BODY(c)                                                                                                              \
    if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[c]) : "v"(k1), "v"(k2));                                   \
    if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[c]) : "v"(k2));                                                \
    if (OP == 2) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[c]) : "v"(n[c]));                                              \
    if (OP == 3) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[c]) : "v"(d[c]));                                              \
    if (OP == 4) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[c]) : "v"(f[c]));                                              \
    if (OP == 5) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[c]) : "v"(n[c] & 1));                                        \
    if (OP == 6) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[c]) : "v"(1.0000001f));                                    \
    if (OP == 7) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(l[c]) : "v"(l[(c + 1) & 7]));                            \
    if (OP == 8) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(n[c]) : "v"(n[(c + 1) & 7]), "v"(0x05040100));              \
    if (OP == 9) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(n[c]) : "v"(n[(c + 1) & 7]), "v"(n[(c + 2) & 7]));          \
    if (OP == 10) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[c]) : "v"(n[(c + 1) & 7]));                                   \
    if (OP == 11) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(n[c]) : "v"(n[(c + 1) & 7]));                                \
    if (OP == 12) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(n[c]) : "v"(n[(c + 1) & 7]));                           \
    if (OP == 13) asm volatile("v_cvt_f64_i32 %0, %1\n v_cvt_f64_i32 %2, %3" : "=v"(d[c]), "=v"(d[(c + 4) & 7]) : "v"(n[c]), "v"(n[(c+1)&7])); \
    if (OP == 14) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[c]) : "v"(k1));                                               \
    if (OP == 15) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(n[c]) : "v"(n[(c + 1) & 7]));                          \
    if (OP == 16) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(l[c]) : "v"(n[c]), "v"(n[(c + 1) & 7]) : "vcc");
        if (CH == 8) { REP8(BODY) } else { BODY(0) }
    }
    const long long t1 = clock64();
    double acc = 0;
    for (int c = 0; c < 8; ++c) acc += d[c] + f[c] + n[c] + (double)l[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP, int CH>
void run_ch(const char *name) {
    const int iters = 2000;
    double *out;
    long long *cyc;
    hipMalloc(&out, sizeof(double) * 512 * 512);
    hipMalloc(&cyc, sizeof(long long) * 512);
    for (int wps = 1; wps <= 2; ++wps) {
        const int threads = 256 * wps; // 4 or 8 waves per CU (one block per CU)
        hipLaunchKernelGGL((k<OP, CH>), dim3(256), dim3(threads), 0, 0, out, cyc, 10);
        hipDeviceSynchronize();
        hipLaunchKernelGGL((k<OP, CH>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        std::vector<long long> h(256);
        hipMemcpy(h.data(), cyc, sizeof(long long) * 256, hipMemcpyDeviceToHost);
        double avg = 0;
        for (long long v : h) avg += (double)v;
        avg /= 256.0;
        const int per = (OP == 13 ? 2 : 1) * CH;
        // clock64 ticks at 100 MHz (s_memrealtime) on this target? print raw ticks per instruction too
        printf("%-16s %s waves/SIMD %d: %.2f ticks per wave-instruction (per SIMD: %.2f)\n", name, CH == 8 ? "8 chains " : "dependent", wps, avg / (iters * per), avg / (iters * per) / wps);
    }
    hipFree(out);
    hipFree(cyc);
}

template <int OP>
void run(const char *name) {
    run_ch<OP, 8>(name);
    run_ch<OP, 1>(name);
}

int main() {
    run<6>("v_fma_f32");
    run<10>("v_add_u32");
    run<0>("v_fma_f64");
    run<1>("v_add_f64");
    run<14>("v_mul_f64");
    run<2>("v_cvt_f64_i32");
    run<3>("v_cvt_f32_f64");
    run<4>("v_cvt_f64_f32");
    run<5>("v_ldexp_f64");
    run<7>("v_lshl_add_u64");
    run<8>("v_perm_b32");
    run<9>("v_max3_i32");
    run<11>("v_mul_lo_u32");
    run<12>("v_lshl_add_u32");
    run<15>("v_cndmask_b32");
    run<16>("v_mad_u64_u32");
    return 0;
}
