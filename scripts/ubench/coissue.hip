// Microbenchmark: can VALU work of one wave overlap i8 MFMA work of another wave on the same SIMD?
// 512-thread blocks: waves 0-3 run an MFMA loop, waves 4-7 run a VALU loop (f64 fma / f32 fma / int).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int MODE> // bit0: mfma waves active, bit1: valu waves active; VK: 0 f64, 1 f32, 2 int
__global__ __launch_bounds__(512) void k(int iters, int vk, double *out) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        if (!(MODE & 1)) return;
        v16i a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        v4i x = {(int)threadIdx.x, 1, 2, 3}, y = {3, 2, 1, (int)threadIdx.x};
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, a3, 0, 0, 0);
        }
        if (a0[0] + a1[1] + a2[2] + a3[3] == 12345) out[0] = 1;
    } else {
        if (!(MODE & 2)) return;
        if (vk == 0) {
            double p = threadIdx.x * 1e-3, q = 1.0000001, r0 = 0, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, r6 = 6, r7 = 7;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    r0 = fma(r0, q, p); r1 = fma(r1, q, p); r2 = fma(r2, q, p); r3 = fma(r3, q, p);
                    r4 = fma(r4, q, p); r5 = fma(r5, q, p); r6 = fma(r6, q, p); r7 = fma(r7, q, p);
                }
            }
            if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 1.2345) out[1] = 1;
        } else if (vk == 1) {
            float p = threadIdx.x * 1e-3f, q = 1.0000001f, r0 = 0, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, r6 = 6, r7 = 7;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    r0 = fmaf(r0, q, p); r1 = fmaf(r1, q, p); r2 = fmaf(r2, q, p); r3 = fmaf(r3, q, p);
                    r4 = fmaf(r4, q, p); r5 = fmaf(r5, q, p); r6 = fmaf(r6, q, p); r7 = fmaf(r7, q, p);
                }
            }
            if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 1.2345f) out[1] = 1;
        } else {
            unsigned p = threadIdx.x, r0 = 0, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, r6 = 6, r7 = 7;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    r0 = r0 * 3 + p; r1 = r1 * 3 + p; r2 = r2 * 3 + p; r3 = r3 * 3 + p;
                    r4 = r4 * 3 + p; r5 = r5 * 3 + p; r6 = r6 * 3 + p; r7 = r7 * 3 + p;
                }
            }
            if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345u) out[1] = 1;
        }
    }
}

template <int MODE> float run(int iters, int vk, double *d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, iters, vk, d);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, iters, vk, d);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    double *d; hipMalloc(&d, 64);
    const int iters = 20000; // 80000 MFMAs (32 cyc each) / 640000 VALU ops per wave
    const char *names[3] = {"f64 fma", "f32 fma", "u32 mad"};
    printf("mfma only: %.3f ms\n", run<1>(iters, 0, d));
    for (int vk = 0; vk < 3; ++vk) {
        float v = run<2>(iters, vk, d), b = run<3>(iters, vk, d);
        printf("%s: valu only %.3f ms, both %.3f ms\n", names[vk], v, b);
    }
    return 0;
}
