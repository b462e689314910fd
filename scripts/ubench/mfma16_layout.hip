// Operand / result layout of v_mfma_i32_16x16x64_i8 on gfx950, found by experiment: A[m][k] = (m + 1) at k = kk, B[k][n] = (n + 1) at
// k = kk, so C[m][n] = (m + 1)(n + 1) iff the lane that holds A's row m byte kk and the lane that holds B's column n byte kk are
// paired by the hardware.  Prints, per lane, what (m, n) its 4 result registers hold, assuming A: lane = (m = l & 15, g = l >> 4),
// bytes 16 g + 4 v + b; B the same with n.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(int *out, int kk) {
    const int l = threadIdx.x, m = l & 15, g = l >> 4;
    v4i a = {0, 0, 0, 0}, b = {0, 0, 0, 0}, c = {0, 0, 0, 0};
    if (kk / 16 == g) {
        const int v = (kk % 16) / 4, sh = 8 * (kk % 4);
        a[v] = (m + 1) << sh;
        b[v] = (m + 1) << sh;
    }
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = c[j];
}
int main() {
    int *d, h[256];
    hipMalloc(&d, sizeof h);
    for (int kk : {0, 5, 17, 40, 63}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, kk);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        bool ok = true;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
                const int n = l & 15, mm = 4 * (l >> 4) + j; // expected: lane = column n, register j = row 4 (l >> 4) + j
                if (h[l * 4 + j] != (mm + 1) * (n + 1)) ok = false;
            }
        printf("kk %2d: C layout lane(n = l & 15, rows 4 (l >> 4) + j) %s; lane 17 regs: %d %d %d %d (expect %d %d %d %d)\n", kk, ok ? "OK" : "MISMATCH",
               h[17 * 4], h[17 * 4 + 1], h[17 * 4 + 2], h[17 * 4 + 3], 5 * 2, 6 * 2, 7 * 2, 8 * 2);
    }
    return 0;
}
