// Which int8 MFMA shape sustains more on a loaded MI355X: v_mfma_i32_32x32x32_i8 or v_mfma_i32_16x16x64_i8?
// MI355X_MICROARCH.md ("DVFS give-back", item 7) reports the 16x16 bf16 shape at ~1.12-1.15x the FLOP/s of the 32x32 one
// on random data at equal cycles per FLOP (the clock the chip holds differs).  This measures the int8 pair in the regime of
// the forward kernel: operands from registers, the A operand 0/1 bytes (the expanded sample bits) or random bytes, the B
// operand random digits, every CU busy with 2 waves per SIMD, wall time over long launches.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_i8_shapes.hip -o mfma_i8_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void k(const v4i *__restrict__ a_in, const v4i *__restrict__ b_in, int *__restrict__ out, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    v4i a[2], b[5];
    for (int i = 0; i < 2; ++i) a[i] = a_in[(tid * 2 + i) & 65535];
    for (int i = 0; i < 5; ++i) b[i] = b_in[(tid * 5 + i) & 65535];
    int sum = 0;
    if (SHAPE == 32) {
        v16i acc[2][5];
        for (int i = 0; i < 2; ++i)
            for (int l = 0; l < 5; ++l)
                for (int e = 0; e < 16; ++e) acc[i][l][e] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int l = 0; l < 5; ++l) acc[i][l] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[l], acc[i][l], 0, 0, 0);
        }
        for (int i = 0; i < 2; ++i)
            for (int l = 0; l < 5; ++l)
                for (int e = 0; e < 16; ++e) sum += acc[i][l][e];
    } else {
        v4i acc[4][2][5]; // the same 160 accumulator registers: 4 sample tiles x 2 node tiles x 5 limbs of 16 x 16
        for (int i = 0; i < 4; ++i)
            for (int n = 0; n < 2; ++n)
                for (int l = 0; l < 5; ++l)
                    for (int e = 0; e < 4; ++e) acc[i][n][l][e] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int l = 0; l < 5; ++l)
                        acc[i][n][l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i & 1], b[l], acc[i][n][l], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i)
            for (int n = 0; n < 2; ++n)
                for (int l = 0; l < 5; ++l)
                    for (int e = 0; e < 4; ++e) sum += acc[i][n][l][e];
    }
    out[tid] = sum;
}

template <int SHAPE>
double run(const v4i *a, const v4i *b, int *out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * 2 * 8; // 2 workgroups per CU resident, 8 rounds
    hipLaunchKernelGGL(k<SHAPE>, dim3(grid), dim3(256), 0, 0, a, b, out, iters / 8);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<SHAPE>, dim3(grid), dim3(256), 0, 0, a, b, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // MACs per wave-iteration: 10 x (32 x 32 x 32) = 327 680, or 40 x (16 x 16 x 64) = 655 360
    const double macs = 3.0 * grid * 4.0 * iters * (SHAPE == 32 ? 327680.0 : 655360.0);
    return 2.0 * macs / (ms * 1e-3) / 1e12; // TOP/s
}

int main() {
    std::vector<int> h01(65536 * 4), hr(65536 * 4);
    uint32_t s = 12345;
    for (auto &v : h01) {
        s = s * 1664525u + 1013904223u;
        v = (int)((s >> 3) & 0x01010101u);
    }
    for (auto &v : hr) {
        s = s * 1664525u + 1013904223u;
        v = (int)s;
    }
    std::vector<int> hz(65536 * 4, 0);
    v4i *a01, *ar, *az;
    int *out;
    hipMalloc(&a01, h01.size() * 4);
    hipMalloc(&ar, hr.size() * 4);
    hipMalloc(&az, hz.size() * 4);
    hipMalloc(&out, sizeof(int) * 256 * 2 * 8 * 256);
    hipMemcpy(a01, h01.data(), h01.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(ar, hr.data(), hr.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(az, hz.data(), hz.size() * 4, hipMemcpyHostToDevice);
    const int iters = 4000;
    for (int rep = 0; rep < 2; ++rep) {
        printf("A = 0/1 bytes, B random : 32x32x32 %.0f TOP/s   16x16x64 %.0f TOP/s\n", run<32>(a01, ar, out, iters), run<16>(a01, ar, out, iters / 2));
        printf("A random,     B random : 32x32x32 %.0f TOP/s   16x16x64 %.0f TOP/s\n", run<32>(ar, ar, out, iters), run<16>(ar, ar, out, iters / 2));
        printf("A zero,       B zero   : 32x32x32 %.0f TOP/s   16x16x64 %.0f TOP/s\n", run<32>(az, az, out, iters), run<16>(az, az, out, iters / 2));
    }
    return 0;
}
