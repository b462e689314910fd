// Would the FP6 block-scaled MFMA carry the forward GEMM of precision i8w faster than the int8 one?  
//
// v_mfma_scale_f32_32x32x64_f8f6f4 with an FP4 A operand (the 0/1 sample bits: 1.0 = 0b0010) and an FP6-E2M3 B operand (digits
// -15..15 in units of 1/8: code = |d| | sign << 5, block scale 2^3) multiplies 64 columns per instruction at the FP4/FP6 rate and is
// EXACT in its FP32 accumulators for these operands (|sum| <= 15 * 1024 < 2^24 over the 1024 columns of the headline problem).  54 bits of
// Theta are 11 such digit planes (4.95 bits each) against 7 int8 planes.  This program (1) checks the exactness claim on the device
// against integer arithmetic and (2) measures what both instructions SUSTAIN, every CU busy with 2 waves per SIMD, operands from
// registers, on the kernels' kind of data (0/1 A, random digits B) -- the int8 instruction holds 3.7 of its nominal 5 POP/s there
// (scripts/ubench/mfma_i8_shapes.hip: the chip lowers its clock under it).
// Build: hipcc -O3 --offload-arch=gfx950 mfma_fp6_rate.hip -o mfma_fp6_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define MFMA_FP6(a, b, c) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4((a), (b), (c), 4 /* A: fp4 */, 2 /* B: fp6 e2m3 */, 0, 127, 0, 130)

// ---- exactness: every lane holds the same B registers, so every output column is the sum of the lane's 32 digits over both K halves
__global__ void k_check(const v8i *a, const v8i *b, float *o) {
    v16f c;
    for (int e = 0; e < 16; ++e) c[e] = 0.0f;
    for (int it = 0; it < 16; ++it) c = MFMA_FP6(a[0], b[it], c); // 16 x 64 = 1024 columns
    for (int e = 0; e < 16; ++e) o[threadIdx.x * 16 + e] = c[e];
}

template <int KIND> // 0: int8 32x32x32, 2 sample tiles x 4 planes (sweep A of k_fwd_i8w); 1: fp4 x fp6 32x32x64, 2 sample tiles x 6 planes
__global__ __launch_bounds__(256, 2) void k_rate(const int *__restrict__ a_in, const int *__restrict__ b_in, float *__restrict__ out, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    float sum = 0.0f;
    if (KIND == 0) {
        v4i a[2], b[4];
        for (int i = 0; i < 2; ++i)
            for (int e = 0; e < 4; ++e) a[i][e] = a_in[(tid * 8 + i * 4 + e) & 0xfffff];
        for (int l = 0; l < 4; ++l)
            for (int e = 0; e < 4; ++e) b[l][e] = b_in[(tid * 16 + l * 4 + e) & 0xfffff];
        v16i acc[2][4];
        for (int i = 0; i < 2; ++i)
            for (int l = 0; l < 4; ++l)
                for (int e = 0; e < 16; ++e) acc[i][l][e] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int l = 0; l < 4; ++l) acc[i][l] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[l], acc[i][l], 0, 0, 0);
        }
        for (int i = 0; i < 2; ++i)
            for (int l = 0; l < 4; ++l)
                for (int e = 0; e < 16; ++e) sum += (float)acc[i][l][e];
    } else {
        v8i a[2], b[6];
        for (int i = 0; i < 2; ++i)
            for (int e = 0; e < 8; ++e) a[i][e] = e < 4 ? a_in[(tid * 8 + i * 4 + e) & 0xfffff] : 0;
        for (int l = 0; l < 6; ++l)
            for (int e = 0; e < 8; ++e) b[l][e] = e < 6 ? b_in[(tid * 36 + l * 6 + e) & 0xfffff] : 0;
        v16f acc[2][6];
        for (int i = 0; i < 2; ++i)
            for (int l = 0; l < 6; ++l)
                for (int e = 0; e < 16; ++e) acc[i][l][e] = 0.0f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int l = 0; l < 6; ++l) acc[i][l] = MFMA_FP6(a[i], b[l], acc[i][l]);
        }
        for (int i = 0; i < 2; ++i)
            for (int l = 0; l < 6; ++l)
                for (int e = 0; e < 16; ++e) sum += acc[i][l][e];
    }
    out[tid] = sum;
}

template <int KIND>
double run(const int *a, const int *b, float *out, int iters, double *ms_out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * 2 * 8;
    hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(256), 0, 0, a, b, out, iters / 8);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(256), 0, 0, a, b, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    *ms_out = ms / 3;
    const double macs_per_iter = KIND == 0 ? 8.0 * 32 * 32 * 32 : 12.0 * 32 * 32 * 64;
    return 2.0 * 3.0 * grid * 4.0 * iters * macs_per_iter / (ms * 1e-3) / 1e12; // TOP/s
}

int main() {
    // ---- (1) exactness
    {
        std::vector<int> ha(8, 0), hb(16 * 8, 0);
        for (int e = 0; e < 4; ++e) ha[e] = 0x22222222; // 8 x fp4 1.0 per dword: all 32 k of the lane
        uint32_t s = 2463534242u;
        long long want = 0;
        for (int it = 0; it < 16; ++it) {
            unsigned char code[32];
            for (int q = 0; q < 32; ++q) {
                s ^= s << 13; s ^= s >> 17; s ^= s << 5;
                const int d = (int)(s % 31) - 15;
                code[q] = (unsigned char)((d < 0 ? -d : d) | (d < 0 ? 32 : 0));
                want += 2 * d; // both lane halves (k 0..31 and 32..63) hold the same registers
            }
            unsigned long long bits[3] = {0, 0, 0}; // 32 x 6 bits = 192 bits, little-endian packing
            for (int q = 0; q < 32; ++q)
                for (int bb = 0; bb < 6; ++bb)
                    if ((code[q] >> bb) & 1) bits[(q * 6 + bb) >> 6] |= 1ull << ((q * 6 + bb) & 63);
            for (int e = 0; e < 6; ++e) hb[it * 8 + e] = (int)(bits[e >> 1] >> (32 * (e & 1)));
        }
        int *da, *db;
        float *dout;
        hipMalloc(&da, 32);
        hipMalloc(&db, 16 * 32);
        hipMalloc(&dout, 64 * 16 * 4);
        hipMemcpy(da, ha.data(), 32, hipMemcpyHostToDevice);
        hipMemcpy(db, hb.data(), 16 * 32, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, (const v8i *)da, (const v8i *)db, dout);
        std::vector<float> ho(64 * 16);
        hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (float v : ho) bad += v != (float)want;
        printf("exactness: sum of 1024 products of 0/1 (fp4) x digits -15..15 (fp6 e2m3, block scale 2^3): expected %lld, got %g in %d of 1024 outputs%s\n",
               want, ho[0], 1024 - bad, bad ? "  ** MISMATCH **" : "  (exact)");
    }
    // ---- (2) sustained rates
    std::vector<int> h01(1 << 20), hr8(1 << 20), h4(1 << 20), h6(1 << 20);
    uint32_t s = 12345;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; };
    for (auto &v : h01) v = (int)(rnd() & 0x01010101u);              // int8 A: 0/1 bytes
    for (auto &v : hr8) v = (int)rnd();                               // int8 B: random digits
    for (auto &v : h4) v = (int)((rnd() & 0x11111111u) << 1);        // fp4 A: nibbles 0b0000 / 0b0010 (0 / 1.0)
    for (auto &v : h6) v = (int)rnd();                                // fp6 B: random codes (all 64 are valid e2m3 values)
    int *d01, *dr8, *d4, *d6;
    float *dout;
    hipMalloc(&d01, 4 << 20); hipMalloc(&dr8, 4 << 20); hipMalloc(&d4, 4 << 20); hipMalloc(&d6, 4 << 20);
    hipMalloc(&dout, 256 * 2 * 8 * 256 * 4);
    hipMemcpy(d01, h01.data(), 4 << 20, hipMemcpyHostToDevice);
    hipMemcpy(dr8, hr8.data(), 4 << 20, hipMemcpyHostToDevice);
    hipMemcpy(d4, h4.data(), 4 << 20, hipMemcpyHostToDevice);
    hipMemcpy(d6, h6.data(), 4 << 20, hipMemcpyHostToDevice);
    const int iters = 20000;
    for (int rep = 0; rep < 3; ++rep) {
        double ms8, ms6;
        const double t8 = run<0>(d01, dr8, dout, iters, &ms8), t6 = run<1>(d4, d6, dout, iters, &ms6);
        // 54 bits of Theta against one 64-column step of one 32-sample tile: 7 planes x 2 int8 MFMAs (K = 32) or 11 planes x 1 fp6 MFMA (K = 64)
        const double c8 = ms8 / (iters * 8.0), c6 = ms6 / (iters * 12.0); // ms per MFMA per wave slot (relative units)
        printf("int8 32x32x32: %.2f POP/s sustained   fp4 x fp6 32x32x64: %.2f POP/s sustained   ratio %.2f   54-bit step: int8 14 MFMAs = %.3f, fp6 11 MFMAs = %.3f -> %.0f %% of the int8 matrix time\n",
               t8 / 1e3, t6 / 1e3, t6 / t8, 14 * c8 * 1e6, 11 * c6 * 1e6, 100.0 * (11 * c6) / (14 * c8));
    }
    return 0;
}
