// Is v_mfma_f32_32x32x2_f32 (and 16x16x4) a sequential float32 FMA chain over k?  D = A B + C with random operands, K = 32 as 16 chained
// instructions, compared bit for bit with fmaf chains in three candidate orders.   hipcc --offload-arch=gfx950 -O2 mfma_f32_order.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// 32x32x2: lane l holds A[row l%32][k l/32], B[k l/32][col l%32]; acc r of lane l: row 8(r/4) + 4(l/32) + r%4, col l%32
__global__ void k32(const float *A, const float *B, float *D, int K) {   // A [32][K], B [K][32], D [32][32]
    const int l = threadIdx.x;
    f32x16 acc = {0};
    for (int k = 0; k < K; k += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l % 32) * K + k + l / 32], B[(k + l / 32) * 32 + l % 32], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[(8 * (r / 4) + 4 * (l / 32) + r % 4) * 32 + l % 32] = acc[r];
}
// 16x16x4: lane l holds A[row l%16][k l/16], B[k l/16][col l%16]; acc r: row 4(l/16) + r, col l%16
__global__ void k16(const float *A, const float *B, float *D, int K) {   // A [16][K], B [K][16]
    const int l = threadIdx.x;
    f32x4 acc = {0};
    for (int k = 0; k < K; k += 4)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l % 16) * K + k + l / 16], B[(k + l / 16) * 16 + l % 16], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l / 16) + r) * 16 + l % 16] = acc[r];
}
int main() {
    const int K = 32;
    float hA[32 * K], hB[K * 32], hD[1024], hD16[256];
    srand(7);
    for (int i = 0; i < 32 * K; ++i) hA[i] = (float)rand() / RAND_MAX * 2 - 1;
    for (int i = 0; i < K * 32; ++i) hB[i] = ((float)rand() / RAND_MAX * 2 - 1) * (i % 7 == 0 ? 1e3f : 1.f);
    float *dA, *dB, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    k32<<<1, 64>>>(dA, dB, dD, K); hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int eq_seq = 0, eq_pair = 0, eq_f64 = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        float s = 0.f; for (int k = 0; k < K; ++k) s = fmaf(hA[i * K + k], hB[k * 32 + j], s);                                 // sequential fma chain
        float p = 0.f; for (int k = 0; k < K; k += 2) p = (float)((double)hA[i * K + k] * hB[k * 32 + j] + (double)hA[i * K + k + 1] * hB[(k + 1) * 32 + j] + (double)p);   // exact pair + acc, one rounding
        double d = 0; for (int k = 0; k < K; ++k) d += (double)hA[i * K + k] * hB[k * 32 + j];
        eq_seq += memcmp(&s, &hD[i * 32 + j], 4) == 0; eq_pair += memcmp(&p, &hD[i * 32 + j], 4) == 0; float df = (float)d; eq_f64 += memcmp(&df, &hD[i * 32 + j], 4) == 0;
    }
    printf("32x32x2 f32, K = 32: bitwise equal to sequential fmaf chain %d / 1024, to (pair exact + acc, one rounding) %d / 1024, to float(f64 sum) %d / 1024\n", eq_seq, eq_pair, eq_f64);
    // 16x16x4 on the first 16 rows / columns
    float hB16[K * 16]; for (int k = 0; k < K; ++k) for (int j = 0; j < 16; ++j) hB16[k * 16 + j] = hB[k * 32 + j];
    hipMemcpy(dB, hB16, sizeof hB16, hipMemcpyHostToDevice);
    k16<<<1, 64>>>(dA, dB, dD, K); hipMemcpy(hD16, dD, sizeof hD16, hipMemcpyDeviceToHost);
    int e1 = 0, e4 = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        float s = 0.f; for (int k = 0; k < K; ++k) s = fmaf(hA[i * K + k], hB16[k * 16 + j], s);
        float p = 0.f; for (int k = 0; k < K; k += 4) { double t = (double)p; for (int q = 0; q < 4; ++q) t += (double)hA[i * K + k + q] * hB16[(k + q) * 16 + j]; p = (float)t; }
        e1 += memcmp(&s, &hD16[i * 16 + j], 4) == 0; e4 += memcmp(&p, &hD16[i * 16 + j], 4) == 0;
    }
    printf("16x16x4 f32, K = 32: bitwise equal to sequential fmaf chain %d / 256, to (four exact + acc, one rounding) %d / 256\n", e1, e4);
    return 0;
}
