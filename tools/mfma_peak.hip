// Microbenchmark: what the chip sustains on back-to-back v_mfma_f32_32x32x16_f16 from registers (no memory traffic), for 1..4 waves
// per SIMD and a duty-cycle knob (s_sleep between bursts) -- the reference point for the irrep GEMM's MFMA fraction.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/mfma_peak.hip -o tools/mfma_peak.so
#include <hip/hip_runtime.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// RANDOM = 1: eight operand sets of pseudo-random fp16 values (every MFMA sees different multiplier inputs, as in a real GEMM);
// RANDOM = 0: one smooth operand pair for every MFMA (minimal switching activity); RANDOM = 2: all-zero operands (no switching at all)
template <int NACC, int RANDOM>
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters, int sleep) {
    f16x8 av[8], bv[8];
    unsigned seed = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int q = 0; q < 8; ++q)
        for (int e = 0; e < 8; ++e) {
            if (RANDOM == 1) {
                seed = seed * 1664525u + 1013904223u; av[q][e] = (_Float16)(((int)(seed >> 8) % 2001 - 1000) * 0.001f);
                seed = seed * 1664525u + 1013904223u; bv[q][e] = (_Float16)(((int)(seed >> 8) % 2001 - 1000) * 0.001f);
            } else if (RANDOM == 2) { av[q][e] = (_Float16)0.f; bv[q][e] = (_Float16)0.f; }
            else { av[q][e] = (_Float16)(0.001f * (threadIdx.x + e)); bv[q][e] = (_Float16)(0.002f * (threadIdx.x - e)); }
        }
    f32x16 acc[NACC];
    for (int q = 0; q < NACC; ++q)
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();          // s_memtime: shader-clock cycles
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
            for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[rep], bv[(rep + q) & 7], acc[q], 0, 0, 0);
        if (sleep) __builtin_amdgcn_s_sleep(8);
    }
    float s = 0.f;
    for (int q = 0; q < NACC; ++q)
        for (int r = 0; r < 16; ++r) s += acc[q][r];
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x < 256) out[(1 << 21) + blockIdx.x] = (float)(t1 - t0);      // cycles of this wave's loop
}

extern "C" double mfma_peak_run(int blocks, int threads, int iters, int sleep, int random, float *out_dev) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto kern = random == 1 ? mfma_loop<4, 1> : random == 2 ? mfma_loop<4, 2> : mfma_loop<4, 0>;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out_dev, iters / 10, sleep);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out_dev, iters, sleep);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * (threads / 64) * iters * 8.0 * 4.0 * 32768.0;
    return flops / (ms * 1e-3) / 1e12;      // TFLOP/s
}

// The same loop with v_mfma_f32_16x16x32_f16: same flops per instruction, a quarter of the accumulator registers per instruction and twice the
// K -- half the accumulator read / write traffic per flop.  Does the part sustain more of it on real data?
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int RANDOM>
__global__ __launch_bounds__(256) void mfma_loop16(float *out, int iters, int sleep) {
    f16x8 av[8], bv[8];
    unsigned seed = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int q = 0; q < 8; ++q)
        for (int e = 0; e < 8; ++e) {
            if (RANDOM == 1) {
                seed = seed * 1664525u + 1013904223u; av[q][e] = (_Float16)(((int)(seed >> 8) % 2001 - 1000) * 0.001f);
                seed = seed * 1664525u + 1013904223u; bv[q][e] = (_Float16)(((int)(seed >> 8) % 2001 - 1000) * 0.001f);
            } else if (RANDOM == 2) { av[q][e] = (_Float16)0.f; bv[q][e] = (_Float16)0.f; }
            else { av[q][e] = (_Float16)(0.001f * (threadIdx.x + e)); bv[q][e] = (_Float16)(0.002f * (threadIdx.x - e)); }
        }
    f32x4 acc[NACC];
    for (int q = 0; q < NACC; ++q)
        for (int r = 0; r < 4; ++r) acc[q][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
            for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[rep], bv[(rep + q) & 7], acc[q], 0, 0, 0);
        if (sleep) __builtin_amdgcn_s_sleep(8);
    }
    float s = 0.f;
    for (int q = 0; q < NACC; ++q)
        for (int r = 0; r < 4; ++r) s += acc[q][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

extern "C" double mfma_peak_run16(int blocks, int threads, int iters, int sleep, int random, float *out_dev) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto kern = random == 1 ? mfma_loop16<8, 1> : random == 2 ? mfma_loop16<8, 2> : mfma_loop16<8, 0>;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out_dev, iters / 10, sleep);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out_dev, iters, sleep);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * (threads / 64) * iters * 8.0 * 8.0 * 16384.0;
    return flops / (ms * 1e-3) / 1e12;      // TFLOP/s
}
