// Store throughput of one CU in the GEMM epilogue's pattern: a 512-thread workgroup (one per CU: 160 KB of LDS requested) writes 256 x 256 float
// tiles, every store instruction = 2 rows x 512 contiguous bytes (or 4 x 256), row pitch 1.2 MB; with all CUs storing at once and with one in
// eight.  hipcc --offload-arch=gfx950 -O3 tools/probe/store_rate.hip -o /tmp/store_rate && /tmp/store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int ROWS_PER_INSTR>
__global__ __launch_bounds__(512) void store_tiles(float *out, size_t pitch, int tiles_per_wg, int ntile_cols, int active_mod) {
    extern __shared__ char smem[];
    if (blockIdx.x % active_mod) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wo = w & 3, wb = w >> 2;
    constexpr int LPR = 64 / ROWS_PER_INSTR;                     // lanes per row
    const int rl = lane % LPR, rr = lane / LPR;
    f32x4 v = {(float)lane, 1.f, 2.f, 3.f};
    for (int t = 0; t < tiles_per_wg; ++t) {
        const int tile = blockIdx.x * tiles_per_wg + t;
        const int mt = tile / ntile_cols, nt = tile % ntile_cols;
        float *base = out + (size_t)(mt * 256 + wo * 64) * pitch + nt * 256 + wb * 128;
        // the wave's 64 rows x 128 columns
#pragma unroll 4
        for (int i = 0; i < 64 * 128 / 256; ++i) {               // 256 floats per instruction
            const int e = i * 256 + rr * (LPR * 4) + rl * 4;      // element index inside an instruction group
            const int row = (i * ROWS_PER_INSTR + rr) % 64, col = ((i * ROWS_PER_INSTR) / 64) * (LPR * 4) + rl * 4;
            (void)e;
            *reinterpret_cast<f32x4 *>(base + (size_t)row * pitch + col) = v;
        }
        __syncthreads();
    }
}
int main() {
    const size_t pitch = 307200;                                 // floats: d = 5, B = 61440
    const int mts = 10, ntc = 1200;
    float *out; hipMalloc(&out, (size_t)mts * 256 * pitch * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void *)store_tiles<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    hipFuncSetAttribute((const void *)store_tiles<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    for (int rows = 2; rows <= 4; rows += 2)
        for (int mod : {1, 2, 8, 32}) {
            const int tpw = 40;
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (rows == 2) hipLaunchKernelGGL(store_tiles<2>, dim3(256), dim3(512), 163840, 0, out, pitch, tpw, ntc, mod);
                else hipLaunchKernelGGL(store_tiles<4>, dim3(256), dim3(512), 163840, 0, out, pitch, tpw, ntc, mod);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            const double bytes_per_wg = (double)tpw * 256 * 256 * 4;
            printf("rows/instr %d, 1 of %2d CUs storing: %.3f ms, %.1f GB/s per storing CU, %.2f TB/s total, %.2f us per 256 KB tile\n", rows, mod, ms,
                   bytes_per_wg / ms / 1e6, bytes_per_wg * (256 / mod) / ms / 1e9, ms * 1e3 / tpw);
        }
    return 0;
}
