// Microbenchmark: achievable HBM rates of simple streaming kernels on this part (what bounds ft_nonlin and the other streaming kernels).
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/hbm_rate.hip -o tools/hbm_rate.so
#include <hip/hip_runtime.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: copy, 1: read-only (sum), 2: write-only, 3: copy with nontemporal hints
__global__ __launch_bounds__(256) void stream_kernel(const f4 *__restrict__ src, f4 *__restrict__ dst, size_t n4, float *sink) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i < n4; i += stride) {
        f4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t j = i + (size_t)k * 256;
            if (MODE == 2) v[k] = f4{1.f, 2.f, 3.f, (float)j};
            else if (j < n4) v[k] = MODE == 3 ? __builtin_nontemporal_load(src + j) : src[j];
            else v[k] = f4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t j = i + (size_t)k * 256;
            if (MODE == 1) acc += v[k];
            else if (j < n4) { if (MODE == 3) __builtin_nontemporal_store(v[k], dst + j); else dst[j] = v[k]; }
        }
    }
    if (MODE == 1 && acc[0] + acc[1] + acc[2] + acc[3] == 1.2345e30f) *sink = acc[0];
}

extern "C" double hbm_rate_run(int mode, int blocks, const void *src, void *dst, size_t bytes, int reps) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t n4 = bytes / 16;
    float *sink = reinterpret_cast<float *>(dst);
    auto launch = [&] {
        switch (mode) {
            case 0: hipLaunchKernelGGL(stream_kernel<0>, dim3(blocks), dim3(256), 0, 0, (const f4 *)src, (f4 *)dst, n4, sink); break;
            case 1: hipLaunchKernelGGL(stream_kernel<1>, dim3(blocks), dim3(256), 0, 0, (const f4 *)src, (f4 *)dst, n4, sink); break;
            case 2: hipLaunchKernelGGL(stream_kernel<2>, dim3(blocks), dim3(256), 0, 0, (const f4 *)src, (f4 *)dst, n4, sink); break;
            default: hipLaunchKernelGGL(stream_kernel<3>, dim3(blocks), dim3(256), 0, 0, (const f4 *)src, (f4 *)dst, n4, sink); break;
        }
    };
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) launch();
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
