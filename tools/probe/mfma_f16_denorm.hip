// Does v_mfma_f32_32x32x16_f16 on gfx950 honour fp16 SUBNORMAL inputs (or flush them to zero)?  The fp16 x 2 operand split leaves the low
// piece of an element below 2^-3 in the subnormal range unless a block scale lifts it.
// Build + run: hipcc --offload-arch=gfx950 -O2 tools/probe/mfma_f16_denorm.hip -o /tmp/mfma_f16_denorm && /tmp/mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void probe(float *out, float aval, float bval) {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)0.f; b[e] = (_Float16)0.f; }
    a[0] = (_Float16)aval;             // every lane: A[row][k0] = aval, B[k0][col] = bval for k0 = 8 * (lane / 32)
    b[0] = (_Float16)bval;
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}

int main() {
    float *d; (void)hipMalloc(&d, 4);
    const float cases[][2] = {{1.0f, 1.0f}, {3.0e-5f, 1024.0f}, {1.0e-6f, 32768.0f}, {6.0e-8f, 32768.0f}, {3.0e-5f, 3.0e-5f}};
    for (auto &cs : cases) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, cs[0], cs[1]);
        float h; (void)hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        const float ah = (float)(_Float16)cs[0], bh = (float)(_Float16)cs[1];
        printf("a = %.6e (fp16 %.6e, %s)  b = %.6e: mfma sum over 2 k-groups = %.9e   expected %.9e\n", cs[0], ah, ah < 6.1e-5f ? "subnormal" : "normal", cs[1], h, 2.0f * ah * bh);
    }
    return 0;
}
