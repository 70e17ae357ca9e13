// Semantics probe for gfx950's ds_read_b64_tr_b16: which LDS elements does lane l receive when every lane supplies its own address?
// LDS is filled with lds[e] = e (16-bit); case 0: lane l supplies byte address 8 * l (lane-linear); case 1: address 8 * (l & 15) + 128 * (l >> 4)
// (the same thing); case 2: a [4 rows][16] block with a row stride of 64 bytes: lane p of a 16-lane group supplies row (p / 4), chunk (p % 4).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(int mode, uint16_t *out) {
    __shared__ uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x, p = l & 15, g = l >> 4;
    unsigned addr;
    if (mode == 0) addr = 8u * l;
    else if (mode == 1) addr = 8u * p + 128u * g;
    else addr = 1024u * g + 64u * (p / 4) + 8u * (p % 4);
    addr += (unsigned)(uintptr_t)lds;          // LDS base offset of the array (0 here, kept for generality)
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)(v >> (16 * j));
}
int main() {
    uint16_t *d; hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, mode, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) { printf("  lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %4d", h[l * 4 + j]); printf("%s", (l % 4 == 3) ? "\n" : "   |"); }
    }
    return 0;
}
