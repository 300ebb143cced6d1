// Sustained bf16 MFMA rate of the whole chip from registers only (no memory traffic), with random or zero
// operands, 1 or 2 wavefronts per SIMD: the practical ceiling for the f32x3 kernels (power / clock limited).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(const u32x4* src, float* out, long long* clk, int iters) {
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, src[(threadIdx.x * 8 + i) % 4096]);
        b[i] = __builtin_bit_cast(bf16x8, src[(threadIdx.x * 8 + 4 + i) % 4096]);
    }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

int main() {
    const int nb = 256 * 2;
    u32x4* src; float* out; long long* clk;
    hipMalloc(&src, 4096 * 16); hipMalloc(&out, nb * 256 * 4); hipMalloc(&clk, nb * 16);
    std::vector<unsigned> h(4096 * 4);
    for (int mode = 0; mode < 2; ++mode) {
        for (auto& v : h) {
            // random bf16 pairs in [-1, 1): sign + exponent 0x3f/0x3e + random mantissa
            unsigned lo = 0x3f00 | (rand() & 0x80ff), hi = 0x3f00 | (rand() & 0x80ff);
            v = mode ? ((hi << 16) | lo) : 0u;
        }
        hipMemcpy(src, h.data(), 4096 * 16, hipMemcpyHostToDevice);
        for (int blocks : {256, 512}) {
            const int iters = 20000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            k<12><<<blocks, 256>>>(src, out, clk, 100);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            k<12><<<blocks, 256>>>(src, out, clk, iters);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
            const double flops = (double)blocks * 4 * iters * 12 * 2.0 * 32 * 32 * 16;
            printf("%s operands, %d blocks (%d wave/SIMD): %.1f TF bf16, %.2f ms, shader clock %.0f MHz, %.1f cycles/MFMA\n",
                   mode ? "random" : "zero", blocks, blocks / 256, flops / ms / 1e9, ms,
                   (double)c[0] / ((double)c[1] / 100.0), (double)c[0] / (iters * 12.0));
        }
    }
    return 0;
}
