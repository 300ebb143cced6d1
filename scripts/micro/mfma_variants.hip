// Which f16 MFMA shape does the most work per joule?  The matrix kernels of the training step are power-limited (DESIGN.md
// section 5): their time is energy.  Register-only loops on RANDOM f16 operands, one wavefront per SIMD on every CU, run for
// ~50 ms each (the clock follows the power of the last milliseconds): sustained TFLOP/s and shader clock of
//   v_mfma_f32_32x32x16_f16 (what the kernels use: 16 accumulator registers per block, 32 cycles),
//   v_mfma_f32_16x16x32_f16 (4 accumulator registers per block, 16 cycles: twice the operand reads per flop),
// and of the 32x32x16 loop with the same A operand for three / five consecutive MFMAs (the kernels' visiting orders).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_variants.hip -o scripts/micro/bin/mfma_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: 32x32x16, 15 accumulators, A changes every 3 MFMAs (5 x 3 blocks, nt inner)
// MODE 1: 32x32x16, 15 accumulators, A changes every MFMA, B every 5 (mt inner)
// MODE 2: 16x16x32, 60 accumulators (same 240 registers), A changes every 6 MFMAs (10 x 6 blocks)
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const u32x4* src, float* out, long long* clk, int iters) {
    f16x8 a[10], b[6];
    for (int i = 0; i < 10; ++i) a[i] = __builtin_bit_cast(f16x8, src[(threadIdx.x * 16 + i) % 4096]);
    for (int i = 0; i < 6; ++i) b[i] = __builtin_bit_cast(f16x8, src[(threadIdx.x * 16 + 10 + i) % 4096]);
    f32x16 acc[15];
    f32x4 acc4[60];
    for (int i = 0; i < 15; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = 0; i < 60; ++i) for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 15; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i / 3], b[i % 3], acc[i], 0, 0, 0);
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 15; ++i)
                acc[(i % 5) * 3 + i / 5] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i % 5], b[i / 5], acc[(i % 5) * 3 + i / 5], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 60; ++i)
                acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i / 6], b[i % 6], acc4[i], 0, 0, 0);
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 15; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 60; ++i) for (int r = 0; r < 4; ++r) s += acc4[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int MODE>
static void run(const char* name, const u32x4* src, float* out, long long* clk, double flops_per_iter) {
    const int iters = MODE == 2 ? 40000 : 160000;    // ~ 50 ms
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<MODE><<<256, 256>>>(src, out, clk, iters);      // warm: the clock settles
    CK(hipEventRecord(e0));
    k<MODE><<<256, 256>>>(src, out, clk, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long c[2]; CK(hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost));
    printf("%-44s %7.1f TF f16, %6.2f ms, shader clock %4.0f MHz\n", name, 256.0 * 4 * iters * flops_per_iter / ms / 1e9, ms,
           (double)c[0] / ((double)c[1] / 100.0));
}

int main() {
    u32x4* src; float* out; long long* clk;
    CK(hipMalloc(&src, 4096 * 16)); CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&clk, 256 * 16));
    std::vector<unsigned short> h(4096 * 8);
    srand(1);
    for (auto& v : h) v = (unsigned short)(((rand() & 1) << 15) | ((12 + (rand() & 3)) << 10) | (rand() & 0x3ff));
    CK(hipMemcpy(src, h.data(), 4096 * 16, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("32x32x16, A shared by 3 consecutive MFMAs", src, out, clk, 15 * 2.0 * 32 * 32 * 16);
        run<1>("32x32x16, B shared by 5 consecutive MFMAs", src, out, clk, 15 * 2.0 * 32 * 32 * 16);
        run<2>("16x16x32, A shared by 6 consecutive MFMAs", src, out, clk, 60 * 2.0 * 16 * 16 * 32);
    }
    return 0;
}
