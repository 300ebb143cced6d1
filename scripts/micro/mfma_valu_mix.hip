// How many independent VALU instructions fit in the shadow of one v_mfma_f32_32x32x16_bf16 (32 cycles) when a
// single wavefront per SIMD issues both?  cycles / MFMA for k = 0..6 VALU ops of three kinds per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int K, int KIND>
__global__ __launch_bounds__(256, 1) void k(const u32x4* src, float* out, long long* clk, int iters) {
    bf16x8 a = __builtin_bit_cast(bf16x8, src[threadIdx.x]), b = __builtin_bit_cast(bf16x8, src[threadIdx.x + 256]);
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f;
    unsigned u0 = 0;
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (KIND == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u0) : "v"(v0), "v"(v1));
                if (KIND == 1) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u0) : "v"(v2));
                if (KIND == 2) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(v5) : "v"(v3), "v"(v4));
            }
        }
    }
    const long long c1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s + v5 + (float)u0;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = c1 - c0;
}

template <int K, int KIND> void run(const u32x4* src, float* out, long long* clk, const char* name) {
    const int iters = 4000;
    k<K, KIND><<<256, 256>>>(src, out, clk, 10);
    k<K, KIND><<<256, 256>>>(src, out, clk, iters);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    printf("%s x%d per MFMA: %.1f cycles/MFMA\n", name, K, (double)c / (iters * 8.0));
}

int main() {
    u32x4* src; float* out; long long* clk;
    hipMalloc(&src, 512 * 16); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&clk, 16);
    hipMemset(src, 0x3f, 512 * 16);
    run<0, 0>(src, out, clk, "none");
    run<1, 0>(src, out, clk, "cvt_pk_bf16"); run<2, 0>(src, out, clk, "cvt_pk_bf16"); run<3, 0>(src, out, clk, "cvt_pk_bf16");
    run<4, 0>(src, out, clk, "cvt_pk_bf16"); run<6, 0>(src, out, clk, "cvt_pk_bf16");
    run<2, 1>(src, out, clk, "lshlrev"); run<4, 1>(src, out, clk, "lshlrev"); run<6, 1>(src, out, clk, "lshlrev"); run<8, 1>(src, out, clk, "lshlrev");
    run<2, 2>(src, out, clk, "sub_f32"); run<4, 2>(src, out, clk, "sub_f32"); run<6, 2>(src, out, clk, "sub_f32"); run<8, 2>(src, out, clk, "sub_f32");
    return 0;
}
