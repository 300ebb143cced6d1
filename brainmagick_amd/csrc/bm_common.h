// Shared device/host helpers for libbmhip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define BM_OK 0
#define BM_ERR_ARG 1001
#define BM_ERR_WORKSPACE 1002
#define BM_ERR_UNSUPPORTED 1003

// thread-local last error message (C-ABI convention: functions return int, message via bm_last_error)
extern thread_local char bm_err_buf[512];
int bm_set_error(int code, const char* fmt, ...);
int bm_check_launch(const char* what);

#define BM_REQUIRE(cond, ...)                                   \
    do {                                                        \
        if (!(cond)) return bm_set_error(BM_ERR_ARG, __VA_ARGS__); \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Channel chunk of the packed-weight layout (see pack.hip / conv_nn.hip).
#define BM_BKC 16
// Wavefront width on CDNA.
#define BM_WAVE 64

// XCD-aware bijective remap of a linear workgroup id: the dispatcher places block i on XCD i % 8;
// give every XCD a contiguous range of logical ids so that neighbouring tiles (which share an
// operand panel) hit the same per-XCD L2.  (cdna_hip_programming.md T1, bijective form.)
__device__ __forceinline__ int bm_xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = id & 7, slot = id >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

// Activations used by SimpleConv (bm/models/simpleconv.py:85-90).
enum { BM_ACT_NONE = 0, BM_ACT_GELU = 1, BM_ACT_RELU = 2, BM_ACT_LEAKY = 3 };

__device__ __forceinline__ float bm_act(float z, int act, float leak) {
    if (act == BM_ACT_GELU) return 0.5f * z * (1.0f + erff(z * 0.70710678118654752440f));
    if (act == BM_ACT_RELU) return z > 0.f ? z : 0.f;
    if (act == BM_ACT_LEAKY) return z > 0.f ? z : z * leak;
    return z;
}

__device__ __forceinline__ float bm_act_grad(float z, int act, float leak) {
    if (act == BM_ACT_GELU) {
        const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752440f));
        const float pdf = 0.39894228040143267794f * expf(-0.5f * z * z);
        return cdf + z * pdf;
    }
    if (act == BM_ACT_RELU) return z > 0.f ? 1.f : 0.f;
    if (act == BM_ACT_LEAKY) return z > 0.f ? 1.f : leak;
    return 1.f;
}

__device__ __forceinline__ float bm_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double bm_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float bm_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
