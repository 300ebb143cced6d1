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
// max |x| of a tensor, published by the kernel that produces it (consumed as the f16x2 scale of the next
// contraction, conv_nn_h2w.hip): every workgroup folds its maximum through LDS and stores it to ws[workgroup id]
// (plain store: no atomics, nothing to zero beforehand); a one-workgroup kernel (bm_amax_finalize, core.hip)
// then reduces the partial maxima into the amax slot.  A slot is BM_AMAX_SHARDS floats whose maximum is the
// answer (the consumers take bm_amax_load); the workspace holds BM_AMAX_WS floats, the upper bound of every
// producer's grid, and can be shared by all producers of one stream.
#define BM_AMAX_SHARDS 8
#define BM_AMAX_WS 16384
__device__ __forceinline__ float bm_wave_max(float v);
__device__ __forceinline__ void bm_publish_amax(float m, float* ws, float* sh /* >= blockDim / 64 floats of LDS */) {
    if (!ws) return;                                   // kernel argument: uniform
    m = bm_wave_max(m);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        float v = sh[0];
        for (int w = 1; w < nw; ++w) v = fmaxf(v, sh[w]);
        ws[blockIdx.y * gridDim.x + blockIdx.x] = v;
    }
}
__device__ __forceinline__ float bm_amax_load(const float* slot) {
    float m = slot[0];
#pragma unroll
    for (int i = 1; i < BM_AMAX_SHARDS; ++i) m = fmaxf(m, slot[i]);
    return m;
}
// host: reduce n partial maxima (ws) into the slot `out`; no-op when out is null
int bm_amax_finalize(const float* ws, int n, float* out, hipStream_t stream);

__device__ __forceinline__ float bm_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
