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

// n / d for 32-bit n by multiply-high (Granlund & Montgomery): the flat-index kernels split an element index into
// (row, column) once per vector; a 64-bit hardware-less division there cost more instructions than the erf.
struct BmFastDiv {
    unsigned mul, shift, d;
};
static inline BmFastDiv bm_fastdiv(unsigned d) {
    BmFastDiv f;
    f.d = d;
    unsigned l = 0;
    while ((1ull << l) < d) ++l;                                  // l = ceil(log2 d)
    f.mul = (unsigned)((((1ull << l) - d) << 32) / d + 1);
    f.shift = l;
    return f;
}
__host__ __device__ __forceinline__ unsigned bm_div(unsigned n, const BmFastDiv& f) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned t = __umulhi(n, f.mul);
#else
    const unsigned t = (unsigned)(((unsigned long long)n * f.mul) >> 32);
#endif
    return f.shift == 0 ? n : (t + ((n - t) >> 1)) >> (f.shift - 1);
}

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

// erf: the library erff.  (A branch-free evaluation of both minimax polynomials of a faithfully-rounded erff --
// 15 FMAs + one exp per element, no per-lane divergence -- was measured against it on the cfg2 step, same box,
// interleaved: 0.15 ms per step SLOWER, profiles/r3_ab_notes.md.  The GELU passes are instruction-bound either way.)
__device__ __forceinline__ float bm_erff(float a) { return erff(a); }

__device__ __forceinline__ float bm_act(float z, int act, float leak) {
    if (act == BM_ACT_GELU) return 0.5f * z * (1.0f + bm_erff(z * 0.70710678118654752440f));
    if (act == BM_ACT_RELU) return z > 0.f ? z : 0.f;
    if (act == BM_ACT_LEAKY) return z > 0.f ? z : z * leak;
    return z;
}

// d/dz GELU(z) = Phi(z) + z phi(z) with ONE exponential and no erf (BM_ACT_GELU_FASTGRAD, what the backward kernels
// use; the erff + expf form below serves the generic bm_act_grad):
//     e = exp(-z^2 / 2),   Phi(-|z|) = e Q(t),  t = 1 / (1 + 0.24 |z|),
// Q = degree-7 polynomial without constant term, fitted (scripts/fit_gelu_grad.py) to |e Q - Phi(-|z|)| <= 3.1e-10 on
// [0, 14]; then  gelu'(z) = [z >= 0] + e (z / sqrt(2 pi)) -+ e Q.  ~14 VALU (2 transcendental) instead of ~45 with a
// divergent branch: the two BatchNorm-backward passes were instruction-bound (3.9 / 4.3 TB/s).  In fp32 arithmetic
// its error against fp64 is BELOW that of the reference's own fp32 formula 0.5 (1 + erf(z / sqrt 2)) + z pdf(z)
// (max 2.6e-7 / rms 2.9e-8 against 2.8e-7 / 4.8e-8, emulated element by element over [-12, 12]; the pointwise GPU
// test holds it to the same 4e-7 as before).  z = +-inf gives NaN like the reference's inf * 0.
__device__ __forceinline__ float bm_gelu_grad_fast(float z) {
    const float a = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(a, 0.24f, 1.0f));
    const float e = __builtin_amdgcn_exp2f(z * z * -0.72134752044448170368f);     // exp(-z^2 / 2)
    float q = -0.12508568359372038f;
    q = fmaf(q, t, 0.5545737304338939f);
    q = fmaf(q, t, -0.646070256151912f);
    q = fmaf(q, t, 0.6255898343952563f);
    q = fmaf(q, t, -0.15404857581855858f);
    q = fmaf(q, t, 0.15551350568092193f);
    q = fmaf(q, t, 0.08952744474497322f);
    const float eq = e * (q * t);                                                  // Phi(-|z|)
    const float base = z >= 0.f ? 1.0f - eq : eq;                                  // Phi(z)
    return fmaf(e, z * 0.39894228040143267794f, base);
}
#define BM_ACT_GELU_FASTGRAD 4      // internal to the backward kernels (norm_act.hip): GELU, derivative as above

__device__ __forceinline__ float bm_act_grad(float z, int act, float leak) {
    if (act == BM_ACT_GELU_FASTGRAD) return bm_gelu_grad_fast(z);
    if (act == BM_ACT_GELU) {
        const float cdf = 0.5f * (1.0f + bm_erff(z * 0.70710678118654752440f));
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
// Where a producer publishes max |output|: ws[workgroup id] = the workgroup's maximum (plain store), folded into the
// tensor's amax slot by the one-workgroup bm_amax_finalize launch.  (Round 3 measured an atomic-max form that needs no
// finalize launch: 0.26-0.33 ms per step SLOWER -- one agent-scope atomic per workgroup delays the retirement of the
// 16 384-workgroup streaming kernels by more than the ~60 finalize launches cost; removed in round 4.)
struct BmAmaxDst {
    float* ws;
};
__device__ __forceinline__ void bm_publish_amax_at(float m, const BmAmaxDst& dst, float* sh /* >= blockDim / 64 floats of LDS */,
                                                   unsigned wg /* index of the partial: workgroup or tile id */) {
    if (!dst.ws) return;                               // kernel argument: uniform
    m = bm_wave_max(m);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        float v = sh[0];
        for (int w = 1; w < nw; ++w) v = fmaxf(v, sh[w]);
        dst.ws[wg] = v;
    }
}
__device__ __forceinline__ void bm_publish_amax(float m, const BmAmaxDst& dst, float* sh) {
    bm_publish_amax_at(m, dst, sh, blockIdx.y * gridDim.x + blockIdx.x);
}
// plain-pointer form: two-stage mode into `ws`
__device__ __forceinline__ void bm_publish_amax(float m, float* ws, float* sh) {
    bm_publish_amax(m, BmAmaxDst{ws}, sh);
}
__device__ __forceinline__ float bm_amax_load(const float* slot) {
    float m = slot[0];
#pragma unroll
    for (int i = 1; i < BM_AMAX_SHARDS; ++i) m = fmaxf(m, slot[i]);
    return m;
}
// host: reduce n partial maxima (ws) into the slot `out`; no-op when out is null
int bm_amax_finalize(const float* ws, int n, float* out, hipStream_t stream);
// host: fold partials laid out [nsplit][C] into the slot AND into per-channel maxima rows_out[C] (nullable)
int bm_amax_finalize_rows(const float* ws, int C, int nsplit, float* out, float* rows_out, hipStream_t stream);
// host: the same launch also folds per-(channel, split) double sums [C][nsplit] into sums_out[C] (out must be non-null)
int bm_amax_finalize_rows_sums(const float* ws, int C, int nsplit, float* out, float* rows_out,
                               const double* sum_partial, float* sums_out, hipStream_t stream);
// host: destination of a producer's maximum, and what follows its launch
static inline BmAmaxDst bm_amax_dst(float* amax_out, float* amax_ws) {
    return BmAmaxDst{amax_out ? amax_ws : nullptr};
}
static inline int bm_amax_done(const BmAmaxDst& d, int nblocks, float* amax_out, hipStream_t stream) {
    return d.ws ? bm_amax_finalize(d.ws, nblocks, amax_out, stream) : BM_OK;
}

__device__ __forceinline__ float bm_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
