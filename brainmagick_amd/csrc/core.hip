// Error convention + version of the C-ABI (include/bm_hip.h).
#include "bm_common.h"
#include <stdarg.h>
#include <stdlib.h>

thread_local char bm_err_buf[512] = {0};

int bm_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(bm_err_buf, sizeof(bm_err_buf), fmt, ap);
    va_end(ap);
    return code;
}

int bm_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return bm_set_error((int)e, "%s: launch failed: %s", what, hipGetErrorString(e));
    return BM_OK;
}

extern "C" const char* bm_last_error(void) { return bm_err_buf; }
extern "C" int bm_version(void) { return 106; }   // 0.1.6: round 6 (bm_act_bn_bwd_fused_fallbacks; the wide conv reads its packed weights straight into fragment registers -- same entry points, same packed layout)

// Number of HIP devices visible; <0 on error.  Lets the Python side fail loudly early.
extern "C" int bm_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { bm_set_error((int)e, "hipGetDeviceCount: %s", hipGetErrorString(e)); return -1; }
    return n;
}

// out[0] = max_i ws[i] (NaN-ignoring fmaxf; partial maxima are >= 0), out[1 .. BM_AMAX_SHARDS) = 0
__global__ __launch_bounds__(1024) void amax_finalize_kernel(const float* __restrict__ ws, int n, float* __restrict__ out) {
    __shared__ float sh[16];
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, ws[i]);
    m = bm_wave_max(m);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x < BM_AMAX_SHARDS) {
        float v = 0.f;
        if (threadIdx.x == 0)
            for (int w = 0; w < 16; ++w) v = fmaxf(v, sh[w]);
        out[threadIdx.x] = v;
    }
}

int bm_amax_finalize(const float* ws, int n, float* out, hipStream_t stream) {
    if (!out) return BM_OK;
    hipLaunchKernelGGL(amax_finalize_kernel, dim3(1), dim3(1024), 0, stream, ws, n, out);
    return bm_check_launch("amax_finalize");
}

// Partials laid out [nsplit][C] (the (channel, split) grids of the BatchNorm / GLU backward kernels): rows_out[c] =
// max over the splits -- the per-channel maxima that give every row of a weight gradient its own f16x2 scale
// (gemm_nt_h2w.hip, RS kernels) -- and the tensor maximum into the slot, in ONE one-workgroup launch.
__global__ __launch_bounds__(1024) void amax_finalize_rows_kernel(const float* __restrict__ ws, int C, int nsplit,
                                                                  float* __restrict__ out, float* __restrict__ rows_out,
                                                                  const double* __restrict__ sum_partial,
                                                                  float* __restrict__ sums_out) {
    __shared__ float sh[16];
    float m = 0.f;
    for (int c = threadIdx.x; c < C; c += 1024) {
        float r = 0.f;
        for (int s = 0; s < nsplit; ++s) r = fmaxf(r, ws[(long)s * C + c]);
        if (rows_out) rows_out[c] = r;
        m = fmaxf(m, r);
        if (sum_partial) {                  // the same launch folds the producer's per-(channel, split) sums (bias gradient)
            double t = 0;
            for (int s = 0; s < nsplit; ++s) t += sum_partial[(long)c * nsplit + s];
            sums_out[c] = (float)t;
        }
    }
    m = bm_wave_max(m);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x < BM_AMAX_SHARDS) {
        float v = 0.f;
        if (threadIdx.x == 0)
            for (int w = 0; w < 16; ++w) v = fmaxf(v, sh[w]);
        out[threadIdx.x] = v;
    }
}

int bm_amax_finalize_rows(const float* ws, int C, int nsplit, float* out, float* rows_out, hipStream_t stream) {
    if (!out) return BM_OK;
    hipLaunchKernelGGL(amax_finalize_rows_kernel, dim3(1), dim3(1024), 0, stream, ws, C, nsplit, out, rows_out,
                       (const double*)nullptr, (float*)nullptr);
    return bm_check_launch("amax_finalize_rows");
}

// ... and, in the same launch, sums_out[c] = sum_split sum_partial[c][split] (doubles; the bias gradient of the layer)
int bm_amax_finalize_rows_sums(const float* ws, int C, int nsplit, float* out, float* rows_out,
                               const double* sum_partial, float* sums_out, hipStream_t stream) {
    hipLaunchKernelGGL(amax_finalize_rows_kernel, dim3(1), dim3(1024), 0, stream, ws, C, nsplit, out, rows_out,
                       sum_partial, sums_out);
    return bm_check_launch("amax_finalize_rows");
}

extern "C" int bm_amax_ws_elems(void) { return BM_AMAX_WS; }
