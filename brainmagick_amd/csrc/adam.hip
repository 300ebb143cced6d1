// Fused Adam over ONE flat fp32 parameter bucket (torch.optim.Adam as configured at
// bm/train.py:118-119: no weight decay, no amsgrad).  One launch for all 58 tensors; the flat
// bucket is also the unit of the RCCL reduce-scatter / all-gather, so a rank can
// update only its shard [lo, hi) (ZeRO-1 style) between the two collectives.
#include "bm_common.h"

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, float lr, float beta1, float beta2, float eps,
                            float bc1, float bc2_sqrt, float grad_scale, float omb1, float omb2) {
    const float step_size = lr / bc1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i] * grad_scale;
        // exp_avg.lerp_(grad, 1 - beta1)
        const float mi = m[i] + (gi - m[i]) * omb1;
        // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        const float vi = v[i] * beta2 + omb2 * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
        m[i] = mi;
        v[i] = vi;
    }
}

extern "C" int bm_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n,
                            int step, double lr, double beta1, double beta2, double eps,
                            double grad_scale, void* stream) {
    BM_REQUIRE(param && grad && exp_avg && exp_avg_sq, "adam_step: null pointer");
    BM_REQUIRE(step >= 1, "adam_step: step is 1-based");
    if (n == 0) return BM_OK;
    const double bc1 = 1.0 - pow(beta1, step);
    const double bc2 = 1.0 - pow(beta2, step);
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, n, (float)lr, (float)beta1, (float)beta2, (float)eps, (float)bc1,
                       (float)sqrt(bc2), (float)grad_scale, (float)(1.0 - beta1), (float)(1.0 - beta2));
    return bm_check_launch("adam_step");
}
