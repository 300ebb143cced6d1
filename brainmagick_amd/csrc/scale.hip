// Fused front end of Solver._process_batch (bm/solver.py:245-246): per-recording robust scaling of
// the MEG window, clamp and rejection statistics in ONE streaming pass.
//   bm/norm.py:86-87   RobustScaler.transform:  (X - center_[c]) / scale_[c]
//   bm/norm.py:255-261 BatchScaler._transform:  python loop over segments, scaler picked by recording
//   bm/norm.py:325-341 ScaleReject.__call__:    clamp_(+-limit) if clip; reject = max|meg| > limit
// The reference does this with a host loop + `.item()` per segment; here the recording index is a
// device tensor, the centre/scale tables live in HBM as [R][C], and max|x| per segment is an
// order-independent atomic max (deterministic).  fp32 op order = the reference's (sub, then div).
#include "bm_common.h"

template <int VEC>
__global__ void center_scale_kernel(const float* __restrict__ x, float* __restrict__ out,
                                    const long* __restrict__ group, const float* __restrict__ center,
                                    const float* __restrict__ scale, int B, int C, int T, int clip,
                                    float limit, unsigned int* __restrict__ maxabs_bits) {
    // grid: (row blocks, B); one row = (b, c), T contiguous
    const int b = blockIdx.y;
    const long g = group ? group[b] : 0;
    const int TV = T / VEC;
    float local_max = 0.f;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < C * TV; e += gridDim.x * blockDim.x) {
        const int c = e / TV;
        const int tv = e - c * TV;
        const float ce = center[g * C + c], sc = scale[g * C + c];
        const long off = ((long)b * C + c) * T + (long)tv * VEC;
        float v[VEC];
        if constexpr (VEC == 4) {
            const float4 t = *reinterpret_cast<const float4*>(x + off);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
            v[0] = x[off];
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float r = __fdiv_rn(__fsub_rn(v[i], ce), sc);
            if (clip) r = fminf(fmaxf(r, -limit), limit);
            local_max = fmaxf(local_max, fabsf(r));
            v[i] = r;
        }
        if constexpr (VEC == 4) {
            *reinterpret_cast<float4*>(out + off) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            out[off] = v[0];
        }
    }
    if (maxabs_bits) {
        local_max = bm_wave_max(local_max);
        // non-negative floats order like their bit patterns
        if ((threadIdx.x & 63) == 0) atomicMax(&maxabs_bits[b], __float_as_uint(local_max));
    }
}

// out[b][c][t] = clamp((x[b][c][t] - center[group[b]][c]) / scale[group[b]][c]); maxabs[b] (optional,
// must be zero-initialised by the caller) receives max_{c,t} |out[b]|.  In-place (out == x) allowed.
extern "C" int bm_center_scale(const float* x, float* out, const long* group, const float* center,
                               const float* scale, int B, int C, int T, int clip, float limit,
                               float* maxabs, void* stream) {
    BM_REQUIRE(x && out && center && scale, "center_scale: null pointer");
    if ((long)B * C * T == 0) return BM_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (T % 4 == 0) && (((uintptr_t)x | (uintptr_t)out) % 16 == 0);
    const int per = vec ? C * (T / 4) : C * T;
    int bx = (per + 255) / 256;
    if (bx > 64) bx = 64;
    if (vec)
        hipLaunchKernelGGL(center_scale_kernel<4>, dim3(bx, B), dim3(256), 0, s, x, out, group, center,
                           scale, B, C, T, clip, limit, (unsigned int*)maxabs);
    else
        hipLaunchKernelGGL(center_scale_kernel<1>, dim3(bx, B), dim3(256), 0, s, x, out, group, center,
                           scale, B, C, T, clip, limit, (unsigned int*)maxabs);
    return bm_check_launch("center_scale");
}
