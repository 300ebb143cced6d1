// ClipLoss (bm/losses.py:77-114): candidate inverse norms, row-wise softmax cross-entropy over the
// [B, B'] score matrix with the target on the diagonal, probabilities, and the score gradient.
// The two dense contractions (scores = est . cand^T over K = F*T, and dEst = dScores . cand) run on
// the MFMA kernels (gemm_nt.hip split-K, conv_nn.hip).
#include "bm_common.h"

// inv_norm[o] = 1 / (1e-8 + ||cand[o]||_2)      (losses.py:91)
__global__ __launch_bounds__(256) void inv_norms_kernel(const float* __restrict__ cand, long K,
                                                        float* __restrict__ inv_norm) {
    __shared__ double sh[4];
    const float* row = cand + (long)blockIdx.x * K;
    double s = 0;
    if ((K & 3) == 0) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (long i = threadIdx.x; i < K / 4; i += blockDim.x) {
            const float4 v = reinterpret_cast<const float4*>(row)[i];
            a0 += v.x * v.x; a1 += v.y * v.y; a2 += v.z * v.z; a3 += v.w * v.w;
        }
        s = (double)a0 + (double)a1 + (double)a2 + (double)a3;
    } else {
        float a = 0.f;
        for (long i = threadIdx.x; i < K; i += blockDim.x) a += row[i] * row[i];
        s = a;
    }
    s = bm_wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double t = sh[0] + sh[1] + sh[2] + sh[3];
        inv_norm[blockIdx.x] = 1.0f / (1e-8f + (float)sqrt(t));
    }
}

extern "C" int bm_clip_inv_norms(const float* cand, int Bc, long K, float* inv_norm, void* stream) {
    BM_REQUIRE(cand && inv_norm, "clip_inv_norms: null pointer");
    if (Bc == 0) return BM_OK;
    hipLaunchKernelGGL(inv_norms_kernel, dim3(Bc), dim3(256), 0, (hipStream_t)stream, cand, K, inv_norm);
    return bm_check_launch("clip_inv_norms");
}

// One wavefront per estimate row b:
//   scores[b][o] = inv_norm[o] * sum_split part[split][b][o]
//   loss_row[b]  = logsumexp_o(scores[b]) - scores[b][tgt], tgt = b + target_offset
//                  (F.cross_entropy, target = arange(B); the offset lets a rank point at its own
//                   block of whole-node gathered candidates without re-ordering them)
//   probs[b][o]  = softmax_o(scores[b])                            (get_probabilities, losses.py:97-102)
//   dscaled[b][o]= (probs - [o==b]) / B * inv_norm[o]              (d loss / d(est.cand[o]) )
__global__ void clip_ce_kernel(const float* __restrict__ part, int nsplit, const float* __restrict__ inv_norm,
                               float* __restrict__ scores, float* __restrict__ probs,
                               float* __restrict__ dscaled, float* __restrict__ loss_row, int B, int Bc,
                               int target_offset) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= B) return;
    const long per = (long)B * Bc;
    const int tgt = b + target_offset;
    float mx = -INFINITY;
    for (int o = lane; o < Bc; o += 64) {
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += part[(long)k * per + (long)b * Bc + o];
        s *= inv_norm[o];
        scores[(long)b * Bc + o] = s;
        mx = fmaxf(mx, s);
    }
    mx = bm_wave_max(mx);
    float sum = 0.f;
    for (int o = lane; o < Bc; o += 64) sum += expf(scores[(long)b * Bc + o] - mx);
    sum = bm_wave_sum(sum);
    const float lse = mx + logf(sum);
    const float inv = 1.f / sum;
    for (int o = lane; o < Bc; o += 64) {
        const float s = scores[(long)b * Bc + o];
        const float p = expf(s - mx) * inv;
        if (probs) probs[(long)b * Bc + o] = p;
        if (dscaled) dscaled[(long)b * Bc + o] = (p - (o == tgt ? 1.f : 0.f)) / (float)B * inv_norm[o];
        if (o == tgt && loss_row) loss_row[b] = lse - s;
    }
}

__global__ void mean_kernel(const float* __restrict__ x, int n, float* __restrict__ out) {
    __shared__ double sh[4];
    double s = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
    s = bm_wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *out = (float)((sh[0] + sh[1] + sh[2] + sh[3]) / n);
}

extern "C" int bm_clip_ce(const float* part, int nsplit, const float* inv_norm, float* scores,
                          float* probs, float* dscaled, float* loss_row, float* loss, int B, int Bc,
                          int target_offset, void* stream) {
    BM_REQUIRE(part && inv_norm && scores, "clip_ce: null pointer");
    BM_REQUIRE(!loss || (target_offset >= 0 && target_offset + B <= Bc),
               "clip_ce: need at least as many targets as estimates");
    BM_REQUIRE(!loss || loss_row, "clip_ce: loss needs loss_row scratch");
    if (B == 0) return BM_OK;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(clip_ce_kernel, dim3((B + 3) / 4), dim3(256), 0, s, part, nsplit, inv_norm, scores,
                       probs, dscaled, loss_row, B, Bc, target_offset);
    if (loss) hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, s, loss_row, B, loss);
    return bm_check_launch("clip_ce");
}
