// ChannelMerger front end (bm/models/common.py:312-362): Fourier positional embedding of the 2-D
// sensor layout, masked softmax over sensors (wave-shuffle reductions, one wavefront per row) and
// its backward.  The dense contractions around it (logits = heads . emb^T, out = weights . meg and
// their gradients) run on the MFMA kernels of gemm_nt.hip / conv_nn.hip.
//
// MI355X-first difference to the reference: the embedding / logits / softmax depend only on the
// sensor LAYOUT, not on the segment, so they are computed once per distinct layout in the batch
// (U rows) instead of once per segment (B rows) -- 115 MMAC/segment of redundant work removed.
#include "bm_common.h"

#define BM_INVALID_POS (-0.1f)

// emb[u][c][d]: d < D/2 -> cos(loc_d), else sin(loc_{d-D/2});  loc_{kx*nf+ky} =
//   (px+margin) * (2*pi*kx/width) + (py+margin) * (2*pi*ky/width)     (common.py:254-271)
// fp32 operation order mirrors the reference's tensor ops (no FMA contraction), so the arguments
// of cos/sin are bit-identical to torch's.
__global__ void fourier_emb_kernel(const float* __restrict__ pos, float* __restrict__ emb, long rows,
                                   int D, int nf, float margin) {
    const int half = D / 2;
    const float two_pi = 6.283185307179586f;
    const float width = __fadd_rn(1.0f, __fmul_rn(2.0f, margin));
    const long total = rows * half;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const long row = i / half;
        const int d = (int)(i - row * half);
        const int kx = d / nf, ky = d - kx * nf;
        const float px = __fadd_rn(pos[row * 2 + 0], margin);
        const float py = __fadd_rn(pos[row * 2 + 1], margin);
        const float fx = __fdiv_rn(__fmul_rn(two_pi, (float)kx), width);
        const float fy = __fdiv_rn(__fmul_rn(two_pi, (float)ky), width);
        const float loc = __fadd_rn(__fmul_rn(px, fx), __fmul_rn(py, fy));
        emb[row * D + d] = cosf(loc);
        emb[row * D + half + d] = sinf(loc);
    }
}

extern "C" int bm_fourier_emb(const float* positions, float* emb, long rows, int D, float margin,
                              void* stream) {
    BM_REQUIRE(positions && emb, "fourier_emb: null pointer");
    const int nf = (int)lroundf(sqrtf((float)(D / 2)));
    BM_REQUIRE(nf * nf * 2 == D, "fourier_emb: dimension %d is not 2*n^2", D);
    const long total = rows * (D / 2);
    if (total == 0) return BM_OK;
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(fourier_emb_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, positions,
                       emb, rows, D, nf, margin);
    return bm_check_launch("fourier_emb");
}

// weights[u][o][:] = softmax_c(scores[u][o][c] + offset[u][c]); offset = -inf for INVALID sensors
// (both coordinates == -0.1, common.py:235-236,340) and, when ban_radius > 0, for sensors within
// ban_radius of the ban centre (common.py:342-346).  An all-masked row yields NaN like the reference.
__global__ void masked_softmax_kernel(const float* __restrict__ scores, const float* __restrict__ pos,
                                      const float* __restrict__ ban_center, float ban_radius,
                                      float* __restrict__ weights, int U, int O, int C) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= (long)U * O) return;
    const int u = (int)(row / O);
    const float* sr = scores + row * C;
    const float* pr = pos + (long)u * C * 2;
    float cx = 0.f, cy = 0.f;
    if (ban_radius > 0.f) { cx = ban_center[0]; cy = ban_center[1]; }
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) {
        const float px = pr[c * 2], py = pr[c * 2 + 1];
        bool masked = (px == BM_INVALID_POS) && (py == BM_INVALID_POS);
        if (ban_radius > 0.f) {
            const float dx = __fsub_rn(px, cx), dy = __fsub_rn(py, cy);
            masked = masked || (sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy))) <= ban_radius);
        }
        const float s = masked ? -INFINITY : sr[c];
        mx = fmaxf(mx, s);
    }
    mx = bm_wave_max(mx);
    float sum = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float px = pr[c * 2], py = pr[c * 2 + 1];
        bool masked = (px == BM_INVALID_POS) && (py == BM_INVALID_POS);
        if (ban_radius > 0.f) {
            const float dx = __fsub_rn(px, cx), dy = __fsub_rn(py, cy);
            masked = masked || (sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy))) <= ban_radius);
        }
        const float s = masked ? -INFINITY : sr[c];
        const float e = expf(s - mx);          // (-inf) - (-inf) = NaN for an all-masked row
        weights[row * C + c] = e;
        sum += e;
    }
    sum = bm_wave_sum(sum);
    const float inv = 1.f / sum;
    for (int c = lane; c < C; c += 64) weights[row * C + c] *= inv;
}

extern "C" int bm_masked_softmax(const float* scores, const float* positions, const float* ban_center,
                                 float ban_radius, float* weights, int U, int O, int C, void* stream) {
    BM_REQUIRE(scores && positions && weights, "masked_softmax: null pointer");
    BM_REQUIRE(ban_radius <= 0.f || ban_center, "masked_softmax: ban radius without centre");
    const long rows = (long)U * O;
    if (rows == 0) return BM_OK;
    hipLaunchKernelGGL(masked_softmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, scores, positions, ban_center, ban_radius, weights, U, O, C);
    return bm_check_launch("masked_softmax");
}

// dscores = w * (dw - sum_c w*dw)    (masked entries have w = 0)
__global__ void softmax_bwd_kernel(const float* __restrict__ w, const float* __restrict__ dw,
                                   float* __restrict__ ds, long rows, int C) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    float dot = 0.f;
    for (int c = lane; c < C; c += 64) dot += w[row * C + c] * dw[row * C + c];
    dot = bm_wave_sum(dot);
    for (int c = lane; c < C; c += 64) ds[row * C + c] = w[row * C + c] * (dw[row * C + c] - dot);
}

extern "C" int bm_softmax_bwd(const float* w, const float* dw, float* ds, long rows, int C,
                              void* stream) {
    BM_REQUIRE(w && dw && ds, "softmax_bwd: null pointer");
    if (rows == 0) return BM_OK;
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, w, dw, ds, rows, C);
    return bm_check_launch("softmax_bwd");
}
