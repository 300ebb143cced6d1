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

// The candidates' three per-step reductions in ONE pass over them (they are inputs: 44 MB at F = 120, 377 MB at
// F = 1024 per 256 segments): inv_norm[o] (losses.py:91), max |cand| for the f16x2 scale of the score contraction
// (raised into the ZEROED amax slot with atomic max) and the reference's finiteness assert (bm/solver.py:258-260).
__global__ __launch_bounds__(256) void cand_prep_kernel(const float* __restrict__ cand, long K,
                                                        float* __restrict__ inv_norm, float* __restrict__ amax_slot,
                                                        int* __restrict__ nonfinite) {
    __shared__ double sh[4];
    __shared__ float shm[4];
    const float* row = cand + (long)blockIdx.x * K;
    double s = 0;
    float mx = 0.f;
    unsigned top = 0u;
    if ((K & 3) == 0 && (((uintptr_t)row & 15) == 0)) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (long i = threadIdx.x; i < K / 4; i += blockDim.x) {
            const float4 v = reinterpret_cast<const float4*>(row)[i];
            a0 += v.x * v.x; a1 += v.y * v.y; a2 += v.z * v.z; a3 += v.w * v.w;
            mx = fmaxf(fmaxf(mx, fabsf(v.x)), fmaxf(fabsf(v.y), fmaxf(fabsf(v.z), fabsf(v.w))));
            top = max(max(top, __float_as_uint(v.x) & 0x7fffffffu),
                      max(__float_as_uint(v.y) & 0x7fffffffu, max(__float_as_uint(v.z) & 0x7fffffffu,
                                                                   __float_as_uint(v.w) & 0x7fffffffu)));
        }
        s = (double)a0 + (double)a1 + (double)a2 + (double)a3;
    } else {
        float a = 0.f;
        for (long i = threadIdx.x; i < K; i += blockDim.x) {
            const float v = row[i];
            a += v * v;
            mx = fmaxf(mx, fabsf(v));
            top = max(top, __float_as_uint(v) & 0x7fffffffu);
        }
        s = a;
    }
    if (nonfinite && top >= 0x7f800000u) atomicOr(nonfinite, 1);
    s = bm_wave_sum_d(s);
    mx = bm_wave_max(mx);
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = s; shm[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double t = sh[0] + sh[1] + sh[2] + sh[3];
        inv_norm[blockIdx.x] = 1.0f / (1e-8f + (float)sqrt(t));
        if (amax_slot) {
            const float m = fmaxf(fmaxf(shm[0], shm[1]), fmaxf(shm[2], shm[3]));
            unsigned* p = reinterpret_cast<unsigned*>(amax_slot) + (blockIdx.x & (BM_AMAX_SHARDS - 1));
            const unsigned bits = __float_as_uint(m);
            if (bits > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                (void)__hip_atomic_fetch_max(p, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// inv_norm as bm_clip_inv_norms (same summation order: bit-identical); amax_slot (nullable): ZEROED amax slot
// that receives max |cand|; nonfinite_flag (nullable): set to 1 when cand holds an inf or a nan.
extern "C" int bm_clip_cand_prep(const float* cand, int Bc, long K, float* inv_norm, float* amax_slot,
                                 int* nonfinite_flag, void* stream) {
    BM_REQUIRE(cand && inv_norm, "clip_cand_prep: null pointer");
    if (Bc == 0) return BM_OK;
    hipLaunchKernelGGL(cand_prep_kernel, dim3(Bc), dim3(256), 0, (hipStream_t)stream, cand, K, inv_norm, amax_slot,
                       nonfinite_flag);
    return bm_check_launch("clip_cand_prep");
}

// ClipLoss.forward asserts `mask.all()` (bm/losses.py:110) -- a host synchronisation in the middle of the step.  Under
// the Solver the verdict goes to the device-side flag word instead: *flag |= 1 when any byte of the bool mask is 0.
__global__ __launch_bounds__(256) void flag_unless_all_set_kernel(const unsigned char* __restrict__ mask, long n,
                                                                  int* __restrict__ flag) {
    bool bad = false;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        bad |= mask[i] == 0;
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

extern "C" int bm_flag_unless_all_set(const unsigned char* mask, long n, int* flag, void* stream) {
    BM_REQUIRE(flag && (mask || n == 0) && n >= 0, "flag_unless_all_set: bad arguments");
    if (n == 0) return BM_OK;
    long blocks = (n + 255) / 256;
    blocks = blocks > 512 ? 512 : blocks;
    hipLaunchKernelGGL(flag_unless_all_set_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, mask, n,
                       flag);
    return bm_check_launch("flag_unless_all_set");
}

// One workgroup (4 wavefronts) per estimate row b:
//   scores[b][o] = inv_norm[o] * sum_split part[split][b][o]      (fixed split order: deterministic)
//   loss_row[b]  = logsumexp_o(scores[b]) - scores[b][tgt], tgt = b + target_offset
//                  (F.cross_entropy, target = arange(B); the offset lets a rank point at its own
//                   block of whole-node gathered candidates without re-ordering them)
//   probs[b][o]  = softmax_o(scores[b])                            (get_probabilities, losses.py:97-102)
//   dscaled[b][o]= (probs - [o==b]) / B * inv_norm[o]              (d loss / d(est.cand[o]) )
// The fold of the split-K partial tiles is the expensive part (nsplit * B * B' * 4 bytes): consecutive threads
// read consecutive columns of one partial row (coalesced), eight splits in flight per thread.
__device__ __forceinline__ float clip_block_max(float v, float* sh) {
    v = bm_wave_max(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    v = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
    return v;
}
__device__ __forceinline__ float clip_block_sum(float v, float* sh) {
    v = bm_wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    v = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    return v;
}

__global__ __launch_bounds__(256) void clip_ce_kernel(const float* __restrict__ part, int nsplit,
                                                      const float* __restrict__ inv_norm, float* __restrict__ scores,
                                                      float* __restrict__ probs, float* __restrict__ dscaled,
                                                      float* __restrict__ loss_row, int B, int Bc, int target_offset,
                                                      const float* __restrict__ col_valid) {
    __shared__ float sh[4];
    const int b = blockIdx.x;
    const long per = (long)B * Bc;
    const int tgt = b + target_offset;
    float* srow = scores + (long)b * Bc;
    float mx = -INFINITY;
    for (int o = threadIdx.x; o < Bc; o += 256) {
        const float* p = part + (long)b * Bc + o;
        float s = 0.f;
        int k = 0;
        for (; k + 8 <= nsplit; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(long)(k + u) * per];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < nsplit; ++k) s += p[(long)k * per];
        s *= inv_norm[o];
        // a masked candidate (padding row of a rank that rejected segments, Solver negatives="node") never matches:
        // score -inf -> probability 0, gradient 0
        if (col_valid && col_valid[o] == 0.f) s = -INFINITY;
        srow[o] = s;
        mx = fmaxf(mx, s);
    }
    mx = clip_block_max(mx, sh);
    float sum = 0.f;
    for (int o = threadIdx.x; o < Bc; o += 256) sum += expf(srow[o] - mx);      // own writes: same thread, same o
    sum = clip_block_sum(sum, sh);
    const float lse = mx + logf(sum);
    const float inv = 1.f / sum;
    for (int o = threadIdx.x; o < Bc; o += 256) {
        const float s = srow[o];
        const float pr = expf(s - mx) * inv;
        if (probs) probs[(long)b * Bc + o] = pr;
        if (dscaled) dscaled[(long)b * Bc + o] = s == -INFINITY ? 0.f : (pr - (o == tgt ? 1.f : 0.f)) / (float)B * inv_norm[o];
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

extern "C" int bm_clip_ce_masked(const float* part, int nsplit, const float* inv_norm, const float* col_valid,
                                 float* scores, float* probs, float* dscaled, float* loss_row, float* loss, int B,
                                 int Bc, int target_offset, void* stream);

extern "C" int bm_clip_ce(const float* part, int nsplit, const float* inv_norm, float* scores,
                          float* probs, float* dscaled, float* loss_row, float* loss, int B, int Bc,
                          int target_offset, void* stream) {
    return bm_clip_ce_masked(part, nsplit, inv_norm, nullptr, scores, probs, dscaled, loss_row, loss, B, Bc,
                             target_offset, stream);
}

// `col_valid` ([Bc] floats, nullable): candidates with col_valid[o] == 0 are masked out of every row's softmax
extern "C" int bm_clip_ce_masked(const float* part, int nsplit, const float* inv_norm, const float* col_valid,
                                 float* scores, float* probs, float* dscaled, float* loss_row, float* loss, int B,
                                 int Bc, int target_offset, void* stream) {
    BM_REQUIRE(part && inv_norm && scores, "clip_ce: null pointer");
    BM_REQUIRE(!loss || (target_offset >= 0 && target_offset + B <= Bc),
               "clip_ce: need at least as many targets as estimates");
    BM_REQUIRE(!loss || loss_row, "clip_ce: loss needs loss_row scratch");
    if (B == 0) return BM_OK;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(clip_ce_kernel, dim3(B), dim3(256), 0, s, part, nsplit, inv_norm, scores,
                       probs, dscaled, loss_row, B, Bc, target_offset, col_valid);
    if (loss) hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, s, loss_row, B, loss);
    return bm_check_launch("clip_ce");
}

// ------------------------------------------------------------------------------------------------
// Column softmax (the "audio -> brain" direction of the symmetric CLIP objective; an opt-in extension, the
// reference's ClipLoss is the row term alone, bm/losses.py:104-114).  For every target candidate o = off + j the
// column scores[:, o] is a B-way classification over the estimates with the answer in row j:
//     loss_col[j] = logsumexp_b scores[b][o] - scores[j][o]
// and the combined gradient  dscaled = w_row * dscaled_row + w_col * (softmax_col - onehot) / B * inv_norm[o]
// is written over the row term in place (candidates that are negatives only keep w_row * their row term).
// One wavefront per column, lanes over the rows (the matrix is L2-resident: B * B' * 4 bytes), shuffle reductions.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void clip_ce_cols_kernel(const float* __restrict__ scores,
                                                           const float* __restrict__ inv_norm,
                                                           float* __restrict__ dscaled, float* __restrict__ loss_col,
                                                           int B, int Bc, int off, float w_row, float w_col) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= Bc) return;
    const int j = o - off;
    if (j < 0 || j >= B) {                       // a negative without a label: only the weight of the row term
        if (dscaled) for (int b = lane; b < B; b += 64) dscaled[(long)b * Bc + o] *= w_row;
        return;
    }
    float mx = -INFINITY;
    for (int b = lane; b < B; b += 64) mx = fmaxf(mx, scores[(long)b * Bc + o]);
    mx = bm_wave_max(mx);
    float sum = 0.f;
    for (int b = lane; b < B; b += 64) sum += expf(scores[(long)b * Bc + o] - mx);
    sum = bm_wave_sum(sum);
    const float lse = mx + logf(sum);
    if (lane == 0 && loss_col) loss_col[j] = lse - scores[(long)j * Bc + o];
    if (!dscaled) return;
    const float inv = 1.f / sum;
    const float g = w_col / (float)B * inv_norm[o];
    for (int b = lane; b < B; b += 64) {
        const float pr = expf(scores[(long)b * Bc + o] - mx) * inv;
        const long at = (long)b * Bc + o;
        dscaled[at] = fmaf(w_row, dscaled[at], g * (pr - (b == j ? 1.f : 0.f)));
    }
}

// loss <- w_row * loss + w_col * mean(loss_col)
__global__ void clip_mix_loss_kernel(const float* __restrict__ loss_col, int n, float* __restrict__ loss,
                                     float w_row, float w_col) {
    __shared__ double sh[4];
    double s = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += loss_col[i];
    s = bm_wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *loss = w_row * *loss + w_col * (float)((sh[0] + sh[1] + sh[2] + sh[3]) / n);
}

// scores [B][Bc] as written by bm_clip_ce; dscaled (nullable) holds the row term and is updated in place; loss
// (nullable) holds the row loss and becomes the weighted sum.  loss_col: [B] scratch / per-column losses.
extern "C" int bm_clip_ce_cols(const float* scores, const float* inv_norm, float* dscaled, float* loss_col,
                               float* loss, int B, int Bc, int target_offset, float w_row, float w_col,
                               void* stream) {
    BM_REQUIRE(scores && inv_norm && loss_col, "clip_ce_cols: null pointer");
    BM_REQUIRE(target_offset >= 0 && target_offset + B <= Bc, "clip_ce_cols: need at least as many targets as estimates");
    if (B == 0) return BM_OK;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(clip_ce_cols_kernel, dim3((Bc + 3) / 4), dim3(256), 0, s, scores, inv_norm, dscaled,
                       loss_col, B, Bc, target_offset, w_row, w_col);
    if (loss) hipLaunchKernelGGL(clip_mix_loss_kernel, dim3(1), dim3(256), 0, s, loss_col, B, loss, w_row, w_col);
    return bm_check_launch("clip_ce_cols");
}

// ------------------------------------------------------------------------------------------------
// Retrieval evaluation (scripts/run_eval_probs.py:237-264 _get_accuracy_from_probs; bm/wer.py:104-111):
// top-k columns of every probability / score row and "is the row's label among the labels of its
// top-k candidates".  One wavefront per row; k selection passes, each a strided scan + wave arg-max
// (ties broken towards the lower column index), no per-lane candidate lists (k stays in SGPR-land).
// ------------------------------------------------------------------------------------------------
__global__ void topk_rows_kernel(const float* __restrict__ x, int rows, int cols, int k,
                                 int* __restrict__ idx_out, float* __restrict__ val_out,
                                 const long* __restrict__ col_labels, const long* __restrict__ row_labels,
                                 int* __restrict__ hit_out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (long)row * cols;
    float prev_v = INFINITY;
    int prev_i = -1;
    int hit = 0;
    const long want = row_labels ? row_labels[row] : 0;
    for (int p = 0; p < k; ++p) {
        float best_v = -INFINITY;
        int best_i = cols;               // sentinel: nothing left
        for (int c = lane; c < cols; c += 64) {
            const float v = xr[c];
            const bool eligible = (v < prev_v) || (v == prev_v && c > prev_i);
            if (eligible && (v > best_v || (v == best_v && c < best_i))) { best_v = v; best_i = c; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best_v, o);
            const int oi = __shfl_xor(best_i, o);
            if (ov > best_v || (ov == best_v && oi < best_i)) { best_v = ov; best_i = oi; }
        }
        if (best_i >= cols) best_i = -1;  // fewer than k columns
        if (lane == 0) {
            if (idx_out) idx_out[(long)row * k + p] = best_i;
            if (val_out) val_out[(long)row * k + p] = best_v;
        }
        if (best_i >= 0 && col_labels && col_labels[best_i] == want) hit = 1;
        prev_v = best_v;
        prev_i = best_i < 0 ? cols : best_i;
    }
    if (lane == 0 && hit_out) hit_out[row] = hit;
}

extern "C" int bm_topk_rows(const float* x, int rows, int cols, int k, int* idx_out, float* val_out,
                            const long* col_labels, const long* row_labels, int* hit_out, void* stream) {
    BM_REQUIRE(x, "topk_rows: null pointer");
    BM_REQUIRE(k >= 1 && cols >= 1, "topk_rows: bad k/cols");
    BM_REQUIRE((col_labels == nullptr) == (row_labels == nullptr), "topk_rows: labels come in pairs");
    BM_REQUIRE(!hit_out || col_labels, "topk_rows: hit_out needs labels");
    if (rows == 0) return BM_OK;
    hipLaunchKernelGGL(topk_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, rows,
                       cols, k, idx_out, val_out, col_labels, row_labels, hit_out);
    return bm_check_launch("topk_rows");
}

// ------------------------------------------------------------------------------------------------
// Gradient of ClipLoss w.r.t. the CANDIDATES (learnable feature model, bm/solver.py:304-320 +
// bm/models/features.py DeepMel):  with s[b,o] = inv_o * <est_b, cand_o>,  inv_o = 1/(1e-8+|cand_o|)
//   dcand_o = sum_b dscaled[b,o] * est_b  -  coef_o * cand_o,
//   coef_o  = (sum_b dscaled[b,o] * s[b,o]) / |cand_o|
// The first term is an MFMA GEMM (conv_nn.hip); this file provides coef and the rank-1 correction.
// ------------------------------------------------------------------------------------------------
__global__ void clip_cand_coef_kernel(const float* __restrict__ dscaled, const float* __restrict__ scores,
                                      const float* __restrict__ inv_norm, const float* __restrict__ alpha,
                                      float* __restrict__ coef, int B, int Bc) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= Bc) return;
    double r = 0;
    // a masked candidate (bm_clip_ce_masked) carries score -inf and gradient 0: its terms are skipped, 0 * -inf is NaN
    for (int b = 0; b < B; ++b) {
        const float d = dscaled[(long)b * Bc + o];
        if (d != 0.f) r += (double)d * (double)scores[(long)b * Bc + o];
    }
    const float norm = 1.0f / inv_norm[o] - 1e-8f;
    const float a = alpha ? *alpha : 1.f;
    coef[o] = norm > 0.f ? a * (float)r / norm : 0.f;
}

extern "C" int bm_clip_cand_coef(const float* dscaled, const float* scores, const float* inv_norm,
                                 const float* alpha, float* coef, int B, int Bc, void* stream) {
    BM_REQUIRE(dscaled && scores && inv_norm && coef, "clip_cand_coef: null pointer");
    if (Bc == 0) return BM_OK;
    hipLaunchKernelGGL(clip_cand_coef_kernel, dim3((Bc + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       dscaled, scores, inv_norm, alpha, coef, B, Bc);
    return bm_check_launch("clip_cand_coef");
}

// y[r][i] -= coef[r] * x[r][i]
__global__ void row_axpy_kernel(float* __restrict__ y, const float* __restrict__ x,
                                const float* __restrict__ coef, long K, long K4) {
    const int r = blockIdx.y;
    const float c = coef[r];
    float* yr = y + (long)r * K;
    const float* xr = x + (long)r * K;
    if (K4) {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < K4; i += (long)gridDim.x * blockDim.x) {
            float4 a = reinterpret_cast<float4*>(yr)[i];
            const float4 b = reinterpret_cast<const float4*>(xr)[i];
            a.x -= c * b.x; a.y -= c * b.y; a.z -= c * b.z; a.w -= c * b.w;
            reinterpret_cast<float4*>(yr)[i] = a;
        }
    } else {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < K; i += (long)gridDim.x * blockDim.x)
            yr[i] -= c * xr[i];
    }
}

extern "C" int bm_row_axpy_sub(float* y, const float* x, const float* coef, int rows, long K, void* stream) {
    BM_REQUIRE(y && x && coef, "row_axpy_sub: null pointer");
    if (rows == 0 || K == 0) return BM_OK;
    const long K4 = (K % 4 == 0 && (((uintptr_t)y | (uintptr_t)x) % 16 == 0)) ? K / 4 : 0;
    const long n = K4 ? K4 : K;
    int bx = (int)((n + 255) / 256);
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(row_axpy_kernel, dim3(bx, rows), dim3(256), 0, (hipStream_t)stream, y, x, coef, K, K4);
    return bm_check_launch("row_axpy_sub");
}

// ------------------------------------------------------------------------------------------------
// Word-level retrieval (bm/wer.py:91-116), batched.  Row softmax of a score matrix, per-row dot
// products est_i . out_i (the segment's own target replaces the last negative, wer.py:93-94), and the
// aggregation of candidate probabilities per vocabulary word (wer.py:100-103) as a deterministic
// segmented sum over columns pre-sorted by word.
// ------------------------------------------------------------------------------------------------
__global__ void row_softmax_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (long)row * cols;
    float mx = -INFINITY;
    for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, xr[c]);
    mx = bm_wave_max(mx);
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += expf(xr[c] - mx);
    s = bm_wave_sum(s);
    const float inv = 1.f / s;
    for (int c = lane; c < cols; c += 64) y[(long)row * cols + c] = expf(xr[c] - mx) * inv;
}

extern "C" int bm_row_softmax(const float* x, float* y, int rows, int cols, void* stream) {
    BM_REQUIRE(x && y, "row_softmax: null pointer");
    if (rows == 0) return BM_OK;
    hipLaunchKernelGGL(row_softmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, y, rows, cols);
    return bm_check_launch("row_softmax");
}

// out[r] = scale[r] * <a[r], b[r]>   (rows of length K)
__global__ __launch_bounds__(256) void rowwise_dot_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          const float* __restrict__ scale, float* __restrict__ out,
                                                          long K) {
    __shared__ double sh[4];
    const float* ar = a + (long)blockIdx.x * K;
    const float* br = b + (long)blockIdx.x * K;
    float s = 0.f;
    for (long i = threadIdx.x; i < K; i += blockDim.x) s += ar[i] * br[i];
    double d = bm_wave_sum_d((double)s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (float)(sh[0] + sh[1] + sh[2] + sh[3]) * (scale ? scale[blockIdx.x] : 1.f);
}

extern "C" int bm_rowwise_dot(const float* a, const float* b, const float* scale, float* out, int rows, long K,
                              void* stream) {
    BM_REQUIRE(a && b && out, "rowwise_dot: null pointer");
    if (rows == 0) return BM_OK;
    hipLaunchKernelGGL(rowwise_dot_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, a, b, scale, out, K);
    return bm_check_launch("rowwise_dot");
}

// pv[r][v] = sum_{k in [seg[v], seg[v+1])} p[r][order[k]]   (columns grouped by vocabulary word)
__global__ void segment_sum_cols_kernel(const float* __restrict__ p, const int* __restrict__ order,
                                        const int* __restrict__ seg, float* __restrict__ pv, int rows,
                                        int cols, int V) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* pr = p + (long)row * cols;
    for (int v = lane; v < V; v += 64) {
        float s = 0.f;
        for (int k = seg[v]; k < seg[v + 1]; ++k) s += pr[order[k]];
        pv[(long)row * V + v] = s;
    }
}

extern "C" int bm_segment_sum_cols(const float* p, const int* order, const int* seg, float* pv, int rows,
                                   int cols, int V, void* stream) {
    BM_REQUIRE(p && order && seg && pv, "segment_sum_cols: null pointer");
    if (rows == 0 || V == 0) return BM_OK;
    hipLaunchKernelGGL(segment_sum_cols_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, p, order,
                       seg, pv, rows, cols, V);
    return bm_check_launch("segment_sum_cols");
}
