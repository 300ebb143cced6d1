// Weight packing for conv_nn.hip, grouping of segments by an index (subject / layout), and the
// reduction of split-K partial tiles into strided destinations.
#include "bm_common.h"

// dst[g][chunk][j][cc][Mpad]  <-  alpha * src[g*sg + m*sm + c*sc + jj*sj],   c = chunk*BKC + cc,
// jj = flip ? KS-1-j : j; zero for m >= M or c >= Cin.  The strides express every weight layout of
// the path: Conv1d (M,Cin,KS) forward; its data-gradient (roles of m and c swapped + tap flip);
// SubjectLayers (S,Cin,Cout) read as W^T; ConvTranspose1d(k=1) (in,out,1).
__global__ void pack_weights_kernel(const float* __restrict__ src, float* __restrict__ dst, int G,
                                    int M, int Cin, int KS, long sg, long sm, long sc, long sj,
                                    int flip, int Mpad, int nchunk, const float* alpha_ptr) {
    const long total = (long)G * nchunk * KS * BM_BKC * Mpad;
    const float alpha = alpha_ptr ? *alpha_ptr : 1.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        long r = i;
        const int m = (int)(r % Mpad); r /= Mpad;
        const int cc = (int)(r % BM_BKC); r /= BM_BKC;
        const int j = (int)(r % KS); r /= KS;
        const int chunk = (int)(r % nchunk);
        const int g = (int)(r / nchunk);
        const int c = chunk * BM_BKC + cc;
        float v = 0.f;
        if (m < M && c < Cin) {
            const int jj = flip ? KS - 1 - j : j;
            v = alpha * src[g * sg + m * sm + c * sc + jj * sj];
        }
        dst[i] = v;
    }
}

extern "C" int bm_conv_mpad(int M);

extern "C" long bm_packed_weight_elems(int G, int M, int Cin, int KS) {
    return (long)G * cdiv(Cin, BM_BKC) * KS * BM_BKC * bm_conv_mpad(M);
}

extern "C" int bm_pack_weights(const float* src, float* dst, int G, int M, int Cin, int KS, long sg,
                               long sm, long sc, long sj, int flip, const float* alpha_ptr,
                               void* stream) {
    BM_REQUIRE(src && dst, "pack_weights: null pointer");
    BM_REQUIRE(G > 0 && M > 0 && Cin > 0 && KS > 0, "pack_weights: bad dims");
    const int Mpad = bm_conv_mpad(M);
    const int nchunk = cdiv(Cin, BM_BKC);
    const long total = (long)G * nchunk * KS * BM_BKC * Mpad;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst,
                       G, M, Cin, KS, sg, sm, sc, sj, flip, Mpad, nchunk, alpha_ptr);
    return bm_check_launch("pack_weights");
}

// Stable counting sort of segments by group index: order[] lists the segments of group 0, then
// group 1, ...; seg[g]..seg[g+1] delimits group g.  Used to turn the per-subject / per-layout
// weight-gradient scatter (bm/models/common.py:55-58 backward) into a deterministic grouped GEMM.
__global__ void group_by_index_kernel(const long* __restrict__ idx, int B, int G,
                                      int* __restrict__ order, int* __restrict__ seg,
                                      int* __restrict__ err) {
    extern __shared__ int counts[];   // [G + 1] running offsets, then [B] the indices (-1 = out of range)
    int* ids = counts + G + 1;
    for (int g = threadIdx.x; g <= G; g += blockDim.x) counts[g] = 0;
    __syncthreads();
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const long v = idx[b];
        if (v < 0 || v >= G) {
            if (err) atomicExch(err, 1);
            ids[b] = -1;
        } else {
            ids[b] = (int)v;
            atomicAdd(&counts[v], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int g = 0; g < G; ++g) { const int c = counts[g]; counts[g] = run; seg[g] = run; run += c; }
        counts[G] = run;
        seg[G] = run;
    }
    __syncthreads();
    // each group is filled in ascending segment order by ONE thread -> deterministic (the scan reads LDS, eight
    // values per round: one dependent global load per element took 28 us at B = 256)
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        int pos = counts[g];
        const int end = counts[g + 1];
        for (int b0 = 0; b0 < B && pos < end; b0 += 8) {      // 8 independent LDS reads per round
            int v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = b0 + u < B ? ids[b0 + u] : -1;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (v[u] == g) order[pos++] = b0 + u;
        }
    }
}

extern "C" int bm_group_by_index(const long* idx, int B, int G, int* order, int* seg, int* err_flag,
                                 void* stream) {
    BM_REQUIRE(idx && order && seg, "group_by_index: null pointer");
    BM_REQUIRE(B >= 0 && B <= 16000 && G > 0 && G <= 16000, "group_by_index: bad dims B=%d G=%d", B, G);
    hipLaunchKernelGGL(group_by_index_kernel, dim3(1), dim3(256), (size_t)(G + 1 + B) * sizeof(int),
                       (hipStream_t)stream, idx, B, G, order, seg, err_flag);
    return bm_check_launch("group_by_index");
}

// int64 group indices (subject_index, layout index) -> int32 for the grouped conv kernels, range-checked:
// an index outside [0, G) sets *err (the reference's W[subjects] gather, bm/models/common.py:57, raises on
// it) and is replaced by 0 so that no kernel ever reads outside the weight table.
__global__ void index_to_i32_kernel(const long* __restrict__ idx, int B, int G, int* __restrict__ out,
                                    int* __restrict__ err) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    long v = idx[i];
    if (v < 0 || v >= G) {
        if (err) atomicExch(err, 1);
        v = 0;
    }
    out[i] = (int)v;
}

extern "C" int bm_index_to_i32(const long* idx, int B, int G, int* out, int* err_flag, void* stream) {
    BM_REQUIRE(idx && out, "index_to_i32: null pointer");
    BM_REQUIRE(B >= 0 && G > 0, "index_to_i32: bad dims B=%d G=%d", B, G);
    if (B == 0) return BM_OK;
    hipLaunchKernelGGL(index_to_i32_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, idx, B,
                       G, out, err_flag);
    return bm_check_launch("index_to_i32");
}

// out[g*sg + m*sm + c*sc + j*sj] = sum_split part[(g*nsplit + split)][m][c*KS + j]
// (fixed summation order -> deterministic split-K).
__global__ void reduce_splits_kernel(const float* __restrict__ part, float* __restrict__ out, int G,
                                     int nsplit, int M, int Cn, int KS, long sg, long sm, long sc,
                                     long sj) {
    const long N = (long)Cn * KS;
    const long per = (long)M * N;
    const long total = (long)G * per;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int g = (int)(i / per);
        const long r = i - (long)g * per;
        const int m = (int)(r / N);
        const int n = (int)(r - (long)m * N);
        const int c = n / KS, j = n - c * KS;
        const float* p = part + (long)g * nsplit * per + r;
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += p[(long)k * per];
        out[g * sg + m * sm + c * sc + j * sj] = s;
    }
}

// Same sum, four consecutive partial-tile elements per thread (one dwordx4 per split, eight splits in
// flight), for per % 4 == 0.  The order of the additions per element is the same k = 0, 1, ... as above.
__global__ void reduce_splits_vec4_kernel(const float* __restrict__ part, float* __restrict__ out, int G,
                                          int nsplit, int M, int Cn, int KS, long sg, long sm, long sc,
                                          long sj) {
    const long N = (long)Cn * KS;
    const long per = (long)M * N;
    const long total4 = (long)G * per / 4;
    for (long i4 = (long)blockIdx.x * blockDim.x + threadIdx.x; i4 < total4;
         i4 += (long)gridDim.x * blockDim.x) {
        const long i = i4 * 4;
        const int g = (int)(i / per);
        const long r = i - (long)g * per;
        const float* p = part + (long)g * nsplit * per + r;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int k = 0;
        for (; k + 8 <= nsplit; k += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (long)(k + u) * per);
#pragma unroll
            for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; k < nsplit; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(p + (long)k * per);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long re = r + e;
            const int m = (int)(re / N);
            const int n = (int)(re - (long)m * N);
            const int c = n / KS, j = n - c * KS;
            out[g * sg + m * sm + c * sc + j * sj] = sv[e];
        }
    }
}

extern "C" int bm_reduce_splits(const float* part, float* out, int G, int nsplit, int M, int Cn,
                                int KS, long sg, long sm, long sc, long sj, void* stream) {
    BM_REQUIRE(part && out, "reduce_splits: null pointer");
    const long total = (long)G * M * Cn * KS;
    if (total <= 0) return BM_OK;
    const long per = (long)M * Cn * KS;
    if (per % 4 == 0 && ((uintptr_t)part % 16) == 0) {
        const long t4 = total / 4;
        const int blocks = (int)((t4 + 255) / 256 > 8192 ? 8192 : (t4 + 255) / 256);
        hipLaunchKernelGGL(reduce_splits_vec4_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, part,
                           out, G, nsplit, M, Cn, KS, sg, sm, sc, sj);
        return bm_check_launch("reduce_splits");
    }
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(reduce_splits_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, part,
                       out, G, nsplit, M, Cn, KS, sg, sm, sc, sj);
    return bm_check_launch("reduce_splits");
}

// out[i] = sum_b x[b][i]   (fixed order).  Used to fold per-layout partial head gradients.
__global__ void sum_over_batch_kernel(const float* __restrict__ x, float* __restrict__ out, int B,
                                      long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += x[(long)b * n + i];
        out[i] = s;
    }
}

extern "C" int bm_sum_over_batch(const float* x, float* out, int B, long n, void* stream) {
    BM_REQUIRE(x && out, "sum_over_batch: null pointer");
    if (n <= 0) return BM_OK;
    const int blocks = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
    hipLaunchKernelGGL(sum_over_batch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, out,
                       B, n);
    return bm_check_launch("sum_over_batch");
}
