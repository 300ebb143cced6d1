// HBM-bound epilogue kernels of the ConvSequence block (bm/models/common.py:113-151):
// BatchNorm1d (train: batch statistics over (B,T); eval: running statistics), GELU/ReLU/LeakyReLU,
// residual add, GLU -- forward and backward.  All tensors are fp32 [B][C][T] with T contiguous;
// every kernel streams rows with 16-byte accesses when T % 4 == 0 (the 360-sample case) and falls
// back to dword accesses otherwise (T = 343 / 361 ...).  Channel reductions are two-stage
// (per-(channel, split) partials in a fixed order -> deterministic).
#include <cstdlib>
#include "bm_common.h"

template <int VEC> struct Pack;
template <> struct Pack<1> {
    float v[1];
    __device__ static Pack ld(const float* p) { Pack r; r.v[0] = *p; return r; }
    __device__ void st(float* p) const { *p = v[0]; }
};
template <> struct Pack<4> {
    float v[4];
    __device__ static Pack ld(const float* p) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        Pack r; r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; return r;
    }
    __device__ void st(float* p) const { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};

// Walk of the flat (segment, vector) index e = threadIdx.x, threadIdx.x + blockDim.x, ... of a workgroup's
// [nb segments][TV vectors] slab without a division per element (the (channel, split) kernels below spend a
// dozen iterations per thread; `e / TV` per iteration was a fifth of their instructions).
struct SlabWalk {
    int bl, tv, dbl, dtv, TV;
    __device__ SlabWalk(int TV_, int first, int step) : TV(TV_) {
        bl = first / TV_; tv = first - bl * TV_;
        dbl = step / TV_; dtv = step - dbl * TV_;
    }
    __device__ void next() {
        tv += dtv; bl += dbl;
        if (tv >= TV) { tv -= TV; ++bl; }
    }
};

// Block reduction of NV doubles (sum); result valid in thread 0.
template <int NV>
__device__ void block_sum(double (&v)[NV], double* sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = bm_wave_sum_d(v[i]);
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < NV; ++i) sh[wave * NV + i] = v[i];
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double s = 0;
            for (int w = 0; w < nw; ++w) s += sh[w * NV + i];
            v[i] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// BatchNorm statistics: fold the per-tile (sum, sumsq) partials written by conv_nn's epilogue.
// torch semantics: normalise with the biased variance, update running_var with the unbiased one,
// running = (1 - momentum) * running + momentum * batch, num_batches_tracked += 1.
// ------------------------------------------------------------------------------------------------
__global__ void bn_finalize_kernel(const float* __restrict__ stats, long tstride, long cstride, int ntiles, int C, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* running_mean, float* running_var, long* num_batches,
                                   float momentum, float eps, float* __restrict__ mean_out,
                                   float* __restrict__ invstd_out, float* __restrict__ scale_out,
                                   float* __restrict__ shift_out) {
    const int c = blockIdx.x;
    double s = 0, s2 = 0;
    for (int t = threadIdx.x; t < ntiles; t += blockDim.x) {
        s += (double)stats[(long)t * tstride + (long)c * cstride + 0];
        s2 += (double)stats[(long)t * tstride + (long)c * cstride + 1];
    }
    s = bm_wave_sum_d(s);
    s2 = bm_wave_sum_d(s2);
    if (threadIdx.x == 0) {
        const double mean = s / count;
        double var = s2 / count - mean * mean;
        if (var < 0) var = 0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
        const float sc = g * invstd;
        mean_out[c] = (float)mean;
        invstd_out[c] = invstd;
        scale_out[c] = sc;
        shift_out[c] = bt - (float)mean * sc;
        // (a batch whose statistics are not finite -- a NaN / inf in the input, which the Solver reports a few launches
        // later through its flag word -- leaves the running estimates alone instead of poisoning them for good)
        if (running_mean && isfinite(mean) && isfinite(var)) {
            const double unbiased = count > 1 ? var * count / (count - 1) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
        if (num_batches && c == 0) *num_batches += 1;
    }
}

extern "C" int bm_bn_finalize(const float* stats, int ntiles, int C, long count, const float* gamma,
                              const float* beta, float* running_mean, float* running_var,
                              long* num_batches, float momentum, float eps, float* mean,
                              float* invstd, float* scale, float* shift, void* stream) {
    BM_REQUIRE(stats && mean && invstd && scale && shift, "bn_finalize: null pointer");
    BM_REQUIRE(C > 0 && ntiles > 0 && count > 0, "bn_finalize: bad dims");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream, stats, (long)C * 2, 2L, ntiles, C,
                       (double)count, gamma, beta, running_mean, running_var, num_batches, momentum,
                       eps, mean, invstd, scale, shift);
    return bm_check_launch("bn_finalize");
}

// The same for channel-major partials stats[C][ntiles][2] (what the wide f16x2 conv writes from its epilogue: a
// channel's ~1 000 partial rows are then one contiguous run for the workgroup that folds them).
extern "C" int bm_bn_finalize_cm(const float* stats, int ntiles, int C, long count, const float* gamma,
                                 const float* beta, float* running_mean, float* running_var,
                                 long* num_batches, float momentum, float eps, float* mean,
                                 float* invstd, float* scale, float* shift, void* stream) {
    BM_REQUIRE(stats && mean && invstd && scale && shift, "bn_finalize_cm: null pointer");
    BM_REQUIRE(C > 0 && ntiles > 0 && count > 0, "bn_finalize_cm: bad dims");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream, stats, 2L, (long)ntiles * 2, ntiles, C,
                       (double)count, gamma, beta, running_mean, running_var, num_batches, momentum,
                       eps, mean, invstd, scale, shift);
    return bm_check_launch("bn_finalize_cm");
}

// eval mode: scale/shift from the running statistics.
__global__ void bn_eval_affine_kernel(int C, const float* gamma, const float* beta,
                                      const float* running_mean, const float* running_var,
                                      float eps, float* mean, float* invstd, float* scale,
                                      float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float is = 1.0f / sqrtf(running_var[c] + eps);
    const float sc = (gamma ? gamma[c] : 1.f) * is;
    mean[c] = running_mean[c];
    invstd[c] = is;
    scale[c] = sc;
    shift[c] = (beta ? beta[c] : 0.f) - running_mean[c] * sc;
}

extern "C" int bm_bn_eval_affine(int C, const float* gamma, const float* beta,
                                 const float* running_mean, const float* running_var, float eps,
                                 float* mean, float* invstd, float* scale, float* shift,
                                 void* stream) {
    BM_REQUIRE(running_mean && running_var && scale && shift, "bn_eval_affine: null pointer");
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, C,
                       gamma, beta, running_mean, running_var, eps, mean, invstd, scale, shift);
    return bm_check_launch("bn_eval_affine");
}

// ------------------------------------------------------------------------------------------------
// forward: out = act(y * scale[c] + shift[c]) + res
// ------------------------------------------------------------------------------------------------
template <int VEC>
__global__ void affine_act_res_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                      const float* __restrict__ shift, const float* __restrict__ res,
                                      float* __restrict__ out, long nvec, int C, int T, int act,
                                      float leak, BmAmaxDst amax_ws, BmFastDiv div_tv, BmFastDiv div_c) {
    __shared__ float amax_sh[4];
    float amx = 0.f;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < nvec;
         e += (long)gridDim.x * blockDim.x) {
        const unsigned row = bm_div((unsigned)e, div_tv);           // nvec < 2^32 (host check)
        const int c = (int)(row - bm_div(row, div_c) * (unsigned)C);
        const float sc = scale ? scale[c] : 1.f, sh = scale ? shift[c] : 0.f;
        Pack<VEC> v = Pack<VEC>::ld(y + e * VEC);
        Pack<VEC> r;
        if (res) r = Pack<VEC>::ld(res + e * VEC);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float z = bm_act(v.v[i] * sc + sh, act, leak);
            if (res) z += r.v[i];
            v.v[i] = z;
            amx = fmaxf(amx, fabsf(z));
        }
        v.st(out + e * VEC);
    }
    bm_publish_amax(amx, amax_ws, amax_sh);
}

extern "C" int bm_affine_act_res(const float* y, const float* scale, const float* shift,
                                 const float* res, float* out, int B, int C, int T, int act,
                                 float leak, float* amax_out, float* amax_ws, void* stream) {
    BM_REQUIRE(y && out, "affine_act_res: null pointer");
    BM_REQUIRE(!amax_out || amax_ws, "affine_act_res: amax_out needs the amax workspace");
    const long n = (long)B * C * T;
    if (n == 0) return BM_OK;
    BM_REQUIRE(n < 0xffffffffL, "affine_act_res: tensor of %ld elements (32-bit element index)", n);
    hipStream_t s = (hipStream_t)stream;
    const BmAmaxDst amax_dst = bm_amax_dst(amax_out, amax_ws);
    int nblk = 0;
    if (T % 4 == 0) {
        const long nvec = n / 4;
        const int blocks = (int)((nvec + 255) / 256 > 16384 ? 16384 : (nvec + 255) / 256);
        hipLaunchKernelGGL(affine_act_res_kernel<4>, dim3(blocks), dim3(256), 0, s, y, scale, shift, res,
                           out, nvec, C, T, act, leak, amax_dst, bm_fastdiv((unsigned)(T / 4)), bm_fastdiv((unsigned)C));
        nblk = blocks;
    } else {
        const int blocks = (int)((n + 255) / 256 > 16384 ? 16384 : (n + 255) / 256);
        hipLaunchKernelGGL(affine_act_res_kernel<1>, dim3(blocks), dim3(256), 0, s, y, scale, shift, res,
                           out, n, C, T, act, leak, amax_dst, bm_fastdiv((unsigned)T), bm_fastdiv((unsigned)C));
        nblk = blocks;
    }
    if (int rc = bm_check_launch("affine_act_res")) return rc;
    return bm_amax_done(amax_dst, nblk, amax_out, s);
}

// ------------------------------------------------------------------------------------------------
// backward of  out = act(bn(y)) [+ res]
//   dz = dout * act'(z),  z = y*scale + shift
//   train BN:  dy = scale * (dz - mean(dz) - xhat * mean(dz*xhat)),  dgamma = sum dz*xhat, dbeta = sum dz
//   eval BN:   dy = scale * dz (dgamma / dbeta as in train mode, from the running statistics' xhat) ;  no BN:  dy = dz
// Pass 1 (bn_bwd_reduce): per-(channel, split) partial sums.  Pass 2 (bn_bwd_apply): dy + sum(dy).
// ------------------------------------------------------------------------------------------------
#define NSPLIT_MAX 32
#ifndef BWD_UNROLL
#define BWD_UNROLL 4      // slab positions per loop trip of the backward kernels (see bn_bwd_reduce_kernel)
#endif

template <int VEC>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const float* __restrict__ dout, const float* __restrict__ y, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
    double* __restrict__ partial, int B, int C, int T, int act, float leak) {
    __shared__ double sh[4 * 2];
    const int c = blockIdx.x, split = blockIdx.y, nsplit = gridDim.y;
    const int b0 = (int)((long)B * split / nsplit), b1 = (int)((long)B * (split + 1) / nsplit);
    const int TV = T / VEC;
    const float sc = scale ? scale[c] : 1.f, shf = scale ? shift[c] : 0.f;
    const float mu = mean ? mean[c] : 0.f, is = invstd ? invstd[c] : 1.f;
    double acc[2] = {0, 0};
    float s0 = 0.f, s1 = 0.f;
    const int nb = b1 - b0;
    // BWD_UNROLL positions per trip, all loads issued before the first use: two 16-byte loads in flight per thread
    // left these (channel, split) kernels latency-bound at ~5.0 TB/s (the flat-grid forward kernels reach 5.8-6.0)
    SlabWalk w(TV, threadIdx.x, blockDim.x);
    while (w.bl < nb) {
        long off[BWD_UNROLL];
        bool ok[BWD_UNROLL];
        Pack<VEC> d[BWD_UNROLL], v[BWD_UNROLL];
#pragma unroll
        for (int u = 0; u < BWD_UNROLL; ++u) {
            // (unconditional loads -- a position past the slab re-reads the trip's first one -- so that nothing but
            // the address arithmetic sits between the 2 * BWD_UNROLL loads)
            ok[u] = w.bl < nb;
            off[u] = ok[u] ? ((long)(b0 + w.bl) * C + c) * T + (long)w.tv * VEC : off[0];
            d[u] = Pack<VEC>::ld(dout + off[u]);
            v[u] = Pack<VEC>::ld(y + off[u]);
            if (ok[u]) w.next();
        }
#pragma unroll
        for (int u = 0; u < BWD_UNROLL; ++u) {
            if (!ok[u]) continue;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float dz = d[u].v[i] * bm_act_grad(v[u].v[i] * sc + shf, act, leak);
                s0 += dz;
                s1 += dz * ((v[u].v[i] - mu) * is);
            }
        }
    }
    acc[0] = s0; acc[1] = s1;
    block_sum<2>(acc, sh);
    if (threadIdx.x == 0) {
        partial[((long)c * nsplit + split) * 2 + 0] = acc[0];
        partial[((long)c * nsplit + split) * 2 + 1] = acc[1];
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const float* __restrict__ dout, const float* __restrict__ y, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
    const double* __restrict__ partial, int bn_train, float* __restrict__ dy,
    double* __restrict__ dy_partial, float* __restrict__ dgamma, float* __restrict__ dbeta, int B,
    int C, int T, int act, float leak, BmAmaxDst amax_ws, int reverse) {
    __shared__ double sh[4];
    __shared__ float amax_sh[4];
    float amx = 0.f;
    // `reverse`: walk the batch splits in the opposite order of the reduce pass that has just streamed the same two
    // tensors -- what that pass read last is what the memory-side cache still holds (236 of its 256 MB)
    const int c = blockIdx.x, nsplit = gridDim.y, split = reverse ? nsplit - 1 - (int)blockIdx.y : (int)blockIdx.y;
    const int b0 = (int)((long)B * split / nsplit), b1 = (int)((long)B * (split + 1) / nsplit);
    const int TV = T / VEC;
    const float sc = scale ? scale[c] : 1.f, shf = scale ? shift[c] : 0.f;
    const float mu = mean ? mean[c] : 0.f, is = invstd ? invstd[c] : 1.f;
    float k1 = 0.f, k2 = 0.f;
    if (scale && (bn_train || dgamma || dbeta)) {   // BatchNorm: the reduce pass ran (train, or eval with affine grads)
        double sdz = 0, sdzx = 0;
        for (int k = 0; k < nsplit; ++k) {
            sdz += partial[((long)c * nsplit + k) * 2 + 0];
            sdzx += partial[((long)c * nsplit + k) * 2 + 1];
        }
        if (bn_train) {
            const double n = (double)B * T;
            k1 = (float)(sdz / n);
            k2 = (float)(sdzx / n);
        }
        if (split == 0 && threadIdx.x == 0) {
            if (dgamma) dgamma[c] = (float)sdzx;
            if (dbeta) dbeta[c] = (float)sdz;
        }
    }
    float sdy = 0.f;
    const int nb = b1 - b0;
    SlabWalk w(TV, threadIdx.x, blockDim.x);
    while (w.bl < nb) {
        long off[BWD_UNROLL];
        bool ok[BWD_UNROLL];
        Pack<VEC> d[BWD_UNROLL], v[BWD_UNROLL];
#pragma unroll
        for (int u = 0; u < BWD_UNROLL; ++u) {
            // (unconditional loads -- a position past the slab re-reads the trip's first one -- so that nothing but
            // the address arithmetic sits between the 2 * BWD_UNROLL loads)
            ok[u] = w.bl < nb;
            off[u] = ok[u] ? ((long)(b0 + w.bl) * C + c) * T + (long)w.tv * VEC : off[0];
            d[u] = Pack<VEC>::ld(dout + off[u]);
            v[u] = Pack<VEC>::ld(y + off[u]);
            if (ok[u]) w.next();
        }
#pragma unroll
        for (int u = 0; u < BWD_UNROLL; ++u) {
            if (!ok[u]) continue;
            Pack<VEC> o;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float dz = d[u].v[i] * bm_act_grad(v[u].v[i] * sc + shf, act, leak);
                float g = dz;
                if (bn_train) g = sc * (dz - k1 - ((v[u].v[i] - mu) * is) * k2);
                else if (scale) g = sc * dz;
                o.v[i] = g;
                sdy += g;
                amx = fmaxf(amx, fabsf(g));
            }
            o.st(dy + off[u]);
        }
    }
    bm_publish_amax(amx, amax_ws, amax_sh);
    double acc[1] = {(double)sdy};
    block_sum<1>(acc, sh);
    if (threadIdx.x == 0 && dy_partial) dy_partial[(long)c * nsplit + split] = acc[0];
}

// ------------------------------------------------------------------------------------------------
// ONE-PASS BatchNorm backward (round 5).  The two-pass form above reads `dout` and `y` twice (236 + 236 MB per 320-channel
// layer at B = 256) because dy needs the channel's sums of dz and dz * xhat before the first element can be written.
// Here a workgroup keeps its slab of the channel IN REGISTERS between the two phases (dz and xhat of up to 256 * MAXIT
// float4 positions), the `nsplit` workgroups of a channel exchange their partial sums through memory inside the launch,
// and dout / y are read ONCE: 354 MB instead of 590 MB per layer, the GELU derivative evaluated once per element
// instead of twice, one launch instead of two.
//
// In-launch hand-off, placement-independent (cdna_hip_programming.md section 6, Guideline 16, form R2 "the data is
// the flag"): each partial sum is ONE naturally aligned 8-byte word, written with one agent-scope relaxed atomic store
// (write-through, sc1) and read with agent-scope relaxed atomic loads; the words start out as a sentinel (all ones,
// hipMemsetAsync in front of every launch; a sum that happens to have that NaN pattern is stored as the canonical NaN).
// Wave 0 polls the channel's 2 * nsplit words (one lane each, s_sleep between polls) until none is the sentinel: no
// fence, no counter, one memory round trip once the partners are done.
// NO assumption about dispatch order or residency: the poll is bounded, and a workgroup whose partners have not
// published in time COMPUTES THEIR SUMS ITSELF (re-reading their slabs with the very same per-thread summation order,
// so the result is bit-identical) and goes on -- a partner that was never resident can therefore not dead-lock it, it
// only makes it slower.  The sums are folded in split order by every workgroup alike: deterministic.
// ------------------------------------------------------------------------------------------------
#define FUSED_MAXIT 10            // float4 positions per thread held between the phases (80 VGPRs; 4 workgroups per CU)
#define FUSED_POLL_LIMIT 64u      // x s_sleep(64) ~ 0.13 ms: partners normally publish within ~10 us (they are dispatched together)
#define FUSED_SENTINEL 0xffffffffffffffffull

// dz and xhat of one element, and the running sums, in ONE place: the keeping pass and the fallback pass must round alike
__device__ __forceinline__ void fused_elem(float dv, float yv, float sc, float shf, float mu, float is, int act,
                                           float leak, bool ok, float& dz, float& xh, float& s0, float& s1) {
    dz = ok ? dv * bm_act_grad(fmaf(yv, sc, shf), act, leak) : 0.f;
    xh = (yv - mu) * is;
    s0 += dz;
    s1 = fmaf(dz, xh, s1);
}

// how many workgroups gave up on a partner and computed its sums themselves since the library was loaded (a slow path,
// never a wrong one): bm_act_bn_bwd_fused_fallbacks() reads it -- bench.py and the tests report it
__device__ unsigned long long bn_fused_fallbacks = 0ull;

template <int MAXIT>
__global__ __launch_bounds__(256, 4) void bn_bwd_fused_kernel(
    const float* __restrict__ dout, const float* __restrict__ y, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
    unsigned long long* partial, int bn_train, float* __restrict__ dy, double* __restrict__ dy_partial,
    float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int C, int T, int nsplit, int act, float leak,
    BmAmaxDst amax_ws, BmFastDiv div_tv, BmFastDiv div_ns, unsigned poll_limit) {
    __shared__ double sh[4 * 2];
    __shared__ float amax_sh[4];
    __shared__ float k_sh[2];
    __shared__ unsigned long long part_sh[2 * NSPLIT_MAX];
    __shared__ unsigned missing_sh;
    // consecutive workgroups = the splits of one channel (they start together under the observed dispatch order; the
    // protocol does not depend on it)
    const int c = (int)bm_div(blockIdx.x, div_ns), split = (int)(blockIdx.x - (unsigned)c * (unsigned)nsplit);
    const int TV = T / 4;
    const float sc = scale[c], shf = shift[c], mu = mean[c], is = invstd[c];
    const int b0 = (int)((long)B * split / nsplit), b1 = (int)((long)B * (split + 1) / nsplit);
    const unsigned nvec = (unsigned)(b1 - b0) * (unsigned)TV;            // <= 256 * MAXIT (host check)

    // phase 1: the slab into registers, all loads in flight before the first use
    float4 d[MAXIT], v[MAXIT];
    // element offset of position `it` of this thread (B * C * T < 2^32, host check); recomputed for the stores of phase 2
    // rather than kept: MAXIT registers decide between 4 and 3 workgroups per CU
    auto offset = [&](int it) {
        const unsigned e = (unsigned)it * 256u + threadIdx.x;
        const unsigned ee = e < nvec ? e : 0u;                           // past the slab: re-read position 0 (unused)
        const unsigned bl = bm_div(ee, div_tv);
        return ((unsigned)(b0 + (int)bl) * (unsigned)C + (unsigned)c) * (unsigned)T + (ee - bl * (unsigned)TV) * 4u;
    };
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const unsigned o = offset(it);
        d[it] = *reinterpret_cast<const float4*>(dout + o);
        v[it] = *reinterpret_cast<const float4*>(y + o);
    }
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const bool ok = (unsigned)it * 256u + threadIdx.x < nvec;
        float* dd = reinterpret_cast<float*>(&d[it]);
        float* vv = reinterpret_cast<float*>(&v[it]);
#pragma unroll
        for (int i = 0; i < 4; ++i) fused_elem(dd[i], vv[i], sc, shf, mu, is, act, leak, ok, dd[i], vv[i], s0, s1);
    }
    double acc[2] = {(double)s0, (double)s1};
    block_sum<2>(acc, sh);

    // publish this split's two sums (one 8-byte write-through word each)
    unsigned long long* chan = partial + (long)c * nsplit * 2;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            unsigned long long bits = __builtin_bit_cast(unsigned long long, acc[j]);
            if (bits == FUSED_SENTINEL) bits = 0x7ff8000000000000ull;
            __hip_atomic_store(chan + split * 2 + j, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            part_sh[split * 2 + j] = bits;
        }
    }
    // wave 0 polls the channel's words, one lane per word
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const bool mine = lane < 2 * nsplit && (lane >> 1) != split;
        unsigned long long bits = 0;
        bool ready = !mine;
        for (unsigned polls = 0;; ++polls) {
            if (!ready) {
                bits = __hip_atomic_load(chan + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ready = bits != FUSED_SENTINEL;
            }
            if (__all(ready) || polls >= poll_limit) break;
            __builtin_amdgcn_s_sleep(64);
        }
        if (mine && ready) part_sh[lane] = bits;
        const unsigned long long lagging = __ballot(!ready);             // bit 2k / 2k+1: split k has not published
        if (lane == 0) {
            unsigned m = 0;
            for (int k = 0; k < nsplit; ++k) m |= ((lagging >> (2 * k)) & 3ull) ? (1u << k) : 0u;
            missing_sh = m;
        }
    }
    __syncthreads();
    // partners that did not publish in time (never under a sane dispatcher): their sums, computed here, bit-identically
    const unsigned missing = missing_sh;
    if (missing) {
        if (threadIdx.x == 0) atomicAdd(&bn_fused_fallbacks, 1ull);
        for (int k = 0; k < nsplit; ++k) {
            if (!((missing >> k) & 1u)) continue;
            const int kb0 = (int)((long)B * k / nsplit), kb1 = (int)((long)B * (k + 1) / nsplit);
            const unsigned knvec = (unsigned)(kb1 - kb0) * (unsigned)TV;
            float t0 = 0.f, t1 = 0.f;
            for (int it = 0; it < MAXIT; ++it) {
                const unsigned e = (unsigned)it * 256u + threadIdx.x;
                const bool ok = e < knvec;
                const unsigned ee = ok ? e : 0u;
                const unsigned bl = bm_div(ee, div_tv);
                const unsigned o = ((unsigned)(kb0 + (int)bl) * (unsigned)C + (unsigned)c) * (unsigned)T + (ee - bl * (unsigned)TV) * 4u;
                const float4 dv = *reinterpret_cast<const float4*>(dout + o);
                const float4 yv = *reinterpret_cast<const float4*>(y + o);
                const float* dd = reinterpret_cast<const float*>(&dv);
                const float* vv = reinterpret_cast<const float*>(&yv);
                float dz, xh;
#pragma unroll
                for (int i = 0; i < 4; ++i) fused_elem(dd[i], vv[i], sc, shf, mu, is, act, leak, ok, dz, xh, t0, t1);
            }
            double a2[2] = {(double)t0, (double)t1};
            __syncthreads();
            block_sum<2>(a2, sh);
            if (threadIdx.x == 0) {
                part_sh[2 * k + 0] = __builtin_bit_cast(unsigned long long, a2[0]);
                part_sh[2 * k + 1] = __builtin_bit_cast(unsigned long long, a2[1]);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double sdz = 0, sdzx = 0;
        for (int k = 0; k < nsplit; ++k) {                               // split order: the same fold in every workgroup
            sdz += __builtin_bit_cast(double, part_sh[2 * k + 0]);
            sdzx += __builtin_bit_cast(double, part_sh[2 * k + 1]);
        }
        const double n = (double)B * T;
        k_sh[0] = bn_train ? (float)(sdz / n) : 0.f;
        k_sh[1] = bn_train ? (float)(sdzx / n) : 0.f;
        if (split == 0) {
            if (dgamma) dgamma[c] = (float)sdzx;
            if (dbeta) dbeta[c] = (float)sdz;
        }
    }
    __syncthreads();
    const float k1 = k_sh[0], k2 = k_sh[1];

    // phase 2: dy from the registers
    float sdy = 0.f, amx = 0.f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        if ((unsigned)it * 256u + threadIdx.x < nvec) {
            const float* dd = reinterpret_cast<const float*>(&d[it]);
            const float* vv = reinterpret_cast<const float*>(&v[it]);
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float g = sc * (dd[i] - k1 - vv[i] * k2);
                o[i] = g;
                sdy += g;
                amx = fmaxf(amx, fabsf(g));
            }
            *reinterpret_cast<float4*>(dy + offset(it)) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    bm_publish_amax_at(amx, amax_ws, amax_sh, (unsigned)split * (unsigned)C + (unsigned)c);
    double acc2[1] = {(double)sdy};
    block_sum<1>(acc2, sh);
    if (threadIdx.x == 0 && dy_partial) dy_partial[(long)c * nsplit + split] = acc2[0];
}

// A/B switch: 0 = the two-pass kernels, 1 = one pass (default; environment BM_BN_BWD_FUSED).
// Measured (profiles/r5_ab_notes.md): two passes 152-154 us stand-alone / ~118 us inside the step (the second pass
// re-reads from the memory-side cache), one pass 113-118 us stand-alone, -0.2 ... -0.27 ms per training step.  Two more
// forms were built, measured and removed: slabs of 5 float4 per thread with 6 workgroups per CU (119-125 us) and a
// software-pipelined persistent grid holding two slabs per workgroup (153 us: every workgroup of the chip reaches its
// hand-off at the same time, so nothing overlaps the ~7 us of hand-off latency per item).
static int g_bn_bwd_fused = -1;
// test hook: polls before a workgroup gives up on its partners and computes their sums itself (0: at once)
static unsigned g_fused_poll_limit = FUSED_POLL_LIMIT;
extern "C" int bm_act_bn_bwd_set_poll_limit(int polls) {
    const int prev = (int)g_fused_poll_limit;
    g_fused_poll_limit = polls < 0 ? FUSED_POLL_LIMIT : (unsigned)polls;
    return prev;
}
// debug counter (synchronises the device): workgroups of the one-pass kernel that took the self-sufficient fallback
extern "C" long bm_act_bn_bwd_fused_fallbacks() {
    unsigned long long n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(bn_fused_fallbacks), sizeof(n)) != hipSuccess) return -1;
    return (long)n;
}
extern "C" int bm_act_bn_bwd_set_fused(int mode) {
    const int prev = g_bn_bwd_fused;
    g_bn_bwd_fused = mode < 0 ? -1 : (mode ? 1 : 0);
    return prev;
}
static int fused_mode() {
    if (g_bn_bwd_fused < 0) {
        const char* e = getenv("BM_BN_BWD_FUSED");
        g_bn_bwd_fused = (e && e[0] == '0') ? 0 : 1;
    }
    return g_bn_bwd_fused;
}

extern "C" int bm_bwd_nsplit(int B);
// splits of the one-pass kernel: the smallest count whose per-split slab fits the registers; 0 = not covered
static int fused_nsplit(int B, int C, int T, int* maxit) {
    const int mode = fused_mode();
    if (!mode || T % 4 != 0 || B <= 0 || (long)B * C * T >= 0xffffffffL) return 0;
    // (slabs of 11 or 12 positions per thread -- 2.8 / 2.5 rounds of resident workgroups instead of 3.1 -- measure the
    // same 115 us stand-alone and the same step: profiles/r5_ab_notes.md)
    const long TV = T / 4, cap = 256L * FUSED_MAXIT;
    if (maxit) *maxit = FUSED_MAXIT;
    if (TV > cap) return 0;
    for (int n = bm_bwd_nsplit(B); n <= NSPLIT_MAX && n <= B; ++n) {
        const long nb = ((long)B + n - 1) / n;          // the largest slab of `n` near-equal splits
        if (nb * TV <= cap) return (long)C * n <= BM_AMAX_WS ? n : 0;
    }
    return 0;
}
extern "C" int bm_act_bn_bwd_fused_covers(int B, int C, int T) { return fused_nsplit(B, C, T, nullptr) > 0; }

// out[c] = sum_split partial[c][split]
__global__ void finalize_channel_sums_kernel(const double* __restrict__ partial, float* __restrict__ out,
                                             int C, int nsplit) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0;
    for (int k = 0; k < nsplit; ++k) s += partial[(long)c * nsplit + k];
    out[c] = (float)s;
}

extern "C" int bm_bwd_nsplit(int B) { return B >= 64 ? 8 : (B >= 8 ? 4 : 1); }

// The backward kernels evaluate the GELU derivative with one exponential and no erf (bm_gelu_grad_fast, bm_common.h);
// the apply pass walks the batch in the OPPOSITE order of the reduce pass (what that pass read last is what the
// memory-side cache still holds; -0.03 ms per step, profiles/r3r_ab_bn_apply_reverse.txt).
static int gelu_grad_code(int act) { return act == BM_ACT_GELU ? BM_ACT_GELU_FASTGRAD : act; }

// workspace: doubles, (2*C*nsplit) for the reduce partials + (C*nsplit) for the dy sums.
// (sized for the largest split count of the one-pass kernel)
extern "C" long bm_act_bn_bwd_workspace_bytes(int B, int C) {
    (void)B;
    return (long)3 * C * NSPLIT_MAX * sizeof(double);
}

extern "C" int bm_act_bn_bwd(const float* dout, const float* y, const float* scale,
                             const float* shift, const float* mean, const float* invstd,
                             int bn_train, float* dy, float* dgamma, float* dbeta, float* dbias,
                             void* workspace, long workspace_bytes, int B, int C, int T, int act,
                             float leak, float* amax_out, float* amax_ws, float* amax_rows_out, void* stream) {
    BM_REQUIRE(dout && y && dy, "act_bn_bwd: null pointer");
    BM_REQUIRE(!amax_out || amax_ws, "act_bn_bwd: amax_out needs the amax workspace");
    BM_REQUIRE(!amax_out || (long)C * bm_bwd_nsplit(B) <= BM_AMAX_WS, "act_bn_bwd: too many channels for the amax workspace");
    BM_REQUIRE(!bn_train || (scale && shift && mean && invstd), "act_bn_bwd: train BN needs saved statistics");
    if ((long)B * C * T == 0) return BM_OK;
    const bool reduce = bn_train || (scale && (dgamma || dbeta));
    int maxit = 0;
    const int nfused = (reduce && scale && shift && mean && invstd) ? fused_nsplit(B, C, T, &maxit) : 0;
    const int nsplit = nfused ? nfused : bm_bwd_nsplit(B);
    if (workspace_bytes < bm_act_bn_bwd_workspace_bytes(B, C))
        return bm_set_error(BM_ERR_WORKSPACE, "act_bn_bwd: workspace too small (%ld < %ld)", workspace_bytes,
                            bm_act_bn_bwd_workspace_bytes(B, C));
    double* partial = (double*)workspace;
    double* dy_partial = partial + (long)2 * C * nsplit;
    hipStream_t s = (hipStream_t)stream;
    const BmAmaxDst amax_dst = bm_amax_dst(amax_out, amax_ws);
    const dim3 grid(C, nsplit);
    act = gelu_grad_code(act);
    if (nfused) {
        // one pass: the slab stays in registers between the sums and the apply (see bn_bwd_fused_kernel)
        hipError_t e = hipMemsetAsync(partial, 0xff, (size_t)2 * C * nsplit * sizeof(double), s);
        if (e != hipSuccess) return bm_set_error((int)e, "act_bn_bwd: hipMemsetAsync: %s", hipGetErrorString(e));
        const dim3 fgrid((unsigned)C * (unsigned)nsplit);
        hipLaunchKernelGGL(bn_bwd_fused_kernel<FUSED_MAXIT>, fgrid, dim3(256), 0, s, dout, y, scale, shift, mean, invstd,
                           (unsigned long long*)partial, bn_train, dy, dy_partial, dgamma, dbeta, B, C, T, nsplit, act,
                           leak, amax_dst, bm_fastdiv((unsigned)(T / 4)), bm_fastdiv((unsigned)nsplit), g_fused_poll_limit);
    } else if (T % 4 == 0) {
        if (reduce)
            hipLaunchKernelGGL(bn_bwd_reduce_kernel<4>, grid, dim3(256), 0, s, dout, y, scale, shift, mean,
                               invstd, partial, B, C, T, act, leak);
        hipLaunchKernelGGL(bn_bwd_apply_kernel<4>, grid, dim3(256), 0, s, dout, y, scale, shift, mean,
                           invstd, partial, bn_train, dy, dy_partial, dgamma, dbeta, B, C, T, act, leak,
                           amax_dst, 1);
    } else {
        if (reduce)
            hipLaunchKernelGGL(bn_bwd_reduce_kernel<1>, grid, dim3(256), 0, s, dout, y, scale, shift, mean,
                               invstd, partial, B, C, T, act, leak);
        hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, grid, dim3(256), 0, s, dout, y, scale, shift, mean,
                           invstd, partial, bn_train, dy, dy_partial, dgamma, dbeta, B, C, T, act, leak,
                           amax_dst, 1);
    }
    // the (channel, split) grid's partials are per channel already: ONE launch folds them into the tensor slot, the
    // per-channel maxima and the bias gradient (two-stage mode; otherwise the sums get their own launch)
    if (amax_dst.ws && amax_out && dbias) {
        if (int rc = bm_check_launch("act_bn_bwd")) return rc;
        return bm_amax_finalize_rows_sums(amax_dst.ws, C, nsplit, amax_out, amax_rows_out, dy_partial, dbias, s);
    }
    if (dbias)
        hipLaunchKernelGGL(finalize_channel_sums_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, dy_partial,
                           dbias, C, nsplit);
    if (int rc = bm_check_launch("act_bn_bwd")) return rc;
    if (amax_dst.ws) return bm_amax_finalize_rows(amax_dst.ws, C, nsplit, amax_out, amax_rows_out, s);
    return bm_amax_done(amax_dst, C * nsplit, amax_out, s);
}

// ------------------------------------------------------------------------------------------------
// per-channel sum over (B,T): bias gradients of plain convs.
// ------------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ x, long bstride,
                                                          double* __restrict__ partial, int B, int C,
                                                          int T) {
    __shared__ double sh[4];
    const int c = blockIdx.x, split = blockIdx.y, nsplit = gridDim.y;
    const int b0 = (int)((long)B * split / nsplit), b1 = (int)((long)B * (split + 1) / nsplit);
    const int TV = T / VEC;
    float s = 0.f;
    const int nb = b1 - b0;
    for (SlabWalk w(TV, threadIdx.x, blockDim.x); w.bl < nb; w.next()) {
        const int bl = w.bl, tv = w.tv;
        const Pack<VEC> v = Pack<VEC>::ld(x + (long)(b0 + bl) * bstride + (long)c * T + (long)tv * VEC);
#pragma unroll
        for (int i = 0; i < VEC; ++i) s += v.v[i];
    }
    double acc[1] = {(double)s};
    block_sum<1>(acc, sh);
    if (threadIdx.x == 0) partial[(long)c * nsplit + split] = acc[0];
}

extern "C" long bm_channel_sum_workspace_bytes(int B, int C) {
    return (long)C * bm_bwd_nsplit(B) * sizeof(double);
}

extern "C" int bm_channel_sum(const float* x, long bstride, float* out, void* workspace,
                              long workspace_bytes, int B, int C, int T, void* stream) {
    BM_REQUIRE(x && out, "channel_sum: null pointer");
    const int nsplit = bm_bwd_nsplit(B);
    if (workspace_bytes < bm_channel_sum_workspace_bytes(B, C))
        return bm_set_error(BM_ERR_WORKSPACE, "channel_sum: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    double* partial = (double*)workspace;
    if (T % 4 == 0 && bstride % 4 == 0)
        hipLaunchKernelGGL(channel_sum_kernel<4>, dim3(C, nsplit), dim3(256), 0, s, x, bstride, partial, B, C, T);
    else
        hipLaunchKernelGGL(channel_sum_kernel<1>, dim3(C, nsplit), dim3(256), 0, s, x, bstride, partial, B, C, T);
    hipLaunchKernelGGL(finalize_channel_sums_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, partial, out, C,
                       nsplit);
    return bm_check_launch("channel_sum");
}

// ------------------------------------------------------------------------------------------------
// out[c][b] = sum_t x[b][c][t]: the per-segment time sums of a gradient, transposed so that a per-group
// (subject / layout) segmented sum over b runs along the contiguous axis (bias term of the composed front end,
// functional.FusedFrontEndFn).  One wavefront per (b, c) row, fixed summation order.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void time_sums_t_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                          int B, int C, int T) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)B * C) return;
    const int lane = threadIdx.x & 63;
    const float* p = x + row * T;
    float s = 0.f;
    for (int t = lane; t < T; t += 64) s += p[t];
    s = bm_wave_sum(s);
    if (lane == 0) {
        const int b = (int)(row / C), c = (int)(row - (long)b * C);
        out[(long)c * B + b] = s;
    }
}

extern "C" int bm_time_sums_t(const float* x, float* out, int B, int C, int T, void* stream) {
    BM_REQUIRE(x && out && B >= 0 && C > 0 && T > 0, "time_sums_t: bad arguments");
    if (B == 0) return BM_OK;
    hipLaunchKernelGGL(time_sums_t_kernel, dim3((unsigned)(((long)B * C + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       x, out, B, C, T);
    return bm_check_launch("time_sums_t");
}

// ------------------------------------------------------------------------------------------------
// per-channel (sum, sumsq) partials of the conv output for BatchNorm1d (train mode), one streaming
// pass; written in the [nsplit][C][2] layout that bn_finalize folds.  (Cheaper than reducing the
// MFMA accumulators across lanes in the conv epilogue: 24 us vs ~70 us per 320-channel layer.)
// ------------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void channel_stats_kernel(const float* __restrict__ x,
                                                            float* __restrict__ stats, int B, int C,
                                                            int T) {
    __shared__ double sh[4 * 2];
    const int c = blockIdx.x, split = blockIdx.y, nsplit = gridDim.y;
    const int b0 = (int)((long)B * split / nsplit), b1 = (int)((long)B * (split + 1) / nsplit);
    const int TV = T / VEC;
    float s = 0.f, s2 = 0.f;
    const int nb = b1 - b0;
    for (SlabWalk w(TV, threadIdx.x, blockDim.x); w.bl < nb; w.next()) {
        const int bl = w.bl, tv = w.tv;
        const Pack<VEC> v = Pack<VEC>::ld(x + ((long)(b0 + bl) * C + c) * T + (long)tv * VEC);
#pragma unroll
        for (int i = 0; i < VEC; ++i) { s += v.v[i]; s2 += v.v[i] * v.v[i]; }
    }
    double acc[2] = {(double)s, (double)s2};
    block_sum<2>(acc, sh);
    if (threadIdx.x == 0) {
        stats[((long)split * C + c) * 2 + 0] = (float)acc[0];
        stats[((long)split * C + c) * 2 + 1] = (float)acc[1];
    }
}

extern "C" int bm_channel_stats_splits(int B) { return B >= 64 ? 16 : (B >= 8 ? 4 : 1); }

extern "C" int bm_channel_stats(const float* x, float* stats, int B, int C, int T, void* stream) {
    BM_REQUIRE(x && stats, "channel_stats: null pointer");
    const int nsplit = bm_channel_stats_splits(B);
    hipStream_t s = (hipStream_t)stream;
    if (T % 4 == 0)
        hipLaunchKernelGGL(channel_stats_kernel<4>, dim3(C, nsplit), dim3(256), 0, s, x, stats, B, C, T);
    else
        hipLaunchKernelGGL(channel_stats_kernel<1>, dim3(C, nsplit), dim3(256), 0, s, x, stats, B, C, T);
    return bm_check_launch("channel_stats");
}

// ------------------------------------------------------------------------------------------------
// GLU(dim=1): out[b][h][t] = u[b][h][t] * sigmoid(u[b][H+h][t])      (bm/models/common.py:135)
// ------------------------------------------------------------------------------------------------
template <int VEC>
__global__ void glu_fwd_kernel(const float* __restrict__ u, float* __restrict__ out, long nvec, int H,
                               int T, BmAmaxDst amax_ws, BmFastDiv div_per) {
    __shared__ float amax_sh[4];
    float amx = 0.f;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < nvec;
         e += (long)gridDim.x * blockDim.x) {
        const long b = bm_div((unsigned)e, div_per);                // nvec < 2^32 (host check); per = H * TV
        const long r = e - b * (long)div_per.d;
        const float* ua = u + (b * 2 * H) * T + r * VEC;
        const Pack<VEC> a = Pack<VEC>::ld(ua), g = Pack<VEC>::ld(ua + (long)H * T);
        Pack<VEC> o;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            o.v[i] = a.v[i] / (1.f + expf(-g.v[i]));
            amx = fmaxf(amx, fabsf(o.v[i]));
        }
        o.st(out + e * VEC);
    }
    bm_publish_amax(amx, amax_ws, amax_sh);
}

extern "C" int bm_glu_fwd(const float* u, float* out, int B, int H, int T, float* amax_out, float* amax_ws,
                          void* stream) {
    BM_REQUIRE(u && out, "glu_fwd: null pointer");
    BM_REQUIRE(!amax_out || amax_ws, "glu_fwd: amax_out needs the amax workspace");
    const long n = (long)B * H * T;
    if (n == 0) return BM_OK;
    BM_REQUIRE(n < 0xffffffffL && (long)H * T < 0xffffffffL, "glu_fwd: tensor of %ld elements (32-bit element index)", n);
    hipStream_t s = (hipStream_t)stream;
    const BmAmaxDst amax_dst = bm_amax_dst(amax_out, amax_ws);
    int nblk = 0;
    if (T % 4 == 0) {
        const long nvec = n / 4;
        const int blocks = (int)((nvec + 255) / 256 > 16384 ? 16384 : (nvec + 255) / 256);
        hipLaunchKernelGGL(glu_fwd_kernel<4>, dim3(blocks), dim3(256), 0, s, u, out, nvec, H, T, amax_dst,
                           bm_fastdiv((unsigned)((long)H * (T / 4))));
        nblk = blocks;
    } else {
        const int blocks = (int)((n + 255) / 256 > 16384 ? 16384 : (n + 255) / 256);
        hipLaunchKernelGGL(glu_fwd_kernel<1>, dim3(blocks), dim3(256), 0, s, u, out, n, H, T, amax_dst,
                           bm_fastdiv((unsigned)((long)H * T)));
        nblk = blocks;
    }
    if (int rc = bm_check_launch("glu_fwd")) return rc;
    return bm_amax_done(amax_dst, nblk, amax_out, s);
}

// du_a = dout * sig(g) ; du_g = dout * a * sig(g) * (1 - sig(g)); also per-channel sums (bias grad).
template <int VEC>
__global__ __launch_bounds__(256) void glu_bwd_kernel(const float* __restrict__ dout,
                                                      const float* __restrict__ u, float* __restrict__ du,
                                                      double* __restrict__ partial, int B, int H, int T,
                                                      BmAmaxDst amax_ws) {
    __shared__ double sh[4 * 2];
    __shared__ float amax_sh[4];
    float amx = 0.f, amx_g = 0.f;                      // max |du| of the value row hch and of the gate row H + hch
    const int hch = blockIdx.x, split = blockIdx.y, nsplit = gridDim.y;
    const int b0 = (int)((long)B * split / nsplit), b1 = (int)((long)B * (split + 1) / nsplit);
    const int TV = T / VEC;
    float sa = 0.f, sg = 0.f;
    const int nb = b1 - b0;
    SlabWalk w(TV, threadIdx.x, blockDim.x);
    while (w.bl < nb) {
        long offa[BWD_UNROLL];
        bool ok[BWD_UNROLL];
        Pack<VEC> d[BWD_UNROLL], a[BWD_UNROLL], g[BWD_UNROLL];
        long offo[BWD_UNROLL];
#pragma unroll
        for (int q = 0; q < BWD_UNROLL; ++q) {
            ok[q] = w.bl < nb;
            const int b = b0 + w.bl;
            offa[q] = ok[q] ? ((long)b * 2 * H + hch) * T + (long)w.tv * VEC : offa[0];
            offo[q] = ok[q] ? ((long)b * H + hch) * T + (long)w.tv * VEC : offo[0];
            d[q] = Pack<VEC>::ld(dout + offo[q]);
            a[q] = Pack<VEC>::ld(u + offa[q]);
            g[q] = Pack<VEC>::ld(u + offa[q] + (long)H * T);
            if (ok[q]) w.next();
        }
#pragma unroll
        for (int q = 0; q < BWD_UNROLL; ++q) {
            if (!ok[q]) continue;
            Pack<VEC> oa, og;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float sgm = 1.f / (1.f + expf(-g[q].v[i]));
                oa.v[i] = d[q].v[i] * sgm;
                og.v[i] = d[q].v[i] * a[q].v[i] * sgm * (1.f - sgm);
                sa += oa.v[i];
                sg += og.v[i];
                amx = fmaxf(amx, fabsf(oa.v[i]));
                amx_g = fmaxf(amx_g, fabsf(og.v[i]));
            }
            oa.st(du + offa[q]);
            og.st(du + offa[q] + (long)H * T);
        }
    }
    if (amax_ws.ws) {
        // two-stage mode: one partial per ROW of du, laid out [nsplit][2 H] (per-channel maxima, see bm_glu_bwd)
        bm_publish_amax_at(amx, amax_ws, amax_sh, (unsigned)(split * 2 * H + hch));
        __syncthreads();
        bm_publish_amax_at(amx_g, amax_ws, amax_sh, (unsigned)(split * 2 * H + H + hch));
    } else {
        bm_publish_amax(fmaxf(amx, amx_g), amax_ws, amax_sh);
    }
    double acc[2] = {(double)sa, (double)sg};
    block_sum<2>(acc, sh);
    if (threadIdx.x == 0 && partial) {
        partial[(long)hch * nsplit + split] = acc[0];
        partial[(long)(H + hch) * nsplit + split] = acc[1];
    }
}

extern "C" long bm_glu_bwd_workspace_bytes(int B, int H) {
    return (long)2 * H * bm_bwd_nsplit(B) * sizeof(double);
}

extern "C" int bm_glu_bwd(const float* dout, const float* u, float* du, float* dbias, void* workspace,
                          long workspace_bytes, int B, int H, int T, float* amax_out, float* amax_ws,
                          float* amax_rows_out, void* stream) {
    BM_REQUIRE(dout && u && du, "glu_bwd: null pointer");
    BM_REQUIRE(!amax_out || amax_ws, "glu_bwd: amax_out needs the amax workspace");
    BM_REQUIRE(!amax_out || (long)2 * H * bm_bwd_nsplit(B) <= BM_AMAX_WS, "glu_bwd: too many channels for the amax workspace");
    if ((long)B * H * T == 0) return BM_OK;
    const int nsplit = bm_bwd_nsplit(B);
    if (workspace_bytes < bm_glu_bwd_workspace_bytes(B, H))
        return bm_set_error(BM_ERR_WORKSPACE, "glu_bwd: workspace too small");
    double* partial = (double*)workspace;
    hipStream_t s = (hipStream_t)stream;
    const BmAmaxDst amax_dst = bm_amax_dst(amax_out, amax_ws);
    if (T % 4 == 0)
        hipLaunchKernelGGL(glu_bwd_kernel<4>, dim3(H, nsplit), dim3(256), 0, s, dout, u, du, partial, B, H, T,
                           amax_dst);
    else
        hipLaunchKernelGGL(glu_bwd_kernel<1>, dim3(H, nsplit), dim3(256), 0, s, dout, u, du, partial, B, H, T,
                           amax_dst);
    if (amax_dst.ws && amax_out && dbias) {
        if (int rc = bm_check_launch("glu_bwd")) return rc;
        return bm_amax_finalize_rows_sums(amax_dst.ws, 2 * H, nsplit, amax_out, amax_rows_out, partial, dbias, s);
    }
    if (dbias)
        hipLaunchKernelGGL(finalize_channel_sums_kernel, dim3(cdiv(2 * H, 256)), dim3(256), 0, s, partial,
                           dbias, 2 * H, nsplit);
    if (int rc = bm_check_launch("glu_bwd")) return rc;
    if (amax_dst.ws) return bm_amax_finalize_rows(amax_dst.ws, 2 * H, nsplit, amax_out, amax_rows_out, s);
    return bm_amax_done(amax_dst, H * nsplit, amax_out, s);
}
