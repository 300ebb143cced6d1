// bf16-input / fp32-accumulate variant of the implicit-GEMM conv (conv_nn.hip) on the CDNA4 matrix
// cores: v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate).  OPT-IN compute mode
// (brainmagick_amd.set_compute_dtype("bf16")): activations and outputs stay fp32 in HBM, the MFMA
// operands are rounded to bf16 (RNE) while they are staged into LDS, accumulation is fp32.  With the
// matrix pipe 16x faster the kernel is bound by HBM/L2 traffic instead of the fp32 matrix rate
// (DESIGN.md §2); its results differ from the fp32 reference by ~2^-9 per product (tolerance 1e-2,
// tests/test_bf16_gpu.py) -- the parity-green default stays the exact-fp32 kernel.
//
// Layouts (K of one MFMA = 16 channels = two groups of 8; lane l feeds group l>>5, row/col l&31):
//   packed weights  [g][chunk of 64 ch][tap][8 groups][Mpad][8 ch]  bf16   (bm_pack_weights_bf16)
//   LDS A slab      [8 groups][BM] x 16 B      -> ds_read_b128, consecutive lanes consecutive slots
//   LDS x window    [8 groups][XW] x 16 B      -> tap j of output column n = slot n + j*dil
// Staging: a thread loads the SAME time sample of 8 consecutive channels (8 coalesced dword loads),
// converts and writes one 16-byte slot.  The K loop is software-pipelined exactly like the fp32
// kernel: stage = (chunk, tap); the A slab of the next stage is in flight during the MFMAs of a
// stage, the whole x window of the next chunk is requested at the chunk's first stage and 1/KS of it
// is converted and written to the other LDS buffer after each stage (flight time = up to KS stages).
#include "conv_common.h"

// Measurement-only knob: issue every MFMA BM_MFMA_REP times (results are garbage for REP > 1); used
// to price the matrix-pipe load of multi-pass bf16 emulation schemes.  Always 1 in the shipped build.
#ifndef BM_MFMA_REP
#define BM_MFMA_REP 1
#endif

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define BKC2 64
#define KG (BKC2 / 8)

template <int N> struct FVecB { typedef float type __attribute__((ext_vector_type(N))); };
template <int N> struct UVecB { typedef unsigned int type __attribute__((ext_vector_type(N))); };

__device__ __forceinline__ u32x4 pack8_bf16(float f0, float f1, float f2, float f3, float f4, float f5,
                                            float f6, float f7) {
    bf16x8 b;
    b[0] = (__bf16)f0; b[1] = (__bf16)f1; b[2] = (__bf16)f2; b[3] = (__bf16)f3;
    b[4] = (__bf16)f4; b[5] = (__bf16)f5; b[6] = (__bf16)f6; b[7] = (__bf16)f7;
    return __builtin_bit_cast(u32x4, b);
}

template <int MT, int KS>
__global__ __launch_bounds__(256, 2) void conv_nn_bf16_kernel(ConvNNArgs a) {
    constexpr int BM = 32 * MT;
    constexpr int BN = 128;
    constexpr int AIT = (KG * BM + 255) / 256;        // 16-byte A slots per thread per stage
    constexpr int NIT = 6;                            // x items per thread per chunk (2 groups x 3 column passes)
    constexpr int IPS = (NIT + KS - 1) / KS;          // x items staged per stage
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4* As = reinterpret_cast<u32x4*>(smem);                     // [2][KG][BM]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int nl = lane & 31;
    const int h = lane >> 5;

    const int nblocks = gridDim.x;
    int id = bm_xcd_remap(blockIdx.x, nblocks);
    const int mtile = id % a.ntiles_m;
    id /= a.ntiles_m;
    const int ntile = id % a.ntiles_n;
    const int b = id / a.ntiles_n;
    const int m0 = mtile * BM;
    const int n0 = ntile * BN;
    const int halo = (KS >> 1) * a.dil;
    const int XW = BN + 2 * halo;                     // <= 192
    u32x4* Xs = As + 2 * KG * BM;                     // [2][KG][XW]

    const int g = a.widx ? a.widx[b] : 0;
    const float* xb = a.x + (long)b * a.x_bstride;
    // packed bf16 weights, in 16-byte slots: [g][chunk][tap][KG][Mpad]
    const u32x4* wg = reinterpret_cast<const u32x4*>(a.wp) + (long)g * a.nchunk * KS * KG * a.Mpad + m0;

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    typename UVecB<4 * AIT>::type areg0, areg1;       // A slab of stage s lives in set s & 1, two stages in flight
    typename FVecB<8 * NIT>::type xreg;               // whole next-chunk window, in flight for KS stages

    // A slab of stage S_ (= chunk*KS + tap): KG runs of BM slots
#define LOAD_A(AREG_, S_)                                                                              \
    {                                                                                             \
        const u32x4* wsrc = wg + (long)(S_) * KG * a.Mpad;                                        \
        _Pragma("unroll") for (int i = 0; i < AIT; ++i) {                                         \
            int e = tid + i * 256;                                                                \
            e = e < KG * BM ? e : KG * BM - 1;                                                    \
            const int kg = e / BM, mm = e - kg * BM;                                              \
            const u32x4 v = wsrc[(long)kg * a.Mpad + mm];                                         \
            AREG_[4 * i] = v[0]; AREG_[4 * i + 1] = v[1]; AREG_[4 * i + 2] = v[2]; AREG_[4 * i + 3] = v[3]; \
        }                                                                                         \
    }
#define STORE_A(AREG_, BUF_)                                                                             \
    {                                                                                             \
        u32x4* dst = As + (BUF_) * KG * BM;                                                       \
        _Pragma("unroll") for (int i = 0; i < AIT; ++i) {                                         \
            const int e = tid + i * 256;                                                          \
            if (e < KG * BM) {                                                                    \
                u32x4 v; v[0] = AREG_[4 * i]; v[1] = AREG_[4 * i + 1]; v[2] = AREG_[4 * i + 2]; v[3] = AREG_[4 * i + 3]; \
                dst[e] = v;                                                                       \
            }                                                                                     \
        }                                                                                         \
    }
    // x items: wavefront w stages channel groups 2w and 2w+1; item it in [0,6): group 2w + it/3,
    // columns lane + 64*(it%3).  Part PART_ of a chunk = items PART_, PART_+KS, ...
#define LOAD_X(CHUNK_)                                                                            \
    {                                                                                             \
        _Pragma("unroll") for (int it = 0; it < NIT; ++it) {                                      \
            const int kg = 2 * wave + it / 3;                                                     \
            const int xx = lane + 64 * (it % 3);                                                  \
            const int t = n0 - halo + xx;                                                         \
            const int cbase = (CHUNK_) * BKC2 + kg * 8;                                           \
            const bool tok = xx < XW && t >= 0 && t < a.T;                                        \
            const float* col = xb + (long)cbase * a.T + t;                                        \
            _Pragma("unroll") for (int r = 0; r < 8; ++r)                                         \
                xreg[8 * it + r] = (tok && cbase + r < a.Cin) ? col[(long)r * a.T] : 0.f;         \
        }                                                                                         \
    }
#define STORE_X(BUF_, PART_)                                                                      \
    {                                                                                             \
        u32x4* dst = Xs + (BUF_) * KG * XW;                                                       \
        _Pragma("unroll") for (int u = 0; u < IPS; ++u) {                                         \
            const int it = (PART_) + u * KS;                                                      \
            if (it < NIT) {                                                                       \
                const int kg = 2 * wave + it / 3;                                                 \
                const int xx = lane + 64 * (it % 3);                                              \
                if (xx < XW)                                                                      \
                    dst[kg * XW + xx] = pack8_bf16(xreg[8 * it], xreg[8 * it + 1], xreg[8 * it + 2], \
                                                   xreg[8 * it + 3], xreg[8 * it + 4],            \
                                                   xreg[8 * it + 5], xreg[8 * it + 6],            \
                                                   xreg[8 * it + 7]);                             \
            }                                                                                     \
        }                                                                                         \
    }

    // prologue: A slab of stage 0 -> LDS, A slab of stage 1 in flight (set 1), window of chunk 0
    const int nstage = a.nchunk * KS;
    LOAD_A(areg0, 0);
    STORE_A(areg0, 0);
    if (nstage > 1) LOAD_A(areg1, 1);
    LOAD_X(0);
#pragma unroll
    for (int part = 0; part < KS; ++part) STORE_X(0, part);
    __syncthreads();

    // One pipeline stage s = chunk*KS + tap.  A(s+2) is requested into the register set that held
    // A(s) (already in LDS), A(s+1) -- requested one stage ago -- is written to the other LDS buffer
    // after the MFMAs: two stages of flight time for the weight slabs.
#define STAGE(CHUNK_, J_, AL_, AS_)                                                               \
    {                                                                                             \
        const int s_ = (CHUNK_) * KS + (J_);                                                      \
        const bool more_x_ = (CHUNK_) + 1 < a.nchunk;                                             \
        if (s_ + 2 < nstage) LOAD_A(AL_, s_ + 2);                                                 \
        if ((J_) == 0 && more_x_) LOAD_X((CHUNK_) + 1);                                           \
        const u32x4* abuf = As + (s_ & 1) * KG * BM;                                              \
        const u32x4* xcol = Xs + ((CHUNK_) & 1) * KG * XW + wave * 32 + nl + (J_) * a.dil;        \
        _Pragma("unroll 1") for (int ks = 0; ks < KG / 2; ++ks) {                                 \
            const bf16x8 bv = __builtin_bit_cast(bf16x8, xcol[(2 * ks + h) * XW]);                \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                   \
                const bf16x8 av = __builtin_bit_cast(bf16x8, abuf[(2 * ks + h) * BM + mt * 32 + nl]); \
                _Pragma("unroll") for (int rep_ = 0; rep_ < BM_MFMA_REP; ++rep_)                  \
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[mt], 0, 0, 0);  \
            }                                                                                     \
        }                                                                                         \
        if (s_ + 1 < nstage) STORE_A(AS_, (s_ + 1) & 1);                                          \
        if (more_x_) STORE_X(((CHUNK_) + 1) & 1, (J_));                                           \
        __syncthreads();                                                                          \
    }
    // the register-set roles alternate with the parity of s; KS is odd, so they flip every chunk:
    // even chunks start with (load -> set 0, store <- set 1), odd chunks with the opposite.
#define CHUNK_EVEN(C_)                                                                            \
    {                                                                                             \
        STAGE(C_, 0, areg0, areg1)                                                                \
        if constexpr (KS >= 3) { STAGE(C_, 1, areg1, areg0) STAGE(C_, 2, areg0, areg1) }          \
        if constexpr (KS >= 5) { STAGE(C_, 3, areg1, areg0) STAGE(C_, 4, areg0, areg1) }          \
    }
#define CHUNK_ODD(C_)                                                                             \
    {                                                                                             \
        STAGE(C_, 0, areg1, areg0)                                                                \
        if constexpr (KS >= 3) { STAGE(C_, 1, areg0, areg1) STAGE(C_, 2, areg1, areg0) }          \
        if constexpr (KS >= 5) { STAGE(C_, 3, areg0, areg1) STAGE(C_, 4, areg1, areg0) }          \
    }
    int chunk = 0;
    for (; chunk + 1 < a.nchunk; chunk += 2) {
        CHUNK_EVEN(chunk)
        CHUNK_ODD(chunk + 1)
    }
    if (chunk < a.nchunk) CHUNK_EVEN(chunk)
#undef STAGE
#undef CHUNK_EVEN
#undef CHUNK_ODD
#undef LOAD_A
#undef STORE_A
#undef LOAD_X
#undef STORE_X
    conv_tile_epilogue<MT>(a, acc, smem, b, ntile, m0, n0, tid);
}

template <int MT, int KS>
static int launch_conv_nn_bf16(const ConvNNArgs& a, hipStream_t stream) {
    constexpr int BM = 32 * MT;
    const int halo = (KS >> 1) * a.dil;
    const int XW = 128 + 2 * halo;
    if (XW > 192)
        return bm_set_error(BM_ERR_UNSUPPORTED, "conv_nn_bf16: (kernel_size/2)*dilation = %d exceeds the 32-sample halo", halo);
    size_t lds = (size_t)(2 * KG * BM + 2 * KG * XW) * 16;
    const size_t lds_red = (size_t)(4 * BM * 2 + 3 * BM) * sizeof(float);   // epilogue scratch
    if (lds < lds_red) lds = lds_red;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_nn_bf16_kernel<MT, KS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return bm_set_error((int)e, "conv_nn_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    const long nblocks = (long)a.B * a.ntiles_n * a.ntiles_m;
    if (nblocks <= 0) return BM_OK;
    hipLaunchKernelGGL((conv_nn_bf16_kernel<MT, KS>), dim3((unsigned)nblocks), dim3(256), lds, stream, a);
    return bm_check_launch("conv_nn_bf16");
}

// Tile heights available on the bf16 path: 3, 4 or 5 MFMA row blocks.
extern "C" int bm_conv_bf16_mt_for(int M) {
    int best = 3;
    long best_cost = -1;
    for (int mt = 3; mt <= 5; ++mt) {
        const long bm = 32L * mt;
        const long cost = (long)cdiv(M, bm) * bm * 16 - mt;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = mt; }
    }
    return best;
}
extern "C" int bm_conv_bf16_mpad(int M) {
    const int mt = bm_conv_bf16_mt_for(M);
    return cdiv(M, 32 * mt) * 32 * mt;
}
// number of bf16 elements of the packed weight buffer
extern "C" long bm_packed_weight_elems_bf16(int G, int M, int Cin, int KS) {
    return (long)G * cdiv(Cin, BKC2) * KS * KG * bm_conv_bf16_mpad(M) * 8;
}

// dst[g][chunk][tap][kg][Mpad][8] (bf16) <- alpha * src[g*sg + m*sm + c*sc + tap'*sj], c = chunk*64 + kg*8 + r
__global__ void pack_weights_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                         int G, int M, int Cin, int KS, long sg, long sm, long sc, long sj,
                                         int flip, int Mpad, int nchunk, const float* alpha_ptr) {
    const long total = (long)G * nchunk * KS * KG * Mpad * 8;
    const float alpha = alpha_ptr ? *alpha_ptr : 1.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i;
        const int e8 = (int)(r % 8); r /= 8;
        const int m = (int)(r % Mpad); r /= Mpad;
        const int kg = (int)(r % KG); r /= KG;
        const int j = (int)(r % KS); r /= KS;
        const int chunk = (int)(r % nchunk);
        const int g = (int)(r / nchunk);
        const int c = chunk * BKC2 + kg * 8 + e8;
        float v = 0.f;
        if (m < M && c < Cin) {
            const int jj = flip ? KS - 1 - j : j;
            v = alpha * src[g * sg + m * sm + c * sc + jj * sj];
        }
        const __bf16 bv = (__bf16)v;
        dst[i] = __builtin_bit_cast(unsigned short, bv);
    }
}

extern "C" int bm_pack_weights_bf16(const float* src, void* dst, int G, int M, int Cin, int KS, long sg,
                                    long sm, long sc, long sj, int flip, const float* alpha_ptr,
                                    void* stream) {
    BM_REQUIRE(src && dst, "pack_weights_bf16: null pointer");
    BM_REQUIRE(G > 0 && M > 0 && Cin > 0 && KS > 0, "pack_weights_bf16: bad dims");
    const int Mpad = bm_conv_bf16_mpad(M);
    const int nchunk = cdiv(Cin, BKC2);
    const long total = (long)G * nchunk * KS * KG * Mpad * 8;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src,
                       (unsigned short*)dst, G, M, Cin, KS, sg, sm, sc, sj, flip, Mpad, nchunk, alpha_ptr);
    return bm_check_launch("pack_weights_bf16");
}

// Same contract as bm_conv1d_nn, with bf16-rounded MFMA operands (weights packed by
// bm_pack_weights_bf16); x, bias, residual, outputs are fp32.
extern "C" int bm_conv1d_nn_bf16(const float* x, long x_bstride, const void* wpacked, const int* widx,
                                 const float* bias, long bias_gstride, const float* ep_scale, const float* ep_shift,
                                 const float* res, long res_bstride, float* y_pre, float* y_out,
                                 long y_bstride, float* stats, int B, int Cin, int M, int T, int KS,
                                 int dil, int act, float leak, void* stream) {
    BM_REQUIRE(x && wpacked, "conv1d_nn_bf16: null x/w");
    BM_REQUIRE(y_pre || y_out, "conv1d_nn_bf16: no output");
    BM_REQUIRE(KS == 1 || KS == 3 || KS == 5, "conv1d_nn_bf16: kernel size %d not supported (1, 3, 5)", KS);
    BM_REQUIRE(B >= 0 && Cin > 0 && M > 0 && T > 0 && dil >= 1, "conv1d_nn_bf16: bad dims");
    BM_REQUIRE((ep_scale == nullptr) == (ep_shift == nullptr), "conv1d_nn_bf16: scale/shift must come together");
    ConvNNArgs a;
    a.x = x; a.x_bstride = x_bstride; a.wp = (const float*)wpacked; a.widx = widx; a.bias = bias; a.bias_gstride = bias_gstride;
    a.ep_scale = ep_scale; a.ep_shift = ep_shift; a.res = res; a.res_bstride = res_bstride;
    a.y_pre = y_pre; a.y_out = y_out; a.y_bstride = y_bstride; a.stats = stats;
    a.B = B; a.Cin = Cin; a.M = M; a.T = T; a.KS = KS; a.dil = dil; a.act = act; a.leak = leak;
    const int mt = bm_conv_bf16_mt_for(M);
    a.Mpad = bm_conv_bf16_mpad(M);
    a.nchunk = cdiv(Cin, BKC2);
    a.ntiles_n = cdiv(T, 128);
    a.ntiles_m = a.Mpad / (32 * mt);
    hipStream_t s = (hipStream_t)stream;
#define DISPATCH_KS(MT_)                                                    \
    switch (KS) {                                                           \
        case 1: return launch_conv_nn_bf16<MT_, 1>(a, s);                   \
        case 3: return launch_conv_nn_bf16<MT_, 3>(a, s);                   \
        default: return launch_conv_nn_bf16<MT_, 5>(a, s);                  \
    }
    switch (mt) {
        case 3: DISPATCH_KS(3)
        case 4: DISPATCH_KS(4)
        default: DISPATCH_KS(5)
    }
#undef DISPATCH_KS
}
