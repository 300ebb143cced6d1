// fp32-ACCURATE implicit-GEMM conv on the bf16 matrix cores ("bf16x3" split emulation).
//
// Every fp32 operand is split EXACTLY into three bf16 numbers  x = hi + mid + lo  (8+8+8 mantissa
// bits); a product a*b is evaluated as the six largest of the nine partial products
//     a.lo*b.hi + a.hi*b.lo + a.mid*b.mid + a.mid*b.hi + a.hi*b.mid + a.hi*b.hi
// (the three dropped ones are <= 2^-26 |a*b|), each an exact bf16 x bf16 product accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16.  Six bf16 MFMAs cost 6/16 of one exact-fp32 MFMA (v_mfma_f32_32x32x2_f32)
// for the same 32x32x16 block, and the result is fp32-class: measured rel-L2 vs fp64 1.4e-7 against
// 3.4e-7 for a plain fp32 FMA chain (tests/test_exact_f32_gpu.py::test_x3_error_is_fp32_class).  This is the compute mode "f32x3"
// (brainmagick_amd.set_compute_dtype("f32x3")); activations, parameters, gradients and every
// elementwise kernel stay fp32.
//
// Three operand planes:
//   packed weights  [g][chunk of 32 ch][tap][plane][4 groups][Mpad][8 ch] bf16  (bm_pack_weights_x3)
//   LDS A slab      [plane][4 groups][BM] x 16 B,   LDS x window  [plane][4 groups][XW] x 16 B
// stage = (chunk, tap): 2 k16-steps x 6 terms x MT MFMAs.  LDS is single-buffered (two barriers per
// stage) so that two workgroups fit a CU; the next A slab and the next chunk's x window travel through
// registers while the MFMAs run.
#include <cstdlib>
#include "conv_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define XC 32             // channels per chunk
#define XG (XC / 8)       // 8-channel groups per chunk

template <int N> struct FVecX { typedef float type __attribute__((ext_vector_type(N))); };
template <int N> struct UVecX { typedef unsigned int type __attribute__((ext_vector_type(N))); };

// exact 3-way split of 8 fp32 values into bf16 planes (hi, mid, lo)
__device__ __forceinline__ void split8(const float* f, u32x4& hi, u32x4& mid, u32x4& lo) {
    bf16x8 h, m, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 a = (__bf16)f[i];
        const float r1 = f[i] - (float)a;
        const __bf16 b = (__bf16)r1;
        const float r2 = r1 - (float)b;
        h[i] = a; m[i] = b; l[i] = (__bf16)r2;
    }
    hi = __builtin_bit_cast(u32x4, h);
    mid = __builtin_bit_cast(u32x4, m);
    lo = __builtin_bit_cast(u32x4, l);
}

template <int MT, int KS>
__global__ __launch_bounds__(256, 2) void conv_nn_x3_kernel(ConvNNArgs a) {
    constexpr int BM = 32 * MT;
    constexpr int BN = 128;
    constexpr int ASLOTS = 3 * XG * BM;               // 16-byte slots of one A slab
    constexpr int AIT = (ASLOTS + 255) / 256;
    constexpr int NIT = 3;                            // x items per thread per chunk (3 column passes)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4* As = reinterpret_cast<u32x4*>(smem);       // [3][XG][BM]
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
    u32x4* Xs = As + ASLOTS;                          // [3][XG][XW]

    const int g = a.widx ? a.widx[b] : 0;
    const float* xb = a.x + (long)b * a.x_bstride;
    // packed weights in 16-byte slots: [g][chunk][tap][plane][XG][Mpad]
    const u32x4* wg = reinterpret_cast<const u32x4*>(a.wp) + (long)g * a.nchunk * KS * 3 * XG * a.Mpad + m0;

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    typename UVecX<4 * AIT>::type areg;
    typename FVecX<8 * NIT>::type xreg;

#define LOAD_A(S_)                                                                                \
    {                                                                                             \
        const u32x4* wsrc = wg + (long)(S_) * 3 * XG * a.Mpad;                                    \
        _Pragma("unroll") for (int i = 0; i < AIT; ++i) {                                         \
            int e = tid + i * 256;                                                                \
            e = e < ASLOTS ? e : ASLOTS - 1;                                                      \
            const int run = e / BM, mm = e - run * BM;          /* run = plane*XG + group */      \
            const u32x4 v = wsrc[(long)run * a.Mpad + mm];                                        \
            areg[4 * i] = v[0]; areg[4 * i + 1] = v[1]; areg[4 * i + 2] = v[2]; areg[4 * i + 3] = v[3]; \
        }                                                                                         \
    }
#define STORE_A()                                                                                 \
    {                                                                                             \
        _Pragma("unroll") for (int i = 0; i < AIT; ++i) {                                         \
            const int e = tid + i * 256;                                                          \
            if (e < ASLOTS) {                                                                     \
                u32x4 v; v[0] = areg[4 * i]; v[1] = areg[4 * i + 1]; v[2] = areg[4 * i + 2]; v[3] = areg[4 * i + 3]; \
                As[e] = v;                                                                        \
            }                                                                                     \
        }                                                                                         \
    }
    // x items: wavefront w stages channel group w; item it: columns lane + 64*it
#define LOAD_X(CHUNK_)                                                                            \
    {                                                                                             \
        const int cbase = (CHUNK_) * XC + wave * 8;                                               \
        _Pragma("unroll") for (int it = 0; it < NIT; ++it) {                                      \
            const int xx = lane + 64 * it;                                                        \
            const int t = n0 - halo + xx;                                                         \
            const bool tok = xx < XW && t >= 0 && t < a.T;                                        \
            const float* col = xb + (long)cbase * a.T + t;                                        \
            _Pragma("unroll") for (int r = 0; r < 8; ++r)                                         \
                xreg[8 * it + r] = (tok && cbase + r < a.Cin) ? col[(long)r * a.T] : 0.f;         \
        }                                                                                         \
    }
#define STORE_X()                                                                                 \
    {                                                                                             \
        _Pragma("unroll") for (int it = 0; it < NIT; ++it) {                                      \
            const int xx = lane + 64 * it;                                                        \
            if (xx < XW) {                                                                        \
                float f[8];                                                                       \
                _Pragma("unroll") for (int r = 0; r < 8; ++r) f[r] = xreg[8 * it + r];            \
                u32x4 hi, mid, lo;                                                                \
                split8(f, hi, mid, lo);                                                           \
                Xs[(0 * XG + wave) * XW + xx] = hi;                                               \
                Xs[(1 * XG + wave) * XW + xx] = mid;                                              \
                Xs[(2 * XG + wave) * XW + xx] = lo;                                               \
            }                                                                                     \
        }                                                                                         \
    }

    const int nstage = a.nchunk * KS;
    LOAD_A(0);
    LOAD_X(0);
    STORE_A();
    STORE_X();
    __syncthreads();

    int s = 0;
    for (int chunk = 0; chunk < a.nchunk; ++chunk) {
        const bool more_x = chunk + 1 < a.nchunk;
#pragma unroll
        for (int j = 0; j < KS; ++j, ++s) {
            const bool more_a = s + 1 < nstage;
            if (more_a) LOAD_A(s + 1);
            if (j == 0 && more_x) LOAD_X(chunk + 1);
            const u32x4* xcol = Xs + wave * 32 + nl + j * a.dil;
#pragma unroll 1
            for (int ks = 0; ks < XG / 2; ++ks) {
                const int kg = 2 * ks + h;
                const bf16x8 bh = __builtin_bit_cast(bf16x8, xcol[(0 * XG + kg) * XW]);
                const bf16x8 bm = __builtin_bit_cast(bf16x8, xcol[(1 * XG + kg) * XW]);
                const bf16x8 bl = __builtin_bit_cast(bf16x8, xcol[(2 * XG + kg) * XW]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, As[(0 * XG + kg) * BM + mt * 32 + nl]);
                    const bf16x8 am = __builtin_bit_cast(bf16x8, As[(1 * XG + kg) * BM + mt * 32 + nl]);
                    const bf16x8 al = __builtin_bit_cast(bf16x8, As[(2 * XG + kg) * BM + mt * 32 + nl]);
                    // smallest terms first
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[mt], 0, 0, 0);
                }
            }
            __syncthreads();                          // every wavefront is done with this stage's LDS
            if (more_a) STORE_A();
            if (j == KS - 1 && more_x) STORE_X();
            __syncthreads();
        }
    }
#undef LOAD_A
#undef STORE_A
#undef LOAD_X
#undef STORE_X
    conv_tile_epilogue<MT>(a, acc, smem, b, ntile, m0, n0, tid);
}

template <int MT, int KS>
static int launch_conv_nn_x3(const ConvNNArgs& a, hipStream_t stream) {
    constexpr int BM = 32 * MT;
    const int halo = (KS >> 1) * a.dil;
    const int XW = 128 + 2 * halo;
    if (XW > 192)
        return bm_set_error(BM_ERR_UNSUPPORTED, "conv_nn_x3: (kernel_size/2)*dilation = %d exceeds the 32-sample halo", halo);
    size_t lds = (size_t)(3 * XG * BM + 3 * XG * XW) * 16;
    const size_t lds_red = (size_t)(4 * BM * 2 + 3 * BM) * sizeof(float);   // epilogue scratch
    if (lds < lds_red) lds = lds_red;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_nn_x3_kernel<MT, KS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return bm_set_error((int)e, "conv_nn_x3: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    const long nblocks = (long)a.B * a.ntiles_n * a.ntiles_m;
    if (nblocks <= 0) return BM_OK;
    hipLaunchKernelGGL((conv_nn_x3_kernel<MT, KS>), dim3((unsigned)nblocks), dim3(256), lds, stream, a);
    return bm_check_launch("conv_nn_x3");
}

// Tile heights of the narrow kernels: 3, 4 or 5 MFMA row blocks (96 / 128 / 160 rows), whichever pads M least.
extern "C" int bm_conv_x3_mt_for(int M) {
    int best = 3;
    long best_cost = -1;
    for (int mt = 3; mt <= 5; ++mt) {
        const long bm = 32L * mt;
        const long cost = (long)cdiv(M, bm) * bm * 16 - mt;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = mt; }
    }
    return best;
}
extern "C" int bm_conv_x3_mpad(int M) {
    const int mt = bm_conv_x3_mt_for(M);
    return cdiv(M, 32 * mt) * 32 * mt;
}

// number of bf16 elements of the packed 3-plane weight buffer
extern "C" long bm_packed_weight_elems_x3(int G, int M, int Cin, int KS) {
    return (long)G * cdiv(Cin, XC) * KS * 3 * XG * bm_conv_x3_mpad(M) * 8;
}

// dst[g][chunk][tap][plane][kg][Mpad][8] (bf16) <- exact 3-way split of alpha * src[...]
__global__ void pack_weights_x3_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                       int G, int M, int Cin, int KS, long sg, long sm, long sc, long sj,
                                       int flip, int Mpad, int nchunk, const float* alpha_ptr) {
    const long per_plane = (long)XG * Mpad * 8;
    const long total = (long)G * nchunk * KS * per_plane;          // one thread per fp32 source element
    const float alpha = alpha_ptr ? *alpha_ptr : 1.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long r = i;
        const int e8 = (int)(r % 8); r /= 8;
        const int m = (int)(r % Mpad); r /= Mpad;
        const int kg = (int)(r % XG); r /= XG;
        const int j = (int)(r % KS); r /= KS;
        const int chunk = (int)(r % nchunk);
        const int g = (int)(r / nchunk);
        const int c = chunk * XC + kg * 8 + e8;
        float v = 0.f;
        if (m < M && c < Cin) {
            const int jj = flip ? KS - 1 - j : j;
            v = alpha * src[g * sg + m * sm + c * sc + jj * sj];
        }
        const __bf16 hi = (__bf16)v;
        const float r1 = v - (float)hi;
        const __bf16 mid = (__bf16)r1;
        const __bf16 lo = (__bf16)(r1 - (float)mid);
        const long stage = ((long)g * nchunk + chunk) * KS + j;
        const long within = ((long)kg * Mpad + m) * 8 + e8;
        unsigned short* base = dst + stage * 3 * per_plane + within;
        base[0 * per_plane] = __builtin_bit_cast(unsigned short, hi);
        base[1 * per_plane] = __builtin_bit_cast(unsigned short, mid);
        base[2 * per_plane] = __builtin_bit_cast(unsigned short, lo);
    }
}

extern "C" int bm_pack_weights_x3(const float* src, void* dst, int G, int M, int Cin, int KS, long sg,
                                  long sm, long sc, long sj, int flip, const float* alpha_ptr,
                                  void* stream) {
    BM_REQUIRE(src && dst, "pack_weights_x3: null pointer");
    BM_REQUIRE(G > 0 && M > 0 && Cin > 0 && KS > 0, "pack_weights_x3: bad dims");
    const int Mpad = bm_conv_x3_mpad(M);
    const int nchunk = cdiv(Cin, XC);
    const long total = (long)G * nchunk * KS * XG * Mpad * 8;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_weights_x3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src,
                       (unsigned short*)dst, G, M, Cin, KS, sg, sm, sc, sj, flip, Mpad, nchunk, alpha_ptr);
    return bm_check_launch("pack_weights_x3");
}

// Same contract as bm_conv1d_nn; weights packed by bm_pack_weights_x3; fp32-accurate (see header).
extern "C" int bm_conv1d_nn_x3(const float* x, long x_bstride, const void* wpacked, const int* widx,
                               const float* bias, long bias_gstride, const float* ep_scale, const float* ep_shift,
                               const float* res, long res_bstride, float* y_pre, float* y_out,
                               long y_bstride, float* stats, int B, int Cin, int M, int T, int KS,
                               int dil, int act, float leak, void* stream) {
    BM_REQUIRE(x && wpacked, "conv1d_nn_x3: null x/w");
    BM_REQUIRE(y_pre || y_out, "conv1d_nn_x3: no output");
    BM_REQUIRE(KS == 1 || KS == 3 || KS == 5, "conv1d_nn_x3: kernel size %d not supported (1, 3, 5)", KS);
    BM_REQUIRE(B >= 0 && Cin > 0 && M > 0 && T > 0 && dil >= 1, "conv1d_nn_x3: bad dims");
    BM_REQUIRE((ep_scale == nullptr) == (ep_shift == nullptr), "conv1d_nn_x3: scale/shift must come together");
    ConvNNArgs a;
    a.x = x; a.x_bstride = x_bstride; a.wp = (const float*)wpacked; a.widx = widx; a.bias = bias; a.bias_gstride = bias_gstride;
    a.ep_scale = ep_scale; a.ep_shift = ep_shift; a.res = res; a.res_bstride = res_bstride;
    a.y_pre = y_pre; a.y_out = y_out; a.y_bstride = y_bstride; a.stats = stats;
    a.B = B; a.Cin = Cin; a.M = M; a.T = T; a.KS = KS; a.dil = dil; a.act = act; a.leak = leak;
    const int mt = bm_conv_x3_mt_for(M);
    a.Mpad = bm_conv_x3_mpad(M);
    a.nchunk = cdiv(Cin, XC);
    a.ntiles_n = cdiv(T, 128);
    a.ntiles_m = a.Mpad / (32 * mt);
    hipStream_t s = (hipStream_t)stream;
#define DISPATCH_KS(MT_)                                                  \
    switch (KS) {                                                         \
        case 1: return launch_conv_nn_x3<MT_, 1>(a, s);                   \
        case 3: return launch_conv_nn_x3<MT_, 3>(a, s);                   \
        default: return launch_conv_nn_x3<MT_, 5>(a, s);                  \
    }
    switch (mt) {
        case 3: DISPATCH_KS(3)
        case 4: DISPATCH_KS(4)
        default: DISPATCH_KS(5)
    }
#undef DISPATCH_KS
}
