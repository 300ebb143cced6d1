// Implicit-GEMM 1-D convolution ("NN" form: weights x activations, time contiguous) on the CDNA4
// matrix cores with exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
//   y[b][m][t] = ep( bias[m] + sum_{c,j} W[g(b)][m][c][j] * x[b][c][t + (j - KS/2)*dil] )
//
// One kernel covers every "weights x [C,T] window" contraction of the SimpleConv hot path:
//   * nn.Conv1d k=3 dilated "same" convs of ConvSequence        (bm/models/common.py:113-114)
//   * the GLU convs and all 1x1 convs / ConvTranspose1d(k=1)     (common.py:133-138, simpleconv.py:113-120,185-189)
//   * SubjectLayers' per-subject matmul (weight picked by widx)  (common.py:55-58)
//   * ChannelMerger's weighted sensor reduction                  (common.py:358)
//   * every data-gradient of the above (weights packed transposed+flipped by pack.hip)
//   * ClipLoss' dEstimate = dScores x candidates                  (losses.py:94, backward)
//
// Tiling (MI355X-first, wave64): a workgroup is 4 wavefronts and owns a [BM = 32*MT] x [BN = 128]
// output tile of ONE segment b; wavefront w owns the 32-column strip w and all MT 32-row blocks
// (MT f32x16 accumulators, 1 B-operand + MT A-operand ds_read_b32 per MT MFMAs).  The input
// window x[b][c0:c0+16][n0-halo : n0+128+halo] is staged ONCE in LDS and re-used by all KS taps
// (the taps are just column offsets j*dil into the same LDS rows), weights arrive pre-packed as
// [chunk][tap][16 channels][Mpad] so that the A slab of a (chunk, tap) stage is a straight,
// 16-byte-vectorised copy and both MFMA operands are conflict-free consecutive-lane LDS reads.
// The K loop is software-pipelined through registers with double-buffered LDS (one barrier per
// stage): global loads of stage s+1 are issued before the 8*MT MFMAs of stage s.
#include "conv_common.h"


template <int MT, bool VEC>
__global__ __launch_bounds__(256, 3) void conv_nn_kernel(ConvNNArgs a) {
    constexpr int BM = 32 * MT;
    constexpr int BN = 128;
    constexpr int BKC = BM_BKC;
    constexpr int Q = BM / 4;                         // float4 per A row
    constexpr int AREG = (BKC * Q + 255) / 256;       // float4 per thread per (chunk, tap) A slab
    constexpr int XROWS = BKC / 4;                    // X rows per wavefront
    constexpr int XCOLS = 3;                          // 64-lane column passes: window <= 192 floats
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int nl = lane & 31;
    const int h = lane >> 5;

    // logical block id -> (b, ntile, mtile); mtile fastest so that blocks sharing an x window
    // are neighbours on the same XCD.
    const int nblocks = gridDim.x;
    int id = bm_xcd_remap(blockIdx.x, nblocks);
    const int mtile = id % a.ntiles_m;
    id /= a.ntiles_m;
    const int ntile = id % a.ntiles_n;
    const int b = id / a.ntiles_n;

    const int m0 = mtile * BM;
    const int n0 = ntile * BN;
    const int halo = (a.KS >> 1) * a.dil;
    // staged x window: [n0 - HP, n0 + BN + HP).  VEC (T % 4 == 0, 16-byte aligned rows): HP = halo
    // rounded up to a multiple of 4 so that every global access is a dwordx4 lying entirely inside
    // or outside [0, T); otherwise HP = halo and the window is filled with dword loads.
    const int HP = VEC ? ((halo + 3) & ~3) : halo;
    const int XW = BN + 2 * HP;            // staged window width (<= 192)
    const int xoff = HP - halo;            // column of tap 0, output column 0 inside the window
    // LDS: two A slabs [BKC][BM] (one per pipeline stage parity) + two X windows [BKC][XW]
    float* As = smem;
    float* Xs = smem + 2 * BKC * BM;

    const int g = a.widx ? a.widx[b] : 0;
    const float* xb = a.x + (long)b * a.x_bstride;
    const float* wg = a.wp + (long)g * a.nchunk * a.KS * BKC * a.Mpad + m0;

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    // Software pipeline over stages s = chunk*KS + tap: while the MFMAs of stage s run from LDS,
    // the A slab of stage s+1 (and, during the last tap of a chunk, the x window of chunk+1) is in
    // flight from HBM/L2 into registers; it is written to the other LDS buffer after the MFMAs,
    // one barrier per stage.
    static_assert(AREG <= 3 && XROWS == 4 && XCOLS == 3, "staging register set");
    float4 a0 = make_float4(0, 0, 0, 0), a1 = a0, a2 = a0;
    float x00 = 0, x01 = 0, x02 = 0, x10 = 0, x11 = 0, x12 = 0, x20 = 0, x21 = 0, x22 = 0, x30 = 0,
          x31 = 0, x32 = 0;
    float4 xv0 = a0, xv1 = a0, xv2 = a0, xv3 = a0;
    const int nstage = a.nchunk * a.KS;

// NB: the staging registers are NAMED scalars handled by macros: arrays (even with fully unrolled
// constant indices, or captured by lambdas) were left in scratch memory by hipcc's SROA.
#define BM_A1(I_, V_)                                                                             \
    if (I_ < AREG) {                                                                              \
        int e = tid + I_ * 256;                                                                   \
        e = e < BKC * Q ? e : BKC * Q - 1;          /* clamp: unconditional load, guarded store */ \
        const int r = e / Q, q = e - r * Q;                                                       \
        V_ = *reinterpret_cast<const float4*>(wsrc + (long)r * a.Mpad + q * 4);                   \
    }
#define BM_LOAD_A(S_)                                                                             \
    {                                                                                             \
        const float* wsrc = wg + (long)(S_) * BKC * a.Mpad;                                       \
        BM_A1(0, a0) BM_A1(1, a1) BM_A1(2, a2)                                                    \
    }
#define BM_SA1(I_, V_)                                                                            \
    if (I_ < AREG && tid + I_ * 256 < BKC * Q)                                                    \
        *reinterpret_cast<float4*>(dst + (tid + I_ * 256) * 4) = V_;
#define BM_STORE_A(BUF_)                                                                          \
    {                                                                                             \
        float* dst = As + (BUF_) * BKC * BM;                                                      \
        BM_SA1(0, a0) BM_SA1(1, a1) BM_SA1(2, a2)                                                 \
    }
#define BM_X1(RR_, K_, V_)                                                                        \
    {                                                                                             \
        const int c = c0 + wave + RR_ * 4;                                                        \
        const int xx = lane + K_ * 64;                                                            \
        const int t = n0 - HP + xx;                                                               \
        V_ = (xx < XW && c < a.Cin && t >= 0 && t < a.T) ? xb[(long)c * a.T + t] : 0.f;           \
    }
#define BM_XV1(RR_, V_)                                                                           \
    {                                                                                             \
        const int c = c0 + wave + RR_ * 4;                                                        \
        const int t = n0 - HP + 4 * lane;                                                         \
        V_ = make_float4(0.f, 0.f, 0.f, 0.f);                                                     \
        if (4 * lane < XW && c < a.Cin && t >= 0 && t < a.T)                                      \
            V_ = *reinterpret_cast<const float4*>(xb + (long)c * a.T + t);                        \
    }
#define BM_LOAD_X(CHUNK_)                                                                         \
    {                                                                                             \
        const int c0 = (CHUNK_) * BKC;                                                            \
        if constexpr (VEC) {                                                                      \
            BM_XV1(0, xv0) BM_XV1(1, xv1) BM_XV1(2, xv2) BM_XV1(3, xv3)                           \
        } else {                                                                                  \
        BM_X1(0, 0, x00) BM_X1(0, 1, x01) BM_X1(0, 2, x02) BM_X1(1, 0, x10) BM_X1(1, 1, x11)      \
        BM_X1(1, 2, x12) BM_X1(2, 0, x20) BM_X1(2, 1, x21) BM_X1(2, 2, x22) BM_X1(3, 0, x30)      \
        BM_X1(3, 1, x31) BM_X1(3, 2, x32)                                                         \
        }                                                                                         \
    }
#define BM_SX1(RR_, K_, V_)                                                                       \
    if (lane + K_ * 64 < XW) dst[(wave + RR_ * 4) * XW + lane + K_ * 64] = V_;
#define BM_STORE_X(BUF_)                                                                          \
    {                                                                                             \
        float* dst = Xs + (BUF_) * BKC * XW;                                                      \
        if constexpr (VEC) {                                                                      \
            if (4 * lane < XW) {                                                                  \
                *reinterpret_cast<float4*>(dst + (wave + 0) * XW + 4 * lane) = xv0;               \
                *reinterpret_cast<float4*>(dst + (wave + 4) * XW + 4 * lane) = xv1;               \
                *reinterpret_cast<float4*>(dst + (wave + 8) * XW + 4 * lane) = xv2;               \
                *reinterpret_cast<float4*>(dst + (wave + 12) * XW + 4 * lane) = xv3;              \
            }                                                                                     \
        } else {                                                                                  \
        BM_SX1(0, 0, x00) BM_SX1(0, 1, x01) BM_SX1(0, 2, x02) BM_SX1(1, 0, x10) BM_SX1(1, 1, x11) \
        BM_SX1(1, 2, x12) BM_SX1(2, 0, x20) BM_SX1(2, 1, x21) BM_SX1(2, 2, x22) BM_SX1(3, 0, x30) \
        BM_SX1(3, 1, x31) BM_SX1(3, 2, x32)                                                       \
        }                                                                                         \
    }

    BM_LOAD_A(0);
    BM_LOAD_X(0);
    BM_STORE_A(0);
    BM_STORE_X(0);
    __syncthreads();
    int s = 0;
    for (int chunk = 0; chunk < a.nchunk; ++chunk) {
        const float* xbuf = Xs + (chunk & 1) * BKC * XW;
        for (int j = 0; j < a.KS; ++j, ++s) {
            const bool more_a = s + 1 < nstage;
            const bool more_x = j == a.KS - 1 && chunk + 1 < a.nchunk;
            if (more_a) BM_LOAD_A(s + 1);
            if (more_x) BM_LOAD_X(chunk + 1);
            // ---- MFMA: k runs over channel pairs; lanes 0-31 feed k even, 32-63 k odd ----
            const float* xrow = xbuf + h * XW + wave * 32 + nl + xoff + j * a.dil;
            const float* arow = As + (s & 1) * BKC * BM + h * BM + nl;
#pragma unroll
            for (int p = 0; p < BKC / 2; ++p) {
                const float bv = xrow[2 * p * XW];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float av = arow[2 * p * BM + mt * 32];
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mt], 0, 0, 0);
                }
            }
            if (more_a) BM_STORE_A((s + 1) & 1);
            if (more_x) BM_STORE_X((chunk + 1) & 1);
            __syncthreads();
        }
    }
#undef BM_LOAD_A
#undef BM_A1
#undef BM_SA1
#undef BM_X1
#undef BM_XV1
#undef BM_SX1
#undef BM_STORE_A
#undef BM_LOAD_X
#undef BM_STORE_X

    conv_tile_epilogue<MT>(a, acc, smem, b, ntile, m0, n0, tid);
}

template <int MT, bool VEC>
static int launch_conv_nn_v(const ConvNNArgs& a, hipStream_t stream) {
    constexpr int BM = 32 * MT;
    const int halo = (a.KS >> 1) * a.dil;
    const int HP = VEC ? ((halo + 3) & ~3) : halo;
    const int XW = 128 + 2 * HP;
    if (XW > 192)
        return bm_set_error(BM_ERR_UNSUPPORTED, "conv_nn: (kernel_size/2)*dilation = %d exceeds the 32-sample halo of the staged window", halo);
    size_t lds = (size_t)(2 * BM_BKC * BM + 2 * BM_BKC * XW) * sizeof(float);
    const size_t lds_red = (size_t)(4 * BM * 2 + 3 * BM) * sizeof(float);   // epilogue scratch
    if (lds < lds_red) lds = lds_red;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_nn_kernel<MT, VEC>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return bm_set_error((int)e, "conv_nn: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    const long nblocks = (long)a.B * a.ntiles_n * a.ntiles_m;
    if (nblocks <= 0) return BM_OK;
    hipLaunchKernelGGL((conv_nn_kernel<MT, VEC>), dim3((unsigned)nblocks), dim3(256), lds, stream, a);
    return bm_check_launch("conv_nn");
}

template <int MT>
static int launch_conv_nn(const ConvNNArgs& a, hipStream_t stream) {
    // the 5-block tile is register-bound at 3 waves/SIMD: its x window stays on the 12-dword staging
    // set (the 4 x dwordx4 set spills); smaller tiles take the vector path.
    const bool vec = MT < 5 && (a.T % 4 == 0) && (a.x_bstride % 4 == 0) && ((uintptr_t)a.x % 16 == 0);
    if constexpr (MT < 5)
        if (vec) return launch_conv_nn_v<MT, true>(a, stream);
    return launch_conv_nn_v<MT, false>(a, stream);
}

// Row-tile height (in 32-row MFMA blocks) that minimises padded rows for M output channels.
extern "C" int bm_conv_mt_for(int M) {
    int best = 1;
    long best_cost = -1;
    for (int mt = 1; mt <= 5; ++mt) {
        const long bm = 32L * mt;
        const long padded = (long)cdiv(M, bm) * bm;
        // prefer less padding; among equals prefer the taller tile (fewer x re-reads)
        const long cost = padded * 16 - mt;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = mt; }
    }
    return best;
}

extern "C" int bm_conv_mpad(int M) {
    const int mt = bm_conv_mt_for(M);
    return cdiv(M, 32 * mt) * 32 * mt;
}

extern "C" int bm_conv_stats_tiles(int B, int T) { return B * cdiv(T, 128); }

// C-ABI: replaces F.conv1d / einsum call sites listed in the header comment.
extern "C" int bm_conv1d_nn(const float* x, long x_bstride, const float* wpacked, const int* widx,
                            const float* bias, long bias_gstride, const float* ep_scale, const float* ep_shift,
                            const float* res, long res_bstride, float* y_pre, float* y_out,
                            long y_bstride, float* stats, int B, int Cin, int M, int T, int KS,
                            int dil, int act, float leak, void* stream) {
    BM_REQUIRE(x && wpacked, "conv1d_nn: null x/w");
    BM_REQUIRE(y_pre || y_out, "conv1d_nn: no output");
    BM_REQUIRE(KS >= 1 && (KS & 1), "conv1d_nn: kernel size must be odd (got %d)", KS);
    BM_REQUIRE(B >= 0 && Cin > 0 && M > 0 && T > 0 && dil >= 1, "conv1d_nn: bad dims");
    BM_REQUIRE((ep_scale == nullptr) == (ep_shift == nullptr), "conv1d_nn: scale/shift must come together");
    ConvNNArgs a;
    a.x = x; a.x_bstride = x_bstride; a.wp = wpacked; a.widx = widx; a.bias = bias; a.bias_gstride = bias_gstride;
    a.ep_scale = ep_scale; a.ep_shift = ep_shift; a.res = res; a.res_bstride = res_bstride;
    a.y_pre = y_pre; a.y_out = y_out; a.y_bstride = y_bstride; a.stats = stats;
    a.B = B; a.Cin = Cin; a.M = M; a.T = T; a.KS = KS; a.dil = dil; a.act = act; a.leak = leak;
    const int mt = bm_conv_mt_for(M);
    a.Mpad = bm_conv_mpad(M);
    a.nchunk = cdiv(Cin, BM_BKC);
    a.ntiles_n = cdiv(T, 128);
    a.ntiles_m = a.Mpad / (32 * mt);
    hipStream_t s = (hipStream_t)stream;
    switch (mt) {
        case 1: return launch_conv_nn<1>(a, s);
        case 2: return launch_conv_nn<2>(a, s);
        case 3: return launch_conv_nn<3>(a, s);
        case 4: return launch_conv_nn<4>(a, s);
        default: return launch_conv_nn<5>(a, s);
    }
}
