// Wide-tile variant of the fp32-accurate ("f32x3") implicit-GEMM conv of conv_nn_x3.hip for the large
// layers of the stack (M a multiple of 320 output channels): ONE workgroup of four wavefronts per CU, one
// wavefront per SIMD, each wavefront holding a 160 x 96 output block as 5 x 3 MFMA accumulators (240
// accumulation registers), workgroup tile 320 x 192 (two time tiles cover a 360-sample segment).
//
// Why: with 5 x 1 blocks per wavefront (conv_nn_x3.hip) every MFMA needs 0.6 ds_read_b128 and the LDS
// pipe is ~75 % busy at matrix-core peak; with 5 x 3 blocks it is 0.27 reads per MFMA.  The weight slab of
// the next (16-channel chunk, tap) stage is copied global -> LDS by the DMA path (global_load_lds_dwordx4,
// no staging registers, no ds_write) while the 90 MFMAs of the current stage run; the input window of the
// next chunk travels through registers (it has to be split into the three bf16 planes).
//
// Packed weights are those of bm_pack_weights_x3 ([g][chunk32][tap][plane][4 groups][Mpad] 16-byte slots);
// a stage uses two of the four 8-channel groups.  LDS: A [2 buffers][plane][2 groups][320 rows],
// X [2 buffers][plane][2 groups][224 columns], 102 KB.
#include <cstdlib>
#include "conv_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define WBM 320           // rows of the workgroup tile (2 wavefront rows x 5 blocks)
#define WBN 192           // columns (2 wavefront columns x 3 blocks)
#define WXWP 256          // padded x-window width (>= 192 + 2 * 16; one column per thread)
#define WASLAB (3 * 2 * WBM)      // 16-byte slots of one A buffer
#define WXSLAB (3 * 2 * WXWP)     // 16-byte slots of one X buffer

__device__ __forceinline__ void split8w(const float* f, u32x4& hi, u32x4& mid, u32x4& lo) {
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

#ifdef WIDE_NO_X
#define WIDE_XLOAD(OFF_) (float)((OFF_) & 3)
#else
#define WIDE_XLOAD(OFF_) __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, (OFF_), 0, 0))
#endif
#ifdef WIDE_NO_DMA
#define WIDE_COPY(SRC_, DST_) (DST_)[lane] = *(SRC_);
#else
#define WIDE_COPY(SRC_, DST_)                                                                     \
    __builtin_amdgcn_global_load_lds((const void*)(SRC_), (__attribute__((address_space(3))) void*)(DST_), 16, 0, 0);
#endif

template <int KS>
__global__ __launch_bounds__(256, 1) void conv_nn_x3w_kernel(ConvNNArgs a) {
    constexpr int MW = 5, NW = 3;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4* As = reinterpret_cast<u32x4*>(smem);       // [2][3][2][WBM]
    u32x4* Xs = As + 2 * WASLAB;                      // [2][3][2][WXWP]
#ifdef WIDE_PROFILE
    const long long pentry = clock64();
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nl = lane & 31;
    const int h = lane >> 5;

    int id = bm_xcd_remap(blockIdx.x, gridDim.x);
    const int mtile = id % a.ntiles_m;
    id /= a.ntiles_m;
    const int ntile = id % a.ntiles_n;
    const int b = id / a.ntiles_n;
    const int m0 = mtile * WBM;
    const int n0 = ntile * WBN;
    const int halo = (KS >> 1) * a.dil;
    const int XW = WBN + 2 * halo;                    // <= 224

    const int g = a.widx ? a.widx[b] : 0;
    const int nchunk32 = a.nchunk;
    const int n16 = (a.Cin + 15) >> 4;
    const int nstage = n16 * KS;
    // packed weights in 16-byte slots: [g][chunk32][tap][plane][4][Mpad]
    const u32x4* wg = reinterpret_cast<const u32x4*>(a.wp) + (long)g * nchunk32 * KS * 12 * a.Mpad + m0 + lane;

    // input window of this segment through a bounds-checked buffer descriptor: channels past Cin read 0
    const unsigned long long xaddr = (unsigned long long)(a.x + (long)b * a.x_bstride);
    const unsigned xlo = __builtin_amdgcn_readfirstlane((unsigned)xaddr);          // (readfirstlane returns int:
    const unsigned xhi = __builtin_amdgcn_readfirstlane((unsigned)(xaddr >> 32));  //  keep the halves unsigned)
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(((unsigned long long)xhi << 32) | xlo), 0, __builtin_amdgcn_readfirstlane(a.Cin * a.T * 4), 0x00020000);
    // thread `tid` stages window column tid (both 8-channel groups); columns outside [0, T) or past the
    // window get an offset that stays out of range for every channel -> they read as 0 (conv zero padding)
    const int tcol = n0 - halo + tid;
    const int xoff0 = (tid < XW && tcol >= 0 && tcol < a.T) ? tcol * 4 : 0x40000000;
    const int crow = a.T * 4;

    f32x16 acc[MW][NW];
#pragma unroll
    for (int i = 0; i < MW; ++i)
#pragma unroll
        for (int k = 0; k < NW; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][k][r] = 0.f;

    float xreg[16];

    // DMA of the A slab of stage S_ into buffer (S_ & 1): 30 copies of 64 rows x 16 B, wavefront w
    // issues copies w, w + 4, ...
#define DMA_A(S_)                                                                                 \
    {                                                                                             \
        const int c16 = (S_) / KS, jj = (S_) - c16 * KS;                                          \
        const u32x4* src = wg + ((long)((c16 >> 1) * KS + jj) * 12 + (c16 & 1) * 2) * a.Mpad;     \
        u32x4* dstb = As + ((S_) & 1) * WASLAB;                                                   \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                           \
            int k = wave + 4 * i;                                                                 \
            k = k < 30 ? k : k - 4;      /* wavefronts 2, 3 repeat their last copy: no branch */  \
            const int run = k / 5, rb = k - run * 5;            /* run = plane * 2 + group */     \
            const int plane = run >> 1, kg = run & 1;                                             \
            WIDE_COPY((src + (long)(plane * 4 + kg) * a.Mpad + rb * 64), (dstb + run * WBM + rb * 64)) \
        }                                                                                         \
    }
#define LOAD_X(C16_)                                                                              \
    {                                                                                             \
        const int cb = (C16_) * 16 * crow + xoff0;                                                \
        _Pragma("unroll") for (int r = 0; r < 16; ++r)                                            \
            xreg[r] = WIDE_XLOAD(cb + r * crow);                                                   \
    }
#define STORE_X(BUF_)                                                                             \
    {                                                                                             \
        u32x4* xd = Xs + (BUF_) * WXSLAB + tid;                                                   \
        _Pragma("unroll") for (int kg = 0; kg < 2; ++kg) {                                        \
            u32x4 hi, mid, lo;                                                                    \
            split8w(xreg + 8 * kg, hi, mid, lo);                                                  \
            xd[(0 * 2 + kg) * WXWP] = hi;                                                         \
            xd[(1 * 2 + kg) * WXWP] = mid;                                                        \
            xd[(2 * 2 + kg) * WXWP] = lo;                                                         \
        }                                                                                         \
    }

    DMA_A(0);
    LOAD_X(0);
    STORE_X(0);
    __syncthreads();

#ifdef WIDE_PROFILE
    long long pt[5] = {0, 0, 0, 0, 0};
#define PSTAMP(I_) { const long long now_ = clock64(); pt[I_] += now_ - plast; plast = now_; }
    long long plast = clock64();
    const long long pc0 = plast, pw0 = wall_clock64();
#else
#define PSTAMP(I_)
#endif
    int s = 0;
    for (int c16 = 0; c16 < n16; ++c16) {
#pragma unroll
        for (int j = 0; j < KS; ++j, ++s) {
            // next stage's weight slab by DMA first: it has the whole stage to land (the last stage re-copies
            // its own slab: no branches inside a stage)
#ifndef WIDE_EXP_NODMA
            DMA_A(s + 1 < nstage ? s + 1 : s);
#endif
            __builtin_amdgcn_sched_barrier(0);
            // this stage's operand fragments, in the order the partial products consume them
            const u32x4* ab = As + (s & 1) * WASLAB + h * WBM + wm * (MW * 32) + nl;
            const u32x4* xb = Xs + (c16 & 1) * WXSLAB + h * WXWP + wn * (NW * 32) + nl + j * a.dil;
            bf16x8 af[3][MW], bf[3][NW];
#define FRAGS(PA_, PB_)                                                                           \
    _Pragma("unroll") for (int nt = 0; nt < NW; ++nt)                                             \
        bf[PB_][nt] = __builtin_bit_cast(bf16x8, xb[(PB_) * 2 * WXWP + nt * 32]);                 \
    _Pragma("unroll") for (int mt = 0; mt < MW; ++mt)                                             \
        af[PA_][mt] = __builtin_bit_cast(bf16x8, ab[(PA_) * 2 * WBM + mt * 32]);
            FRAGS(2, 0) FRAGS(0, 2) FRAGS(1, 1)
#undef FRAGS
#ifndef WIDE_EXP_NOX
            if (j == 0) LOAD_X(c16 + 1);   // next chunk's input window into registers (zeros past the last chunk)
#endif
            PSTAMP(0)
#ifdef WIDE_PROFILE
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            PSTAMP(1)
#ifndef WIDE_EXP_NOX
            if (j == KS - 1) STORE_X((c16 + 1) & 1);
#endif
            // six partial products, smallest first (planes: 0 = hi, 1 = mid, 2 = lo)
#define TERM(PA_, PB_)                                                                            \
    _Pragma("unroll") for (int mt = 0; mt < MW; ++mt)                                             \
        _Pragma("unroll") for (int nt = 0; nt < NW; ++nt)                                         \
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA_][mt], bf[PB_][nt], acc[mt][nt], 0, 0, 0);
            TERM(2, 0) TERM(0, 2) TERM(1, 1) TERM(1, 0) TERM(0, 1) TERM(0, 0)
#undef TERM
            PSTAMP(2)
            PSTAMP(3)
#ifndef WIDE_EXP_NOBARRIER
            __syncthreads();      // next A slab landed (vmcnt) and visible, this stage's LDS reads are done
#endif
            PSTAMP(4)
        }
    }
#undef DMA_A
#undef LOAD_X
#undef STORE_X

#ifdef WIDE_PROFILE
    if (a.stats && blockIdx.x == 0 && lane == 0) {
        for (int i = 0; i < 5; ++i) a.stats[wave * 8 + i] = (float)pt[i] / (float)nstage;
        a.stats[wave * 8 + 5] = (float)(clock64() - pc0);
        a.stats[wave * 8 + 6] = (float)(wall_clock64() - pw0);
        a.stats[wave * 8 + 7] = (float)(pc0 - pentry);
    }
#endif
    // epilogue: bias, optional pre-activation store, per-channel affine, activation, residual.
    // One literal-indexed expansion per accumulator block keeps the 240 accumulators in registers.
    float* ep = smem;                // the operand buffers are free after the last barrier of the main loop
    conv_ep_stage_params(a, ep, WBM, m0, tid, 256);
    __syncthreads();
#define EPI(MT_, NT_)                                                                             \
    {                                                                                             \
        float v_[16];                                                                             \
        conv_ep_store_block(a, acc[MT_][NT_], ep, WBM, b, m0, wm * (MW * 32) + (MT_) * 32 + 4 * h, \
                            n0 + wn * (NW * 32) + (NT_) * 32 + nl, v_);                           \
    }
#define EPI_ROW(MT_) EPI(MT_, 0) EPI(MT_, 1) EPI(MT_, 2)
#ifdef WIDE_NO_EPI
    if (a.y_pre && tid == 0 && blockIdx.x == 0) a.y_pre[0] = acc[0][0][0] + acc[4][2][15] + acc[2][1][7];
#else
    EPI_ROW(0) EPI_ROW(1) EPI_ROW(2) EPI_ROW(3) EPI_ROW(4)
#endif
#undef EPI_ROW
#undef EPI
#ifdef WIDE_PROFILE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.stats && blockIdx.x == 0 && lane == 0) a.stats[32 + wave] = (float)(clock64() - pentry);
#endif
}

template <int KS>
static int launch_conv_nn_x3w(ConvNNArgs a, hipStream_t stream) {
    const size_t lds = (size_t)(2 * WASLAB + 2 * WXSLAB) * 16;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_nn_x3w_kernel<KS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return bm_set_error((int)e, "conv_nn_x3w: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    a.ntiles_n = cdiv(a.T, WBN);
    a.ntiles_m = a.Mpad / WBM;
    const long nblocks = (long)a.B * a.ntiles_n * a.ntiles_m;
    if (nblocks <= 0) return BM_OK;
    hipLaunchKernelGGL((conv_nn_x3w_kernel<KS>), dim3((unsigned)nblocks), dim3(256), lds, stream, a);
    return bm_check_launch("conv_nn_x3w");
}

static int x3w_enabled() {
    static int wide = -1;
    if (wide < 0) {
        const char* e = getenv("BM_X3_WIDE");      // BM_X3_WIDE=0: A/B runs against conv_nn_x3_kernel
        wide = !(e && e[0] == '0');
    }
    return wide;
}

static bool x3w_covers(int Cin, int Mpad, int T, int KS, int dil, bool with_stats) {
#ifndef WIDE_PROFILE
    if (with_stats) return false;
#endif
    if (!x3w_enabled() || Mpad % WBM != 0 || (KS != 1 && KS != 3)) return false;
    if ((KS >> 1) * dil > 16 || T <= 128) return false;
    return (long)Cin * T * 4 < 0x40000000L;
}

extern "C" int bm_conv_bf16_mpad(int M);
extern "C" int bm_conv_x3_is_wide(int Cin, int M, int T, int KS, int dil, int with_stats) {
    return x3w_covers(Cin, bm_conv_bf16_mpad(M), T, KS, dil, with_stats != 0) ? 1 : 0;
}

// Eligibility + launch of the wide-tile kernel; returns -1 when the shape is not one it covers (the caller
// then takes conv_nn_x3_kernel).  `a` comes fully populated from bm_conv1d_nn_x3.
int bm_conv_nn_x3w_try(const ConvNNArgs& a, hipStream_t stream) {
    if (!x3w_covers(a.Cin, a.Mpad, a.T, a.KS, a.dil, a.stats != nullptr)) return -1;
    return a.KS == 1 ? launch_conv_nn_x3w<1>(a, stream) : launch_conv_nn_x3w<3>(a, stream);
}
