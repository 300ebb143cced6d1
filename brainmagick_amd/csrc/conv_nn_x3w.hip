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
// a stage uses two of the four 8-channel groups.  LDS: A [3 buffers][plane][2 groups][320 rows],
// X [2 buffers][plane][2 groups][256 columns], 141 KB.  The slab of stage s + 2 is requested at the start
// of stage s; the barrier of a stage sits before its last 30 MFMAs, and the first fragments of the next
// stage are read under them.
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

typedef int i32x4w __attribute__((ext_vector_type(4)));

// The input-window loads of the main loop are issued through inline asm and the barriers are raw s_barrier
// with hand-counted waits: __syncthreads() would drain the VMEM queue (vmcnt(0)), i.e. wait for the weight
// slab that has just been requested two stages ahead.  Rules that keep this safe: no value produced by the
// asm loads is read before the matching CW_WAIT_X; the kernel compiles without spills in the main loop
// (tests/test_host_cpu.py); everything is drained before the epilogue.
__device__ __forceinline__ float cw_ld32(i32x4w rs, int voff) {
    float v;
    asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=&v"(v) : "v"(voff), "s"(rs) : "memory");
    return v;
}

template <int KS>
__global__ __launch_bounds__(256, 1) void conv_nn_x3w_kernel(ConvNNArgs a) {
#ifdef WIDE_PROFILE
    const long long pentry = clock64();
#endif
    constexpr int MW = 5, NW = 3;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4* As = reinterpret_cast<u32x4*>(smem);       // [3 buffers][3 planes][2 groups][WBM]
    u32x4* Xs = As + 3 * WASLAB;                      // [2 buffers][3 planes][2 groups][WXWP]
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
    const int n16 = (a.Cin + 15) >> 4;                // channels past Cin read 0 through the bounds check
    const int nstage = n16 * KS;
    // packed weights in 16-byte slots: [g][chunk32][tap][plane][4][Mpad]
    const u32x4* wg = reinterpret_cast<const u32x4*>(a.wp) + (long)g * nchunk32 * KS * 12 * a.Mpad + m0 + lane;

    // input window of this segment through a bounds-checked buffer descriptor: channels past Cin read 0
    const unsigned long long xaddr = (unsigned long long)(a.x + (long)b * a.x_bstride);
    i32x4w xr;
    xr[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)xaddr);
    xr[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(xaddr >> 32) & 0xffffu));
    xr[2] = __builtin_amdgcn_readfirstlane(a.Cin * a.T * 4);
    xr[3] = 0x00020000;
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
    bf16x8 af[3][MW], bf[3][NW];                      // operand fragments [plane 0 = hi, 1 = mid, 2 = lo]


    // DMA of the A slab of stage S_ (clamped to the last stage) into A buffer S_ % 3
#define DMA_A(S_, BUF3_)                                                                          \
    {                                                                                             \
        const int sc = (S_) < nstage ? (S_) : nstage - 1;                                         \
        const int c16 = sc / KS, jj = sc - c16 * KS;                                              \
        const u32x4* src = wg + ((long)((c16 >> 1) * KS + jj) * 12 + (c16 & 1) * 2) * a.Mpad;     \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                           \
            int k = wave + 4 * i;                                                                 \
            k = k < 30 ? k : k - 4;                                                               \
            const int run = k / 5, rb = k - run * 5;            /* run = plane * 2 + group */     \
            const int plane = run >> 1, kg = run & 1;                                             \
            __builtin_amdgcn_global_load_lds(                                                     \
                (const void*)(src + (long)(plane * 4 + kg) * a.Mpad + rb * 64),                   \
                (__attribute__((address_space(3))) void*)(As + (BUF3_) * WASLAB + run * WBM + rb * 64), 16, 0, 0); \
        }                                                                                         \
    }
    // 16 channels x 1 column of the input window of chunk C16_ (zeros past the last chunk: offset out of range)
#define LOAD_X(C16_)                                                                              \
    {                                                                                             \
        const int cb = (C16_) * 16 * crow + xoff0;                                                \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) xreg[r] = cw_ld32(xr, cb + r * crow);      \
    }
#define CW_WAIT_X(N_)                                                                             \
    asm volatile("s_waitcnt vmcnt(" #N_ ")"                                                       \
                 : "+v"(xreg[0]), "+v"(xreg[1]), "+v"(xreg[2]), "+v"(xreg[3]), "+v"(xreg[4]), "+v"(xreg[5]),  \
                   "+v"(xreg[6]), "+v"(xreg[7]), "+v"(xreg[8]), "+v"(xreg[9]), "+v"(xreg[10]), "+v"(xreg[11]), \
                   "+v"(xreg[12]), "+v"(xreg[13]), "+v"(xreg[14]), "+v"(xreg[15])::"memory");
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
    // fragments a stage needs first (B.hi, A.lo, B.lo); ABUF_ = its A buffer, XB_/J_ = its X buffer and tap
#define FRAGS_EARLY(ABUF_, XB_, J_)                                                               \
    {                                                                                             \
        const u32x4* ab = As + (ABUF_) * WASLAB + h * WBM + wm * (MW * 32) + nl;                  \
        const u32x4* xb = Xs + (XB_) * WXSLAB + h * WXWP + wn * (NW * 32) + nl + (J_) * a.dil;    \
        _Pragma("unroll") for (int nt = 0; nt < NW; ++nt) bf[0][nt] = __builtin_bit_cast(bf16x8, xb[0 * 2 * WXWP + nt * 32]); \
        _Pragma("unroll") for (int mt = 0; mt < MW; ++mt) af[2][mt] = __builtin_bit_cast(bf16x8, ab[2 * 2 * WBM + mt * 32]); \
        _Pragma("unroll") for (int nt = 0; nt < NW; ++nt) bf[2][nt] = __builtin_bit_cast(bf16x8, xb[2 * 2 * WXWP + nt * 32]); \
    }
#define FRAGS_LATE(ABUF_, XB_, J_)                                                                \
    {                                                                                             \
        const u32x4* ab = As + (ABUF_) * WASLAB + h * WBM + wm * (MW * 32) + nl;                  \
        const u32x4* xb = Xs + (XB_) * WXSLAB + h * WXWP + wn * (NW * 32) + nl + (J_) * a.dil;    \
        _Pragma("unroll") for (int mt = 0; mt < MW; ++mt) af[0][mt] = __builtin_bit_cast(bf16x8, ab[0 * 2 * WBM + mt * 32]); \
        _Pragma("unroll") for (int nt = 0; nt < NW; ++nt) bf[1][nt] = __builtin_bit_cast(bf16x8, xb[1 * 2 * WXWP + nt * 32]); \
        _Pragma("unroll") for (int mt = 0; mt < MW; ++mt) af[1][mt] = __builtin_bit_cast(bf16x8, ab[1 * 2 * WBM + mt * 32]); \
    }
#define TERM(PA_, PB_)                                                                            \
    _Pragma("unroll") for (int mt = 0; mt < MW; ++mt)                                             \
        _Pragma("unroll") for (int nt = 0; nt < NW; ++nt)                                         \
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA_][mt], bf[PB_][nt], acc[mt][nt], 0, 0, 0);
    // "slab s + 1 landed" + workgroup barrier: N_ younger VMEM instructions may stay in flight
#define CW_BARRIER(N_)                                                                            \
    asm volatile("s_waitcnt vmcnt(" #N_ ") lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // prologue: slabs 0 and 1, input window of chunk 0
    DMA_A(0, 0)
    DMA_A(1, 1)
    LOAD_X(0)
    CW_WAIT_X(0)
    STORE_X(0)
    if (KS == 1) LOAD_X(1)          // 1x1 convs: the window of chunk c + 2 is requested in stage c (see below)
    CW_BARRIER(0)
    FRAGS_EARLY(0, 0, 0)

    // Stage s = (16-channel chunk c16, tap j), A buffer s % 3, X buffer c16 & 1.  VMEM queue order inside a
    // stage: [input window of chunk c16 + 1, 16 loads, if j == 0], slab s + 2 (8 DMA copies).  Before the
    // barrier slab s + 1 must have landed: younger than it are 8 (+ 16 if j == 0) instructions.  The input
    // window is consumed in the stage with j == KS - 1: younger than it are the KS slabs issued since.
    // KS == 1 (every stage is a window stage): the window of chunk c + 1 was requested in stage c - 1 after its
    // slab; it is split in stage c, and only then is the window of chunk c + 2 requested into the same
    // registers - queue order per stage: slab s + 2 (8), window s + 2 (16); younger than slab s + 1 at the
    // barrier: window s + 1, slab s + 2, window s + 2 = 40.
#ifdef WIDE_PROFILE
    long long pt[3] = {0, 0, 0};
    long long plast = clock64();
    const long long pstart = plast;
#define CW_PSTAMP(I_) { const long long now_ = clock64(); pt[I_] += now_ - plast; plast = now_; }
#else
#define CW_PSTAMP(I_)
#endif
    int s = 0;
    int ab3 = 0;                                      // s % 3
    for (int c16 = 0; c16 < n16; ++c16) {
        const int xbuf = c16 & 1;
#pragma unroll
        for (int j = 0; j < KS; ++j, ++s) {
            const int ab_next = ab3 == 2 ? 0 : ab3 + 1;
            const int ab_next2 = ab_next == 2 ? 0 : ab_next + 1;
            if (KS != 1 && j == 0) LOAD_X(c16 + 1)
            DMA_A(s + 2, ab_next2)
            FRAGS_LATE(ab3, xbuf, j)
            if (j == KS - 1) {
                if (KS == 3) CW_WAIT_X(24) else CW_WAIT_X(8)
                STORE_X(xbuf ^ 1)
                if (KS == 1) LOAD_X(c16 + 2)
            }
            TERM(2, 0) TERM(0, 2) TERM(1, 0) TERM(0, 0)
            if (j == KS - 1) {       // spread the split arithmetic of the input window between the MFMAs
                _Pragma("unroll") for (int g_ = 0; g_ < 45; ++g_) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                }
            }
            CW_PSTAMP(0)
            if (KS == 1) CW_BARRIER(40) else if (j == 0) CW_BARRIER(24) else CW_BARRIER(8)
            CW_PSTAMP(1)
            if (j == KS - 1) FRAGS_EARLY(ab_next, xbuf ^ 1, 0) else FRAGS_EARLY(ab_next, xbuf, j + 1)
            TERM(1, 1) TERM(0, 1)
            CW_PSTAMP(2)
            ab3 = ab_next;
        }
    }
#ifdef WIDE_PROFILE
    if (a.stats && blockIdx.x == 0 && lane == 0) {   // profiling builds only: the stats buffer carries the stamps
        float* d = a.stats + wave * 8;
        for (int i = 0; i < 3; ++i) d[i] = (float)(pt[i] / nstage);
        d[3] = (float)((clock64() - pstart) / nstage);
        d[4] = (float)(pstart - pentry);              // prologue
        d[5] = (float)(clock64() - pstart);           // main loop
    }
    const long long pmain_end = clock64();
#endif
    // drain the (clamped, unused) copies of the last stages before the LDS is re-used by the epilogue
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#undef DMA_A
#undef LOAD_X
#undef CW_WAIT_X
#undef STORE_X
#undef FRAGS_EARLY
#undef FRAGS_LATE
#undef TERM
#undef CW_BARRIER

    // epilogue: bias, optional pre-activation store, per-channel affine, activation, residual.
    // One literal-indexed expansion per accumulator block keeps the 240 accumulators in registers.
    float* ep = smem;                // the operand buffers are free after the last barrier of the main loop
    conv_ep_stage_params(a, ep, WBM, m0, tid, 256, b);
    __syncthreads();
    // Common case (every wide launch of the training step): one output tensor, no affine / activation, tile
    // fully inside M.  Row addresses are wave-uniform bases + one per-lane offset + an immediate per column
    // block, the 80 row biases are read once: ~3 VALU issue slots per stored element instead of ~10.
    const bool simple = ((a.y_pre != nullptr) != (a.y_out != nullptr)) && !a.ep_scale && a.act == BM_ACT_NONE &&
                        m0 + WBM <= a.M;
    if (simple) {
        float* yb = (a.y_pre ? a.y_pre : a.y_out) + (long)b * a.y_bstride;
        const float* rb = (a.y_out && a.res) ? a.res + (long)b * a.res_bstride : nullptr;
        const int rowu = m0 + wm * (MW * 32);                       // wave-uniform first row
        const int li = 4 * h * a.T + n0 + wn * (NW * 32) + nl;      // per-lane element offset inside a row block
        const float* epl = ep + wm * (MW * 32) + 4 * h;
        float bia[MW][16];
#pragma unroll
        for (int mt = 0; mt < MW; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) bia[mt][r] = epl[mt * 32 + (r & 3) + 8 * (r >> 2)];
#define EPS(MT_, NT_)                                                                             \
    {                                                                                             \
        float rv_[16];                                                                            \
        if (rb) {                                                                                 \
            _Pragma("unroll") for (int r = 0; r < 16; ++r)                                        \
                rv_[r] = rb[(long)(rowu + (MT_) * 32 + (r & 3) + 8 * (r >> 2)) * a.T + li + (NT_) * 32]; \
        }                                                                                         \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                          \
            float v_ = acc[MT_][NT_][r] + bia[MT_][r];                                            \
            if (rb) v_ += rv_[r];                                                                 \
            yb[(long)(rowu + (MT_) * 32 + (r & 3) + 8 * (r >> 2)) * a.T + li + (NT_) * 32] = v_;  \
        }                                                                                         \
    }
#define EPS_COL(NT_)                                                                              \
    if (n0 + wn * (NW * 32) + (NT_) * 32 + nl < a.T) { EPS(0, NT_) EPS(1, NT_) EPS(2, NT_) EPS(3, NT_) EPS(4, NT_) }
        EPS_COL(0) EPS_COL(1) EPS_COL(2)
#undef EPS_COL
#undef EPS
    } else {
#define EPI(MT_, NT_)                                                                             \
    {                                                                                             \
        float v_[16];                                                                             \
        conv_ep_store_block(a, acc[MT_][NT_], ep, WBM, b, m0, wm * (MW * 32) + (MT_) * 32 + 4 * h, \
                            n0 + wn * (NW * 32) + (NT_) * 32 + nl, v_);                           \
    }
#define EPI_ROW(MT_) EPI(MT_, 0) EPI(MT_, 1) EPI(MT_, 2)
        EPI_ROW(0) EPI_ROW(1) EPI_ROW(2) EPI_ROW(3) EPI_ROW(4)
#undef EPI_ROW
#undef EPI
    }
#ifdef WIDE_PROFILE
    {
        const long long issued = clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (a.stats && blockIdx.x == 0 && lane == 0) {
            a.stats[wave * 8 + 6] = (float)(issued - pmain_end);        // epilogue instructions issued
            a.stats[wave * 8 + 7] = (float)(clock64() - pmain_end);     // ... and stores drained
        }
    }
#endif
}

template <int KS>
static int launch_conv_nn_x3w(ConvNNArgs a, hipStream_t stream) {
    const size_t lds = (size_t)(3 * WASLAB + 2 * WXSLAB) * 16;
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
